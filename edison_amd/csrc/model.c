/*
 * model.c -- host-side (plain C) loader of the int8 CNN parameters.
 *
 * Input: an .ednn blob (tools/import_weights_h.py) holding the numbers of an NNoM weights.h -- for the
 * reference that is firmware/src/ai/nnom/kws_nnom/weights.h (arrays :3-45, shifts :50-105, graph :138-161).
 * The two fast kernels are specialised for that graph (kws_conv: conv5x5x16 / pool(2,1) / conv3x3x32 / pool(2,1) /
 * conv3x3x64 / conv3x3x32 / dense10 / softmax on a 31x13x1 input); this parser answers EDISON_E_SIZE for any other
 * topology, and the caller (edison_model_load_mem) then serves the model with the general plan of model_net.c alone.
 * Weight VALUES and all shifts are free (a retrained model loads).
 *
 * Output: ed_cnn_model_t -- weights as dwords [K/4][out_channel] (OHWI order inside K), accumulator seeds
 * (bias << bias_lshift) + NN_ROUND(out_rshift) precomputed (arm_convolve_HWC_q7_basic_nonsquare.c:196).
 */
#include <stdio.h>
#include <string.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"
#include "cnn_mfma_cols.h"

enum { T_CONV = 1, T_POOL = 2, T_DENSE = 3, T_SOFTMAX = 4 };

typedef struct { int32_t v[12]; } rec_t;

static int fail(char *err, size_t cap, const char *msg)
{
	if (err && cap) snprintf(err, cap, "%s", msg);
	return EDISON_E_SIZE;
}

static int32_t seed(int8_t b, int bl, int rs) { return (int32_t)((uint32_t)(int32_t)b << bl) + (int32_t)((1u << rs) >> 1); }

/* pack int8 weights w[O][K] (K padded to K4*4) into dwords out[K4][OS] (OS >= O, extra columns zero) */
static void pack(const int8_t *w, int O, int K, int K4, int OS, int32_t *out)
{
	memset(out, 0, sizeof(int32_t) * (size_t)K4 * OS);
	for (int o = 0; o < O; o++)
		for (int k = 0; k < K; k++)
		{
			uint32_t byte = (uint32_t)(uint8_t)w[(size_t)o * K + k];
			out[(size_t)(k / 4) * OS + o] |= (int32_t)(byte << (8 * (k % 4)));
		}
}

/* Which logical row of a 32-row tile sits in physical row p of the MFMA's A operand. The D operand of v_mfma_i32_32x32x32_i8 gives
 * lane half h the physical rows 8 g + 4 h + j (register group g = 0..3, j = 0..3): with logical row 16 h + 4 g + j there, the
 * four packed dwords of a lane are 16 CONSECUTIVE rows (output channels) of its column -- one ds_write_b128 as it stands. (With
 * rows in natural order a lane held 4-byte pieces 8 bytes apart and needed two v_permlane32_swap per 16 bytes to trade them with
 * the other lane half: 38 swaps of 8 cycles per group of four utterances, in a kernel bound by vector issue.) */
static int tile_row(int p) { return 16 * ((p >> 2) & 1) + 4 * (p >> 3) + (p & 3); }

/* Fill one MFMA A-operand fragment: lane l, byte j  <-  a(row0 + tile_row(l & 31), 32*kstep + 16*(l >> 5) + j). */
typedef int8_t (*a_elem_fn)(const int8_t *w, int row, int k);

static void fill_frag(int8_t *frag, const int8_t *w, a_elem_fn a, int row0, int kstep)
{
	for (int l = 0; l < 64; l++)
		for (int j = 0; j < 16; j++)
			frag[l * 16 + j] = a(w, row0 + tile_row(l & 31), 32 * kstep + 16 * (l >> 5) + j);
}

/* conv1 as Toeplitz: row = x*16 + o, k = ky*16 + xx; w1 is OHWI [16][5][5][1] */
static int8_t a1_elem(const int8_t *w, int row, int k)
{
	const int x = row >> 4, o = row & 15, ky = k >> 4, xx = k & 15;
	if (x >= 9 || ky >= 5 || xx - x < 0 || xx - x >= 5) return 0;
	return w[(o * 5 + ky) * 5 + (xx - x)];
}
static int8_t a2_elem(const int8_t *w, int row, int k) { return k < 144 ? w[row * 144 + k] : 0; } /* k = tap*16+ci */
static int8_t a3_elem(const int8_t *w, int row, int k) { return w[row * 288 + k]; }                /* k = tap*32+ci */
static int8_t a4_elem(const int8_t *w, int row, int k) { return w[row * 576 + k]; }                /* k = tap*64+ci */
static int8_t afc_elem(const int8_t *w, int row, int k) { return row < ED_FC_O && k < ED_FC_I ? w[row * ED_FC_I + k] : 0; }

/* One A-operand fragment of v_mfma_i32_16x16x64_i8: lane l, byte j  <-  a(row0 + (l & 15), 64*kstep + 16*(l >> 4) + j). */
static void fill_frag16(int8_t *frag, const int8_t *w, a_elem_fn a, int row0, int kstep)
{
	for (int l = 0; l < 64; l++)
		for (int j = 0; j < 16; j++)
			frag[l * 16 + j] = a(w, row0 + (l & 15), 64 * kstep + 16 * (l >> 4) + j);
}

int ed_parse_model(const void *blob, size_t blob_bytes, ed_cnn_model_t *out, ed_cnn_mfma_model_t *out_mfma, char *err,
                   size_t err_cap)
{
	const unsigned char *p = (const unsigned char *)blob;
	if (blob == NULL || blob_bytes < 40 || memcmp(p, "EDNNOM1\0", 8) != 0) return fail(err, err_cap, "not an .ednn model blob");
	int32_t head[8];
	memcpy(head, p + 8, sizeof(head));
	const int in_h = head[0], in_w = head[1], in_c = head[2], n_layers = head[3], payload_bytes = head[4];
	if (n_layers < 1 || n_layers > 64 || payload_bytes < 0 ||
	    blob_bytes < 40 + (size_t)n_layers * 48 + (size_t)payload_bytes)
		return fail(err, err_cap, "truncated .ednn model blob");
	const int8_t *payload = (const int8_t *)(p + 40 + (size_t)n_layers * 48);

	/* expected topology: type, out_ch, kh, kw, (stride 1 conv / stride == kernel pool), in_ch */
	static const int expect[8][6] = {
		{T_CONV, ED_C1_O, 5, 5, 1, 1}, {T_POOL, 0, 2, 1, 2, 1}, {T_CONV, ED_C2_O, 3, 3, 1, 1}, {T_POOL, 0, 2, 1, 2, 1},
		{T_CONV, ED_C3_O, 3, 3, 1, 1}, {T_CONV, ED_C4_O, 3, 3, 1, 1}, {T_DENSE, ED_FC_O, 0, 0, 0, 0}, {T_SOFTMAX, 0, 0, 0, 0, 0}};
	if (in_h != ED_IN_H || in_w != ED_IN_W || in_c != 1 || n_layers != 8)
		return fail(err, err_cap, "model is not the 31x13x1 kws_conv graph this path accelerates");
	rec_t r[8];
	memcpy(r, p + 40, sizeof(r));
	for (int i = 0; i < 8; i++)
	{
		const int32_t *v = r[i].v;
		int ok = v[0] == expect[i][0];
		if (ok && v[0] == T_CONV) ok = v[1] == expect[i][1] && v[2] == expect[i][2] && v[3] == expect[i][3] && v[4] == 1 && v[5] == 1 && v[8] == 1;
		/* v[8]: bit 0 = ReLU tail, bit 1 = PADDING_SAME. The matrix-core kernel implements valid pooling (27 -> 13 rows)
		 * and a dense layer without ReLU only; anything else falls through to the general plan (model_net.c) */
		if (ok && v[0] == T_POOL) ok = v[2] == 2 && v[3] == 1 && v[4] == 2 && v[5] == 1 && v[8] == 0;
		if (ok && v[0] == T_DENSE) ok = v[1] == ED_FC_O && v[11] == ED_FC_I && v[8] == 0;
		if (ok && (v[0] == T_CONV || v[0] == T_DENSE))
			ok = v[6] >= 0 && v[6] < 24 && v[7] >= 0 && v[7] < 31 && v[9] >= 0 && v[10] >= 0 && v[9] < payload_bytes && v[10] < payload_bytes;
		if (!ok) return fail(err, err_cap, "model layer list does not match the kws_conv graph (weights.h:138-161)");
	}
	static const int cin[8] = {1, 0, ED_C1_O, 0, ED_C2_O, ED_C3_O, 0, 0};
	for (int i = 0; i < 8; i++)
		if (r[i].v[0] == T_CONV)
		{
			if (r[i].v[11] != cin[i]) return fail(err, err_cap, "conv input channel count mismatch");
			size_t need = (size_t)r[i].v[1] * r[i].v[2] * r[i].v[3] * cin[i];
			if ((size_t)r[i].v[9] + need > (size_t)payload_bytes || (size_t)r[i].v[10] + (size_t)r[i].v[1] > (size_t)payload_bytes)
				return fail(err, err_cap, "weight tensor outside the payload");
		}
	if ((size_t)r[6].v[9] + ED_FC_O * ED_FC_I > (size_t)payload_bytes || (size_t)r[6].v[10] + ED_FC_O > (size_t)payload_bytes)
		return fail(err, err_cap, "dense tensor outside the payload");

	memset(out, 0, sizeof(*out));
	pack(payload + r[0].v[9], ED_C1_O, 25, 7, ED_C1_O, &out->w1[0][0]);
	pack(payload + r[2].v[9], ED_C2_O, 144, 36, ED_C2_O, &out->w2[0][0]);
	pack(payload + r[4].v[9], ED_C3_O, 288, 72, ED_C3_O, &out->w3[0][0]);
	pack(payload + r[5].v[9], ED_C4_O, 576, 144, ED_C4_O, &out->w4[0][0]);
	pack(payload + r[6].v[9], ED_FC_O, ED_FC_I, 24, 16, &out->wfc[0][0]);
	for (int o = 0; o < ED_C1_O; o++) out->b1[o] = seed(payload[r[0].v[10] + o], r[0].v[6], r[0].v[7]);
	for (int o = 0; o < ED_C2_O; o++) out->b2[o] = seed(payload[r[2].v[10] + o], r[2].v[6], r[2].v[7]);
	for (int o = 0; o < ED_C3_O; o++) out->b3[o] = seed(payload[r[4].v[10] + o], r[4].v[6], r[4].v[7]);
	for (int o = 0; o < ED_C4_O; o++) out->b4[o] = seed(payload[r[5].v[10] + o], r[5].v[6], r[5].v[7]);
	for (int o = 0; o < ED_FC_O; o++) out->bfc[o] = seed(payload[r[6].v[10] + o], r[6].v[6], r[6].v[7]);
	out->rs1 = r[0].v[7]; out->rs2 = r[2].v[7]; out->rs3 = r[4].v[7]; out->rs4 = r[5].v[7]; out->rsfc = r[6].v[7];

	if (out_mfma)
	{
		ed_cnn_mfma_model_t *m = out_mfma;
		memset(m, 0, sizeof(*m));
		for (int rt = 0; rt < 5; rt++)
			for (int s = 0; s < 3; s++) fill_frag(m->a1[rt * 3 + s], payload + r[0].v[9], a1_elem, 32 * rt, s);
		for (int s = 0; s < 5; s++) fill_frag(m->a2[s], payload + r[2].v[9], a2_elem, 0, s);
		for (int rt = 0; rt < 2; rt++)
			for (int s = 0; s < 9; s++) fill_frag(m->a3[rt * 9 + s], payload + r[4].v[9], a3_elem, 32 * rt, s);
		for (int rt = 0; rt < 2; rt++)
			for (int s = 0; s < 9; s++) fill_frag16(m->a4[rt * 9 + s], payload + r[5].v[9], a4_elem, 16 * rt, s);
		for (int s = 0; s < 2; s++) fill_frag16(m->afc[s], payload + r[6].v[9], afc_elem, 0, s);
		/* the accumulator seeds of the 32-row tiles in PHYSICAL row order (conv1: a lane half owns all 16 channels of one x
		 * position, so its seeds are the 16 channels in order whatever the tile) */
		memcpy(m->b1, out->b1, sizeof(m->b1));
		for (int p_ = 0; p_ < 32; p_++) m->b2[p_] = out->b2[tile_row(p_)];
		for (int p_ = 0; p_ < 64; p_++) m->b3[p_] = out->b3[32 * (p_ >> 5) + tile_row(p_ & 31)];
		memcpy(m->b4, out->b4, sizeof(m->b4));
		memcpy(m->bfc, out->bfc, sizeof(m->bfc));
		m->rs1 = out->rs1; m->rs2 = out->rs2; m->rs3 = out->rs3; m->rs4 = out->rs4; m->rsfc = out->rsfc;
		/* the column tables of a full group (edison_internal.h): byte offsets of the column's operand reads and of its outputs;
		 * an idle lane re-reads the last live column of its tile */
		for (int t = 0; t < 2; t++)
		{
			uint32_t last = 0;
			for (int pass = 0; pass < 2; pass++) /* pass 0 finds the tile's last live column, pass 1 fills */
				for (int c = 0; c < 32; c++)
				{
					const int q = ED_CNN_COLS_CONV1[t][c];
					if (q >= 0)
					{
						const int u = q / 13, py = q % 13;
						last = (uint32_t)(u * EDM_UTT + py * 16) | ((uint32_t)(u * EDM_UTT + EDM_REGA + py * 9 * 16) << 16);
						if (pass) m->cols1[t][c] = last;
					}
					else if (pass) m->cols1[t][c] = (last & 0x7fffu) | ED_CNN_COL_IDLE;
				}
		}
		for (int t = 0; t < 5; t++)
		{
			uint32_t last = 0;
			for (int pass = 0; pass < 2; pass++)
				for (int c = 0; c < 32; c++)
				{
					const int q = ED_CNN_COLS_CONV2[t][c];
					if (q >= 0)
					{
						const int u = q / 35, r_ = q % 35, py = r_ / 7, x = r_ % 7;
						last = (uint32_t)(u * EDM_UTT + EDM_REGA + ((2 * py) * 9 + x) * 16) | ((uint32_t)(u * EDM_UTT + (py * 7 + x) * 16) << 16);
						if (pass) m->cols2[t][c] = last;
					}
					else if (pass) m->cols2[t][c] = (last & 0x7fffu) | ED_CNN_COL_IDLE;
				}
		}
		for (int t = 0; t < 2; t++)
		{
			uint32_t last = 0;
			for (int pass = 0; pass < 2; pass++)
				for (int c = 0; c < 32; c++)
				{
					const int q = ED_CNN_COLS_CONV3[t][c];
					if (q >= 0)
					{
						const int u = q / 15, r_ = q % 15, y = r_ / 5, x = r_ % 5;
						last = (uint32_t)(u * EDM_UTT + (y * 7 + x) * 16) | ((uint32_t)(u * EDM_UTT + EDM_REGA + (y * 5 + x) * 16) << 16);
						if (pass) m->cols3[t][c] = last;
					}
					else if (pass) m->cols3[t][c] = (last & 0x7fffu) | ED_CNN_COL_IDLE;
				}
		}
	}
	return EDISON_OK;
}
