/*
 * model_net_mm.c -- host-side planner of the matrix-core path for ANY sequential NNoM int8 graph (the layer list
 * ed_plan_net built): operand-fragment packing of every Conv2D / Dense layer, the padded activation layouts and the
 * LDS budget. The arithmetic it feeds is the reference's (arm_convolve_HWC_q7_basic_nonsquare.c:188-221,
 * arm_fully_connected_q7_opt.c:374-473: out = sat8((sum x*w + (bias << BL) + NN_ROUND(RS)) >> RS)); only the order of
 * the exact integer sums differs. See edison_internal.h (ed_mm_plan_t) for the scheme.
 */
#include <stdlib.h>
#include <string.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

typedef struct { int32_t v[12]; } rec_t;

static int imax(int a, int b) { return a > b ? a : b; }
static int up16(int v) { return (v + 15) & ~15; }

int ed_plan_net_mm(const void *blob, size_t blob_bytes, const ed_net_plan_t *plan, ed_mm_plan_t *mm, int8_t **frag, int32_t **seeds)
{
	const unsigned char *p = (const unsigned char *)blob;
	memset(mm, 0, sizeof(*mm));
	*frag = NULL;
	*seeds = NULL;
	if (blob_bytes < 40) return EDISON_E_SIZE;
	const int n_layers = plan->n_layers;
	const int8_t *payload = (const int8_t *)(p + 40 + (size_t)n_layers * 48);

	/* pass 1: shapes, counts */
	size_t frag_bytes = 0;
	int n_seeds = 0, n_koff = 0;
	for (int i = 0; i < n_layers; i++)
	{
		const ed_net_layer_t *L = &plan->L[i];
		ed_mm_layer_t *M = &mm->L[i];
		/* default: the layer reads a compact (unpadded) image */
		M->in_hp = L->in_h; M->in_wp = L->in_w; M->in_py = 0; M->in_px = 0;
		M->pp = L->type == ED_NET_DENSE ? L->in_n : L->in_c;
		/* (a dense layer reads the same compact HWC image: it is a 1x1 convolution over ONE row of in_n bytes) */
		if (L->type != ED_NET_CONV && L->type != ED_NET_DENSE)
		{
			M->in_img = up16(L->in_n) + 16;
			continue;
		}
		const int dense = L->type == ED_NET_DENSE;
		const int in_c = dense ? L->in_n : L->in_c, kh = dense ? 1 : L->kh, kw = dense ? 1 : L->kw;
		const int sh = dense ? 1 : L->sh, out_w = dense ? 1 : L->out_w, out_h = dense ? 1 : L->out_h;
		if (!dense)
		{
			/* zero border so that every window lies inside the stored image: pad rows / columns in front
			 * (nnom_conv2d.c:66-71) and whatever the last windows overhang behind */
			M->in_py = L->pad_h; M->in_px = L->pad_w;
			M->in_hp = imax(L->in_h + L->pad_h, (out_h - 1) * L->sh + kh);
			M->in_wp = imax(L->in_w + L->pad_w, (out_w - 1) * L->sw + kw);
		}
		const int seg = kw * in_c;                 /* contiguous bytes under one kernel row */
		M->mm = 1;
		M->cpr = (seg + 15) / 16;
		M->n_ks = (kh * M->cpr + 1) / 2;
		M->n_rt = (L->out_c + 31) / 32;
		M->expand = (in_c % 16) != 0;
		/* Row-Toeplitz form of a convolution over a NARROW image with few channels (a spectrogram: C_in = 1): the GEMM's rows
		 * are (output x, output channel) pairs, its k runs over (kernel row, byte of the WHOLE zero-padded input row, 16-byte
		 * aligned by construction), its columns are the output rows. A[(x, oc)][ky][q] = w[oc][ky][q / C_in - x * sw][q % C_in]
		 * where that tap exists, else 0. No expanded copy of the input (the expansion pass and its LDS buffer -- 4.4 KB per
		 * input on kws_conv, more than both activation buffers together -- disappear), and the output row of a column is
		 * contiguous in the consumer's HWC layout, so the epilogue is the ordinary one with out_w * C_out "channels".
		 * Taken when the plain form would need the expansion, the padded row fits four chunks and the matrix-core work does
		 * not grow by more than half. EDISON_NET_NO_TOEPLITZ=1: A/B knob. */
		if (!dense && M->expand)
		{
			const char *env = getenv("EDISON_NET_NO_TOEPLITZ");
			const int rowb = up16(M->in_wp * in_c), rows = L->out_w * L->out_c;
			const int64_t plain = (int64_t)M->n_rt * ((out_h * out_w + 31) / 32) * M->n_ks;
			const int64_t toep = (int64_t)((rows + 31) / 32) * ((out_h + 31) / 32) * ((kh * (rowb / 16) + 1) / 2);
			if (!(env && atoi(env)) && rowb <= 64 && rowb % in_c == 0 && rows <= 1024 && 2 * toep <= 3 * plain)
			{
				M->toep = 1;
				M->in_wp = rowb / in_c;               /* zero columns behind the image up to the aligned row */
				M->cpr = rowb / 16;
				M->n_ks = (kh * M->cpr + 1) / 2;
				M->n_rt = (rows + 31) / 32;
				M->expand = 0;
			}
		}
		M->in_img = (dense ? up16(L->in_n) : up16(M->in_hp * M->in_wp * in_c)) + 16; /* dense: the compact image, C_in = its whole length */
		if (M->toep)
		{
			M->pitch_x = 0;                           /* one GEMM column per output ROW */
			M->pitch_y = M->in_wp * in_c;
			M->x_img = 0;
		}
		else if (M->expand)
		{
			M->pitch_x = 16 * M->cpr;
			M->pitch_y = out_w * 16 * M->cpr;
			M->x_img = (dense ? 1 : M->in_hp) * M->pitch_y + 16;
		}
		else
		{
			M->pitch_x = (dense ? 1 : L->sw) * in_c;
			M->pitch_y = M->in_wp * in_c;
			M->x_img = 0;
		}
		(void)sh;
		/* the small-tile form of the same layer (chosen in pass 2, once the per-wave batch is known): room for either */
		M->n_ks16 = (kh * M->cpr + 3) / 4;
		M->n_rt16 = ((M->toep ? L->out_w * L->out_c : L->out_c) + 15) / 16;
		M->frag_off = (int32_t)frag_bytes;
		frag_bytes += (size_t)imax(M->n_rt * M->n_ks, M->n_rt16 * M->n_ks16) * 1024;
		M->seed_off = n_seeds;
		n_seeds += 32 * M->n_rt;
		M->koff_off = n_koff;
		n_koff += imax(2 * M->n_ks, 4 * M->n_ks16);
		if (n_koff > ED_MM_MAX_KOFF || 2 * M->n_ks > 256 || frag_bytes > ((size_t)64 << 20)) return EDISON_OK; /* mm->ok stays 0 */
	}

	/* MaxPool right behind a matrix-core layer, window = stride, no padding, at most 4 positions: fused into that layer */
	for (int i = 0; i + 1 < n_layers; i++)
	{
		const ed_net_layer_t *C = &plan->L[i], *Q = &plan->L[i + 1];
		if (!mm->L[i].mm || C->type != ED_NET_CONV || Q->type != ED_NET_POOL) continue;
		if (Q->pad_h || Q->pad_w || Q->check_taps || Q->kh != Q->sh || Q->kw != Q->sw || (Q->kh * Q->kw != 2 && Q->kh * Q->kw != 4)) continue;
		if (Q->out_h * Q->kh > C->out_h || Q->out_w * Q->kw > C->out_w) continue;
		if (mm->L[i].toep && (Q->kw != 1 || Q->out_w != C->out_w)) continue; /* x lives in the GEMM's rows there: only windows along y fuse */
		mm->L[i].pool_h = Q->kh; mm->L[i].pool_w = Q->kw;
		mm->L[i + 1].skip = 1;
		mm->L[i + 1].in_img = 0; /* the unpooled tensor never exists */
	}

	/* LDS banks. A B fragment is 16 bytes per lane and neighbouring lanes are neighbouring pixels; the epilogue of the layer in
	 * front stores one dword per lane, neighbouring lanes again neighbouring pixels. With C_in = 32 or 64 bytes per pixel the 16
	 * lanes of a read phase fall on every second / fourth 16-byte bank group and the stores on 8 / 4 of the 64 banks (the own
	 * kernel of kws_conv: LDS pipe 86 % busy, 43 % of it conflicts, profiles/r03_net_counters.txt). One 16-byte gap behind every
	 * pixel whose C_in is an EVEN multiple of 16 makes the pixel pitch an odd multiple: consecutive pixels then walk through all
	 * 16 bank groups. Only the tables change (chunk offsets, column offsets, the producer's output pitch); the gaps are never
	 * read. Not for layers that read an expanded copy or a Toeplitz row, not behind a Toeplitz producer (it stores whole rows),
	 * not for Dense (one column per image: nothing to conflict with). EDISON_NET_NO_PIXEL_GAP=1: A/B knob. */
	{
		const char *env = getenv("EDISON_NET_NO_PIXEL_GAP");
		for (int i = 0; i < n_layers && !(env && atoi(env)); i++)
		{
			const ed_net_layer_t *L = &plan->L[i];
			ed_mm_layer_t *M = &mm->L[i];
			if (!M->mm || L->type != ED_NET_CONV || M->expand || M->toep || (L->in_c % 32) != 0) continue;
			int j = i - 1; /* the producer: the layer in front, or the one in front of a fused MaxPool */
			if (j >= 0 && mm->L[j].skip) j--;
			if (j >= 0 && mm->L[j].toep) continue;
			if (j < 0 && !(plan->in_n <= ED_MM_MAX_INTAB)) continue; /* the input stage places pixels by table only */
			M->pp = L->in_c + 16;
			M->pitch_x = L->sw * M->pp;
			M->pitch_y = M->in_wp * M->pp;
			M->in_img = up16(M->in_hp * M->in_wp * M->pp) + 16;
		}
	}

	/* column tables: (B offset, output offset) of every stored pixel, for the layers that fit ED_MM_MAX_COLS together */
	int n_cols = 0;
	for (int i = 0; i < n_layers; i++)
	{
		const ed_net_layer_t *L = &plan->L[i];
		ed_mm_layer_t *M = &mm->L[i];
		M->col_off = -1;
		if (!M->mm) continue;
		const int dense = L->type == ED_NET_DENSE, fused = M->pool_h > 0, nx = fused ? i + 2 : i + 1;
		const int st_h = dense ? 1 : (fused ? plan->L[i + 1].out_h : L->out_h), st_w = dense || M->toep ? 1 : (fused ? plan->L[i + 1].out_w : L->out_w);
		if (n_cols + st_h * st_w > ED_MM_MAX_COLS) continue;
		const int ph = fused ? M->pool_h : 1, pw = fused ? M->pool_w : 1, sh = dense ? 1 : L->sh;
		int owp, opy, opx, opp; /* the consumer's layout (the last layer's output is compact) */
		if (nx < n_layers) { owp = mm->L[nx].in_wp; opy = mm->L[nx].in_py; opx = mm->L[nx].in_px; opp = plan->L[nx].type == ED_NET_CONV ? mm->L[nx].pp : L->out_c; }
		else { owp = fused ? plan->L[i + 1].out_w : L->out_w; opy = 0; opx = 0; opp = L->out_c; }
		M->col_off = n_cols;
		for (int y = 0; y < st_h; y++)
			for (int x = 0; x < st_w; x++)
			{
				mm->coltab[2 * n_cols] = (y * ph * sh) * M->pitch_y + (x * pw) * M->pitch_x;
				mm->coltab[2 * n_cols + 1] = ((y + opy) * owp + x + opx) * opp;
				n_cols++;
			}
	}
	mm->n_cols = n_cols;

	/* expansion tables and the input table, under the same rule (what does not fit is worked out by the kernel) */
	int n_xtab = 0;
	for (int i = 0; i < n_layers; i++)
	{
		const ed_net_layer_t *L = &plan->L[i];
		ed_mm_layer_t *M = &mm->L[i];
		M->xtab_off = -1;
		if (!M->mm || !M->expand) continue;
		const int dense = L->type == ED_NET_DENSE;
		const int in_c = dense ? L->in_n : L->in_c, seg = (dense ? 1 : L->kw) * in_c, sw = dense ? 1 : L->sw, out_w = dense ? 1 : L->out_w;
		const int rows = dense ? 1 : M->in_hp, rec = rows * out_w * M->cpr;
		if (n_xtab + rec > ED_MM_MAX_XTAB || M->in_img >= (1 << 24)) continue;
		M->xtab_off = n_xtab;
		for (int r = 0; r < rows; r++)
			for (int xo = 0; xo < out_w; xo++)
				for (int j = 0; j < M->cpr; j++)
				{
					const int keep = seg - 16 * j < 16 ? seg - 16 * j : 16;
					mm->xtab[2 * n_xtab] = ((r * M->in_wp + xo * sw) * in_c + 16 * j) | (keep << 24);
					mm->xtab[2 * n_xtab + 1] = r * M->pitch_y + xo * M->pitch_x + 16 * j;
					n_xtab++;
				}
	}
	mm->n_xtab = n_xtab;
	mm->n_intab = 0;
	if (plan->in_n <= ED_MM_MAX_INTAB && mm->L[0].in_img <= 65536)
	{
		const ed_mm_layer_t *M0 = &mm->L[0];
		for (int e = 0; e < plan->in_n; e++)
		{
			const int pix = e / plan->in_c, c = e - pix * plan->in_c, y = pix / plan->in_w, x = pix - y * plan->in_w;
			mm->intab[e] = (uint16_t)(((y + M0->in_py) * M0->in_wp + x + M0->in_px) * (plan->L[0].type == ED_NET_CONV ? M0->pp : plan->in_c) + c);
		}
		mm->n_intab = plan->in_n;
		if (plan->in_n >= 4 && plan->in_n < ED_MM_INTAB_PAD) /* see ED_MM_INTAB_PAD */
		{
			for (int e = plan->in_n; e < ED_MM_INTAB_PAD; e++) mm->intab[e] = (uint16_t)(M0->in_img - 1);
			mm->n_intab = ED_MM_INTAB_PAD;
		}
	}

	/* LDS budget: a wave's activation region for `batch` inputs -- a layer's input images at one end, its output images at
	 * the other, the next layer the other way round: max(in + out) over the layers that run, not twice the largest image --
	 * the expansion buffer, the koff table */
	int max_pair = 0, max_x = 0;
	for (int i = 0; i < n_layers; i++)
	{
		if (mm->L[i].x_img > max_x) max_x = mm->L[i].x_img;
		if (mm->L[i].skip) continue;
		const int nx = mm->L[i].pool_h > 0 ? i + 2 : i + 1;
		const int st_h = mm->L[i].pool_h > 0 ? plan->L[i + 1].out_h : plan->L[i].out_h, st_w = mm->L[i].pool_h > 0 ? plan->L[i + 1].out_w : plan->L[i].out_w;
		const int out_img = nx < n_layers ? mm->L[nx].in_img : up16(st_h * st_w * plan->L[i].out_c) + 16; /* the last layer's output is compact */
		if (mm->L[i].in_img + out_img > max_pair) max_pair = mm->L[i].in_img + out_img;
	}
	const int max_img = (max_pair / 2 + 15) & ~15; /* half of the region per input */
	/* Where the weight fragments live: (2) ALL layers resident in LDS for the whole launch, shared by the waves of the
	 * workgroup (one L2 read per workgroup), or (0) streamed from L2 per MFMA. Every wave gets its own activation slice
	 * (two ping-pong buffers + the expansion buffer for `batch` inputs); as many waves as fit, at most 12, at least 4; the
	 * per-wave batch grows (up to 4) only while 8 waves still fit -- independent waves hide each other's latencies, a
	 * bigger batch only fills the 32-column tiles of small late layers better. */
	const int lds_cap = 156 * 1024; /* of the CU's 160 KB */
	const int tbl = up16(4 * n_koff) + up16(4 * n_seeds) + up16(8 * n_cols) + up16(8 * n_xtab) + up16(2 * mm->n_intab);
	if (tbl > 24 * 1024) return EDISON_OK;
	int batch = 0, waves = 0, frag_lds = 0, frag_mode = 0;
	/* EDISON_NET_BATCH=b / EDISON_NET_MIN_WAVES=w: A/B knobs (tools/bench_net.py): force the per-wave batch, relax the wave floor */
	const char *env_b = getenv("EDISON_NET_BATCH"), *env_w = getenv("EDISON_NET_MIN_WAVES");
	const int force_b = env_b ? atoi(env_b) : 0, min_w = env_w ? atoi(env_w) : 0;
	for (int mode = 2; mode >= 0 && !batch; mode -= 2)
	{
		const int64_t fl = mode == 2 ? (int64_t)frag_bytes : 0;
		if (fl > 96 * 1024) continue;
		for (int b = 4; b >= 1 && !batch; b >>= 1)
		{
			if (force_b > 0 && b != force_b) continue;
			const int64_t per_wave = 2 * (int64_t)b * max_img + (int64_t)b * up16(max_x);
			int64_t w = (lds_cap - tbl - fl) / per_wave;
			if (w > 12) w = 12; /* the kernel is built for 768 threads: 168 VGPRs a wave */
			if (w >= (min_w > 0 ? min_w : (b > 1 ? 8 : 4))) { batch = b; waves = (int)w; frag_lds = (int)fl; frag_mode = mode; }
		}
	}
	if (!batch) return EDISON_OK;
	mm->batch = batch;
	mm->waves = waves;
	mm->buf_bytes = batch * max_img;
	mm->x_bytes = batch * up16(max_x);
	mm->frag_lds = frag_lds;
	mm->frag_mode = frag_mode;
	mm->tbl_bytes = tbl;
	mm->lds_bytes = tbl + frag_lds + waves * (2 * mm->buf_bytes + mm->x_bytes);
	mm->frag_bytes = (int32_t)frag_bytes;
	mm->n_seeds = n_seeds;
	mm->n_koff = n_koff;

	/* layers whose columns (stored pixels x per-wave batch) fit 16 and that fuse no MaxPool run on 16 x 16 x 64 tiles: a
	 * quarter of the accumulator registers to requantise per tile and half the k-steps, where a 32-column tile would be
	 * mostly padding (the late layers of a classifier: 3 pixels, 1 pixel). EDISON_NET_NO_SMALL_TILES=1: A/B knob. */
	{
		const char *env = getenv("EDISON_NET_NO_SMALL_TILES");
		const int no_small = env && atoi(env);
		for (int i = 0; i < n_layers; i++)
		{
			const ed_net_layer_t *L = &plan->L[i];
			ed_mm_layer_t *M = &mm->L[i];
			if (!M->mm) continue;
			const int dense = L->type == ED_NET_DENSE;
			const int pix = dense ? 1 : L->out_h * (M->toep ? 1 : L->out_w);
			M->small = !no_small && M->pool_h == 0 && batch * pix <= 16;
		}
		/* now that every layer's form is known: the exact fragment layout (pass 1 reserved room for either form), and
		 * the waves that fit beside it */
		size_t fexact = 0;
		for (int i = 0; i < n_layers; i++)
		{
			ed_mm_layer_t *M = &mm->L[i];
			if (!M->mm) continue;
			M->frag_off = (int32_t)fexact;
			fexact += (size_t)(M->small ? M->n_rt16 * M->n_ks16 : M->n_rt * M->n_ks) * 1024;
		}
		frag_bytes = fexact;
		if (frag_mode == 2) frag_lds = (int)fexact;
		{
			const int64_t per_wave = 2 * (int64_t)batch * max_img + (int64_t)batch * up16(max_x);
			int64_t w = (lds_cap - tbl - frag_lds) / per_wave;
			if (w > 12) w = 12;
			if (w > waves) waves = (int)w;
		}
		mm->waves = waves;
		mm->frag_lds = frag_lds;
		mm->lds_bytes = tbl + frag_lds + waves * (2 * mm->buf_bytes + mm->x_bytes);
		mm->frag_bytes = (int32_t)frag_bytes;
	}

	/* pass 2: fragments, seeds, chunk offsets */
	int8_t *fb = (int8_t *)calloc(frag_bytes + 16, 1);
	int32_t *sb = (int32_t *)calloc((size_t)n_seeds + 4, sizeof(int32_t));
	if (!fb || !sb) { free(fb); free(sb); return EDISON_E_NO_MEMORY; }
	int64_t acc_bound[ED_NET_MAX_LAYERS]; /* the largest |accumulator| a layer can reach, for the choice of its requantisation (ED_RUN_RS_HI) */
	memset(acc_bound, 0, sizeof(acc_bound));
	for (int i = 0; i < n_layers; i++)
	{
		const ed_net_layer_t *L = &plan->L[i];
		const ed_mm_layer_t *M = &mm->L[i];
		if (!M->mm) continue;
		rec_t r;
		memcpy(&r, p + 40 + (size_t)i * 48, sizeof(r));
		const int dense = L->type == ED_NET_DENSE;
		const int in_c = dense ? L->in_n : L->in_c, kh = dense ? 1 : L->kh, kw = dense ? 1 : L->kw;
		const int8_t *w = payload + r.v[9], *bias = payload + r.v[10]; /* OHWI / [out][in]: row o = kh segments of seg bytes */
		int8_t *wt = NULL;
		int seg = kw * in_c, n_rows = L->out_c;
		for (int o = 0; o < L->out_c; o++) /* |sum x w + seed| <= 128 sum |w| + |seed| */
		{
			int64_t sw = 0;
			for (int q = 0; q < kh * seg; q++) sw += w[(size_t)o * kh * seg + q] < 0 ? -(int64_t)w[(size_t)o * kh * seg + q] : w[(size_t)o * kh * seg + q];
			const int64_t sd = (int64_t)(int32_t)((uint32_t)(int32_t)bias[o] << r.v[6]) + (int64_t)((1u << r.v[7]) >> 1);
			const int64_t bd = 128 * sw + (sd < 0 ? -sd : sd);
			if (bd > acc_bound[i]) acc_bound[i] = bd;
		}
		if (M->toep)
		{
			/* the Toeplitz matrix, in the layout the packing below expects: [row = x * C_out + oc][ky][q < rowb] */
			const int rowb = M->in_wp * in_c;
			n_rows = L->out_w * L->out_c;
			wt = (int8_t *)calloc((size_t)n_rows * kh * rowb, 1);
			if (!wt) { free(fb); free(sb); return EDISON_E_NO_MEMORY; }
			for (int x = 0; x < L->out_w; x++)
				for (int o = 0; o < L->out_c; o++)
					for (int ky = 0; ky < kh; ky++)
						for (int t = 0; t < kw; t++)
							for (int c = 0; c < in_c; c++)
								wt[(((size_t)x * L->out_c + o) * kh + ky) * rowb + (x * L->sw + t) * in_c + c] = w[((size_t)o * kh + ky) * seg + t * in_c + c];
			w = wt;
			seg = rowb;
		}
		if (M->small)
		{
			/* lane l of k-step s of row tile rt: row 16 rt + (l & 15), chunk 4 s + (l >> 4) */
			for (int rt = 0; rt < M->n_rt16; rt++)
				for (int s = 0; s < M->n_ks16; s++)
				{
					int8_t *f = fb + M->frag_off + ((size_t)rt * M->n_ks16 + s) * 1024;
					for (int l = 0; l < 64; l++)
					{
						const int row = 16 * rt + (l & 15), c = 4 * s + (l >> 4);
						const int ky = c / M->cpr, jc = c - ky * M->cpr;
						for (int j = 0; j < 16; j++)
						{
							const int q = 16 * jc + j;
							f[l * 16 + j] = (row < n_rows && ky < kh && q < seg) ? w[((size_t)row * kh + ky) * seg + q] : 0;
						}
					}
				}
		}
		else
		for (int rt = 0; rt < M->n_rt; rt++)
			for (int s = 0; s < M->n_ks; s++)
			{
				int8_t *f = fb + M->frag_off + ((size_t)rt * M->n_ks + s) * 1024;
				for (int l = 0; l < 64; l++)
				{
					const int row = 32 * rt + (l & 31), c = 2 * s + (l >> 5);
					const int ky = c / M->cpr, jc = c - ky * M->cpr;
					for (int j = 0; j < 16; j++)
					{
						const int q = 16 * jc + j;
						f[l * 16 + j] = (row < n_rows && ky < kh && q < seg) ? w[((size_t)row * kh + ky) * seg + q] : 0;
					}
				}
			}
		free(wt);
		for (int o = 0; o < n_rows; o++)
			sb[M->seed_off + o] = (int32_t)((uint32_t)(int32_t)bias[o % L->out_c] << r.v[6]) + (int32_t)((1u << r.v[7]) >> 1);
		for (int c = 0; c < (M->small ? 4 * M->n_ks16 : 2 * M->n_ks); c++)
		{
			const int ky = c / M->cpr, jc = c - ky * M->cpr;
			/* chunk jc of a kernel row: 16 bytes of pixel jc / (C_in / 16) -- behind a pixel gap the row is not contiguous */
			const int in_row = (!dense && M->pp != in_c) ? (jc / (in_c / 16)) * M->pp + (jc % (in_c / 16)) * 16 : 16 * jc;
			mm->koff[M->koff_off + c] = ky < kh ? ky * M->pitch_y + in_row : 0; /* chunks past the end meet zero weights */
		}
	}
	/* bounds of everything the kernel will address in LDS, checked here on the host (a plan that fails stays off the
	 * matrix-core path instead of reaching the device): B fragment reads and epilogue stores of every layer */
	int in_off = 0; /* the first layer's input sits at the low end of the region */
	const int region = 2 * mm->buf_bytes;
	for (int i = 0; i < n_layers; i++)
	{
		const ed_net_layer_t *L = &plan->L[i];
		const ed_mm_layer_t *M = &mm->L[i];
		if (M->skip) continue; /* fused MaxPool: nothing runs for it */
		int ohp, owp, opy, opx, oimg, opp; /* the consumer's layout = where this layer's epilogue stores */
		const int fused = M->pool_h > 0, nx = fused ? i + 2 : i + 1;          /* the consumer */
		const int st_h = fused ? plan->L[i + 1].out_h : L->out_h, st_w = fused ? plan->L[i + 1].out_w : L->out_w; /* what is stored */
		if (nx < n_layers) { ohp = mm->L[nx].in_hp; owp = mm->L[nx].in_wp; opy = mm->L[nx].in_py; opx = mm->L[nx].in_px; oimg = mm->L[nx].in_img; opp = plan->L[nx].type == ED_NET_CONV ? mm->L[nx].pp : L->out_c; }
		else { ohp = st_h; owp = st_w; opy = 0; opx = 0; oimg = up16(st_h * st_w * L->out_c) + 16; opp = L->out_c; }
		const int64_t last_store = ((int64_t)(st_h - 1 + opy) * owp + (st_w - 1 + opx)) * opp + L->out_c;
		/* input at one end of the region, output at the other */
		const int o_off = in_off == 0 ? region - batch * oimg : 0;
		int bad = last_store > oimg || ohp < st_h + opy || owp < st_w + opx || batch * (M->in_img + oimg) > region || o_off < 0 || (o_off & 15) || (in_off & 15);
		bad |= in_off + batch * M->in_img > region || (in_off == 0 ? batch * M->in_img > o_off : batch * oimg > in_off);
		if (M->mm)
		{
			const int dense = L->type == ED_NET_DENSE;
			const int out_h = dense ? 1 : L->out_h, out_w = dense || M->toep ? 1 : L->out_w, sh = dense ? 1 : L->sh;
			const int img = M->expand ? M->x_img : M->in_img;
			int max_koff = 0;
			for (int c = 0; c < (M->small ? 4 * M->n_ks16 : 2 * M->n_ks); c++) if (mm->koff[M->koff_off + c] > max_koff) max_koff = mm->koff[M->koff_off + c];
			const int64_t last_read = (int64_t)(out_h - 1) * sh * M->pitch_y + (int64_t)(out_w - 1) * M->pitch_x + max_koff + 16;
			bad |= last_read > img || (M->pitch_x & 15) || (M->pitch_y & 15) || (M->in_img & 15) || (M->x_img & 15);
			bad |= M->expand ? (batch * up16(M->x_img) > mm->x_bytes + 0) : 0;
			/* the tables the kernel follows blindly: every entry re-checked against the buffers it indexes */
			if (M->col_off >= 0)
			{
				const int pix = dense ? 1 : st_h * (M->toep ? 1 : st_w), wmax = fused ? ((M->pool_h - 1) * sh) * M->pitch_y + (M->pool_w - 1) * M->pitch_x : 0;
				for (int q = 0; q < pix; q++)
				{
					const int boff = mm->coltab[2 * (M->col_off + q)], ooff = mm->coltab[2 * (M->col_off + q) + 1];
					bad |= boff < 0 || (boff & 15) || (int64_t)boff + wmax + max_koff + 16 > img || ooff < 0 || ooff + (M->toep ? L->out_w : 1) * L->out_c > oimg;
				}
			}
			if (M->xtab_off >= 0)
			{
				const int rec = (dense ? 1 : M->in_hp) * out_w * M->cpr;
				for (int q = 0; q < rec; q++)
				{
					const int w0 = mm->xtab[2 * (M->xtab_off + q)], doff = mm->xtab[2 * (M->xtab_off + q) + 1];
					const int soff = w0 & 0xffffff, keep = w0 >> 24;
					/* the gather reads five aligned dwords around soff: up to soff + 20 */
					bad |= keep < 1 || keep > 16 || soff + 20 > M->in_img + 4 || doff < 0 || (doff & 15) || doff + 16 > M->x_img;
				}
			}
		}
		if (i == 0)
			for (int e = 0; e < mm->n_intab; e++) bad |= mm->intab[e] >= M->in_img;
		bad |= M->toep && opp != L->out_c; /* a Toeplitz layer stores whole rows: its consumer's pixels must be gap-free */
		if (bad) { free(fb); free(sb); return EDISON_OK; } /* mm->ok stays 0 */
		/* the run record */
		ed_mm_run_t *R = &mm->R[i];
		const int dense = L->type == ED_NET_DENSE;
		R->kind = M->mm ? ED_RUN_MM : (L->type == ED_NET_POOL ? ((L->in_c & 3) == 0 ? ED_RUN_POOL4 : ED_RUN_POOL1) : ED_RUN_SOFTMAX);
		R->zero_border = ohp != st_h || owp != st_w;
		R->in_img = M->in_img; R->o_img = oimg;
		R->oc_pitch = opp; R->o_origin = (opy * owp + opx) * opp; R->o_row = owp * opp;
		R->li_out = fused ? i + 1 : i;
		R->expand = M->expand; R->x_img = M->x_img; R->xtab_off = M->xtab_off;
		R->rec_per_img = M->mm ? (dense ? 1 : M->in_hp) * (dense ? 1 : L->out_w) * M->cpr : 0;
		R->pitch_x = M->pitch_x; R->pitch_y = M->pitch_y; R->sh = dense ? 1 : L->sh;
		R->ph = fused ? M->pool_h : 1; R->pw = fused ? M->pool_w : 1;
		R->small = M->small;
		R->n_ks = M->small ? M->n_ks16 : M->n_ks; R->n_rt = M->small ? M->n_rt16 : M->n_rt; R->frag_off = M->frag_off; R->seed_off = M->seed_off; R->koff_off = M->koff_off; R->col_off = M->col_off;
		R->pix_per_img = dense ? 1 : st_h * (M->toep ? 1 : st_w); R->col_w = dense || M->toep ? 1 : st_w;
		R->out_c = M->toep ? L->out_w * L->out_c : L->out_c; /* rows of the GEMM */
		R->rs = L->rs; R->lo_clamp = L->relu ? 0 : -128;
		/* sat8(v >> rs) is the HIGH byte of sat16(v >> (rs - 8)) (arithmetic shifts compose, and saturating to 16 bits commutes
		 * with dropping 8 more): two v_cvt_pk_i16_i32 and one v_perm_b32 per four values instead of four shift / clamp pairs.
		 * For rs < 8 the inner shift goes LEFT, which is exact only while v << (8 - rs) stays inside 32 bits: decided here from
		 * the layer's own weights and seeds. */
		if (M->mm && L->rs >= 0 && L->rs <= 31 && (L->rs >= 8 || (acc_bound[i] << (8 - L->rs)) < ((int64_t)1 << 31))) R->rs |= ED_RUN_RS_HI;
		R->in_n = L->in_n;
		R->in_off = in_off; R->o_off = o_off;
		in_off = o_off; /* the consumer reads where this layer stored */
	}
	*frag = fb;
	*seeds = sb;
	mm->ok = 1;
	return EDISON_OK;
}
