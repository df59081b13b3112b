/*
 * model_net.c -- host-side (plain C) planner for ANY sequential NNoM int8 graph the importer understands
 * (tools/import_weights_h.py): Input -> {Conv2D [+ReLU] | MaxPool | Dense [+ReLU] | Softmax}* -> Output.
 *
 * It does for the GPU what model_compile() + the layers' build functions do in the reference (nnom.c:758-900,
 * nnom_conv2d.c:79-108, nnom_maxpool.c:80-106, nnom_dense.c:70-91): derive every tensor shape, decide where the
 * activations live (two LDS buffers here instead of NNoM's memory blocks) and pre-compute what does not depend on
 * the input -- the accumulator seeds (bias << bias_lshift) + NN_ROUND(out_rshift) of
 * arm_convolve_HWC_q7_basic_nonsquare.c:196 / arm_fully_connected_q7_opt.c:383.
 *
 * Where the reference's own dispatch would not compute the plain formula, the graph is refused instead of being run
 * differently from the reference:
 *   - square images go to the CMSIS-NN "square" kernels, which are handed kernel.w / stride.w / pad.w only
 *     (nnom_conv2d.c:146-153,177-184, nnom_maxpool.c:124-134) and index with `signed char` (arm_convolve_HWC_q7_basic.c);
 *   - a 1x1 convolution with C_in % 4 == 0, C_out % 2 == 0 and a stride other than 1 makes
 *     arm_convolve_1x1_HWC_q7_fast_nonsquare return ARM_MATH_SIZE_MISMATCH (:204-209).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

typedef struct { int32_t v[12]; } rec_t;

static int fail(char *err, size_t cap, int code, const char *fmt, int a, int b)
{
	if (err && cap) snprintf(err, cap, fmt, a, b);
	return code;
}

static int ceil_div(int a, int b) { return (a + b - 1) / b; }

int ed_plan_net(const void *blob, size_t blob_bytes, ed_net_plan_t *plan, int8_t **weights, int32_t **seeds, char *err,
                size_t err_cap)
{
	const unsigned char *p = (const unsigned char *)blob;
	*weights = NULL;
	*seeds = NULL;
	if (blob == NULL || blob_bytes < 40 || memcmp(p, "EDNNOM1\0", 8) != 0)
		return fail(err, err_cap, EDISON_E_SIZE, "not an .ednn model blob", 0, 0);
	int32_t head[8];
	memcpy(head, p + 8, sizeof(head));
	const int n_layers = head[3], payload_bytes = head[4];
	if (n_layers < 1 || n_layers > ED_NET_MAX_LAYERS || payload_bytes < 0 ||
	    blob_bytes < 40 + (size_t)n_layers * 48 + (size_t)payload_bytes)
		return fail(err, err_cap, EDISON_E_SIZE, "truncated .ednn model blob, or more than %d layers", ED_NET_MAX_LAYERS, 0);
	const int8_t *payload = (const int8_t *)(p + 40 + (size_t)n_layers * 48);

	memset(plan, 0, sizeof(*plan));
	int h = head[0], w = head[1], c = head[2];
	if (h < 1 || w < 1 || c < 1 || (int64_t)h * w * c > ED_NET_MAX_LDS / 2)
		return fail(err, err_cap, EDISON_E_SIZE, "model input shape out of range", 0, 0);
	plan->n_layers = n_layers;
	plan->in_h = h; plan->in_w = w; plan->in_c = c; plan->in_n = h * w * c;

	/* every tensor re-aligned to 16; one seed per bias byte at most. Tensors of a well-formed blob are disjoint, so their
	 * sizes add up to at most the payload; records that point at overlapping ranges are refused when the sum outgrows
	 * these buffers (w_cap / s_cap checks below), not written past their end. */
	const size_t w_cap = (size_t)payload_bytes + 16 * ((size_t)n_layers + 1), s_cap = (size_t)payload_bytes + 16;
	int8_t *wbuf = (int8_t *)calloc(w_cap, 1);
	int32_t *sbuf = (int32_t *)calloc(s_cap, sizeof(int32_t));
	if (!wbuf || !sbuf) { free(wbuf); free(sbuf); return fail(err, err_cap, EDISON_E_NO_MEMORY, "host allocation failed", 0, 0); }
	int w_used = 0, s_used = 0, acts = 0, max_act = (plan->in_n + 15) & ~15, rc = EDISON_OK;

	for (int i = 0; i < n_layers && rc == EDISON_OK; i++)
	{
		rec_t r;
		memcpy(&r, p + 40 + (size_t)i * 48, sizeof(r));
		const int32_t *v = r.v;
		ed_net_layer_t *L = &plan->L[i];
		L->type = v[0];
		L->in_h = h; L->in_w = w; L->in_c = c; L->in_n = h * w * c;
		if (v[0] == ED_NET_CONV || v[0] == ED_NET_POOL)
		{
			const int same = (v[8] >> 1) & 1;
			L->kh = v[2]; L->kw = v[3]; L->sh = v[4]; L->sw = v[5];
			if (L->kh < 1 || L->kw < 1 || L->sh < 1 || L->sw < 1 || L->kh > 255 || L->kw > 255)
			{ rc = fail(err, err_cap, EDISON_E_SIZE, "layer %d: kernel or stride out of range", i, 0); break; }
			L->pad_h = same ? (L->kh - 1) / 2 : 0; /* nnom_conv2d.c:66-71, nnom_maxpool.c:65-70 */
			L->pad_w = same ? (L->kw - 1) / 2 : 0;
			L->out_h = same ? ceil_div(h, L->sh) : ceil_div(h - L->kh + 1, L->sh);
			L->out_w = same ? ceil_div(w, L->sw) : ceil_div(w - L->kw + 1, L->sw);
			if (L->out_h < 1 || L->out_w < 1)
			{ rc = fail(err, err_cap, EDISON_E_SIZE, "layer %d: kernel larger than its %d-row input", i, h); break; }
			L->check_taps = L->pad_h > 0 || L->pad_w > 0 || (L->out_h - 1) * L->sh - L->pad_h + L->kh > h ||
			                (L->out_w - 1) * L->sw - L->pad_w + L->kw > w;
			const int square_in = h == w;
			if (v[0] == ED_NET_CONV)
			{
				L->out_c = v[1]; L->relu = v[8] & 1; L->rs = v[7];
				if (L->out_c < 1 || v[11] != c || v[6] < 0 || v[6] > 23 || v[7] < 0 || v[7] > 30)
				{ rc = fail(err, err_cap, EDISON_E_SIZE, "layer %d: channel count or shifts inconsistent (input has %d channels)", i, c); break; }
				if (square_in && (L->kh != L->kw || L->sh != L->sw || h + L->kh > 127))
				{ rc = fail(err, err_cap, EDISON_E_NO_IMPL, "layer %d: on a square image (%d rows) the reference's CMSIS-NN kernels use kernel.w/stride.w for both axes and 8-bit indices; refused", i, h); break; }
				if (L->kh == 1 && L->kw == 1 && c % 4 == 0 && L->out_c % 2 == 0 && (L->sh != 1 || L->sw != 1))
				{ rc = fail(err, err_cap, EDISON_E_SIZE, "layer %d: the reference's 1x1 convolution kernel rejects strides other than 1 (NN_SIZE_MISMATCH)", i, 0); break; }
				const int64_t wn = (int64_t)L->out_c * L->kh * L->kw * c;
				if (v[9] < 0 || v[10] < 0 || v[9] + wn > payload_bytes || (int64_t)v[10] + L->out_c > payload_bytes)
				{ rc = fail(err, err_cap, EDISON_E_SIZE, "layer %d: weight tensor outside the payload", i, 0); break; }
				w_used = (w_used + 15) & ~15;
				if ((size_t)w_used + (size_t)wn > w_cap || (size_t)s_used + (size_t)L->out_c > s_cap)
				{ rc = fail(err, err_cap, EDISON_E_SIZE, "layer %d: tensors overlap (their sizes exceed the %d-byte payload)", i, payload_bytes); break; }
				L->w_off = w_used;
				/* stored OHWI (arm_convolve_HWC_q7_basic_nonsquare.c:209-211 reads w[o][ky][kx][ci]); the kernel wants the
				 * output channel innermost so that consecutive lanes read consecutive addresses: [k/4][o] dwords of four
				 * input channels when C_in % 4 == 0, [k][o] bytes otherwise, k = (ky*kw + kx)*C_in + ci */
				{
					const int K = L->kh * L->kw * c, oc = L->out_c, g = (c % 4 == 0) ? 4 : 1;
					const int8_t *src = payload + v[9];
					int8_t *dst = wbuf + w_used;
					for (int o = 0; o < oc; o++)
						for (int k = 0; k < K; k++) dst[((size_t)(k / g) * oc + o) * g + (k % g)] = src[(size_t)o * K + k];
				}
				w_used += (int)wn;
				L->seed_off = s_used;
				for (int o = 0; o < L->out_c; o++)
					sbuf[s_used++] = (int32_t)((uint32_t)(int32_t)payload[v[10] + o] << v[6]) + (int32_t)((1u << v[7]) >> 1);
			}
			else
			{
				L->out_c = c;
				const int square_out = L->out_h == L->out_w;
				if (square_in && square_out && (L->kh != L->kw || L->sh != L->sw))
				{ rc = fail(err, err_cap, EDISON_E_NO_IMPL, "layer %d: on a square image (%d rows) the reference pools with kernel.w/stride.w on both axes; refused", i, h); break; }
			}
			h = L->out_h; w = L->out_w; c = L->out_c;
		}
		else if (v[0] == ED_NET_DENSE)
		{
			L->out_h = 1; L->out_w = 1; L->out_c = v[1]; L->relu = v[8] & 1; L->rs = v[7];
			L->kh = L->kw = L->sh = L->sw = 1;
			const int64_t wn = (int64_t)v[1] * L->in_n;
			if (v[1] < 1 || v[11] != L->in_n || v[6] < 0 || v[6] > 23 || v[7] < 0 || v[7] > 30)
			{ rc = fail(err, err_cap, EDISON_E_SIZE, "layer %d: dense input width or shifts inconsistent (flattened input is %d)", i, L->in_n); break; }
			if (v[9] < 0 || v[10] < 0 || v[9] + wn > payload_bytes || (int64_t)v[10] + v[1] > payload_bytes)
			{ rc = fail(err, err_cap, EDISON_E_SIZE, "layer %d: dense tensor outside the payload", i, 0); break; }
			w_used = (w_used + 15) & ~15;
			if ((size_t)w_used + (size_t)wn > w_cap || (size_t)s_used + (size_t)v[1] > s_cap)
			{ rc = fail(err, err_cap, EDISON_E_SIZE, "layer %d: tensors overlap (their sizes exceed the %d-byte payload)", i, payload_bytes); break; }
			L->w_off = w_used;
			memcpy(wbuf + w_used, payload + v[9], (size_t)wn); /* [out][in], de-interleaved by the importer */
			w_used += (int)wn;
			L->seed_off = s_used;
			for (int o = 0; o < v[1]; o++)
				sbuf[s_used++] = (int32_t)((uint32_t)(int32_t)payload[v[10] + o] << v[6]) + (int32_t)((1u << v[7]) >> 1);
			h = 1; w = 1; c = v[1];
		}
		else if (v[0] == ED_NET_SOFTMAX)
		{
			if (i != n_layers - 1 || i == 0)
			{ rc = fail(err, err_cap, EDISON_E_NO_IMPL, "layer %d: Softmax is only supported as the last layer", i, 0); break; }
			L->out_h = h; L->out_w = w; L->out_c = c;
			plan->has_softmax = 1;
		}
		else
		{ rc = fail(err, err_cap, EDISON_E_NO_IMPL, "layer %d: layer type %d is not built on this path", i, v[0]); break; }
		L->out_n = L->out_h * L->out_w * L->out_c;
		L->acts_off = acts;
		acts += L->out_n;
		if (((L->out_n + 15) & ~15) > max_act) max_act = (L->out_n + 15) & ~15;
		if (2 * max_act > ED_NET_MAX_LDS)
		{ rc = fail(err, err_cap, EDISON_E_NO_IMPL, "layer %d: activations of %d bytes do not fit the two LDS buffers", i, L->out_n); break; }
	}
	if (rc != EDISON_OK) { free(wbuf); free(sbuf); return rc; }
	for (int i = 0; i < n_layers; i++)
	{
		plan->L[i].in_buf = (i & 1) ? max_act : 0;
		plan->L[i].out_buf = (i & 1) ? 0 : max_act;
	}
	plan->logits_layer = plan->has_softmax ? n_layers - 2 : n_layers - 1;
	plan->out_n = plan->L[n_layers - 1].out_n;
	plan->lds_bytes = 2 * max_act;
	plan->acts_bytes = acts;
	plan->weights_bytes = (w_used + 15) & ~15;
	plan->n_seeds = s_used;
	*weights = wbuf;
	*seeds = sbuf;
	return EDISON_OK;
}
