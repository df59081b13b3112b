/*
 * oracle/kws_cnn_ref.c -- TEST INFRASTRUCTURE (see oracle.h). Plain-C restatement of the int8 CNN
 * arithmetic the reference runs through NNoM 0.3.0 + CMSIS-NN (portable branches, i.e. what the
 * x86 build of the reference executes):
 *
 *   conv2d    arm_convolve_HWC_q7_basic_nonsquare.c:188-221 / arm_convolve_HWC_q7_fast_nonsquare.c (same
 *             arithmetic in its portable branch); dispatch nnom_conv2d.c:141-213
 *             out = ssat8( (sum x*w + (bias << bias_shift) + NN_ROUND(out_shift)) >> out_shift )
 *   relu      arm_relu_q7.c:57-105, run in place as NNoM's tail activation (nnom.c:986-989)
 *   maxpool   local_maxpool_q7_HWC, nnom_local.c:117-159 (dispatch nnom_maxpool.c:138-150)
 *   dense     arm_fully_connected_q7_opt.c:374-473 -- the weight stream is de-interleaved once by
 *             tools/import_weights_h.py, so this file multiplies a plain row-major [out][in] matrix
 *   softmax   arm_softmax_q7.c:215-260 (portable branch)
 *   argmax    nnom_predict, nnom_utils.c:275-284: strict '>' => first maximum
 *
 * Tensors are HWC int8, accumulators int32, shifts arithmetic on signed int32.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define NN_ROUND(s) ((int32_t)((0x1u << (s)) >> 1)) /* arm_nnsupportfunctions.h */

static int8_t ssat8(int32_t v) { return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v)); }
static int32_t usat(int32_t v, int bits) { int32_t hi = (1 << bits) - 1; return v < 0 ? 0 : (v > hi ? hi : v); }

static void conv2d_q7(const int8_t *in, int H, int W, int C, const oracle_layer_t *L, int8_t *out, int OH, int OW)
{
	(void)H;
	const int O = L->out_ch, KH = L->kh, KW = L->kw;
	for (int o = 0; o < O; o++)
		for (int y = 0; y < OH; y++)
			for (int x = 0; x < OW; x++)
			{
				int32_t acc = ((int32_t)L->b[o] << L->bias_lshift) + NN_ROUND(L->out_rshift);
				for (int m = 0; m < KH; m++)
					for (int n = 0; n < KW; n++)
					{
						const int8_t *px = in + ((size_t)(y * L->sh + m) * W + (x * L->sw + n)) * C;
						const int8_t *pw = L->w + (size_t)o * C * KH * KW + (size_t)(m * KW + n) * C;
						for (int l = 0; l < C; l++) acc += (int32_t)px[l] * (int32_t)pw[l];
					}
				int8_t v = ssat8(acc >> L->out_rshift);
				if (L->relu && v < 0) v = 0;
				out[o + ((size_t)y * OW + x) * O] = v;
			}
}

static void maxpool_q7(const int8_t *in, int H, int W, int C, const oracle_layer_t *L, int8_t *out, int OH, int OW)
{
	(void)H;
	for (int c = 0; c < C; c++)
		for (int y = 0; y < OH; y++)
			for (int x = 0; x < OW; x++)
			{
				int mx = -129;
				for (int ky = y * L->sh; ky < y * L->sh + L->kh; ky++)
					for (int kx = x * L->sw; kx < x * L->sw + L->kw; kx++)
					{
						int v = in[c + (size_t)C * (kx + (size_t)ky * W)];
						if (v > mx) mx = v;
					}
				out[c + (size_t)C * (x + (size_t)y * OW)] = (int8_t)mx;
			}
}

static void dense_q7(const int8_t *in, int n_in, const oracle_layer_t *L, int8_t *out)
{
	for (int r = 0; r < L->out_ch; r++)
	{
		int32_t acc = ((int32_t)L->b[r] << L->bias_lshift) + NN_ROUND(L->out_rshift);
		for (int j = 0; j < n_in; j++) acc += (int32_t)in[j] * (int32_t)L->w[(size_t)r * n_in + j];
		out[r] = ssat8(acc >> L->out_rshift);
	}
}

static void softmax_q7(const int8_t *in, int n, int8_t *out)
{
	int32_t base = -128;
	for (int i = 0; i < n; i++) if (in[i] > base) base = in[i];
	base -= 8; /* Q7BITS */
	int32_t sum = 0;
	for (int i = 0; i < n; i++) sum += 0x1 << usat(in[i] - base, 3); /* LOG2Q7BITS */
	int32_t output_base = (1 << 20) / sum;
	for (int i = 0; i < n; i++) out[i] = ssat8(output_base >> usat(13 + base - in[i], 5));
}

static int run_one(const oracle_layer_t *layers, int n_layers, int h, int w, int c, const int8_t *in,
                   int8_t *buf0, int8_t *buf1, int8_t *acts, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	const int8_t *cur = in;
	int8_t *nxt = buf0;
	size_t act_off = 0;
	int last_n = h * w * c;
	for (int i = 0; i < n_layers; i++)
	{
		const oracle_layer_t *L = &layers[i];
		int oh = h, ow = w, oc = c;
		switch (L->type)
		{
		case ORACLE_L_CONV:
			oh = (h - L->kh) / L->sh + 1; ow = (w - L->kw) / L->sw + 1; oc = L->out_ch;
			conv2d_q7(cur, h, w, c, L, nxt, oh, ow);
			break;
		case ORACLE_L_POOL:
			oh = (h - L->kh) / L->sh + 1; ow = (w - L->kw) / L->sw + 1;
			maxpool_q7(cur, h, w, c, L, nxt, oh, ow);
			break;
		case ORACLE_L_DENSE:
			oh = 1; ow = 1; oc = L->out_ch;
			dense_q7(cur, h * w * c, L, nxt);
			if (logits) memcpy(logits, nxt, (size_t)oc);
			break;
		case ORACLE_L_SOFTMAX:
			softmax_q7(cur, h * w * c, nxt);
			if (softmax) memcpy(softmax, nxt, (size_t)(h * w * c));
			break;
		default:
			return -1;
		}
		h = oh; w = ow; c = oc; last_n = h * w * c;
		if (acts) { memcpy(acts + act_off, nxt, (size_t)last_n); act_off += (size_t)last_n; }
		cur = nxt;
		nxt = (nxt == buf0) ? buf1 : buf0;
	}
	if (argmax)
	{
		int best = 0; int8_t mx = cur[0];
		for (int i = 1; i < last_n; i++) if (cur[i] > mx) { mx = cur[i]; best = i; }
		*argmax = best;
	}
	return 0;
}

int oracle_cnn_run(const oracle_layer_t *layers, int n_layers, int in_h, int in_w, int in_c,
                   const int8_t *in, int64_t n, int8_t *acts, int64_t acts_stride,
                   int8_t *logits, int8_t *softmax, int32_t *argmax, int n_threads)
{
	/* largest activation of any layer bounds the ping-pong buffers */
	size_t maxact = (size_t)in_h * in_w * in_c;
	int n_out = 0, n_logits = 0;
	{
		int h = in_h, w = in_w, c = in_c;
		for (int i = 0; i < n_layers; i++)
		{
			const oracle_layer_t *L = &layers[i];
			if (L->type == ORACLE_L_CONV) { h = (h - L->kh) / L->sh + 1; w = (w - L->kw) / L->sw + 1; c = L->out_ch; }
			else if (L->type == ORACLE_L_POOL) { h = (h - L->kh) / L->sh + 1; w = (w - L->kw) / L->sw + 1; }
			else if (L->type == ORACLE_L_DENSE) { h = 1; w = 1; c = L->out_ch; n_logits = c; }
			if (h < 1 || w < 1) return -1;
			if ((size_t)h * w * c > maxact) maxact = (size_t)h * w * c;
			n_out = h * w * c;
		}
	}
	const size_t in_sz = (size_t)in_h * in_w * in_c;
	int err = 0;
	if (n_threads < 1) n_threads = 1;
	#pragma omp parallel num_threads(n_threads)
	{
		int8_t *b0 = (int8_t *)malloc(maxact), *b1 = (int8_t *)malloc(maxact);
		if (!b0 || !b1)
		{
			#pragma omp atomic write
			err = -2;
		}
		else
		{
			#pragma omp for schedule(static)
			for (int64_t u = 0; u < n; u++)
			{
				int r = run_one(layers, n_layers, in_h, in_w, in_c, in + (size_t)u * in_sz, b0, b1,
				                acts ? acts + (size_t)u * acts_stride : NULL,
				                logits ? logits + (size_t)u * n_logits : NULL,
				                softmax ? softmax + (size_t)u * n_out : NULL,
				                argmax ? argmax + u : NULL);
				if (r != 0)
				{
					#pragma omp atomic write
					err = r;
				}
			}
		}
		free(b0); free(b1);
	}
	return err;
}
