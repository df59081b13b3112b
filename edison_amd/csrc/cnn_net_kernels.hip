/*
 * cnn_net_kernels.hip -- layer-by-layer int8 inference of ANY sequential NNoM graph the planner accepts
 * (model_net.c), hand-written HIP for gfx950. It is the GPU's model_run() (nnom.c:975-1040): one workgroup owns an
 * utterance and walks the layer list; activations (HWC int8) ping-pong between two LDS buffers, weights (output
 * channel innermost, see model_net.c) and the accumulator seeds are read from L1/L2 where every workgroup shares them.
 *
 * Arithmetic restated (all integer, results are bit-exact, not "within tolerance"):
 *   Conv2D   sat8((sum x*w + (bias << BL) + NN_ROUND(RS)) >> RS), zero padding by skipping taps outside the image
 *            (arm_convolve_HWC_q7_basic_nonsquare.c:188-221; the fast / 1x1 / RGB / square variants compute the same
 *            on their portable branches), ReLU = arm_relu_q7 as a tail activation (nnom.c:986-989)
 *   MaxPool  maximum over the part of the window inside the image, starting from -129 (nnom_local.c:117-159,
 *            arm_pool_q7_HWC.c portable branch)
 *   Dense    the same requantisation over the flattened HWC input (arm_fully_connected_q7_opt.c:374-473; the
 *            importer undoes the weight interleave)
 *   Softmax  arm_softmax_q7.c:215-260, portable branch; argmax = first maximum of the last layer (nnom_utils.c:275-284)
 *
 * The shipped kws_conv graph does not come here: cnn_mfma_kernels.hip runs it on the matrix cores. This kernel trades
 * peak speed for generality: 4-channel groups use v_dot4_i32_i8, everything else plain multiply-adds; convolutions
 * with C_out % 4 == 0 are register-blocked (4 channels x 2 pixels per thread).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

#define EN_THREADS 256

__device__ __forceinline__ int en_ssat8(int v) { return v > 127 ? 127 : (v < -128 ? -128 : v); }
__device__ __forceinline__ int en_usat(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

/* Consecutive lanes own consecutive output channels of a pixel: the weights (re-laid [k/4][o] or [k][o] by model_net.c)
 * are read as one coalesced row per step, the activations as an LDS broadcast, the output bytes are contiguous. */
template <bool C4, bool PADDED>
__device__ __forceinline__ void en_conv(const ed_net_layer_t &L, const int8_t *in, int8_t *out, const int8_t *__restrict__ W,
                                        const int32_t *__restrict__ S)
{
	const int oc = L.out_c, cg = C4 ? L.in_c / 4 : L.in_c; /* weight rows per tap */
	for (int idx = threadIdx.x; idx < L.out_n; idx += EN_THREADS)
	{
		const int pix = idx / oc, o = idx - pix * oc;
		const int y = pix / L.out_w, x = pix - y * L.out_w;
		int acc = S[L.seed_off + o];
		for (int ky = 0; ky < L.kh; ky++)
		{
			const int iy = y * L.sh - L.pad_h + ky;
			if (PADDED && (unsigned)iy >= (unsigned)L.in_h) continue;
			for (int kx = 0; kx < L.kw; kx++)
			{
				const int ix = x * L.sw - L.pad_w + kx;
				if (PADDED && (unsigned)ix >= (unsigned)L.in_w) continue;
				const int8_t *a = in + (iy * L.in_w + ix) * L.in_c;
				const int row0 = (ky * L.kw + kx) * cg;
				if (C4)
				{
					const int *a4 = reinterpret_cast<const int *>(a);
					const int *w4 = reinterpret_cast<const int *>(W + L.w_off) + (int64_t)row0 * oc + o;
#pragma unroll 8
					for (int c = 0; c < cg; c++) acc = __builtin_amdgcn_sdot4(a4[c], w4[(int64_t)c * oc], acc, false);
				}
				else
				{
					const int8_t *w1 = W + L.w_off + (int64_t)row0 * oc + o;
#pragma unroll 4
					for (int c = 0; c < cg; c++) acc += (int)a[c] * (int)w1[(int64_t)c * oc];
				}
			}
		}
		int v = en_ssat8(acc >> L.rs);
		if (L.relu && v < 0) v = 0;
		out[idx] = (int8_t)v;
	}
}

/* The same convolution, register-blocked for C_out % 4 == 0: a thread owns 4 output channels (oq + j * C_out/4, so
 * that lanes still read consecutive weights) of EN_PB consecutive pixels. One activation read feeds 4 multiply-adds
 * and one weight read EN_PB: 4 + EN_PB loads per 4 * EN_PB instead of 8 * EN_PB. Taps outside the image contribute
 * a zero activation; a pixel tail recomputes the last pixel and stores it once. */
/* ---- lab knobs: only a lab build (ED_LAB, tools/lab/mkvariant.py) may set them; the product build has none, and
 * tests/test_host_cpu.py checks the values below against what edison_amd/build.py compiles */
#if !defined(ED_LAB) && (defined(EN_PB))
#error "EN_PB lab knob defined without ED_LAB (tools/lab/mkvariant.py builds lab variants)"
#endif
#if defined(ED_LAB)
/* a lab build says so: the product library exports no ed_lab_build_* symbol (tests/test_host_cpu.py) */
extern "C" { extern const int ed_lab_build_cnn_net; const int ed_lab_build_cnn_net = 1; }
#endif
#ifndef EN_PB
#define EN_PB 2 /* 4 measured slower on the test graphs: too few work items left in the small late layers */
#endif
template <bool C4, bool PADDED>
__device__ __forceinline__ void en_conv_blocked(const ed_net_layer_t &L, const int8_t *in, int8_t *out, const int8_t *__restrict__ W,
                                                const int32_t *__restrict__ S)
{
	const int oc = L.out_c, oq_n = oc >> 2, cg = C4 ? L.in_c / 4 : L.in_c;
	const int npix = L.out_h * L.out_w, ngrp = (npix + EN_PB - 1) / EN_PB;
	for (int idx = threadIdx.x; idx < ngrp * oq_n; idx += EN_THREADS)
	{
		const int pg = idx / oq_n, oq = idx - pg * oq_n;
		int pix[EN_PB], ybase[EN_PB], xbase[EN_PB], acc[EN_PB][4];
#pragma unroll
		for (int i = 0; i < EN_PB; i++)
		{
			pix[i] = EN_PB * pg + i < npix ? EN_PB * pg + i : npix - 1;
			const int y = pix[i] / L.out_w;
			ybase[i] = y * L.sh - L.pad_h;
			xbase[i] = (pix[i] - y * L.out_w) * L.sw - L.pad_w;
#pragma unroll
			for (int j = 0; j < 4; j++) acc[i][j] = S[L.seed_off + oq + j * oq_n];
		}
		for (int ky = 0; ky < L.kh; ky++)
			for (int kx = 0; kx < L.kw; kx++)
			{
				bool inside[EN_PB];
				const int8_t *a[EN_PB];
#pragma unroll
				for (int i = 0; i < EN_PB; i++)
				{
					const int iy = ybase[i] + ky, ix = xbase[i] + kx;
					inside[i] = !PADDED || ((unsigned)iy < (unsigned)L.in_h && (unsigned)ix < (unsigned)L.in_w);
					a[i] = in + (inside[i] ? (iy * L.in_w + ix) * L.in_c : 0);
				}
				const int64_t row0 = (int64_t)(ky * L.kw + kx) * cg * oc + oq;
				for (int c = 0; c < cg; c++)
				{
					int v[EN_PB], w[4];
#pragma unroll
					for (int i = 0; i < EN_PB; i++)
						v[i] = !inside[i] ? 0 : C4 ? reinterpret_cast<const int *>(a[i])[c] : (int)a[i][c];
#pragma unroll
					for (int j = 0; j < 4; j++)
						w[j] = C4 ? (reinterpret_cast<const int *>(W + L.w_off) + row0)[(int64_t)c * oc + j * oq_n]
						          : (int)(W + L.w_off + row0)[(int64_t)c * oc + j * oq_n];
#pragma unroll
					for (int i = 0; i < EN_PB; i++)
#pragma unroll
						for (int j = 0; j < 4; j++)
							acc[i][j] = C4 ? __builtin_amdgcn_sdot4(v[i], w[j], acc[i][j], false) : acc[i][j] + v[i] * w[j];
				}
			}
#pragma unroll
		for (int i = 0; i < EN_PB; i++)
		{
			if (EN_PB * pg + i >= npix) continue;
#pragma unroll
			for (int j = 0; j < 4; j++)
			{
				int r = en_ssat8(acc[i][j] >> L.rs);
				if (L.relu && r < 0) r = 0;
				out[pix[i] * oc + oq + j * oq_n] = (int8_t)r;
			}
		}
	}
}

__device__ __forceinline__ void en_pool(const ed_net_layer_t &L, const int8_t *in, int8_t *out)
{
	for (int idx = threadIdx.x; idx < L.out_n; idx += EN_THREADS)
	{
		const int pix = idx / L.in_c, c = idx - pix * L.in_c;
		const int y = pix / L.out_w, x = pix - y * L.out_w;
		int mx = -129;
		for (int ky = 0; ky < L.kh; ky++)
		{
			const int iy = y * L.sh - L.pad_h + ky;
			if ((unsigned)iy >= (unsigned)L.in_h) continue;
			for (int kx = 0; kx < L.kw; kx++)
			{
				const int ix = x * L.sw - L.pad_w + kx;
				if ((unsigned)ix >= (unsigned)L.in_w) continue;
				const int v = in[(iy * L.in_w + ix) * L.in_c + c];
				mx = v > mx ? v : mx;
			}
		}
		out[idx] = (int8_t)mx;
	}
}

/* one wavefront per output unit, the input vector split over its lanes; integer sums are exact in any order */
__device__ __forceinline__ void en_dense(const ed_net_layer_t &L, const int8_t *in, int8_t *out, const int8_t *__restrict__ W,
                                         const int32_t *__restrict__ S)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int o = wave; o < L.out_c; o += EN_THREADS / 64)
	{
		const int8_t *wrow = W + L.w_off + (int64_t)o * L.in_n;
		int acc = 0;
		if ((L.in_n & 3) == 0)
		{
			const int *a4 = reinterpret_cast<const int *>(in), *w4 = reinterpret_cast<const int *>(wrow);
			for (int c = lane; c < L.in_n / 4; c += 64) acc = __builtin_amdgcn_sdot4(a4[c], w4[c], acc, false);
		}
		else
			for (int c = lane; c < L.in_n; c += 64) acc += (int)in[c] * (int)wrow[c];
		for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
		if (lane == 0)
		{
			int v = en_ssat8((acc + S[L.seed_off + o]) >> L.rs);
			if (L.relu && v < 0) v = 0;
			out[o] = (int8_t)v;
		}
	}
}

__device__ __forceinline__ void en_softmax(const ed_net_layer_t &L, const int8_t *in, int8_t *out)
{
	if (threadIdx.x != 0) return;
	int base = -128;
	for (int i = 0; i < L.in_n; i++) base = in[i] > base ? in[i] : base;
	base -= 8;
	int sum = 0;
	for (int i = 0; i < L.in_n; i++) sum += 1 << en_usat(in[i] - base, 7);
	const int output_base = (1 << 20) / sum;
	for (int i = 0; i < L.in_n; i++) out[i] = (int8_t)en_ssat8(output_base >> en_usat(13 + base - in[i], 31));
}

__global__ __launch_bounds__(EN_THREADS) void ed_net_kernel(const ed_net_plan_t *__restrict__ P, const int8_t *__restrict__ W,
                                                            const int32_t *__restrict__ S, const int8_t *__restrict__ in, int64_t n,
                                                            int64_t in_stride, int8_t *__restrict__ logits, int8_t *__restrict__ softmax,
                                                            int32_t *__restrict__ argmax, int8_t *__restrict__ acts)
{
	extern __shared__ __attribute__((aligned(16))) int8_t en_lds[];
	const int n_layers = P->n_layers, in_n = P->in_n, out_n = P->out_n, acts_bytes = P->acts_bytes;
	const int logits_layer = P->logits_layer, has_softmax = P->has_softmax;
	for (int64_t u = blockIdx.x; u < n; u += gridDim.x)
	{
		const int8_t *src = in + u * in_stride;
		for (int i = threadIdx.x; i < in_n; i += EN_THREADS) en_lds[i] = src[i]; /* layer 0 reads buffer 0 */
		__syncthreads();
		for (int li = 0; li < n_layers; li++)
		{
			const ed_net_layer_t L = P->L[li]; /* uniform: scalar loads */
			const int8_t *a = en_lds + L.in_buf;
			int8_t *o = en_lds + L.out_buf;
			if (L.type == ED_NET_CONV)
			{
				const bool padded = L.check_taps != 0; /* windows that stay inside the image need no tap tests */
				const bool c4 = (L.in_c & 3) == 0;
				if ((L.out_c & 3) == 0)
				{
					if (c4) { if (padded) en_conv_blocked<true, true>(L, a, o, W, S); else en_conv_blocked<true, false>(L, a, o, W, S); }
					else { if (padded) en_conv_blocked<false, true>(L, a, o, W, S); else en_conv_blocked<false, false>(L, a, o, W, S); }
				}
				else if (c4) { if (padded) en_conv<true, true>(L, a, o, W, S); else en_conv<true, false>(L, a, o, W, S); }
				else { if (padded) en_conv<false, true>(L, a, o, W, S); else en_conv<false, false>(L, a, o, W, S); }
			}
			else if (L.type == ED_NET_POOL) en_pool(L, a, o);
			else if (L.type == ED_NET_DENSE) en_dense(L, a, o, W, S);
			else en_softmax(L, a, o);
			__syncthreads();
			if (acts)
				for (int i = threadIdx.x; i < L.out_n; i += EN_THREADS) acts[u * acts_bytes + L.acts_off + i] = o[i];
			if (li == logits_layer && logits)
				for (int i = threadIdx.x; i < out_n; i += EN_THREADS) logits[u * out_n + i] = o[i];
			if (li == n_layers - 1)
			{
				if (has_softmax && softmax)
					for (int i = threadIdx.x; i < out_n; i += EN_THREADS) softmax[u * out_n + i] = o[i];
				if (argmax && threadIdx.x == 0)
				{
					int best = 0, mx = -129;
					for (int i = 0; i < out_n; i++)
						if (o[i] > mx) { mx = o[i]; best = i; }
					argmax[u] = best;
				}
			}
		}
		__syncthreads(); /* the next utterance overwrites both buffers */
	}
}

extern "C" int ed_launch_net(const ed_net_plan_t *dev_plan, const int8_t *dev_w, const int32_t *dev_seeds, int lds_bytes,
                             const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax, int32_t *argmax,
                             int8_t *acts, int n_cu, hipStream_t stream)
{
	if (n <= 0) return 0;
	int per_cu = lds_bytes > 0 ? (160 * 1024) / (lds_bytes + 256) : 8;
	if (per_cu > 8) per_cu = 8;
	if (per_cu < 1) per_cu = 1;
	int64_t blocks = n;
	if (blocks > (int64_t)n_cu * per_cu) blocks = (int64_t)n_cu * per_cu;
	hipLaunchKernelGGL(ed_net_kernel, dim3((unsigned)blocks), dim3(EN_THREADS), (size_t)lds_bytes, stream, dev_plan, dev_w,
	                   dev_seeds, in, n, in_stride, logits, softmax, argmax, acts);
	return (int)hipGetLastError();
}
