/*
 * cnn_kernels.hip -- the int8 keyword-spotting CNN (NNoM "kws_conv": 4 x Conv2D+ReLU, 2 x MaxPool(2,1),
 * Dense, Softmax) for gfx950, hand-written HIP. Bit-exact with the reference's CPU path:
 *
 *   conv   arm_convolve_HWC_q7_basic_nonsquare.c:188-221 / arm_convolve_HWC_q7_fast_nonsquare.c (portable):
 *          out = ssat8((sum x*w + (bias << bias_shift) + NN_ROUND(out_shift)) >> out_shift), OHWI weights
 *   relu   arm_relu_q7.c:57-105 (in place, NNoM tail activation nnom.c:986-989)
 *   pool   local_maxpool_q7_HWC nnom_local.c:117-159, kernel (2,1) stride (2,1) VALID
 *   dense  arm_fully_connected_q7_opt.c:374-473 (weights de-interleaved at import time)
 *   softmax arm_softmax_q7.c:215-260 (portable branch)      argmax nnom_utils.c:275-284 (first maximum)
 *
 * One 256-thread workgroup walks utterances in a persistent loop. All weights (42.8 KB) are staged once per
 * workgroup in LDS as dwords [k/4][out_channel] (4 consecutive K bytes per dword) so a wavefront's weight
 * read is 64 consecutive dwords (conflict-free) and v_dot4_i32_i8 consumes 4 MACs per lane per instruction.
 * Activations of the utterance in flight (<= 3888 B per layer) never leave LDS.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "edison_internal.h"

#define ED_CNN_THREADS 256

struct ed_cnn_smem
{
	ed_cnn_model_t m;                                  /* weights + biases + shifts                       */
	__attribute__((aligned(16))) int8_t in[416];       /* 31x13x1 (+ pad so dword reads stay in bounds)   */
	__attribute__((aligned(16))) int8_t c1[3888 + 16]; /* conv1+relu 27x9x16                              */
	__attribute__((aligned(16))) int8_t p1[1872 + 16]; /* pool1 13x9x16                                   */
	__attribute__((aligned(16))) int8_t c2[2464 + 16]; /* conv2+relu 11x7x32                              */
	__attribute__((aligned(16))) int8_t p2[1120 + 16]; /* pool2 5x7x32                                    */
	__attribute__((aligned(16))) int8_t c3[960 + 16];  /* conv3+relu 3x5x64                               */
	__attribute__((aligned(16))) int8_t c4[96 + 16];   /* conv4+relu 1x3x32                               */
	__attribute__((aligned(16))) int8_t fc[16];        /* dense logits                                    */
	__attribute__((aligned(16))) int8_t sm[16];        /* softmax                                         */
};

__device__ __forceinline__ int ed_ssat8(int v) { return v > 127 ? 127 : (v < -128 ? -128 : v); }
__device__ __forceinline__ int ed_usat(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

/* VALID conv, stride 1, Cin % 4 == 0, O a power of two; weights w[K/4][O] dwords; ReLU fused. */
template <int H, int W, int C, int KH, int KW, int O>
__device__ __forceinline__ void ed_conv_c4(const int8_t *__restrict__ in, const int32_t *__restrict__ w,
                                           const int32_t *__restrict__ bias, int rshift, int8_t *__restrict__ out)
{
	constexpr int OH = H - KH + 1, OW = W - KW + 1, C4 = C / 4;
	const int32_t *in4 = reinterpret_cast<const int32_t *>(in);
	for (int e = threadIdx.x; e < OH * OW * O; e += ED_CNN_THREADS)
	{
		const int o = e % O, pix = e / O;
		const int y = pix / OW, x = pix % OW;
		int acc = bias[o];
#pragma unroll
		for (int ky = 0; ky < KH; ky++)
		{
			const int32_t *row = in4 + ((y + ky) * W + x) * C4;   /* KW*C contiguous bytes */
			const int32_t *wr = w + (ky * KW * C4) * O + o;
#pragma unroll 4
			for (int j = 0; j < KW * C4; j++) acc = __builtin_amdgcn_sdot4(row[j], wr[j * O], acc, false);
		}
		int v = ed_ssat8(acc >> rshift);
		out[e] = (int8_t)(v < 0 ? 0 : v);
	}
}

/* conv1: Cin = 1, 5x5, K = 25 padded to 28. Bytes are gathered from the 31x13 input. */
__device__ __forceinline__ void ed_conv1(const int8_t *__restrict__ in, const int32_t *__restrict__ w,
                                         const int32_t *__restrict__ bias, int rshift, int8_t *__restrict__ out)
{
	constexpr int OH = 27, OW = 9, O = ED_C1_O;
	for (int e = threadIdx.x; e < OH * OW * O; e += ED_CNN_THREADS)
	{
		const int o = e % O, pix = e / O;
		const int y = pix / OW, x = pix % OW;
		int acc = bias[o];
#pragma unroll
		for (int k4 = 0; k4 < 7; k4++)
		{
			uint32_t packed = 0;
#pragma unroll
			for (int b = 0; b < 4; b++)
			{
				const int k = 4 * k4 + b;
				if (k < 25)
				{
					const int ky = k / 5, kx = k % 5;
					packed |= (uint32_t)(uint8_t)in[(y + ky) * ED_IN_W + x + kx] << (8 * b);
				}
			}
			acc = __builtin_amdgcn_sdot4((int)packed, w[k4 * O + o], acc, false);
		}
		int v = ed_ssat8(acc >> rshift);
		out[e] = (int8_t)(v < 0 ? 0 : v);
	}
}

/* MaxPool kernel (2,1) stride (2,1) VALID on HWC: out[y][x][c] = max(in[2y][x][c], in[2y+1][x][c]); 4 ch / thread */
template <int H, int W, int C>
__device__ __forceinline__ void ed_pool21(const int8_t *__restrict__ in, int8_t *__restrict__ out)
{
	constexpr int OH = (H - 2) / 2 + 1, ROW4 = W * C / 4;
	const int32_t *in4 = reinterpret_cast<const int32_t *>(in);
	int32_t *out4 = reinterpret_cast<int32_t *>(out);
	for (int e = threadIdx.x; e < OH * ROW4; e += ED_CNN_THREADS)
	{
		const int y = e / ROW4, j = e % ROW4;
		const uint32_t a = (uint32_t)in4[(2 * y) * ROW4 + j], b = (uint32_t)in4[(2 * y + 1) * ROW4 + j];
		uint32_t r = 0;
#pragma unroll
		for (int s = 0; s < 32; s += 8)
		{
			const int va = (int8_t)(a >> s), vb = (int8_t)(b >> s);
			r |= (uint32_t)(uint8_t)(va > vb ? va : vb) << s;
		}
		out4[e] = (int32_t)r;
	}
}

template <bool LAYERS>
__global__ __launch_bounds__(ED_CNN_THREADS) void ed_cnn_kernel(const ed_cnn_model_t *__restrict__ model,
                                                                const int8_t *__restrict__ feat, int64_t n_utt,
                                                                int8_t *__restrict__ logits,
                                                                int8_t *__restrict__ softmax,
                                                                int32_t *__restrict__ argmax,
                                                                int8_t *__restrict__ acts)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	ed_cnn_smem &s = *reinterpret_cast<ed_cnn_smem *>(smem_raw);

	{ /* stage the model once per workgroup */
		const int32_t *src = reinterpret_cast<const int32_t *>(model);
		int32_t *dst = reinterpret_cast<int32_t *>(&s.m);
		for (int i = threadIdx.x; i < (int)(sizeof(ed_cnn_model_t) / 4); i += ED_CNN_THREADS) dst[i] = src[i];
		if (threadIdx.x < 13) s.in[403 + threadIdx.x] = 0;
	}
	__syncthreads();

	for (int64_t u = blockIdx.x; u < n_utt; u += gridDim.x)
	{
		const int8_t *fin = feat + u * 403;
		for (int i = threadIdx.x; i < 403; i += ED_CNN_THREADS) s.in[i] = fin[i];
		__syncthreads();
		ed_conv1(s.in, &s.m.w1[0][0], s.m.b1, s.m.rs1, s.c1);
		__syncthreads();
		ed_pool21<27, 9, 16>(s.c1, s.p1);
		__syncthreads();
		ed_conv_c4<13, 9, 16, 3, 3, ED_C2_O>(s.p1, &s.m.w2[0][0], s.m.b2, s.m.rs2, s.c2);
		__syncthreads();
		ed_pool21<11, 7, 32>(s.c2, s.p2);
		__syncthreads();
		ed_conv_c4<5, 7, 32, 3, 3, ED_C3_O>(s.p2, &s.m.w3[0][0], s.m.b3, s.m.rs3, s.c3);
		__syncthreads();
		ed_conv_c4<3, 5, 64, 3, 3, ED_C4_O>(s.c3, &s.m.w4[0][0], s.m.b4, s.m.rs4, s.c4);
		__syncthreads();
		if (threadIdx.x < ED_FC_O)
		{
			const int32_t *x4 = reinterpret_cast<const int32_t *>(s.c4);
			int acc = s.m.bfc[threadIdx.x];
#pragma unroll
			for (int j = 0; j < ED_FC_I / 4; j++) acc = __builtin_amdgcn_sdot4(x4[j], s.m.wfc[j][threadIdx.x], acc, false);
			s.fc[threadIdx.x] = (int8_t)ed_ssat8(acc >> s.m.rsfc);
		}
		__syncthreads();
		if (threadIdx.x == 0)
		{
			/* arm_softmax_q7 portable branch, then first-max argmax over its output */
			int base = -128;
			for (int i = 0; i < ED_FC_O; i++) base = s.fc[i] > base ? s.fc[i] : base;
			base -= 8;
			int sum = 0;
			for (int i = 0; i < ED_FC_O; i++) sum += 1 << ed_usat(s.fc[i] - base, 7);
			const int output_base = (1 << 20) / sum;
			int best = 0, mx = -129;
			for (int i = 0; i < ED_FC_O; i++)
			{
				const int v = ed_ssat8(output_base >> ed_usat(13 + base - s.fc[i], 31));
				s.sm[i] = (int8_t)v;
				if (v > mx) { mx = v; best = i; }
			}
			if (argmax) argmax[u] = best;
		}
		__syncthreads();
		if (threadIdx.x < ED_FC_O)
		{
			if (logits) logits[u * ED_FC_O + threadIdx.x] = s.fc[threadIdx.x];
			if (softmax) softmax[u * ED_FC_O + threadIdx.x] = s.sm[threadIdx.x];
		}
		if (LAYERS && acts)
		{
			int8_t *a = acts + u * ED_CNN_ACT_BYTES;
			for (int i = threadIdx.x; i < 3888; i += ED_CNN_THREADS) a[i] = s.c1[i];
			for (int i = threadIdx.x; i < 1872; i += ED_CNN_THREADS) a[3888 + i] = s.p1[i];
			for (int i = threadIdx.x; i < 2464; i += ED_CNN_THREADS) a[5760 + i] = s.c2[i];
			for (int i = threadIdx.x; i < 1120; i += ED_CNN_THREADS) a[8224 + i] = s.p2[i];
			for (int i = threadIdx.x; i < 960; i += ED_CNN_THREADS) a[9344 + i] = s.c3[i];
			for (int i = threadIdx.x; i < 96; i += ED_CNN_THREADS) a[10304 + i] = s.c4[i];
			if (threadIdx.x < 10) { a[10400 + threadIdx.x] = s.fc[threadIdx.x]; a[10410 + threadIdx.x] = s.sm[threadIdx.x]; }
		}
		__syncthreads();
	}
}

extern "C" int ed_launch_cnn(const ed_cnn_model_t *dev_model, const int8_t *feat, int64_t n_utt, int8_t *logits,
                             int8_t *softmax, int32_t *argmax, int8_t *acts, int n_cu, hipStream_t stream)
{
	if (n_utt <= 0) return 0;
	const size_t lds = sizeof(ed_cnn_smem);
	int64_t blocks = n_utt;
	const int64_t cap = (int64_t)n_cu * 2;
	if (blocks > cap) blocks = cap;
	dim3 grid((unsigned)blocks), block(ED_CNN_THREADS);
	if (acts)
		hipLaunchKernelGGL(ed_cnn_kernel<true>, grid, block, lds, stream, dev_model, feat, n_utt, logits, softmax,
		                   argmax, acts);
	else
		hipLaunchKernelGGL(ed_cnn_kernel<false>, grid, block, lds, stream, dev_model, feat, n_utt, logits, softmax,
		                   argmax, acts);
	return (int)hipGetLastError();
}
