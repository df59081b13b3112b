/*
 * mfcc_kernels.hip -- batched per-frame MFCC for gfx950 (CDNA4), hand-written HIP.
 *
 * Computes what the reference computes one frame at a time in Python/numpy float64
 *   variant A  audio/edison/mfcc/mfcc_utils.py:160-197   variant B  audio/edison/mfcc/mfcc_utils.py:287-322
 * in fp32, one 64-lane wavefront per 1024-sample frame, everything between the int16 load and the 13
 * output coefficients kept in registers/LDS (no intermediate HBM traffic: 2048 B in, <=128 B out per frame).
 *
 * Pipeline per wavefront (lane = 0..63):
 *   1. load      z[n] = x[2n] + i*x[2n+1], n = lane + 64a (a = 0..7): 8 coalesced 256-B wave loads
 *   2. FFT512    3 radix-8 passes over the digits of n = 64a + 8b + c, k = p + 8q + 64r; the two digit
 *                transposes go through a wave-private, padded LDS buffer (conflict-free ds_*_b64)
 *   3. split     X[k] = E[k] + W1024^k O[k] from Z[k], conj Z[512-k]; lane handles the pair (k, 512-k)
 *   4. |X|       -> wave-private LDS spectrum S[0..512]
 *   5. mel       32 banded dot products: lane (band j = lane&31, half h = lane>>5) walks its taps,
 *                halves combined with one cross-lane shuffle
 *   6. ln / DCT  optional ln(x+1e-6); DCT-II from 16 per-lane table registers, halves combined by shuffle
 *   7. store     n_coef fp32 and/or int8 (clip, round-half-even) per frame
 *
 * Waves never share LDS data, so there is no workgroup barrier inside the frame loop.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "edison_internal.h"

#define ED_WAVES_PER_BLOCK 4
#define ED_XBUF_FLOATS 1160 /* per-wave LDS: 576 complex exchange slots (also Pz + S) + 8 pad */

__device__ __forceinline__ void ed_wave_sync()
{
	/* Order this wave's LDS writes before its following LDS reads. A wave's DS instructions execute in
	 * order; the fence only stops the compiler from moving them across each other. */
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void ed_dft4(float y0r, float y0i, float y1r, float y1i, float y2r, float y2i, float y3r,
                                        float y3i, float &o0r, float &o0i, float &o1r, float &o1i, float &o2r,
                                        float &o2i, float &o3r, float &o3i)
{
	float a0r = y0r + y2r, a0i = y0i + y2i;
	float a1r = y0r - y2r, a1i = y0i - y2i;
	float a2r = y1r + y3r, a2i = y1i + y3i;
	float a3r = y1i - y3i, a3i = y3r - y1r; /* (y1 - y3) * (-i) */
	o0r = a0r + a2r; o0i = a0i + a2i;
	o2r = a0r - a2r; o2i = a0i - a2i;
	o1r = a1r + a3r; o1i = a1i + a3i;
	o3r = a1r - a3r; o3i = a1i - a3i;
}

/* In-place 8-point forward DFT, natural order in and out. */
__device__ __forceinline__ void ed_radix8(float (&r)[8], float (&i)[8])
{
	const float h = 0.70710678118654752440f;
	float ur[4], ui[4], vr[4], vi[4];
#pragma unroll
	for (int a = 0; a < 4; a++)
	{
		ur[a] = r[a] + r[a + 4]; ui[a] = i[a] + i[a + 4];
		vr[a] = r[a] - r[a + 4]; vi[a] = i[a] - i[a + 4];
	}
	float t;
	t = vr[1]; vr[1] = h * (vr[1] + vi[1]); vi[1] = h * (vi[1] - t);   /* * (1 - i)/sqrt2  */
	t = vr[2]; vr[2] = vi[2]; vi[2] = -t;                               /* * (-i)           */
	t = vr[3]; vr[3] = h * (vi[3] - vr[3]); vi[3] = -h * (vi[3] + t);   /* * (-1 - i)/sqrt2 */
	ed_dft4(ur[0], ui[0], ur[1], ui[1], ur[2], ui[2], ur[3], ui[3], r[0], i[0], r[2], i[2], r[4], i[4], r[6], i[6]);
	ed_dft4(vr[0], vi[0], vr[1], vi[1], vr[2], vi[2], vr[3], vi[3], r[1], i[1], r[3], i[3], r[5], i[5], r[7], i[7]);
}

template <bool STAGES>
__global__ __launch_bounds__(64 * ED_WAVES_PER_BLOCK) void ed_mfcc_kernel(ed_mfcc_args_t args,
                                                                          const ed_mfcc_tables_t *__restrict__ tab)
{
	extern __shared__ __attribute__((aligned(16))) float smem[];
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int T = tab->mel_T;
	float *melw = smem;                                   /* [T][64] shared by the block      */
	float *xbuf = smem + ED_MEL_T_MAX * 64 + wave * ED_XBUF_FLOATS; /* wave-private               */
	float2 *xc = reinterpret_cast<float2 *>(xbuf);

	for (int t = threadIdx.x; t < T * 64; t += blockDim.x) melw[t] = (&tab->mel_w[0][0])[t];
	__syncthreads();

	/* per-lane constants, resident in registers for the whole persistent loop */
	float t1r[8], t1i[8], t2r[8], t2i[8], tpr[4], tpi[4], dct[16];
#pragma unroll
	for (int p = 1; p < 8; p++)
	{
		t1r[p] = tab->tw1[lane][p][0]; t1i[p] = tab->tw1[lane][p][1];
		t2r[p] = tab->tw2[lane & 7][p][0]; t2i[p] = tab->tw2[lane & 7][p][1];
	}
#pragma unroll
	for (int m = 0; m < 4; m++) { tpr[m] = tab->twp[m][lane][0]; tpi[m] = tab->twp[m][lane][1]; }
#pragma unroll
	for (int n = 0; n < 16; n++) dct[n] = tab->dct[n][lane];
	const int mel_start = tab->mel_start[lane];
	const float spec_scale = tab->spec_scale;
	const float log_offset = tab->log_offset;
	const bool do_log = tab->always_log || args.use_log;

	const int hi3 = lane >> 3, lo3 = lane & 7;

	for (int64_t f = (int64_t)blockIdx.x * ED_WAVES_PER_BLOCK + wave; f < args.n_frames;
	     f += (int64_t)gridDim.x * ED_WAVES_PER_BLOCK)
	{
		const int64_t g = f / args.frames_per_group;
		const int64_t start = g * args.group_stride + (f - g * args.frames_per_group) * args.frame_step;
		const int16_t *fp = args.audio + start;

		/* ---- 1. load: lane gets z[lane + 64a] */
		float re[8], im[8];
		if ((reinterpret_cast<uintptr_t>(fp) & 3) == 0)
		{
			const uint32_t *fp32 = reinterpret_cast<const uint32_t *>(fp);
			uint32_t v[8];
#pragma unroll
			for (int a = 0; a < 8; a++) v[a] = fp32[lane + 64 * a];
#pragma unroll
			for (int a = 0; a < 8; a++)
			{
				re[a] = (float)(int16_t)(v[a] & 0xffffu);
				im[a] = (float)(int16_t)(v[a] >> 16);
			}
		}
		else
		{
#pragma unroll
			for (int a = 0; a < 8; a++)
			{
				re[a] = (float)fp[2 * (lane + 64 * a)];
				im[a] = (float)fp[2 * (lane + 64 * a) + 1];
			}
		}

		/* ---- 2a. pass 1: DFT over a, twiddle W512^(lane*p) */
		ed_radix8(re, im);
#pragma unroll
		for (int p = 1; p < 8; p++)
		{
			float xr = re[p], xi = im[p];
			re[p] = xr * t1r[p] - xi * t1i[p];
			im[p] = xr * t1i[p] + xi * t1r[p];
		}
		/* transpose 1: (lane = 8b+c, reg p) -> (lane = 8p+c, reg b); slot = 72p + 8b + c */
#pragma unroll
		for (int p = 0; p < 8; p++) xc[72 * p + lane] = make_float2(re[p], im[p]);
		ed_wave_sync();
#pragma unroll
		for (int b = 0; b < 8; b++)
		{
			float2 v = xc[72 * hi3 + 8 * b + lo3];
			re[b] = v.x; im[b] = v.y;
		}
		ed_wave_sync();

		/* ---- 2b. pass 2: DFT over b, twiddle W64^(c*q) */
		ed_radix8(re, im);
#pragma unroll
		for (int q = 1; q < 8; q++)
		{
			float xr = re[q], xi = im[q];
			re[q] = xr * t2r[q] - xi * t2i[q];
			im[q] = xr * t2i[q] + xi * t2r[q];
		}
		/* transpose 2: (lane = 8p+c, reg q) -> (lane = p+8q, reg c); slot = 66c + p + 8q */
#pragma unroll
		for (int q = 0; q < 8; q++) xc[66 * lo3 + hi3 + 8 * q] = make_float2(re[q], im[q]);
		ed_wave_sync();
#pragma unroll
		for (int c = 0; c < 8; c++)
		{
			float2 v = xc[66 * c + lane];
			re[c] = v.x; im[c] = v.y;
		}
		ed_wave_sync();

		/* ---- 2c. pass 3: DFT over c  ->  reg r holds Z[lane + 64r] */
		ed_radix8(re, im);

		/* ---- 3. real-FFT split. Partner buffer Pz[j] = Z[256 + j] (regs 4..7), read back reversed. */
#pragma unroll
		for (int r = 4; r < 8; r++) xc[lane + 64 * (r - 4)] = make_float2(re[r], im[r]);
		ed_wave_sync();
		float slo[4], shi[4];
		float flr[4], fli[4], fhr[4], fhi[4]; /* X2[k], X2[512-k] for the stage dump */
#pragma unroll
		for (int m = 0; m < 4; m++)
		{
			float2 pz = xc[256 - lane - 64 * m]; /* Z[512 - k]; slot 256 (lane 0, m 0) is Z[512] = Z[0] */
			if (m == 0 && lane == 0) pz = make_float2(re[0], im[0]);
			float ar = re[m] + pz.x, ai = im[m] - pz.y; /* A  = Z[k] + conj Z[512-k]          = 2 E[k]      */
			float br = re[m] - pz.x, bi = im[m] + pz.y; /* B  = Z[k] - conj Z[512-k]; O2 = -i*B = 2 O[k]    */
			float tr = tpr[m] * bi + tpi[m] * br;       /* T  = W1024^k * (bi - i*br)                        */
			float ti = tpi[m] * bi - tpr[m] * br;
			float xr = ar + tr, xi = ai + ti;           /* 2 X[k]                                            */
			float yr = ar - tr, yi = ai - ti;           /* conj(2 X[512-k])                                  */
			slo[m] = __fsqrt_rn(xr * xr + xi * xi) * spec_scale;
			shi[m] = __fsqrt_rn(yr * yr + yi * yi) * spec_scale;
			if (STAGES) { flr[m] = 0.5f * xr; fli[m] = 0.5f * xi; fhr[m] = 0.5f * yr; fhi[m] = -0.5f * yi; }
		}
		/* k = 256 pairs with itself: X[256] = conj(Z[256]) (lane 0, reg 4) */
		const float s256 = 2.0f * __fsqrt_rn(re[4] * re[4] + im[4] * im[4]) * spec_scale;

		/* ---- 4. spectrum to LDS (floats 576..1088 of the wave buffer: disjoint from Pz) */
		float *S = xbuf + 576;
#pragma unroll
		for (int m = 0; m < 4; m++)
		{
			S[lane + 64 * m] = slo[m];
			S[512 - lane - 64 * m] = shi[m];
		}
		if (lane == 0) S[256] = s256;
		if (STAGES)
		{
			if (args.fft)
			{
				float2 *F = reinterpret_cast<float2 *>(args.fft) + f * 513;
#pragma unroll
				for (int m = 0; m < 4; m++)
				{
					F[lane + 64 * m] = make_float2(flr[m], fli[m]);
					F[512 - lane - 64 * m] = make_float2(fhr[m], fhi[m]);
				}
				if (lane == 0) F[256] = make_float2(re[4], -im[4]);
			}
		}
		ed_wave_sync();
		if (STAGES && args.spec)
		{
			for (int k = lane; k < 513; k += 64) args.spec[f * 513 + k] = S[k];
		}

		/* ---- 5. mel filterbank: lane (band, half) walks its taps */
		float acc = 0.0f;
		for (int t = 0; t < T; t++) acc = fmaf(S[mel_start + t], melw[t * 64 + lane], acc);
		float e = acc + __shfl_xor(acc, 32);
		float lm = do_log ? logf(e + log_offset) : e;
		if (STAGES && lane < 32)
		{
			if (args.mel) args.mel[f * 32 + lane] = e;
			if (args.logmel) args.logmel[f * 32 + lane] = lm;
		}

		/* ---- 6. DCT-II: lane (c = lane&31, h) sums n = 16h..16h+15 */
		float *Lb = xbuf + 1104; /* 32 floats, 16-B aligned */
		if (lane < 32) Lb[lane] = lm;
		ed_wave_sync();
		const float4 *L4 = reinterpret_cast<const float4 *>(Lb + 16 * (lane >> 5));
		float d = 0.0f;
#pragma unroll
		for (int n4 = 0; n4 < 4; n4++)
		{
			float4 v = L4[n4];
			d = fmaf(v.x, dct[4 * n4 + 0], d);
			d = fmaf(v.y, dct[4 * n4 + 1], d);
			d = fmaf(v.z, dct[4 * n4 + 2], d);
			d = fmaf(v.w, dct[4 * n4 + 3], d);
		}
		d += __shfl_xor(d, 32);
		ed_wave_sync(); /* Lb / S are rewritten by the next frame */

		/* ---- 7. store */
		if (lane < args.n_coef)
		{
			if (args.mfcc) args.mfcc[f * args.n_coef + lane] = d;
			if (args.feat)
			{
				float q = d * args.feat_scale;
				q = fminf(fmaxf(q, -128.0f), 127.0f);
				args.feat[f * args.n_coef + lane] = (int8_t)__float2int_rn(q);
			}
		}
	}
}

extern "C" int ed_launch_mfcc(const ed_mfcc_args_t *args, const ed_mfcc_tables_t *dev_tab, int stages, int n_cu,
                              hipStream_t stream)
{
	if (args->n_frames <= 0) return 0;
	const size_t lds = sizeof(float) * (ED_MEL_T_MAX * 64 + ED_WAVES_PER_BLOCK * ED_XBUF_FLOATS);
	int64_t blocks = (args->n_frames + ED_WAVES_PER_BLOCK - 1) / ED_WAVES_PER_BLOCK;
	const int64_t cap = (int64_t)n_cu * 8;
	if (blocks > cap) blocks = cap;
	dim3 grid((unsigned)blocks), block(64 * ED_WAVES_PER_BLOCK);
	if (stages)
		hipLaunchKernelGGL(ed_mfcc_kernel<true>, grid, block, lds, stream, *args, dev_tab);
	else
		hipLaunchKernelGGL(ed_mfcc_kernel<false>, grid, block, lds, stream, *args, dev_tab);
	return (int)hipGetLastError();
}
