/*
 * mfcc_f32_kernels.hip -- MFCC variant D on gfx950: the firmware's float32 ML-KWS extractor
 * (mfcc_compute, firmware/src/audio/mfcc.c:174-255), one wavefront per frame.
 *
 *   1. (audio[i] - audio[i-1] * preempha) / 2^15, Hann window, zero padding          (mfcc.c:178-193)
 *   2. FFT of the padded frame: radix-2 in LDS on bit-reversed input (the firmware's arm_rfft_fast_f32 tables are
 *      not in the reference snapshot; any float32 FFT differs from it by rounding only)
 *   3. |X[k]| = sqrtf(re^2 + im^2), k = 0..padded/2                                   (mfcc.c:196-206,218)
 *   4. 26 mel bands, FLT_MIN when a band is exactly zero, logf                         (mfcc.c:208-232)
 *   5. DCT rows feature_offset..num_features-1, * 2^dec_bits, round half away, saturate to q7 (mfcc.c:234-254)
 *
 * This is the path of the reference's dormant NNoM example (app.c:497-623, frame 512, hop 256, 12 features); it is
 * built for completeness of the call surface (mfcc_create / mfcc_compute), not tuned: HBM traffic per frame is
 * 2 * frame_len bytes in, <= 26 bytes out.
 */
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

#define EF_WPB 4

__device__ __forceinline__ void ef_wave_sync() { __builtin_amdgcn_wave_barrier(); }

__global__ __launch_bounds__(64 * EF_WPB) void ed_mfcc_f32_kernel(ed_mfcc_f32_args_t a, const ed_f32_tables_t *__restrict__ T)
{
	extern __shared__ __attribute__((aligned(16))) float2 s_buf[]; /* [EF_WPB][padded]: sized by the launcher */
	__shared__ float2 s_tw[ED_F32_MAX_FRAME / 2];
	__shared__ float s_lm[EF_WPB][32];
	__shared__ float s_melw[ED_F32_MAX_W];
	__shared__ float s_dct[ED_F32_NUM_FBANK * ED_F32_NUM_FBANK];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int N = T->frame_len, P = T->padded, L2 = T->log2p, half = P >> 1;
	const int n_out = T->n_features - T->offset;
	const float preempha = T->preempha, scale = T->scale;
	for (int i = threadIdx.x; i < half; i += 64 * EF_WPB) s_tw[i] = make_float2(T->tw[i][0], T->tw[i][1]);
	/* mel weights and DCT rows are read serially by few lanes (the firmware's summation order is kept): from LDS,
	 * with the loads of several taps in flight, not one global load per dependent add */
	for (int i = threadIdx.x; i < ED_F32_MAX_W; i += 64 * EF_WPB) s_melw[i] = T->mel_w[i];
	for (int i = threadIdx.x; i < ED_F32_NUM_FBANK * ED_F32_NUM_FBANK; i += 64 * EF_WPB) s_dct[i] = T->dct[i];
	__syncthreads();
	float2 *buf = s_buf + w * P;
	float *mag = reinterpret_cast<float *>(buf); /* magnitudes overwrite the transform in place (k <= P/2 < P floats) */
	float *lm = s_lm[w];

	for (int64_t f = (int64_t)blockIdx.x * EF_WPB + w; f < a.n_frames; f += (int64_t)gridDim.x * EF_WPB)
	{
		const int16_t *x = a.audio + f * a.frame_step;
		/* 1. pre-emphasis, window, padding; stored bit-reversed for the in-place decimation-in-time FFT */
		for (int i = lane; i < P; i += 64)
		{
			float v = 0.0f;
			if (i < N)
			{
				v = (i == 0) ? (float)x[0] : ((float)x[i] - (float)x[i - 1] * preempha) / 32768.0f;
				v *= T->window[i];
			}
			buf[__builtin_bitreverse32((unsigned)i) >> (32 - L2)] = make_float2(v, 0.0f);
		}
		ef_wave_sync();
		/* 2. log2(P) radix-2 stages */
		for (int s = 0; s < L2; s++)
		{
			const int hs = 1 << s;
			for (int t = lane; t < half; t += 64)
			{
				const int j = t & (hs - 1);
				const int i0 = ((t >> s) << (s + 1)) | j, i1 = i0 + hs;
				const float2 wv = s_tw[j << (L2 - 1 - s)];
				const float2 p = buf[i0], q = buf[i1];
				const float xr = q.x * wv.x - q.y * wv.y, xi = q.x * wv.y + q.y * wv.x;
				buf[i1] = make_float2(p.x - xr, p.y - xi);
				buf[i0] = make_float2(p.x + xr, p.y + xi);
			}
			ef_wave_sync();
		}
		/* 3. magnitudes of bins 0..P/2; every lane reads its bins before any lane overwrites (in-order LDS, and
		 *    bin k is written at float k which only aliases complex slots k/2 <= k) */
		float m[ED_F32_MAX_FRAME / 128 + 1];
#pragma unroll
		for (int c = 0; c < ED_F32_MAX_FRAME / 128 + 1; c++)
		{
			const int k = lane + 64 * c;
			m[c] = 0.0f;
			if (k <= half)
			{
				const float2 v = buf[k];
				m[c] = sqrtf(v.x * v.x + v.y * v.y);
			}
		}
		ef_wave_sync();
#pragma unroll
		for (int c = 0; c < ED_F32_MAX_FRAME / 128 + 1; c++)
			if (lane + 64 * c <= half) mag[lane + 64 * c] = m[c];
		ef_wave_sync();
		/* 4. mel bands and log */
		if (lane < ED_F32_NUM_FBANK)
		{
			float e = 0.0f;
			const int first = T->mel_first[lane], last = T->mel_last[lane];
			const float *wgt = s_melw + T->mel_off[lane];
			if (first >= 0)
			{
#pragma unroll 8
				for (int i = first; i <= last; i++) e += mag[i] * wgt[i - first];
			}
			if (e == 0.0f) e = FLT_MIN;
			const float l = logf(e);
			lm[lane] = l;
			if (a.logmel) a.logmel[f * ED_F32_NUM_FBANK + lane] = l;
		}
		ef_wave_sync();
		/* 5. DCT rows, scale, round half away from zero, saturate */
		if (lane < n_out)
		{
			const float *row = s_dct + (T->offset + lane) * ED_F32_NUM_FBANK;
			float sum = 0.0f;
#pragma unroll
			for (int j = 0; j < ED_F32_NUM_FBANK; j++) sum += row[j] * lm[j];
			sum *= scale;
			if (a.out_f32) a.out_f32[f * n_out + lane] = sum;
			const float r = roundf(sum);
			a.out[f * n_out + lane] = (int8_t)(r >= 127.0f ? 127 : (r <= -128.0f ? -128 : (int)r));
		}
		ef_wave_sync();
	}
}

extern "C" int ed_launch_mfcc_f32(const ed_mfcc_f32_args_t *args, const ed_f32_tables_t *dev_tab, int padded, int n_cu, hipStream_t stream)
{
	if (args->n_frames <= 0) return 0;
	int64_t blocks = (args->n_frames + EF_WPB - 1) / EF_WPB;
	const size_t lds = (size_t)EF_WPB * (size_t)padded * sizeof(float2);
	int per_cu = (int)((160u * 1024u) / (lds + 12u * 1024u)); /* + the static tables */
	if (per_cu > 8) per_cu = 8;
	if (per_cu < 1) per_cu = 1;
	const int64_t cap = (int64_t)n_cu * per_cu;
	if (blocks > cap) blocks = cap;
	hipLaunchKernelGGL(ed_mfcc_f32_kernel, dim3((unsigned)blocks), dim3(64 * EF_WPB), lds, stream, *args, dev_tab);
	return (int)hipGetLastError();
}

/* ================================================================================================================
 * Fast path for frames padded to 512 points (the firmware's configuration: frame 512, hop 256, app.c:497,583).
 *
 * The 512-point real transform of a frame is taken as a 512-point COMPLEX transform with a zero imaginary part on the
 * register-level radix-8 machinery of the variant A / B kernel (mfcc_fft.h): three radix-8 passes, the first digit
 * transpose on the VALU (v_permlane swaps + DPP), the second through a wave-private LDS buffer; no real-FFT split is
 * needed, lane l ends with X[l + 64 r] in register r and bins 0..256 are registers 0..3 (+ register 4 of lane 0).
 * Two frames ride in the two halves of packed fp32 registers (v_pk_*_f32): independent lanes of arithmetic, so a silent
 * frame next to a loud one stays exactly silent (which pairing two real frames into ONE complex transform would not
 * give). Frames are handed to the 16 waves of a CU-wide workgroup through a counter in LDS, as in ed_mfcc2_kernel.
 * Mel: two lanes per band (first / second half of the band's taps), log, DCT row per lane, scale, round half away,
 * saturate -- float32 like the firmware; the order of the partial sums differs from its serial loops, inside the bars
 * tests/test_gpu_f32.py states (the variant is pinned on the reference's compiled mfcc_compute + CMSIS transform: tests/golden/mfccf32_golden.npz).
 */
#include "mfcc_fft.h"

/* ---- lab knobs: only a lab build (ED_LAB, tools/lab/mkvariant.py) may set them; the product build has none, and
 * tests/test_host_cpu.py checks the values below against what edison_amd/build.py compiles */
#if !defined(ED_LAB) && (defined(EF2_WPB) || defined(EF2_PRIO))
#error "EF2_* lab knob defined without ED_LAB (tools/lab/mkvariant.py builds lab variants)"
#endif
#if defined(ED_LAB)
/* a lab build says so: the product library exports no ed_lab_build_* symbol (tests/test_host_cpu.py) */
extern "C" { extern const int ed_lab_build_mfcc_f32; const int ed_lab_build_mfcc_f32 = 1; }
#endif
#ifndef EF2_WPB
#define EF2_WPB 16 /* 4 waves per SIMD at 128 VGPRs: +4.5 % over 12 once the wave priorities are in (1.33-1.37 -> 1.40-1.41 G frames/s) */
#endif
#define EF2_S_OFF 1088     /* float offset of the interleaved spectra S2[k] = (|X_A[k]|, |X_B[k]|), k = 0..256 */
#define EF2_L_OFF 1664     /* log-mel energies of both frames, float2[32] */
#define EF2_XBUF_FLOATS 2208
#define EF2_TAB_FLOATS (2 * 7 * 64 * 2 + ED_F32_MAX_W + ED_F32_NUM_FBANK * ED_F32_NUM_FBANK + 4) /* twiddles | mel_w | dct */

/* wave priority rising with the progress through a pair (see ED2_PRIO in mfcc_kernels.hip): 0 in pass 1, 1 in pass 2, 2 in pass 3,
 * 3 from the magnitudes to the next pair's loads. EF2_PRIO=0: none (A/B) */
#ifndef EF2_PRIO
#define EF2_PRIO 1
#endif
#if EF2_PRIO
#define EF2_PR(p) __builtin_amdgcn_s_setprio(p)
#else
#define EF2_PR(p) ((void)0)
#endif
__global__ __launch_bounds__(64 * EF2_WPB) void ed_mfcc_f32_fast_kernel(ed_mfcc_f32_args_t a, const ed_f32_tables_t *__restrict__ T,
                                                                      const ed_mfcc_tables_t *__restrict__ F)
{
	extern __shared__ __attribute__((aligned(16))) float ef_smem[];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	float2 *twl = reinterpret_cast<float2 *>(ef_smem);                /* [2][7][64]: pass-1 / pass-2 twiddles per lane */
	float *melw = ef_smem + 2 * 7 * 64 * 2;
	float *dctl = melw + ED_F32_MAX_W;
	float *xbuf = ef_smem + EF2_TAB_FLOATS + wave * EF2_XBUF_FLOATS;  /* wave-private */
	unsigned *queue = reinterpret_cast<unsigned *>(ef_smem + EF2_TAB_FLOATS + EF2_WPB * EF2_XBUF_FLOATS);

	const int N = T->frame_len, n_out = T->n_features - T->offset;
	const float preempha = T->preempha, scale = T->scale;
	const uint32_t n_frames = (uint32_t)a.n_frames, n_pairs = (n_frames + 1) >> 1;
	const uint32_t s0 = (uint32_t)(((uint64_t)blockIdx.x * n_pairs) / gridDim.x);
	const uint32_t cnt = (uint32_t)(((uint64_t)(blockIdx.x + 1) * n_pairs) / gridDim.x) - s0;

	/* per-lane constants of the 8 samples n = lane + 64 a this lane feeds: window / 2^15 (the firmware scales before it
	 * windows; the power of two commutes exactly), the pre-emphasis factor (0 for n = 0, which the firmware passes
	 * through unscaled: mfcc.c:178-181), and the sample offsets clamped into the frame */
	float ws[8], pe[8];
	int off0[8], off1[8];
#pragma unroll
	for (int q = 0; q < 8; q++)
	{
		const int n = lane + 64 * q;
		const bool in = n < N;
		ws[q] = !in ? 0.0f : (n == 0 ? T->window[0] : T->window[n] * (1.0f / 32768.0f));
		pe[q] = (in && n > 0) ? preempha : 0.0f;
		off0[q] = in ? n : 0;
		off1[q] = (in && n > 0) ? n - 1 : 0;
	}
	for (int t = threadIdx.x; t < 7 * 64; t += 64 * EF2_WPB)
	{
		twl[t] = *reinterpret_cast<const float2 *>(&F->tw1[1 + t / 64][t & 63][0]);
		twl[7 * 64 + t] = *reinterpret_cast<const float2 *>(&F->tw2[1 + t / 64][t & 63][0]);
	}
	for (int t = threadIdx.x; t < ED_F32_MAX_W; t += 64 * EF2_WPB) melw[t] = T->mel_w[t];
	for (int t = threadIdx.x; t < ED_F32_NUM_FBANK * ED_F32_NUM_FBANK; t += 64 * EF2_WPB) dctl[t] = T->dct[t];
	if (threadIdx.x == 0) *queue = 2 * EF2_WPB;
	/* mel: lane (band = lane & 31, part = lane >> 5) sums one half of its band's taps */
	const int band = lane & 31, part = lane >> 5;
	int t_first = 0, t_cnt = 0, t_woff = 0;
	if (band < ED_F32_NUM_FBANK && T->mel_first[band] >= 0)
	{
		const int first = T->mel_first[band], last = T->mel_last[band], n_t = last - first + 1, h0 = (n_t + 1) >> 1;
		t_first = part ? first + h0 : first;
		t_cnt = part ? n_t - h0 : h0;
		t_woff = T->mel_off[band] + (part ? h0 : 0);
	}
	const int dct_row = (T->offset + (lane < n_out ? lane : 0)) * ED_F32_NUM_FBANK;
	__syncthreads();
	const float2 *tw1l = twl + lane, *tw2l = twl + 7 * 64 + lane;
	const int hi3 = lane >> 3, lo3 = lane & 7;
	float4 *xc4 = reinterpret_cast<float4 *>(xbuf);
	ed_f2 *S2 = reinterpret_cast<ed_f2 *>(xbuf + EF2_S_OFF);
	ed_f2 *LM = reinterpret_cast<ed_f2 *>(xbuf + EF2_L_OFF);

	auto load_pair = [&](uint32_t pr, int (&xa)[8], int (&ya)[8], int (&xb)[8], int (&yb)[8]) {
		const uint32_t fA = 2 * pr, fB = fA + 1 < n_frames ? fA + 1 : fA;
		const int16_t *pa = a.audio + (int64_t)fA * a.frame_step, *pb = a.audio + (int64_t)fB * a.frame_step;
#pragma unroll
		for (int q = 0; q < 8; q++) { xa[q] = pa[off0[q]]; ya[q] = pa[off1[q]]; xb[q] = pb[off0[q]]; yb[q] = pb[off1[q]]; }
	};

	uint32_t i_cur = wave, i_next = wave + EF2_WPB;
	int xa[8], ya[8], xb[8], yb[8];
	if (i_cur < cnt) load_pair(s0 + i_cur, xa, ya, xb, yb);
	while (i_cur < cnt)
	{
		const uint32_t fA = 2 * (s0 + i_cur);
		const bool haveB = fA + 1 < n_frames;
		/* ---- 1. pre-emphasis, scale, window; the imaginary parts start as zero */
		ed_f2 re[8], im[8];
#pragma unroll
		for (int q = 0; q < 8; q++)
		{
			const float va = __fsub_rn((float)xa[q], __fmul_rn((float)ya[q], pe[q])), vb = __fsub_rn((float)xb[q], __fmul_rn((float)yb[q], pe[q]));
			re[q] = ed_mk2(va * ws[q], vb * ws[q]);
			im[q] = ed_splat(0.0f);
		}
		load_pair(s0 + (i_next < cnt ? i_next : cnt - 1), xa, ya, xb, yb); /* unconditional, see ed_mfcc2_kernel */
		uint32_t drawn = 0;
		if (lane == 0) drawn = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

		EF2_PR(0);
		/* ---- 2. 512-point complex FFT: pass 1 + twiddles, transpose 1 (VALU), pass 2 + twiddles, transpose 2 (LDS), pass 3 */
		ed_radix8_2(re, im);
#pragma unroll
		for (int q = 1; q < 8; q++)
		{
			const float2 w_ = tw1l[64 * (q - 1)];
			const ed_f2 wr = ed_splat(w_.x), wi = ed_splat(w_.y), xr = re[q], xi = im[q];
			re[q] = xr * wr - xi * wi;
			im[q] = xr * wi + xi * wr;
		}
		EF2_PR(1);
		ed_transpose8_2<3, 4, 5>(re, lane);
		ed_transpose8_2<3, 4, 5>(im, lane);
		ed_radix8_2(re, im);
#pragma unroll
		for (int q = 1; q < 8; q++)
		{
			const float2 w_ = tw2l[64 * (q - 1)];
			const ed_f2 wr = ed_splat(w_.x), wi = ed_splat(w_.y), xr = re[q], xi = im[q];
			re[q] = xr * wr - xi * wi;
			im[q] = xr * wi + xi * wr;
		}
		EF2_PR(2);
#pragma unroll
		for (int q = 0; q < 8; q++) xc4[66 * lo3 + hi3 + 8 * q] = make_float4(re[q].x, re[q].y, im[q].x, im[q].y);
		ed_wave_sync();
#pragma unroll
		for (int c = 0; c < 8; c++)
		{
			const float4 v = xc4[66 * c + lane];
			re[c] = ed_mk2(v.x, v.y); im[c] = ed_mk2(v.z, v.w);
		}
		ed_wave_sync();
		ed_radix8_2(re, im);

		EF2_PR(3);
		/* ---- 3. |X[k]|, k = lane + 64 r (r < 4), and k = 256 (lane 0, r = 4): sqrtf(re^2 + im^2) (mfcc.c:196-206) */
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
			const ed_f2 e = re[r] * re[r] + im[r] * im[r];
			S2[lane + 64 * r] = ed_mk2(__fsqrt_rn(e.x), __fsqrt_rn(e.y));
		}
		if (lane == 0)
		{
			const ed_f2 e = re[4] * re[4] + im[4] * im[4];
			S2[256] = ed_mk2(__fsqrt_rn(e.x), __fsqrt_rn(e.y));
		}
		ed_wave_sync();

		/* ---- 4. mel bands: two lanes per band, each over half of the taps; FLT_MIN for an exactly empty band, logf */
		ed_f2 acc = ed_splat(0.0f);
		for (int i = 0; i < t_cnt; i++) acc = ed_fma2(S2[t_first + i], ed_splat(melw[t_woff + i]), acc);
		ed_f2 e = ed_mk2(ed_sum_halves(acc.x), ed_sum_halves(acc.y));
		if (e.x == 0.0f) e.x = FLT_MIN;
		if (e.y == 0.0f) e.y = FLT_MIN;
		const ed_f2 lg = ed_mk2(logf(e.x), logf(e.y));
		/* the lane number is made opaque for the two output stages: the 64-bit per-lane output addresses derived from it would
		 * otherwise be hoisted out of the loop and, at 128 registers (4 waves per SIMD), live in scratch */
		int ln = lane;
		asm volatile("" : "+v"(ln));
		if (ln < ED_F32_NUM_FBANK)
		{
			LM[ln] = lg;
			if (a.logmel)
			{
				a.logmel[(int64_t)fA * ED_F32_NUM_FBANK + ln] = lg.x;
				if (haveB) a.logmel[(int64_t)(fA + 1) * ED_F32_NUM_FBANK + ln] = lg.y;
			}
		}
		ed_wave_sync();

		/* ---- 5. DCT rows, scale, round half away from zero, saturate to q7 (mfcc.c:234-254) */
		if (ln < n_out)
		{
			ed_f2 sum = ed_splat(0.0f);
#pragma unroll
			for (int j = 0; j < ED_F32_NUM_FBANK; j++) sum = ed_fma2(ed_splat(dctl[dct_row + j]), LM[j], sum);
			sum = sum * ed_splat(scale);
			const float ra = roundf(sum.x), rb = roundf(sum.y);
			const int64_t oa = (int64_t)fA * n_out + ln;
			a.out[oa] = (int8_t)(ra >= 127.0f ? 127 : (ra <= -128.0f ? -128 : (int)ra));
			if (a.out_f32) a.out_f32[oa] = sum.x;
			if (haveB)
			{
				a.out[oa + n_out] = (int8_t)(rb >= 127.0f ? 127 : (rb <= -128.0f ? -128 : (int)rb));
				if (a.out_f32) a.out_f32[oa + n_out] = sum.y;
			}
		}
		ed_wave_sync(); /* S2 / LM are rewritten by the next pair */
		i_cur = i_next; i_next = __builtin_amdgcn_readfirstlane(drawn);
	}
}

extern "C" int ed_launch_mfcc_f32_fast(const ed_mfcc_f32_args_t *args, const ed_f32_tables_t *dev_tab, const ed_mfcc_tables_t *dev_fft_tab,
                                       int n_cu, hipStream_t stream)
{
	if (args->n_frames <= 0) return 0;
	const size_t lds = sizeof(float) * (EF2_TAB_FLOATS + EF2_WPB * EF2_XBUF_FLOATS) + 16;
	static int ready_dev[16]; /* per device: the attribute belongs to the function on the current device */
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	int &ready = ready_dev[dev_ & 15];
	if (!ready)
	{
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ed_mfcc_f32_fast_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return (int)e;
		ready = 1;
	}
	const int64_t n_pairs = (args->n_frames + 1) / 2;
	int64_t blocks = (n_pairs + EF2_WPB - 1) / EF2_WPB;
	if (blocks > n_cu) blocks = n_cu;
	hipLaunchKernelGGL(ed_mfcc_f32_fast_kernel, dim3((unsigned)blocks), dim3(64 * EF2_WPB), lds, stream, *args, dev_tab, dev_fft_tab);
	return (int)hipGetLastError();
}
