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
