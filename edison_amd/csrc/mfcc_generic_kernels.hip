/*
 * mfcc_generic_kernels.hip -- MFCC variants A, B and TF for ANY geometry the reference's Python functions accept: frame_len (any length,
 * also no power of two: the reference calls numpy.fft.fft), mel_nbins, filterbank edges, sample rate
 * (audio/edison/mfcc/mfcc_utils.py:134-199 `mfcc`, :255-323 `mfcc_mcu`, :75-131 `batch_mfcc`).
 *
 * The hot path (mfcc_kernels.hip) is built around the one geometry every caller in the reference passes (audio/config.py: 1024-sample
 * frames, 32 bins); this kernel is the generality path behind the same Python mirror, not a throughput kernel: one 256-thread workgroup
 * per frame, float64 throughout (gfx950's vector fp64 runs at half the fp32 rate; the reference computes in float64, so the outputs are
 * the reference's to ~1e-12 and no tolerance discussion is needed), the transform as a direct DFT -- exact for every length, N^2 / 2
 * multiply-adds per frame (0.5 M for N = 1024: microseconds) -- against a cos / sin table built on the host, then |X|, the dense mel
 * product, ln, DCT-II with the variant's constants:
 *   A (:170-196)  X = fft(x)[:N/2]; s = |X|; e = s . W(N/2 bins); l = ln(e + 1e-6); mfcc = dct2(l) / sqrt(2 * mel_nbins)
 *   B (:296-319)  X = fft(x) / 1024 (the constant, whatever N is); s = |X| / sqrt(2); e = (s[:N/2+1] . (scale * W(N/2+1 bins))) / scale;
 *                 l = use_log ? ln(e + 1e-6) : e; mfcc = dct2(l) / 64 (the constant)
 *   TF (:201-253) x = float32(x) * hann_periodic_float32; X = rfft(x) (N/2+1 bins); s = |X|; e = s . W(N/2+1 bins); l = ln(e + 1e-6);
 *                 mfcc = dct2(l) / sqrt(2 * mel_nbins) (tf.signal.mfccs_from_log_mel_spectrograms). fft_len == frame_len. PARITY UNPINNED (no TensorFlow here).
 * Outputs are float64 like the reference's dict entries: fft [n][N/2 (A) | N (B) | N/2+1 (TF)][2], spectrogram likewise, mel / logmel / mfcc [n][mel_nbins].
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "edison_internal.h"

__global__ __launch_bounds__(256) void ed_mfcc_generic_kernel(ed_mfcc_gen_args_t a)
{
	extern __shared__ __attribute__((aligned(16))) double gsm[];
	const int N = a.frame_len, nb = a.n_bins, nm = a.n_mel, t = threadIdx.x;
	double *x = gsm;                 /* [N]      samples                         */
	double *cs = x + N;              /* [N][2]   cos, sin of 2 pi j / N          */
	double *re = cs + 2 * N;         /* [N/2+1]  X[k] of the non-redundant half  */
	double *im = re + (N / 2 + 1);
	double *sp = im + (N / 2 + 1);   /* [N/2+1]  spectrum (scaled)               */
	double *me = sp + (N / 2 + 1);   /* [nm]     mel / log-mel                   */
	for (int j = t; j < 2 * N; j += 256) cs[j] = a.tw[j];
	for (int64_t f = blockIdx.x; f < a.n_frames; f += gridDim.x)
	{
		const int16_t *fp = a.audio + f * a.frame_step;
		__syncthreads();
		/* variant TF: the samples become float32, the window multiplies them in float32 (tf.signal.stft on a float32 tensor), the rest is float64 */
		if (a.window)
			for (int n = t; n < N; n += 256) x[n] = (double)__fmul_rn((float)fp[n], a.window[n]);
		else
			for (int n = t; n < N; n += 256) x[n] = (double)fp[n];
		__syncthreads();
		/* direct DFT of the real frame, bins 0 .. N/2: X[k] = sum_n x[n] (cos(2 pi k n / N) - i sin(2 pi k n / N)); the table index
		 * k n mod N is kept by addition */
		for (int k = t; k <= N / 2; k += 256)
		{
			double sr = 0.0, si = 0.0;
			int j = 0;
			for (int n = 0; n < N; n++)
			{
				const double v = x[n];
				sr = fma(v, cs[2 * j], sr);
				si = fma(-v, cs[2 * j + 1], si);
				j += k;
				if (j >= N) j -= N;
			}
			re[k] = sr * a.fft_scale;
			im[k] = si * a.fft_scale;
			sp[k] = sqrt(re[k] * re[k] + im[k] * im[k]) * a.spec_scale;
		}
		__syncthreads();
		/* the reference's dict entries: A keeps bins 0 .. N/2-1, B all N (the upper half is the conjugate mirror of a real frame) */
		if (a.fft)
			for (int k = t; k < a.fft_out; k += 256)
			{
				const int m = k <= N / 2 ? k : N - k;
				double *o = a.fft + ((size_t)f * a.fft_out + k) * 2;
				o[0] = re[m];
				o[1] = k <= N / 2 ? im[m] : -im[m];
			}
		if (a.spec)
			for (int k = t; k < a.fft_out; k += 256) a.spec[(size_t)f * a.fft_out + k] = sp[k <= N / 2 ? k : N - k];
		/* mel: e[j] = sum_k s[k] W[k][j] over the variant's nb bins (W is row-major [bin][mel]: consecutive threads, consecutive weights) */
		for (int j = t; j < nm; j += 256)
		{
			double e = 0.0;
			for (int k = 0; k < nb; k++) e = fma(sp[k], a.W[(size_t)k * nm + j], e);
			e = e / a.mel_div;
			if (a.mel) a.mel[(size_t)f * nm + j] = e;
			const double l = a.take_log ? log(e + 1e-6) : e;
			if (a.logmel) a.logmel[(size_t)f * nm + j] = l;
			me[j] = l;
		}
		__syncthreads();
		/* DCT-II, scipy's unnormalised definition y[c] = 2 sum_n l[n] cos(pi c (2 n + 1) / (2 nm)), over the variant's divisor */
		for (int c = t; c < nm; c += 256)
		{
			double y = 0.0;
			for (int n = 0; n < nm; n++) y = fma(me[n], a.dct[(size_t)c * nm + n], y);
			y = y / a.dct_div;
			if (a.mfcc) a.mfcc[(size_t)f * nm + c] = y;
			if (a.feat && c < a.n_coef)
			{
				/* kws_nnom.py:359-361: float32 array * scale -> clip -> round half even -> int8 */
				float v = (float)y * a.feat_scale;
				v = fminf(fmaxf(v, -128.0f), 127.0f);
				a.feat[(size_t)f * a.n_coef + c] = (int8_t)__float2int_rn(v);
			}
		}
	}
}

extern "C" int ed_launch_mfcc_generic(const ed_mfcc_gen_args_t *a, int n_cu, hipStream_t stream)
{
	if (a->n_frames <= 0) return 0;
	const int N = a->frame_len;
	if (N < 2 || N > ED_GEN_MAX_FRAME || a->n_mel < 1 || a->n_mel > ED_GEN_MAX_MEL) return (int)hipErrorInvalidValue;
	const size_t lds = sizeof(double) * ((size_t)N + 2 * (size_t)N + 3 * ((size_t)N / 2 + 1) + (size_t)a->n_mel);
	static bool attr_set[16];
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	if (!attr_set[dev_ & 15])
	{
		if (hipFuncSetAttribute((const void *)ed_mfcc_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * (3 * ED_GEN_MAX_FRAME + 3 * (ED_GEN_MAX_FRAME / 2 + 1) + ED_GEN_MAX_MEL))) != hipSuccess)
			return (int)hipGetLastError();
		attr_set[dev_ & 15] = true;
	}
	int64_t blocks = a->n_frames;
	const int64_t cap = (int64_t)n_cu * 2;
	if (blocks > cap) blocks = cap;
	hipLaunchKernelGGL(ed_mfcc_generic_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, *a);
	return (int)hipGetLastError();
}
