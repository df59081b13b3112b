/*
 * oracle/mfcc_ref.c -- TEST INFRASTRUCTURE (see oracle.h). Plain-C float64 restatement of the
 * reference's Python host MFCC, variants A and B:
 *
 *   frames                 audio/edison/mfcc/mfcc_utils.py:16-27
 *   hertz_to_mel           audio/edison/mfcc/mfcc_utils.py:30-34   (1127*ln(1+f/700), audio/config.py:35-36)
 *   gen_mel_weight_matrix  audio/edison/mfcc/mfcc_utils.py:36-73
 *   mfcc       (variant A) audio/edison/mfcc/mfcc_utils.py:134-199
 *   mfcc_mcu   (variant B) audio/edison/mfcc/mfcc_utils.py:255-323
 *   mfcc_tf    (variant TF) audio/edison/mfcc/mfcc_utils.py:201-253 -- PARITY UNPINNED, see oracle.h
 *
 * The FFT and the DCT live in third-party dependencies that are not under /root/reference
 * (numpy==1.18.2 np.fft.fft, scipy==1.4.1 scipy.fftpack.dct, audio/requirements.txt:33,58). They are
 * restated from their published definitions: the forward DFT X[k]=sum x[n] exp(-2*pi*i*n*k/N) and the
 * unnormalised DCT-II y[k]=2*sum x[n] cos(pi*k*(2n+1)/(2N)), both evaluated in float64. Parity is
 * anchored on the reference's own call sites through tests/golden/mfcc_golden.npz.
 *
 * Written for clarity, not speed: a full-length complex radix-2 FFT with imag=0 exactly like
 * np.fft.fft(chunk) on real input; the mel product is the dense dot the reference computes.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

#define MEL_HIGH_FREQUENCY_Q 1127.0      /* audio/config.py:35 */
#define MEL_BREAK_FREQUENCY_HERTZ 700.0  /* audio/config.py:36 */

int oracle_num_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

static double hertz_to_mel(double f) /* mfcc_utils.py:30-34 */
{
	return MEL_HIGH_FREQUENCY_Q * log(1.0 + (f / MEL_BREAK_FREQUENCY_HERTZ));
}

/* numpy.linspace(start, stop, num)[i]: start + i*step, the last sample pinned to stop. */
static double linspace_at(double start, double stop, int num, int i)
{
	if (num == 1) return start;
	if (i == num - 1) return stop;
	double step = (stop - start) / (double)(num - 1);
	return (double)i * step + start;
}

void oracle_mel_weight_matrix(int num_mel_bins, int num_spectrogram_bins, double sample_rate,
                              double lower_edge_hertz, double upper_edge_hertz, double *W)
{
	/* mfcc_utils.py:43-48: linear bins without DC, converted to mel */
	double nyquist = sample_rate / 2.0;
	double mlo = hertz_to_mel(lower_edge_hertz), mhi = hertz_to_mel(upper_edge_hertz);
	/* mfcc_utils.py:73: DC row re-added as zeros */
	for (int j = 0; j < num_mel_bins; j++) W[j] = 0.0;
	for (int i = 1; i < num_spectrogram_bins; i++)
	{
		double m = hertz_to_mel(linspace_at(0.0, nyquist, num_spectrogram_bins, i));
		for (int j = 0; j < num_mel_bins; j++)
		{
			/* mfcc_utils.py:54-60: (lower, center, upper) = consecutive triples of num_mel_bins+2 edges */
			double lo = linspace_at(mlo, mhi, num_mel_bins + 2, j);
			double ce = linspace_at(mlo, mhi, num_mel_bins + 2, j + 1);
			double up = linspace_at(mlo, mhi, num_mel_bins + 2, j + 2);
			double lower_slope = (m - lo) / (ce - lo);  /* :64-65 */
			double upper_slope = (up - m) / (up - ce);  /* :66-67 */
			double v = lower_slope < upper_slope ? lower_slope : upper_slope;
			W[(size_t)i * num_mel_bins + j] = v > 0.0 ? v : 0.0; /* :70 */
		}
	}
}

/* In-place iterative radix-2 DIT FFT, forward sign, float64. tw = exp(-2*pi*i*k/n), k<n/2. */
static void fft_c2c(double *re, double *im, int n, const double *twr, const double *twi)
{
	for (int i = 1, j = 0; i < n; i++)
	{
		int bit = n >> 1;
		for (; j & bit; bit >>= 1) j ^= bit;
		j ^= bit;
		if (i < j)
		{
			double t = re[i]; re[i] = re[j]; re[j] = t;
			t = im[i]; im[i] = im[j]; im[j] = t;
		}
	}
	for (int len = 2; len <= n; len <<= 1)
	{
		int half = len >> 1, step = n / len;
		for (int s = 0; s < n; s += len)
			for (int k = 0; k < half; k++)
			{
				double wr = twr[k * step], wi = twi[k * step];
				double xr = re[s + k + half], xi = im[s + k + half];
				double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
				re[s + k + half] = re[s + k] - tr; im[s + k + half] = im[s + k] - ti;
				re[s + k] += tr; im[s + k] += ti;
			}
	}
}

int oracle_mfcc(const int16_t *x, int64_t n_frames, int frame_len, int64_t frame_step, int variant,
                int num_mel_bins, double sample_rate, double lower_edge_hertz, double upper_edge_hertz,
                double mel_mtx_scale, int use_log,
                double *spectrogram, double *mel, double *logmel, double *mfcc, int n_threads)
{
	const int N = frame_len, nmel = num_mel_bins;
	if (N < 2 || (N & (N - 1)) != 0 || nmel < 1) return -1;
	if (variant != ORACLE_MFCC_VARIANT_A && variant != ORACLE_MFCC_VARIANT_B && variant != ORACLE_MFCC_VARIANT_TF) return -1;
	/* A: spectrogram = |fft|[:N/2], mel matrix built for N/2 bins (mfcc_utils.py:171-181)
	 * B: spectrogram = |fft/N|/sqrt2 over all N bins, mel uses the first N/2+1 (:297-308)
	 * TF: spectrogram = |rfft(hann * frame)| over the N/2+1 unique bins, all of them into the mel product (:218-231) */
	const int nspec_out = (variant == ORACLE_MFCC_VARIANT_A) ? N / 2 : (variant == ORACLE_MFCC_VARIANT_TF) ? N / 2 + 1 : N;
	const int nbins = (variant == ORACLE_MFCC_VARIANT_A) ? N / 2 : N / 2 + 1;

	double *W = (double *)malloc(sizeof(double) * (size_t)nbins * nmel);
	double *twr = (double *)malloc(sizeof(double) * (size_t)N / 2);
	double *twi = (double *)malloc(sizeof(double) * (size_t)N / 2);
	double *dct = (double *)malloc(sizeof(double) * (size_t)nmel * nmel);
	if (!W || !twr || !twi || !dct) { free(W); free(twr); free(twi); free(dct); return -2; }

	oracle_mel_weight_matrix(nmel, nbins, sample_rate, lower_edge_hertz, upper_edge_hertz, W);
	if (variant == ORACLE_MFCC_VARIANT_B) /* :282 mel_mtx_scale * gen_mel_weight_matrix(...) */
		for (size_t i = 0; i < (size_t)nbins * nmel; i++) W[i] *= mel_mtx_scale;
	for (int k = 0; k < N / 2; k++)
	{
		twr[k] = cos(-2.0 * M_PI * (double)k / (double)N);
		twi[k] = sin(-2.0 * M_PI * (double)k / (double)N);
	}
	/* scipy.fftpack.dct(type=2), norm=None: y[k] = 2 * sum_n x[n] cos(pi*k*(2n+1)/(2*nmel)) */
	for (int k = 0; k < nmel; k++)
		for (int n = 0; n < nmel; n++)
			dct[(size_t)k * nmel + n] = 2.0 * cos(M_PI * (double)k * (double)(2 * n + 1) / (double)(2 * nmel));
	/* A: / sqrt(2*mel_nbins) (:193); B: * 1.0/64 (:318, a literal 64 in the reference)
	 * TF: tf.signal.mfccs_from_log_mel_spectrograms = dct(type 2) * rsqrt(2 * num_mel_bins), the same scale as A */
	const double dct_div = (variant == ORACLE_MFCC_VARIANT_B) ? 64.0 : sqrt(2.0 * (double)nmel);

	int err = 0;
#ifdef _OPENMP
	if (n_threads < 1) n_threads = 1;
	#pragma omp parallel num_threads(n_threads)
#endif
	{
		double *re = (double *)malloc(sizeof(double) * (size_t)N);
		double *im = (double *)malloc(sizeof(double) * (size_t)N);
		double *sp = (double *)malloc(sizeof(double) * (size_t)N);
		double *me = (double *)malloc(sizeof(double) * (size_t)nmel);
		double *lm = (double *)malloc(sizeof(double) * (size_t)nmel);
		if (!re || !im || !sp || !me || !lm)
		{
			#pragma omp atomic write
			err = -2;
		}
		else
		{
#ifdef _OPENMP
			#pragma omp for schedule(static)
#endif
			for (int64_t f = 0; f < n_frames; f++)
			{
				const int16_t *chunk = x + f * frame_step; /* :168 / :293 */
				for (int i = 0; i < N; i++) { re[i] = (double)chunk[i]; im[i] = 0.0; }
				if (variant == ORACLE_MFCC_VARIANT_TF) /* tf.signal.hann_window(N, periodic=True), stft's default window_fn */
					for (int i = 0; i < N; i++) re[i] *= 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)N);
				fft_c2c(re, im, N, twr, twi);
				if (variant == ORACLE_MFCC_VARIANT_A)
				{
					for (int k = 0; k < N / 2; k++) sp[k] = hypot(re[k], im[k]); /* :171-174 */
				}
				else if (variant == ORACLE_MFCC_VARIANT_TF)
				{
					for (int k = 0; k <= N / 2; k++) sp[k] = hypot(re[k], im[k]); /* :222 tf.abs(stfts) */
				}
				else
				{
					/* :297 fft * 1/1024 -- a literal 1024 in the reference, equal to N at the shipped config */
					for (int k = 0; k < N; k++)
						sp[k] = 1.0 / sqrt(2.0) * hypot(re[k] * (1.0 / 1024), im[k] * (1.0 / 1024)); /* :300 */
				}
				for (int j = 0; j < nmel; j++) /* np.dot(spectrogram[:nbins], W)  :185 / :308 */
				{
					double acc = 0.0;
					for (int k = 0; k < nbins; k++) acc += sp[k] * W[(size_t)k * nmel + j];
					me[j] = acc;
				}
				if (variant == ORACLE_MFCC_VARIANT_B)
					for (int j = 0; j < nmel; j++) me[j] /= mel_mtx_scale; /* :309 */
				for (int j = 0; j < nmel; j++)
				{
					if (variant != ORACLE_MFCC_VARIANT_B) lm[j] = log(me[j] + 1e-6);  /* :189 / :233 */
					else lm[j] = use_log ? log(me[j] + 1e-6) : me[j];                 /* :313-315 */
				}
				if (spectrogram) memcpy(spectrogram + (size_t)f * nspec_out, sp, sizeof(double) * (size_t)nspec_out);
				if (mel) memcpy(mel + (size_t)f * nmel, me, sizeof(double) * (size_t)nmel);
				if (logmel) memcpy(logmel + (size_t)f * nmel, lm, sizeof(double) * (size_t)nmel);
				if (mfcc)
					for (int k = 0; k < nmel; k++)
					{
						double acc = 0.0;
						for (int n = 0; n < nmel; n++) acc += lm[n] * dct[(size_t)k * nmel + n];
						mfcc[(size_t)f * nmel + k] = (variant != ORACLE_MFCC_VARIANT_B) ? acc / dct_div
						                                                                : 1.0 / dct_div * acc;
					}
			}
		}
		free(re); free(im); free(sp); free(me); free(lm);
	}
	free(W); free(twr); free(twi); free(dct);
	return err;
}

void oracle_net_input(const double *mfcc, int64_t n_rows, int stride, int n_coef, double scale,
                      double clip_lo, double clip_hi, int8_t *out)
{
	/* kws_nnom.py:359-361: float32 array * quantise_factor -> np.clip -> .round() (half to even) -> int8 */
	for (int64_t r = 0; r < n_rows; r++)
		for (int c = 0; c < n_coef; c++)
		{
			float v = (float)mfcc[(size_t)r * stride + c] * (float)scale;
			if (v < (float)clip_lo) v = (float)clip_lo;
			if (v > (float)clip_hi) v = (float)clip_hi;
			out[(size_t)r * n_coef + c] = (int8_t)rintf(v);
		}
}
