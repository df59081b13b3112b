/*
 * oracle/mfcc_f32_ref.c -- TEST INFRASTRUCTURE. CPU restatement of the firmware's float32 MFCC ("variant D"),
 * the ARM ML-KWS feature extractor that the dormant NNoM example uses (firmware/src/app.c:497-623).
 *
 * What it follows (reference file:line):
 *   mfcc_create        firmware/src/audio/mfcc.c:47-84   (frame_len padded to a power of two, Hann window in float)
 *   create_dct_matrix  firmware/src/audio/mfcc.c:102-117 (sqrt(2/N) cos(pi/N (n+0.5) k), float)
 *   create_mel_fbank   firmware/src/audio/mfcc.c:119-172 (26 triangular filters 20..4000 Hz, weights linear in mel)
 *   mfcc_compute       firmware/src/audio/mfcc.c:174-255 (pre-emphasis, window, real FFT, power, sqrt, mel, FLT_MIN
 *                      guard, logf, DCT, * 2^dec_bits, round(), saturate to q7)
 *   MelScale           firmware/src/audio/mfcc.h:53-55
 *
 * Float32 everywhere the reference is float32, in the reference's operation order. The FFT is the exception: the
 * reference calls CMSIS arm_rfft_fast_f32, whose tables (arm_common_tables.c) are missing from the snapshot; here the
 * transform is evaluated in double and rounded to float, i.e. at least as accurate as any float32 FFT.
 *
 * PARITY STATUS: pinned (round 3). The reference's own mfcc_compute with CMSIS-DSP's float transform compiled under it
 * (oracle/_ref/libmfcc_f32_ref.so, ref_shim/mfcc_f32_ref_shim.c; the transform's table values regenerated, its bit-reversal
 * list derived from the routine) answers the same frames: int8 equal but for rounding-boundary values, log-mel within 1e-3
 * clear of the float32 rounding floor (tests/test_oracle_refpins.py, fixture tests/golden/mfccf32_golden.npz). In the
 * reference itself the path is switched off (#if'd "NNoM example") and feeds a 63 x 12 input that the shipped 31 x 13
 * network does not accept.
 */
/* no fused multiply-adds: mfcc.c is compiled at -O0 for the MCU (firmware/Makefile:42) and plain C rounds every operation;
 * with -march=x86-64-v3 gcc would contract a * b + c (tests/test_oracle_refpins.py: tables bit-identical to the reference's) */
#pragma GCC optimize("fp-contract=off")

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define F32_NUM_FBANK 26
#define F32_SAMP_FREQ 16000
#define F32_MEL_LOW 20
#define F32_MEL_HIGH 4000

struct oracle_f32_mfcc {
	int n_features, offset, frame_len, padded, dec_bits;
	float preempha;
	float *window;
	int first[F32_NUM_FBANK], last[F32_NUM_FBANK];
	float *fbank[F32_NUM_FBANK];
	float *dct;
};

static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

oracle_f32_mfcc_t *oracle_f32_mfcc_new(int num_mfcc_features, int feature_offset, int frame_len, int mfcc_dec_bits, float preempha)
{
	if (frame_len < 2 || frame_len > 4096 || num_mfcc_features < 1 || num_mfcc_features > F32_NUM_FBANK ||
	    feature_offset < 0 || feature_offset >= num_mfcc_features)
		return NULL;
	oracle_f32_mfcc_t *m = (oracle_f32_mfcc_t *)calloc(1, sizeof(*m));
	if (!m) return NULL;
	m->n_features = num_mfcc_features; m->offset = feature_offset; m->frame_len = frame_len;
	m->dec_bits = mfcc_dec_bits; m->preempha = preempha;
	m->padded = (int)powf(2, ceilf(logf((float)frame_len) / logf(2)));
	m->window = (float *)malloc(sizeof(float) * (size_t)frame_len);
	for (int i = 0; i < frame_len; i++)
		m->window[i] = 0.5f - 0.5f * cosf((float)6.283185307179586476925286766559005 * ((float)i) / (frame_len));
	/* mel filterbank */
	const int nbins = m->padded / 2;
	const float bin_width = ((float)F32_SAMP_FREQ) / m->padded;
	const float lo = mel_scale(F32_MEL_LOW), hi = mel_scale(F32_MEL_HIGH);
	const float delta = (hi - lo) / (F32_NUM_FBANK + 1);
	float *tmp = (float *)malloc(sizeof(float) * (size_t)nbins);
	for (int b = 0; b < F32_NUM_FBANK; b++)
	{
		const float left = lo + b * delta, center = lo + (b + 1) * delta, right = lo + (b + 2) * delta;
		int first = -1, last = -1;
		for (int i = 0; i < nbins; i++)
		{
			const float mel = mel_scale(bin_width * i);
			tmp[i] = 0.0f;
			if (mel > left && mel < right)
			{
				tmp[i] = mel <= center ? (mel - left) / (center - left) : (right - mel) / (right - center);
				if (first == -1) first = i;
				last = i;
			}
		}
		m->first[b] = first; m->last[b] = last;
		const int n = (first >= 0) ? last - first + 1 : 0;
		m->fbank[b] = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
		for (int i = 0; i < n; i++) m->fbank[b][i] = tmp[first + i];
	}
	free(tmp);
	/* DCT matrix */
	m->dct = (float *)malloc(sizeof(float) * F32_NUM_FBANK * (size_t)num_mfcc_features);
	const float normalizer = sqrtf(2.0f / (float)F32_NUM_FBANK);
	for (int k = 0; k < num_mfcc_features; k++)
		for (int n = 0; n < F32_NUM_FBANK; n++)
			m->dct[k * F32_NUM_FBANK + n] = normalizer * cosf(((float)3.14159265358979323846264338327950288) / F32_NUM_FBANK * (n + 0.5f) * k);
	return m;
}

void oracle_f32_mfcc_free(oracle_f32_mfcc_t *m)
{
	if (!m) return;
	free(m->window); free(m->dct);
	for (int b = 0; b < F32_NUM_FBANK; b++) free(m->fbank[b]);
	free(m);
}

int oracle_f32_mfcc_n_out(const oracle_f32_mfcc_t *m) { return m->n_features - m->offset; }

/* the restated tables, for the comparison with the reference's own create_dct_matrix / create_mel_fbank (oracle/_ref/
 * libmfcc_f32_ref.so): dct [n_features][26]; first / last [26]; weights = the rows back to back; returns their count */
int oracle_f32_mfcc_tables_get(const oracle_f32_mfcc_t *m, float *dct, int32_t *first, int32_t *last, float *weights, int cap)
{
	if (!m) return -1;
	if (dct) memcpy(dct, m->dct, sizeof(float) * F32_NUM_FBANK * (size_t)m->n_features);
	int pos = 0;
	for (int b = 0; b < F32_NUM_FBANK; b++)
	{
		const int n = m->last[b] - m->first[b] + 1;
		if (first) first[b] = m->first[b];
		if (last) last[b] = m->last[b];
		if (weights)
		{
			if (pos + n > cap) return -1;
			memcpy(weights + pos, m->fbank[b], sizeof(float) * (size_t)n);
		}
		pos += n;
	}
	return pos;
}

/* radix-2 FFT in double, natural order in/out */
static void fft_double(double *re, double *im, int n)
{
	for (int i = 1, j = 0; i < n; i++)
	{
		int bit = n >> 1;
		for (; j & bit; bit >>= 1) j ^= bit;
		j ^= bit;
		if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
	}
	for (int len = 2; len <= n; len <<= 1)
	{
		const double ang = -2.0 * M_PI / len;
		for (int i = 0; i < n; i += len)
			for (int k = 0; k < len / 2; k++)
			{
				const double wr = cos(ang * k), wi = sin(ang * k);
				const int a = i + k, b = i + k + len / 2;
				const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
				re[b] = re[a] - xr; im[b] = im[a] - xi;
				re[a] += xr; im[a] += xi;
			}
	}
}

static void frame_f32(const oracle_f32_mfcc_t *m, const int16_t *audio, int8_t *out, float *out_f32, float *mel_out)
{
	const int N = m->frame_len, P = m->padded, half = P / 2;
	float frame[4096];
	double re[4096], im[4096];
	float power[2049], mel[F32_NUM_FBANK];
	/* 1./2. normalise and pre-emphasise (mfcc.c:178-185): element 0 is left unscaled, the window zeroes it */
	float last = (float)audio[0];
	frame[0] = last;
	for (int i = 1; i < N; i++)
	{
		frame[i] = ((float)audio[i] - last * m->preempha) / (1 << 15);
		last = (float)audio[i];
	}
	for (int i = N; i < P; i++) frame[i] = 0.0f;
	for (int i = 0; i < N; i++) frame[i] *= m->window[i];
	for (int i = 0; i < P; i++) { re[i] = frame[i]; im[i] = 0.0; }
	fft_double(re, im, P);
	/* power spectrum (mfcc.c:196-206) */
	power[0] = (float)re[0] * (float)re[0];
	power[half] = (float)re[half] * (float)re[half];
	for (int i = 1; i < half; i++)
	{
		const float r = (float)re[i], q = (float)im[i];
		power[i] = r * r + q * q;
	}
	/* mel filterbank on the magnitudes, FLT_MIN guard, log (mfcc.c:208-232) */
	for (int b = 0; b < F32_NUM_FBANK; b++)
	{
		float e = 0;
		if (m->first[b] >= 0)
			for (int i = m->first[b], j = 0; i <= m->last[b]; i++) e += sqrtf(power[i]) * m->fbank[b][j++];
		if (e == 0.0f) e = FLT_MIN;
		mel[b] = logf(e);
		if (mel_out) mel_out[b] = mel[b];
	}
	/* DCT, scale, round half away from zero, saturate (mfcc.c:234-254) */
	int o = 0;
	for (int i = m->offset; i < m->n_features; i++, o++)
	{
		float sum = 0.0f;
		for (int j = 0; j < F32_NUM_FBANK; j++) sum += m->dct[i * F32_NUM_FBANK + j] * mel[j];
		sum *= (float)(0x1 << m->dec_bits);
		if (out_f32) out_f32[o] = sum;
		sum = (float)round((double)sum);
		out[o] = sum >= 127 ? 127 : (sum <= -128 ? -128 : (int8_t)sum);
	}
}

int oracle_f32_mfcc_run(const oracle_f32_mfcc_t *m, const int16_t *x, int64_t n_frames, int64_t frame_step, int8_t *out,
                        float *out_f32, float *logmel, int n_threads)
{
	if (!m || !x || !out) return -1;
	const int n_out = m->n_features - m->offset;
	if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(static)
	for (int64_t f = 0; f < n_frames; f++)
		frame_f32(m, x + f * frame_step, out + f * n_out, out_f32 ? out_f32 + f * n_out : NULL,
		          logmel ? logmel + f * F32_NUM_FBANK : NULL);
	return 0;
}
