/*
 * oracle/mfcc_q15_ref.c -- TEST INFRASTRUCTURE. CPU restatement of the firmware's fixed-point MFCC ("variant C").
 *
 * What it follows (reference file:line):
 *   audioCalcMFCCs                   firmware/src/audioprocessing.c:116-215  (complex-FFT branch: USE_REAL_FFT is off, :51;
 *                                    compact mel matrix: USE_MEL_MTX_COMPACT is on, :60)
 *   cmpl_mag_q15                     firmware/src/audioprocessing.c:299-312
 *   dct2_q15                         firmware/src/audioprocessing.c:330-436  (even/odd reorder, RFFT, real parts; no weights)
 *   arm_cfft_q15                     lib/CMSIS/DSP/Source/TransformFunctions/arm_cfft_q15.c:695-745
 *   arm_radix4_butterfly_q15         .../arm_cfft_radix4_q15.c:147-563   (the ARM_MATH_DSP branch: the firmware is built
 *                                    with -DARM_MATH_CM4 -DARM_MATH_DSP, firmware/Makefile:30)
 *   arm_rfft_q15, arm_split_rfft_q15 .../arm_rfft_q15.c:76-123, 241-325  (ARM_MATH_DSP branch)
 *   arm_sqrt_q31                     lib/CMSIS/DSP/Source/FastMathFunctions/arm_sqrt_q31.c:50-139
 *   mel_constants.h generator        audio/edison/mfcc/mfcc_on_mcu.py:26-145 (int16(128*W), per-band first index + count)
 *
 * The packed-halfword DSP instructions are written out per component; complex values are (re, im) int16 pairs.
 *
 * THIRD-PARTY TABLES: CMSIS-DSP's arm_common_tables.c (twiddleCoef_1024_q15, twiddleCoef_16_q15, realCoefAQ15,
 * realCoefBQ15, the bit-reversal index tables) is ABSENT from the reference snapshot. The tables are regenerated
 * here from the formulas CMSIS documents for them:
 *     twiddleCoef_N_q15[2i], [2i+1] = cos(2 pi i / N), sin(2 pi i / N),  i < 3N/4
 *     realCoefAQ15[2i], [2i+1]      = 0.5 (1 - sin(2 pi i / 8192)),  -0.5 cos(2 pi i / 8192)
 *     realCoefBQ15[2i], [2i+1]      = 0.5 (1 + sin(2 pi i / 8192)),   0.5 cos(2 pi i / 8192)
 * The float -> Q15 conversion of the published tables is selectable (tw_mode / rc_mode) because the snapshot
 * cannot settle it: 0 = floor(x * 2^15) (what taking the top halfword of the Q31 tables gives), 1 = round half
 * away. The bit-reversal tables are not needed: five radix-4 stages with the middle outputs swapped leave
 * X[k] at the bit-reversed position of k (see bitrev below), which is all arm_bitreversal_16 undoes.
 *
 * PARITY STATUS: the only evidence the reference holds for this path is the host-vs-board comparison printed in
 * README.md:121-139 for data/edison_16k_16b.wav (rmse / scale / correlation / deviation extremes of the 31x13
 * network input). tests/golden/gen_fixtures_q15.py evaluates those statistics on this restatement; see DESIGN.md
 * for the outcome. Beyond that printed summary: parity unpinned.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef struct { int16_t re, im; } cq15;

static inline int16_t sat16(int32_t v) { return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }
static inline cq15 c_qadd(cq15 a, cq15 b) { cq15 r = { sat16((int32_t)a.re + b.re), sat16((int32_t)a.im + b.im) }; return r; }
static inline cq15 c_qsub(cq15 a, cq15 b) { cq15 r = { sat16((int32_t)a.re - b.re), sat16((int32_t)a.im - b.im) }; return r; }
/* halving add / subtract: the 17-bit sum shifted right once, no saturation */
static inline cq15 c_hadd(cq15 a, cq15 b) { cq15 r = { (int16_t)(((int32_t)a.re + b.re) >> 1), (int16_t)(((int32_t)a.im + b.im) >> 1) }; return r; }
static inline cq15 c_hsub(cq15 a, cq15 b) { cq15 r = { (int16_t)(((int32_t)a.re - b.re) >> 1), (int16_t)(((int32_t)a.im - b.im) >> 1) }; return r; }
static inline cq15 c_half(cq15 a) { cq15 r = { (int16_t)(a.re >> 1), (int16_t)(a.im >> 1) }; return r; }
/* s + i*t and s - i*t, saturating (QASX / QSAX) and halving (SHASX / SHSAX) */
static inline cq15 c_q_plus_i(cq15 s, cq15 t) { cq15 r = { sat16((int32_t)s.re - t.im), sat16((int32_t)s.im + t.re) }; return r; }
static inline cq15 c_q_minus_i(cq15 s, cq15 t) { cq15 r = { sat16((int32_t)s.re + t.im), sat16((int32_t)s.im - t.re) }; return r; }
static inline cq15 c_h_plus_i(cq15 s, cq15 t) { cq15 r = { (int16_t)(((int32_t)s.re - t.im) >> 1), (int16_t)(((int32_t)s.im + t.re) >> 1) }; return r; }
static inline cq15 c_h_minus_i(cq15 s, cq15 t) { cq15 r = { (int16_t)(((int32_t)s.re + t.im) >> 1), (int16_t)(((int32_t)s.im - t.re) >> 1) }; return r; }
/* x * conj(w), Q15 x Q15 -> bits 31..16 of the 32-bit (wrapping) dual multiply-accumulate (SMUAD / SMUSDX) */
static inline cq15 c_twiddle(cq15 x, const int16_t *w)
{
	uint32_t re = (uint32_t)((int32_t)w[0] * x.re) + (uint32_t)((int32_t)w[1] * x.im);
	uint32_t im = (uint32_t)((int32_t)w[0] * x.im) - (uint32_t)((int32_t)w[1] * x.re);
	cq15 r = { (int16_t)(re >> 16), (int16_t)(im >> 16) };
	return r;
}

/* arm_radix4_butterfly_q15 (DSP branch), forward, n = 16 or 1024, tw = twiddleCoef_n_q15. Output is left in
 * bit-reversed order exactly as the routine leaves it. */
static void radix4_q15(cq15 *x, int n, const int16_t *tw)
{
	int n2 = n >> 2, mod = 1;
	/* first stage: inputs pre-scaled by 1/4 (arm_cfft_radix4_q15.c:181-321) */
	for (int j = 0; j < n2; j++)
	{
		cq15 a = c_half(c_half(x[j])), b = c_half(c_half(x[j + n2]));
		cq15 c = c_half(c_half(x[j + 2 * n2])), d = c_half(c_half(x[j + 3 * n2]));
		cq15 r = c_qadd(a, c), s = c_qsub(a, c), t = c_qadd(b, d);
		x[j] = c_hadd(r, t);
		x[j + n2] = c_twiddle(c_qsub(r, t), tw + 4 * j);
		t = c_qsub(b, d);
		x[j + 2 * n2] = c_twiddle(c_q_minus_i(s, t), tw + 2 * j);
		x[j + 3 * n2] = c_twiddle(c_q_plus_i(s, t), tw + 6 * j);
	}
	mod <<= 2;
	/* middle stages (:335-455) */
	for (int k = n / 4; k > 4; k >>= 2)
	{
		int n1 = n2;
		n2 >>= 2;
		for (int j = 0; j < n2; j++)
		{
			const int ic = j * mod;
			for (int i0 = j; i0 < n; i0 += n1)
			{
				cq15 a = x[i0], b = x[i0 + n2], c = x[i0 + 2 * n2], d = x[i0 + 3 * n2];
				cq15 r = c_qadd(a, c), s = c_qsub(a, c), t = c_qadd(b, d);
				x[i0] = c_half(c_hadd(r, t));
				x[i0 + n2] = c_twiddle(c_hsub(r, t), tw + 4 * ic);
				t = c_qsub(b, d);
				x[i0 + 2 * n2] = c_twiddle(c_h_minus_i(s, t), tw + 2 * ic);
				x[i0 + 3 * n2] = c_twiddle(c_h_plus_i(s, t), tw + 6 * ic);
			}
		}
		mod <<= 2;
	}
	/* last stage, no twiddles (:470-561) */
	for (int g = 0; g < n; g += 4)
	{
		cq15 a = x[g], b = x[g + 1], c = x[g + 2], d = x[g + 3];
		cq15 r = c_qadd(a, c), t = c_qadd(b, d), s = c_qsub(a, c), u = c_qsub(b, d);
		x[g] = c_hadd(r, t);
		x[g + 1] = c_hsub(r, t);
		x[g + 2] = c_h_minus_i(s, u);
		x[g + 3] = c_h_plus_i(s, u);
	}
}

static inline int bitrev(int v, int bits)
{
	int r = 0;
	for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
	return r;
}

/* arm_sqrt_q31: float seed from the bit trick, three Newton steps on 1/sqrt in Q29-ish integers */
static int32_t sqrt_q31(int32_t in)
{
	if (in <= 0) return 0;
	int sign_bits = __builtin_clz((uint32_t)in) - 1;
	int sh = (sign_bits & 1) ? sign_bits - 1 : sign_bits;
	int32_t number = (int32_t)((uint32_t)in << sh);
	int32_t half = number >> 1, keep = number;
	union { int32_t i; float f; } cv;
	volatile float seed = (float)number * 4.6566128731e-010f;
	cv.f = seed;
	cv.i = 0x5f3759df - (cv.i >> 1);
	volatile float scaled = cv.f * 1073741824.0f;
	int32_t v = (int32_t)scaled;
	for (int it = 0; it < 3; it++)
	{
		int32_t vv = (int32_t)(((int64_t)v * v) >> 31);
		int32_t hv = (int32_t)(((int64_t)vv * (int64_t)half) >> 31);
		v = (int32_t)((uint32_t)(int32_t)(((int64_t)v * (int64_t)(0x30000000 - hv)) >> 31) << 2);
	}
	v = (int32_t)((uint32_t)(int32_t)(((int64_t)keep * v) >> 31) << 1);
	return v >> (sh / 2);
}

/* ---------------------------------------------------------------------------------------------- tables */

static int16_t to_q15(double x, int mode)
{
	double v = x * 32768.0;
	double q = mode == 0 ? floor(v) : (v >= 0 ? floor(v + 0.5) : -floor(-v + 0.5));
	return sat16((int32_t)q);
}

struct oracle_q15_tables {
	int16_t tw1024[1536];
	int16_t tw16[24];
	int16_t rfa[32], rfb[32];   /* realCoefA/B pairs at index 256*i, i = 0..15 (twidCoefRModifier for N = 32) */
	int16_t mel_coef[2048];
	int16_t mel_start[64], mel_count[64];
	int n_mel, n_coef_total;
};

oracle_q15_tables_t *oracle_q15_tables_new(int tw_mode, int rc_mode, int num_mel_bins, double sample_rate,
                                           double lower_edge_hertz, double upper_edge_hertz, int mel_mtx_scale)
{
	if (num_mel_bins != 32) return NULL; /* the DCT stage is a 32-point real FFT */
	oracle_q15_tables_t *t = (oracle_q15_tables_t *)calloc(1, sizeof(*t));
	if (!t) return NULL;
	for (int i = 0; i < 768; i++)
	{
		t->tw1024[2 * i] = to_q15(cos(2.0 * M_PI * i / 1024.0), tw_mode);
		t->tw1024[2 * i + 1] = to_q15(sin(2.0 * M_PI * i / 1024.0), tw_mode);
	}
	for (int i = 0; i < 12; i++)
	{
		t->tw16[2 * i] = to_q15(cos(2.0 * M_PI * i / 16.0), tw_mode);
		t->tw16[2 * i + 1] = to_q15(sin(2.0 * M_PI * i / 16.0), tw_mode);
	}
	for (int i = 0; i < 16; i++)
	{
		double a = 2.0 * M_PI * (256.0 * i) / 8192.0;
		t->rfa[2 * i] = to_q15(0.5 * (1.0 - sin(a)), rc_mode);
		t->rfa[2 * i + 1] = to_q15(-0.5 * cos(a), rc_mode);
		t->rfb[2 * i] = to_q15(0.5 * (1.0 + sin(a)), rc_mode);
		t->rfb[2 * i + 1] = to_q15(0.5 * cos(a), rc_mode);
	}
	/* mel_constants.h: int16(scale * W) truncated, then per band the first non-zero bin and the non-zero count */
	const int nbins = 513;
	double *W = (double *)malloc(sizeof(double) * nbins * num_mel_bins);
	if (!W) { free(t); return NULL; }
	oracle_mel_weight_matrix(num_mel_bins, nbins, sample_rate, lower_edge_hertz, upper_edge_hertz, W);
	int pos = 0;
	for (int m = 0; m < num_mel_bins; m++)
	{
		int first = -1, cnt = 0;
		for (int k = 0; k < nbins; k++)
			if ((int16_t)(mel_mtx_scale * W[k * num_mel_bins + m]) != 0) { if (first < 0) first = k; cnt++; }
		if (first < 0) first = 0;
		t->mel_start[m] = (int16_t)first;
		t->mel_count[m] = (int16_t)cnt;
		for (int k = first; k < first + cnt && pos < 2048; k++)
			t->mel_coef[pos++] = (int16_t)(mel_mtx_scale * W[k * num_mel_bins + m]);
	}
	free(W);
	t->n_mel = num_mel_bins;
	t->n_coef_total = pos;
	return t;
}

void oracle_q15_tables_free(oracle_q15_tables_t *t) { free(t); }

int oracle_q15_tables_get(const oracle_q15_tables_t *t, int16_t *tw1024, int16_t *tw16, int16_t *rfa, int16_t *rfb,
                          int16_t *mel_coef, int16_t *mel_start, int16_t *mel_count)
{
	if (tw1024) memcpy(tw1024, t->tw1024, sizeof(t->tw1024));
	if (tw16) memcpy(tw16, t->tw16, sizeof(t->tw16));
	if (rfa) memcpy(rfa, t->rfa, sizeof(t->rfa));
	if (rfb) memcpy(rfb, t->rfb, sizeof(t->rfb));
	if (mel_coef) memcpy(mel_coef, t->mel_coef, sizeof(int16_t) * (size_t)t->n_coef_total);
	if (mel_start) memcpy(mel_start, t->mel_start, sizeof(int16_t) * (size_t)t->n_mel);
	if (mel_count) memcpy(mel_count, t->mel_count, sizeof(int16_t) * (size_t)t->n_mel);
	return t->n_coef_total;
}

/* ---------------------------------------------------------------------------------------------- one frame */

static void frame_q15(const oracle_q15_tables_t *t, const int16_t *in, int mel_mtx_scale,
                      int16_t *fft_out, int16_t *spec_out, int16_t *mel_out, int16_t *mfcc_out)
{
	cq15 x[1024];
	int16_t spec[513], mel[32];
	/* [1] real samples into the real parts, arm_cfft_q15(len 1024, forward, bit reversal on) (:133-139) */
	for (int i = 0; i < 1024; i++) { x[i].re = in[i]; x[i].im = 0; }
	radix4_q15(x, 1024, t->tw1024);
	/* [2] magnitude: sqrt of re^2 + im^2 as Q31, top halfword (:299-312). Only bins below 513 are used later. */
	for (int k = 0; k <= 512; k++)
	{
		cq15 v = x[bitrev(k, 10)];
		if (fft_out) { fft_out[2 * k] = v.re; fft_out[2 * k + 1] = v.im; }
		int32_t sum = (int32_t)((uint32_t)((int32_t)v.re * v.re) + (uint32_t)((int32_t)v.im * v.im));
		spec[k] = (int16_t)(sqrt_q31(sum) >> 16);
	}
	if (fft_out)
		for (int k = 513; k < 1024; k++) { cq15 v = x[bitrev(k, 10)]; fft_out[2 * k] = v.re; fft_out[2 * k + 1] = v.im; }
	/* [3] compact mel matrix, 32-bit accumulator, C division by the scale, cast to q15 (:158-172) */
	const int16_t *coef = t->mel_coef;
	for (int m = 0; m < 32; m++)
	{
		int32_t acc = 0;
		for (int f = t->mel_start[m]; f < t->mel_start[m] + t->mel_count[m]; f++)
			acc = (int32_t)((uint32_t)acc + (uint32_t)((int32_t)spec[f] * (int32_t)*coef++));
		mel[m] = (int16_t)(acc / mel_mtx_scale);
	}
	/* [5] dct2_q15: v[i] = x[2i], v[31-i] = x[2i+1]; 32-point real FFT = 16-point complex FFT + split; real parts */
	int16_t v[32];
	for (int i = 0; i < 16; i++) { v[i] = mel[2 * i]; v[31 - i] = mel[2 * i + 1]; }
	cq15 z[16], zn[16];
	for (int i = 0; i < 16; i++) { z[i].re = v[2 * i]; z[i].im = v[2 * i + 1]; }
	radix4_q15(z, 16, t->tw16);
	for (int i = 0; i < 16; i++) zn[i] = z[bitrev(i, 4)];
	int16_t out[32];
	for (int i = 1; i < 16; i++)
	{
		const int16_t *A = t->rfa + 2 * i, *B = t->rfb + 2 * i;
		cq15 p = zn[i], q = zn[16 - i];
		uint32_t r = (uint32_t)((int32_t)p.re * A[0]) - (uint32_t)((int32_t)p.im * A[1]);
		r += (uint32_t)((int32_t)q.re * B[0]) + (uint32_t)((int32_t)q.im * B[1]);
		out[i] = (int16_t)(r >> 16);
		out[32 - i] = out[i];
	}
	out[16] = (int16_t)(((int32_t)zn[0].re - zn[0].im) >> 1);
	out[0] = (int16_t)(((int32_t)zn[0].re + zn[0].im) >> 1);
	if (spec_out) memcpy(spec_out, spec, sizeof(spec));
	if (mel_out) memcpy(mel_out, mel, sizeof(mel));
	memcpy(mfcc_out, out, sizeof(out));
}

int oracle_mfcc_q15(const oracle_q15_tables_t *t, const int16_t *x, int64_t n_frames, int64_t frame_step,
                    int mel_mtx_scale, int16_t *fft, int16_t *spec, int16_t *mel, int16_t *mfcc, int n_threads)
{
	if (!t || !x || !mfcc || mel_mtx_scale == 0) return -1;
	if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(static)
	for (int64_t n = 0; n < n_frames; n++)
		frame_q15(t, x + n * frame_step, mel_mtx_scale, fft ? fft + n * 2048 : NULL, spec ? spec + n * 513 : NULL,
		          mel ? mel + n * 32 : NULL, mfcc + n * 32);
	return 0;
}

/* mfccToNetInput, NNoM branch (firmware/src/app.c:686-694): C division by NNOM_INPUT_SCALE, clip, cast */
void oracle_net_input_q15(const int16_t *mfcc, int64_t n_rows, int stride, int n_coef, int scale, int clip_lo,
                          int clip_hi, int8_t *out)
{
	for (int64_t r = 0; r < n_rows; r++)
		for (int c = 0; c < n_coef; c++)
		{
			int16_t v = (int16_t)(mfcc[r * stride + c] / scale);
			v = v > clip_hi ? (int16_t)clip_hi : v;
			v = v < clip_lo ? (int16_t)clip_lo : v;
			out[r * n_coef + c] = (int8_t)v;
		}
}
