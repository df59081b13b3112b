/*
 * tables_q15.c -- host-side construction of the constant tables of MFCC variant C, the firmware's fixed-point
 * pipeline (firmware/src/audioprocessing.c:116-215 over CMSIS-DSP).
 *
 *   twiddleCoef_1024_q15 / twiddleCoef_16_q15   cos, sin of 2*pi*i/N for i < 3N/4 (arm_cfft_radix4_q15.c reads pairs
 *                                               ic, 2ic, 3ic of them, :236-288)
 *   realCoefAQ15 / realCoefBQ15                 0.5(1 -/+ sin), -/+0.5 cos of 2*pi*i/8192, read with stride
 *                                               twidCoefRModifier = 256 for the 32-point real FFT of the DCT stage
 *                                               (arm_rfft_init_q15.c:213-214, arm_rfft_q15.c:260-320)
 *   melMtxCompact / melCompFStarts / melCompFCount   int16(mel_mtx_scale * W) without its zeros
 *                                               (audio/edison/mfcc/mfcc_on_mcu.py:26-62,82-113)
 *
 * CMSIS-DSP's arm_common_tables.c is not part of the reference snapshot, so the first two groups are regenerated
 * from their documented formulas. The float -> Q15 step is: twiddles = top halfword of the Q31 value, i.e.
 * floor(x * 2^15); split coefficients = round(x * 2^15). That is the one combination that reproduces the
 * host-vs-board statistics the reference publishes (README.md:121-139) digit for digit; see DESIGN.md and
 * tests/golden/gen_fixtures_q15.py.
 *
 * Every coefficient is stored twice, as the two packed operands of v_dot2_i32_i16:
 *   x * conj(w):  re = w.cos*x.re + w.sin*x.im  -> (lo = cos, hi = sin)
 *                 im = w.cos*x.im - w.sin*x.re  -> (lo = -sin, hi = cos)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

static int q15_floor(double x)
{
	double v = floor(x * 32768.0);
	return v > 32767.0 ? 32767 : (v < -32768.0 ? -32768 : (int)v);
}

static int q15_round(double x)
{
	double v = x * 32768.0;
	v = v >= 0 ? floor(v + 0.5) : -floor(-v + 0.5);
	return v > 32767.0 ? 32767 : (v < -32768.0 ? -32768 : (int)v);
}

static uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

static uint32_t mul_q31(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 31); }

/* arm_sqrt_q31 (CMSIS-DSP FastMathFunctions/arm_sqrt_q31.c:50-139) for in > 0, table construction only: the
 * exponent-trick seed of 1/sqrt, three Newton steps, one multiply back, the normalisation shift undone. */
static uint32_t sqrt_q31_host(uint32_t in)
{
	const int sh = (__builtin_clz(in) - 1) & ~1;
	const uint32_t number = in << sh;
	union { int32_t i; float f; } cv;
	cv.f = (float)(int32_t)number * 4.6566128731e-010f;
	cv.i = 0x5f3759df - (cv.i >> 1);
	uint32_t v = (uint32_t)(int32_t)(cv.f * 1073741824.0f);
	for (int it = 0; it < 3; it++)
	{
		const uint32_t hv = mul_q31(mul_q31(v, v), number >> 1);
		v = mul_q31(v, 0x30000000u - hv) << 2;
	}
	return (mul_q31(number, v) << 1) >> (sh >> 1);
}

int ed_build_q15_tables(double sample_rate, double lower_edge_hertz, double upper_edge_hertz, double mel_mtx_scale,
                        ed_q15_tables_t *out, char *err, size_t err_cap)
{
	memset(out, 0, sizeof(*out));
	for (int n = 0; n < 2; n++)
	{
		const int N = n == 0 ? 1024 : 16, cnt = 3 * N / 4;
		uint32_t *w = n == 0 ? out->tw1024 : out->tw16, *wx = n == 0 ? out->tw1024x : out->tw16x;
		for (int i = 0; i < cnt; i++)
		{
			const int c = q15_floor(cos(2.0 * M_PI * i / N)), s = q15_floor(sin(2.0 * M_PI * i / N));
			w[i] = pack16(c, s);
			/* -sin must fit a halfword: sin = -1 is pair 3N/4, one past the last pair the transform reads */
			wx[i] = pack16(s == -32768 ? 32767 : -s, c);
		}
	}
	for (int i = 0; i < 16; i++)
	{
		const double a = 2.0 * M_PI * (256.0 * i) / 8192.0;
		const int are = q15_round(0.5 * (1.0 - sin(a))), aim = q15_round(-0.5 * cos(a));
		const int bre = q15_round(0.5 * (1.0 + sin(a))), bim = q15_round(0.5 * cos(a));
		out->rfa[i] = pack16(are, -aim); /* outR = p.re*A.re - p.im*A.im + q.re*B.re + q.im*B.im */
		out->rfb[i] = pack16(bre, bim);
	}
	/* magnitudes: the kernel takes floor(sqrt(x/2)) where that provably equals arm_sqrt_q31(x) >> 16; for x = 2 c^2
	 * the routine lands on c or one below, recorded here */
	for (uint32_t c = 1; c < 32768; c++)
	{
		const uint32_t mag = sqrt_q31_host(2u * c * c) >> 16;
		if (mag == c - 1) out->sqbit[c >> 5] |= 1u << (c & 31);
		else if (mag != c)
		{
			if (err) snprintf(err, err_cap, "variant C: arm_sqrt_q31(2*%u^2) >> 16 = %u, expected %u or %u", c, mag, c - 1, c);
			return EDISON_E_RUNTIME;
		}
	}

	const int nbins = EDISON_FRAME_LEN / 2 + 1, NMEL = EDISON_NUM_MEL;
	const int scale = (int)mel_mtx_scale;
	if ((double)scale != mel_mtx_scale || scale < 1 || scale > 32767)
	{
		if (err) snprintf(err, err_cap, "variant C needs an integer mel_mtx_scale in 1..32767");
		return EDISON_E_NO_IMPL;
	}
	double *W = (double *)malloc(sizeof(double) * (size_t)nbins * NMEL);
	if (!W) return EDISON_E_NO_MEMORY;
	int r = ed_gen_mel_weight_matrix(NMEL, nbins, sample_rate, lower_edge_hertz, upper_edge_hertz, W);
	if (r != EDISON_OK) { free(W); return r; }
	/* the firmware's compact form: per band the first non-zero bin and the number of non-zero entries; the generator
	 * takes `count` consecutive entries from the first one (mfcc_on_mcu.py:44-48) */
	int first[EDISON_NUM_MEL], cnt[EDISON_NUM_MEL], total = 0;
	for (int m = 0; m < NMEL; m++)
	{
		first[m] = -1; cnt[m] = 0;
		for (int k = 0; k < nbins; k++)
			if ((int16_t)(scale * W[(size_t)k * NMEL + m]) != 0) { if (first[m] < 0) first[m] = k; cnt[m]++; }
		if (first[m] < 0) first[m] = 0;
		if (first[m] + cnt[m] > nbins)
		{
			if (err) snprintf(err, err_cap, "variant C: band %d of the compact mel matrix runs past the spectrum", m);
			free(W);
			return EDISON_E_NO_IMPL;
		}
		total += cnt[m];
	}
	int nlo = 1, nhi = 1;
	for (int b = 0; b < 16; b++)
	{
		if ((cnt[b] + 3) / 4 > nlo) nlo = (cnt[b] + 3) / 4;
		if ((cnt[31 - b] + 3) / 4 > nhi) nhi = (cnt[31 - b] + 3) / 4;
	}
	if (nlo > ED_Q15_NLO_MAX || nhi > ED_Q15_NHI_MAX)
	{
		if (err) snprintf(err, err_cap, "variant C: mel bands need %d+%d taps per lane, the kernel's budget is %d+%d", nlo, nhi,
		                  ED_Q15_NLO_MAX, ED_Q15_NHI_MAX);
		free(W);
		return EDISON_E_NO_IMPL;
	}
	/* two compiled shapes (see edison_internal.h) */
	if (nlo <= 6 && nhi <= 18) { nlo = 6; nhi = 18; } else { nlo = ED_Q15_NLO_MAX; nhi = ED_Q15_NHI_MAX; }
	out->mel_nlo = nlo; out->mel_nhi = nhi;
	for (int l = 0; l < 64; l++)
	{
		const int b = l & 15, rr = l >> 4;
		for (int part = 0; part < 2; part++)
		{
			const int m = part == 0 ? b : 31 - b, N = part == 0 ? nlo : nhi;
			const int per = (cnt[m] + 3) / 4;
			const int k0 = first[m] + rr * per;                       /* this lane's run: [k0, k1) */
			int k1 = k0 + per;
			if (k1 > first[m] + cnt[m]) k1 = first[m] + cnt[m];
			int s = k0;
			if (s > nbins - N) s = nbins - N;                         /* keep every read inside spec[0..512] */
			if (s < 0) s = 0;
			if (part == 0) out->mel_lo_bin[l] = s; else out->mel_hi_bin[l] = s;
			for (int t = 0; t < N; t++)
			{
				const int k = s + t;
				const int mine = k >= k0 && k < k1 && k < nbins;
				const int c = mine ? (int16_t)(scale * W[(size_t)k * NMEL + m]) : 0;
				out->mel_tap[(part == 0 ? 0 : nlo) + t][l] = c;
				if (c != 0 && k == nbins - 1) out->need_nyquist = 1;
			}
		}
	}
	free(W);
	/* packed form for the kernel (see edison_internal.h) */
	for (int l = 0; l < 64; l++)
		for (int part = 0; part < 2; part++)
		{
			const int n = part == 0 ? nlo : nhi, s = part == 0 ? out->mel_lo_bin[l] : out->mel_hi_bin[l];
			const int row0 = part == 0 ? 0 : nlo, pair0 = part == 0 ? 0 : ED_Q15_PAIRS(nlo);
			if (part == 0) out->mel_lo_pair[l] = s >> 1; else out->mel_hi_pair[l] = s >> 1;
			for (int tp = 0; tp < ED_Q15_PAIRS(n); tp++)
			{
				uint32_t v = 0;
				for (int h = 0; h < 2; h++)
				{
					const int t = 2 * ((s >> 1) + tp) + h - s; /* tap index of bin 2(p0+tp)+h */
					if (t >= 0 && t < n) v |= ((uint32_t)out->mel_tap[row0 + t][l] & 0xffffu) << (16 * h);
				}
				out->mel_tap2[pair0 + tp][l] = v;
			}
		}
	out->mel_scale = scale;
	out->n_mel_coef = total;
	return EDISON_OK;
}
