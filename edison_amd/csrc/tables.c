/*
 * tables.c -- host-side (plain C) construction of the constant tables the MFCC kernel reads.
 *
 * Mirrors, for the two host MFCC variants of the reference:
 *   gen_mel_weight_matrix   audio/edison/mfcc/mfcc_utils.py:36-73  (TF-style triangular weights, DC row zero)
 *   hertz_to_mel            audio/edison/mfcc/mfcc_utils.py:30-34, constants audio/config.py:35-36
 *   variant A scaling       mfcc_utils.py:171-193  (|fft|[:512], 512-bin matrix, ln(x+1e-6), dct2/sqrt(2*32))
 *   variant B scaling       mfcc_utils.py:282-318  (fft/1024, |.|/sqrt2, 513-bin matrix, dct2 * 1/64)
 *   variant TF              mfcc_utils.py:201-253  (tf.signal.stft: periodic Hann window, rfft; |.|, 513-bin matrix,
 *                           ln(x+1e-6), tf.signal.mfccs_from_log_mel_spectrograms = dct2 * rsqrt(2*32))
 *
 * The device works on Z = FFT512(x[2n] + i*x[2n+1]) and on 2*X[k] (see mfcc_kernels.hip), so the factor 1/2
 * of the real-FFT split is folded into spec_scale together with the variant's own normalisation.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

#define MEL_HIGH_FREQUENCY_Q 1127.0
#define MEL_BREAK_FREQUENCY_HERTZ 700.0

static double hz2mel(double hz) { return MEL_HIGH_FREQUENCY_Q * log(1.0 + hz / MEL_BREAK_FREQUENCY_HERTZ); }

/* np.linspace(a, b, n): a + i*step with step = (b-a)/(n-1); the endpoint is returned exactly */
static void linspace(double a, double b, int n, double *out)
{
	double step = n > 1 ? (b - a) / (double)(n - 1) : 0.0;
	for (int i = 0; i < n; i++) out[i] = (double)i * step + a;
	if (n > 1) out[n - 1] = b;
}

int ed_gen_mel_weight_matrix(int num_mel_bins, int num_spectrogram_bins, double sample_rate,
                             double lower_edge_hertz, double upper_edge_hertz, double *W)
{
	if (num_mel_bins < 1 || num_spectrogram_bins < 2 || W == NULL) return EDISON_E_ARGUMENT;
	double *lin = (double *)malloc(sizeof(double) * (size_t)num_spectrogram_bins);
	double *edge = (double *)malloc(sizeof(double) * (size_t)(num_mel_bins + 2));
	if (!lin || !edge) { free(lin); free(edge); return EDISON_E_NO_MEMORY; }
	linspace(0.0, sample_rate / 2.0, num_spectrogram_bins, lin);
	linspace(hz2mel(lower_edge_hertz), hz2mel(upper_edge_hertz), num_mel_bins + 2, edge);
	memset(W, 0, sizeof(double) * (size_t)num_mel_bins); /* the DC bin is excluded, then padded back as zeros */
	for (int k = 1; k < num_spectrogram_bins; k++)
	{
		double m = hz2mel(lin[k]);
		double *row = W + (size_t)k * num_mel_bins;
		for (int j = 0; j < num_mel_bins; j++)
		{
			double up = (m - edge[j]) / (edge[j + 1] - edge[j]);
			double dn = (edge[j + 2] - m) / (edge[j + 2] - edge[j + 1]);
			double w = up < dn ? up : dn;
			row[j] = w > 0.0 ? w : 0.0;
		}
	}
	free(lin); free(edge);
	return EDISON_OK;
}

/* ---- bank-conflict-aware lane assignment of the mel stage (fast kernel, ed_mfcc2_kernel) -------------------------
 * The kernel reads the interleaved spectra of its two frames with ds_read_b128: quad q of the spectrum is the two
 * 16-byte slots 2q and 2q+1, and a lane reads both (one per instruction, in the order its `half` bit says). The LDS
 * serves such a read in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
 * (MI355X_MICROARCH.md, LDS table) -- and a group costs as many passes as its most loaded slot (address / 16 mod 16)
 * holds distinct addresses. Lane (column c = lane & 15, row r = lane >> 4) works on quarter r of band col_band[c]
 * and of band 31 - col_band[c]; any column order and any half order give the same sums, so both are chosen here,
 * by a deterministic hill climb from a few starting points, to minimise the passes of one frame pair. */
static const unsigned char g_b128_group[4][16] = {
	{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
	{4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
	{32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
	{36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

static int mel_read_passes(int first4[2][EDISON_NUM_MEL][4], int NLO, int NHI, const int *col_band, const int *half)
{
	int total = 0;
	for (int part = 0; part < 2; part++)
		for (int t = 0; t < (part == 0 ? NLO : NHI); t++)
			for (int second = 0; second < 2; second++)
				for (int g = 0; g < 4; g++)
				{
					int n_in_slot[16] = {0}, held[16][16], worst = 1;
					for (int i = 0; i < 16; i++)
					{
						const int l = g_b128_group[g][i], b = col_band[l & 15];
						const int q = first4[part][part == 0 ? b : 31 - b][l >> 4] + t;
						const int addr = 2 * q + (second ? 1 - half[l] : half[l]), slot = addr & 15;
						int k = 0;
						while (k < n_in_slot[slot] && held[slot][k] != addr) k++; /* equal addresses share a pass */
						if (k == n_in_slot[slot]) held[slot][n_in_slot[slot]++] = addr;
						if (n_in_slot[slot] > worst) worst = n_in_slot[slot];
					}
					total += worst;
				}
	return total;
}

static void choose_lane_assignment(int first4[2][EDISON_NUM_MEL][4], int NLO, int NHI, int *col_band, int *half)
{
	int best = -1, cb[16], hf[64];
	uint32_t rng = 12345u;
	for (int start = 0; start < 8; start++)
	{
		for (int c = 0; c < 16; c++) cb[c] = c;
		for (int l = 0; l < 64; l++) hf[l] = l & 1;
		if (start > 0) /* shuffled start (LCG, so the tables are the same on every run) */
		{
			for (int c = 15; c > 0; c--)
			{
				rng = rng * 1664525u + 1013904223u;
				const int k = (int)((rng >> 8) % (uint32_t)(c + 1)), tmp = cb[c];
				cb[c] = cb[k]; cb[k] = tmp;
			}
			for (int l = 0; l < 64; l++) { rng = rng * 1664525u + 1013904223u; hf[l] = (int)((rng >> 16) & 1u); }
		}
		int cost = mel_read_passes(first4, NLO, NHI, cb, hf), improved = 1;
		while (improved)
		{
			improved = 0;
			for (int l = 0; l < 64; l++)
			{
				hf[l] ^= 1;
				const int c2 = mel_read_passes(first4, NLO, NHI, cb, hf);
				if (c2 < cost) { cost = c2; improved = 1; } else hf[l] ^= 1;
			}
			for (int a = 0; a < 16; a++)
				for (int b = a + 1; b < 16; b++)
				{
					int tmp = cb[a]; cb[a] = cb[b]; cb[b] = tmp;
					const int c2 = mel_read_passes(first4, NLO, NHI, cb, hf);
					if (c2 < cost) { cost = c2; improved = 1; } else { tmp = cb[a]; cb[a] = cb[b]; cb[b] = tmp; }
				}
		}
		if (best < 0 || cost < best)
		{
			best = cost;
			memcpy(col_band, cb, sizeof(cb));
			memcpy(half, hf, sizeof(hf));
		}
	}
}

int ed_build_mfcc_tables(int variant, double sample_rate, double lower_edge_hertz, double upper_edge_hertz,
                         double mel_mtx_scale, ed_mfcc_tables_t *out, char *err, size_t err_cap)
{
	const int NMEL = EDISON_NUM_MEL;
	const int nbins = (variant == EDISON_MFCC_A) ? EDISON_FRAME_LEN / 2 : EDISON_FRAME_LEN / 2 + 1;
	if (variant != EDISON_MFCC_A && variant != EDISON_MFCC_B && variant != EDISON_MFCC_TF) return EDISON_E_ARGUMENT;
	memset(out, 0, sizeof(*out));
	/* scale of the reference's spectrogram relative to the kernel's 2|X[k]|: A |X| (mfcc_utils.py:174),
	 * B |X/1024|/sqrt2 (mfcc_utils.py:297-300),
	 * TF |X| (tf.abs(stfts), mfcc_utils.py:222) */
	const double spec_scale = (variant == EDISON_MFCC_B) ? 0.5 / 1024.0 / sqrt(2.0) : 0.5;

	/* --- FFT twiddles, rounded once from float64 */
	for (int l = 0; l < 64; l++)
		for (int p = 0; p < 8; p++)
		{
			double a = -2.0 * M_PI * (double)(l * p) / 512.0;
			out->tw1[p][l][0] = (float)cos(a); out->tw1[p][l][1] = (float)sin(a);
			a = -2.0 * M_PI * (double)((l & 7) * p) / 64.0;
			out->tw2[p][l][0] = (float)cos(a); out->tw2[p][l][1] = (float)sin(a);
		}
	for (int m = 0; m < 4; m++)
		for (int l = 0; l < 64; l++)
		{
			const int k0 = ED_K0(l); /* lane l holds Z[k0 + 64r] after the last FFT pass */
			double a = -2.0 * M_PI * (double)(k0 + 64 * m) / 1024.0;
			out->twp[m][l][0] = (float)cos(a); out->twp[m][l][1] = (float)sin(a);
		}

	/* --- mel filterbank -> per-lane tap lists */
	double *W = (double *)malloc(sizeof(double) * (size_t)nbins * NMEL);
	if (!W) return EDISON_E_NO_MEMORY;
	int r = ed_gen_mel_weight_matrix(NMEL, nbins, sample_rate, lower_edge_hertz, upper_edge_hertz, W);
	if (r != EDISON_OK) { free(W); return r; }
	/* variant B multiplies the matrix by mel_mtx_scale and divides the product by it again
	 * (mfcc_utils.py:282,309); in float arithmetic that is the identity up to rounding, so the device
	 * uses the unscaled matrix. mel_mtx_scale is accepted for signature compatibility. */
	(void)mel_mtx_scale;
	/* Balanced tap assignment. The spectrum is read as 16-byte quads; band j spans the quads q0[j]..q1[j].
	 * Band widths grow with frequency (7..71 bins), so lane (b = lane&15, r = lane>>4) takes quarter r of the
	 * narrow band b AND quarter r of the wide band 31-b: n = ceil(quads/4) consecutive quads of each. */
	int q0[EDISON_NUM_MEL], nq[EDISON_NUM_MEL], ks[EDISON_NUM_MEL], ke[EDISON_NUM_MEL];
	for (int j = 0; j < NMEL; j++)
	{
		int first = -1, last = -1;
		for (int k = 0; k < nbins; k++)
			if (W[(size_t)k * NMEL + j] != 0.0) { if (first < 0) first = k; last = k; }
		if (first < 0) { first = 0; last = -1; } /* empty band (degenerate edges) */
		ks[j] = first; ke[j] = last + 1;
		q0[j] = first / 4;
		nq[j] = last >= first ? last / 4 - first / 4 + 1 : 0;
	}
	int NLO = 1, NHI = 1;
	for (int b = 0; b < 16; b++)
	{
		if ((nq[b] + 3) / 4 > NLO) NLO = (nq[b] + 3) / 4;
		if ((nq[31 - b] + 3) / 4 > NHI) NHI = (nq[31 - b] + 3) / 4;
	}
	if (NLO > ED_MEL_NLO_MAX || NHI > ED_MEL_NHI_MAX)
	{
		if (err) snprintf(err, err_cap, "mel filterbank needs %d+%d spectrum quads per lane, the kernel's budget is %d+%d",
		                  NLO, NHI, ED_MEL_NLO_MAX, ED_MEL_NHI_MAX);
		free(W);
		return EDISON_E_NO_IMPL;
	}
	/* the kernel is compiled for two table shapes: 2+5 quads per lane (the shipped filterbank) and 3+6 */
	/* EDISON_FORCE_WIDE_MEL=1 selects the larger shape regardless (lets the tests cover that kernel instance) */
	const char *force_wide = getenv("EDISON_FORCE_WIDE_MEL");
	if (NLO <= 2 && NHI <= 5 && !(force_wide && force_wide[0] == '1')) { NLO = 2; NHI = 5; }
	else { NLO = ED_MEL_NLO_MAX; NHI = ED_MEL_NHI_MAX; }
	out->mel_NLO = NLO; out->mel_NHI = NHI;
	/* first quad of quarter rr of band j, for the narrow (part 0, N = NLO) and the wide (part 1, N = NHI) role */
	int first4[2][EDISON_NUM_MEL][4];
	for (int j = 0; j < NMEL; j++)
		for (int rr = 0; rr < 4; rr++)
			for (int part = 0; part < 2; part++)
			{
				const int N = part == 0 ? NLO : NHI;
				int s4 = q0[j] + rr * ((nq[j] + 3) / 4);
				if (s4 > ED_SPEC_QUADS - N) s4 = ED_SPEC_QUADS - N; /* keep every read inside the padded spectrum */
				if (s4 < 0) s4 = 0;
				first4[part][j][rr] = s4;
			}
	/* which column of the wavefront serves which band pair, and which half of a quad each lane reads first: free
	 * choices (the sums do not depend on them), made to spread the fast kernel's 16-byte LDS reads over the banks */
	int col_band[16], half[64];
	choose_lane_assignment(first4, NLO, NHI, col_band, half);
	for (int l = 0; l < 64; l++)
	{
		const int b = col_band[l & 15], rr = l >> 4;
		out->mel_band[l] = b;
		out->mel_half[l] = half[l];
		for (int part = 0; part < 2; part++)
		{
			const int j = part == 0 ? b : 31 - b, N = part == 0 ? NLO : NHI;
			const int per = (nq[j] + 3) / 4;                   /* quads per quarter of this band            */
			const int pq0 = q0[j] + rr * per;                  /* this lane's quads: [pq0, pq1)             */
			int pq1 = pq0 + per;
			if (pq1 > q0[j] + nq[j]) pq1 = q0[j] + nq[j];
			const int s4 = first4[part][j][rr];
			if (part == 0) out->mel_slo4[l] = s4; else out->mel_shi4[l] = s4;
			for (int t = 0; t < N; t++)
				for (int c = 0; c < 4; c++)
				{
					const int q = s4 + t, k = 4 * q + c;
					const int mine = q >= pq0 && q < pq1 && k >= ks[j] && k < ke[j] && k < nbins;
					/* the kernel's spectrum is 2|X[k]|: the factor 1/2 and the variant's normalisation ride on the weights */
					out->mel_w4[(part == 0 ? 0 : NLO) + t][l][c] = mine ? (float)(spec_scale * W[(size_t)k * NMEL + j]) : 0.0f;
				}
		}
	}
	free(W);

	/* --- DCT-II (scipy.fftpack.dct type 2, norm=None: y[c] = 2*sum x[n] cos(pi*c*(2n+1)/(2N))) + scaling */
	double dscale = (variant == EDISON_MFCC_B) ? 1.0 / 64.0 : 1.0 / sqrt(2.0 * (double)NMEL);
	for (int l = 0; l < 64; l++)
	{
		int c = l & 31, h = l >> 5;
		for (int n = 0; n < 8; n++)
		{
			int nn = n + 8 * h; /* 0..15; the kernel folds in n' = 31 - nn through cos symmetry */
			out->dct4[n / 4][l][n % 4] =
			    (float)(dscale * 2.0 * cos(M_PI * (double)c * (double)(2 * nn + 1) / (double)(2 * NMEL)));
		}
	}
	out->spec_scale = (float)spec_scale;
	out->log_offset = 1e-6f;
	out->always_log = (variant == EDISON_MFCC_B) ? 0 : 1;
	if (variant == EDISON_MFCC_TF)
	{
		/* tf.signal.hann_window(frame_length, periodic=True): 0.5 - 0.5 cos(2 pi n / N), rounded once from float64 */
		out->has_window = 1;
		for (int n = 0; n < EDISON_FRAME_LEN; n++)
			out->window2[n / 2][n % 2] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)EDISON_FRAME_LEN));
	}
	return EDISON_OK;
}
