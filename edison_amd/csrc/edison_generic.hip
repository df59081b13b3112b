/*
 * edison_generic.hip -- entry points of the generality path: MFCC variants A / B / TF for any geometry the reference's Python functions
 * accept (mfcc_utils.py:134-199, 255-323), kernel in mfcc_generic_kernels.hip. Tables (cos / sin of the frame length, the mel
 * matrix, the DCT matrix) are built per call on the host in float64 -- this is not a throughput path.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "edison_ctx.h"

extern "C" int ed_launch_mfcc_generic(const ed_mfcc_gen_args_t *a, int n_cu, hipStream_t stream);

struct gen_tables
{
	double *d;  /* one device block: tw | W | dct | window (float32, variant TF) */
	size_t tw_off, w_off, dct_off, win_off;
	gen_tables() : d(NULL), tw_off(0), w_off(0), dct_off(0), win_off(0) {}
	~gen_tables() { if (d) (void)hipFree(d); }
};

static int build_tables(edison_ctx *ctx, int variant, int N, int nmel, double fs, double lo, double hi, double mel_mtx_scale, gen_tables *t, int *n_bins)
{
	const int v = variant & 0xff;
	const int nb = v == EDISON_MFCC_A ? N / 2 : N / 2 + 1;
	*n_bins = nb;
	const size_t n_tw = 2 * (size_t)N, n_w = (size_t)nb * nmel, n_d = (size_t)nmel * nmel;
	const size_t n_win = v == EDISON_MFCC_TF ? ((size_t)N + 1) / 2 : 0; /* N floats in doubles' worth of space */
	double *h = (double *)malloc(sizeof(double) * (n_tw + n_w + n_d + n_win));
	if (!h) return ed_set_err(ctx, EDISON_E_NO_MEMORY, "host allocation failed");
	for (int j = 0; j < N; j++)
	{
		const double ang = 2.0 * M_PI * (double)j / (double)N;
		h[2 * j] = cos(ang);
		h[2 * j + 1] = sin(ang);
	}
	double *W = h + n_tw;
	int r = ed_gen_mel_weight_matrix(nmel, nb, fs, lo, hi, W);
	if (r != EDISON_OK) { free(h); return ed_set_err(ctx, r, "edison_mfcc_generic: mel matrix (needs >= 2 spectrum bins, >= 1 mel bin)"); }
	if (v == EDISON_MFCC_B) for (size_t i = 0; i < n_w; i++) W[i] = mel_mtx_scale * W[i]; /* mfcc_utils.py:281-284 */
	double *D = W + n_w;
	for (int c = 0; c < nmel; c++)
		for (int n = 0; n < nmel; n++) D[(size_t)c * nmel + n] = 2.0 * cos(M_PI * (double)c * (double)(2 * n + 1) / (double)(2 * nmel));
	if (n_win)
	{
		/* tf.signal.hann_window(frame_length, periodic=True), float32: 0.5 - 0.5 cos(2 pi n / N) rounded once from float64 (as tables.c does for the fast path) */
		float *wf = (float *)(D + n_d);
		for (int n = 0; n < N; n++) wf[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)N));
	}
	hipError_t e = hipMalloc((void **)&t->d, sizeof(double) * (n_tw + n_w + n_d + n_win));
	if (e == hipSuccess) e = hipMemcpy(t->d, h, sizeof(double) * (n_tw + n_w + n_d + n_win), hipMemcpyHostToDevice);
	free(h);
	ED_HIP(ctx, e);
	t->tw_off = 0; t->w_off = n_tw; t->dct_off = n_tw + n_w; t->win_off = n_win ? n_tw + n_w + n_d : 0;
	return EDISON_OK;
}

static int check_geometry(edison_ctx *ctx, int64_t n_frames, int frame_len, int64_t frame_step, int variant, int mel_nbins, double fs, double lo, double hi,
                          double scale, int n_coef)
{
	const int v = variant & 0xff;
	if (v != EDISON_MFCC_A && v != EDISON_MFCC_B && v != EDISON_MFCC_TF) return ed_set_err(ctx, EDISON_E_NO_IMPL, "edison_mfcc_generic: variants A, B and TF");
	if (v != EDISON_MFCC_B && (variant & EDISON_MFCC_USE_LOG)) return ed_set_err(ctx, EDISON_E_ARGUMENT, "variants A and TF always take the logarithm");
	if (n_frames < 0 || frame_step < 0) return EDISON_E_ARGUMENT;
	if (n_frames > INT32_MAX) return ed_set_err(ctx, EDISON_E_SIZE, "edison_mfcc_generic: more than 2^31 frames in one call");
	if (frame_len < 4 || frame_len > ED_GEN_MAX_FRAME) return ed_set_err(ctx, EDISON_E_NO_IMPL, "edison_mfcc_generic: frame_len 4 .. 4096");
	if (mel_nbins < 1 || mel_nbins > ED_GEN_MAX_MEL) return ed_set_err(ctx, EDISON_E_NO_IMPL, "edison_mfcc_generic: mel_nbins 1 .. 256");
	if (!(fs > 0) || !(lo >= 0) || !(hi > lo) || !(scale > 0)) return ed_set_err(ctx, EDISON_E_ARGUMENT, "edison_mfcc_generic: bad filterbank edges / scale");
	if (n_coef < 0 || n_coef > mel_nbins) return ed_set_err(ctx, EDISON_E_ARGUMENT, "edison_mfcc_generic: n_coef 0 .. mel_nbins");
	return EDISON_OK;
}

static void fill_args(ed_mfcc_gen_args_t *a, const gen_tables &t, int variant, int N, int nmel, int nb, double scale)
{
	const int v = variant & 0xff;
	a->frame_len = N; a->n_bins = nb; a->n_mel = nmel;
	const bool b = v == EDISON_MFCC_B; /* A and TF share the constants: no scales, ln, dct2 / sqrt(2 nmel) (mfcc_utils.py:193; tf.signal.mfccs_from_log_mel_spectrograms) */
	a->fft_out = v == EDISON_MFCC_A ? N / 2 : (b ? N : N / 2 + 1);
	a->take_log = !b || (variant & EDISON_MFCC_USE_LOG);
	a->fft_scale = b ? 1.0 / 1024.0 : 1.0;          /* mfcc_utils.py:297: the constant 1024, whatever the frame length */
	a->spec_scale = b ? 1.0 / sqrt(2.0) : 1.0;        /* :300 */
	a->mel_div = b ? scale : 1.0;                     /* :309 */
	a->dct_div = b ? 64.0 : sqrt(2.0 * (double)nmel); /* :318, :193 */
	a->tw = t.d + t.tw_off; a->W = t.d + t.w_off; a->dct = t.d + t.dct_off;
	a->window = t.win_off ? (const float *)(t.d + t.win_off) : NULL;
}

extern "C" int edison_mfcc_generic_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int frame_len, int64_t frame_step, int variant, int mel_nbins,
                                       double sample_rate, double lower_edge_hertz, double upper_edge_hertz, double mel_mtx_scale, double *fft, double *spec,
                                       double *mel, double *logmel, double *mfcc, int n_coef, int8_t *feat, float feat_scale)
{
	if (!ctx || (!audio && n_frames > 0)) return EDISON_E_ARGUMENT;
	{ const int r = check_geometry(ctx, n_frames, frame_len, frame_step, variant, mel_nbins, sample_rate, lower_edge_hertz, upper_edge_hertz, mel_mtx_scale, n_coef); if (r != EDISON_OK) return r; }
	if (n_frames == 0) return EDISON_OK;
	if (feat && n_coef < 1) return ed_set_err(ctx, EDISON_E_ARGUMENT, "edison_mfcc_generic: feat needs n_coef >= 1");
	ED_HIP(ctx, hipSetDevice(ctx->device));
	gen_tables t;
	int nb = 0;
	{ const int r = build_tables(ctx, variant, frame_len, mel_nbins, sample_rate, lower_edge_hertz, upper_edge_hertz, mel_mtx_scale, &t, &nb); if (r != EDISON_OK) return r; }
	ed_mfcc_gen_args_t a;
	memset(&a, 0, sizeof(a));
	fill_args(&a, t, variant, frame_len, mel_nbins, nb, mel_mtx_scale);
	a.audio = audio; a.n_frames = n_frames; a.frame_step = frame_step;
	a.fft = fft; a.spec = spec; a.mel = mel; a.logmel = logmel; a.mfcc = mfcc;
	a.n_coef = n_coef; a.feat = feat; a.feat_scale = feat_scale;
	const int e = ed_launch_mfcc_generic(&a, ctx->n_cu, ctx->stream);
	if (e != 0)
	{
		snprintf(ctx->err, sizeof(ctx->err), "generic MFCC kernel launch failed: %s", hipGetErrorString((hipError_t)e));
		return EDISON_E_RUNTIME;
	}
	/* the tables are freed when this function returns: wait for the kernel (a generality path, not a pipeline stage) */
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

struct gbuf
{
	void *p;
	gbuf() : p(NULL) {}
	~gbuf() { if (p) (void)hipFree(p); }
	hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};

extern "C" int edison_mfcc_generic(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int frame_len, int64_t frame_step, int variant, int mel_nbins,
                                   double sample_rate, double lower_edge_hertz, double upper_edge_hertz, double mel_mtx_scale, double *fft, double *spec,
                                   double *mel, double *logmel, double *mfcc, int n_coef, int8_t *feat, float feat_scale)
{
	if (!ctx || (!audio && n_frames > 0)) return EDISON_E_ARGUMENT;
	{ const int r = check_geometry(ctx, n_frames, frame_len, frame_step, variant, mel_nbins, sample_rate, lower_edge_hertz, upper_edge_hertz, mel_mtx_scale, n_coef); if (r != EDISON_OK) return r; }
	if (n_frames == 0) return EDISON_OK;
	ED_HIP(ctx, hipSetDevice(ctx->device));
	const size_t n = (size_t)n_frames, na = ((size_t)(n_frames - 1) * (size_t)frame_step + (size_t)frame_len) * sizeof(int16_t);
	const size_t fo = (size_t)((variant & 0xff) == EDISON_MFCC_A ? frame_len / 2 : ((variant & 0xff) == EDISON_MFCC_B ? frame_len : frame_len / 2 + 1));
	gbuf a, f, s, m, l, c, q;
	ED_HIP(ctx, a.alloc(na));
	if (fft) ED_HIP(ctx, f.alloc(n * fo * 2 * sizeof(double)));
	if (spec) ED_HIP(ctx, s.alloc(n * fo * sizeof(double)));
	if (mel) ED_HIP(ctx, m.alloc(n * mel_nbins * sizeof(double)));
	if (logmel) ED_HIP(ctx, l.alloc(n * mel_nbins * sizeof(double)));
	if (mfcc) ED_HIP(ctx, c.alloc(n * mel_nbins * sizeof(double)));
	if (feat) ED_HIP(ctx, q.alloc(n * (size_t)(n_coef > 0 ? n_coef : 1)));
	ED_HIP(ctx, hipMemcpyAsync(a.p, audio, na, hipMemcpyHostToDevice, ctx->stream));
	const int r = edison_mfcc_generic_dev(ctx, (const int16_t *)a.p, n_frames, frame_len, frame_step, variant, mel_nbins, sample_rate, lower_edge_hertz,
	                                      upper_edge_hertz, mel_mtx_scale, (double *)f.p, (double *)s.p, (double *)m.p, (double *)l.p, (double *)c.p, n_coef,
	                                      (int8_t *)q.p, feat_scale);
	if (r != EDISON_OK) return r;
	if (fft) ED_HIP(ctx, hipMemcpyAsync(fft, f.p, n * fo * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	if (spec) ED_HIP(ctx, hipMemcpyAsync(spec, s.p, n * fo * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	if (mel) ED_HIP(ctx, hipMemcpyAsync(mel, m.p, n * mel_nbins * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	if (logmel) ED_HIP(ctx, hipMemcpyAsync(logmel, l.p, n * mel_nbins * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	if (mfcc) ED_HIP(ctx, hipMemcpyAsync(mfcc, c.p, n * mel_nbins * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	if (feat) ED_HIP(ctx, hipMemcpyAsync(feat, q.p, n * (size_t)n_coef, hipMemcpyDeviceToHost, ctx->stream));
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}
