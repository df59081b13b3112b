/*
 * edison_q15.hip -- C-ABI entry points of MFCC variant C, the firmware's fixed-point audioCalcMFCCs
 * (firmware/src/audioprocessing.c:116-215), with its native int16 outputs. See include/edison_hip.h.
 * The kernel is mfcc_q15_kernels.hip; the tables are tables_q15.c. No CPU path.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "edison_ctx.h"

int ed_ctx_mfcc_q15_launch(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                           int64_t frame_step, int n_coef, int16_t *mfcc_i16, float *mfcc_f32, int8_t *feat, int stages,
                           int16_t *fft, int16_t *spec, int16_t *mel)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	return ed_ctx_mfcc_q15_launch_on(ctx, ctx->stream, audio, n_frames, fpg, group_stride, frame_step, n_coef, mfcc_i16, mfcc_f32, feat,
	                                 stages, fft, spec, mel);
}

int ed_ctx_mfcc_q15_launch_on(edison_ctx *ctx, hipStream_t stream, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                              int64_t frame_step, int n_coef, int16_t *mfcc_i16, float *mfcc_f32, int8_t *feat, int stages,
                              int16_t *fft, int16_t *spec, int16_t *mel)
{
	if (!ctx || (!audio && n_frames > 0)) return EDISON_E_ARGUMENT;
	if (n_coef < 1 || n_coef > EDISON_NUM_MEL) return ed_set_err(ctx, EDISON_E_ARGUMENT, "n_coef must be 1..32");
	if (n_frames < 0 || n_frames >= ((int64_t)1 << 31) || frame_step < 0 || fpg < 1)
		return ed_set_err(ctx, EDISON_E_ARGUMENT, "bad frame count / step");
	if (!ctx->d_q15)
	{
		snprintf(ctx->err, sizeof(ctx->err), "MFCC variant C is not available for the configured filterbank: %s",
		         ctx->q15_err[0] ? ctx->q15_err : "tables not built");
		return EDISON_E_NO_IMPL;
	}
	if (n_frames == 0) return EDISON_OK;
	ed_mfcc_q15_args_t a;
	memset(&a, 0, sizeof(a));
	a.audio = audio; a.n_frames = n_frames; a.frames_per_group = fpg; a.group_stride = group_stride;
	a.frame_step = frame_step; a.n_coef = n_coef;
	a.mfcc_i16 = mfcc_i16; a.mfcc_f32 = mfcc_f32; a.feat = feat;
	a.fft = fft; a.spec = spec; a.mel = mel;
	int e = ed_launch_mfcc_q15(&a, ctx->d_q15, ctx->q15_nlo, ctx->q15_nhi, stages, ctx->n_cu, stream);
	if (e != 0)
	{
		snprintf(ctx->err, sizeof(ctx->err), "Q15 MFCC kernel launch failed: %s", hipGetErrorString((hipError_t)e));
		return EDISON_E_RUNTIME;
	}
	return EDISON_OK;
}

extern "C" int edison_mfcc_q15_batch_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                         int n_coef, int16_t *mfcc, int8_t *feat)
{
	return ed_ctx_mfcc_q15_launch(ctx, audio, n_frames, n_frames > 0 ? n_frames : 1, 0, frame_step, n_coef, mfcc, NULL,
	                              feat, 0, NULL, NULL, NULL);
}

extern "C" int edison_mfcc_q15_stages_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                          int16_t *fft, int16_t *spec, int16_t *mel, int16_t *mfcc32)
{
	return ed_ctx_mfcc_q15_launch(ctx, audio, n_frames, n_frames > 0 ? n_frames : 1, 0, frame_step, EDISON_NUM_MEL,
	                              mfcc32, NULL, NULL, 1, fft, spec, mel);
}

/* ------------------------------------------------------------------------------------------ host pointers */
namespace {
struct dev_buf
{
	void *p;
	dev_buf() : p(NULL) {}
	~dev_buf() { if (p) (void)hipFree(p); }
	hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};
} // namespace

#define EQ_DOWN(ctx, dst, src, n) \
	do { if (dst) ED_HIP(ctx, hipMemcpyAsync((dst), (src), (n), hipMemcpyDeviceToHost, (ctx)->stream)); } while (0)

static int q15_host(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step, int n_coef, int stages,
                    int16_t *mfcc, int8_t *feat, int16_t *fft, int16_t *spec, int16_t *mel)
{
	if (!ctx || n_frames < 0 || (!audio && n_frames > 0) || frame_step < 0) return EDISON_E_ARGUMENT;
	if (n_frames == 0) return EDISON_OK;
	if (n_coef < 1 || n_coef > EDISON_NUM_MEL) return ed_set_err(ctx, EDISON_E_ARGUMENT, "n_coef must be 1..32");
	ED_HIP(ctx, hipSetDevice(ctx->device));
	dev_buf a, m, q, f, s, l;
	const size_t n = (size_t)n_frames;
	const size_t na = ((size_t)(n_frames - 1) * (size_t)frame_step + EDISON_FRAME_LEN) * sizeof(int16_t);
	ED_HIP(ctx, a.alloc(na));
	if (mfcc) ED_HIP(ctx, m.alloc(n * n_coef * sizeof(int16_t)));
	if (feat) ED_HIP(ctx, q.alloc(n * n_coef));
	if (fft) ED_HIP(ctx, f.alloc(n * 513 * 2 * sizeof(int16_t)));
	if (spec) ED_HIP(ctx, s.alloc(n * 513 * sizeof(int16_t)));
	if (mel) ED_HIP(ctx, l.alloc(n * 32 * sizeof(int16_t)));
	ED_HIP(ctx, hipMemcpyAsync(a.p, audio, na, hipMemcpyHostToDevice, ctx->stream));
	int r = ed_ctx_mfcc_q15_launch(ctx, (const int16_t *)a.p, n_frames, n_frames, 0, frame_step, n_coef, (int16_t *)m.p,
	                               NULL, (int8_t *)q.p, stages, (int16_t *)f.p, (int16_t *)s.p, (int16_t *)l.p);
	if (r != EDISON_OK) return r;
	EQ_DOWN(ctx, mfcc, m.p, n * n_coef * sizeof(int16_t));
	EQ_DOWN(ctx, feat, q.p, n * n_coef);
	EQ_DOWN(ctx, fft, f.p, n * 513 * 2 * sizeof(int16_t));
	EQ_DOWN(ctx, spec, s.p, n * 513 * sizeof(int16_t));
	EQ_DOWN(ctx, mel, l.p, n * 32 * sizeof(int16_t));
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

extern "C" int edison_mfcc_q15_batch(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                     int n_coef, int16_t *mfcc, int8_t *feat)
{
	return q15_host(ctx, audio, n_frames, frame_step, n_coef, 0, mfcc, feat, NULL, NULL, NULL);
}

extern "C" int edison_mfcc_q15_stages(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                      int16_t *fft, int16_t *spec, int16_t *mel, int16_t *mfcc32)
{
	return q15_host(ctx, audio, n_frames, frame_step, EDISON_NUM_MEL, 1, mfcc32, NULL, fft, spec, mel);
}
