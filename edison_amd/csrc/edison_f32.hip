/*
 * edison_f32.hip -- C-ABI entry points of MFCC variant D, the firmware's float32 ML-KWS extractor, under the
 * firmware's own names (firmware/src/audio/mfcc.h:64-67: mfcc_create / mfcc_compute / mfcc_delete) plus the batched
 * forms. The kernel is mfcc_f32_kernels.hip, the tables tables_f32.c. No CPU path.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "edison_ctx.h"

extern "C" int ed_launch_mfcc_f32(const ed_mfcc_f32_args_t *args, const ed_f32_tables_t *dev_tab, int padded, int n_cu, hipStream_t stream);

/* the handle: opaque to callers (the firmware's struct fields are its private scratch) */
struct _mfcc_t
{
	edison_ctx *ctx;
	ed_f32_tables_t *d_tab;
	int n_out, frame_len, padded;
};

extern "C" mfcc_t *edison_mfcc_f32_create(edison_ctx *ctx, int num_mfcc_features, int feature_offset, int frame_len,
                                          int mfcc_dec_bits, float preemph)
{
	if (!ctx) return NULL;
	ed_f32_tables_t *h = (ed_f32_tables_t *)malloc(sizeof(ed_f32_tables_t));
	mfcc_t *m = (mfcc_t *)calloc(1, sizeof(mfcc_t));
	if (!h || !m) { free(h); free(m); ed_set_err(ctx, EDISON_E_NO_MEMORY, "host allocation failed"); return NULL; }
	int r = ed_build_f32_tables(num_mfcc_features, feature_offset, frame_len, mfcc_dec_bits, preemph, h, ctx->err, sizeof(ctx->err));
	hipError_t e = hipSuccess;
	const int padded = r == EDISON_OK ? h->padded : 0; /* the kernel's per-wave LDS buffer is sized by it */
	if (r == EDISON_OK)
	{
		e = hipSetDevice(ctx->device);
		if (e == hipSuccess) e = hipMalloc((void **)&m->d_tab, sizeof(ed_f32_tables_t));
		if (e == hipSuccess) e = hipMemcpy(m->d_tab, h, sizeof(ed_f32_tables_t), hipMemcpyHostToDevice);
		if (e != hipSuccess) snprintf(ctx->err, sizeof(ctx->err), "mfcc_create: %s", hipGetErrorString(e));
	}
	free(h);
	if (r != EDISON_OK || e != hipSuccess)
	{
		if (m->d_tab) (void)hipFree(m->d_tab);
		free(m);
		return NULL;
	}
	m->ctx = ctx;
	m->n_out = num_mfcc_features - feature_offset;
	m->frame_len = frame_len;
	m->padded = padded;
	return m;
}

extern "C" void mfcc_delete(mfcc_t *mfcc)
{
	if (!mfcc) return;
	if (mfcc->d_tab) { (void)hipSetDevice(mfcc->ctx->device); (void)hipStreamSynchronize(mfcc->ctx->stream); (void)hipFree(mfcc->d_tab); }
	free(mfcc);
}

extern "C" int edison_mfcc_f32_n_out(const mfcc_t *mfcc) { return mfcc ? mfcc->n_out : EDISON_E_ARGUMENT; }

extern "C" int edison_mfcc_f32_batch_dev(mfcc_t *mfcc, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                         int8_t *out, float *out_f32, float *logmel)
{
	if (!mfcc || n_frames < 0 || frame_step < 0 || ((!audio || !out) && n_frames > 0)) return EDISON_E_ARGUMENT;
	if (n_frames == 0) return EDISON_OK;
	edison_ctx *ctx = mfcc->ctx;
	ed_mfcc_f32_args_t a;
	a.audio = audio; a.n_frames = n_frames; a.frame_step = frame_step; a.out = out; a.out_f32 = out_f32; a.logmel = logmel;
	int e = ed_launch_mfcc_f32(&a, mfcc->d_tab, mfcc->padded, ctx->n_cu, ctx->stream);
	if (e != 0)
	{
		snprintf(ctx->err, sizeof(ctx->err), "float32 MFCC kernel launch failed: %s", hipGetErrorString((hipError_t)e));
		return EDISON_E_RUNTIME;
	}
	return EDISON_OK;
}

namespace {
struct dev_buf
{
	void *p;
	dev_buf() : p(NULL) {}
	~dev_buf() { if (p) (void)hipFree(p); }
	hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};
} // namespace

extern "C" int edison_mfcc_f32_batch(mfcc_t *mfcc, const int16_t *audio, int64_t n_frames, int64_t frame_step, int8_t *out,
                                     float *out_f32, float *logmel)
{
	if (!mfcc || n_frames < 0 || frame_step < 0 || ((!audio || !out) && n_frames > 0)) return EDISON_E_ARGUMENT;
	if (n_frames == 0) return EDISON_OK;
	edison_ctx *ctx = mfcc->ctx;
	ED_HIP(ctx, hipSetDevice(ctx->device));
	dev_buf a, o, f, l;
	const size_t n = (size_t)n_frames, no = (size_t)mfcc->n_out;
	const size_t na = ((size_t)(n_frames - 1) * (size_t)frame_step + (size_t)mfcc->frame_len) * sizeof(int16_t);
	ED_HIP(ctx, a.alloc(na));
	ED_HIP(ctx, o.alloc(n * no));
	if (out_f32) ED_HIP(ctx, f.alloc(n * no * sizeof(float)));
	if (logmel) ED_HIP(ctx, l.alloc(n * ED_F32_NUM_FBANK * sizeof(float)));
	ED_HIP(ctx, hipMemcpyAsync(a.p, audio, na, hipMemcpyHostToDevice, ctx->stream));
	int r = edison_mfcc_f32_batch_dev(mfcc, (const int16_t *)a.p, n_frames, frame_step, (int8_t *)o.p, (float *)f.p, (float *)l.p);
	if (r != EDISON_OK) return r;
	ED_HIP(ctx, hipMemcpyAsync(out, o.p, n * no, hipMemcpyDeviceToHost, ctx->stream));
	if (out_f32) ED_HIP(ctx, hipMemcpyAsync(out_f32, f.p, n * no * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	if (logmel) ED_HIP(ctx, hipMemcpyAsync(logmel, l.p, n * ED_F32_NUM_FBANK * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

/* ---- the firmware's own names, on the process-global context (created like aiInitialize does) ---- */
extern "C" mfcc_t *mfcc_create(int num_mfcc_features, int feature_offset, int frame_len, int mfcc_dec_bits, float preemph)
{
	edison_ctx *ctx = edison_global_ctx();
	if (!ctx)
	{
		if (aiInitialize() != EDISON_OK) return NULL;
		ctx = edison_global_ctx();
	}
	return edison_mfcc_f32_create(ctx, num_mfcc_features, feature_offset, frame_len, mfcc_dec_bits, preemph);
}

extern "C" void mfcc_compute(mfcc_t *mfcc, const int16_t *audio_data, int8_t *mfcc_out)
{
	if (!mfcc || !audio_data || !mfcc_out) return;
	if (edison_mfcc_f32_batch(mfcc, audio_data, 1, mfcc->frame_len, mfcc_out, NULL, NULL) != EDISON_OK)
	{
		fprintf(stderr, "mfcc_compute: GPU MFCC failed: %s\n", edison_last_error(mfcc->ctx));
		memset(mfcc_out, 0, (size_t)mfcc->n_out);
	}
}
