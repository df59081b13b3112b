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
extern "C" int ed_launch_mfcc_f32_fast(const ed_mfcc_f32_args_t *args, const ed_f32_tables_t *dev_tab, const ed_mfcc_tables_t *dev_fft_tab,
                                       int n_cu, hipStream_t stream);

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
	/* frames padded to 512 points (the firmware's configuration) take the register-FFT kernel; its FFT twiddles are those
	 * of the variant A / B tables (a property of the 512-point transform, not of the filterbank). EDISON_F32_GENERIC=1
	 * keeps the generic radix-2 kernel, for A/B measurements */
	static const int generic = getenv("EDISON_F32_GENERIC") ? atoi(getenv("EDISON_F32_GENERIC")) : 0;
	int e = (mfcc->padded == 512 && ctx->d_tab[0] && !generic)
	            ? ed_launch_mfcc_f32_fast(&a, mfcc->d_tab, ctx->d_tab[0], ctx->n_cu, ctx->stream)
	            : ed_launch_mfcc_f32(&a, mfcc->d_tab, mfcc->padded, ctx->n_cu, ctx->stream);
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

/* ================================================================================================================
 * The front end of the firmware's NNoM keyword-spotting example around mfcc_compute (appNnomKwsRun, app.c:545-623):
 * every audio event brings AUDIO_FRAME_LEN = 512 new samples; they are appended behind the last 256 old ones
 * (audio_buffer_16bit, app.c:507,567-575), TWO frames are extracted at offsets 0 and 256 (50 % overlap, app.c:583) and
 * go into a ring of MFCC_LEN = 63 feature rows (app.c:508,594-596); the network input is the ring unrolled oldest row
 * first (mfcc_features_seq, app.c:600-604). Here the state (256 samples, window_rows feature rows, both starting as
 * zeros like the firmware's static buffers) lives in HBM, a push takes any number of events, runs ONE launch of the
 * variant D kernel over all 2 * n_events frames and returns the window after EVERY event.
 */
struct edison_f32_stream
{
	edison_ctx *ctx;
	mfcc_t *mfcc;
	int rows, n_out, max_events;
	int16_t *d_audio; /* [256 + max_events * 512] */
	int8_t *d_feat;   /* [rows + 2 * max_events][n_out]: the last `rows` rows, then the rows of the running push */
	int64_t events_seen;
};

/* window e (after event e of this push) = feature rows 2e+2 .. 2e+1+rows of d_feat; then the state moves up */
__global__ void ed_f32_windows_kernel(const int8_t *__restrict__ feat, int rows, int n_out, int n_events, int8_t *__restrict__ win)
{
	const int64_t per = (int64_t)rows * n_out, total = per * n_events;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
	{
		const int64_t e = i / per, r = i - e * per;
		win[i] = feat[(2 * e + 2) * n_out + r];
	}
}

/* one workgroup: the last `rows` feature rows and the last 256 samples become the state of the next push (read
 * everything, barrier, write: the ranges may overlap) */
__global__ __launch_bounds__(256) void ed_f32_shift_kernel(int8_t *feat, int rows, int n_out, int n_events, int16_t *audio)
{
	extern __shared__ int8_t sh[];
	const int nf = rows * n_out;
	int16_t *sa = reinterpret_cast<int16_t *>(sh + ((nf + 15) & ~15));
	for (int i = threadIdx.x; i < nf; i += 256) sh[i] = feat[(int64_t)2 * n_events * n_out + i];
	for (int i = threadIdx.x; i < 256; i += 256) sa[i] = audio[(int64_t)n_events * 512 + i];
	__syncthreads();
	for (int i = threadIdx.x; i < nf; i += 256) feat[i] = sh[i];
	for (int i = threadIdx.x; i < 256; i += 256) audio[i] = sa[i];
}

extern "C" int edison_f32_stream_create(edison_ctx *ctx, mfcc_t *mfcc, int window_rows, int max_events, edison_f32_stream **out)
{
	if (!ctx || !mfcc || !out) return EDISON_E_ARGUMENT;
	*out = NULL;
	if (mfcc->ctx != ctx) return ed_set_err(ctx, EDISON_E_ARGUMENT, "f32 stream: the extractor belongs to another context");
	if (mfcc->frame_len != 512) return ed_set_err(ctx, EDISON_E_NO_IMPL, "f32 stream: the firmware's front end is defined for 512-sample frames (AUDIO_FRAME_LEN, app.c:497)");
	if (window_rows < 2 || window_rows > 1024 || max_events < 1 || max_events > (1 << 20)) return ed_set_err(ctx, EDISON_E_ARGUMENT, "f32 stream: bad window_rows / max_events");
	edison_f32_stream *s = (edison_f32_stream *)calloc(1, sizeof(edison_f32_stream));
	if (!s) return ed_set_err(ctx, EDISON_E_NO_MEMORY, "host allocation failed");
	s->ctx = ctx; s->mfcc = mfcc; s->rows = window_rows; s->n_out = mfcc->n_out; s->max_events = max_events;
	const size_t na = (256 + (size_t)max_events * 512) * sizeof(int16_t), nf = ((size_t)window_rows + 2 * (size_t)max_events) * mfcc->n_out;
	hipError_t e = hipSetDevice(ctx->device);
	if (e == hipSuccess) e = hipMalloc((void **)&s->d_audio, na);
	if (e == hipSuccess) e = hipMalloc((void **)&s->d_feat, nf);
	if (e == hipSuccess) e = hipMemsetAsync(s->d_audio, 0, na, ctx->stream);
	if (e == hipSuccess) e = hipMemsetAsync(s->d_feat, 0, nf, ctx->stream);
	if (e != hipSuccess)
	{
		snprintf(ctx->err, sizeof(ctx->err), "f32 stream: %s", hipGetErrorString(e));
		if (s->d_audio) (void)hipFree(s->d_audio);
		if (s->d_feat) (void)hipFree(s->d_feat);
		free(s);
		return EDISON_E_RUNTIME;
	}
	*out = s;
	return EDISON_OK;
}

extern "C" void edison_f32_stream_destroy(edison_f32_stream *s)
{
	if (!s) return;
	(void)hipSetDevice(s->ctx->device);
	(void)hipStreamSynchronize(s->ctx->stream);
	(void)hipFree(s->d_audio);
	(void)hipFree(s->d_feat);
	free(s);
}

extern "C" int edison_f32_stream_reset(edison_f32_stream *s)
{
	if (!s) return EDISON_E_ARGUMENT;
	edison_ctx *ctx = s->ctx;
	ED_HIP(ctx, hipMemsetAsync(s->d_audio, 0, 256 * sizeof(int16_t), ctx->stream));
	ED_HIP(ctx, hipMemsetAsync(s->d_feat, 0, (size_t)s->rows * s->n_out, ctx->stream));
	s->events_seen = 0;
	return EDISON_OK;
}

extern "C" int64_t edison_f32_stream_events_seen(const edison_f32_stream *s) { return s ? s->events_seen : 0; }

/* samples: n_events x 512 new int16 samples (device); windows: [n_events][window_rows][n_out] int8 (device), window e =
 * what mfcc_features_seq holds after event e. Asynchronous on the context's stream. */
extern "C" int edison_f32_stream_push_dev(edison_f32_stream *s, const int16_t *samples, int n_events, int8_t *windows)
{
	if (!s || n_events < 0 || (n_events > 0 && (!samples || !windows))) return EDISON_E_ARGUMENT;
	if (n_events == 0) return EDISON_OK;
	edison_ctx *ctx = s->ctx;
	if (n_events > s->max_events) return ed_set_err(ctx, EDISON_E_SIZE, "f32 stream: more events than the stream was created for");
	ED_HIP(ctx, hipMemcpyAsync(s->d_audio + 256, samples, (size_t)n_events * 512 * sizeof(int16_t), hipMemcpyDeviceToDevice, ctx->stream));
	int r = edison_mfcc_f32_batch_dev(s->mfcc, s->d_audio, 2 * (int64_t)n_events, 256, s->d_feat + (size_t)s->rows * s->n_out, NULL, NULL);
	if (r != EDISON_OK) return r;
	const int64_t total = (int64_t)n_events * s->rows * s->n_out;
	int blocks = (int)((total + 255) / 256);
	if (blocks > 4096) blocks = 4096;
	hipLaunchKernelGGL(ed_f32_windows_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, s->d_feat, s->rows, s->n_out, n_events, windows);
	const size_t lds = (((size_t)s->rows * s->n_out + 15) & ~(size_t)15) + 512;
	hipLaunchKernelGGL(ed_f32_shift_kernel, dim3(1), dim3(256), lds, ctx->stream, s->d_feat, s->rows, s->n_out, n_events, s->d_audio);
	ED_HIP(ctx, hipGetLastError());
	s->events_seen += n_events;
	return EDISON_OK;
}

/* the same with host pointers; synchronous */
extern "C" int edison_f32_stream_push(edison_f32_stream *s, const int16_t *samples, int n_events, int8_t *windows)
{
	if (!s || n_events < 0 || (n_events > 0 && (!samples || !windows))) return EDISON_E_ARGUMENT;
	if (n_events == 0) return EDISON_OK;
	edison_ctx *ctx = s->ctx;
	if (n_events > s->max_events) return ed_set_err(ctx, EDISON_E_SIZE, "f32 stream: more events than the stream was created for");
	ED_HIP(ctx, hipSetDevice(ctx->device));
	dev_buf a, w;
	const size_t na = (size_t)n_events * 512 * sizeof(int16_t), nw = (size_t)n_events * s->rows * s->n_out;
	ED_HIP(ctx, a.alloc(na));
	ED_HIP(ctx, w.alloc(nw));
	ED_HIP(ctx, hipMemcpyAsync(a.p, samples, na, hipMemcpyHostToDevice, ctx->stream));
	int r = edison_f32_stream_push_dev(s, (const int16_t *)a.p, n_events, (int8_t *)w.p);
	if (r != EDISON_OK) return r;
	ED_HIP(ctx, hipMemcpyAsync(windows, w.p, nw, hipMemcpyDeviceToHost, ctx->stream));
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}
