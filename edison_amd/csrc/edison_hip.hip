/*
 * edison_hip.hip -- the thin C-ABI shim between the host-side C (tables.c, model.c, legacy.c) and the HIP
 * kernels (mfcc_kernels.hip, cnn_kernels.hip). Implements include/edison_hip.h; see that header for the
 * reference interface each entry point replaces.
 *
 * No CPU fallback lives here: every compute entry point ends in a kernel launch on a gfx950 device, and
 * edison_init fails with EDISON_E_NO_DEVICE when there is none.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "edison_ctx.h"

static char g_init_err[512] = "";

static int set_err(edison_ctx *ctx, int code, const char *msg) { return ed_set_err(ctx, code, msg); }

static inline int tab_index(int variant) { return variant == EDISON_MFCC_TF ? 2 : variant; }

static int upload_tables(edison_ctx *ctx, double fs, double lo, double hi, double scale)
{
	/* Tables are treated like the model: graphs captured by edison_stream objects hold the kernel instance picked for the
	 * table shape (mel_NLO / mel_NHI), the table addresses and, for variant C, d_q15 itself, and they run on the streams'
	 * private HIP streams. So: wait for the whole device before a table is overwritten or freed, and bump the epoch that
	 * edison_stream_push* checks -- a stream created before this call refuses further pushes. */
	ED_HIP(ctx, hipDeviceSynchronize());
	ctx->tables_epoch++;
	static const int variants[3] = {EDISON_MFCC_A, EDISON_MFCC_B, EDISON_MFCC_TF}; /* slot = tab_index(variant) */
	for (int v = 0; v < 3; v++)
	{
		ed_mfcc_tables_t *h = (ed_mfcc_tables_t *)malloc(sizeof(ed_mfcc_tables_t));
		if (!h) return set_err(ctx, EDISON_E_NO_MEMORY, "host allocation failed");
		int r = ed_build_mfcc_tables(variants[v], fs, lo, hi, scale, h, ctx->err, sizeof(ctx->err));
		if (r != EDISON_OK) { free(h); return r; }
		if (!ctx->d_tab[v]) ED_HIP(ctx, hipMalloc((void **)&ctx->d_tab[v], sizeof(ed_mfcc_tables_t)));
		/* synchronous w.r.t. the stream: a kernel in flight may still be reading the old tables */
		ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
		hipError_t e = hipMemcpy(ctx->d_tab[v], h, sizeof(ed_mfcc_tables_t), hipMemcpyHostToDevice);
		ctx->mel_NLO[v] = h->mel_NLO;
		ctx->mel_NHI[v] = h->mel_NHI;
		free(h);
		ED_HIP(ctx, e);
	}
	/* variant C: a filterbank / scale it cannot express only switches variant C off (EDISON_E_NO_IMPL on use) */
	{
		ed_q15_tables_t *h = (ed_q15_tables_t *)malloc(sizeof(ed_q15_tables_t));
		if (!h) return set_err(ctx, EDISON_E_NO_MEMORY, "host allocation failed");
		ctx->q15_err[0] = 0;
		int r = ed_build_q15_tables(fs, lo, hi, scale, h, ctx->q15_err, sizeof(ctx->q15_err));
		ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (r == EDISON_OK)
		{
			if (!ctx->d_q15) ED_HIP(ctx, hipMalloc((void **)&ctx->d_q15, sizeof(ed_q15_tables_t)));
			hipError_t e = hipMemcpy(ctx->d_q15, h, sizeof(ed_q15_tables_t), hipMemcpyHostToDevice);
			ctx->q15_nlo = h->mel_nlo;
			ctx->q15_nhi = h->mel_nhi;
			free(h);
			ED_HIP(ctx, e);
		}
		else
		{
			free(h);
			if (r != EDISON_E_NO_IMPL) return r;
			if (ctx->d_q15) { (void)hipFree(ctx->d_q15); ctx->d_q15 = NULL; }
		}
	}
	return EDISON_OK;
}

extern "C" int edison_init(int device, edison_ctx **out)
{
	if (!out) return EDISON_E_ARGUMENT;
	*out = NULL;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
	{
		snprintf(g_init_err, sizeof(g_init_err), "no HIP device visible (%s); libedison_hip has no CPU path",
		         e == hipSuccess ? "count 0" : hipGetErrorString(e));
		return EDISON_E_NO_DEVICE;
	}
	if (device < 0 || device >= n)
	{
		snprintf(g_init_err, sizeof(g_init_err), "device %d out of range (0..%d)", device, n - 1);
		return EDISON_E_ARGUMENT;
	}
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess)
	{
		snprintf(g_init_err, sizeof(g_init_err), "hipGetDeviceProperties failed");
		return EDISON_E_RUNTIME;
	}
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
	{
		snprintf(g_init_err, sizeof(g_init_err), "device %d is %s; this library is built for gfx950 only", device,
		         prop.gcnArchName);
		return EDISON_E_NO_DEVICE;
	}
	edison_ctx *ctx = (edison_ctx *)calloc(1, sizeof(edison_ctx));
	if (!ctx) return EDISON_E_NO_MEMORY;
	ctx->device = device;
	ctx->n_cu = prop.multiProcessorCount;
	ctx->hbm_bytes = prop.totalGlobalMem;
	snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
	int r = EDISON_OK;
	do {
		if (hipSetDevice(device) != hipSuccess) { r = EDISON_E_RUNTIME; break; }
		if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { r = EDISON_E_RUNTIME; break; }
		ctx->stream = ctx->own_stream;
		r = upload_tables(ctx, EDISON_FS, 80.0, 7600.0, 128.0); /* audio/config.py:14-15 */
	} while (0);
	if (r != EDISON_OK)
	{
		snprintf(g_init_err, sizeof(g_init_err), "edison_init: %s", ctx->err[0] ? ctx->err : "HIP setup failed");
		edison_shutdown(ctx);
		return r;
	}
	*out = ctx;
	return EDISON_OK;
}

extern "C" void edison_shutdown(edison_ctx *ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)hipDeviceSynchronize();
	(void)edison_dist_shutdown(ctx);
	for (int v = 0; v < 3; v++) if (ctx->d_tab[v]) (void)hipFree(ctx->d_tab[v]);
	if (ctx->d_q15) (void)hipFree(ctx->d_q15);
	if (ctx->d_model) (void)hipFree(ctx->d_model);
	if (ctx->d_model_mfma) (void)hipFree(ctx->d_model_mfma);
	if (ctx->d_net_plan) (void)hipFree(ctx->d_net_plan);
	if (ctx->d_net_w) (void)hipFree(ctx->d_net_w);
	if (ctx->d_net_seeds) (void)hipFree(ctx->d_net_seeds);
	ed_ctx_net_spec_drop(ctx);
	free(ctx->h_mm_plan);
	if (ctx->d_mm_plan) (void)hipFree(ctx->d_mm_plan);
	if (ctx->d_mm_frag) (void)hipFree(ctx->d_mm_frag);
	if (ctx->d_mm_seeds) (void)hipFree(ctx->d_mm_seeds);
	if (ctx->scratch) (void)hipFree(ctx->scratch);
	if (ctx->pipe_ready)
	{
		for (int k = 0; k < 5; k++) (void)hipStreamDestroy(ctx->pipe_cand[k]);
		for (int k = 0; k < 2; k++) (void)hipEventDestroy(ctx->pipe_join[k]);
		(void)hipEventDestroy(ctx->pipe_fork); (void)hipEventDestroy(ctx->pipe_t0); (void)hipEventDestroy(ctx->pipe_t1);
	}
	if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
	free(ctx);
}

extern "C" const char *edison_last_error(const edison_ctx *ctx) { return ctx ? ctx->err : g_init_err; }

extern "C" int edison_set_stream(edison_ctx *ctx, void *hip_stream)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	ctx->stream = (hipStream_t)hip_stream; /* NULL = HIP's default stream */
	return EDISON_OK;
}

extern "C" int edison_reset_stream(edison_ctx *ctx)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	ctx->stream = ctx->own_stream;
	return EDISON_OK;
}

extern "C" int edison_sync(edison_ctx *ctx)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

extern "C" int edison_device_info(edison_ctx *ctx, char *name, int name_cap, int *n_cu, int64_t *hbm_bytes)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", ctx->name);
	if (n_cu) *n_cu = ctx->n_cu;
	if (hbm_bytes) *hbm_bytes = (int64_t)ctx->hbm_bytes;
	return EDISON_OK;
}

extern "C" int edison_mfcc_configure(edison_ctx *ctx, double sample_rate, double lower_edge_hertz,
                                     double upper_edge_hertz, double mel_mtx_scale)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	if (!(sample_rate > 0) || !(lower_edge_hertz >= 0) || !(upper_edge_hertz > lower_edge_hertz) ||
	    !(upper_edge_hertz <= sample_rate / 2.0) || !(mel_mtx_scale > 0))
		return set_err(ctx, EDISON_E_ARGUMENT, "edison_mfcc_configure: bad filterbank edges");
	ED_HIP(ctx, hipSetDevice(ctx->device));
	return upload_tables(ctx, sample_rate, lower_edge_hertz, upper_edge_hertz, mel_mtx_scale);
}

extern "C" int edison_gen_mel_weight_matrix(int num_mel_bins, int num_spectrogram_bins, double sample_rate,
                                            double lower_edge_hertz, double upper_edge_hertz, double *W)
{
	return ed_gen_mel_weight_matrix(num_mel_bins, num_spectrogram_bins, sample_rate, lower_edge_hertz,
	                                upper_edge_hertz, W);
}

extern "C" int edison_model_load_mem(edison_ctx *ctx, const void *blob, size_t blob_bytes)
{
	if (!ctx || !blob) return EDISON_E_ARGUMENT;
	/* every graph gets the general plan; the kws_conv graph additionally gets the two specialised weight images */
	ed_net_plan_t *plan = (ed_net_plan_t *)malloc(sizeof(ed_net_plan_t));
	ed_cnn_model_t *h = (ed_cnn_model_t *)malloc(sizeof(ed_cnn_model_t));
	ed_cnn_mfma_model_t *hm = (ed_cnn_mfma_model_t *)malloc(sizeof(ed_cnn_mfma_model_t));
	int8_t *w = NULL;
	int32_t *seeds = NULL;
	int r = (plan && h && hm) ? ed_plan_net(blob, blob_bytes, plan, &w, &seeds, ctx->err, sizeof(ctx->err))
	                          : set_err(ctx, EDISON_E_NO_MEMORY, "host allocation failed");
	hipError_t e = hipSuccess;
	if (r == EDISON_OK)
	{
		char why[256];
		const int fast = ed_parse_model(blob, blob_bytes, h, hm, why, sizeof(why)) == EDISON_OK;
		e = hipSetDevice(ctx->device);
		if (e == hipSuccess) e = hipDeviceSynchronize(); /* streams of edison_stream objects may still read the old model */
		if (e == hipSuccess && fast && !ctx->d_model) e = hipMalloc((void **)&ctx->d_model, sizeof(ed_cnn_model_t));
		if (e == hipSuccess && fast && !ctx->d_model_mfma) e = hipMalloc((void **)&ctx->d_model_mfma, sizeof(ed_cnn_mfma_model_t));
		if (e == hipSuccess && fast) e = hipMemcpy(ctx->d_model, h, sizeof(ed_cnn_model_t), hipMemcpyHostToDevice);
		if (e == hipSuccess && fast) e = hipMemcpy(ctx->d_model_mfma, hm, sizeof(ed_cnn_mfma_model_t), hipMemcpyHostToDevice);
		ctx->have_model = 0;
		if (ctx->d_net_w) { (void)hipFree(ctx->d_net_w); ctx->d_net_w = NULL; }
		if (ctx->d_net_seeds) { (void)hipFree(ctx->d_net_seeds); ctx->d_net_seeds = NULL; }
		if (e == hipSuccess && !ctx->d_net_plan) e = hipMalloc((void **)&ctx->d_net_plan, sizeof(ed_net_plan_t));
		if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_net_w, (size_t)plan->weights_bytes + 16);
		if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_net_seeds, ((size_t)plan->n_seeds + 4) * sizeof(int32_t));
		if (e == hipSuccess) e = hipMemcpy(ctx->d_net_plan, plan, sizeof(ed_net_plan_t), hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMemcpy(ctx->d_net_w, w, (size_t)plan->weights_bytes, hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMemcpy(ctx->d_net_seeds, seeds, (size_t)plan->n_seeds * sizeof(int32_t), hipMemcpyHostToDevice);
		/* the matrix-core plan of the same graph (any graph: model_net_mm.c); without one the graph stays on the
		 * layer-by-layer kernel */
		ctx->mm_ok = 0;
		ed_ctx_net_spec_drop(ctx); /* the previous graph's own kernel (the device is idle: synchronised above) */
		free(ctx->h_mm_plan);
		ctx->h_mm_plan = NULL;
		if (ctx->d_mm_frag) { (void)hipFree(ctx->d_mm_frag); ctx->d_mm_frag = NULL; }
		if (ctx->d_mm_seeds) { (void)hipFree(ctx->d_mm_seeds); ctx->d_mm_seeds = NULL; }
		if (e == hipSuccess)
		{
			ed_mm_plan_t *mm = (ed_mm_plan_t *)malloc(sizeof(ed_mm_plan_t));
			int8_t *frag = NULL;
			int32_t *mseeds = NULL;
			if (mm && ed_plan_net_mm(blob, blob_bytes, plan, mm, &frag, &mseeds) == EDISON_OK && mm->ok)
			{
				if (!ctx->d_mm_plan) e = hipMalloc((void **)&ctx->d_mm_plan, sizeof(ed_mm_plan_t));
				if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_mm_frag, (size_t)mm->frag_bytes + 16);
				if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_mm_seeds, ((size_t)mm->n_seeds + 4) * sizeof(int32_t));
				if (e == hipSuccess) e = hipMemcpy(ctx->d_mm_plan, mm, sizeof(ed_mm_plan_t), hipMemcpyHostToDevice);
				if (e == hipSuccess) e = hipMemcpy(ctx->d_mm_frag, frag, (size_t)mm->frag_bytes, hipMemcpyHostToDevice);
				if (e == hipSuccess) e = hipMemcpy(ctx->d_mm_seeds, mseeds, (size_t)mm->n_seeds * sizeof(int32_t), hipMemcpyHostToDevice);
				if (e == hipSuccess)
				{
					ctx->mm_ok = 1; ctx->mm_lds = mm->lds_bytes; ctx->mm_batch = mm->batch; ctx->mm_waves = mm->waves; ctx->mm_frag_mode = mm->frag_mode;
					ctx->h_mm_plan = mm; /* kept: edison_net_specialize() generates the graph's own kernel from it */
					mm = NULL;
				}
				else
				{
					/* the matrix-core plan is an accelerator, not the model: if its upload fails the graph still loads and
					 * runs on the layer-by-layer kernel */
					(void)hipGetLastError();
					if (ctx->d_mm_frag) { (void)hipFree(ctx->d_mm_frag); ctx->d_mm_frag = NULL; }
					if (ctx->d_mm_seeds) { (void)hipFree(ctx->d_mm_seeds); ctx->d_mm_seeds = NULL; }
					e = hipSuccess;
				}
			}
			free(mm); free(frag); free(mseeds);
		}
		if (e == hipSuccess)
		{
			ctx->net = *plan;
			ctx->fast_model = fast;
			ctx->have_model = 1;
			ctx->model_epoch++;
		}
	}
	free(plan); free(h); free(hm); free(w); free(seeds);
	if (r != EDISON_OK) return r;
	ED_HIP(ctx, e);
	/* The graph's own kernel (edison_net_specialize, edison_net_jit.hip) is an explicit call, or an explicit wish: a model load by
	 * itself starts no compiler, writes no file and loads no code object from a cache (round-3 review: surprising in a drop-in
	 * library). EDISON_NET_SPECIALIZE (read at every load): unset / 0 = nothing; 1 = specialise every graph that has a
	 * matrix-core plan at load time (from the cache, else compiled now, ~1 s once per graph and machine); cache = take the own
	 * kernel only if an earlier call left it in the cache. A failure there (no compiler on the machine, ...) is not a failed load:
	 * the graph runs on the general kernel, edison_net_specialized() says which. */
	const char *env_spec = getenv("EDISON_NET_SPECIALIZE");
	const int always = env_spec && !strcmp(env_spec, "1"), cache_only = env_spec && !strcmp(env_spec, "cache");
	if (ctx->mm_ok && (always || cache_only))
	{
		if (always)
		{
			char keep[sizeof(ctx->err)];
			memcpy(keep, ctx->err, sizeof(keep));
			if (edison_net_specialize(ctx) != EDISON_OK) memcpy(ctx->err, keep, sizeof(keep));
		}
		else ed_ctx_net_spec_from_cache(ctx);
	}
	return EDISON_OK;
}

extern "C" int edison_model_load(edison_ctx *ctx, const char *ednn_path)
{
	if (!ctx || !ednn_path) return EDISON_E_ARGUMENT;
	FILE *f = fopen(ednn_path, "rb");
	if (!f)
	{
		snprintf(ctx->err, sizeof(ctx->err), "cannot open model file %s", ednn_path);
		return EDISON_E_ARGUMENT;
	}
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	if (n <= 0 || n > (16L << 20)) { fclose(f); return set_err(ctx, EDISON_E_LENGTH, "model file has an implausible size"); }
	void *buf = malloc((size_t)n);
	if (!buf) { fclose(f); return set_err(ctx, EDISON_E_NO_MEMORY, "host allocation failed"); }
	size_t got = fread(buf, 1, (size_t)n, f);
	fclose(f);
	int r = got == (size_t)n ? edison_model_load_mem(ctx, buf, (size_t)n) : set_err(ctx, EDISON_E_LENGTH, "short read on model file");
	free(buf);
	return r;
}

/* ---------------------------------------------------------------------------------------- device memory */
extern "C" int edison_dev_alloc(edison_ctx *ctx, size_t bytes, void **dptr)
{
	if (!ctx || !dptr) return EDISON_E_ARGUMENT;
	ED_HIP(ctx, hipSetDevice(ctx->device));
	hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
	if (e == hipErrorOutOfMemory) return set_err(ctx, EDISON_E_NO_MEMORY, "hipMalloc: out of HBM");
	ED_HIP(ctx, e);
	return EDISON_OK;
}
extern "C" int edison_dev_free(edison_ctx *ctx, void *dptr)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	ED_HIP(ctx, hipFree(dptr));
	return EDISON_OK;
}
extern "C" int edison_dev_upload(edison_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	ED_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}
extern "C" int edison_dev_download(edison_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	ED_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

static int ensure_scratch(edison_ctx *ctx, size_t bytes)
{
	if (bytes <= ctx->scratch_bytes) return EDISON_OK;
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (ctx->scratch) ED_HIP(ctx, hipFree(ctx->scratch));
	ctx->scratch = NULL; ctx->scratch_bytes = 0;
	hipError_t e = hipMalloc(&ctx->scratch, bytes);
	if (e == hipErrorOutOfMemory) return set_err(ctx, EDISON_E_NO_MEMORY, "scratch hipMalloc: out of HBM");
	ED_HIP(ctx, e);
	ctx->scratch_bytes = bytes;
	return EDISON_OK;
}

/* ---------------------------------------------------------------------------------------- hot path, device */
static int mfcc_launch_on(edison_ctx *ctx, hipStream_t stream, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                          int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale,
                          int stages, float *fft, float *spec, float *mel, float *logmel);

/* on the context's current stream */
static int mfcc_launch(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                       int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale,
                       int stages, float *fft, float *spec, float *mel, float *logmel)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	return mfcc_launch_on(ctx, ctx->stream, audio, n_frames, fpg, group_stride, frame_step, variant, n_coef, mfcc, feat, feat_scale,
	                      stages, fft, spec, mel, logmel);
}

int ed_ctx_mfcc_launch(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                       int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale,
                       int stages, float *fft, float *spec, float *mel, float *logmel)
{
	return mfcc_launch(ctx, audio, n_frames, fpg, group_stride, frame_step, variant, n_coef, mfcc, feat, feat_scale,
	                   stages, fft, spec, mel, logmel);
}

/* The one-frame microphone push in ONE launch (ed_kws1_kernel, cnn_mfma_kernels.hip): the MFCC (variant A / B, 13 int8 features,
 * scale 1) of the 1024 samples at `audio`, its row written to feat_row (the host's ring) and completing the 31-row window at
 * `window`, the kws_conv CNN on that window, outputs and the completion flag written by the kernel. EDISON_E_NO_IMPL when the
 * loaded model / variant has no such kernel (the caller launches the two kernels one after the other then). */
int ed_ctx_kws1_launch_on(edison_ctx *ctx, hipStream_t stream, const int16_t *audio, int variant, int8_t *feat_row, const int8_t *window,
                          int8_t *logits, int8_t *softmax, int32_t *argmax, unsigned *flag, unsigned seq, const ed_out_filter_t *filter)
{
	const int v = variant & 0xff;
	if (!ctx || !audio || !feat_row || !window) return EDISON_E_ARGUMENT;
	if (!ctx->have_model || !ctx->fast_model || (v != EDISON_MFCC_A && v != EDISON_MFCC_B)) return EDISON_E_NO_IMPL;
	ed_mfcc_args_t a;
	memset(&a, 0, sizeof(a));
	a.audio = audio; a.n_frames = 1; a.frames_per_group = 1; a.group_stride = 0; a.frame_step = EDISON_FRAME_LEN;
	a.n_coef = EDISON_NUM_MFCC; a.use_log = (variant & EDISON_MFCC_USE_LOG) ? 1 : 0;
	a.mel_NLO = ctx->mel_NLO[v];
	a.mel_NHI = ctx->mel_NHI[v];
	a.feat = feat_row; a.feat_scale = 1.0f;
	const int e = ed_launch_kws1(&a, ctx->d_tab[v], ctx->d_model_mfma, window, logits, softmax, argmax, flag, seq, filter, stream);
	if (e == (int)hipErrorInvalidValue) return EDISON_E_NO_IMPL;
	if (e != 0)
	{
		snprintf(ctx->err, sizeof(ctx->err), "one-frame KWS kernel launch failed: %s", hipGetErrorString((hipError_t)e));
		return EDISON_E_RUNTIME;
	}
	return EDISON_OK;
}

/* the same on an explicit stream (edison_stream.hip: a stream object launches on its private stream without touching
 * ctx->stream, which another thread's call on the context may be reading) */
int ed_ctx_mfcc_launch_on(edison_ctx *ctx, hipStream_t stream, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                          int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale,
                          int stages, float *fft, float *spec, float *mel, float *logmel)
{
	return mfcc_launch_on(ctx, stream, audio, n_frames, fpg, group_stride, frame_step, variant, n_coef, mfcc, feat, feat_scale,
	                      stages, fft, spec, mel, logmel);
}

static int mfcc_launch_on(edison_ctx *ctx, hipStream_t stream, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                          int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale,
                          int stages, float *fft, float *spec, float *mel, float *logmel)
{
	const int v = variant & 0xff;
	if (!ctx || (!audio && n_frames > 0)) return EDISON_E_ARGUMENT;
	if (v == EDISON_MFCC_C)
	{
		/* the firmware's integers through the float interface: mfcc = (float)int16 like the Cube branch of
		 * mfccToNetInput (app.c:680-683), feat = its NNoM branch (app.c:686-694, NNOM_INPUT_SCALE 1) */
		if ((variant & EDISON_MFCC_USE_LOG) || stages || feat_scale != 1.0f)
			return set_err(ctx, EDISON_E_NO_IMPL, "variant C: no log, feat_scale 1, stages through edison_mfcc_q15_stages");
		return ed_ctx_mfcc_q15_launch_on(ctx, stream, audio, n_frames, fpg, group_stride, frame_step, n_coef, NULL, mfcc, feat, 0,
		                                 NULL, NULL, NULL);
	}
	if (v != EDISON_MFCC_A && v != EDISON_MFCC_B && v != EDISON_MFCC_TF) return set_err(ctx, EDISON_E_ARGUMENT, "unknown MFCC variant");
	if (v == EDISON_MFCC_TF && (variant & EDISON_MFCC_USE_LOG)) return set_err(ctx, EDISON_E_ARGUMENT, "variant TF always takes the logarithm");
	const int ti = tab_index(v);
	if (n_coef < 1 || n_coef > EDISON_NUM_MEL) return set_err(ctx, EDISON_E_ARGUMENT, "n_coef must be 1..32");
	if (n_frames < 0 || n_frames >= ((int64_t)1 << 31) || frame_step < 0 || fpg < 1)
		return set_err(ctx, EDISON_E_ARGUMENT, "bad frame count / step");
	if (n_frames == 0) return EDISON_OK;
	ed_mfcc_args_t a;
	memset(&a, 0, sizeof(a));
	a.audio = audio; a.n_frames = n_frames; a.frames_per_group = fpg; a.group_stride = group_stride;
	a.frame_step = frame_step; a.n_coef = n_coef; a.use_log = (variant & EDISON_MFCC_USE_LOG) ? 1 : 0;
	a.mel_NLO = ctx->mel_NLO[ti];
	a.mel_NHI = ctx->mel_NHI[ti];
	a.mfcc = mfcc; a.feat = feat; a.feat_scale = feat_scale;
	a.fft = fft; a.spec = spec; a.mel = mel; a.logmel = logmel;
	a.window = v == EDISON_MFCC_TF;
	int e = ed_launch_mfcc(&a, ctx->d_tab[ti], stages, ctx->n_cu, stream);
	if (e != 0)
	{
		snprintf(ctx->err, sizeof(ctx->err), "MFCC kernel launch failed: %s", hipGetErrorString((hipError_t)e));
		return EDISON_E_RUNTIME;
	}
	return EDISON_OK;
}

extern "C" int edison_mfcc_batch_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                     int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale)
{
	return mfcc_launch(ctx, audio, n_frames, n_frames > 0 ? n_frames : 1, 0, frame_step, variant, n_coef, mfcc, feat,
	                   feat_scale, 0, NULL, NULL, NULL, NULL);
}

extern "C" int edison_mfcc_rows_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_rows, int64_t row_stride,
                                    int64_t frames_per_row, int64_t frame_step, int variant, int n_coef, float *mfcc,
                                    int8_t *feat, float feat_scale)
{
	if (!ctx || n_rows < 0 || frames_per_row < 0 || row_stride < 0) return EDISON_E_ARGUMENT;
	if (n_rows == 0 || frames_per_row == 0) return EDISON_OK;
	if (n_rows > INT32_MAX / frames_per_row) return set_err(ctx, EDISON_E_SIZE, "edison_mfcc_rows: more than 2^31 frames in one call");
	return mfcc_launch(ctx, audio, n_rows * frames_per_row, frames_per_row, row_stride, frame_step, variant, n_coef, mfcc, feat,
	                   feat_scale, 0, NULL, NULL, NULL, NULL);
}

/* Up to 16 INDEPENDENT batches -- any addresses, own outputs -- per launch; more than 16 go out as several launches. Variants A, B
 * (+ USE_LOG) on the two-frame kernel; the batches of one call share n_frames_each, frame_step, variant, n_coef and feat_scale. */
extern "C" int edison_mfcc_batches_dev(edison_ctx *ctx, int n_batches, const int16_t *const *audio, int64_t n_frames_each, int64_t frame_step,
                                       int variant, int n_coef, float *const *mfcc, int8_t *const *feat, float feat_scale)
{
	const int v = variant & 0xff;
	if (!ctx || n_batches < 0 || (n_batches > 0 && !audio) || n_frames_each < 0 || frame_step < 0) return EDISON_E_ARGUMENT;
	if (v != EDISON_MFCC_A && v != EDISON_MFCC_B) return set_err(ctx, EDISON_E_NO_IMPL, "edison_mfcc_batches_dev: variants A and B (the other variants: one call per batch)");
	if (n_coef < 1 || n_coef > EDISON_NUM_MEL) return set_err(ctx, EDISON_E_ARGUMENT, "n_coef must be 1..32");
	if (n_batches == 0 || n_frames_each == 0) return EDISON_OK;
	if (n_frames_each > INT32_MAX / ED_MFCC_LIST_MAX) return set_err(ctx, EDISON_E_SIZE, "edison_mfcc_batches_dev: more than 2^31 frames in one launch");
	for (int b = 0; b < n_batches; b++)
		if (!audio[b] || (mfcc && !mfcc[b]) || (feat && !feat[b])) return set_err(ctx, EDISON_E_ARGUMENT, "edison_mfcc_batches_dev: a NULL pointer in a batch list");
	for (int b0 = 0; b0 < n_batches; b0 += ED_MFCC_LIST_MAX)
	{
		const int nb = n_batches - b0 < ED_MFCC_LIST_MAX ? n_batches - b0 : ED_MFCC_LIST_MAX;
		ed_mfcc_list_t list;
		memset(&list, 0, sizeof(list));
		for (int b = 0; b < nb; b++)
		{
			list.audio[b] = audio[b0 + b];
			list.mfcc[b] = mfcc ? mfcc[b0 + b] : NULL;
			list.feat[b] = feat ? feat[b0 + b] : NULL;
		}
		ed_mfcc_args_t a;
		memset(&a, 0, sizeof(a));
		a.audio = list.audio[0]; a.n_frames = (int64_t)nb * n_frames_each; a.frames_per_group = n_frames_each; a.group_stride = 0;
		a.frame_step = frame_step; a.n_coef = n_coef; a.use_log = (variant & EDISON_MFCC_USE_LOG) ? 1 : 0;
		a.mel_NLO = ctx->mel_NLO[v];
		a.mel_NHI = ctx->mel_NHI[v];
		a.mfcc = list.mfcc[0]; a.feat = list.feat[0]; a.feat_scale = feat_scale; /* flags: which outputs the launch writes */
		const int e = ed_launch_mfcc_list(&a, &list, nb, ctx->d_tab[v], ctx->n_cu, ctx->stream);
		if (e != 0)
		{
			snprintf(ctx->err, sizeof(ctx->err), "MFCC list kernel launch failed: %s", hipGetErrorString((hipError_t)e));
			return EDISON_E_RUNTIME;
		}
	}
	return EDISON_OK;
}

struct dev_buf
{
	void *p;
	dev_buf() : p(NULL) {}
	~dev_buf() { if (p) (void)hipFree(p); }
	hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};

/* ---- two queues for independent batches: the NEXT launch is in flight while this one drains -------------------------------------
 * A 65 536-frame launch idles ~15 % of its window (256 workgroups starting up, waves leaving over the last pair time) and the next
 * launch of the same queue starts only when this one has completed. On a second hardware queue the next launch's workgroups are
 * dispatched onto CUs as this launch's leave them. Worth +4 ... 5 % from a host that keeps both queues fed (a C host: 47.7 -> 45.7 us
 * per batch; profiles/r05_mfcc_two_queues_notes.txt), nothing when the host needs as long per call as the GPU per batch. */
static int pipe_setup(edison_ctx *ctx)
{
	if (ctx->pipe_ready) return EDISON_OK;
	ED_HIP(ctx, hipSetDevice(ctx->device));
	int least = 0, greatest = 0;
	ED_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
	hipError_t e = hipSuccess;
	int n_streams = 0, n_events = 0;
	hipEvent_t *ev[5] = {&ctx->pipe_fork, &ctx->pipe_join[0], &ctx->pipe_join[1], &ctx->pipe_t0, &ctx->pipe_t1};
	for (int k = 0; k < 5 && e == hipSuccess; k++)
	{
		e = hipStreamCreateWithPriority(&ctx->pipe_cand[k], hipStreamNonBlocking, k < 3 ? least : greatest);
		if (e == hipSuccess) n_streams++;
	}
	for (int k = 0; k < 5 && e == hipSuccess; k++)
	{
		e = k < 3 ? hipEventCreateWithFlags(ev[k], hipEventDisableTiming) : hipEventCreate(ev[k]); /* the last two time the calibration */
		if (e == hipSuccess) n_events++;
	}
	if (e != hipSuccess)
	{
		/* all or nothing: a half-built set is taken down again, the next call starts over */
		for (int k = 0; k < n_events; k++) (void)hipEventDestroy(*ev[k]);
		for (int k = 0; k < n_streams; k++) (void)hipStreamDestroy(ctx->pipe_cand[k]);
		ED_HIP(ctx, e);
	}
	/* until a calibration says otherwise: one stream of each priority -- different hardware queues by construction */
	ctx->pipe_pair[0] = 0; ctx->pipe_pair[1] = 3;
	ctx->pipe_q[0] = ctx->pipe_cand[0]; ctx->pipe_q[1] = ctx->pipe_cand[3];
	ctx->pipe_ready = 1;
	return EDISON_OK;
}

extern "C" int edison_queues_fork(edison_ctx *ctx)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	{ const int r = pipe_setup(ctx); if (r != EDISON_OK) return r; }
	/* both queues start behind everything the caller's stream holds so far (the producers of the batches) */
	ED_HIP(ctx, hipEventRecord(ctx->pipe_fork, ctx->stream));
	ED_HIP(ctx, hipStreamWaitEvent(ctx->pipe_q[0], ctx->pipe_fork, 0));
	ED_HIP(ctx, hipStreamWaitEvent(ctx->pipe_q[1], ctx->pipe_fork, 0));
	ctx->pipe_forked = 1;
	return EDISON_OK;
}

extern "C" int edison_queues_join(edison_ctx *ctx)
{
	if (!ctx) return EDISON_E_ARGUMENT;
	if (!ctx->pipe_forked) return EDISON_OK;
	for (int k = 0; k < 2; k++)
	{
		ED_HIP(ctx, hipEventRecord(ctx->pipe_join[k], ctx->pipe_q[k]));
		ED_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_join[k], 0));
	}
	ctx->pipe_forked = 0;
	return EDISON_OK;
}

/* `n` launches of the caller's batch, alternating over streams a and b (a == b: the serial sequence), timed with events on the
 * context's stream around a fork / join; microseconds per launch */
static int pipe_time_pair(edison_ctx *ctx, hipStream_t a, hipStream_t b, const int16_t *audio, int64_t n_frames, int64_t frame_step, int variant,
                          float *const out[2], int n, double *us)
{
	hipStream_t q[2] = {a, b};
	for (int timed = 0; timed < 2; timed++)
	{
		ED_HIP(ctx, hipEventRecord(timed ? ctx->pipe_t0 : ctx->pipe_fork, ctx->stream));
		for (int k = 0; k < 2; k++) ED_HIP(ctx, hipStreamWaitEvent(q[k], timed ? ctx->pipe_t0 : ctx->pipe_fork, 0));
		const int m = timed ? n : 6;
		for (int i = 0; i < m; i++)
		{
			const int r = mfcc_launch_on(ctx, q[i & 1], audio, n_frames, n_frames, 0, frame_step, variant, EDISON_NUM_MFCC, out[i & 1], NULL, 1.0f, 0, NULL, NULL, NULL, NULL);
			if (r != EDISON_OK) return r;
		}
		for (int k = 0; k < 2; k++)
		{
			ED_HIP(ctx, hipEventRecord(ctx->pipe_join[k], q[k]));
			ED_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_join[k], 0));
		}
	}
	ED_HIP(ctx, hipEventRecord(ctx->pipe_t1, ctx->stream));
	ED_HIP(ctx, hipEventSynchronize(ctx->pipe_t1));
	float ms = 0;
	ED_HIP(ctx, hipEventElapsedTime(&ms, ctx->pipe_t0, ctx->pipe_t1));
	*us = (double)ms * 1e3 / n;
	return EDISON_OK;
}

static double median_of(double *v, int n)
{
	for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (v[j] < v[i]) { const double t = v[i]; v[i] = v[j]; v[j] = t; }
	return v[n / 2];
}

/* Which two of the context's candidate streams let two launches of THIS workload overlap profitably -- if any. Whether the next
 * launch backfills the CUs this one leaves depends on where the runtime and the driver put the streams' hardware queues, which
 * nothing in the HIP API controls: of all pairs of nine streams in one process about a third gained (+1 % on a box whose serial launch
 * takes 43.5 us, +4 ... 7 % on boxes at 47-48 us), a third changed nothing and a third LOST 8-10 % (profiles/r05_mfcc_two_queues_notes.txt).
 * So the library measures: every pair of its five candidates and the serial sequence, interleaved, on the caller's own batch (device
 * pointer, read only; outputs go to scratch), then the winner against the serial sequence once more; it keeps the pair only if it
 * is at least 1 % faster, otherwise both queue indices mean ONE stream and the queue calls are the serial sequence. ~0.17 s at 65 536 frames. */
extern "C" int edison_queues_calibrate(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step, int variant,
                                       double *serial_us, double *best_us, int *pair_kept)
{
	if (!ctx || !audio || n_frames < 1 || frame_step < 0) return EDISON_E_ARGUMENT;
	if (ctx->pipe_forked) return set_err(ctx, EDISON_E_ARGUMENT, "edison_queues_calibrate between fork and join");
	const int v = variant & 0xff;
	if (v != EDISON_MFCC_A && v != EDISON_MFCC_B) return set_err(ctx, EDISON_E_NO_IMPL, "edison_queues_calibrate: variants A and B");
	{ const int r = pipe_setup(ctx); if (r != EDISON_OK) return r; }
	dev_buf o0, o1;
	const size_t ob = (size_t)n_frames * EDISON_NUM_MFCC * sizeof(float);
	ED_HIP(ctx, o0.alloc(ob));
	ED_HIP(ctx, o1.alloc(ob));
	float *const out[2] = {(float *)o0.p, (float *)o1.p};
	enum { NC = 5, NP = NC * (NC - 1) / 2, ROUNDS = 5, LAUNCHES = 32 };
	int pa[NP + 1], pb[NP + 1], np = 0;
	pa[np] = 0; pb[np] = 0; np++;                                  /* entry 0: the serial sequence */
	for (int i = 0; i < NC; i++) for (int j = i + 1; j < NC; j++) { pa[np] = i; pb[np] = j; np++; }
	double t[NP + 1][ROUNDS];
	/* A launch that takes milliseconds has nothing to gain from a second queue (its fixed cost is a few microseconds) and the calibration would
	 * take seconds: one short serial block decides, and such a batch keeps one queue. */
	{
		double probe = 0;
		const int rc = pipe_time_pair(ctx, ctx->pipe_cand[0], ctx->pipe_cand[0], audio, n_frames, frame_step, variant, out, 4, &probe);
		if (rc != EDISON_OK) return rc;
		if (probe > 1000.0)
		{
			ctx->pipe_pair[0] = ctx->pipe_pair[1] = 0;
			ctx->pipe_q[0] = ctx->pipe_q[1] = ctx->pipe_cand[0];
			ctx->pipe_cal_serial_us = ctx->pipe_cal_best_us = probe;
			if (serial_us) *serial_us = probe;
			if (best_us) *best_us = probe;
			if (pair_kept) *pair_kept = 0;
			ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
			return EDISON_OK;
		}
	}
	for (int r = 0; r < ROUNDS; r++)
		for (int k0 = 0; k0 < np; k0++)
		{
			const int k = (r & 1) ? np - 1 - k0 : k0;                 /* interleaved, the order reversed every other round */
			const int rc = pipe_time_pair(ctx, ctx->pipe_cand[pa[k]], ctx->pipe_cand[pb[k]], audio, n_frames, frame_step, variant, out, LAUNCHES, &t[k][r]);
			if (rc != EDISON_OK) return rc;
		}
	const double serial = median_of(t[0], ROUNDS);
	int best = 0;
	double best_t = serial;
	for (int k = 1; k < np; k++) { const double m = median_of(t[k], ROUNDS); if (m < best_t) { best_t = m; best = k; } }
	int keep = 0;
	double s2 = serial, b2 = best_t;
	if (best > 0)
	{
		/* the winner against the serial sequence once more, longer: a pair that won on noise does not win twice */
		double ts[ROUNDS], tb[ROUNDS];
		for (int r = 0; r < ROUNDS; r++)
		{
			int rc = pipe_time_pair(ctx, ctx->pipe_cand[0], ctx->pipe_cand[0], audio, n_frames, frame_step, variant, out, 2 * LAUNCHES, &ts[r]);
			if (rc == EDISON_OK) rc = pipe_time_pair(ctx, ctx->pipe_cand[pa[best]], ctx->pipe_cand[pb[best]], audio, n_frames, frame_step, variant, out, 2 * LAUNCHES, &tb[r]);
			if (rc != EDISON_OK) return rc;
		}
		s2 = median_of(ts, ROUNDS); b2 = median_of(tb, ROUNDS);
		keep = b2 < 0.99 * s2;
	}
	ctx->pipe_pair[0] = keep ? pa[best] : 0;
	ctx->pipe_pair[1] = keep ? pb[best] : 0;
	ctx->pipe_q[0] = ctx->pipe_cand[ctx->pipe_pair[0]];
	ctx->pipe_q[1] = ctx->pipe_cand[ctx->pipe_pair[1]];
	ctx->pipe_cal_serial_us = s2; ctx->pipe_cal_best_us = keep ? b2 : s2;
	if (serial_us) *serial_us = s2;
	if (best_us) *best_us = keep ? b2 : s2;
	if (pair_kept) *pair_kept = keep ? 10 * pa[best] + pb[best] : 0;
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

extern "C" int edison_mfcc_batch_queue_dev(edison_ctx *ctx, int queue, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                           int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale)
{
	if (!ctx || queue < 0 || queue > 1) return EDISON_E_ARGUMENT;
	if (!ctx->pipe_forked) return set_err(ctx, EDISON_E_ARGUMENT, "edison_mfcc_batch_queue_dev outside edison_queues_fork ... edison_queues_join");
	return mfcc_launch_on(ctx, ctx->pipe_q[queue], audio, n_frames, n_frames > 0 ? n_frames : 1, 0, frame_step, variant, n_coef, mfcc, feat, feat_scale,
	                      0, NULL, NULL, NULL, NULL);
}

extern "C" int edison_mfcc_stages_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                      int variant, float *fft, float *spec, float *mel, float *logmel, float *mfcc32)
{
	return mfcc_launch(ctx, audio, n_frames, n_frames > 0 ? n_frames : 1, 0, frame_step, variant, EDISON_NUM_MEL,
	                   mfcc32, NULL, 1.0f, 1, fft, spec, mel, logmel);
}

static int kws_shaped(const ed_net_plan_t *p)
{
	return p->in_h == EDISON_UTT_FRAMES && p->in_w == EDISON_NUM_MFCC && p->in_c == 1 && p->out_n == EDISON_NET_OUT && p->has_softmax;
}

/* any loaded graph without per-layer dumps: the matrix-core kernel when the graph has a plan for it (EDISON_NET_NO_MFMA=1
 * keeps the layer-by-layer kernel, for A/B measurements), else the layer-by-layer kernel */
int ed_ctx_net_launch(edison_ctx *ctx, const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	return ed_ctx_net_launch_on(ctx, ctx->stream, in, n, in_stride, logits, softmax, argmax);
}

int ed_ctx_net_launch_on(edison_ctx *ctx, hipStream_t stream, const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	return ed_ctx_net_launch_flag(ctx, stream, in, n, in_stride, logits, softmax, argmax, NULL, 0, NULL);
}

int ed_ctx_net_launch_flag(edison_ctx *ctx, hipStream_t stream, const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax, int32_t *argmax,
                           unsigned *flag, unsigned seq, int *flag_written)
{
	if (flag_written) *flag_written = 0;
	static const int no_mfma = getenv("EDISON_NET_NO_MFMA") ? atoi(getenv("EDISON_NET_NO_MFMA")) : 0;
	if (ctx->mm_ok && !no_mfma && ctx->spec_fn && ctx->spec_epoch == ctx->model_epoch)
		return ed_ctx_net_spec_launch(ctx, stream, in, n, in_stride, logits, softmax, argmax, flag, seq, flag_written);
	if (ctx->mm_ok && !no_mfma)
		return ed_launch_net_mfma(ctx->d_net_plan, ctx->d_mm_plan, ctx->d_mm_frag, ctx->d_mm_seeds, ctx->mm_lds, ctx->mm_batch, ctx->mm_waves, ctx->mm_frag_mode, in, n,
		                          in_stride, logits, softmax, argmax, ctx->n_cu, stream, flag, seq, flag_written);
	return ed_launch_net(ctx->d_net_plan, ctx->d_net_w, ctx->d_net_seeds, ctx->net.lds_bytes, in, n, in_stride, logits, softmax, argmax,
	                     NULL, ctx->n_cu, stream);
}

int ed_ctx_kws_cnn_launch(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int64_t feat_stride, int8_t *logits,
                          int8_t *softmax, int32_t *argmax)
{
	return ed_ctx_kws_cnn_launch_on(ctx, ctx->stream, feat, n_utt, feat_stride, logits, softmax, argmax);
}

int ed_ctx_kws_cnn_launch_on(edison_ctx *ctx, hipStream_t stream, const int8_t *feat, int64_t n_utt, int64_t feat_stride, int8_t *logits,
                             int8_t *softmax, int32_t *argmax)
{
	return ed_ctx_kws_cnn_launch_flag(ctx, stream, feat, n_utt, feat_stride, logits, softmax, argmax, NULL, 0, NULL);
}

int ed_ctx_kws_cnn_launch_flag(edison_ctx *ctx, hipStream_t stream, const int8_t *feat, int64_t n_utt, int64_t feat_stride, int8_t *logits,
                               int8_t *softmax, int32_t *argmax, unsigned *flag, unsigned seq, int *flag_written)
{
	if (flag_written) *flag_written = 0;
	if (!ctx->have_model) return set_err(ctx, EDISON_E_NO_MODEL, "no CNN model loaded (edison_model_load)");
	if (!ctx->fast_model && !kws_shaped(&ctx->net))
		return set_err(ctx, EDISON_E_SIZE, "the loaded model is not a 31x13x1 -> 10 softmax classifier; use edison_net_batch");
	int e = ctx->fast_model
	            ? ed_launch_cnn_mfma_flag(ctx->d_model_mfma, feat, n_utt, feat_stride, logits, softmax, argmax, ctx->n_cu, stream, flag, seq, flag_written)
	            : ed_ctx_net_launch_flag(ctx, stream, feat, n_utt, feat_stride, logits, softmax, argmax, flag, seq, flag_written);
	if (e != 0)
	{
		snprintf(ctx->err, sizeof(ctx->err), "CNN kernel launch failed: %s", hipGetErrorString((hipError_t)e));
		return EDISON_E_RUNTIME;
	}
	return EDISON_OK;
}

static int cnn_launch(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *logits, int8_t *softmax,
                      int32_t *argmax, int8_t *acts)
{
	if (!ctx || n_utt < 0 || (!feat && n_utt > 0)) return EDISON_E_ARGUMENT;
	if (!ctx->have_model) return set_err(ctx, EDISON_E_NO_MODEL, "no CNN model loaded (edison_model_load)");
	if (n_utt == 0) return EDISON_OK;
	if (!acts) return ed_ctx_kws_cnn_launch(ctx, feat, n_utt, EDISON_NET_IN, logits, softmax, argmax);
	/* per-layer activations in the kws_conv layout come from that graph's layer-by-layer kernel */
	if (!ctx->fast_model)
		return set_err(ctx, EDISON_E_SIZE, "edison_cnn_layers dumps the kws_conv layout; use edison_net_layers for this model");
	int e = ed_launch_cnn(ctx->d_model, feat, n_utt, logits, softmax, argmax, acts, ctx->n_cu, ctx->stream);
	if (e != 0)
	{
		snprintf(ctx->err, sizeof(ctx->err), "CNN kernel launch failed: %s", hipGetErrorString((hipError_t)e));
		return EDISON_E_RUNTIME;
	}
	return EDISON_OK;
}

extern "C" int edison_cnn_batch_dev(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *logits,
                                    int8_t *softmax, int32_t *argmax)
{
	return cnn_launch(ctx, feat, n_utt, logits, softmax, argmax, NULL);
}

extern "C" int edison_cnn_layers_dev(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *acts)
{
	if (!acts && n_utt > 0) return EDISON_E_ARGUMENT;
	return cnn_launch(ctx, feat, n_utt, NULL, NULL, NULL, acts);
}

static int kws_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride, int variant, int8_t *feat,
                   int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	if (!ctx || n_utt < 0 || (!audio && n_utt > 0)) return EDISON_E_ARGUMENT;
	if (!ctx->have_model) return set_err(ctx, EDISON_E_NO_MODEL, "no CNN model loaded (edison_model_load)");
	if (utt_stride < 0) return set_err(ctx, EDISON_E_ARGUMENT, "negative utterance stride");
	if (n_utt == 0) return EDISON_OK;
	if (n_utt * EDISON_UTT_FRAMES >= ((int64_t)1 << 31)) return set_err(ctx, EDISON_E_ARGUMENT, "too many utterances per call");
	int8_t *f = feat;
	if (!f)
	{
		int r = ensure_scratch(ctx, (size_t)n_utt * EDISON_NET_IN);
		if (r != EDISON_OK) return r;
		f = (int8_t *)ctx->scratch;
	}
	/* first 13 coefficients, scale 1 (nnom_net_input_scale, audio/config.py:41; NNOM_INPUT_SCALE, weights.h:162) */
	int r = mfcc_launch(ctx, audio, n_utt * EDISON_UTT_FRAMES, EDISON_UTT_FRAMES, utt_stride, EDISON_FRAME_LEN,
	                    variant, EDISON_NUM_MFCC, NULL, f, 1.0f, 0, NULL, NULL, NULL, NULL);
	if (r != EDISON_OK) return r;
	return cnn_launch(ctx, f, n_utt, logits, softmax, argmax, NULL);
}

extern "C" int edison_kws_batch_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride,
                                    int8_t *feat, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	return kws_dev(ctx, audio, n_utt, utt_stride, EDISON_MFCC_B, feat, logits, softmax, argmax);
}

extern "C" int edison_kws_batch_q15_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride,
                                        int8_t *feat, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	return kws_dev(ctx, audio, n_utt, utt_stride, EDISON_MFCC_C, feat, logits, softmax, argmax);
}

/* ---------------------------------------------------------------------------------------- hot path, host  */

#define ED_UP(ctx, dst, src, n) ED_HIP(ctx, hipMemcpyAsync((dst), (src), (n), hipMemcpyHostToDevice, (ctx)->stream))
#define ED_DOWN(ctx, dst, src, n) \
	do { if (dst) ED_HIP(ctx, hipMemcpyAsync((dst), (src), (n), hipMemcpyDeviceToHost, (ctx)->stream)); } while (0)

static size_t audio_span(int64_t n_frames, int64_t frame_step) { return (size_t)((n_frames - 1) * frame_step + EDISON_FRAME_LEN); }

extern "C" int edison_mfcc_batch(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                 int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale)
{
	if (!ctx || n_frames < 0 || (!audio && n_frames > 0) || frame_step < 0) return EDISON_E_ARGUMENT;
	if (n_frames == 0) return EDISON_OK;
	if (n_coef < 1 || n_coef > EDISON_NUM_MEL) return set_err(ctx, EDISON_E_ARGUMENT, "n_coef must be 1..32");
	ED_HIP(ctx, hipSetDevice(ctx->device));
	dev_buf a, m, q;
	const size_t na = audio_span(n_frames, frame_step) * sizeof(int16_t);
	ED_HIP(ctx, a.alloc(na));
	if (mfcc) ED_HIP(ctx, m.alloc((size_t)n_frames * n_coef * sizeof(float)));
	if (feat) ED_HIP(ctx, q.alloc((size_t)n_frames * n_coef));
	ED_UP(ctx, a.p, audio, na);
	int r = edison_mfcc_batch_dev(ctx, (const int16_t *)a.p, n_frames, frame_step, variant, n_coef, (float *)m.p,
	                              (int8_t *)q.p, feat_scale);
	if (r != EDISON_OK) return r;
	ED_DOWN(ctx, mfcc, m.p, (size_t)n_frames * n_coef * sizeof(float));
	ED_DOWN(ctx, feat, q.p, (size_t)n_frames * n_coef);
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

extern "C" int edison_mfcc_rows(edison_ctx *ctx, const int16_t *audio, int64_t n_rows, int64_t row_stride, int64_t frames_per_row,
                                int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale)
{
	if (!ctx || n_rows < 0 || frames_per_row < 0 || row_stride < 0 || frame_step < 0 || (!audio && n_rows > 0)) return EDISON_E_ARGUMENT;
	if (n_rows == 0 || frames_per_row == 0) return EDISON_OK;
	if (n_coef < 1 || n_coef > EDISON_NUM_MEL) return set_err(ctx, EDISON_E_ARGUMENT, "n_coef must be 1..32");
	if (n_rows > INT32_MAX / frames_per_row) return set_err(ctx, EDISON_E_SIZE, "edison_mfcc_rows: more than 2^31 frames in one call");
	/* the staging buffer spans (n_rows - 1) * row_stride + the frames of one row: computed in 128 bits, refused beyond 2^46
	 * samples (a wrapped size_t would allocate a small buffer and let the kernel read past it) */
	const size_t span = audio_span(frames_per_row, frame_step);
	const unsigned __int128 na128 = ((unsigned __int128)(n_rows - 1) * (unsigned __int128)row_stride + span) * sizeof(int16_t);
	if (na128 > ((unsigned __int128)1 << 47)) return set_err(ctx, EDISON_E_SIZE, "edison_mfcc_rows: row_stride x n_rows too large");
	ED_HIP(ctx, hipSetDevice(ctx->device));
	dev_buf a, m, q;
	const size_t n = (size_t)(n_rows * frames_per_row);
	const size_t na = (size_t)na128;
	ED_HIP(ctx, a.alloc(na));
	if (mfcc) ED_HIP(ctx, m.alloc(n * n_coef * sizeof(float)));
	if (feat) ED_HIP(ctx, q.alloc(n * n_coef));
	ED_UP(ctx, a.p, audio, na);
	int r = edison_mfcc_rows_dev(ctx, (const int16_t *)a.p, n_rows, row_stride, frames_per_row, frame_step, variant, n_coef,
	                             (float *)m.p, (int8_t *)q.p, feat_scale);
	if (r != EDISON_OK) return r;
	ED_DOWN(ctx, mfcc, m.p, n * n_coef * sizeof(float));
	ED_DOWN(ctx, feat, q.p, n * n_coef);
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

extern "C" int edison_mfcc_stages(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                                  int variant, float *fft, float *spec, float *mel, float *logmel, float *mfcc32)
{
	if (!ctx || n_frames < 0 || (!audio && n_frames > 0) || frame_step < 0) return EDISON_E_ARGUMENT;
	if (n_frames == 0) return EDISON_OK;
	ED_HIP(ctx, hipSetDevice(ctx->device));
	dev_buf a, f, s, m, l, c;
	const size_t na = audio_span(n_frames, frame_step) * sizeof(int16_t), n = (size_t)n_frames;
	ED_HIP(ctx, a.alloc(na));
	if (fft) ED_HIP(ctx, f.alloc(n * 513 * 2 * sizeof(float)));
	if (spec) ED_HIP(ctx, s.alloc(n * 513 * sizeof(float)));
	if (mel) ED_HIP(ctx, m.alloc(n * 32 * sizeof(float)));
	if (logmel) ED_HIP(ctx, l.alloc(n * 32 * sizeof(float)));
	if (mfcc32) ED_HIP(ctx, c.alloc(n * 32 * sizeof(float)));
	ED_UP(ctx, a.p, audio, na);
	int r = edison_mfcc_stages_dev(ctx, (const int16_t *)a.p, n_frames, frame_step, variant, (float *)f.p, (float *)s.p,
	                               (float *)m.p, (float *)l.p, (float *)c.p);
	if (r != EDISON_OK) return r;
	ED_DOWN(ctx, fft, f.p, n * 513 * 2 * sizeof(float));
	ED_DOWN(ctx, spec, s.p, n * 513 * sizeof(float));
	ED_DOWN(ctx, mel, m.p, n * 32 * sizeof(float));
	ED_DOWN(ctx, logmel, l.p, n * 32 * sizeof(float));
	ED_DOWN(ctx, mfcc32, c.p, n * 32 * sizeof(float));
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

static int cnn_host(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *logits, int8_t *softmax,
                    int32_t *argmax, int8_t *acts)
{
	if (!ctx || n_utt < 0 || (!feat && n_utt > 0)) return EDISON_E_ARGUMENT;
	if (!ctx->have_model) return set_err(ctx, EDISON_E_NO_MODEL, "no CNN model loaded (edison_model_load)");
	if (n_utt == 0) return EDISON_OK;
	ED_HIP(ctx, hipSetDevice(ctx->device));
	dev_buf f, l, s, a, t;
	const size_t n = (size_t)n_utt;
	ED_HIP(ctx, f.alloc(n * EDISON_NET_IN));
	if (logits) ED_HIP(ctx, l.alloc(n * EDISON_NET_OUT));
	if (softmax) ED_HIP(ctx, s.alloc(n * EDISON_NET_OUT));
	if (argmax) ED_HIP(ctx, a.alloc(n * sizeof(int32_t)));
	if (acts) ED_HIP(ctx, t.alloc(n * EDISON_CNN_ACT_BYTES));
	ED_UP(ctx, f.p, feat, n * EDISON_NET_IN);
	int r = cnn_launch(ctx, (const int8_t *)f.p, n_utt, (int8_t *)l.p, (int8_t *)s.p, (int32_t *)a.p, (int8_t *)t.p);
	if (r != EDISON_OK) return r;
	ED_DOWN(ctx, logits, l.p, n * EDISON_NET_OUT);
	ED_DOWN(ctx, softmax, s.p, n * EDISON_NET_OUT);
	ED_DOWN(ctx, argmax, a.p, n * sizeof(int32_t));
	ED_DOWN(ctx, acts, t.p, n * EDISON_CNN_ACT_BYTES);
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

extern "C" int edison_cnn_batch(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *logits, int8_t *softmax,
                                int32_t *argmax)
{
	return cnn_host(ctx, feat, n_utt, logits, softmax, argmax, NULL);
}

extern "C" int edison_cnn_layers(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *acts)
{
	if (!acts && n_utt > 0) return EDISON_E_ARGUMENT;
	return cnn_host(ctx, feat, n_utt, NULL, NULL, NULL, acts);
}

static int kws_host(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride, int variant, int8_t *feat,
                    int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	if (!ctx || n_utt < 0 || (!audio && n_utt > 0) || utt_stride < 0) return EDISON_E_ARGUMENT;
	if (n_utt == 0) return EDISON_OK;
	ED_HIP(ctx, hipSetDevice(ctx->device));
	dev_buf au, f, l, s, a;
	const size_t n = (size_t)n_utt;
	const size_t na = ((size_t)(n_utt - 1) * (size_t)utt_stride + (size_t)EDISON_UTT_FRAMES * EDISON_FRAME_LEN) * sizeof(int16_t);
	ED_HIP(ctx, au.alloc(na));
	ED_HIP(ctx, f.alloc(n * EDISON_NET_IN));
	if (logits) ED_HIP(ctx, l.alloc(n * EDISON_NET_OUT));
	if (softmax) ED_HIP(ctx, s.alloc(n * EDISON_NET_OUT));
	if (argmax) ED_HIP(ctx, a.alloc(n * sizeof(int32_t)));
	ED_UP(ctx, au.p, audio, na);
	int r = kws_dev(ctx, (const int16_t *)au.p, n_utt, utt_stride, variant, (int8_t *)f.p, (int8_t *)l.p, (int8_t *)s.p,
	                (int32_t *)a.p);
	if (r != EDISON_OK) return r;
	ED_DOWN(ctx, feat, f.p, n * EDISON_NET_IN);
	ED_DOWN(ctx, logits, l.p, n * EDISON_NET_OUT);
	ED_DOWN(ctx, softmax, s.p, n * EDISON_NET_OUT);
	ED_DOWN(ctx, argmax, a.p, n * sizeof(int32_t));
	ED_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return EDISON_OK;
}

extern "C" int edison_kws_batch(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride,
                                int8_t *feat, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	return kws_host(ctx, audio, n_utt, utt_stride, EDISON_MFCC_B, feat, logits, softmax, argmax);
}

extern "C" int edison_kws_batch_q15(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride,
                                    int8_t *feat, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	return kws_host(ctx, audio, n_utt, utt_stride, EDISON_MFCC_C, feat, logits, softmax, argmax);
}
