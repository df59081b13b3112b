/*
 * edison_ctx.h -- the context object behind the opaque `edison_ctx *` of include/edison_hip.h, shared by the
 * HIP shim files (edison_hip.hip, edison_stream.hip). Not part of the public ABI.
 */
#ifndef EDISON_CTX_H
#define EDISON_CTX_H

#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

extern "C" int ed_launch_mfcc(const ed_mfcc_args_t *args, const ed_mfcc_tables_t *dev_tab, int stages, int n_cu,
                              hipStream_t stream);
extern "C" int ed_launch_mfcc_list(const ed_mfcc_args_t *args, const ed_mfcc_list_t *list, int n_batches, const ed_mfcc_tables_t *dev_tab, int n_cu,
                                   hipStream_t stream);
extern "C" int ed_launch_cnn(const ed_cnn_model_t *dev_model, const int8_t *feat, int64_t n_utt, int8_t *logits,
                             int8_t *softmax, int32_t *argmax, int8_t *acts, int n_cu, hipStream_t stream);
extern "C" int ed_launch_cnn_mfma(const ed_cnn_mfma_model_t *dev_model, const int8_t *feat, int64_t n_utt,
                                  int64_t feat_stride, int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu,
                                  hipStream_t stream);

extern "C" int ed_launch_cnn_mfma_flag(const ed_cnn_mfma_model_t *dev_model, const int8_t *feat, int64_t n_utt,
                                       int64_t feat_stride, int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu,
                                       hipStream_t stream, unsigned *done_flag, unsigned done_seq, int *flag_written);

extern "C" int ed_launch_net(const ed_net_plan_t *dev_plan, const int8_t *dev_w, const int32_t *dev_seeds, int lds_bytes,
                             const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax, int32_t *argmax,
                             int8_t *acts, int n_cu, hipStream_t stream);

extern "C" int ed_launch_net_mfma(const ed_net_plan_t *dev_plan, const ed_mm_plan_t *dev_mm, const int8_t *dev_frag,
                                  const int32_t *dev_seeds, int lds_bytes, int batch, int waves, int frag_mode, const int8_t *in, int64_t n,
                                  int64_t in_stride, int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu, hipStream_t stream,
                                  unsigned *done_flag, unsigned done_seq, int *flag_written);

extern "C" int ed_launch_mfcc_q15(const ed_mfcc_q15_args_t *args, const ed_q15_tables_t *dev_tab, int mel_nlo, int mel_nhi,
                                  int stages, int n_cu, hipStream_t stream);

struct edison_ctx
{
	int device;
	int n_cu;
	size_t hbm_bytes;
	char name[128];
	hipStream_t own_stream;
	hipStream_t stream;
	ed_mfcc_tables_t *d_tab[3]; /* variant A, B, TF (ed_tab_index) */
	int mel_NLO[3], mel_NHI[3];
	ed_q15_tables_t *d_q15; /* variant C (firmware Q15); NULL when the configured filterbank does not fit it */
	int q15_nlo, q15_nhi;   /* host copy of the table shape: selects the kernel instance */
	char q15_err[160];      /* why variant C is unavailable, when it is */
	ed_cnn_model_t *d_model;           /* layer-by-layer diagnostic kernel (edison_cnn_layers) */
	ed_cnn_mfma_model_t *d_model_mfma; /* MFMA fast path                                       */
	int have_model;                    /* a model is loaded (any graph the planner accepts)    */
	int fast_model;                    /* ... and it is the kws_conv graph the two kernels above are specialised for */
	int tables_epoch;                  /* counts edison_mfcc_configure calls: captured graphs also bake in the table shape and addresses */
	int model_epoch;                   /* counts loads: captured graphs (edison_stream) hold device addresses of one load */
	/* the general layer-by-layer path (cnn_net_kernels.hip): plan (host copy + device copy), weights, seeds */
	ed_net_plan_t net;
	ed_net_plan_t *d_net_plan;
	int8_t *d_net_w;
	int32_t *d_net_seeds;
	/* ... and its matrix-core plan (cnn_net_mfma_kernels.hip), when the graph has one */
	int mm_ok, mm_lds, mm_batch, mm_waves, mm_frag_mode;
	ed_mm_plan_t *d_mm_plan;
	int8_t *d_mm_frag;
	int32_t *d_mm_seeds;
	/* ... and the graph's OWN kernel (edison_net_jit.hip: the same source compiled by hipRTC with this plan as constants) */
	ed_mm_plan_t *h_mm_plan; /* host copy of the matrix-core plan: what the specialisation is generated from */
	void *spec_mod, *spec_fn; /* hipModule_t / hipFunction_t */
	int spec_epoch;           /* the model load (model_epoch) it was compiled for */
	int spec_state;           /* 1: compiled just now by the hipcc child process, 2: loaded from the on-disk cache, 3: compiled just now by hipRTC in this process */
	/* growable device scratch for the host-pointer entry points and the fused KWS path */
	void *scratch;
	size_t scratch_bytes;
	/* multi-GPU (edison_dist.hip): the RCCL communicator this context belongs to, NULL for a single-GPU context */
	void *dist_comm;
	int dist_rank, dist_world;
	/* two library-owned HIP queues for independent batches (edison_queues_fork / edison_mfcc_batch_queue_dev / edison_queues_join):
	 * created at the first fork with DIFFERENT priorities, i.e. on different hardware queues by construction (HIP shares a pool of
	 * hardware queues among the streams of one priority) */
	hipStream_t pipe_q[2];       /* the pair in use (two of pipe_cand, or twice the same one after a calibration that found no winning pair) */
	hipStream_t pipe_cand[5];    /* candidates: 3 of the least, 2 of the greatest priority */
	hipEvent_t pipe_fork, pipe_join[2], pipe_t0, pipe_t1;
	int pipe_ready, pipe_forked;
	int pipe_pair[2];            /* indices into pipe_cand */
	double pipe_cal_serial_us, pipe_cal_best_us; /* edison_queues_calibrate's findings (0: never calibrated) */
	void *dist_scratch; /* padded send + receive blocks of edison_dist_allgather_logits_total (unequal shards) */
	size_t dist_scratch_bytes;
	char err[512];
};

#define ED_HIP(ctx, call)                                                                                     \
	do {                                                                                                      \
		hipError_t e_ = (call);                                                                               \
		if (e_ != hipSuccess)                                                                                 \
		{                                                                                                     \
			snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
			         __FILE__, __LINE__);                                                                     \
			return EDISON_E_RUNTIME;                                                                          \
		}                                                                                                     \
	} while (0)

static inline int ed_set_err(edison_ctx *ctx, int code, const char *msg)
{
	if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", msg);
	return code;
}

/* The loaded model on 31x13x1 -> 10 features (the geometry of every kws / stream entry point): matrix-core kernel for
 * the kws_conv graph, the general kernel for any other graph of that shape. feat_stride = bytes between utterances. */
int ed_ctx_kws_cnn_launch(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int64_t feat_stride, int8_t *logits,
                          int8_t *softmax, int32_t *argmax);

/* Any loaded graph, no per-layer dumps: matrix-core kernel if the graph has a plan for it, else the layer-by-layer one. */
int ed_ctx_net_launch(edison_ctx *ctx, const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax, int32_t *argmax);

/* Fill the launch arguments of the MFCC kernel for `variant` and enqueue it on the context's stream. */
int ed_ctx_mfcc_launch(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                       int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale,
                       int stages, float *fft, float *spec, float *mel, float *logmel);
/* Variant C: same frame addressing, int16 / float / int8 outputs and int16 stage dumps (edison_q15.hip). */
int ed_ctx_mfcc_q15_launch(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                           int64_t frame_step, int n_coef, int16_t *mfcc_i16, float *mfcc_f32, int8_t *feat, int stages,
                           int16_t *fft, int16_t *spec, int16_t *mel);

/* the same launches on an explicit stream instead of ctx->stream (edison_stream.hip's private streams) */
int ed_ctx_kws_cnn_launch_on(edison_ctx *ctx, hipStream_t stream, const int8_t *feat, int64_t n_utt, int64_t feat_stride, int8_t *logits,
                             int8_t *softmax, int32_t *argmax);
/* ... and, where the specialised kernel runs a single group, let it write `seq` to the host-mapped `flag` behind its outputs;
 * *flag_written = 0 when the caller has to signal completion itself */
int ed_ctx_kws_cnn_launch_flag(edison_ctx *ctx, hipStream_t stream, const int8_t *feat, int64_t n_utt, int64_t feat_stride, int8_t *logits,
                               int8_t *softmax, int32_t *argmax, unsigned *flag, unsigned seq, int *flag_written);
int ed_ctx_net_launch_on(edison_ctx *ctx, hipStream_t stream, const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax,
                         int32_t *argmax);
/* the one-frame microphone push in one launch (ed_kws1_kernel) */
extern "C" int ed_launch_kws1(const ed_mfcc_args_t *margs, const ed_mfcc_tables_t *dev_tab, const ed_cnn_mfma_model_t *dev_model, const int8_t *feat,
                              int8_t *logits, int8_t *softmax, int32_t *argmax, unsigned *done_flag, unsigned done_seq, const ed_out_filter_t *filter,
                              hipStream_t stream);
int ed_ctx_kws1_launch_on(edison_ctx *ctx, hipStream_t stream, const int16_t *audio, int variant, int8_t *feat_row, const int8_t *window,
                          int8_t *logits, int8_t *softmax, int32_t *argmax, unsigned *flag, unsigned seq, const ed_out_filter_t *filter);
/* edison_net_jit.hip */
int ed_ctx_net_spec_launch(edison_ctx *ctx, hipStream_t stream, const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax,
                           int32_t *argmax, unsigned *done_flag, unsigned done_seq, int *flag_written);
/* ed_ctx_net_launch_on with the completion flag of the one-window microphone push (see ed_ctx_kws_cnn_launch_flag) */
int ed_ctx_net_launch_flag(edison_ctx *ctx, hipStream_t stream, const int8_t *in, int64_t n, int64_t in_stride, int8_t *logits, int8_t *softmax,
                           int32_t *argmax, unsigned *flag, unsigned seq, int *flag_written);
void ed_ctx_net_spec_drop(edison_ctx *ctx);
void ed_ctx_net_spec_from_cache(edison_ctx *ctx);
int ed_ctx_mfcc_launch_on(edison_ctx *ctx, hipStream_t stream, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                          int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale,
                          int stages, float *fft, float *spec, float *mel, float *logmel);
int ed_ctx_mfcc_q15_launch_on(edison_ctx *ctx, hipStream_t stream, const int16_t *audio, int64_t n_frames, int64_t fpg, int64_t group_stride,
                              int64_t frame_step, int n_coef, int16_t *mfcc_i16, float *mfcc_f32, int8_t *feat, int stages,
                              int16_t *fft, int16_t *spec, int16_t *mel);

#endif
