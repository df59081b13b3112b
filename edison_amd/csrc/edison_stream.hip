/*
 * edison_stream.hip -- continuous-microphone mode on the GPU: the counterpart of the firmware's
 * appMicMfccInfereContinuous (firmware/src/app.c:288-371) and appAudioEvent (app.c:635-663):
 *
 *     per new 1024-sample frame:  audioCalcMFCCs -> mfccToNetInputPush (drop the oldest of 31 rows, append the
 *                                 newest, app.c:706-719) -> aiRunInference on the 31 x 13 window
 *                                 -> moving average over the outputs, arm_max_f32, threshold (app.c:341-356)
 *
 * Here a push delivers `chunk` hops of new samples at once (chunk = 1 for lowest latency, thousands for
 * throughput). Frames may overlap (hop 512 = the 50 % overlap of BASELINE config 5; hop 1024 = the shipped
 * firmware cadence, audio/config.py:27). Device-resident state: the last 1024 - hop samples, the last 30
 * feature rows and the ten filtered outputs. The sliding window is never copied per frame: window i of a push is
 * rows i..i+30 of one [30 + chunk][13] int8 buffer, which the CNN kernel reads with a 13-byte "utterance" stride.
 *
 * The device operations of a push are the MFCC kernel, the CNN kernel, the output filter and the history shift, launched
 * directly on a private stream (a captured hipGraph of the same nodes exists -- EDISON_STREAM_GRAPH=1 -- but replaying it
 * measured slower than the plain launches on this platform; host pushes of 16 KB..1 MB still use the staged graph that
 * also holds the upload and the download). Host-pointer pushes of a few frames (the microphone case) run against
 * host-mapped buffers with two or three launches and no copies at all (see the struct). The net input
 * starts as zeros like the firmware's static netInput buffer, the filter state as zeros like netOutFilt (app.c:299-300).
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <immintrin.h>

#include "edison_ctx.h"
#include "edison_fsm_core.h"

struct edison_stream
{
	edison_ctx *ctx;
	int hop, chunk, tail; /* tail = 1024 - hop samples of history */
	int last_n;           /* inferences of the LAST push (chunk, or fewer after edison_stream_push_n_dev): what the filtered / state outputs hold */
	int variant;          /* EDISON_MFCC_B or EDISON_MFCC_C */
	int filter;
	int use_graph;        /* edison_stream_opts.launch_mode */
	double alpha, one_minus_alpha, threshold;
	int16_t *d_audio;     /* [tail + slots*chunk*hop]                */
	int8_t *d_feat;       /* [(30 + slots*chunk) * 13]               */
	/* Direct device pushes SLIDE through buffers `slots` pushes long instead of moving the history back to the front after
	 * every push: a_pos / f_pos = where the history (tail samples / 30 rows) starts now. The shift kernel runs when the next
	 * push would not fit -- one launch in `slots` pushes instead of one per push (a push of 4096 frames is ~29 us of GPU time,
	 * most of it kernel boundaries). Every other path (captured graphs hold addresses) first brings the history to the front. */
	int slots, a_pos, f_pos;
	int8_t *d_soft;       /* [chunk * 10]                            */
	int8_t *d_logits;     /* [chunk * 10]                            */
	int32_t *d_argmax;    /* [chunk]                                 */
	float *d_filt_state;  /* [10]  netOutFilt                        */
	int fsm;              /* the state machine runs behind the filter (opts.fsm) */
	edison_fsm *d_fsm;    /* the machine, device memory              */
	ed_fsm_roles_t roles;
	uint32_t dt_us;       /* time between two inferences = hop / 16 kHz */
	float *d_filt;        /* [chunk * 10] netOutFilt after each inference of the push */
	int32_t *d_likely;    /* [chunk] arm_max_f32 index               */
	int32_t *d_spotted;   /* [chunk] that index if the maximum exceeds the threshold, else -1 */
	hipStream_t own;      /* the graph always runs on this private stream (the default stream cannot be captured);  */
	hipEvent_t ev_in, ev_out; /* ordered against the context's current stream with events                      */
	hipGraph_t graph;
	hipGraphExec_t exec;
	/* host-pointer pushes of small chunks: pinned staging buffers and a second graph that also holds the upload and the
	 * downloads, so that a push is one memcpy in, ONE graph launch, one wait, one memcpy out (microphone latency path) */
	unsigned char *d_out; /* one block: logits | softmax | argmax | filt | likely | spotted at the offsets below */
	int16_t *h_in;
	unsigned char *h_out;
	size_t off_soft, off_argmax, off_filt, off_likely, off_spotted, off_fsm_states, off_fsm, h_out_bytes;
	hipGraph_t graph_h;
	hipGraphExec_t exec_h;
	int last_push_staged;
	/* host-pointer pushes of a few frames (the microphone case, one frame per push): NO copy nodes at all. The audio ring,
	 * the feature rows and the outputs live in pinned host memory that the kernels read and write over the bus (2 KB of
	 * samples, 403 feature bytes, 24 bytes of results per frame); the host shifts the ring and the rows itself. The device
	 * work is MFCC -> CNN (-> filter), launched DIRECTLY: measured on MI355X (tools/lab/lat_parts.py, bench.py streaming),
	 * an empty kernel's launch + synchronize is 16 us, the staged six-node graph 43 us, the same two kernels as a graph
	 * 39 us, launched directly 32 us -- hipGraphLaunch costs more than two plain launches. */
	int16_t *m_audio;     /* [tail + chunk*hop], pinned + mapped */
	int8_t *m_feat;       /* [(30 + chunk) * 13]                 */
	unsigned char *m_out; /* the d_out block's layout            */
	int mapped;           /* 1: this stream has the host-mapped path */
	/* completion of a mapped push: the command processor writes a sequence number into host-mapped memory behind the
	 * last kernel (hipStreamWriteValue32) and the host spins on it: 2.8 us less than hipStreamSynchronize on this
	 * platform (tools/ubench/launch_lat: 13.7 us -> 10.9 us for two dependent empty kernels) */
	volatile unsigned *m_flag;
	unsigned *md_flag;
	unsigned flag_seq;
	const int16_t *md_audio; int8_t *md_feat; unsigned char *md_out; /* their device addresses */
	/* Direct-launch device pushes run on the CALLER's stream (round 3): no event pair around the push, the CNN writes the
	 * caller's output buffers itself -- a push is one copy and three launches instead of eleven API calls (the loop over a
	 * long recording in 4096-frame pushes was bound by the host's calls, not by the GPU). q_last / q_pending: the stream the
	 * last such push went to; whatever next touches the stream's state on another HIP stream waits for it first. */
	hipStream_t q_last;
	int q_pending;
	int state_host;       /* 1: the newest tail samples / 30 feature rows are in m_audio / m_feat, 0: in d_audio / d_feat */
	int last_push_mapped;
	int64_t frames_seen;
	int tables_epoch;     /* likewise for the MFCC tables (edison_mfcc_configure) */
	int model_epoch;      /* the graphs hold the device addresses of the model that was loaded when they were captured */
};

#define ED_STREAM_STAGED_MAX_BYTES (1u << 20) /* chunks above 1 MB of samples go through plain async copies */
#define ED_STREAM_MAPPED_MAX_BYTES (16u << 10) /* pushes of at most 16 KB of samples run against host-mapped buffers */

__global__ void ed_stream_shift_kernel(int16_t *dst_audio, const int16_t *src_audio, int tail, int8_t *dst_feat, const int8_t *src_feat)
{
	/* the newest `tail` samples and the newest 30 feature rows to the front of their buffers. One workgroup: read
	 * everything into registers first, then barrier, then write (source and destination may overlap). */
	const int t = threadIdx.x;
	int16_t a[4];
	int8_t f[2];
	for (int i = 0; i < 4; i++) { const int j = t + i * 256; a[i] = j < tail ? src_audio[j] : (int16_t)0; }
	for (int i = 0; i < 2; i++) { const int j = t + i * 256; f[i] = j < 30 * EDISON_NUM_MFCC ? src_feat[j] : (int8_t)0; }
	__syncthreads();
	for (int i = 0; i < 4; i++) { const int j = t + i * 256; if (j < tail) dst_audio[j] = a[i]; }
	for (int i = 0; i < 2; i++) { const int j = t + i * 256; if (j < 30 * EDISON_NUM_MFCC) dst_feat[j] = f[i]; }
}

/*
 * The firmware's post-processing of the network output (app.c:332-356), for the `chunk` inferences of a push:
 *   netOutFloat[i] = (float)netOutput[i]
 *   netOutFilt[i]  = (ALPHA*netOutFilt[i] + (1.0-ALPHA)*netOutFloat[i])      double arithmetic (the constants are
 *                                                                           doubles), rounded to float on the store
 *   arm_max_f32(netOutFilt, 10, &predMax, &predMaxIdx)                      first maximum
 *   spotted = predMax > TRUE_THRESHOLD
 * The recurrence rounds at every step, so it is sequential in time by definition; the ten classes run on ten
 * lanes, the products (1-ALPHA)*x come from a 256-entry table (x is an int8), and the per-frame maximum is taken in
 * parallel afterwards. One workgroup.
 */
/* the state machine behind the filter: where its state lives and what a step needs (edison_fsm_core.h); fsm = NULL: none */
struct ed_fsm_stage_t
{
	edison_fsm *fsm;      /* device memory, read and written                       */
	int32_t *states;      /* [chunk] out: the state after each inference           */
	edison_fsm *copy;     /* out: the machine after the push (the host's view), or NULL */
	uint32_t dt_us;
	ed_fsm_roles_t roles;
};

__global__ __launch_bounds__(256) void ed_stream_filter_kernel(const int8_t *soft, int chunk, double alpha, double one_minus_alpha,
                                                               double threshold, float *state, float *filt, int32_t *likely,
                                                               int32_t *spotted, ed_fsm_stage_t fs)
{
	__shared__ double bx[256];
	const int t = threadIdx.x;
	bx[t] = one_minus_alpha * (double)(float)(int8_t)(t - 128);
	__syncthreads();
	if (t < EDISON_NET_OUT)
	{
		/* the product and the sum are rounded SEPARATELY (the Cortex-M4 computes doubles in software, nothing is fused): HIP's
		 * __dmul_rn / __dadd_rn are plain * and +, which the compiler's default contraction turns into one v_fma_f64 -- a
		 * different double in the last place, and a different float wherever that sum sits next to a float rounding boundary
		 * (found in round 4 by scripted softmax streams: one value in ~4 000; network outputs of 0 and 127 never showed it) */
		float y = state[t];
		for (int i = 0; i < chunk; i++)
		{
#pragma clang fp contract(off)
			const int x = soft[(size_t)i * EDISON_NET_OUT + t];
			const double a = alpha * (double)y; /* (not __dmul_rn / __dadd_rn: their bodies are compiled with the header's own contraction) */
			y = (float)(a + bx[x + 128]);
			filt[(size_t)i * EDISON_NET_OUT + t] = y;
		}
		state[t] = y;
	}
	__syncthreads();
	for (int i = t; i < chunk; i += 256)
	{
		const float *row = filt + (size_t)i * EDISON_NET_OUT;
		float best = row[0];
		int idx = 0;
		for (int c = 1; c < EDISON_NET_OUT; c++)
			if (best < row[c]) { best = row[c]; idx = c; }
		likely[i] = idx;
		spotted[i] = ((double)best > threshold) ? idx : -1;
	}
	if (!fs.fsm) return;
	/* edisonFSM (app.c:371, 727-928), one step per inference in time order: sequential by definition, so ONE lane walks the
	 * (likely, spotted) pairs of the push -- a handful of integer operations per step; the machine stays in registers */
	__syncthreads();
	if (t == 0)
	{
		edison_fsm m = *fs.fsm;
		for (int i = 0; i < chunk; i++)
			fs.states[i] = ed_fsm_step_core(&m, spotted[i] >= 0, (uint32_t)likely[i], fs.dt_us, &fs.roles);
		*fs.fsm = m;
		if (fs.copy) *fs.copy = m;
	}
}

static int enqueue_mapped_push(edison_stream *s, unsigned seq, int *flag_written);

/* the state-machine stage of a push whose outputs go to the block `out` (device view: the stream's own block, or the mapped one) */
static ed_fsm_stage_t fsm_stage(const edison_stream *s, unsigned char *out)
{
	ed_fsm_stage_t fs;
	memset(&fs, 0, sizeof(fs));
	if (!s->fsm) return fs;
	fs.fsm = s->d_fsm;
	fs.states = (int32_t *)(out + s->off_fsm_states);
	fs.copy = (edison_fsm *)(out + s->off_fsm);
	fs.dt_us = s->dt_us;
	fs.roles = s->roles;
	return fs;
}

/* enqueue the device operations of a push on hipStream q (passed explicitly: ctx->stream is not touched); the CNN writes
 * logits / softmax / argmax where it is told to (the stream's own block, or the caller's buffers) */
static int enqueue_push_on(edison_stream *s, hipStream_t q, int8_t *logits, int8_t *softmax, int32_t *argmax, int slide = 0, int n = 0)
{
	edison_ctx *ctx = s->ctx;
	if (n <= 0) n = s->chunk;       /* frames of this push: the stream's chunk, or fewer (edison_stream_push_n_dev: a ragged last push) */
	if (n != s->chunk) slide = 0;   /* the sliding bookkeeping counts whole chunks: a short push brings the history back to the front */
	int16_t *audio = s->d_audio + s->a_pos;                          /* history, then the new samples */
	int8_t *feat = s->d_feat + (size_t)s->f_pos * EDISON_NUM_MFCC;   /* 30 rows of history, then the new rows */
	/* 13 coefficients, int8 net input (scale 1): rows 30.. of the feature buffer */
	int r = ed_ctx_mfcc_launch_on(ctx, q, audio, n, n, 0, s->hop, s->variant, EDISON_NUM_MFCC, NULL,
	                              feat + 30 * EDISON_NUM_MFCC, 1.0f, 0, NULL, NULL, NULL, NULL);
	if (r != EDISON_OK) return r;
	/* window i of the push = rows i..i+30 of the feature buffer: a 13-byte utterance stride, nothing is copied */
	r = ed_ctx_kws_cnn_launch_on(ctx, q, feat, n, EDISON_NUM_MFCC, logits, softmax, argmax);
	if (r != EDISON_OK) return r;
	if (s->filter)
	{
		hipLaunchKernelGGL(ed_stream_filter_kernel, dim3(1), dim3(256), 0, q, softmax, n, s->alpha,
		                   s->one_minus_alpha, s->threshold, s->d_filt_state, s->d_filt, s->d_likely, s->d_spotted, fsm_stage(s, s->d_out));
		if (hipGetLastError() != hipSuccess) return ed_set_err(ctx, EDISON_E_RUNTIME, "stream: filter launch failed");
	}
	const size_t nnew = (size_t)n * s->hop; /* < 2^30: edison_stream_create_ex */
	if (slide && (size_t)s->a_pos / nnew + 2 <= (size_t)s->slots)
	{
		/* the next push finds its history behind this push's samples / rows and still fits: nothing moves */
		s->a_pos += (int)nnew;
		s->f_pos += n;
		return EDISON_OK;
	}
	hipLaunchKernelGGL(ed_stream_shift_kernel, dim3(1), dim3(256), 0, q, s->d_audio, audio + nnew, s->tail,
	                   s->d_feat, feat + (size_t)n * EDISON_NUM_MFCC);
	s->a_pos = s->f_pos = 0; /* (a captured graph always runs at position 0 and ends here) */
	return hipGetLastError() == hipSuccess ? EDISON_OK : ed_set_err(ctx, EDISON_E_RUNTIME, "stream: shift launch failed");
}

/* every path but the sliding device push wants the history at the front of the buffers */
static int history_to_front(edison_stream *s, hipStream_t q)
{
	if (!s->a_pos && !s->f_pos) return EDISON_OK;
	hipLaunchKernelGGL(ed_stream_shift_kernel, dim3(1), dim3(256), 0, q, s->d_audio, s->d_audio + s->a_pos, s->tail,
	                   s->d_feat, s->d_feat + (size_t)s->f_pos * EDISON_NUM_MFCC);
	s->a_pos = s->f_pos = 0;
	return hipGetLastError() == hipSuccess ? EDISON_OK : ed_set_err(s->ctx, EDISON_E_RUNTIME, "stream: shift launch failed");
}

static int enqueue_push(edison_stream *s) { return enqueue_push_on(s, s->own, s->d_logits, s->d_soft, s->d_argmax); }

/* Work that a direct device push left on the caller's stream must be behind us before the private stream (or another caller
 * stream `q`) touches the stream's state: one event pair, paid only at such a change of streams. */
static int drain_q(edison_stream *s, hipStream_t q)
{
	edison_ctx *ctx = s->ctx;
	if (!s->q_pending || s->q_last == q) return EDISON_OK;
	if (hipEventRecord(s->ev_in, s->q_last) == hipSuccess) ED_HIP(ctx, hipStreamWaitEvent(q, s->ev_in, 0));
	else (void)hipGetLastError(); /* the caller destroyed that stream (which drains it) */
	s->q_pending = 0;
	return EDISON_OK;
}

/* the device work of a push against the host-mapped buffers, on the stream's private hipStream. When the CNN is the last
 * kernel (no output filter, or the one-launch kernel that filters itself) and runs as one group, it writes the completion
 * sequence number itself (*flag_written = 1). */
static int enqueue_mapped_push(edison_stream *s, unsigned seq, int *flag_written)
{
	edison_ctx *ctx = s->ctx;
	unsigned char *o8 = s->md_out;
	/* one frame, float features, the kws_conv model: MFCC and CNN in ONE launch (ed_kws1_kernel) -- the
	 * dependent-kernel boundary, the second launch call and the wait of the CNN's weight staging behind the MFCC are gone
	 * (C host, one box: p50 18.3 -> see DESIGN 7). EDISON_STREAM_NO_FUSED=1: A/B knob. */
	static const int no_fused = getenv("EDISON_STREAM_NO_FUSED") ? atoi(getenv("EDISON_STREAM_NO_FUSED")) : 0;
	int r = EDISON_E_NO_IMPL;
	if (s->chunk == 1 && !no_fused)
	{
		/* the stream's output filter (app.c:341-356) for this one inference is done by the same kernel, behind its softmax */
		ed_out_filter_t f;
		memset(&f, 0, sizeof(f));
		f.alpha = s->alpha; f.one_minus_alpha = s->one_minus_alpha; f.threshold = s->threshold;
		f.state = s->d_filt_state; f.filt = (float *)(o8 + s->off_filt); f.likely = (int32_t *)(o8 + s->off_likely); f.spotted = (int32_t *)(o8 + s->off_spotted);
		if (s->fsm)
		{
			f.fsm = s->d_fsm; f.fsm_state = (int32_t *)(o8 + s->off_fsm_states); f.fsm_copy = o8 + s->off_fsm;
			f.dt_us = s->dt_us; f.wake_idx = s->roles.wake_idx; f.loc_mask = s->roles.loc_mask; f.val_mask = s->roles.val_mask;
		}
		r = ed_ctx_kws1_launch_on(ctx, s->own, s->md_audio, s->variant, s->md_feat + 30 * EDISON_NUM_MFCC, s->md_feat, (int8_t *)o8,
		                          (int8_t *)(o8 + s->off_soft), (int32_t *)(o8 + s->off_argmax), s->md_flag, seq, s->filter ? &f : NULL);
		if (r == EDISON_OK) { *flag_written = 1; return EDISON_OK; }
		if (r != EDISON_E_NO_IMPL) return r;
	}
	if (r == EDISON_E_NO_IMPL) /* no one-launch kernel for this model / variant / chunk: the two kernels one after the other */
	{
		r = ed_ctx_mfcc_launch_on(ctx, s->own, s->md_audio, s->chunk, s->chunk, 0, s->hop, s->variant, EDISON_NUM_MFCC, NULL,
		                          s->md_feat + 30 * EDISON_NUM_MFCC, 1.0f, 0, NULL, NULL, NULL, NULL);
		if (r == EDISON_OK)
			r = ed_ctx_kws_cnn_launch_flag(ctx, s->own, s->md_feat, s->chunk, EDISON_NUM_MFCC, (int8_t *)o8, (int8_t *)(o8 + s->off_soft), (int32_t *)(o8 + s->off_argmax),
			                               s->filter ? NULL : s->md_flag, seq, flag_written);
	}
	if (r == EDISON_OK && s->filter)
	{
		hipLaunchKernelGGL(ed_stream_filter_kernel, dim3(1), dim3(256), 0, s->own, (const int8_t *)(o8 + s->off_soft), s->chunk, s->alpha,
		                   s->one_minus_alpha, s->threshold, s->d_filt_state, (float *)(o8 + s->off_filt), (int32_t *)(o8 + s->off_likely),
		                   (int32_t *)(o8 + s->off_spotted), fsm_stage(s, o8));
		if (hipGetLastError() != hipSuccess) r = ed_set_err(ctx, EDISON_E_RUNTIME, "stream: filter launch failed");
	}
	return r;
}

extern "C" void edison_stream_destroy(edison_stream *s)
{
	if (!s) return;
	if (s->own && s->ev_in) (void)drain_q(s, s->own);
	if (s->own) (void)hipStreamSynchronize(s->own);
	if (s->exec) (void)hipGraphExecDestroy(s->exec);
	if (s->graph) (void)hipGraphDestroy(s->graph);
	if (s->exec_h) (void)hipGraphExecDestroy(s->exec_h);
	if (s->graph_h) (void)hipGraphDestroy(s->graph_h);
	if (s->m_audio) (void)hipHostFree(s->m_audio);
	if (s->m_feat) (void)hipHostFree(s->m_feat);
	if (s->m_out) (void)hipHostFree(s->m_out);
	if (s->m_flag) (void)hipHostFree((void *)s->m_flag);
	if (s->h_in) (void)hipHostFree(s->h_in);
	if (s->h_out) (void)hipHostFree(s->h_out);
	if (s->d_audio) (void)hipFree(s->d_audio);
	if (s->d_feat) (void)hipFree(s->d_feat);
	if (s->d_out) (void)hipFree(s->d_out); /* logits, softmax, argmax and the filter outputs live in this one block */
	if (s->d_filt_state) (void)hipFree(s->d_filt_state);
	if (s->d_fsm) (void)hipFree(s->d_fsm);
	if (s->ev_in) (void)hipEventDestroy(s->ev_in);
	if (s->ev_out) (void)hipEventDestroy(s->ev_out);
	if (s->own) (void)hipStreamDestroy(s->own);
	free(s);
}

extern "C" int edison_stream_reset(edison_stream *s)
{
	if (!s) return EDISON_E_ARGUMENT;
	edison_ctx *ctx = s->ctx;
	{ const int rq = drain_q(s, s->own); if (rq != EDISON_OK) return rq; }
	s->a_pos = s->f_pos = 0;
	ED_HIP(ctx, hipMemsetAsync(s->d_audio, 0, sizeof(int16_t) * ((size_t)s->tail + (size_t)s->chunk * s->hop), s->own));
	ED_HIP(ctx, hipMemsetAsync(s->d_feat, 0, (size_t)(30 + s->chunk) * EDISON_NUM_MFCC, s->own));
	if (s->filter) ED_HIP(ctx, hipMemsetAsync(s->d_filt_state, 0, sizeof(float) * EDISON_NET_OUT, s->own));
	if (s->fsm)
	{
		edison_fsm start;
		edison_fsm_init(&start); /* EDI_RESET, as the firmware enters appMicMfccInfereContinuous (app.c:288-300) */
		ED_HIP(ctx, hipMemcpyAsync(s->d_fsm, &start, sizeof(start), hipMemcpyHostToDevice, s->own));
	}
	ED_HIP(ctx, hipStreamSynchronize(s->own));
	if (s->m_audio) memset(s->m_audio, 0, sizeof(int16_t) * ((size_t)s->tail + (size_t)s->chunk * s->hop));
	if (s->m_feat) memset(s->m_feat, 0, (size_t)(30 + s->chunk) * EDISON_NUM_MFCC);
	s->frames_seen = 0;
	return EDISON_OK;
}

/* Moves the stream's state (newest tail samples, newest 30 feature rows: always at the FRONT of their buffers between
 * pushes) to where the next push wants it. Only a caller that alternates host and device pushes pays for this. */
static int stream_state_to(edison_stream *s, int host)
{
	edison_ctx *ctx = s->ctx;
	if (!s->m_audio || s->state_host == host) return EDISON_OK;
	{ const int rq = drain_q(s, s->own); if (rq != EDISON_OK) return rq; }
	{ const int rf = history_to_front(s, s->own); if (rf != EDISON_OK) return rf; }
	const hipMemcpyKind kind = host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice;
	if (s->tail) ED_HIP(ctx, hipMemcpyAsync(host ? (void *)s->m_audio : (void *)s->d_audio, host ? (void *)s->d_audio : (void *)s->m_audio, sizeof(int16_t) * (size_t)s->tail, kind, s->own));
	ED_HIP(ctx, hipMemcpyAsync(host ? (void *)s->m_feat : (void *)s->d_feat, host ? (void *)s->d_feat : (void *)s->m_feat, 30 * EDISON_NUM_MFCC, kind, s->own));
	ED_HIP(ctx, hipStreamSynchronize(s->own));
	s->state_host = host;
	return EDISON_OK;
}

extern "C" void edison_stream_default_opts(edison_stream_opts *o)
{
	if (!o) return;
	memset(o, 0, sizeof(*o));
	o->hop = EDISON_FRAME_LEN;
	o->chunk_frames = 1;
	o->mfcc_variant = EDISON_MFCC_B;
	o->filter = 0;
	o->filter_alpha = 0.9;   /* NET_OUT_MOVING_AVG_ALPHA for NET_TYPE_NNOM, app.c:38 */
	o->true_threshold = 0.5; /* TRUE_THRESHOLD, app.c:34 */
	o->fsm = 0;
	const char *g = getenv("EDISON_STREAM_GRAPH");
	o->launch_mode = (g && atoi(g)) ? EDISON_STREAM_LAUNCH_GRAPH : EDISON_STREAM_LAUNCH_DIRECT;
}

extern "C" int edison_stream_create_ex(edison_ctx *ctx, const edison_stream_opts *o, edison_stream **out)
{
	if (!ctx || !out || !o) return EDISON_E_ARGUMENT;
	*out = NULL;
	const int hop = o->hop, chunk_frames = o->chunk_frames;
	if (!ctx->have_model) return ed_set_err(ctx, EDISON_E_NO_MODEL, "no CNN model loaded (edison_model_load)");
	if (!ctx->fast_model && !(ctx->net.in_h == EDISON_UTT_FRAMES && ctx->net.in_w == EDISON_NUM_MFCC && ctx->net.in_c == 1 &&
	                          ctx->net.out_n == EDISON_NET_OUT && ctx->net.has_softmax))
		return ed_set_err(ctx, EDISON_E_SIZE, "stream: the loaded model is not a 31x13x1 -> 10 softmax classifier");
	if (hop < 2 || hop > EDISON_FRAME_LEN || (hop & 1) || chunk_frames < 1 || chunk_frames > (1 << 22))
		return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: hop must be even and 2..1024, chunk 1..4M frames");
	/* positions and sample counts of a push are ints in places (a_pos, kernel arguments): a push of 2^30 samples or more (2 GiB of
	 * int16) is refused here instead of wrapping there */
	if ((int64_t)chunk_frames * hop >= ((int64_t)1 << 30))
		return ed_set_err(ctx, EDISON_E_SIZE, "stream: chunk_frames x hop must stay below 2^30 samples per push");
	if (o->mfcc_variant != EDISON_MFCC_B && o->mfcc_variant != EDISON_MFCC_C)
		return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: MFCC variant must be EDISON_MFCC_B or EDISON_MFCC_C");
	if (o->filter && !(o->filter_alpha >= 0.0 && o->filter_alpha <= 1.0))
		return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: filter_alpha must be within [0, 1]");
	if (o->fsm && !o->filter) return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: the state machine (fsm) works on the filtered outputs: filter = 1 too");
	edison_stream *s = (edison_stream *)calloc(1, sizeof(edison_stream));
	if (!s) return ed_set_err(ctx, EDISON_E_NO_MEMORY, "host allocation failed");
	s->ctx = ctx; s->hop = hop; s->chunk = chunk_frames; s->tail = EDISON_FRAME_LEN - hop;
	s->last_n = chunk_frames; /* before the first push the (zeroed) blocks read as a whole chunk */
	s->variant = o->mfcc_variant;
	s->model_epoch = ctx->model_epoch;
	s->tables_epoch = ctx->tables_epoch;
	s->filter = o->filter ? 1 : 0;
	s->fsm = o->fsm ? 1 : 0;
	s->dt_us = (uint32_t)(((uint64_t)hop * 1000000u) / EDISON_FS); /* one inference per hop (app.c:635-663: one per audio event) */
	edison_fsm_roles(&s->roles.wake_idx, &s->roles.loc_mask, &s->roles.val_mask);
	s->use_graph = o->launch_mode == EDISON_STREAM_LAUNCH_GRAPH;
	s->alpha = o->filter_alpha;
	s->one_minus_alpha = 1.0 - o->filter_alpha; /* the firmware's (1.0-NET_OUT_MOVING_AVG_ALPHA), folded in double */
	s->threshold = o->true_threshold;
	hipError_t e = hipSetDevice(ctx->device);
	if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming);
	if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming);
	/* eight pushes of room for the sliding device pushes, as long as that stays below 64 MB of samples */
	s->slots = (!s->use_graph && (size_t)s->chunk * s->hop * sizeof(int16_t) * 8 <= ((size_t)64 << 20)) ? 8 : 1;
	if (e == hipSuccess) e = hipMalloc((void **)&s->d_audio, sizeof(int16_t) * ((size_t)s->tail + (size_t)s->slots * s->chunk * s->hop) + 16);
	if (e == hipSuccess) e = hipMalloc((void **)&s->d_feat, (size_t)(30 + (size_t)s->slots * s->chunk) * EDISON_NUM_MFCC + 16);
	{
		/* every output of a push in one device block, so that the host path fetches them with a single copy */
		const size_t c = (size_t)s->chunk;
		size_t off = c * EDISON_NET_OUT;                                /* logits at 0 */
		s->off_soft = off; off += c * EDISON_NET_OUT;
		off = (off + 15) & ~(size_t)15; s->off_argmax = off; off += c * sizeof(int32_t);
		off = (off + 15) & ~(size_t)15; s->off_filt = off; off += s->filter ? c * EDISON_NET_OUT * sizeof(float) : 0;
		s->off_likely = off; off += s->filter ? c * sizeof(int32_t) : 0;
		s->off_spotted = off; off += s->filter ? c * sizeof(int32_t) : 0;
		s->off_fsm_states = off; off += s->fsm ? c * sizeof(int32_t) : 0;
		off = (off + 15) & ~(size_t)15; s->off_fsm = off; off += s->fsm ? sizeof(edison_fsm) : 0;
		s->h_out_bytes = off;
		if (e == hipSuccess) e = hipMalloc((void **)&s->d_out, s->h_out_bytes + 16);
		if (e == hipSuccess)
		{
			s->d_logits = (int8_t *)s->d_out;
			s->d_soft = (int8_t *)(s->d_out + s->off_soft);
			s->d_argmax = (int32_t *)(s->d_out + s->off_argmax);
			s->d_filt = (float *)(s->d_out + s->off_filt);
			s->d_likely = (int32_t *)(s->d_out + s->off_likely);
			s->d_spotted = (int32_t *)(s->d_out + s->off_spotted);
		}
	}
	if (s->filter)
	{
		if (e == hipSuccess) e = hipMalloc((void **)&s->d_filt_state, sizeof(float) * EDISON_NET_OUT);
	}
	if (s->fsm && e == hipSuccess) e = hipMalloc((void **)&s->d_fsm, sizeof(edison_fsm));
	if (e != hipSuccess)
	{
		edison_stream_destroy(s);
		return ed_set_err(ctx, e == hipErrorOutOfMemory ? EDISON_E_NO_MEMORY : EDISON_E_RUNTIME, "stream: device allocation failed");
	}
	int r = edison_stream_reset(s);
	/* one eager push on silence: sizes the persistent grids / sets kernel attributes outside of graph capture */
	if (r == EDISON_OK) r = enqueue_push(s);
	if (r == EDISON_OK && hipStreamSynchronize(s->own) != hipSuccess) r = ed_set_err(ctx, EDISON_E_RUNTIME, "stream: warm-up failed");
	if (r == EDISON_OK) r = edison_stream_reset(s);
	if (r == EDISON_OK)
	{
		/* capture one push into a graph; every later push is a single hipGraphLaunch */
		e = hipStreamBeginCapture(s->own, hipStreamCaptureModeThreadLocal);
		if (e == hipSuccess)
		{
			r = enqueue_push(s);
			hipError_t e2 = hipStreamEndCapture(s->own, &s->graph);
			if (r == EDISON_OK && e2 != hipSuccess) r = ed_set_err(ctx, EDISON_E_RUNTIME, "stream: graph capture failed");
			if (r == EDISON_OK && hipGraphInstantiate(&s->exec, s->graph, NULL, NULL, 0) != hipSuccess)
				r = ed_set_err(ctx, EDISON_E_RUNTIME, "stream: graph instantiation failed");
		}
		else
			r = ed_set_err(ctx, EDISON_E_RUNTIME, "stream: cannot begin graph capture on this stream");
	}
	const size_t in_bytes = sizeof(int16_t) * (size_t)s->chunk * s->hop;
	if (r == EDISON_OK && in_bytes <= ED_STREAM_STAGED_MAX_BYTES)
	{
		/* the staged variant: upload + push + downloads in one graph, against pinned host buffers */
		e = hipHostMalloc((void **)&s->h_in, in_bytes, hipHostMallocDefault);
		if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_out, s->h_out_bytes, hipHostMallocDefault);
		if (e == hipSuccess) e = hipStreamBeginCapture(s->own, hipStreamCaptureModeThreadLocal);
		if (e == hipSuccess)
		{
			hipError_t c1 = hipMemcpyAsync(s->d_audio + s->tail, s->h_in, in_bytes, hipMemcpyHostToDevice, s->own);
			r = enqueue_push(s);
			if (c1 == hipSuccess) c1 = hipMemcpyAsync(s->h_out, s->d_out, s->h_out_bytes, hipMemcpyDeviceToHost, s->own);
			hipError_t e2 = hipStreamEndCapture(s->own, &s->graph_h);
			if (r == EDISON_OK && (c1 != hipSuccess || e2 != hipSuccess)) r = ed_set_err(ctx, EDISON_E_RUNTIME, "stream: staged graph capture failed");
			if (r == EDISON_OK && hipGraphInstantiate(&s->exec_h, s->graph_h, NULL, NULL, 0) != hipSuccess)
				r = ed_set_err(ctx, EDISON_E_RUNTIME, "stream: staged graph instantiation failed");
		}
		else
			r = ed_set_err(ctx, EDISON_E_RUNTIME, "stream: pinned staging buffers / capture unavailable");
	}
	if (r == EDISON_OK && in_bytes <= ED_STREAM_MAPPED_MAX_BYTES)
	{
		const size_t audio_bytes = sizeof(int16_t) * ((size_t)s->tail + (size_t)s->chunk * s->hop) + 16, feat_bytes = (size_t)(30 + s->chunk) * EDISON_NUM_MFCC + 16;
		e = hipHostMalloc((void **)&s->m_audio, audio_bytes, hipHostMallocMapped);
		if (e == hipSuccess) e = hipHostMalloc((void **)&s->m_feat, feat_bytes, hipHostMallocMapped);
		if (e == hipSuccess) e = hipHostMalloc((void **)&s->m_out, s->h_out_bytes + 16, hipHostMallocMapped);
		void *da = NULL, *df = NULL, *dout = NULL, *dflag = NULL;
		if (e == hipSuccess) e = hipHostMalloc((void **)&s->m_flag, 64, hipHostMallocMapped);
		if (e == hipSuccess) e = hipHostGetDevicePointer(&dflag, (void *)s->m_flag, 0);
		if (e == hipSuccess) { *s->m_flag = 0; s->md_flag = (unsigned *)dflag; s->flag_seq = 0; }
		if (e == hipSuccess) e = hipHostGetDevicePointer(&da, s->m_audio, 0);
		if (e == hipSuccess) e = hipHostGetDevicePointer(&df, s->m_feat, 0);
		if (e == hipSuccess) e = hipHostGetDevicePointer(&dout, s->m_out, 0);
		if (e == hipSuccess)
		{
			memset(s->m_audio, 0, audio_bytes); memset(s->m_feat, 0, feat_bytes); memset(s->m_out, 0, s->h_out_bytes + 16);
			s->md_audio = (const int16_t *)da; s->md_feat = (int8_t *)df; s->md_out = (unsigned char *)dout;
			s->mapped = 1;
		}
		else
		{
			/* no host-mapped memory on this system: the staged and the plain paths serve the host pushes */
			(void)hipGetLastError();
			if (s->m_audio) { (void)hipHostFree(s->m_audio); s->m_audio = NULL; }
			if (s->m_feat) { (void)hipHostFree(s->m_feat); s->m_feat = NULL; }
			if (s->m_out) { (void)hipHostFree(s->m_out); s->m_out = NULL; }
			if (s->m_flag) { (void)hipHostFree((void *)s->m_flag); s->m_flag = NULL; }
			s->mapped = 0;
		}
	}
	if (r != EDISON_OK) { edison_stream_destroy(s); return r; }
	*out = s;
	return EDISON_OK;
}

extern "C" int edison_stream_create(edison_ctx *ctx, int hop, int chunk_frames, edison_stream **out)
{
	edison_stream_opts o;
	edison_stream_default_opts(&o);
	o.hop = hop;
	o.chunk_frames = chunk_frames;
	return edison_stream_create_ex(ctx, &o, out);
}

/* samples: chunk*hop NEW int16 samples in device memory. Outputs (device, each may be NULL): softmax / logits
 * [chunk][10], argmax [chunk]; entry i belongs to the window that ends with the i-th new frame. Asynchronous. */
extern "C" int edison_stream_push_dev(edison_stream *s, const int16_t *samples, int8_t *logits, int8_t *softmax,
                                      int32_t *argmax)
{
	return edison_stream_push_n_dev(s, samples, s ? s->chunk : 0, logits, softmax, argmax);
}

/* the same for n_frames <= chunk new frames (n_frames * hop samples; outputs [n_frames][..]): the ragged last push of a recording whose
 * length the chunk does not divide. A short push needs the directly launched kernels (a captured graph has the chunk baked in). */
extern "C" int edison_stream_push_n_dev(edison_stream *s, const int16_t *samples, int n_frames, int8_t *logits, int8_t *softmax,
                                        int32_t *argmax)
{
	if (!s || !samples) return EDISON_E_ARGUMENT;
	edison_ctx *ctx = s->ctx;
	if (n_frames < 1 || n_frames > s->chunk) return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: n_frames must be 1 .. chunk_frames");
	if (s->model_epoch != ctx->model_epoch) return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: the model was reloaded after this stream was created; create a new stream");
	if (s->tables_epoch != ctx->tables_epoch) return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: edison_mfcc_configure was called after this stream was created; create a new stream");
	const size_t nnew = (size_t)n_frames * s->hop;
	{ const int rs = stream_state_to(s, 0); if (rs != EDISON_OK) return rs; }
	s->last_push_mapped = 0;
	if (!s->use_graph)
	{
		/* direct launches, on the caller's stream: one copy + MFCC + CNN (+ filter) + shift, the CNN writing the caller's
		 * buffers; the private stream is idle here (everything it ever does ends in a synchronisation) */
		hipStream_t q = ctx->stream;
		{ const int rq = drain_q(s, q); if (rq != EDISON_OK) return rq; }
		ED_HIP(ctx, hipMemcpyAsync(s->d_audio + s->a_pos + s->tail, samples, nnew * sizeof(int16_t), hipMemcpyDeviceToDevice, q));
		/* the output filter reads the softmax from the stream's own block */
		int8_t *so = s->filter ? s->d_soft : softmax;
		{ const int rd = enqueue_push_on(s, q, logits, so, argmax, 1, n_frames); if (rd != EDISON_OK) return rd; }
		if (s->filter && softmax) ED_HIP(ctx, hipMemcpyAsync(softmax, s->d_soft, (size_t)n_frames * EDISON_NET_OUT, hipMemcpyDeviceToDevice, q));
		s->q_last = q;
		s->q_pending = 1;
		s->last_push_staged = 0;
		s->last_n = n_frames;
		s->frames_seen += n_frames;
		return EDISON_OK;
	}
	/* graph replay: on the private stream (a graph is captured on one stream), ordered against the caller's with events.
	 * The caller produced `samples` on the context's stream: the private stream waits for that point ... */
	ED_HIP(ctx, hipEventRecord(s->ev_in, ctx->stream));
	ED_HIP(ctx, hipStreamWaitEvent(s->own, s->ev_in, 0));
	ED_HIP(ctx, hipMemcpyAsync(s->d_audio + s->tail, samples, nnew * sizeof(int16_t), hipMemcpyDeviceToDevice, s->own));
	if (n_frames == s->chunk)
	{
		/* launch_mode = EDISON_STREAM_LAUNCH_GRAPH: the captured nodes of a whole chunk (on this platform the replay measures SLOWER
		 * than its three or four plain launches -- 60.0 M frames/s against 67.2 M over the 1 h stream, 39 against 32 us per
		 * one-frame push -- which is why direct launches are the default) */
		ED_HIP(ctx, hipGraphLaunch(s->exec, s->own));
	}
	else
	{
		/* the ragged last push of a recording (n_frames < chunk): the captured graph has the chunk baked in, so these few frames take
		 * the same kernels launched directly, on the same private stream and on the same state (history at the front, as the graph
		 * leaves it) -- a graph-mode stream consumes every frame of a recording too (round 5; it used to end short of the tail) */
		const int rd = enqueue_push_on(s, s->own, s->d_logits, s->d_soft, s->d_argmax, 0, n_frames);
		if (rd != EDISON_OK) return rd;
	}
	s->last_push_staged = 0;
	if (logits) ED_HIP(ctx, hipMemcpyAsync(logits, s->d_logits, (size_t)n_frames * EDISON_NET_OUT, hipMemcpyDeviceToDevice, s->own));
	if (softmax) ED_HIP(ctx, hipMemcpyAsync(softmax, s->d_soft, (size_t)n_frames * EDISON_NET_OUT, hipMemcpyDeviceToDevice, s->own));
	if (argmax) ED_HIP(ctx, hipMemcpyAsync(argmax, s->d_argmax, (size_t)n_frames * sizeof(int32_t), hipMemcpyDeviceToDevice, s->own));
	/* ... and the context's stream continues only after the outputs are written */
	ED_HIP(ctx, hipEventRecord(s->ev_out, s->own));
	ED_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ev_out, 0));
	s->last_n = n_frames;
	s->frames_seen += n_frames;
	return EDISON_OK;
}

/* the same with host pointers; synchronous */
extern "C" int edison_stream_push(edison_stream *s, const int16_t *samples, int8_t *logits, int8_t *softmax, int32_t *argmax)
{
	if (!s || !samples) return EDISON_E_ARGUMENT;
	edison_ctx *ctx = s->ctx;
	if (s->model_epoch != ctx->model_epoch) return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: the model was reloaded after this stream was created; create a new stream");
	if (s->tables_epoch != ctx->tables_epoch) return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: edison_mfcc_configure was called after this stream was created; create a new stream");
	const size_t nnew = (size_t)s->chunk * s->hop;
	static const int no_mapped = getenv("EDISON_STREAM_NO_MAPPED") ? atoi(getenv("EDISON_STREAM_NO_MAPPED")) : 0; /* A/B knob: the staged graph */
	if (s->mapped && !no_mapped && !s->use_graph)
	{
		const size_t c = (size_t)s->chunk;
		{ const int rs = stream_state_to(s, 1); if (rs != EDISON_OK) return rs; }
		memcpy(s->m_audio + s->tail, samples, nnew * sizeof(int16_t));
		const unsigned seq = ++s->flag_seq;
		int kernel_writes_flag = 0;
		{ const int rd = enqueue_mapped_push(s, seq, &kernel_writes_flag); if (rd != EDISON_OK) return rd; }
		{
			/* wait for the answer: spin on the sequence number the command processor writes behind the last kernel; if
			 * that stream operation is unavailable, or nothing arrives within 20 ms (by the clock, looked at every 1024
			 * spins), synchronize the ordinary way (which also surfaces a device error). The flag is read with acquire
			 * semantics: the outputs copied below must not be read before it. */
			int waited = 0;
			if (kernel_writes_flag || hipStreamWriteValue32(s->own, s->md_flag, seq, 0) == hipSuccess)
			{
				struct timespec t0, t1;
				clock_gettime(CLOCK_MONOTONIC, &t0);
				for (unsigned spins = 1;; spins++)
				{
					if (__atomic_load_n((const unsigned *)s->m_flag, __ATOMIC_ACQUIRE) == seq) { waited = 1; break; }
					_mm_pause();
					if ((spins & 1023u) == 0)
					{
						clock_gettime(CLOCK_MONOTONIC, &t1);
						if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 20000000L) break;
					}
				}
			}
			else
				(void)hipGetLastError();
			if (!waited) ED_HIP(ctx, hipStreamSynchronize(s->own));
		}
		if (logits) memcpy(logits, s->m_out, c * EDISON_NET_OUT);
		if (softmax) memcpy(softmax, s->m_out + s->off_soft, c * EDISON_NET_OUT);
		if (argmax) memcpy(argmax, s->m_out + s->off_argmax, c * sizeof(int32_t));
		/* what the shift kernel does on the device path: the newest tail samples and 30 rows to the front */
		if (s->tail) memmove(s->m_audio, s->m_audio + nnew, sizeof(int16_t) * (size_t)s->tail);
		memmove(s->m_feat, s->m_feat + c * EDISON_NUM_MFCC, 30 * EDISON_NUM_MFCC);
		s->last_push_staged = 0;
		s->last_push_mapped = 1;
		s->last_n = s->chunk;
	s->frames_seen += s->chunk;
		return EDISON_OK;
	}
	{ const int rs = stream_state_to(s, 0); if (rs != EDISON_OK) return rs; }
	{ const int rq = drain_q(s, s->own); if (rq != EDISON_OK) return rq; } /* device pushes may have left work on the caller's stream */
	{ const int rf = history_to_front(s, s->own); if (rf != EDISON_OK) return rf; } /* ... and the history somewhere behind the front */
	s->last_push_mapped = 0;
	if (s->exec_h)
	{
		const size_t c = (size_t)s->chunk;
		memcpy(s->h_in, samples, nnew * sizeof(int16_t));
		ED_HIP(ctx, hipGraphLaunch(s->exec_h, s->own));
		ED_HIP(ctx, hipStreamSynchronize(s->own));
		if (logits) memcpy(logits, s->h_out, c * EDISON_NET_OUT);
		if (softmax) memcpy(softmax, s->h_out + s->off_soft, c * EDISON_NET_OUT);
		if (argmax) memcpy(argmax, s->h_out + s->off_argmax, c * sizeof(int32_t));
		s->last_push_staged = 1;
		s->last_n = s->chunk;
	s->frames_seen += s->chunk;
		return EDISON_OK;
	}
	s->last_push_staged = 0;
	ED_HIP(ctx, hipMemcpyAsync(s->d_audio + s->tail, samples, nnew * sizeof(int16_t), hipMemcpyHostToDevice, s->own));
	{ const int rd = enqueue_push(s); if (rd != EDISON_OK) return rd; } /* plain launches (see edison_stream_push_dev) */
	if (logits) ED_HIP(ctx, hipMemcpyAsync(logits, s->d_logits, (size_t)s->chunk * EDISON_NET_OUT, hipMemcpyDeviceToHost, s->own));
	if (softmax) ED_HIP(ctx, hipMemcpyAsync(softmax, s->d_soft, (size_t)s->chunk * EDISON_NET_OUT, hipMemcpyDeviceToHost, s->own));
	if (argmax) ED_HIP(ctx, hipMemcpyAsync(argmax, s->d_argmax, (size_t)s->chunk * sizeof(int32_t), hipMemcpyDeviceToHost, s->own));
	ED_HIP(ctx, hipStreamSynchronize(s->own));
	s->last_n = s->chunk;
	s->frames_seen += s->chunk;
	return EDISON_OK;
}

/* Filtered outputs of the LAST push (filter enabled in the options): filt [n][10] fp32, likely / spotted [n], n = the frames of that
 * push (chunk_frames, or fewer after edison_stream_push_n_dev).
 * host = 1: host pointers, synchronous; host = 0: device pointers, ordered on the context's stream like a push. */
static int stream_filter_out(edison_stream *s, float *filt, int32_t *likely, int32_t *spotted, int host)
{
	if (!s) return EDISON_E_ARGUMENT;
	edison_ctx *ctx = s->ctx;
	if (!s->filter) return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: created without the output filter");
	/* entries of the last push only: after a ragged push (edison_stream_push_n_dev, n < chunk) the rows n.. of the stream's blocks
	 * are an earlier push's, and a caller that sized its buffers [n][..] must not be written past them */
	const size_t c = (size_t)s->last_n;
	if (s->last_push_mapped)
	{
		/* the mapped push wrote them into pinned host memory */
		if (host)
		{
			if (filt) memcpy(filt, s->m_out + s->off_filt, c * EDISON_NET_OUT * sizeof(float));
			if (likely) memcpy(likely, s->m_out + s->off_likely, c * sizeof(int32_t));
			if (spotted) memcpy(spotted, s->m_out + s->off_spotted, c * sizeof(int32_t));
			return EDISON_OK;
		}
		if (filt) ED_HIP(ctx, hipMemcpyAsync(filt, s->m_out + s->off_filt, c * EDISON_NET_OUT * sizeof(float), hipMemcpyHostToDevice, s->own));
		if (likely) ED_HIP(ctx, hipMemcpyAsync(likely, s->m_out + s->off_likely, c * sizeof(int32_t), hipMemcpyHostToDevice, s->own));
		if (spotted) ED_HIP(ctx, hipMemcpyAsync(spotted, s->m_out + s->off_spotted, c * sizeof(int32_t), hipMemcpyHostToDevice, s->own));
		ED_HIP(ctx, hipEventRecord(s->ev_out, s->own));
		ED_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ev_out, 0));
		return EDISON_OK;
	}
	{ const int rq = drain_q(s, s->own); if (rq != EDISON_OK) return rq; }
	if (host && s->last_push_staged)
	{
		/* the staged push already brought them to the host */
		if (filt) memcpy(filt, s->h_out + s->off_filt, c * EDISON_NET_OUT * sizeof(float));
		if (likely) memcpy(likely, s->h_out + s->off_likely, c * sizeof(int32_t));
		if (spotted) memcpy(spotted, s->h_out + s->off_spotted, c * sizeof(int32_t));
		return EDISON_OK;
	}
	const hipMemcpyKind kind = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
	if (filt) ED_HIP(ctx, hipMemcpyAsync(filt, s->d_filt, sizeof(float) * c * EDISON_NET_OUT, kind, s->own));
	if (likely) ED_HIP(ctx, hipMemcpyAsync(likely, s->d_likely, sizeof(int32_t) * c, kind, s->own));
	if (spotted) ED_HIP(ctx, hipMemcpyAsync(spotted, s->d_spotted, sizeof(int32_t) * c, kind, s->own));
	if (host) ED_HIP(ctx, hipStreamSynchronize(s->own));
	else
	{
		ED_HIP(ctx, hipEventRecord(s->ev_out, s->own));
		ED_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ev_out, 0));
	}
	return EDISON_OK;
}

/* the state machine after the LAST push and the state after each of its inferences */
static int stream_fsm_out(edison_stream *s, edison_fsm *fsm, int32_t *states, int host)
{
	if (!s) return EDISON_E_ARGUMENT;
	edison_ctx *ctx = s->ctx;
	if (!s->fsm) return ed_set_err(ctx, EDISON_E_ARGUMENT, "stream: created without the state machine (opts.fsm)");
	const size_t c = (size_t)s->last_n; /* as in stream_filter_out: the last push's entries only */
	if (s->last_push_mapped)
	{
		if (host)
		{
			if (fsm) memcpy(fsm, s->m_out + s->off_fsm, sizeof(*fsm));
			if (states) memcpy(states, s->m_out + s->off_fsm_states, c * sizeof(int32_t));
			return EDISON_OK;
		}
		if (states) ED_HIP(ctx, hipMemcpyAsync(states, s->m_out + s->off_fsm_states, c * sizeof(int32_t), hipMemcpyHostToDevice, s->own));
		ED_HIP(ctx, hipEventRecord(s->ev_out, s->own));
		ED_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ev_out, 0));
		return EDISON_OK;
	}
	{ const int rq = drain_q(s, s->own); if (rq != EDISON_OK) return rq; }
	if (host && s->last_push_staged)
	{
		if (fsm) memcpy(fsm, s->h_out + s->off_fsm, sizeof(*fsm));
		if (states) memcpy(states, s->h_out + s->off_fsm_states, c * sizeof(int32_t));
		return EDISON_OK;
	}
	const hipMemcpyKind kind = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
	if (states) ED_HIP(ctx, hipMemcpyAsync(states, s->d_out + s->off_fsm_states, c * sizeof(int32_t), kind, s->own));
	if (host && fsm) ED_HIP(ctx, hipMemcpyAsync(fsm, s->d_fsm, sizeof(*fsm), hipMemcpyDeviceToHost, s->own));
	if (host) ED_HIP(ctx, hipStreamSynchronize(s->own));
	else
	{
		ED_HIP(ctx, hipEventRecord(s->ev_out, s->own));
		ED_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ev_out, 0));
	}
	return EDISON_OK;
}

extern "C" int edison_stream_fsm(edison_stream *s, edison_fsm *fsm, int32_t *states) { return stream_fsm_out(s, fsm, states, 1); }
extern "C" int edison_stream_fsm_dev(edison_stream *s, int32_t *states) { return stream_fsm_out(s, NULL, states, 0); }

/* The firmware's post-processing chain (app.c:332-371) on n network outputs in time order, without a stream: the filter kernel
 * with the state machine behind it, one launch. Host pointers; synchronous. */
extern "C" int edison_postproc(edison_ctx *ctx, const int8_t *softmax, int64_t n, double alpha, double true_threshold, uint32_t dt_us,
                               float *filt_state, edison_fsm *fsm, float *filt, int32_t *likely, int32_t *spotted, int32_t *states)
{
	/* n < 2^30: the kernel indexes with ints and strides of 256; the chain is sequential in time by definition (ten lanes filter, one
	 * lane walks the machine), so the cost grows linearly with n in ONE workgroup -- a post-processing stage, not a batch kernel */
	if (!ctx || !softmax || !filt_state || n < 0 || n >= ((int64_t)1 << 30) || !(alpha >= 0.0 && alpha <= 1.0)) return EDISON_E_ARGUMENT;
	if (!(true_threshold == true_threshold)) return ed_set_err(ctx, EDISON_E_ARGUMENT, "edison_postproc: true_threshold is not a number");
	/* what edison_fsm_step answers for the same machine: an unknown state is an argument error, not n states of -1 */
	if (fsm && (fsm->state < EDISON_FSM_RESET || fsm->state > EDISON_FSM_SET)) return ed_set_err(ctx, EDISON_E_ARGUMENT, "edison_postproc: fsm->state is not a state of the machine");
	if (n == 0) return EDISON_OK;
	const size_t c = (size_t)n;
	size_t off = c * EDISON_NET_OUT;                                   /* softmax at 0 */
	off = (off + 15) & ~(size_t)15; const size_t o_state = off; off += EDISON_NET_OUT * sizeof(float);
	off = (off + 15) & ~(size_t)15; const size_t o_filt = off; off += c * EDISON_NET_OUT * sizeof(float);
	const size_t o_likely = off; off += c * sizeof(int32_t);
	const size_t o_spotted = off; off += c * sizeof(int32_t);
	const size_t o_states = off; off += c * sizeof(int32_t);
	off = (off + 15) & ~(size_t)15; const size_t o_fsm = off; off += sizeof(edison_fsm);
	unsigned char *d = NULL;
	ED_HIP(ctx, hipSetDevice(ctx->device));
	ED_HIP(ctx, hipMalloc((void **)&d, off));
	hipStream_t q = ctx->stream;
	hipError_t e = hipMemcpyAsync(d, softmax, c * EDISON_NET_OUT, hipMemcpyHostToDevice, q);
	if (e == hipSuccess) e = hipMemcpyAsync(d + o_state, filt_state, EDISON_NET_OUT * sizeof(float), hipMemcpyHostToDevice, q);
	if (e == hipSuccess && fsm) e = hipMemcpyAsync(d + o_fsm, fsm, sizeof(*fsm), hipMemcpyHostToDevice, q);
	if (e == hipSuccess)
	{
		ed_fsm_stage_t fs;
		memset(&fs, 0, sizeof(fs));
		if (fsm)
		{
			fs.fsm = (edison_fsm *)(d + o_fsm); fs.states = (int32_t *)(d + o_states); fs.dt_us = dt_us;
			edison_fsm_roles(&fs.roles.wake_idx, &fs.roles.loc_mask, &fs.roles.val_mask);
		}
		hipLaunchKernelGGL(ed_stream_filter_kernel, dim3(1), dim3(256), 0, q, (const int8_t *)d, (int)n, alpha, 1.0 - alpha, true_threshold,
		                   (float *)(d + o_state), (float *)(d + o_filt), (int32_t *)(d + o_likely), (int32_t *)(d + o_spotted), fs);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(filt_state, d + o_state, EDISON_NET_OUT * sizeof(float), hipMemcpyDeviceToHost, q);
	if (e == hipSuccess && fsm) e = hipMemcpyAsync(fsm, d + o_fsm, sizeof(*fsm), hipMemcpyDeviceToHost, q);
	if (e == hipSuccess && filt) e = hipMemcpyAsync(filt, d + o_filt, c * EDISON_NET_OUT * sizeof(float), hipMemcpyDeviceToHost, q);
	if (e == hipSuccess && likely) e = hipMemcpyAsync(likely, d + o_likely, c * sizeof(int32_t), hipMemcpyDeviceToHost, q);
	if (e == hipSuccess && spotted) e = hipMemcpyAsync(spotted, d + o_spotted, c * sizeof(int32_t), hipMemcpyDeviceToHost, q);
	if (e == hipSuccess && states && fsm) e = hipMemcpyAsync(states, d + o_states, c * sizeof(int32_t), hipMemcpyDeviceToHost, q);
	if (e == hipSuccess) e = hipStreamSynchronize(q);
	(void)hipFree(d);
	ED_HIP(ctx, e);
	return EDISON_OK;
}

extern "C" int edison_stream_filtered(edison_stream *s, float *filt, int32_t *likely, int32_t *spotted)
{
	return stream_filter_out(s, filt, likely, spotted, 1);
}

extern "C" int edison_stream_filtered_dev(edison_stream *s, float *filt, int32_t *likely, int32_t *spotted)
{
	return stream_filter_out(s, filt, likely, spotted, 0);
}

extern "C" int64_t edison_stream_frames_seen(const edison_stream *s) { return s ? s->frames_seen : -1; }
