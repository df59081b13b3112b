/*
 * include/edison_hip.h -- C-ABI of libedison_hip.so: edison's keyword-spotting hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary. Plain C, plain pointers and sizes, int return codes; no torch types.
 * Every entry point names the reference interface it replaces (paths relative to the reference repo):
 *
 *   per-frame MFCC   audio/edison/mfcc/mfcc_utils.py:134 `mfcc` (variant A), :255 `mfcc_mcu` (variant B),
 *                    :75 `batch_mfcc`; C call surface firmware/src/audioprocessing.h:21-22
 *                    (`audioInit`, `audioCalcMFCCs`) and firmware/src/audio/mfcc.h:64-67 (`mfcc_compute`)
 *   net-input glue   firmware/src/app.c:675-695 `mfccToNetInput`, :706-719 `mfccToNetInputPush`
 *   int8 CNN         firmware/src/ai/ai.h:74-80 (`aiInitialize`, `aiGetInputShape`, `aiRunInference`,
 *                    `aiGetKeywordFromIndex`, `aiGetKeywordCount`), firmware/src/ai/ai_nnom.c:64-132
 *                    (`aiNnomInit`, `aiNnomRunInference`, `aiNnomPredict`, `aiNnomGet{Input,Output}Buffer`),
 *                    model = firmware/src/ai/nnom/kws_nnom/weights.h
 *
 * The reference processes ONE frame / ONE utterance per call on a Cortex-M4 (or in a Python loop). The
 * batched `edison_*_batch*` calls are the MI355X-native form of the same computation; the legacy names at
 * the bottom are batch=1 wrappers with the reference's own signatures and ownership rules.
 *
 * Conventions
 *   - return 0 (EDISON_OK == NNoM's NN_SUCCESS) or a negative code; edison_last_error() gives text.
 *   - `_dev` calls take DEVICE pointers (hipMalloc / torch CUDA tensors) and enqueue on the context's
 *     stream without synchronising; the others take HOST pointers and are synchronous.
 *   - one context per process per GPU; calls on one context are serialised by the caller.
 *   - there is NO CPU fallback: without a gfx950 device edison_init fails.
 */
#ifndef EDISON_HIP_H
#define EDISON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants of the hot path (audio/config.py:11-32, firmware/src/ai/ai.h:55-60) ------------------ */
#define EDISON_FS 16000
#define EDISON_FRAME_LEN 1024          /* frame_length = fft_len = frame_step                            */
#define EDISON_NUM_MEL 32              /* num_mel_bins                                                   */
#define EDISON_NUM_MFCC 13             /* num_mfcc (first_mfcc = 0)                                      */
#define EDISON_UTT_FRAMES 31           /* n_frames of a 2 s / 32000-sample utterance                     */
#define EDISON_NET_IN (EDISON_UTT_FRAMES * EDISON_NUM_MFCC) /* 403 int8, HWC [31][13][1]                 */
#define EDISON_NET_OUT 10

/* ---- status codes: 0 and the negatives mirror nnom_status_t (firmware/src/ai/nnom/inc/nnom.h:34-45) -- */
#define EDISON_OK 0                     /* NN_SUCCESS                                                    */
#define EDISON_E_ARGUMENT (-1)          /* NN_ARGUMENT_ERROR                                             */
#define EDISON_E_LENGTH (-2)            /* NN_LENGTH_ERROR                                               */
#define EDISON_E_SIZE (-3)              /* NN_SIZE_MISMATCH: model blob does not have the kws_conv shape  */
#define EDISON_E_NO_MEMORY (-7)         /* NN_NO_MEMORY (host or HBM allocation failed)                  */
#define EDISON_E_MORE_TODO (-8)         /* NN_MORE_TODO                                                  */
/* codes below have no NNoM counterpart */
#define EDISON_E_RUNTIME (-16)          /* HIP error, text in edison_last_error()                        */
#define EDISON_E_NO_IMPL (-17)          /* valid for the reference, not built on this path yet           */
#define EDISON_E_NO_DEVICE (-18)        /* no gfx950 GPU visible: the product has no CPU path            */
#define EDISON_E_NO_MODEL (-19)         /* CNN call before edison_model_load                             */

/* ---- MFCC variants ----------------------------------------------------------- */
#define EDISON_MFCC_A 0 /* mfcc_utils.mfcc    : fft[:512] -> |.| -> mel(512x32) -> ln(x+1e-6) -> dct2/sqrt(64) */
#define EDISON_MFCC_B 1 /* mfcc_utils.mfcc_mcu: fft/1024 -> |.|/sqrt2 -> mel(513x32) -> [ln] -> dct2/64        */
#define EDISON_MFCC_C 2 /* audioCalcMFCCs, the firmware's Q15 pipeline (firmware/src/audioprocessing.c:116-215):
                         * arm_cfft_q15(1024) -> arm_sqrt_q31 magnitude -> compact int16 mel / 128 -> RFFT-based dct2_q15.
                         * Integer-exact. Through the fp32 entry points below the outputs are the int16 values as
                         * floats (Cube branch of mfccToNetInput, app.c:680-683) and feat is the NNoM branch
                         * (app.c:686-694); feat_scale must be 1, no log. Native int16 interface: edison_mfcc_q15_*.  */
#define EDISON_MFCC_TF 3 /* mfcc_utils.mfcc_tf (mfcc_utils.py:201-253), the TensorFlow curve of `main.py mfcc host`:
                          * periodic Hann window (tf.signal.stft's default) -> rfft -> |.| -> mel(513x32) -> ln(x+1e-6) ->
                          * dct2/sqrt(64). Batch and stage entry points only (edison_mfcc_batch / _rows / _stages and
                          * their _dev forms); TensorFlow is not in this image, so this variant's parity is UNPINNED:
                          * it is checked against a float64 restatement of tf.signal's published definitions only.     */
#define EDISON_MFCC_USE_LOG 0x100 /* OR into variant: mfcc_mcu(..., use_log=True)                       */

typedef struct edison_ctx edison_ctx;

/* ---- lifecycle ---------------------------------------------------------------------------------------- */
int edison_init(int device, edison_ctx **out);
void edison_shutdown(edison_ctx *ctx);
const char *edison_last_error(const edison_ctx *ctx); /* ctx may be NULL: last init error                */
/* Enqueue on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). The handle is used as
 * given: NULL is HIP's default (null) stream, NOT "no stream". edison_reset_stream returns to the context's own
 * private non-blocking stream (the state after edison_init). */
int edison_set_stream(edison_ctx *ctx, void *hip_stream);
int edison_reset_stream(edison_ctx *ctx);
int edison_sync(edison_ctx *ctx);
int edison_device_info(edison_ctx *ctx, char *name, int name_cap, int *n_cu, int64_t *hbm_bytes);

/* Mel filterbank parameters (defaults = audio/config.py). Rebuilds the device tables of both variants.
 * num_mel_bins must be 32 and the frame length 1024 on this path (EDISON_E_NO_IMPL otherwise).          */
int edison_mfcc_configure(edison_ctx *ctx, double sample_rate, double lower_edge_hertz, double upper_edge_hertz,
                          double mel_mtx_scale);

/* gen_mel_weight_matrix (mfcc_utils.py:36-73), host-side, float64: W[num_spectrogram_bins][num_mel_bins]. */
int edison_gen_mel_weight_matrix(int num_mel_bins, int num_spectrogram_bins, double sample_rate,
                                 double lower_edge_hertz, double upper_edge_hertz, double *W);

/* Load the int8 CNN (an .ednn blob written by tools/import_weights_h.py from an NNoM weights.h): the GPU's
 * nnom_model_create() + model_compile() (weights.h:138-161, nnom.c:758-900). Accepted: any chain
 * Input -> {Conv2D valid|same [+ReLU] | MaxPool valid|same | Dense [+ReLU] | Flatten | Softmax (last)}* -> Output whose
 * activations fit 2 x 32 KB. The shipped kws_conv graph (and any retrained model of that shape) runs on the matrix
 * cores; every other graph runs on the general layer-by-layer kernel. EDISON_E_SIZE: malformed graph or one the
 * reference itself would reject; EDISON_E_NO_IMPL: a layer/shape the reference runs but this path does not.  */
int edison_model_load(edison_ctx *ctx, const char *ednn_path);
int edison_model_load_mem(edison_ctx *ctx, const void *blob, size_t blob_bytes);

/* ---- any NNoM graph: shapes come from the loaded model ---------------------------------------------------
 * edison_net_batch = model_run() (nnom.c:975-1040) + nnom_predict()'s first-maximum argmax (nnom_utils.c:258-305)
 * for n inputs; edison_net_layers = the outputs of every layer, as a layer callback installed with
 * model_set_callback() (nnom.c:1043) would see them.
 *   in      [n][in_h*in_w*in_c] int8, HWC                  logits  [n][n_out] int8: output of the layer before a final
 *   softmax [n][n_out] int8 (NULL if the graph has none)            Softmax, or of the last layer if there is none
 *   argmax  [n] int32 first maximum of the LAST layer's output      acts    [n][acts_bytes]: layer outputs back to back
 * The fixed-shape entry points below (edison_cnn_*, edison_kws_*, edison_stream_*) accept any loaded model that maps
 * 31x13x1 features to 10 softmax classes. */
typedef struct edison_net_info {
	int32_t in_h, in_w, in_c; /* Input(shape(h, w, c))                                                          */
	int32_t n_out;            /* elements of the last layer's output                                            */
	int32_t n_layers;         /* compute layers (Input/Output/Flatten are not counted)                          */
	int32_t acts_bytes;       /* bytes per input of edison_net_layers                                           */
	int32_t has_softmax;
	int32_t accelerated;      /* 1: the kws_conv graph, served by its specialised matrix-core kernel; 2: another graph,
	                           * served by the general matrix-core kernel; 0: layer-by-layer VALU kernel only      */
} edison_net_info;
typedef struct edison_net_layer_info_t {
	int32_t type;             /* 1 Conv2D, 2 MaxPool, 3 Dense, 4 Softmax                                        */
	int32_t out_h, out_w, out_c;
	int32_t acts_offset;      /* where this layer's output starts inside one input's acts record                */
	int32_t relu;
} edison_net_layer_info_t;
int edison_net_get_info(edison_ctx *ctx, edison_net_info *out);
int edison_net_layer_info(edison_ctx *ctx, int layer, edison_net_layer_info_t *out);
int edison_net_batch_dev(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *logits, int8_t *softmax, int32_t *argmax);
int edison_net_layers_dev(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *acts);
int edison_net_batch(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *logits, int8_t *softmax, int32_t *argmax);
int edison_net_layers(edison_ctx *ctx, const int8_t *in, int64_t n, int8_t *acts);
/* The loaded graph's OWN kernel. NNoM fixes shapes, buffers and per-layer kernels once, in model_compile()
 * (nnom.c:758-900); edison_net_specialize() goes one step further and compiles the general matrix-core kernel's source with
 * this graph's plan as constants (0.8-1.8 s with the installed hipcc as a child process -- $EDISON_HIPCC, else $ROCM_PATH/bin/hipcc,
 * else /opt/rocm/bin/hipcc, never $PATH -- or hipRTC in this process). Code objects are cached on disk: $EDISON_JIT_CACHE, else
 * $XDG_CACHE_HOME/edison_amd, else $HOME/.cache/edison_amd ("off" disables the cache); the directory is created 0700 and used only
 * if it belongs to this user and nobody else can write it; an entry is keyed by graph, kernel text, compiler options and the
 * compiler's identity. From then on every entry point that runs the general kernel for this load (edison_net_batch*, and
 * edison_cnn_* / edison_kws_* / edison_stream_* for graphs other than kws_conv) runs the graph's own: same arithmetic,
 * bit-identical outputs, 1.7-3 x the general kernel on the fixture graphs (profiles/r03_net_own_kernel_all_graphs.txt).
 * EDISON_E_NO_IMPL: the graph has no matrix-core plan, or neither hipcc nor libhiprtc.so is installed -- the graph stays on the general
 * kernel. A model load by itself does NONE of this (no compiler, no file, no cached code object): it is this call, or the
 * caller's wish in the environment, read at every load -- EDISON_NET_SPECIALIZE=1: every load ends with this call (a failure
 * there never fails the load), =cache: a load takes the own kernel if an earlier call left it in the cache.
 * edison_net_specialized: 0 general kernel; own kernel: 1 compiled just now by a hipcc child process (the installed ROCm's
 * compiler, tried first), 2 taken from the cache, 3 compiled just now by hipRTC in this process (EDISON_JIT_COMPILER=hipcc|hiprtc
 * picks one).
 * edison_net_spec_source (host only, no context): the generated constants of an .ednn blob, for inspection / offline builds;
 * EDISON_E_SIZE with *need = bytes wanted when cap is too small. */
int edison_net_specialize(edison_ctx *ctx);
int edison_net_specialized(edison_ctx *ctx);
int edison_net_spec_source(const void *ednn_blob, size_t blob_bytes, char *out, size_t cap, size_t *need);
/* Host only, for tests and inspection: the plans of an .ednn blob as the matrix-core kernels get them (raw structs of
 * csrc/edison_internal.h: ed_net_plan_t, ed_mm_plan_t), the packed weight fragments and the accumulator seeds; any out pointer
 * may be NULL, *_need report the bytes wanted. edison_net_plan_layout(i): struct sizes / field offsets (see net_spec.c) so that a
 * reader outside C need not restate the layout -- tests/plan_emulator.py walks a plan in numpy exactly as the kernel does. */
int edison_net_plan_dump(const void *ednn_blob, size_t blob_bytes, void *plan_out, size_t plan_cap, void *mm_out, size_t mm_cap,
                         void *frag_out, size_t frag_cap, size_t *frag_need, void *seeds_out, size_t seeds_cap, size_t *seeds_need);
size_t edison_net_plan_layout(int which);

/* ---- device memory helpers (so a C caller needs nothing but this library) ---------------------------- */
int edison_dev_alloc(edison_ctx *ctx, size_t bytes, void **dptr);
int edison_dev_free(edison_ctx *ctx, void *dptr);
int edison_dev_upload(edison_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int edison_dev_download(edison_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---- multi-GPU: one process (one context) per GPU, one RCCL all-gather of the logits ---------------------------
 * The reference has no counterpart (single MCU, UART host link: hostinterface.c:96-112). Utterances are
 * sharded contiguously over the ranks (edison_dist_shard_range); frames and utterances are independent, so the data
 * path has no collective; the only exchange is the all-gather of the per-class int8 logits (10 B per utterance) over
 * xGMI. Protocol: rank 0 calls edison_dist_unique_id and hands the 128 bytes to every rank by whatever means the host
 * program has (MPI, a file, a socket, torch.distributed); every rank then calls edison_dist_init on its context.
 * RCCL is bound at run time (librccl.so.1, or EDISON_RCCL_LIB); EDISON_E_NO_IMPL when it cannot be found. */
#define EDISON_DIST_ID_BYTES 128
int edison_dist_available(void); /* EDISON_OK when RCCL can be bound in this process (binds it, creates nothing) */
int edison_dist_unique_id(void *id128);
int edison_dist_init(edison_ctx *ctx, const void *id128, int rank, int world_size);
int edison_dist_info(const edison_ctx *ctx, int *rank, int *world_size);
int edison_dist_shutdown(edison_ctx *ctx); /* also done by edison_shutdown */
int edison_dist_shard_range(int64_t n_items, int rank, int world_size, int64_t *lo, int64_t *hi);
/* device pointers; every rank passes the same n_local_utt; rank r's rows land at all_logits + r * n_local_utt * 10.
 * Asynchronous on the context's stream. A context outside any communicator is a world of one (plain copy). */
int edison_dist_allgather_logits(edison_ctx *ctx, const int8_t *local_logits, int64_t n_local_utt, int8_t *all_logits);
/* a batch the world size does not divide: every rank passes the TOTAL n_total_utt, holds the shard
 * edison_dist_shard_range(n_total_utt, rank, world) in local_logits, and receives all n_total_utt rows in rank order
 * (shards padded to the largest one inside, still one ncclAllGather). The equal-shard call above must NOT be used with
 * different n_local_utt on different ranks (it cannot tell; ncclAllGather would mis-place or overrun the rows). */
int edison_dist_allgather_logits_total(edison_ctx *ctx, const int8_t *local_logits, int64_t n_total_utt, int8_t *all_logits);
/* edison_kws_batch_dev on this rank's shard followed by the all-gather: BASELINE config 4 in one call per rank */
int edison_kws_batch_sharded_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_local_utt, int64_t utt_stride,
                                 int8_t *feat, int8_t *logits_local, int8_t *softmax, int32_t *argmax, int8_t *logits_all);
/* ... with shards cut by edison_dist_shard_range from n_total_utt (audio = this rank's shard only) */
int edison_kws_batch_sharded_total_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_total_utt, int64_t utt_stride,
                                       int8_t *feat, int8_t *logits_local, int8_t *softmax, int32_t *argmax, int8_t *logits_all);

/* ---- the hot path, batched, device pointers ----------------------------------------------------------- */

/*
 * n_frames frames of 1024 int16 samples; frame i starts at audio + i*frame_step (samples; 1024 = the
 * reference's hop, 512 = 50 % overlap). Writes the first n_coef (1..32) coefficients of every frame:
 *   mfcc  [n_frames][n_coef] fp32            (may be NULL)
 *   feat  [n_frames][n_coef] int8 net input  (may be NULL) = round_half_even(clip(mfcc*feat_scale,-128,127)),
 *         the reference's np.clip(x*scale,-128,127).round().astype(int8) (kws_nnom.py:359-361)
 * Replaces the per-frame loops mfcc_utils.py:160-197 (A) and :287-322 (B).
 */
int edison_mfcc_batch_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                          int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale);

/*
 * The same over n_rows independent rows of samples (the reference's batch_mfcc, mfcc_utils.py:75-131: one utterance per
 * row of data[n, samples]): row r starts at audio + r*row_stride (samples), its frame i at + i*frame_step; every row
 * yields frames_per_row frames. Outputs are [n_rows * frames_per_row][n_coef], row-major over (row, frame). ONE kernel
 * launch for the whole array (the kernel's grouped addressing), not one call per row.
 */
int edison_mfcc_rows_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_rows, int64_t row_stride, int64_t frames_per_row,
                         int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale);

/*
 * The same over n_batches INDEPENDENT batches -- batch b's samples at audio[b] (any address), its outputs at mfcc[b] / feat[b] (either
 * ARRAY may be NULL: that output is not written; an array that is given holds no NULL entries) -- of n_frames_each frames each, in ONE
 * launch per 16 batches (`audio`, `mfcc`, `feat` are HOST arrays of DEVICE pointers, read during the call). Rows of batch_mfcc that
 * do not share an allocation (mfcc_utils.py:75-131: rows are independent), or batches arriving from different producers. Why it
 * exists: a 65 536-frame launch idles ~15 % of its window (start-up, drain), a launch that runs on does not (0.41-0.42 of 8 TB/s
 * against 0.37-0.40), and keeping the NEXT launch in flight on a second HIP queue (below) recovers a quarter of it at best
 * (profiles/r05_mfcc_two_queues_notes.txt). Variants A and B (EDISON_E_NO_IMPL otherwise). Results are bit-identical to one
 * edison_mfcc_batch_dev call per batch.
 */
int edison_mfcc_batches_dev(edison_ctx *ctx, int n_batches, const int16_t *const *audio, int64_t n_frames_each, int64_t frame_step,
                            int variant, int n_coef, float *const *mfcc, int8_t *const *feat, float feat_scale);

/*
 * One call per batch with the NEXT launch already in flight: two library-owned HIP queues per context (created with different
 * priorities by default: HIP shares a pool of hardware queues among the streams of ONE priority, two such streams may serialise).
 *   edison_queues_fork(ctx)                  both queues wait for what the context's stream holds so far (the batches' producers)
 *   edison_mfcc_batch_queue_dev(ctx, q, ...) edison_mfcc_batch_dev on queue q = 0 / 1; batches in flight together must be independent
 *                                            (own input, own outputs); alternate q from call to call
 *   edison_queues_join(ctx)                  the context's stream waits for both queues; outputs are ordered as usual after it
 * Between fork and join the context's stream must not be changed (edison_set_stream) and only the queue call may be used. Results
 * are bit-identical to the plain calls. From a host that keeps both queues fed (a C host, ~5 us per call) a 65 536-frame batch
 * takes 45.7 us instead of 47.7 (+4.5 %; +1 ... +2.6 % on boards that sit at their power cap); a host that needs as long per call as the
 * GPU per batch gains nothing
 * (profiles/r05_mfcc_two_queues_notes.txt). A fork / join pair itself costs ~25-30 us of cross-queue signalling: keep MANY batches between
 * them -- at 65 536 frames per batch the two queues break even near 25 batches per fork (profiles/r05_two_queues_region_length.txt).
 * The list call above is the faster way whenever the batches are known together.
 * Which two streams: whether the next launch really backfills the CUs this one leaves depends on where runtime and driver put the two
 * streams' hardware queues, which HIP does not let a program choose -- of all pairs of nine streams in one process a third gained, a
 * third changed nothing, a third LOST 8-10 %. edison_queues_calibrate measures it: every pair of the context's five candidate streams
 * (three of the least, two of the greatest priority) and the serial sequence, interleaved, on the caller's own batch (`audio`: device
 * pointer, n_frames frames, read only), the winner once more against the serial sequence; the pair is kept only if it is at least 1 %
 * faster, otherwise both queue indices mean one stream and the queue calls ARE the serial sequence (never slower than it). ~0.17 s for a
 * 65 536-frame batch; call it once, after edison_init / edison_mfcc_configure, with a batch of the size that will be used. Outputs (each
 * may be NULL): microseconds per batch of the serial sequence and of what was kept, pair_kept = 10 * i + j (candidate indices) or 0.
 * A batch whose launch takes more than a millisecond keeps one queue without further measurement (its fixed cost is microseconds).
 * Without a calibration the queues are one stream of each priority.
 */
int edison_queues_calibrate(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step, int variant, double *serial_us,
                            double *best_us, int *pair_kept);
int edison_queues_fork(edison_ctx *ctx);
int edison_queues_join(edison_ctx *ctx);
int edison_mfcc_batch_queue_dev(edison_ctx *ctx, int queue, const int16_t *audio, int64_t n_frames, int64_t frame_step, int variant,
                                int n_coef, float *mfcc, int8_t *feat, float feat_scale);

/*
 * Same, plus every intermediate the reference returns in its per-frame dict (mfcc_utils.py:161-197):
 *   fft   [n][513][2] fp32  un-normalised X[k], k=0..512 (re,im)      spec   [n][513] fp32 |X[k]| (A) or |X[k]|/1024/sqrt2 (B)
 *   mel   [n][32] fp32 mel_spectrogram                                 logmel [n][32] fp32 log_mel_spectrogram
 * Any of them may be NULL. This is the diagnostic path (parity tests, `main.py mfcc host`), not the fast one.
 */
int edison_mfcc_stages_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                           int variant, float *fft, float *spec, float *mel, float *logmel, float *mfcc32);

/*
 * The generality path: variants A, B and TF for ANY geometry the reference's Python functions accept -- frame_len (4 .. 4096, also no power
 * of two: the reference calls numpy.fft.fft), mel_nbins (1 .. 256), sample rate, filterbank edges, mel_mtx_scale (mfcc_utils.py:134-199,
 * 255-323; every caller in the reference passes audio/config.py's 1024 / 32, which the entry points above serve). Float64 on the GPU
 * (direct DFT against a host-built table, dense mel product, ln, DCT-II with the variant's constants -- for B the literal 1/1024 and
 * 1/64 of mfcc_utils.py:297,318 whatever the frame length): the outputs are the reference's to ~1e-12. One workgroup per frame; not a
 * throughput path. Outputs, float64, each may be NULL: fft [n][F][2] (re, im), spec [n][F] with F = frame_len/2 (A) or frame_len (B) as
 * in the reference's per-frame dict; mel, logmel, mfcc [n][mel_nbins]; feat [n][n_coef] int8 = the net-input rounding of the first
 * n_coef coefficients (kws_nnom.py:359-361). EDISON_E_NO_IMPL outside the limits above or for other variants.
 * Variant TF (mfcc_utils.py:201-253 with fft_len == frame_len): the samples as float32 times tf.signal's periodic Hann window in float32,
 * then float64 -- rfft, |.|, the (frame_len/2+1)-bin mel matrix, ln(x + 1e-6), DCT-II / sqrt(2 mel_nbins); F = frame_len/2 + 1 (the caller cuts the
 * DC bin as the reference does); mel_mtx_scale is ignored. PARITY UNPINNED like the fast path's variant TF: no TensorFlow in this image.
 */
int edison_mfcc_generic_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int frame_len, int64_t frame_step, int variant, int mel_nbins,
                            double sample_rate, double lower_edge_hertz, double upper_edge_hertz, double mel_mtx_scale, double *fft, double *spec,
                            double *mel, double *logmel, double *mfcc, int n_coef, int8_t *feat, float feat_scale);
int edison_mfcc_generic(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int frame_len, int64_t frame_step, int variant, int mel_nbins,
                        double sample_rate, double lower_edge_hertz, double upper_edge_hertz, double mel_mtx_scale, double *fft, double *spec,
                        double *mel, double *logmel, double *mfcc, int n_coef, int8_t *feat, float feat_scale);

/*
 * n_utt feature maps [31][13] int8 (403 B each, HWC, as aiRunInference's in_data) ->
 *   logits  [n_utt][10] int8 dense output      softmax [n_utt][10] int8 (= aiRunInference's out_data)
 *   argmax  [n_utt] int32 first maximum of the softmax output (nnom_predict, nnom_utils.c:275-284)
 * Each output may be NULL. Replaces model_run() (nnom.c:1037) over the graph of weights.h:138-161.
 */
int edison_cnn_batch_dev(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *logits, int8_t *softmax,
                         int32_t *argmax);

/* Per-layer activations of the CNN for parity tests: acts[n_utt][6496] = conv1+relu(3888) | pool1(1872) is
 * not kept separately on the fast path, so this diagnostic entry runs the unfused kernels:
 * conv1 3888 | pool1 1872 | conv2 2464 | pool2 1120 | conv3 960 | conv4 96 | dense 10 | softmax 10 = 10420 B. */
#define EDISON_CNN_ACT_BYTES 10420
int edison_cnn_layers_dev(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *acts);

/*
 * Full keyword spotting: utterance u = 31 frames of 1024 samples starting at audio + u*utt_stride (samples;
 * the reference's utterances are 32000 samples of which 31*1024 are used, audio/config.py:12,25,31).
 * Variant B features (the ones the net was trained on, kws_keras.py:47) -> int8 -> CNN.
 * feat [n_utt][403] (may be NULL) receives the int8 net input.
 */
int edison_kws_batch_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride,
                         int8_t *feat, int8_t *logits, int8_t *softmax, int32_t *argmax);

/*
 * Variant C with its native types: what the firmware keeps in bufDctInline / sends with audioDumpToHost
 * (audioprocessing.c:221-231). mfcc [n_frames][n_coef] int16; feat [n_frames][n_coef] int8 = mfccToNetInput's NNoM
 * branch (C division by NNOM_INPUT_SCALE = 1, clip to [-128,127], app.c:686-694). Stage dumps, all int16:
 *   fft [n][513][2] X[k] (re,im), k = 0..512, natural order     spec [n][513] bufSpect     mel [n][32] bufMelSpect
 * EDISON_E_NO_IMPL when the configured filterbank cannot be expressed as the firmware's compact tables.
 */
int edison_mfcc_q15_batch_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step, int n_coef,
                              int16_t *mfcc, int8_t *feat);
int edison_mfcc_q15_stages_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step,
                               int16_t *fft, int16_t *spec, int16_t *mel, int16_t *mfcc32);
/* edison_kws_batch_dev with the firmware's own features: variant C -> NNoM clip -> CNN, i.e. what the board
 * answers for the same samples (appHifMfccAndInference, app.c:167-221). */
int edison_kws_batch_q15_dev(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride, int8_t *feat,
                             int8_t *logits, int8_t *softmax, int32_t *argmax);

/* ---- the same with HOST pointers (upload, run, download, synchronise) -------------------------------- */
int edison_mfcc_q15_batch(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step, int n_coef,
                          int16_t *mfcc, int8_t *feat);
int edison_mfcc_q15_stages(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step, int16_t *fft,
                           int16_t *spec, int16_t *mel, int16_t *mfcc32);
int edison_kws_batch_q15(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride, int8_t *feat,
                         int8_t *logits, int8_t *softmax, int32_t *argmax);
int edison_mfcc_batch(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step, int variant,
                      int n_coef, float *mfcc, int8_t *feat, float feat_scale);
int edison_mfcc_stages(edison_ctx *ctx, const int16_t *audio, int64_t n_frames, int64_t frame_step, int variant,
                       float *fft, float *spec, float *mel, float *logmel, float *mfcc32);
int edison_mfcc_rows(edison_ctx *ctx, const int16_t *audio, int64_t n_rows, int64_t row_stride, int64_t frames_per_row,
                     int64_t frame_step, int variant, int n_coef, float *mfcc, int8_t *feat, float feat_scale);
int edison_cnn_batch(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *logits, int8_t *softmax,
                     int32_t *argmax);
int edison_cnn_layers(edison_ctx *ctx, const int8_t *feat, int64_t n_utt, int8_t *acts);
int edison_kws_batch(edison_ctx *ctx, const int16_t *audio, int64_t n_utt, int64_t utt_stride, int8_t *feat,
                     int8_t *logits, int8_t *softmax, int32_t *argmax);

/* ---- continuous-microphone mode ------------------------------------------------------------------------
 * The firmware's appMicMfccInfereContinuous / appAudioEvent loop (firmware/src/app.c:288-371, 635-663): every new
 * frame -> MFCC -> mfccToNetInputPush (31-row sliding window, app.c:706-719) -> inference. A push delivers
 * chunk_frames hops of `hop` NEW samples (hop 1024 = firmware cadence, 512 = 50 % overlap); the stream keeps the
 * last 1024-hop samples and the last 30 feature rows on the device and starts from silence / a zero window like
 * the firmware's static buffers. Output i of a push belongs to the window ending with its i-th new frame.
 * One push = one hipGraph launch (MFCC kernel, CNN kernel over sliding windows, history shift).            */
typedef struct edison_stream edison_stream;
int edison_stream_create(edison_ctx *ctx, int hop, int chunk_frames, edison_stream **out);
void edison_stream_destroy(edison_stream *s);
int edison_stream_reset(edison_stream *s);
int edison_stream_push_dev(edison_stream *s, const int16_t *samples /* device, chunk*hop */, int8_t *logits,
                           int8_t *softmax, int32_t *argmax /* device, may be NULL */);
/* n_frames <= chunk_frames new frames (n_frames * hop samples, outputs [n_frames][..]): the ragged last push of a recording that the
 * chunk does not divide (the firmware has no counterpart: its microphone never ends, app.c:288-371). Either launch mode (under
 * EDISON_STREAM_LAUNCH_GRAPH a short push runs the same kernels launched directly: the captured graph holds the chunk). The filtered
 * outputs of such a push: its n_frames entries. */
int edison_stream_push_n_dev(edison_stream *s, const int16_t *samples /* device, n_frames*hop */, int n_frames, int8_t *logits,
                             int8_t *softmax, int32_t *argmax /* device, may be NULL */);
int edison_stream_push(edison_stream *s, const int16_t *samples /* host */, int8_t *logits, int8_t *softmax,
                       int32_t *argmax);
int64_t edison_stream_frames_seen(const edison_stream *s);

/* Stream options beyond hop / chunk. mfcc_variant: EDISON_MFCC_B (host float model, default) or EDISON_MFCC_C (the
 * firmware's own Q15 features). filter = 1 adds the firmware's output post-processing to every inference of a push
 * (app.c:332-356): netOutFilt[c] = (float)(alpha*netOutFilt[c] + (1.0-alpha)*(float)netOutput[c]) in double
 * arithmetic, arm_max_f32 (first maximum) over the ten filtered outputs, "spotted" if that maximum exceeds
 * true_threshold. Defaults (edison_stream_default_opts): alpha 0.9 = NET_OUT_MOVING_AVG_ALPHA for NNoM (app.c:38),
 * threshold 0.5 = TRUE_THRESHOLD (app.c:34); the filter state starts at zero (app.c:299-300) and survives pushes. */
typedef struct edison_stream_opts
{
	int hop, chunk_frames;
	int mfcc_variant;
	int filter;
	double filter_alpha;
	double true_threshold;
	int launch_mode;  /* EDISON_STREAM_LAUNCH_DIRECT (default): the kernels of a push are launched one by one;
	                   * EDISON_STREAM_LAUNCH_GRAPH: every push replays the hipGraph captured at creation (device pushes: the
	                   * MFCC / CNN / filter / shift nodes; host pushes: upload + those + download). Same results; on this
	                   * platform the replay measures slower. Default from EDISON_STREAM_GRAPH=1. */
	int fsm;          /* 1 (needs filter = 1): the firmware's state machine (edisonFSM, app.c:727-928; edison_fsm below) as the last
	                   * stage of every push, on the GPU: the inferences of the push walk through it in order, the time between
	                   * two of them being the hop (dt = hop / 16 kHz, the cadence of appAudioEvent); state per inference and
	                   * the machine itself: edison_stream_fsm*. The machine starts in RESET and survives pushes. */
} edison_stream_opts;
#define EDISON_STREAM_LAUNCH_DIRECT 0
#define EDISON_STREAM_LAUNCH_GRAPH 1
void edison_stream_default_opts(edison_stream_opts *o);
int edison_stream_create_ex(edison_ctx *ctx, const edison_stream_opts *o, edison_stream **out);
/* Filtered outputs of the LAST push: filt [n][10] fp32 (netOutFilt after each inference), likely [n] int32 (predMaxIdx), spotted [n]
 * int32 (predMaxIdx where predMax > threshold, else -1); n = the frames of that push (chunk_frames, or the n_frames of
 * edison_stream_push_n_dev: exactly n entries are written). Each may be NULL. */
int edison_stream_filtered(edison_stream *s, float *filt /* host */, int32_t *likely, int32_t *spotted);
int edison_stream_filtered_dev(edison_stream *s, float *filt /* device */, int32_t *likely, int32_t *spotted);
/* The state machine of a stream created with fsm = 1, after the LAST push: *fsm = the machine (may be NULL), states [n] int32 =
 * the state it was in after each of the n inferences of that push (may be NULL; n as for edison_stream_filtered). */
struct edison_fsm;
int edison_stream_fsm(edison_stream *s, struct edison_fsm *fsm /* host */, int32_t *states /* host */);
int edison_stream_fsm_dev(edison_stream *s, int32_t *states /* device */);

/* ---- the firmware's home-automation state machine (edisonFSM, app.c:727-928), host side, without the LEDs -------
 * RESET -> IDLE -(wake word "edison" spotted)-> HOT -(a location spotted)-> LOC -(a value spotted)-> SET -> IDLE;
 * HOT and LOC fall back to IDLE after EDI_LOC_TIMEOUT = 5000 ms (app.c:48). Time advances by dt_us per call exactly
 * like the firmware's `hotTimeout += dt/1000` (integer milliseconds per call). One step per inference. */
#define EDISON_FSM_RESET 0
#define EDISON_FSM_IDLE 1
#define EDISON_FSM_HOT 2
#define EDISON_FSM_LOC 3
#define EDISON_FSM_SET 4
typedef struct edison_fsm
{
	int state;
	uint32_t hot_timeout_ms;
	int wake_idx;
	int loc_idx, val_idx;       /* keyword indices of the pending location / value */
	int last_loc, last_val;     /* the last executed command (keyword indices), -1 before the first */
	uint32_t commands;          /* how many "location value" commands were executed */
} edison_fsm;
void edison_fsm_init(edison_fsm *f);
/* pred_max / pred_idx: arm_max_f32 of the filtered outputs; returns the new state. */
int edison_fsm_step(edison_fsm *f, float pred_max, uint32_t pred_idx, uint32_t dt_us, double true_threshold);
/* The whole post-processing chain of the firmware (app.c:332-371) on n network outputs in time order, as one GPU stage and without a
 * stream: moving average -> first maximum -> threshold -> state machine. softmax [n][10] int8 (host); filt_state [10] fp32 in/out
 * (netOutFilt, zeros at the start), fsm in/out (edison_fsm_init at the start; NULL: no state machine), dt_us = time between two
 * inferences; outputs (host, each may be NULL): filt [n][10], likely [n], spotted [n], states [n]. n < 2^30. The recurrence rounds at every
 * step, so the chain is sequential in time by definition: ONE workgroup (ten lanes filter, one walks the machine), cost linear in n -- a
 * post-processing stage for the outputs of a stream, not a batch kernel. EDISON_E_ARGUMENT: alpha outside [0, 1], a threshold that is
 * not a number, an fsm->state that is not a state of the machine (what edison_fsm_step answers for it). */
int edison_postproc(edison_ctx *ctx, const int8_t *softmax, int64_t n, double alpha, double true_threshold, uint32_t dt_us,
                    float *filt_state, edison_fsm *fsm, float *filt, int32_t *likely, int32_t *spotted, int32_t *states);
/* roles of the ten classes in the state machine: the wake word's class index (-1: none), bit masks of the locations and values */
void edison_fsm_roles(int32_t *wake_idx, uint32_t *loc_mask, uint32_t *val_mask);

/* ---- legacy call surface of the reference firmware (batch = 1, process-global context) --------------- */
/* firmware/src/ai/ai.h:74-80. aiInitialize() creates the global context on device $EDISON_DEVICE (default 0)
 * and loads $EDISON_MODEL (default: kws_nnom.ednn next to the library). in_data: 403 int8, out_data: 10 int8. */
int aiInitialize(void);
void aiPrintInfo(void);
void aiGetInputShape(uint16_t *x, uint16_t *y);
int aiRunInference(void *in_data, void *out_data);
const char *aiGetKeywordFromIndex(uint32_t idx);
uint32_t aiGetKeywordCount(void);
/* firmware/src/ai/ai_nnom.c:64-132 */
void aiNnomTest(void);
void aiNnomInit(void);
void aiNnomPrintInfo(void);
int aiNnomRunInference(void *in_data, void *out_data);
int aiNnomPredict(uint32_t *label, float *prob);
int8_t *aiNnomGetInputBuffer(void);
int8_t *aiNnomGetOutputBuffer(void);
/* firmware/src/app.c:152-153: write / push one frame's MFCCs into the process-global 403-byte net input
 * (clip to [-128,127] after integer division by NNOM_INPUT_SCALE = 1, app.c:685-693).                    */
void mfccToNetInput(int16_t *mfcc, uint16_t in_x, uint16_t in_y, uint32_t xoffset);
void mfccToNetInputPush(int16_t *mfcc, uint16_t in_x, uint16_t in_y);
/* firmware/src/audioprocessing.h:21-22: one 1024-sample frame -> pointer to a callee-owned static buffer of 32
 * int16 (valid until the next call): MFCC variant C, the firmware's own Q15 arithmetic. */
void audioInit(void);
void audioCalcMFCCs(int16_t *inp, int16_t **oup);
/* firmware/src/audio/mfcc.h:64-67 -- MFCC variant D, the float32 ML-KWS extractor of the firmware's NNoM example
 * (app.c:540,583: mfcc_create(13, 1, 512, 8, 0.97f), hop 256): pre-emphasis, Hann window, FFT of the frame padded to a
 * power of two, 26 mel bands 20..4000 Hz, logf, DCT rows feature_offset..num_mfcc_features-1, * 2^mfcc_dec_bits,
 * round, saturate to q7. The handle is opaque (the firmware's struct fields are private scratch); `mfcc_out` receives
 * num_mfcc_features - feature_offset int8. mfcc_create returns NULL on failure (bad arguments, padded length outside
 * 128..1024, no GPU). Same struct tag as the firmware's header, so both headers can be included together. */
#ifndef __KWS_MFCC_H__
typedef struct _mfcc_t mfcc_t;
#endif
mfcc_t *mfcc_create(int num_mfcc_features, int feature_offset, int frame_len, int mfcc_dec_bits, float preemph);
void mfcc_delete(mfcc_t *mfcc);
void mfcc_compute(mfcc_t *mfcc, const int16_t *audio_data, int8_t *mfcc_out);
/* mfcc.h:61, mfcc.c:101-115: the DCT-II matrix of that extractor, float32, [coefficient][input], malloc'd (host) */
float *create_dct_matrix(int32_t input_length, int32_t coefficient_count);
/* the same on an explicit context, and batched: frame i starts at audio + i*frame_step; out [n][n_out] q7,
 * out_f32 [n][n_out] (the scaled sums before round/saturate) and logmel [n][26] may be NULL */
mfcc_t *edison_mfcc_f32_create(edison_ctx *ctx, int num_mfcc_features, int feature_offset, int frame_len,
                               int mfcc_dec_bits, float preemph);
int edison_mfcc_f32_n_out(const mfcc_t *mfcc);
int edison_mfcc_f32_batch_dev(mfcc_t *mfcc, const int16_t *audio, int64_t n_frames, int64_t frame_step, int8_t *out,
                              float *out_f32, float *logmel);
int edison_mfcc_f32_batch(mfcc_t *mfcc, const int16_t *audio, int64_t n_frames, int64_t frame_step, int8_t *out,
                          float *out_f32, float *logmel);

/* The front end of the firmware's NNoM keyword-spotting example around mfcc_compute (appNnomKwsRun, app.c:545-623): every
 * audio event delivers 512 new samples behind the last 256 old ones (app.c:567-575), two frames are extracted at offsets 0
 * and 256 (app.c:583) into a ring of window_rows (MFCC_LEN = 63) feature rows, and the network input is that ring oldest
 * row first (mfcc_features_seq, app.c:600-604). A push takes n_events <= max_events events (one kernel launch for all
 * their frames) and returns the window after every event: windows [n_events][window_rows][n_out] int8. State (samples and
 * rows) starts as zeros, like the firmware's static buffers. The extractor must be a 512-sample one on the same context and
 * must outlive the stream (edison_f32_stream_destroy before mfcc_delete). */
typedef struct edison_f32_stream edison_f32_stream;
int edison_f32_stream_create(edison_ctx *ctx, mfcc_t *mfcc, int window_rows, int max_events, edison_f32_stream **out);
void edison_f32_stream_destroy(edison_f32_stream *s);
int edison_f32_stream_reset(edison_f32_stream *s);
int64_t edison_f32_stream_events_seen(const edison_f32_stream *s);
int edison_f32_stream_push_dev(edison_f32_stream *s, const int16_t *samples, int n_events, int8_t *windows);
int edison_f32_stream_push(edison_f32_stream *s, const int16_t *samples, int n_events, int8_t *windows);
/* One 1024-sample frame through the GPU MFCC, any variant; out32 fp32. */
int edison_mfcc_frame(const int16_t *frame1024, int variant, float *out32);
edison_ctx *edison_global_ctx(void);

#ifdef __cplusplus
}
#endif
#endif
