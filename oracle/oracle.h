/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE. CPU restatement of edison's keyword-spotting hot path.
 *
 * This is the parity CHECKER and the timed CPU baseline ("port"); it is never the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. The product
 * (edison_amd/, include/edison_hip.h) must not link, import or call anything declared here.
 *
 * Pinning: the int8 CNN restatement is checked bit-for-bit against the reference itself
 * (oracle/_ref/libnnom_ref.so = NNoM 0.3.0 + CMSIS-NN + weights.h compiled from /root/reference) and
 * against tests/golden/cnn_golden.npz produced from it; the MFCC restatement is checked against
 * tests/golden/mfcc_golden.npz produced by importing the reference's audio/edison/mfcc/mfcc_utils.py
 * (tests/golden/gen_fixtures.py is the generating script).
 */
#ifndef EDISON_ORACLE_H
#define EDISON_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- MFCC (float64, like the numpy reference) -------------------------------------------------- */

#define ORACLE_MFCC_VARIANT_A 0 /* mfcc_utils.mfcc      (audio/edison/mfcc/mfcc_utils.py:134-199) */
#define ORACLE_MFCC_VARIANT_B 1 /* mfcc_utils.mfcc_mcu  (audio/edison/mfcc/mfcc_utils.py:255-323) */
/* mfcc_utils.mfcc_tf (audio/edison/mfcc/mfcc_utils.py:201-253). PARITY UNPINNED: the arithmetic lives in TensorFlow
 * (tensorflow==2.1.0, audio/requirements.txt), which is neither under /root/reference nor installed here, and the reference
 * holds no output of it. Restated in float64 from tf.signal's published definitions (stft: periodic Hann window, rfft;
 * linear_to_mel_weight_matrix -- which gen_mel_weight_matrix above is the reference's own numpy port of;
 * mfccs_from_log_mel_spectrograms: DCT-II * rsqrt(2 * num_mel_bins)). spectrogram is [n_frames][frame_len/2 + 1]. */
#define ORACLE_MFCC_VARIANT_TF 3

/* gen_mel_weight_matrix (mfcc_utils.py:36-73): W is [num_spectrogram_bins][num_mel_bins] row-major. */
void oracle_mel_weight_matrix(int num_mel_bins, int num_spectrogram_bins, double sample_rate,
                              double lower_edge_hertz, double upper_edge_hertz, double *W);

/*
 * n_frames frames of frame_len (must be 1024-like power of two) samples taken every frame_step samples
 * from x. variant A: fft[:N/2] -> |.| -> mel(N/2 bins) -> log(+1e-6) -> dct2/sqrt(2*nmel).
 * variant B: fft/N -> |.|/sqrt2 -> [:N/2+1] . (scale*mel(N/2+1 bins)) / scale -> [log] -> dct2/64.
 * Outputs (each may be NULL): spectrogram [n][nspec] (A: N/2, B: N), mel [n][nmel],
 * logmel [n][nmel] (B without use_log: copy of mel), mfcc [n][nmel].
 */
int oracle_mfcc(const int16_t *x, int64_t n_frames, int frame_len, int64_t frame_step, int variant,
                int num_mel_bins, double sample_rate, double lower_edge_hertz, double upper_edge_hertz,
                double mel_mtx_scale, int use_log,
                double *spectrogram, double *mel, double *logmel, double *mfcc, int n_threads);

/* np.clip(x*scale, lo, hi) then np.round (half to even) -> int8  (kws_nnom.py:354-361 host mirror of
 * firmware/src/app.c:675-695). in: [n][stride] doubles, first n_coef of each row are used. */
void oracle_net_input(const double *mfcc, int64_t n_rows, int stride, int n_coef, double scale,
                      double clip_lo, double clip_hi, int8_t *out);

/* ---- MFCC variant C: the firmware's Q15 pipeline (mfcc_q15_ref.c; tables regenerated, see its header) ---- */

typedef struct oracle_q15_tables oracle_q15_tables_t;

/* tw_mode / rc_mode: float -> Q15 conversion of the CMSIS twiddle / real-FFT split tables (0 floor, 1 round).
 * The mel tables follow audio/edison/mfcc/mfcc_on_mcu.py:26-145. num_mel_bins must be 32. */
oracle_q15_tables_t *oracle_q15_tables_new(int tw_mode, int rc_mode, int num_mel_bins, double sample_rate,
                                           double lower_edge_hertz, double upper_edge_hertz, int mel_mtx_scale);
void oracle_q15_tables_free(oracle_q15_tables_t *t);
/* copies out (each pointer may be NULL): tw1024[1536], tw16[24], rfa[32], rfb[32], mel_coef[<=2048],
 * mel_start[32], mel_count[32]; returns the number of compact mel coefficients */
int oracle_q15_tables_get(const oracle_q15_tables_t *t, int16_t *tw1024, int16_t *tw16, int16_t *rfa, int16_t *rfb,
                          int16_t *mel_coef, int16_t *mel_start, int16_t *mel_count);
/* audioCalcMFCCs per frame (firmware/src/audioprocessing.c:116-215). Outputs (fft/spec/mel may be NULL):
 * fft [n][2048] (re,im of X[0..1023] in natural order), spec [n][513], mel [n][32], mfcc [n][32], all int16. */
int oracle_mfcc_q15(const oracle_q15_tables_t *t, const int16_t *x, int64_t n_frames, int64_t frame_step,
                    int mel_mtx_scale, int16_t *fft, int16_t *spec, int16_t *mel, int16_t *mfcc, int n_threads);
/* mfccToNetInput, NNoM branch (firmware/src/app.c:686-694) */
void oracle_net_input_q15(const int16_t *mfcc, int64_t n_rows, int stride, int n_coef, int scale, int clip_lo,
                          int clip_hi, int8_t *out);

/* ---- MFCC variant D: the firmware's float32 ML-KWS extractor (mfcc_f32_ref.c; pinned on the reference's object code, see its header) ---- */

typedef struct oracle_f32_mfcc oracle_f32_mfcc_t;
/* mfcc_create (firmware/src/audio/mfcc.c:47-84); NULL on bad arguments */
oracle_f32_mfcc_t *oracle_f32_mfcc_new(int num_mfcc_features, int feature_offset, int frame_len, int mfcc_dec_bits,
                                       float preempha);
void oracle_f32_mfcc_free(oracle_f32_mfcc_t *m);
int oracle_f32_mfcc_n_out(const oracle_f32_mfcc_t *m);
int oracle_f32_mfcc_tables_get(const oracle_f32_mfcc_t *m, float *dct, int32_t *first, int32_t *last, float *weights, int cap);
/* mfcc_compute on n_frames frames starting every frame_step samples: out int8 [n][n_out]; out_f32 (may be NULL) the
 * scaled sums before round/saturate; logmel (may be NULL) [n][26] */
int oracle_f32_mfcc_run(const oracle_f32_mfcc_t *m, const int16_t *x, int64_t n_frames, int64_t frame_step, int8_t *out,
                        float *out_f32, float *logmel, int n_threads);

/* ---- int8 CNN (NNoM/CMSIS-NN arithmetic) -------------------------------------------------------- */

#define ORACLE_L_CONV 1
#define ORACLE_L_POOL 2
#define ORACLE_L_DENSE 3
#define ORACLE_L_SOFTMAX 4

typedef struct {
	int32_t type;
	int32_t out_ch, kh, kw, sh, sw;  /* conv / pool geometry (VALID padding)          */
	int32_t bias_lshift, out_rshift; /* conv / dense requantisation                   */
	int32_t relu;                    /* conv: in-place ReLU tail activation           */
	const int8_t *w;                 /* conv: OHWI; dense: row-major [out][in]        */
	const int8_t *b;
} oracle_layer_t;

/*
 * Run n utterances through the layer list. in: [n][in_h*in_w*in_c] int8 HWC.
 * acts (may be NULL): every layer's output back to back per utterance, stride acts_stride bytes.
 * logits = output of the last dense layer, softmax = output of the softmax layer, argmax = first-max
 * index over the softmax output (nnom_predict, nnom_utils.c:275-284). Returns 0 or a negative code.
 */
int oracle_cnn_run(const oracle_layer_t *layers, int n_layers, int in_h, int in_w, int in_c,
                   const int8_t *in, int64_t n, int8_t *acts, int64_t acts_stride,
                   int8_t *logits, int8_t *softmax, int32_t *argmax, int n_threads);

/* app.c:332-356 for n consecutive inferences: state[10] (netOutFilt) is read and updated; filt [n][10], likely [n],
 * spotted [n] may be NULL. */
void oracle_output_filter(const int8_t *soft, int64_t n, double alpha, double threshold, float *state, float *filt,
                          int32_t *likely, int32_t *spotted);

int oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
