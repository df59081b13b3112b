/*
 * edison_internal.h -- shared between the host-side C (tables.c, model.c, legacy.c) and the HIP shim
 * (edison_hip.hip, mfcc_kernels.hip, cnn_kernels.hip). Not part of the public C-ABI (include/edison_hip.h).
 */
#ifndef EDISON_INTERNAL_H
#define EDISON_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ MFCC device tables (one per variant) */
#define ED_MEL_NLO_MAX 3  /* spectrum quads per lane for the narrow band of its pair (shipped filterbank: 2)  */
#define ED_MEL_NHI_MAX 6  /* ... and for the wide band (shipped: 5; the widest band spans 19 quads / 4 lanes) */
#define ED_MEL_TQ_MAX (ED_MEL_NLO_MAX + ED_MEL_NHI_MAX)
#define ED_SPEC_QUADS 129 /* spectrum buffer = 513 bins padded to 516 floats                                  */
/* Which spectrum index a lane ends up with: lane l holds Z[ED_K0(l) + 64 r] in register r after the last pass (natural
 * order: the FFT's second digit transpose goes through LDS; an all-register FFT would leave a digit-reversed order). */
#define ED_K0(l) (l)

typedef struct {
	/* Per-lane register constants, lane-minor so that a wavefront loads each of them with one coalesced read.
	 * 512-point complex FFT of the packed real frame, 3 radix-8 passes (see mfcc_kernels.hip):               */
	float tw1[8][64][2]; /* [p][lane]   W512^(lane*p), p = 0..7 (p = 0 unused)                               */
	float tw2[8][64][2]; /* [q][lane]   W64^((lane&7)*q)                                                     */
	/* mel filterbank, balanced: lane (b = lane&15, r = lane>>4) owns quarter r of TWO bands -- the narrow band
	 * b and the wide band 31-b -- and reads the spectrum as 16-byte quads: mel_slo4[lane] + t, t < mel_NLO,
	 * against mel_w4[t][lane], then mel_shi4[lane] + t, t < mel_NHI, against mel_w4[mel_NLO + t][lane].
	 * The four quarters of a band are summed across the lane rows (xor 16, xor 32).                          */
	int32_t mel_slo4[64], mel_shi4[64];
	/* Which band pair a column serves and which 16-byte half of a quad a lane reads first are free choices; tables.c
	 * makes them so that the fast kernel's LDS reads spread over the banks. mel_band[lane] = the narrow band b of
	 * the lane's column (its wide band is 31-b); mel_half[lane] = 0/1.                                           */
	int32_t mel_band[64], mel_half[64];
	/* ---- the block below is copied verbatim into LDS by every workgroup ---- */
	/* DCT-II with the variant's normalisation folded in, using D[31-n][c] = (-1)^c D[n][c]:
	 * y[c] = sum_{n<16} D[n][c] * (L[n] + (-1)^c L[31-n]); lane (c = lane&31, h = lane>>5) sums n = 8h..8h+7,
	 * dct4[n4][lane][j] = scale * 2*cos(pi*c*(2*(8h+4*n4+j)+1)/64)                                           */
	float dct4[2][64][4];
	float twp[4][64][2]; /* [m][lane]   W1024^(k0+64m), k0 = (lane>>3) + 8*(lane&7): real-FFT split twiddles    */
	float mel_w4[ED_MEL_TQ_MAX][64][4];
	/* ---- end of the LDS image ---- */
	int32_t mel_NLO, mel_NHI;
	float spec_scale; /* applied to 2|X[k]|: A 0.5, B 0.5/1024/sqrt(2)                                       */
	float log_offset; /* 1e-6                                                                                */
	int32_t always_log; /* variant A                                                                         */
	int32_t has_window; /* variant TF: the frame is multiplied by window2 before the FFT                     */
	/* (w[2m], w[2m+1]), m = 0..511: the window as the packed frame meets it, one float2 per complex input point.
	 * Read by the one-frame kernel's WINDOW instances only (all zero for A and B). */
	float window2[512][2];
} ed_mfcc_tables_t;

/* mfcc_utils.gen_mel_weight_matrix in float64 (W[nbins][nmel]); returns 0 or a negative EDISON_E_* code.   */
int ed_gen_mel_weight_matrix(int num_mel_bins, int num_spectrogram_bins, double sample_rate,
                             double lower_edge_hertz, double upper_edge_hertz, double *W);
/* Build the device tables of one variant (EDISON_MFCC_A, _B or _TF). */
int ed_build_mfcc_tables(int variant, double sample_rate, double lower_edge_hertz, double upper_edge_hertz,
                         double mel_mtx_scale, ed_mfcc_tables_t *out, char *err, size_t err_cap);

/* ------------------------------------------------------------------ MFCC variant C (firmware Q15) tables */
/* The compact mel matrix (mel_constants.h: melMtxCompact[915] + per-band first bin and count) is spread over the
 * wavefront like the float kernel's filterbank: lane (b = lane&15, r = lane>>4) sums quarter r of the narrow band b
 * (mel_nlo taps) and quarter r of the wide band 31-b (mel_nhi taps); 32-bit wrap-around addition is associative, so
 * the split does not change the firmware's result. The kernel is compiled for two shapes: 6+18 (shipped) and 8+24. */
#define ED_Q15_NLO_MAX 8
#define ED_Q15_NHI_MAX 24
#define ED_Q15_TAPS_MAX (ED_Q15_NLO_MAX + ED_Q15_NHI_MAX)
#define ED_Q15_PAIRS(n) (((n) + 2) / 2)
#define ED_Q15_PAIRS_MAX (ED_Q15_PAIRS(ED_Q15_NLO_MAX) + ED_Q15_PAIRS(ED_Q15_NHI_MAX))

typedef struct {
	/* complex Q15 coefficients as the two packed operands of v_dot2_i32_i16 (see tables_q15.c):
	 * tw*[i] = (cos, sin), tw*x[i] = (-sin, cos) of 2 pi i / N                                                */
	uint32_t tw1024[768], tw1024x[768];
	uint32_t tw16[12], tw16x[12];
	uint32_t rfa[16], rfb[16]; /* real-FFT split of the DCT stage, pair 256*i of realCoefA/BQ15: (A.re,-A.im), (B.re,B.im) */
	int32_t mel_tap[ED_Q15_TAPS_MAX][64]; /* [t][lane]: t < mel_nlo narrow-band taps, then mel_nhi wide-band taps (0 = padding) */
	int32_t mel_lo_bin[64], mel_hi_bin[64]; /* first spectrum bin of the lane's two runs (every read stays below 513) */
	/* the same taps as the kernel reads them: the spectrum sits in LDS as int16, a lane reads it as dwords (bins 2p,
	 * 2p+1) from pair mel_*_pair[lane] on and multiplies with v_dot2_i32_i16 against (tap[2p] | tap[2p+1] << 16);
	 * (n+2)/2 pairs cover n taps from an odd or even first bin, the taps outside the run are 0 */
	uint32_t mel_tap2[ED_Q15_PAIRS_MAX][64];
	int32_t mel_lo_pair[64], mel_hi_pair[64];
	int32_t mel_nlo, mel_nhi;
	int32_t mel_scale, n_mel_coef;
	int32_t need_nyquist; /* some band reads spectrum bin 512 */
	int32_t pad_;
	/* bit c (c < 32768): arm_sqrt_q31(2 c^2) >> 16 is c - 1 instead of c (the kernel's magnitude shortcut,
	 * tools/verify/sqrt_q31_floor.c) */
	uint32_t sqbit[1024];
} ed_q15_tables_t;

int ed_build_q15_tables(double sample_rate, double lower_edge_hertz, double upper_edge_hertz, double mel_mtx_scale,
                        ed_q15_tables_t *out, char *err, size_t err_cap);

typedef struct {
	const int16_t *audio;
	int64_t n_frames;
	int64_t frames_per_group; /* as ed_mfcc_args_t */
	int64_t group_stride;
	int64_t frame_step;
	int n_coef;
	int16_t *mfcc_i16; /* [n_frames][n_coef] or NULL: bufDctInline                                              */
	float *mfcc_f32;   /* [n_frames][n_coef] or NULL: the same numbers as float (Cube branch of mfccToNetInput)   */
	int8_t *feat;      /* [n_frames][n_coef] or NULL: NNoM branch of mfccToNetInput (app.c:686-694), scale 1      */
	/* stage dumps (diagnostic kernel only): what audioDumpToHost sends (audioprocessing.c:221-231)              */
	int16_t *fft;  /* [n][513][2] */
	int16_t *spec; /* [n][513]    */
	int16_t *mel;  /* [n][32]     */
} ed_mfcc_q15_args_t;

/* ------------------------------------------------------------------ MFCC variant D (firmware float32 extractor) */
#define ED_F32_NUM_FBANK 26   /* NUM_FBANK_BINS, firmware/src/audio/mfcc.h:30 */
#define ED_F32_MAX_FRAME 1024 /* padded frame length supported on this path (the firmware uses 512, app.c:497) */
#define ED_F32_MAX_W 1100     /* mel weights of all bands back to back (bands overlap: < 2 * bins)               */

typedef struct {
	int32_t n_features, offset, frame_len, padded, log2p, dec_bits;
	float preempha, scale;                 /* scale = 2^dec_bits */
	float window[ED_F32_MAX_FRAME];        /* Hann, float (mfcc.c:66-68) */
	float tw[ED_F32_MAX_FRAME / 2][2];     /* exp(-2 pi i k / padded) */
	int32_t mel_first[ED_F32_NUM_FBANK], mel_last[ED_F32_NUM_FBANK], mel_off[ED_F32_NUM_FBANK];
	float mel_w[ED_F32_MAX_W];
	float dct[ED_F32_NUM_FBANK * ED_F32_NUM_FBANK]; /* [feature][band], rows < n_features used */
} ed_f32_tables_t;

int ed_build_f32_tables(int num_mfcc_features, int feature_offset, int frame_len, int mfcc_dec_bits, float preempha,
                        ed_f32_tables_t *out, char *err, size_t err_cap);

typedef struct {
	const int16_t *audio;
	int64_t n_frames, frame_step;
	int8_t *out;    /* [n][n_features - offset] q7 */
	float *out_f32; /* same shape, before round / saturate, or NULL */
	float *logmel;  /* [n][26] or NULL */
} ed_mfcc_f32_args_t;

/* ------------------------------------------------------------------ int8 CNN model (kws_conv topology)   */
/* Geometry of the one network this path accelerates (weights.h:138-161).                    */
#define ED_IN_H 31
#define ED_IN_W 13
#define ED_C1_O 16  /* conv1 5x5x1  -> 27x9x16, pooled 13x9x16  */
#define ED_C2_O 32  /* conv2 3x3x16 -> 11x7x32, pooled 5x7x32   */
#define ED_C3_O 64  /* conv3 3x3x32 -> 3x5x64                   */
#define ED_C4_O 32  /* conv4 3x3x64 -> 1x3x32                   */
#define ED_FC_I 96
#define ED_FC_O 10
#define ED_CNN_ACT_BYTES 10420 /* == EDISON_CNN_ACT_BYTES */

typedef struct {
	/* weights re-laid for conflict-free LDS reads: dword (k4, o) = 4 consecutive k of output channel o,
	 * k = (ky*KW + kx)*Cin + ci in OHWI order; conv1's K = 25 is zero-padded to 28                           */
	int32_t w1[7][ED_C1_O];
	int32_t w2[36][ED_C2_O];
	int32_t w3[72][ED_C3_O];
	int32_t w4[144][ED_C4_O];
	int32_t wfc[24][16];      /* dense, rows 10..15 zero */
	/* (bias << bias_lshift) + NN_ROUND(out_rshift), ready to seed the accumulator */
	int32_t b1[ED_C1_O], b2[ED_C2_O], b3[ED_C3_O], b4[ED_C4_O], bfc[16];
	int32_t rs1, rs2, rs3, rs4, rsfc; /* output right shifts */
	int32_t pad_[3];
} ed_cnn_model_t;

/*
 * The same parameters laid out for v_mfma_i32_32x32x32_i8 (fast path, cnn_mfma_kernels.hip). Every layer is the
 * GEMM D[out_channel][pixel] = sum_k A[out_channel][k] * B[k][pixel]; A (weights) is stored as ready-made
 * operand fragments: fragment = 64 lanes x 16 bytes, lane l holds A[row = tile_row(l&31)][k = 32*kstep + 16*(l>>5) + j],
 * j = 0..15, so a wavefront fetches one fragment with a single conflict-free ds_read_b128; tile_row (model.c) orders the 32 rows of
 * a tile so that a lane's accumulators are 16 consecutive output channels (the seeds b1 .. b3 are stored in that physical order).
 *   a1  conv1 as a Toeplitz GEMM: row = x*16 + o (9 x 16 = 144 rows -> 5 row tiles), k = ky*16 + xx over five
 *       16-byte-padded input rows (K = 80 -> 3 k-steps); A = w1[o][ky][xx - x] inside the 5-tap window, else 0
 *   a2  conv2: 32 rows, k = tap*16 + ci (K = 144 -> 5 k-steps, the last half zero)
 *   a3  conv3: 64 rows (2 row tiles), k = tap*32 + ci (9 k-steps)
 *   a4  conv4: 32 rows, k = tap*64 + ci, as fragments of v_mfma_i32_16x16x64_i8 (lane l: row l & 15, k chunk l >> 4):
 *       2 row tiles x 9 k-steps (one tap each) -- the layer has 12 columns per wave, a 32-column tile would be 5/8 padding
 *   afc dense: 10 rows padded to 16, k = 96 padded to 128, the same 16 x 16 x 64 fragments (2 k-steps)
 */
/* LDS layout of ONE utterance inside a wavefront's slice (cnn_mfma_kernels.hip), shared with model.c, which tabulates addresses: */
/* Region A: in' = the 31 input rows padded to 16 bytes, EVEN rows then ODD rows (2 x 16 slots of 16 B = 512)  ->
 *           p2 = conv2's pooled output [5][7] pixels as TWO PLANES of 16 channels (2 x 35 slots = 1120)  ->  c4 [3][32] (96)
 * Region B: p1 [13][9][16] (1872)  ->  c3 = conv3's output [3][5] pixels as FOUR PLANES of 16 channels (4 x 15 slots = 960)
 * Planes instead of pixel-major records: a B operand is 16 bytes (16 channels of one tap) per lane, and the lanes of a
 * ds_read_b128 group are neighbouring columns = neighbouring pixels. With [pixel][32 or 64 channels] records they sat 32 / 64
 * bytes apart and used every second / fourth 16-byte bank group: 2-way (conv1's row pairs, conv3) and 4-way (conv4) bank
 * conflicts on every operand read, 39 % of all LDS cycles (profiles/r02_cnn_counters.txt). In a plane neighbouring pixels
 * are neighbouring 16-byte slots. */
#define EDM_REGA 1120
#define EDM_REGB 1872
#define EDM_IN_ODD 256   /* byte offset of the odd input rows inside region A */
#define EDM_P2_PLANE 560 /* bytes per 16-channel plane of p2 */
#define EDM_C3_PLANE 240 /* bytes per 16-channel plane of c3 */
#define EDM_UTT (EDM_REGA + EDM_REGB)
/* Which column of a layer's GEMM a lane computes is free (every lane derives its own addresses); for a FULL group of four
 * utterances the columns are dealt to the (tile, lane) slots so that the 16 lanes of a ds_read_b128 group read 16 different
 * 16-byte bank groups and the 8 lanes of a ds_write_b128 group write 8 different ones (csrc/cnn_mfma_cols.h, chosen and checked by
 * tools/dev/cnn_lds_model.py: natural order 457 extra LDS cycles per group = 38 % of all LDS cycles, profiles/r04_cnn_lds_by_phase.txt).
 * A table entry = byte offsets inside the wave's slice: bits 0-14 where the column's operand reads start, bits 16-30 where its
 * outputs go, bit 31 = idle lane (reads the tile's last live column again, stores to the dummy slot). */
#define ED_CNN_COL_IDLE 0x80000000u
typedef struct {
	int8_t a1[15][1024]; /* [row_tile*3 + kstep] */
	int8_t a2[5][1024];
	int8_t a3[18][1024]; /* [row_tile*9 + kstep] */
	int8_t a4[18][1024]; /* [row_tile*9 + tap], 16 x 16 x 64 fragments */
	int8_t afc[2][1024]; /* 16 x 16 x 64 fragments */
	int32_t b1[ED_C1_O], b2[ED_C2_O], b3[ED_C3_O], b4[ED_C4_O], bfc[16]; /* accumulator seeds as above */
	int32_t rs1, rs2, rs3, rs4, rsfc;
	int32_t pad_[3];
	uint32_t cols1[2][32], cols2[5][32], cols3[2][32]; /* column tables of conv1 / conv2 / conv3, see above */
} ed_cnn_mfma_model_t;

int ed_parse_model(const void *blob, size_t blob_bytes, ed_cnn_model_t *out, ed_cnn_mfma_model_t *out_mfma, char *err,
                   size_t err_cap);

/* ------------------------------------------------------------------ any sequential NNoM int8 graph (cnn_net_kernels.hip)
 * The plan of the layer-by-layer kernel: what model_compile() (nnom.c:758-900) works out for the chain
 * Input -> {Conv2D [+ReLU] | MaxPool | Dense [+ReLU] | Flatten | Softmax}* -> Output. One workgroup owns an
 * utterance; activations ping-pong between two LDS buffers; weights stay OHWI int8 in HBM/L2. */
#define ED_NET_MAX_LAYERS 32
#define ED_NET_MAX_LDS (64 * 1024 - 256) /* both activation buffers */
enum { ED_NET_CONV = 1, ED_NET_POOL = 2, ED_NET_DENSE = 3, ED_NET_SOFTMAX = 4 };

typedef struct {
	int32_t type, relu;
	int32_t in_h, in_w, in_c, out_h, out_w, out_c;
	int32_t kh, kw, sh, sw, pad_h, pad_w;
	int32_t rs;       /* output right shift (conv, dense)                                                     */
	int32_t w_off;    /* byte offset of the layer's weights in the device weight buffer (16-byte aligned)      */
	int32_t seed_off; /* index of its first accumulator seed (bias << bias_lshift) + NN_ROUND(out_rshift)      */
	int32_t in_buf, out_buf; /* LDS byte offsets                                                             */
	int32_t in_n, out_n;     /* elements                                                                     */
	int32_t acts_off; /* offset of the layer's output inside one utterance's activation dump                  */
	int32_t check_taps; /* some window reaches outside the image: padding, or SAME with an even kernel / stride (the
	                     * output is ceil(in/stride) wide, so the last windows overhang the right / bottom edge)   */
	int32_t pad_;
} ed_net_layer_t;

typedef struct {
	int32_t n_layers;
	int32_t in_h, in_w, in_c, in_n;
	int32_t out_n;        /* width of logits / softmax                                                        */
	int32_t logits_layer; /* the layer whose output is reported as logits: the one before a final Softmax      */
	int32_t has_softmax;
	int32_t lds_bytes, acts_bytes, weights_bytes, n_seeds;
	ed_net_layer_t L[ED_NET_MAX_LAYERS];
} ed_net_plan_t;

/* Builds the plan, the device weight image and the seeds from an .ednn blob; *weights / *seeds are malloc'd.
 * EDISON_E_SIZE for a malformed graph, EDISON_E_NO_IMPL for one the reference accepts but this path does not run. */
int ed_plan_net(const void *blob, size_t blob_bytes, ed_net_plan_t *plan, int8_t **weights, int32_t **seeds, char *err,
                size_t err_cap);

/* ------------------------------------------------------------------ the same graphs on the matrix cores (cnn_net_mfma_kernels.hip)
 * Every Conv2D / Dense layer as the implicit GEMM  D[out_channel][pixel] = sum_k A[out_channel][k] * B[k][pixel]  on
 * v_mfma_i32_32x32x32_i8. k runs over (kernel row ky, 16-byte chunk j of that row's CONTIGUOUS input segment): in HWC
 * memory the kw * C_in bytes under one kernel row are adjacent, so a B fragment is one aligned ds_read_b128
 *   - straight from the (zero-padded) input image when C_in % 16 == 0, or
 *   - from an expanded copy with one 16-byte-aligned record per (input row, output x) otherwise (first layers, RGB).
 * A = the layer's weights, packed by the planner as ready operand fragments ([row tile][k-step][64 lanes][16 B], zeros
 * beyond kw * C_in, beyond the last chunk and beyond C_out); C = the accumulator seeds. MaxPool, Softmax and the argmax
 * stay on the VALU. A WAVEFRONT takes `batch` inputs through the whole layer list by itself, in its own slice of LDS: no
 * workgroup barrier (the first version ran a workgroup in lockstep phases and spent a third of its time in barriers). */
#define ED_MM_MAX_KOFF 1024
#define ED_MM_MAX_COLS 1024 /* column-table entries of all layers together (8 bytes each in LDS) */
#define ED_MM_MAX_XTAB 1536 /* expansion-table entries of all layers together (8 bytes each in LDS) */
#define ED_MM_MAX_INTAB 4096 /* input-table entries (2 bytes each in LDS) */
/* images of 4 .. ED_MM_INTAB_PAD bytes take the kernel's prefetching input stage (two dwords per lane): their table is padded to
 * ED_MM_INTAB_PAD entries that all point at the LAST byte of the image's 16 bytes of slack -- the stage then scatters every byte of
 * both dwords without a test (a byte past the image goes to the slack, which is only ever read against zero weights or masked) */
#define ED_MM_INTAB_PAD 512
typedef struct {
	int32_t mm;                 /* 1: Conv2D / Dense on the matrix cores                                              */
	int32_t in_hp, in_wp;       /* the input image as this layer wants it in LDS: padded height / width ...          */
	int32_t in_py, in_px;       /* ... and where the unpadded image starts inside it                                  */
	int32_t in_img;             /* bytes per input image in that layout (multiple of 16, + 16 bytes of slack)         */
	int32_t expand;             /* 1: B comes from the expanded copy                                                  */
	int32_t x_img;              /* bytes per expanded image (0 when direct)                                           */
	int32_t cpr, n_ks, n_rt;    /* 16-byte chunks per kernel row, k-steps = ceil(kh * cpr / 2), row tiles             */
	int32_t pitch_x, pitch_y;   /* B addressing in bytes: per output x, per input row                                 */
	int32_t frag_off;           /* byte offset of the fragments in the fragment buffer                                */
	int32_t seed_off;           /* index of the 32 * n_rt padded seeds                                                */
	int32_t koff_off;           /* index into koff[]: byte offset of chunk c = 2 s + h inside the image (ky * pitch_y + 16 j) */
	int32_t pool_h, pool_w;     /* > 0: the MaxPool that follows (window = stride, no padding) is taken in this layer's
	                             * epilogue -- maximum of the accumulator tiles of the window, then ONE requantisation
	                             * (exact: the requantisation is monotone) -- and the pool layer itself is skipped       */
	int32_t skip;               /* 1: this MaxPool is fused into the layer in front of it                              */
	int32_t col_off;            /* >= 0: index of this layer's column table in coltab[] (pairs), -1: the kernel divides    */
	int32_t xtab_off;           /* >= 0: index of this layer's expansion table in xtab[] (pairs), -1: the kernel divides   */
	int32_t small;              /* 1: at most 16 columns per wave and nothing fused: v_mfma_i32_16x16x64_i8 tiles (16 rows x
	                             * 16 columns, four 16-byte chunks per k-step); n_ks16 / n_rt16 count those                */
	int32_t n_ks16, n_rt16;
	int32_t pp;                 /* bytes from one input pixel to the next in this layer's LDS layout: C_in, or C_in + 16 where that
	                             * takes the B reads (16 bytes per lane, neighbouring lanes = neighbouring pixels) and the
	                             * producer's dword stores off the same LDS banks (C_in = 32, 64, ...: model_net_mm.c)          */
	int32_t toep;               /* 1: row-Toeplitz form (model_net_mm.c): the GEMM's rows are (output x, output channel), k runs over
	                             * (kernel row, byte of the whole padded input row), one column per output row; no expansion     */
} ed_mm_layer_t;

/* What one pass of the kernel's layer loop needs, worked out on the host: one 128-byte record per layer that the wave
 * reads with two scalar loads (the kernel used to derive it from ed_net_layer_t + ed_mm_layer_t + the consumer's record:
 * a chain of dependent scalar loads and ~100 scalar instructions per layer and input). */
enum { ED_RUN_SKIP = 0, ED_RUN_MM = 1, ED_RUN_POOL4 = 2, ED_RUN_POOL1 = 3, ED_RUN_SOFTMAX = 4 };
/* ed_mm_run_t.rs: the planner found that sat8(v >> rs) may be taken as the high byte of sat16(v >> (rs - 8)) -- always for
 * rs >= 8, for rs < 8 when no accumulator of the layer can leave 32 bits under the left shift (model_net_mm.c) */
#define ED_RUN_RS_MASK 0xff
#define ED_RUN_RS_HI 0x100
typedef struct {
	int32_t kind;               /* ED_RUN_*                                                                            */
	int32_t zero_border;        /* 1: the consumer wants a zero border: clear the output images first                  */
	int32_t in_img, o_img;      /* bytes per image: this layer's input layout, the layout it stores into               */
	int32_t o_origin, o_row, oc_pitch; /* output pixel (y, x) goes to o_origin + y * o_row + x * oc_pitch              */
	int32_t li_out;             /* the layer whose output this pass stores (the fused MaxPool's when there is one)      */
	int32_t expand, x_img, rec_per_img, xtab_off; /* expansion: bytes per expanded image, records per image, table      */
	int32_t pitch_x, pitch_y, sh, ph, pw; /* B addressing; rows per output row; fused pooling window (1x1: none)        */
	int32_t n_ks, n_rt, frag_off, seed_off, koff_off, col_off;
	int32_t pix_per_img, col_w; /* stored pixels per image, per row                                                    */
	int32_t out_c, rs, lo_clamp; /* rs: the output shift in the low byte (ED_RUN_RS_MASK) | ED_RUN_RS_HI                  */
	int32_t in_n;               /* Softmax: classes                                                                    */
	int32_t small;              /* 1: the 16 x 16 x 64 tiles (n_ks / n_rt then count those)                            */
	int32_t in_off, o_off;      /* where the layer's input / output images start inside the wave's activation region: the
	                             * planner places them at opposite ends, so the region is max(in + out) over the layers, not
	                             * twice the largest image                                                                  */
} ed_mm_run_t;
#if defined(__cplusplus)
static_assert(sizeof(ed_mm_run_t) == 128, "ed_mm_run_t: one 128-byte record per layer (two s_load_dwordx16)");
#else
_Static_assert(sizeof(ed_mm_run_t) == 128, "ed_mm_run_t: one 128-byte record per layer (two s_load_dwordx16)");
#endif

typedef struct {
	int32_t ok;                 /* 0: this graph stays on the layer-by-layer kernel (why: the loader's error text)     */
	int32_t batch;              /* inputs a WAVEFRONT takes through the layer list at a time                          */
	int32_t waves;              /* wavefronts per workgroup: each works alone in its own LDS slice (no workgroup barrier) */
	int32_t buf_bytes;          /* HALF of a wave's activation region (the two ends the layers ping-pong between)      */
	int32_t x_bytes;            /* a wave's expansion buffer                                                          */
	int32_t lds_bytes;
	int32_t frag_lds;           /* bytes of LDS reserved for weight fragments                                         */
	int32_t frag_mode;          /* 2: all layers resident in LDS, 0: streamed from L2                                  */
	int32_t tbl_bytes;          /* LDS copy of the small tables: koff | seeds | column tables (read once per workgroup) */
	int32_t frag_bytes, n_seeds, n_koff;
	int32_t n_cols;             /* entries of coltab[] in use                                                          */
	int32_t n_xtab;             /* entries of xtab[] in use                                                            */
	int32_t n_intab;            /* entries of intab[] in use: in_n, or 0 (the kernel divides)                          */
	int32_t pad_;
	ed_mm_layer_t L[ED_NET_MAX_LAYERS];
	ed_mm_run_t R[ED_NET_MAX_LAYERS];
	int32_t koff[ED_MM_MAX_KOFF];
	/* per stored pixel (y, x) of a matrix-core layer, in pixel order: byte offset of its first window's first chunk in
	 * the layer's B source, and of its output pixel in the consumer's layout (what the kernel would otherwise work out
	 * with two divisions per 32-column tile) */
	int32_t coltab[2 * ED_MM_MAX_COLS];
	/* per expansion record of a layer (one per input row, output x and 16-byte chunk): source byte offset inside the
	 * input image | bytes of the chunk that belong to the kernel-row segment << 24, and the destination byte offset
	 * inside the expanded image */
	int32_t xtab[2 * ED_MM_MAX_XTAB];
	/* byte offset of input element e (HWC order) inside layer 0's LDS layout */
	uint16_t intab[ED_MM_MAX_INTAB];
} ed_mm_plan_t;

/* Adds the matrix-core plan to a graph ed_plan_net accepted. *frag / *seeds are malloc'd when mm->ok. */
int ed_plan_net_mm(const void *blob, size_t blob_bytes, const ed_net_plan_t *plan, ed_mm_plan_t *mm, int8_t **frag,
                   int32_t **seeds);

/* The text that specialises ed_net_mfma_kernel for one graph (net_spec.c): returns the bytes it needs without the final 0. */
size_t ed_emit_net_spec(const ed_net_plan_t *P, const ed_mm_plan_t *M, char *out, size_t cap);
uint64_t ed_net_spec_hash(const ed_net_plan_t *P, const ed_mm_plan_t *M);

/* The firmware's output post-processing (app.c:332-356) for ONE inference, done by the one-launch microphone kernel itself
 * (ed_kws1_kernel) behind the softmax: moving average in double arithmetic rounded to float, first maximum, threshold. */
typedef struct {
	double alpha, one_minus_alpha, threshold;
	float *state;     /* [10] netOutFilt, device memory: read and written */
	float *filt;      /* [10] out                                          */
	int32_t *likely;  /* [1]  out                                          */
	int32_t *spotted; /* [1]  out: the class, or -1 below the threshold    */
	/* the state machine behind it (edison_fsm_core.h), or fsm = NULL */
	void *fsm;            /* edison_fsm, device memory: read and written   */
	int32_t *fsm_state;   /* [1] out: the state after this inference       */
	void *fsm_copy;       /* edison_fsm out (the host's view), or NULL     */
	uint32_t dt_us;
	int32_t wake_idx;
	uint32_t loc_mask, val_mask;
} ed_out_filter_t;

/* ------------------------------------------------------------------ kernel launchers (HIP side)           */
typedef struct {
	const int16_t *audio;
	int64_t n_frames;
	int64_t frames_per_group; /* frame f starts at (f / fpg) * group_stride + (f % fpg) * frame_step        */
	int64_t group_stride;
	int64_t frame_step;
	int n_coef;
	int use_log;
	int mel_NLO, mel_NHI;     /* host copies of the variant's ed_mfcc_tables_t fields (size the LDS tables) */
	float *mfcc;      /* [n_frames][n_coef] or NULL */
	int8_t *feat;     /* [n_frames][n_coef] or NULL */
	float feat_scale;
	int window;       /* the variant windows its frames (ed_mfcc_tables_t.window2): one-frame kernel's WINDOW instances. (Sits
	                   * in what was alignment padding: the kernel arguments of every other instance keep their offsets.) */
	/* stage dumps (diagnostic kernel only) */
	float *fft, *spec, *mel, *logmel;
} ed_mfcc_args_t;

/* Batches of a list launch (edison_mfcc_batches_dev): up to ED_MFCC_LIST_MAX independent batches of the same frame count, each at its
 * own address with its own outputs, as the groups of ONE launch -- group g's samples start at audio[g], its rows at mfcc[g] / feat[g].
 * Passed by value as the list kernel's third argument (kernarg memory: a uniform index is one scalar load). */
#define ED_MFCC_LIST_MAX 16
typedef struct {
	const int16_t *audio[ED_MFCC_LIST_MAX];
	float *mfcc[ED_MFCC_LIST_MAX];
	int8_t *feat[ED_MFCC_LIST_MAX];
} ed_mfcc_list_t;

/* the generality path (mfcc_generic_kernels.hip): variants A / B / TF for any frame length / number of mel bins, float64 */
#define ED_GEN_MAX_FRAME 4096
#define ED_GEN_MAX_MEL 256
typedef struct {
	const int16_t *audio;
	int64_t n_frames, frame_step;
	int frame_len, n_bins, fft_out, n_mel; /* n_bins: spectrum bins under the mel matrix (A: N/2, B and TF: N/2+1); fft_out: entries of the fft / spectrogram dumps (A: N/2, B: N, TF: N/2+1) */
	int take_log, n_coef;
	double fft_scale, spec_scale, mel_div, dct_div;
	const double *tw;  /* [N][2]  cos, sin of 2 pi j / N (device) */
	const double *W;   /* [n_bins][n_mel], for B already multiplied by mel_mtx_scale (mel_div divides it out again, as the reference does) */
	const double *dct; /* [n_mel][n_mel] 2 cos(pi c (2 n + 1) / (2 n_mel)) */
	double *fft, *spec, *mel, *logmel, *mfcc; /* device, each may be NULL */
	int8_t *feat;      /* [n_frames][n_coef] int8 net input, or NULL */
	float feat_scale;
	const float *window; /* variant TF: [N] periodic Hann window in float32 (tf.signal.hann_window), multiplied into the float32 samples; NULL: none */
} ed_mfcc_gen_args_t;



#ifdef __cplusplus
}
#endif
#endif
