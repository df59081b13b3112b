/*
 * cnn_mfma_kernels.hip -- the int8 keyword-spotting CNN on the gfx950 matrix cores (fast path).
 *
 * Same arithmetic, bit for bit, as the reference's NNoM/CMSIS-NN CPU path (see cnn_kernels.hip for the
 * reference citations per layer); only the order of the exact int32 accumulation differs, which integer
 * addition does not notice. Requantisation is out = relu(ssat8((acc + (bias << bl) + round) >> rs)); max-pool
 * is applied to the int32 accumulators BEFORE requantisation, which is exact because the requantisation is a
 * monotone non-decreasing function of the accumulator.
 *
 * Every layer is the GEMM  D[out_channel][pixel] = sum_k A[out_channel][k] * B[k][pixel]  on
 * v_mfma_i32_32x32x32_i8 (A, B: 16 bytes per lane; lane l: A[row l&31][k 16*(l>>5)+j], B[k 16*(l>>5)+j][col l&31];
 * D: col l&31, row (reg&3) + 8*(reg>>2) + 4*(l>>5) -- verified with exact integer data, tools/ubench/mfma_i8.hip):
 *   - A = weights, pre-packed on the host as operand fragments (ed_cnn_mfma_model_t), staged once per
 *     workgroup in LDS, fetched with one conflict-free ds_read_b128 per MFMA;
 *   - B = activations straight from the HWC int8 buffers in LDS: the 16 bytes a lane needs are 16 consecutive
 *     input channels of one tap (conv2-4) or one 16-byte-padded input row (conv1, Toeplitz form), i.e. one
 *     aligned ds_read_b128 -- no im2col buffer;
 *   - C = the accumulator seeds (bias << bias_lshift) + NN_ROUND(out_rshift) of the tile's rows, so the epilogue is
 *     shift, clamp, pack;
 *   - D puts 4 consecutive rows into 4 consecutive registers of a lane, and which output channel a ROW of a 32-row tile is, is
 *     the host's choice (model.c tile_row): the four packed dwords of a lane are 16 consecutive output channels of its pixel, so
 *     the epilogue stores one 16-byte record directly where the next layer reads -- no exchange between lanes.
 *   - the two rows of a max-pool window are computed as two accumulator tiles over the same lanes
 *     (even / odd input row), pooled by an element-wise max.
 *
 * Work distribution (round 2). A WAVEFRONT owns a group of EDM_G = 4 utterances and takes it through all layers by
 * itself in wave-private LDS: there is NO workgroup barrier in the loop. The round-1 kernel dealt the tiles of a layer
 * to the 8 waves of a workgroup with a barrier between layers; all waves were then in the same phase at the same time
 * (matrix pipe busy 27 % of the time, VALU 35 %, profiles/r02_cnn_phase_ablation_r1_kernel.txt), tile counts 13/35/15/3
 * did not divide the waves and the dense + softmax stage ran on one wave. Independent waves drift apart, so the
 * requantisation epilogue (VALU) of one wave runs under the MFMAs of its SIMD partner. Per group the column tiles are
 * 52 / 140 / 60 / 12 / 4 columns in 2 / 5 / 2 / 1 / 1 tiles of 32 (167 MFMAs per 4 utterances against 133 before: the
 * price of the independence). Logits are parked in LDS and the serial softmax / argmax runs once per 8 groups on 32
 * lanes. Groups are handed out through a counter in LDS (ds_add_rtn_u32) as in the MFCC kernel.
 * LDS: 60 KB of weight fragments, seeds and column tables + 8 waves x (4 x 2992 B activations + 512 B parked logits + 160 B) = 158.5 KB.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>

#include "edison_internal.h"
#include "edison_fsm_core.h"

/* ---- lab knobs: only a lab build (ED_LAB, tools/lab/mkvariant.py) may set them; the product build has none, and
 * tests/test_host_cpu.py checks the values below against what edison_amd/build.py compiles */
#if !defined(ED_LAB) && (defined(EDM_PRIO) || defined(EDM_SKIP) || defined(EDM_W1V) || defined(EDM_W2V) || defined(EDM_W3V))
#error "EDM_* lab knob defined without ED_LAB (tools/lab/mkvariant.py builds lab variants)"
#endif
#if defined(ED_LAB)
/* a lab build says so: the product library exports no ed_lab_build_* symbol (tests/test_host_cpu.py) */
extern "C" { extern const int ed_lab_build_cnn_mfma; const int ed_lab_build_cnn_mfma = 1; }
#endif
/* lab only: timing / counter ablations by layer (results are WRONG when non-zero; tools/lab/cnn_lds_by_phase.sh): 1 conv1, 2 conv2,
 * 4 conv3, 8 conv4, 16 dense */
#ifndef EDM_SKIP
#define EDM_SKIP 0
#endif
/* vector instructions the scheduler is asked to put behind each MFMA of the next tile while a tile is requantised (EDM_WEAVE), for
 * the shift-8 epilogues of conv1 / conv2 and for conv3 */
#ifndef EDM_W1V
#define EDM_W1V 5
#endif
#ifndef EDM_W2V
#define EDM_W2V 4
#endif
#ifndef EDM_W3V
#define EDM_W3V 6
#endif
#define EDM_G 4       /* utterances per wavefront group */
#define EDM_WAVES 8
#define EDM_THREADS (64 * EDM_WAVES)
/* (the LDS layout of an utterance -- EDM_REGA, EDM_REGB, EDM_IN_ODD, EDM_P2_PLANE, EDM_C3_PLANE, EDM_UTT -- is in edison_internal.h:
 * model.c builds the column tables from it) */
#define EDM_PARK 8    /* groups whose logits are parked before one softmax pass (8 x 4 = 32 lanes) */
#define EDM_WAVE_LDS (EDM_G * EDM_UTT + EDM_PARK * EDM_G * 16 + 32 + 128) /* + park_group[8] + the idle lanes' 8 dummy slots */

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

/* Wave priority per phase of a group (see ED2_PRIO in mfcc_kernels.hip), two bits per point: 0 input rows, 1 conv1, 2 conv2,
 * 3 conv3, 4 conv4, 5 dense, 6 softmax / stores. Two waves per SIMD: +0.9 ... +1.1 % only. 0 = none */
#ifndef EDM_PRIO
#define EDM_PRIO 0x3f93
#endif
#if EDM_PRIO
#define EDM_PR(pt) __builtin_amdgcn_s_setprio((EDM_PRIO >> (2 * (pt))) & 3);
#else
#define EDM_PR(pt)
#endif
__device__ __forceinline__ v4i edm_ld16(const unsigned char *p) { return *reinterpret_cast<const v4i *>(p); }

__device__ __forceinline__ int edm_med3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* requantise 4 consecutive accumulators with ReLU and pack them into one HWC dword. The accumulators already hold the
 * seeds (bias << bias_lshift) + NN_ROUND(out_rshift): they are the C operand of each tile's first MFMA. */
__device__ __forceinline__ uint32_t edm_pack_relu(int a0, int a1, int a2, int a3, int rs)
{
	/* relu(ssat8(a >> rs)) = clamp(a >> rs, 0, 127) = clamp(a >> (rs - 1), 0, 255) >> 1 for rs >= 1 (arithmetic shifts compose
	 * and 255 >> 1 = 127), and the clamp to 0..255 of two values at a time is what v_cvt_pk_i16_i32 (saturating to int16:
	 * harmless in front of a tighter clamp) followed by v_sat_pk_u8_i16 does: 4 shifts + 2 + 2 + one merge + shift and mask
	 * of the whole dword = 11 instructions, 7 of them in the cheap 4-byte encodings, against 4 shifts + 4 v_med3 + 3 merges
	 * (7 in 8-byte encodings). The kernel is bound by vector issue beside the MFMAs, not by the MFMAs. */
	typedef short s2 __attribute__((ext_vector_type(2)));
	const s2 p01 = __builtin_amdgcn_cvt_pk_i16(a0 >> (rs - 1), a1 >> (rs - 1)), p23 = __builtin_amdgcn_cvt_pk_i16(a2 >> (rs - 1), a3 >> (rs - 1));
	uint32_t q01, q23;
	asm("v_sat_pk_u8_i16 %0, %1" : "=v"(q01) : "v"(p01));
	asm("v_sat_pk_u8_i16 %0, %1" : "=v"(q23) : "v"(p23));
	return (((q23 << 16) | q01) >> 1) & 0x7f7f7f7fu;
}

/* The same for an output shift of exactly 8 (conv1 and conv2 of the shipped model), with or without the max-pool partner o:
 * ssat8(a >> 8) is the HIGH BYTE of a saturated to int16 -- clamp(a, -32768, 32767) >> 8 lies in -128..127 and equals a >> 8
 * inside the range -- so v_cvt_pk_i16_i32 (two accumulators saturated into one dword) does shift, narrowing and saturation at
 * once, v_max3_i32(e, o, 0) does the pool maximum and the ReLU (both commute with the monotone requantisation), and one
 * v_perm_b32 picks the four high bytes: 7 instructions per packed dword (5 without pooling) instead of 15 (11). 60 of the 80
 * dwords a lane requantises per group are conv1's and conv2's. */
__device__ __forceinline__ uint32_t edm_pack_relu8(int a0, int a1, int a2, int a3)
{
	typedef short s2 __attribute__((ext_vector_type(2)));
	const s2 p01 = __builtin_amdgcn_cvt_pk_i16(a0, a1), p23 = __builtin_amdgcn_cvt_pk_i16(a2, a3);
	uint32_t u01, u23;
	__builtin_memcpy(&u01, &p01, 4); __builtin_memcpy(&u23, &p23, 4);
	return __builtin_amdgcn_perm(u23, u01, 0x07050301u); /* bytes 1, 3 of p01, then bytes 1, 3 of p23 */
}
__device__ __forceinline__ int edm_max3z(int a, int b) { const int m = a > b ? a : b; return m > 0 ? m : 0; } /* v_max3_i32 a, b, 0 */

/* the accumulator tile that starts a 32-row output tile: D register r of lane half h is row (r&3) + 8*(r>>2) + 4*h, so
 * register group g (4 registers) is rows 8*g + 4*h .. +3 */
__device__ __forceinline__ v16i edm_seed_tile(const int32_t *seed4h)
{
	v16i c;
#pragma unroll
	for (int g = 0; g < 4; g++)
	{
		const v4i s = *reinterpret_cast<const v4i *>(seed4h + 8 * g);
		c[4 * g] = s.x; c[4 * g + 1] = s.y; c[4 * g + 2] = s.z; c[4 * g + 3] = s.w;
	}
	return c;
}

__device__ __forceinline__ int edm_max(int a, int b) { return a > b ? a : b; }

/* Weave: N times (1 MFMA, V VALU instructions) in the scheduling region that ends here. An MFMA holds vector issue for
 * 8 of its 32 cycles, so ~5 VALU instructions fit under each one (guide: cycle constants); placed in one block after
 * the MFMAs they would only start when the last MFMA has been issued. */
#define EDM_WEAVE(N, V)                                                        \
	_Pragma("unroll") for (int w_ = 0; w_ < (N); w_++)                           \
	{                                                                           \
		__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       \
		__builtin_amdgcn_sched_group_barrier(0x002, (V), 0);                     \
	}
#define EDM_FENCE()

/* Order this wave's LDS writes before its following LDS reads: DS instructions of a wave are issued and serviced in
 * order, the (code-less) wave barrier only keeps the compiler from moving memory operations across. */
__device__ __forceinline__ void edm_wave_sync() { __builtin_amdgcn_wave_barrier(); }

/*
 * The 13-byte feature rows of a group (4 utterances x 31 rows = 124 rows, two per lane) as RAW 16-byte loads: row y of
 * utterance u starts 13 bytes after row y - 1, so a load takes 3 bytes of the next row along -- except for the very last
 * row of the batch, which is fetched 3 bytes early instead (the 16 bytes END with the row). Nothing here looks at the
 * loaded values: what is needed of them (edm_fix_row: drop the foreign bytes, shift the early row into place) happens
 * where they are USED, an iteration later. (The first version masked and byte-assembled right here; the compiler put a
 * vmcnt(0) behind every load -- inside the code that was meant to prefetch.)
 */
__device__ __forceinline__ bool edm_row_is_last(int64_t base, int u, int y, int64_t n_utt) { return base + u == n_utt - 1 && y == ED_IN_H - 1; }

__device__ __forceinline__ void edm_load_rows(const int8_t *feat, int64_t feat_stride, int64_t base, int nb, int64_t n_utt,
                                              int lane, uint4 (&rows)[2])
{
#pragma unroll
	for (int pass = 0; pass < 2; pass++)
	{
		const int r = lane + 64 * pass;
		const int u = r / ED_IN_H, y = r - u * ED_IN_H;
		/* rows past the group / past the batch re-read the group's first row (in bounds) and are zeroed at the use */
		const bool have = r < EDM_G * ED_IN_H && u < nb;
		const uint8_t *g = reinterpret_cast<const uint8_t *>(feat) + (base + (have ? u : 0)) * feat_stride + (have ? y : 0) * ED_IN_W;
		if (have && edm_row_is_last(base, u, y, n_utt)) g -= 3;
		uint4 d;
		__builtin_memcpy(&d, g, 16);
		rows[pass] = d;
	}
}

/* the 13 bytes of the row out of what edm_load_rows fetched for it, zero-padded to 16 */
__device__ __forceinline__ uint4 edm_fix_row(uint4 d, int64_t base, int nb, int64_t n_utt, int r)
{
	const int u = r / ED_IN_H, y = r - u * ED_IN_H;
	if (edm_row_is_last(base, u, y, n_utt)) /* fetched 3 bytes early */
		d = make_uint4(__builtin_amdgcn_alignbit(d.y, d.x, 24), __builtin_amdgcn_alignbit(d.z, d.y, 24), __builtin_amdgcn_alignbit(d.w, d.z, 24), d.w >> 24);
	d.w &= 0xffu; /* bytes 13..15 belong to the next row */
	return u < nb ? d : make_uint4(0, 0, 0, 0);
}

/* The kernel in two parts, so that ed_kws1_kernel (below) can put the MFCC of the newest frame between them:
 * edm_prologue requests the first group's feature rows and stages the weights (no barrier), edm_main waits at the workgroup
 * barrier and does everything else. row30: null, or 16 bytes in LDS that hold the LAST feature row of the (single) utterance
 * at +3 -- where edm_load_rows would have fetched it from, had it been in memory already. */
__device__ __forceinline__ void edm_prologue(const ed_cnn_mfma_model_t *__restrict__ model, const int8_t *__restrict__ feat, int64_t n_utt,
                                             int64_t feat_stride, unsigned char *smem, uint4 (&rows)[2])
{
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	unsigned *queue = reinterpret_cast<unsigned *>(smem + sizeof(ed_cnn_mfma_model_t) + EDM_WAVES * EDM_WAVE_LDS);
	const int64_t n_groups = (n_utt + EDM_G - 1) / EDM_G;
	const int64_t g_lo = (int64_t)blockIdx.x * n_groups / gridDim.x;
	const uint32_t cnt = (uint32_t)((int64_t)(blockIdx.x + 1) * n_groups / gridDim.x - g_lo);
	/* The feature rows of the wave's first group go in flight BEFORE the weight staging (round 3): loads return in issue order,
	 * so the staging below waits behind them anyway, and for a one-window push of the microphone path these rows come over
	 * the bus from host-mapped memory (~2 us) -- with the rows requested after the staging barrier the two latencies added up.
	 * The rows of a wave's next group are fetched while it works on the current one. */
	if ((uint32_t)wave < cnt)
	{
		const int64_t b0 = (g_lo + wave) * EDM_G;
		edm_load_rows(feat, feat_stride, b0, (int)((n_utt - b0) < EDM_G ? (n_utt - b0) : EDM_G), n_utt, lane, rows);
	}
	{ /* stage the weight fragments once per workgroup */
		const v4i *src = reinterpret_cast<const v4i *>(model);
		v4i *dst = reinterpret_cast<v4i *>(smem);
		for (int i = threadIdx.x; i < (int)(sizeof(ed_cnn_mfma_model_t) / 16); i += EDM_THREADS) dst[i] = src[i];
		if (threadIdx.x == 0) *queue = EDM_WAVES;
	}
}

/* F8: the output shifts of conv1 and conv2 are both 8 (edm_pack_relu8); the kernels choose the instantiation from the model */
template <bool HAS_FILTER, bool F8>
__device__ __forceinline__ void edm_main(const int8_t *__restrict__ feat, int64_t n_utt, int64_t feat_stride, int8_t *__restrict__ logits,
                                         int8_t *__restrict__ softmax, int32_t *__restrict__ argmax, unsigned *done_flag, unsigned done_seq,
                                         unsigned char *smem, uint4 (&rows)[2], const unsigned char *row30, const ed_out_filter_t &flt, int with_filter)
{
	const ed_cnn_mfma_model_t &M = *reinterpret_cast<const ed_cnn_mfma_model_t *>(smem);
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	unsigned char *acts = smem + sizeof(ed_cnn_mfma_model_t) + wave * EDM_WAVE_LDS; /* wave-private */
	unsigned char *park = acts + EDM_G * EDM_UTT;                                     /* [EDM_PARK][EDM_G][16] logits */
	int *park_group = reinterpret_cast<int *>(park + EDM_PARK * EDM_G * 16);          /* [EDM_PARK] group index      */
	/* where idle columns store (no branch): a ds_write_b128 is served in groups of 8 consecutive lanes, and idle lanes that all wrote ONE
	 * slot were served one after the other -- every lane of a group has a slot of its own */
	unsigned char *dummy = park + EDM_PARK * EDM_G * 16 + 32 + 16 * (lane & 7);
	unsigned *queue = reinterpret_cast<unsigned *>(smem + sizeof(ed_cnn_mfma_model_t) + EDM_WAVES * EDM_WAVE_LDS);

	/* this workgroup's contiguous slice of the utterance groups; its waves draw from it */
	const int64_t n_groups = (n_utt + EDM_G - 1) / EDM_G;
	const int64_t g_lo = (int64_t)blockIdx.x * n_groups / gridDim.x;
	const uint32_t cnt = (uint32_t)((int64_t)(blockIdx.x + 1) * n_groups / gridDim.x - g_lo);
	__syncthreads();
	if (row30 && wave == 0 && lane == ED_IN_H - 1) /* uniform: ed_kws1_kernel -- the newest row was computed by this very wave a moment ago */
	{
		uint4 d;
		__builtin_memcpy(&d, row30, 16);
		rows[0] = d;
	}
	const int col = lane & 31, h = lane >> 5;
	const unsigned char *afrag = reinterpret_cast<const unsigned char *>(&M) + lane * 16;
	const int a1_off = (int)offsetof(ed_cnn_mfma_model_t, a1), a2_off = (int)offsetof(ed_cnn_mfma_model_t, a2);
	const int a3_off = (int)offsetof(ed_cnn_mfma_model_t, a3), a4_off = (int)offsetof(ed_cnn_mfma_model_t, a4);
	const int afc_off = (int)offsetof(ed_cnn_mfma_model_t, afc);
	/* the output shifts live in scalar registers: read from LDS inside the epilogues they cost a wait per dword */
	const int rs1 = __builtin_amdgcn_readfirstlane(M.rs1), rs2 = __builtin_amdgcn_readfirstlane(M.rs2);
	const int rs3 = __builtin_amdgcn_readfirstlane(M.rs3), rs4 = __builtin_amdgcn_readfirstlane(M.rs4);
	const int rsfc = __builtin_amdgcn_readfirstlane(M.rsfc);

	int parked = 0;
	for (uint32_t idx = wave; idx < cnt;)
	{
		/* the group after this one: drawn now, its index is needed only after the input rows are in LDS */
		uint32_t drawn = 0;
		if (lane == 0) drawn = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

		/* utterances in this group (4 but for the batch's last group -- and 1 for a one-window push of the microphone path):
		 * column tiles that hold none of them are skipped under wave-uniform branches (conv1: 13 columns per utterance,
		 * conv2: 35, conv3: 15), which is what a one-window launch's latency is made of */
		const int64_t b_cur = (g_lo + idx) * EDM_G;
		const int nb_cur = (int)((n_utt - b_cur) < EDM_G ? (n_utt - b_cur) : EDM_G);
		const bool full = nb_cur == EDM_G; /* wave-uniform */

		EDM_PR(0)
		/* ---- input: feat[u][31][13] -> in'[u][31][16] (3 zero bytes of padding per row) */
#pragma unroll
		for (int pass = 0; pass < 2; pass++)
		{
			const int r = lane + 64 * pass;
			if (r < EDM_G * ED_IN_H)
			{
				const int u = r / ED_IN_H, y = r - u * ED_IN_H;
				*reinterpret_cast<uint4 *>(acts + u * EDM_UTT + (y & 1) * EDM_IN_ODD + (y >> 1) * 16) =
				    edm_fix_row(rows[pass], b_cur, nb_cur, n_utt, r);
			}
		}
		const uint32_t next = __builtin_amdgcn_readfirstlane(drawn);
		if (next < cnt)
		{
			const int64_t bn = (g_lo + next) * EDM_G;
			edm_load_rows(feat, feat_stride, bn, (int)((n_utt - bn) < EDM_G ? (n_utt - bn) : EDM_G), n_utt, lane, rows);
		}
		edm_wave_sync();

		EDM_PR(1)
		/* ---- conv1 5x5x1->16 + ReLU + pool(2,1): Toeplitz GEMM, 144 rows (x,o) x 80 k (5 padded input rows).
		 *      columns = (utt, pooled row py): 4 x 13 = 52 in 2 column tiles; two accumulators = input rows 2py / 2py+1 */
		if (!(EDM_SKIP & 1))
		{
			/* rows of a conv1 tile are (x, o) in the order model.c gives the A fragments (tile_row): lane half h owns x = 2 rt + h,
			 * its register group g holds channels 4 g .. 4 g + 3 -- the seeds are the 16 channels in order */
			v16i seed1;
#pragma unroll
			for (int g = 0; g < 4; g++)
			{
				const v4i s_ = *reinterpret_cast<const v4i *>(&M.b1[4 * g]);
				seed1[4 * g] = s_.x; seed1[4 * g + 1] = s_.y; seed1[4 * g + 2] = s_.z; seed1[4 * g + 3] = s_.w;
			}
			/* the 15 weight fragments stay in registers for both column tiles: the MFMA stream then never waits for LDS */
			v4i A1[15];
#pragma unroll
			for (int i = 0; i < 15; i++) A1[i] = edm_ld16(afrag + a1_off + i * 1024);
			/* One column tile: where its lanes read and store, its six B fragments. For a full group the column a lane computes comes
			 * from the order that keeps the LDS accesses of a lane group on different banks (M.cols1, edison_internal.h), for a partial
			 * one it is the natural order (only the tiles that hold a live utterance run). */
			struct tile1_t { v4i be[3], bo[3]; unsigned char *p1; bool live; };
			auto load_tile = [&](int t, tile1_t &c) {
				int rd_off, st_off;
				if (full)
				{
					const uint32_t e = M.cols1[t][col];
					rd_off = (int)(e & 0x7fffu); st_off = (int)((e >> 16) & 0x7fffu); c.live = !(e & ED_CNN_COL_IDLE);
				}
				else
				{
					const int q = t * 32 + col;
					c.live = q < EDM_G * 13;
					const int qq = c.live ? q : EDM_G * 13 - 1; /* idle columns recompute the last one and store nothing */
					const int u = qq / 13, py = qq - u * 13;
					rd_off = u * EDM_UTT + py * 16; st_off = u * EDM_UTT + EDM_REGA + (py * 9) * 16;
				}
				const unsigned char *inb = acts + rd_off; /* row 2 py of the even plane; row 2 py + 1 of the odd plane is EDM_IN_ODD further */
#pragma unroll
				for (int s = 0; s < 3; s++)
				{
					const int k = (2 * s + h) < 4 ? (2 * s + h) : 4; /* k-chunk = input row 2 py + k; chunk 5 meets zero weights */
					/* row r lies in plane r & 1 at slot r >> 1 */
					c.be[s] = edm_ld16(inb + (k & 1) * EDM_IN_ODD + (k >> 1) * 16);
					c.bo[s] = edm_ld16(inb + ((k + 1) & 1) * EDM_IN_ODD + ((k + 1) >> 1) * 16);
				}
				c.p1 = acts + st_off;
			};
			/* software pipeline: the MFMAs of the next row tile are issued BEFORE the requantisation of this one, so that VALU work
			 * runs while the matrix pipe is busy (an MFMA blocks vector issue for 8 of its 32 cycles) */
			v16i ae[2], ao[2];
			auto issue = [&](const tile1_t &c, int rt, int slot) {
				ae[slot] = seed1; ao[slot] = seed1;
#pragma unroll
				for (int s = 0; s < 3; s++)
				{
					ae[slot] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1[rt * 3 + s], c.be[s], ae[slot], 0, 0, 0);
					ao[slot] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1[rt * 3 + s], c.bo[s], ao[slot], 0, 0, 0);
				}
			};
			auto requant1 = [&](const tile1_t &c, int rt, int slot) {
				const v16i &e = ae[slot], &o = ao[slot];
				uint32_t d[4];
#pragma unroll
				for (int g = 0; g < 4; g++)
					d[g] = F8 ? edm_pack_relu8(edm_max3z(e[4 * g], o[4 * g]), edm_max3z(e[4 * g + 1], o[4 * g + 1]),
					                           edm_max3z(e[4 * g + 2], o[4 * g + 2]), edm_max3z(e[4 * g + 3], o[4 * g + 3]))
					          : edm_pack_relu(edm_max(e[4 * g], o[4 * g]), edm_max(e[4 * g + 1], o[4 * g + 1]),
					                          edm_max(e[4 * g + 2], o[4 * g + 2]), edm_max(e[4 * g + 3], o[4 * g + 3]), rs1);
				/* the whole 16-byte record of x = 2 rt + h, as it stands (row order of the A fragments: model.c tile_row) */
				const uint4 rec = make_uint4(d[0], d[1], d[2], d[3]);
				*reinterpret_cast<uint4 *>((2 * rt + h < 9 && c.live) ? c.p1 + (2 * rt + h) * 16 : dummy) = rec;
			};
			if (full)
			{
				/* both column tiles as ONE pipeline of ten (column tile, row tile) steps: the second tile's B fragments are requested
				 * a step ahead, and its first MFMAs go out before the first tile's last requantisation -- between
				 * the tiles the matrix pipe used to drain (a requantisation with nothing behind it) and refill (six MFMAs with
				 * nothing to do beside them, behind six LDS reads) */
				tile1_t c0, c1;
				load_tile(0, c0);
				issue(c0, 0, 0);
#pragma unroll
				for (int k = 0; k < 10; k++)
				{
					const int rt = k % 5;
					if (k + 1 < 10) { if (k + 1 < 5) issue(c0, (k + 1) % 5, (k + 1) & 1); else issue(c1, (k + 1) % 5, (k + 1) & 1); }
					if (k == 3) load_tile(1, c1); /* behind the first tile's last MFMAs (its fragments are free), a whole step ahead of its own first */
					EDM_FENCE();
					if (k < 5) requant1(c0, rt, k & 1); else requant1(c1, rt, k & 1);
					if (k + 1 < 10) { if (F8) { EDM_WEAVE(6, EDM_W1V) } else { EDM_WEAVE(6, 13) } }
					__builtin_amdgcn_sched_barrier(0);
				}
			}
			else
			{
#pragma unroll 1
				for (int t = 0; t < 2; t++)
				{
					if (t * 32 >= nb_cur * 13) break; /* no live column in this tile */
					tile1_t c;
					load_tile(t, c);
					issue(c, 0, 0);
#pragma unroll
					for (int rt = 0; rt < 5; rt++)
					{
						if (rt + 1 < 5) issue(c, rt + 1, (rt + 1) & 1);
						EDM_FENCE();
						requant1(c, rt, rt & 1);
						if (rt + 1 < 5) { EDM_WEAVE(6, 13) }
						__builtin_amdgcn_sched_barrier(0);
					}
				}
			}
		}
		edm_wave_sync();

		EDM_PR(2)
		/* ---- conv2 3x3x16->32 + ReLU + pool(2,1): K = 9 taps x 16 ch (5 k-steps of 2 taps); columns =
		 *      (utt, py, x): 4 x 35 = 140 in 5 column tiles; two accumulators = conv rows 2py / 2py+1 */
		if (!(EDM_SKIP & 2))
		{
			const v16i seed2 = edm_seed_tile(&M.b2[4 * h]);
			/* three-stage software pipeline per column tile: fetch (LDS reads of the B fragments) two tiles ahead, MFMAs
			 * one tile ahead, requantisation of the current tile -- the matrix pipe never waits for LDS and the VALU work
			 * runs under the MFMAs. The 5 weight fragments stay in registers. */
			v4i A2[5];
#pragma unroll
			for (int i = 0; i < 5; i++) A2[i] = edm_ld16(afrag + a2_off + i * 1024);
			v4i B0[2][5], B1[2][5];
			v16i ae[2], ao[2];
			unsigned char *p2[3];
			bool live[3];
			/* bs: which of the two fragment sets, ls: which of the three (live, p2) slots */
			auto fetch = [&](int t, int bs, int ls, auto full_) {
				const unsigned char *p1;
				if (decltype(full_)::value) /* a full group: the conflict-free column order (M.cols2) */
				{
					const uint32_t e = M.cols2[t][col];
					live[ls] = !(e & ED_CNN_COL_IDLE);
					p1 = acts + (e & 0x7fffu);
					p2[ls] = acts + ((e >> 16) & 0x7fffu); /* plane h is EDM_P2_PLANE * h further */
				}
				else
				{
					const int q = t * 32 + col;
					live[ls] = q < nb_cur * 35;
					const int qq = live[ls] ? q : nb_cur * 35 - 1;
					const int u = qq / 35, r = qq - u * 35, py = r / 7, x = r - py * 7;
					p1 = acts + u * EDM_UTT + EDM_REGA + ((2 * py) * 9 + x) * 16;
					p2[ls] = acts + u * EDM_UTT + (py * 7 + x) * 16;
				}
#pragma unroll
				for (int s = 0; s < 5; s++)
				{
					const int tap = (2 * s + h) < 8 ? (2 * s + h) : 8; /* tap 9 meets zero weights */
					const int ky = tap / 3, kx = tap - 3 * ky;
					B0[bs][s] = edm_ld16(p1 + (ky * 9 + kx) * 16);
					B1[bs][s] = edm_ld16(p1 + ((ky + 1) * 9 + kx) * 16);
				}
			};
			auto mma = [&](int bs) {
				ae[bs] = seed2; ao[bs] = seed2;
#pragma unroll
				for (int s = 0; s < 5; s++)
				{
					ae[bs] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A2[s], B0[bs][s], ae[bs], 0, 0, 0);
					ao[bs] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A2[s], B1[bs][s], ao[bs], 0, 0, 0);
				}
			};
			auto requant = [&](int bs, int ls, auto f8_) {
				constexpr bool R8 = decltype(f8_)::value; /* (the partial-group road below keeps the general form) */
				const v16i &e = ae[bs], &o = ao[bs];
				uint32_t d[4];
#pragma unroll
				for (int g = 0; g < 4; g++)
					d[g] = R8 ? edm_pack_relu8(edm_max3z(e[4 * g], o[4 * g]), edm_max3z(e[4 * g + 1], o[4 * g + 1]),
					                           edm_max3z(e[4 * g + 2], o[4 * g + 2]), edm_max3z(e[4 * g + 3], o[4 * g + 3]))
					          : edm_pack_relu(edm_max(e[4 * g], o[4 * g]), edm_max(e[4 * g + 1], o[4 * g + 1]),
					                          edm_max(e[4 * g + 2], o[4 * g + 2]), edm_max(e[4 * g + 3], o[4 * g + 3]), rs2);
				const uint4 rec = make_uint4(d[0], d[1], d[2], d[3]); /* channels 16h .. 16h + 15 (model.c tile_row) */
				*reinterpret_cast<uint4 *>(live[ls] ? p2[ls] + EDM_P2_PLANE * h : dummy) = rec;
			};
			if (nb_cur == EDM_G)
			{
				fetch(0, 0, 0, std::true_type()); fetch(1, 1, 1, std::true_type());
				mma(0);
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int t = 0; t < 5; t++)
				{
					if (t + 1 < 5) mma((t + 1) & 1);
					EDM_FENCE();
					if (t + 2 < 5) fetch(t + 2, t & 1, (t + 2) % 3, std::true_type()); /* into the fragment registers tile t's MFMAs have consumed */
					requant(t & 1, t % 3, std::integral_constant<bool, F8>());
					if (t + 1 < 5) { if (F8) { EDM_WEAVE(10, EDM_W2V) } else { EDM_WEAVE(10, 7) } }
					__builtin_amdgcn_sched_barrier(0);
				}
			}
			else
			{
				/* a partial group (the batch's last one, or the single window of a microphone push): only the column tiles
				 * that hold a live column, one after the other */
				const int n_t2 = (nb_cur * 35 + 31) >> 5;
#pragma unroll 1
				for (int t = 0; t < n_t2; t++)
				{
					fetch(t, 0, 0, std::false_type());
					mma(0);
					requant(0, 0, std::false_type());
				}
			}
		}
		edm_wave_sync();

		EDM_PR(3)
		/* ---- conv3 3x3x32->64 + ReLU: 9 k-steps (tap, 16-channel half); columns = (utt, y, x): 4 x 15 = 60 in 2
		 *      column tiles; two accumulators = output channels 0-31 / 32-63 */
		if (!(EDM_SKIP & 4))
		{
			const v16i seed3a = edm_seed_tile(&M.b3[4 * h]), seed3b = edm_seed_tile(&M.b3[32 + 4 * h]);
			v16i a0[2], a1[2];
			unsigned char *c3[2];
			bool live[2];
			auto issue = [&](int t, int slot) {
				const unsigned char *p2;
				if (full) /* uniform: the conflict-free column order (M.cols3) */
				{
					const uint32_t e = M.cols3[t][col];
					live[slot] = !(e & ED_CNN_COL_IDLE);
					p2 = acts + (e & 0x7fffu) + EDM_P2_PLANE * h;
					c3[slot] = acts + ((e >> 16) & 0x7fffu); /* plane j (channels 16 j ..) is EDM_C3_PLANE * j further */
				}
				else
				{
					const int q = t * 32 + col;
					live[slot] = q < nb_cur * 15;
					const int qq = live[slot] ? q : nb_cur * 15 - 1;
					const int u = qq / 15, r = qq - u * 15, y = r / 5, x = r - y * 5;
					p2 = acts + u * EDM_UTT + EDM_P2_PLANE * h + (y * 7 + x) * 16;
					c3[slot] = acts + u * EDM_UTT + EDM_REGA + (y * 5 + x) * 16;
				}
				a0[slot] = seed3a; a1[slot] = seed3b;
#pragma unroll
				for (int s = 0; s < 9; s++)
				{
					const int ky = s / 3, kx = s - 3 * ky;
					const v4i b = edm_ld16(p2 + (ky * 7 + kx) * 16);
					a0[slot] = __builtin_amdgcn_mfma_i32_32x32x32_i8(edm_ld16(afrag + a3_off + s * 1024), b, a0[slot], 0, 0, 0);
					a1[slot] = __builtin_amdgcn_mfma_i32_32x32x32_i8(edm_ld16(afrag + a3_off + (9 + s) * 1024), b, a1[slot], 0, 0, 0);
				}
			};
			auto requant3 = [&](int slot) {
				const v16i &x0 = a0[slot], &x1 = a1[slot];
				uint32_t d0[4], d1[4];
#pragma unroll
				for (int g = 0; g < 4; g++)
				{
					d0[g] = edm_pack_relu(x0[4 * g], x0[4 * g + 1], x0[4 * g + 2], x0[4 * g + 3], rs3);
					d1[g] = edm_pack_relu(x1[4 * g], x1[4 * g + 1], x1[4 * g + 2], x1[4 * g + 3], rs3);
				}
				const uint4 ra = make_uint4(d0[0], d0[1], d0[2], d0[3]), rb = make_uint4(d1[0], d1[1], d1[2], d1[3]); /* (model.c tile_row) */
				*reinterpret_cast<uint4 *>(live[slot] ? c3[slot] + EDM_C3_PLANE * h : dummy) = ra;       /* channels 16h ..: plane h */
				*reinterpret_cast<uint4 *>(live[slot] ? c3[slot] + EDM_C3_PLANE * (2 + h) : dummy) = rb; /* channels 32 + 16h ..: plane 2 + h */
			};
			issue(0, 0);
			if (nb_cur * 15 > 32) /* uniform: the second column tile holds a live column */
			{
				issue(1, 1);
				EDM_FENCE();
				requant3(0);
				EDM_WEAVE(18, EDM_W3V)
				__builtin_amdgcn_sched_barrier(0);
				requant3(1);
			}
			else
				requant3(0);
			__builtin_amdgcn_sched_barrier(0);
		}
		edm_wave_sync();

		EDM_PR(4)
		/* ---- conv4 3x3x64->32 + ReLU on v_mfma_i32_16x16x64_i8: columns = (utt, x): 4 x 3 = 12 of a 16-column tile, two row
		 *      tiles of 16 channels, 9 k-steps = taps (64 input channels each: lane quarter kq reads plane kq of c3). A
		 *      32 x 32 x 32 tile was 5/8 padding here: 18 MFMAs of 32 cycles, now 18 of 16, and 8 accumulators to requantise
		 *      instead of 16. Lane (col, kq) ends with channels 16 rt + 4 kq .. +3 of its pixel. */
		if (!(EDM_SKIP & 8))
		{
			const int c16 = lane & 15, kq = lane >> 4;
			const bool live = c16 < EDM_G * 3;
			const int qq = live ? c16 : EDM_G * 3 - 1;
			const int u = qq / 3, x = qq - u * 3;
			const unsigned char *c3 = acts + u * EDM_UTT + EDM_REGA + EDM_C3_PLANE * kq + x * 16;
			v4i acc0 = *reinterpret_cast<const v4i *>(&M.b4[4 * kq]), acc1 = *reinterpret_cast<const v4i *>(&M.b4[16 + 4 * kq]);
#pragma unroll
			for (int s = 0; s < 9; s++)
			{
				const int ky = s / 3, kx = s - 3 * ky;
				const v4i b = edm_ld16(c3 + (ky * 5 + kx) * 16);
				acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(edm_ld16(afrag + a4_off + s * 1024), b, acc0, 0, 0, 0);
				acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(edm_ld16(afrag + a4_off + (9 + s) * 1024), b, acc1, 0, 0, 0);
			}
			unsigned char *c4 = acts + u * EDM_UTT + x * 32 + 4 * kq; /* c4 [3][32]: channel 16 rt + 4 kq of pixel x */
			const uint32_t d0 = edm_pack_relu(acc0.x, acc0.y, acc0.z, acc0.w, rs4), d1 = edm_pack_relu(acc1.x, acc1.y, acc1.z, acc1.w, rs4);
			if (live)
			{
				*reinterpret_cast<uint32_t *>(c4) = d0;
				*reinterpret_cast<uint32_t *>(c4 + 16) = d1;
			}
		}
		edm_wave_sync();

		EDM_PR(5)
		/* ---- dense 96->10, same tile shape: rows = 10 logits of 16, columns = utterances (4 live), K = 96 in two k-steps of
		 *      64: bytes 0..95 of c4 and 32 bytes behind it that meet zero weights. Lane (utt, kq) holds logits 4 kq .. +3
		 *      (kq = 2: 8, 9 and two padding rows; kq = 3: padding only): parked as 10 int8 per utterance */
		if (!(EDM_SKIP & 16))
		{
			const int c16 = lane & 15, kq = lane >> 4;
			const int uu = c16 < EDM_G ? c16 : EDM_G - 1;
			const unsigned char *c4 = acts + uu * EDM_UTT + 16 * kq;
			v4i acc = {0, 0, 0, 0};
#pragma unroll
			for (int s = 0; s < 2; s++)
				acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(edm_ld16(afrag + afc_off + s * 1024), edm_ld16(c4 + 64 * s), acc, 0, 0, 0);
			if (c16 < EDM_G && kq < 3)
			{
				const v4i bias = *reinterpret_cast<const v4i *>(&M.bfc[4 * kq]);
				const uint32_t w = (uint32_t)(uint8_t)edm_med3((acc.x + bias.x) >> rsfc, -128, 127) | ((uint32_t)(uint8_t)edm_med3((acc.y + bias.y) >> rsfc, -128, 127) << 8) |
				                   ((uint32_t)(uint8_t)edm_med3((acc.z + bias.z) >> rsfc, -128, 127) << 16) | ((uint32_t)(uint8_t)edm_med3((acc.w + bias.w) >> rsfc, -128, 127) << 24);
				*reinterpret_cast<uint32_t *>(park + (parked * EDM_G + c16) * 16 + 4 * kq) = w; /* logits 4 kq .. 4 kq + 3 (10, 11: unused) */
			}
			if (lane == 0) park_group[parked] = (int)idx;
			parked++;
		}
		edm_wave_sync();

		EDM_PR(6)
		/* ---- softmax + argmax + stores for the parked utterances: every EDM_PARK groups, and after the wave's last */
		if (parked == EDM_PARK || next >= cnt)
		{
			const int slot = lane >> 2, u = lane & 3; /* lanes 0..31 */
			if (lane < EDM_PARK * EDM_G && slot < parked)
			{
				const int64_t utt = (g_lo + park_group[slot]) * EDM_G + u;
				if (utt < n_utt)
				{
					const uint4 raw = *reinterpret_cast<const uint4 *>(park + (slot * EDM_G + u) * 16);
					const uint32_t lw[3] = {raw.x, raw.y, raw.z};
					int lg[10];
#pragma unroll
					for (int i = 0; i < 10; i++) lg[i] = (int)(int8_t)(lw[i >> 2] >> (8 * (i & 3)));
					/* arm_softmax_q7 (portable branch) and nnom_predict's first-maximum rule */
					int mx = -128;
#pragma unroll
					for (int i = 0; i < 10; i++) mx = lg[i] > mx ? lg[i] : mx;
					const int sbase = mx - 8;
					int sum = 0;
#pragma unroll
					for (int i = 0; i < 10; i++) sum += 1 << edm_med3(lg[i] - sbase, 0, 7);
					const int output_base = (1 << 20) / sum;
					int best = 0, bv = -129;
					uint32_t sw[3] = {0, 0, 0};
#pragma unroll
					for (int i = 0; i < 10; i++)
					{
						const int v = edm_med3(output_base >> edm_med3(13 + sbase - lg[i], 0, 31), -128, 127);
						if (v > bv) { bv = v; best = i; }
						sw[i >> 2] |= (uint32_t)(uint8_t)v << (8 * (i & 3));
					}
					const int64_t uo = utt * ED_FC_O; /* 10-byte records: 2-byte aligned */
					if (logits)
					{
						uint16_t *p = reinterpret_cast<uint16_t *>(logits + uo);
						p[0] = (uint16_t)lw[0]; p[1] = (uint16_t)(lw[0] >> 16); p[2] = (uint16_t)lw[1];
						p[3] = (uint16_t)(lw[1] >> 16); p[4] = (uint16_t)lw[2];
					}
					if (softmax)
					{
						uint16_t *p = reinterpret_cast<uint16_t *>(softmax + uo);
						p[0] = (uint16_t)sw[0]; p[1] = (uint16_t)(sw[0] >> 16); p[2] = (uint16_t)sw[1];
						p[3] = (uint16_t)(sw[1] >> 16); p[4] = (uint16_t)sw[2];
					}
					if (argmax) argmax[utt] = best;
					if (HAS_FILTER && with_filter && utt == 0) /* ed_kws1_kernel with the stream's output filter: app.c:341-356 for this one inference, in this lane */
					{
						float ymax = 0.0f;
						int imax = 0;
#pragma unroll
						for (int i = 0; i < 10; i++)
						{
#pragma clang fp contract(off) /* separately rounded product and sum: see ed_stream_filter_kernel */
							const int v = (int)(int8_t)(sw[i >> 2] >> (8 * (i & 3)));
							/* separately rounded double multiply / add (the Cortex-M4 has no double FPU, nothing is fused) */
							const double pa = flt.alpha * (double)flt.state[i], pb = flt.one_minus_alpha * (double)(float)v; /* (not __dmul_rn / __dadd_rn: see ed_stream_filter_kernel) */
							const float y = (float)(pa + pb);
							flt.state[i] = y;
							flt.filt[i] = y;
							if (i == 0 || ymax < y) { ymax = y; imax = i; } /* arm_max_f32: the first maximum */
						}
						*flt.likely = imax;
						const int hit = (double)ymax > flt.threshold;
						*flt.spotted = hit ? imax : -1;
						if (flt.fsm) /* ... and the state machine's step for it (app.c:371) */
						{
							edison_fsm *fsm = reinterpret_cast<edison_fsm *>(flt.fsm);
							edison_fsm m = *fsm;
							const ed_fsm_roles_t roles = {flt.wake_idx, flt.loc_mask, flt.val_mask};
							*flt.fsm_state = ed_fsm_step_core(&m, hit, (uint32_t)imax, flt.dt_us, &roles);
							*fsm = m;
							if (flt.fsm_copy) *reinterpret_cast<edison_fsm *>(flt.fsm_copy) = m;
						}
					}
				}
			}
			parked = 0;
			edm_wave_sync();
		}
		idx = next;
	}
	/* one-group launches of the microphone path (edison_stream.hip): the wave that did the work tells the host itself, in
	 * host-mapped memory, behind a system-scope fence -- a command-processor write behind the kernel costs ~2 us more
	 * (tools/ubench/launch_lat). The launcher passes a flag only when the launch has exactly one group. */
	if (done_flag && blockIdx.x == 0 && wave == 0)
	{
		__threadfence_system();
		if (lane == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

__global__ __launch_bounds__(EDM_THREADS) void ed_cnn_mfma_kernel(const ed_cnn_mfma_model_t *__restrict__ model,
                                                                 const int8_t *__restrict__ feat, int64_t n_utt,
                                                                 int64_t feat_stride, int8_t *__restrict__ logits,
                                                                 int8_t *__restrict__ softmax,
                                                                 int32_t *__restrict__ argmax, unsigned *done_flag, unsigned done_seq)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint4 rows[2];
	edm_prologue(model, feat, n_utt, feat_stride, smem, rows);
	ed_out_filter_t none;
	__builtin_memset(&none, 0, sizeof(none));
	if (model->rs1 == 8 && model->rs2 == 8) /* uniform, from the kernel argument: the shipped model and any retrained one with these shifts */
		edm_main<false, true>(feat, n_utt, feat_stride, logits, softmax, argmax, done_flag, done_seq, smem, rows, nullptr, none, 0);
	else
		edm_main<false, false>(feat, n_utt, feat_stride, logits, softmax, argmax, done_flag, done_seq, smem, rows, nullptr, none, 0);
}

/*
 * ed_kws1_kernel -- the one-frame microphone push in ONE launch (edison_stream.hip, host-mapped path, chunk 1): the MFCC of
 * the newest frame (variant A / B: ed_mfcc1_body, the one-frame kernel's body -- bit-identical to the batch kernel, which the
 * tests hold against it) and the CNN on the 31-row window it completes. Two dependent launches cost 8.7 us where one costs
 * 6.0 (tools/ubench/launch_lat), the host issues one launch instead of two, and the CNN's weight staging and its reads of
 * the 30 older rows (over the bus, from the host's ring) overlap the MFCC instead of following it.
 *   all waves: request the older rows (wave 0) and stage the CNN weights; stage the MFCC tables; workgroup barrier
 *   wave 0:    MFCC of the frame -> the row to the host's ring (history for the next pushes) and to 16 bytes of LDS
 *   all waves: the CNN kernel's barrier; wave 0 runs the one group, takes the newest row from LDS, writes outputs + flag
 * LDS: the CNN kernel's layout; the MFCC tables and wave 0's transform buffer live in the activation slots of waves 1..,
 * which have no group in a one-utterance launch.
 */
#include "mfcc_one_frame.h"
#define EDK1_MFCC_FLOATS(NLO, NHI) (ED_FIXTAB_FLOATS + ((NLO) + (NHI)) * 256 + ED_XBUF_FLOATS)

template <int NLO, int NHI>
__global__ __launch_bounds__(EDM_THREADS) void ed_kws1_kernel(ed_mfcc_args_t margs, const ed_mfcc_tables_t *__restrict__ tab,
                                                             const ed_cnn_mfma_model_t *__restrict__ model, const int8_t *__restrict__ feat,
                                                             int8_t *__restrict__ logits, int8_t *__restrict__ softmax, int32_t *__restrict__ argmax,
                                                             unsigned *done_flag, unsigned done_seq, ed_out_filter_t flt, int with_filter)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	static_assert(sizeof(float) * EDK1_MFCC_FLOATS(NLO, NHI) + 16 <= (size_t)(EDM_WAVES - 1) * EDM_WAVE_LDS, "the MFCC tables do not fit the idle waves' slots");
	float *msmem = reinterpret_cast<float *>(smem + sizeof(ed_cnn_mfma_model_t) + EDM_WAVE_LDS); /* behind wave 0's activations */
	unsigned char *row30 = reinterpret_cast<unsigned char *>(msmem + EDK1_MFCC_FLOATS(NLO, NHI));
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint4 rows[2];
	edm_prologue(model, feat, 1, ED_IN_W, smem, rows);
	/* one frame, taken by wave 0 (every wave passes wave index 0 for the LDS layout: only wave 0's transform buffer exists) */
	/* (ALIGNED: the launcher checks that the frame starts on a dword -- the host's ring does -- so the samples come over the bus
	 * as 8 dword loads per lane instead of 16 halfword loads) */
	ed_mfcc1_body<false, true, NLO, NHI>(margs, tab, msmem, 0, wave == 0 ? 0u : 1u, 1u << 30, reinterpret_cast<int8_t *>(row30) + 3);
	edm_main<true, false>(feat, 1, ED_IN_W, logits, softmax, argmax, done_flag, done_seq, smem, rows, row30, flt, with_filter); /* one window: latency, not throughput */
}

extern "C" int ed_launch_kws1(const ed_mfcc_args_t *margs, const ed_mfcc_tables_t *dev_tab, const ed_cnn_mfma_model_t *dev_model, const int8_t *feat,
                              int8_t *logits, int8_t *softmax, int32_t *argmax, unsigned *done_flag, unsigned done_seq, const ed_out_filter_t *filter,
                              hipStream_t stream);

/* "the kernel's dynamic-LDS limit has been raised" is a property of the function ON A DEVICE: one flag per device, so that
 * two contexts on different GPUs of one process both get it */
static int g_cnn_mfma_ready[16] = {0};

/* feat_stride = bytes between consecutive utterances' feature maps: 403 for packed utterances, 13 for the
 * sliding windows of a stream (window i = feature rows i..i+30 of one long [rows][13] buffer). */
extern "C" int ed_launch_cnn_mfma_flag(const ed_cnn_mfma_model_t *dev_model, const int8_t *feat, int64_t n_utt,
                                       int64_t feat_stride, int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu,
                                       hipStream_t stream, unsigned *done_flag, unsigned done_seq, int *flag_written);

extern "C" int ed_launch_cnn_mfma(const ed_cnn_mfma_model_t *dev_model, const int8_t *feat, int64_t n_utt,
                                  int64_t feat_stride, int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu,
                                  hipStream_t stream)
{
	return ed_launch_cnn_mfma_flag(dev_model, feat, n_utt, feat_stride, logits, softmax, argmax, n_cu, stream, nullptr, 0, nullptr);
}

/* done_flag (device address of host-mapped memory) / done_seq: written by the kernel behind its outputs when the launch is a
 * single group (n_utt <= 4); *flag_written tells the caller whether that is the case (else it must signal completion itself) */
extern "C" int ed_launch_cnn_mfma_flag(const ed_cnn_mfma_model_t *dev_model, const int8_t *feat, int64_t n_utt,
                                       int64_t feat_stride, int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu,
                                       hipStream_t stream, unsigned *done_flag, unsigned done_seq, int *flag_written)
{
	if (flag_written) *flag_written = 0;
	if (n_utt <= 0) return 0;
	const size_t lds = sizeof(ed_cnn_mfma_model_t) + (size_t)EDM_WAVES * EDM_WAVE_LDS + 16 /* queue */;
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	dev_ &= 15;
	if (!g_cnn_mfma_ready[dev_])
	{
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ed_cnn_mfma_kernel),
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return (int)e;
		g_cnn_mfma_ready[dev_] = 1;
	}
	const int64_t n_groups = (n_utt + EDM_G - 1) / EDM_G;
	int64_t blocks = (n_groups + EDM_WAVES - 1) / EDM_WAVES;
	if (blocks > n_cu) blocks = n_cu; /* 157 KB of LDS: one workgroup per CU */
	unsigned *flag = (done_flag && n_groups == 1) ? done_flag : nullptr;
	hipLaunchKernelGGL(ed_cnn_mfma_kernel, dim3((unsigned)blocks), dim3(EDM_THREADS), lds, stream, dev_model, feat, n_utt,
	                   feat_stride, logits, softmax, argmax, flag, done_seq);
	if (flag && flag_written) *flag_written = 1;
	return (int)hipGetLastError();
}


/* margs: ONE frame (n_frames = 1) with its int8 row going to margs->feat (the host's ring); feat: the 31-row window that row
 * completes (its last row is read from LDS, not from there). done_flag / done_seq as in ed_launch_cnn_mfma_flag: written when given
 * (null: something else follows on the stream and signals completion). filter: null, or the stream's output filter, applied to
 * this inference by the kernel (the softmax must be asked for then). */
extern "C" int ed_launch_kws1(const ed_mfcc_args_t *margs, const ed_mfcc_tables_t *dev_tab, const ed_cnn_mfma_model_t *dev_model, const int8_t *feat,
                              int8_t *logits, int8_t *softmax, int32_t *argmax, unsigned *done_flag, unsigned done_seq, const ed_out_filter_t *filter,
                              hipStream_t stream)
{
	if (filter && (!softmax || !filter->state || !filter->filt || !filter->likely || !filter->spotted)) return (int)hipErrorInvalidValue;
	if (margs->n_frames != 1 || !margs->feat || (reinterpret_cast<uintptr_t>(margs->audio) & 3)) return (int)hipErrorInvalidValue;
	const size_t lds = sizeof(ed_cnn_mfma_model_t) + (size_t)EDM_WAVES * EDM_WAVE_LDS + 16 /* queue */;
	const bool narrow = margs->mel_NLO == 2 && margs->mel_NHI == 5;
	if (!narrow && !(margs->mel_NLO == ED_MEL_NLO_MAX && margs->mel_NHI == ED_MEL_NHI_MAX)) return (int)hipErrorInvalidValue;
	const void *fn = narrow ? (const void *)ed_kws1_kernel<2, 5> : (const void *)ed_kws1_kernel<ED_MEL_NLO_MAX, ED_MEL_NHI_MAX>;
	static int ready[16][2];
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	dev_ &= 15;
	if (!ready[dev_][narrow])
	{
		hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return (int)e;
		ready[dev_][narrow] = 1;
	}
	ed_out_filter_t f;
	memset(&f, 0, sizeof(f));
	if (filter) f = *filter;
	int with_filter = filter != nullptr;
	void *kargs[] = {(void *)margs, (void *)&dev_tab, (void *)&dev_model, (void *)&feat, (void *)&logits, (void *)&softmax, (void *)&argmax,
	                 (void *)&done_flag, (void *)&done_seq, (void *)&f, (void *)&with_filter};
	return (int)hipLaunchKernel(fn, dim3(1), dim3(EDM_THREADS), kargs, lds, stream);
}
