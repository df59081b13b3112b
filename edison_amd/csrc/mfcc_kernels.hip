/*
 * mfcc_kernels.hip -- batched per-frame MFCC for gfx950 (CDNA4), hand-written HIP.
 *
 * Computes what the reference computes one frame at a time in Python/numpy float64
 *   variant A  audio/edison/mfcc/mfcc_utils.py:160-197   variant B  audio/edison/mfcc/mfcc_utils.py:287-322
 * in fp32, one 64-lane wavefront per 1024-sample frame, everything between the int16 load and the 13
 * output coefficients kept in registers/LDS (no intermediate HBM traffic: 2048 B in, <=128 B out per frame).
 *
 * Pipeline per wavefront (lane = 0..63):
 *   1. load      z[n] = x[2n] + i*x[2n+1], n = lane + 64a (a = 0..7): 8 coalesced 256-B wave loads, issued one
 *                frame ahead (software prefetch) so HBM latency hides under the previous frame's arithmetic
 *   2. FFT512    3 radix-8 passes over the digits of n = 64a + 8b + c, k = p + 8q + 64r. Each of the two digit
 *                transposes swaps the register index with three lane-index bits. Transpose 1 (lane bits 3-5) is a
 *                butterfly exchange on the VALU: v_permlane32_swap / v_permlane16_swap for bits 5 / 4, DPP
 *                row_shr/row_shl:8 with bank masks for bit 3. Transpose 2 (lane bits 0-2) goes through a padded
 *                wave-private LDS buffer -- with both transposes in LDS the write-heavy
 *                LDS pipe saturates, with both on the VALU (DPP row_shr/shl:4 and quad_perm +
 *                select) the VALU does; split, the two pipes are about equally loaded.
 *   3. split     X[k] = E[k] + W1024^k O[k] from Z[k], conj Z[512-k]; lane handles the pair (k, 512-k); the
 *                partner value comes through ds_bpermute (LDS crossbar, no LDS memory)
 *   4. |X|       -> wave-private LDS spectrum S[0..512] (+3 pad)
 *   5. mel       32 banded dot products, balanced: lane (b = lane&15, r = lane>>4) sums quarter r of the narrow
 *                band b and of the wide band 31-b as 16-byte quads (ds_read_b128 spectrum + weights); quarters are
 *                summed over the lane rows with VALU row swaps
 *   6. ln / DCT  optional ln(x+1e-6); DCT-II through cos symmetry against a per-lane LDS table
 *   7. store     n_coef fp32 and/or int8 (clip, round-half-even) per frame
 *
 * Waves never share LDS data, so there is no workgroup barrier inside the frame loop. The constant tables
 * (mel taps, DCT, split twiddles) are staged once per workgroup in LDS; only the pass-1/2 twiddles live in
 * registers.
 *
 * Two kernels implement this pipeline. ed_mfcc_kernel (one frame per wavefront) is the reference implementation of
 * the design and the one that dumps the intermediate stages; ed_mfcc2_kernel further down carries TWO frames per
 * wavefront in packed fp32 and is what every batched call without stage dumps runs (ED_MFCC_ONE_FRAME=1 selects the
 * former for A/B measurements).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "edison_internal.h"

#include "mfcc_one_frame.h"

template <bool STAGES, bool ALIGNED, int NLO, int NHI, bool WINDOW = false>
__global__ ED_MFCC_BOUNDS void ed_mfcc_kernel(ed_mfcc_args_t args, const ed_mfcc_tables_t *__restrict__ tab)
{
	extern __shared__ __attribute__((aligned(16))) float smem[];
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	ed_mfcc1_body<STAGES, ALIGNED, NLO, NHI, WINDOW>(args, tab, smem, wave, blockIdx.x * ED_WPB + wave, gridDim.x * ED_WPB, nullptr);
}

/* ================================================================================================================
 * Two frames per wavefront, packed fp32.
 *
 * Measured on gfx950 (tools/ubench/fft_pk.hip): a radix-8 pass with its twiddles costs ~3.7 cycles per VALU
 * instruction per wavefront at every occupancy from 1 to 8 waves/SIMD -- and exactly the same when every instruction
 * is the packed form (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) working on two independent values per lane. The
 * kernel is bound by instruction issue, not by ALU width, so the fast path carries TWO frames per wavefront: frame A in
 * the .x halves and frame B in the .y halves of 64-bit register pairs. All arithmetic (the three radix-8 passes, the
 * twiddles, the real-FFT split, the mel and DCT dot products) then costs one instruction for both frames; only the
 * data movement that works on 32-bit registers (int16 -> fp32, DPP / permlane exchanges, ds_bpermute, sqrt, log) is
 * issued once per frame. Same algorithm, same operation order per frame as ed_mfcc_kernel above, which stays as the
 * stage-dump kernel (the two agree to the last places -- the compiler fuses multiply-adds differently in the two texts --
 * and are held to the oracle with the same bars; the two-frame instances among themselves are bit-identical:
 * plain / grouped / list, aligned or not, one queue or two).
 * LDS per wave: 526 transpose slots of 16 B (re A, re B, im A, im B: register pairs stay pairs), the two spectra interleaved as float2[516]
 * behind them (aliased like above), the DCT inputs of both frames.
 */
#define ED2_S_OFF 1088    /* float offset of the interleaved spectra: their zero padding lies beyond the 2104 transpose floats */
#define ED2_L_OFF 2128
#define ED2_XBUF_FLOATS 2208

/* ---- what a lab build may change (tools/lab/mkvariant.py defines ED_LAB; the product build must not: tests/test_host_cpu.py
 * preprocesses this file as edison_amd/build.py compiles it and checks every value below). Everything that was measured and
 * decided in rounds 1-3 (transposes on the VALU / in LDS, twiddles in LDS, queue forms, barrier placements, phase ablations)
 * is no longer a switch: the decisions are in DESIGN.md section 4.1, the loop below is the one that ships. */
#if !defined(ED_LAB) && (defined(ED2_WPB) || defined(ED2_STAGGER_SLEEP) || defined(ED2_PRIO) || defined(ED2_STAMP))
#error "ED2_* lab knob defined without ED_LAB: the product build has no knobs (tools/lab/mkvariant.py builds lab variants)"
#endif
#if defined(ED_LAB)
/* a lab build says so: the product library exports no ed_lab_build_* symbol (tests/test_host_cpu.py) */
extern "C" { extern const int ed_lab_build_mfcc; const int ed_lab_build_mfcc = 1; }
#endif
#ifndef ED2_WPB
#define ED2_WPB 12  /* wavefronts per workgroup = per CU: 3 per SIMD (160 VGPRs, 116 KB LDS) */
#endif
#ifndef ED2_STAGGER_SLEEP
#define ED2_STAGGER_SLEEP 8 /* s_sleep units (64 cycles each) between the first sample loads of the wave groups, see the prologue */
#endif
/* Wave priorities (round 3): s_setprio at the phase boundaries of the loop, two bits per boundary (boundary i = end of phase i:
 * 0 unpack + loads of the next pair, 1 pass 1, 2 transpose 1, 3 pass 2, 4 transpose 2 (LDS), 5 pass 3, 6 split, 7 spectrum to
 * LDS, 8 mel, 9 fold, 10 DCT (LDS), 11 store). A SIMD issues from its oldest ready wave; with equal priorities the three waves
 * of a SIMD end up in the same phase (all in their arithmetic, then all waiting for LDS) -- a priority that RISES with the
 * progress through a pair (0 in pass 1, 1 in pass 2, 2 in pass 3, 3 from the split to the next pair's loads) lets the wave that
 * is ahead stay ahead, so the waves spread over the phases and one wave's LDS phases lie under the others' arithmetic:
 * +5.3 ... +7.0 % (interleaved A/B on six boxes, profiles/r03_wave_priorities.txt; bit-identical). 0 = none. */
#ifndef ED2_PRIO
#define ED2_PRIO 0xfffa50
#endif
/* 1 / 2: diagnostic builds with s_memtime stamps (1: at every phase boundary, 2: around the loop only); per-wave cycle sums go
 * to a debug buffer that nothing else reads (tools/lab/stamp_run.py). The product build contains no stamp. */
#ifndef ED2_STAMP
#define ED2_STAMP 0
#endif
#if ED2_STAMP
#define ED2_NPH 19
__device__ unsigned long long *g_ed2_dbg = nullptr;
extern "C" void ed_set_debug_buffer(void *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ed2_dbg), &p, sizeof(p)); }
/* launches in flight together (tools/lab/stamp_overlap.py) stamp into separate slots of the debug buffer: the slot of a launch is
 * where its output lies inside one allocation of n_slots outputs of slot_bytes each; slot_waves = stamp records per slot */
__device__ unsigned long long g_ed2_slot[4] = {0, 0, 0, 0}; /* base, slot_bytes, n_slots, slot_waves */
extern "C" void ed_set_debug_slots(void *base, unsigned long long slot_bytes, unsigned long long n_slots, unsigned long long slot_waves)
{
	const unsigned long long v[4] = {(unsigned long long)base, slot_bytes, n_slots, slot_waves};
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_ed2_slot), v, sizeof(v));
}
__device__ __forceinline__ unsigned long long ed2_now()
{
	unsigned long long t;
	__builtin_amdgcn_sched_barrier(0);
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
	__builtin_amdgcn_sched_barrier(0);
	return t;
}
#if ED2_STAMP == 1
#define ED2_ST(i) { const unsigned long long n_ = ed2_now(); ph[i] += n_ - tlast; tlast = n_; }
#else
#define ED2_ST(i)
#endif
#elif ED2_PRIO
#define ED2_ST(i) __builtin_amdgcn_s_setprio((ED2_PRIO >> (2 * (i))) & 3);
#else
#define ED2_ST(i)
#endif

/* The two frames of pair `pr`: A = 2 pr, B = 2 pr + 1 (the last pair of an odd batch repeats A). PLAIN: one group, frame f
 * starts at f * frame_step; otherwise f = (g, i) with ONE division per pair, B follows A by increment. LIST: group g's first frame
 * is list->audio[g] instead of audio + g * group_stride; (g, i) of frame A are handed back for the store. */
template <bool PLAIN, bool LIST>
__device__ __forceinline__ void ed_pair_ptrs(const ed_mfcc_args_t &a, const ed_mfcc_list_t *list, uint32_t pr, const int16_t *&pa, const int16_t *&pb,
                                             uint32_t &g_out, uint32_t &i_out)
{
	const uint32_t fA = 2 * pr;
	const bool haveB = fA + 1 < (uint32_t)a.n_frames;
	if (PLAIN)
	{
		pa = a.audio + (int64_t)fA * a.frame_step;
		pb = haveB ? pa + a.frame_step : pa;
	}
	else
	{
		const uint32_t fpg = (uint32_t)a.frames_per_group;
		const uint32_t g = fA / fpg, i = fA - g * fpg;
		if (LIST)
		{
			pa = list->audio[g] + (int64_t)i * a.frame_step;
			pb = !haveB ? pa : (i + 1 < fpg ? pa + a.frame_step : list->audio[g + 1 < ED_MFCC_LIST_MAX ? g + 1 : g]);
			g_out = g; i_out = i;
		}
		else
		{
			pa = a.audio + ((int64_t)g * a.group_stride + (int64_t)i * a.frame_step);
			pb = !haveB ? pa : (i + 1 < fpg ? pa + a.frame_step : a.audio + (int64_t)(g + 1) * a.group_stride);
		}
	}
}

/*
 * Work distribution. One workgroup of ED2_WPB wavefronts per CU owns a contiguous slice of the frame pairs and hands
 * them to its waves through a counter in LDS (ds_add_rtn_u32): a wave that runs faster simply draws more pairs. With a
 * static stride per wave the kernel drained unevenly -- the SIMD arbitrates VALU issue by age, so the waves of the
 * workgroups dispatched first ran their pairs in 3.0 us each, the youngest third in 4.6 us, and the last 25 % of the
 * launch ran with a third of the waves (profiles/r02_mfcc_timeline_static_stride.txt). A wave owns TWO pairs at a time -- the one
 * it computes and the one whose samples it prefetches: the draw (one exec-masked ds_add_rtn_u32 in asm; the builtin atomic goes
 * through the compiler's wave-aggregation code and waits on the spot) is issued at the top of an iteration, read behind pass 1
 * (~450 cycles later, never waited for), and the drawn pair's sample loads go out there. With the draw a whole iteration ahead
 * the last pairs of a CU sat reserved in its slowest waves while the others had retired.
 * Across CUs the slices are static: a rank-based pool at the end of the batch (one device atomic per workgroup) levels the
 * workgroups' finishing times and buys nothing -- the board sits at its power cap and takes the recovered idle time back as
 * clock (profiles/r04_mfcc_launch_structure_notes.txt).
 */
template <bool ALIGNED, bool PLAIN, int NLO, int NHI, bool LIST, bool WINDOW = false>
__device__ __forceinline__ void ed_mfcc2_body(const ed_mfcc_args_t &args, const ed_mfcc_tables_t *__restrict__ tab, const ed_mfcc_list_t *list)
{
	extern __shared__ __attribute__((aligned(16))) float smem[];
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const float4 *dctl = reinterpret_cast<const float4 *>(smem);                     /* [2][64] x 4 coefficients */
	const float2 *tpl = reinterpret_cast<const float2 *>(smem + 512);                /* [4][64] split twiddles   */
	const float4 *melw4 = reinterpret_cast<const float4 *>(smem + ED_FIXTAB_FLOATS); /* [NLO+NHI][64] quads      */
	float *xbuf = smem + ED_FIXTAB_FLOATS + (NLO + NHI) * 256 + wave * ED2_XBUF_FLOATS; /* wave-private */
	unsigned *queue = reinterpret_cast<unsigned *>(smem + ED_FIXTAB_FLOATS + (NLO + NHI) * 256 + ED2_WPB * ED2_XBUF_FLOATS);
	const uint32_t queue_addr = (uint32_t)(sizeof(float) * (ED_FIXTAB_FLOATS + (NLO + NHI) * 256 + ED2_WPB * ED2_XBUF_FLOATS)); /* its LDS byte address (dynamic LDS starts at 0: the kernel has no static LDS) */
#if ED2_STAMP
	unsigned long long rt_entry;
	asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_entry) :: "memory");
#endif

	/* this workgroup's slice of the pairs; its first WPB pairs are the waves' first pairs, the rest is drawn from the queue */
	const uint32_t n_frames = (uint32_t)args.n_frames;
	const uint32_t n_pairs = (n_frames + 1) >> 1;
	const uint32_t s0 = (uint32_t)(((uint64_t)blockIdx.x * n_pairs) / gridDim.x);
	const uint32_t cnt = (uint32_t)(((uint64_t)(blockIdx.x + 1) * n_pairs) / gridDim.x) - s0;
	uint32_t i_cur = wave, i_next;
	uint32_t g_cur = 0, gi_cur = 0, g_nxt = 0, gi_nxt = 0; /* LIST: batch and frame-in-batch of the current / the prefetched pair's frame A */
	uint32_t rawA[8], rawB[8];
	/* No workgroup barrier behind the table staging: the first four waves (one per SIMD) stage the tables and raise a counter in
	 * LDS; every wave checks that counter once, in front of its first use of the tables (the split of its first pair), by which
	 * time it has long been raised. Queue and staging counter are initialised behind a barrier that costs nothing -- no wave has
	 * asked memory for anything yet -- and wave groups 1 and 2 issue their first sample loads a few hundred cycles later than
	 * group 0, so that group 0's samples are not queued behind theirs: a SIMD starts computing ~1.5 us after launch instead of
	 * when the last of the CU's 48 KB has arrived. (A barrier in front of the first split instead: -1.4 %, the fast waves wait
	 * there for the slowest.) */
	if (threadIdx.x == 0) { queue[0] = ED2_WPB; queue[1] = 0; }
	__syncthreads();
	if (wave >= 4) __builtin_amdgcn_s_sleep(ED2_STAGGER_SLEEP);
	if (wave >= 8) __builtin_amdgcn_s_sleep(ED2_STAGGER_SLEEP);
	if (i_cur < cnt)
	{
		const int16_t *pa, *pb;
		ed_pair_ptrs<PLAIN, LIST>(args, list, s0 + i_cur, pa, pb, g_cur, gi_cur);
		ed_load_frame<ALIGNED>(pa, lane, rawA);
		ed_load_frame<ALIGNED>(pb, lane, rawB);
	}
	/* every table load of the prologue goes in flight before the first wait */
	float t1r[8], t1i[8], t2r[8], t2i[8];
#pragma unroll
	for (int q = 1; q < 8; q++)
	{
		const float2 a = *reinterpret_cast<const float2 *>(&tab->tw1[q][lane][0]);
		const float2 b = *reinterpret_cast<const float2 *>(&tab->tw2[q][lane][0]);
		t1r[q] = a.x; t1i[q] = a.y; t2r[q] = b.x; t2i[q] = b.y;
	}
	const int mel_slo4 = tab->mel_slo4[lane], mel_shi4 = tab->mel_shi4[lane];
	const int band = tab->mel_band[lane]; /* this column's narrow band b; its wide band is 31 - b */
	const int mel_half = tab->mel_half[lane];
	const float log_offset = tab->log_offset;
	const bool do_log = tab->always_log || args.use_log;
	{
		const float4 *src = reinterpret_cast<const float4 *>(&tab->dct4[0][0][0]);
		float4 *dst = reinterpret_cast<float4 *>(smem);
		constexpr int n4 = (ED_FIXTAB_FLOATS + (NLO + NHI) * 256) / 4;
		/* lanes with mel_half = 1 read the two 16-byte halves of a spectrum quad in the opposite order (see the mel
		 * stage), so their weight quads are stored (z, w, x, y); quad t belongs to lane t & 63 (ED_FIXTAB_FLOATS / 4 is
		 * a multiple of 64) and threadIdx.x + k * blockDim.x keeps that lane: the flag is this thread's own mel_half */
		static_assert((64 * ED2_WPB) % 64 == 0 && (ED_FIXTAB_FLOATS / 4) % 64 == 0, "weight quad <-> lane mapping");
		if (wave < 4)
		{
			/* thread t of the first 256 stages quads t, t + 256, t + 512: quad <-> lane mapping as above (256 % 64 == 0) */
			constexpr int NT = (n4 + 255) / 256;
			float4 tv[NT];
#pragma unroll
			for (int k = 0; k < NT; k++)
			{
				const int t = threadIdx.x + k * 256;
				tv[k] = src[t < n4 ? t : 0];
			}
#pragma unroll
			for (int k = 0; k < NT; k++)
			{
				const int t = threadIdx.x + k * 256;
				float4 v = tv[k];
				if (t >= ED_FIXTAB_FLOATS / 4 && mel_half) v = make_float4(v.z, v.w, v.x, v.y);
				if (t < n4) dst[t] = v;
			}
			/* this wave's stores are complete (release) before its count: a wave's DS operations execute in order */
			if (lane == 0) __hip_atomic_fetch_add(queue + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
	bool staged = false;
	/* where this lane puts its DCT input (float index into Lb2 = float2 u[16] | v[16]): rows 0/1 hold frame A's
	 * u/v of the column's band, rows 2/3 frame B's */
	const int l_idx = 2 * (16 * ((lane >> 4) & 1) + band) + (lane >> 5);
	const int k0 = ED_K0(lane);
	const int k0p = (64 - k0) & 63;
	const int pull = k0p << 2;
	const int hi3 = lane >> 3, lo3 = lane & 7;
	float4 *xc4 = reinterpret_cast<float4 *>(xbuf);
	ed_f2 *S2 = reinterpret_cast<ed_f2 *>(xbuf + ED2_S_OFF);
	if (lane < 3) S2[513 + lane] = ed_splat(0.0f);

#if ED2_STAMP
	unsigned long long ph[ED2_NPH];
	for (int i_ = 0; i_ < ED2_NPH; i_++) ph[i_] = 0;
	unsigned long long rt0, rt1;
	asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0) :: "memory");
	const unsigned long long tfirst = ed2_now();
	unsigned long long tlast = tfirst;
#endif
	ed_f2 re[8], im[8];
	if (i_cur < cnt)
	{
#pragma unroll
		for (int a = 0; a < 8; a++)
		{
			re[a] = ed_mk2((float)(int16_t)(rawA[a] & 0xffffu), (float)(int16_t)(rawB[a] & 0xffffu));
			im[a] = ed_mk2((float)(int16_t)(rawA[a] >> 16), (float)(int16_t)(rawB[a] >> 16));
		}
		if (WINDOW) /* variant TF: tf.signal.stft's Hann window on the float32 samples, (w[2m], w[2m+1]) per packed point, the same for both frames */
		{
#pragma unroll
			for (int a = 0; a < 8; a++)
			{
				const float2 w = *reinterpret_cast<const float2 *>(&tab->window2[lane + 64 * a][0]);
				re[a] = re[a] * ed_splat(w.x);
				im[a] = im[a] * ed_splat(w.y);
			}
		}
	}
	while (i_cur < cnt)
	{
		const uint32_t fA = 2 * (s0 + i_cur);
		const bool haveB = fA + 1 < n_frames;
		/* ---- 1. draw the next pair (read behind pass 1) */
		uint32_t drawn = 0;
		if (lane == 0)
			asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(drawn) : "v"(queue_addr), "v"(1u) : "memory");

		ED2_ST(0)
		/* ---- 2a. pass 1 + twiddle W512^(lane*p) */
		ed_radix8_2(re, im);
#pragma unroll
		for (int q = 1; q < 8; q++)
		{
			const ed_f2 wr = ed_splat(t1r[q]), wi = ed_splat(t1i[q]);
			const ed_f2 xr = re[q], xi = im[q];
			re[q] = xr * wr - xi * wi;
			im[q] = xr * wi + xi * wr;
		}
		{
			/* the draw issued at the top has returned (~450 cycles of pass 1 ago): the drawn pair's samples go in flight now
			 * and have the rest of the iteration to arrive. Unconditional -- a conditional load makes the frame registers a
			 * merge of two definitions and costs 16 copies per iteration (-12 % measured): past the end of the slice the
			 * slice's last pair is re-read from L2 and thrown away. */
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(drawn));
			i_next = __builtin_amdgcn_readfirstlane(drawn);
			const int16_t *pa, *pb;
			ed_pair_ptrs<PLAIN, LIST>(args, list, s0 + (i_next < cnt ? i_next : cnt - 1), pa, pb, g_nxt, gi_nxt);
			ed_load_frame<ALIGNED>(pa, lane, rawA);
			ed_load_frame<ALIGNED>(pb, lane, rawB);
		}
		ED2_ST(1)
		/* transpose 1 on the VALU (lane bits 3-5): with both transposes in LDS the LDS pipe saturates (-5.5 % with the priorities on) */
		ed_transpose8_2<3, 4, 5>(re, lane);
		ed_transpose8_2<3, 4, 5>(im, lane);

		ED2_ST(2)
		/* ---- 2b. pass 2 + twiddle W64^(c*q) */
		ed_radix8_2(re, im);
#pragma unroll
		for (int q = 1; q < 8; q++)
		{
			const ed_f2 wr = ed_splat(t2r[q]), wi = ed_splat(t2i[q]);
			const ed_f2 xr = re[q], xi = im[q];
			re[q] = xr * wr - xi * wi;
			im[q] = xr * wi + xi * wr;
		}
		ED2_ST(3)
		/* transpose 2 through LDS, both frames in one 16-byte slot: (lane 8p+c, reg q) -> (lane p+8q, reg c); slot
		 * 66c + p + 8q keeps the ds_write_b128 (8-lane groups, stride 66 slots = 8 banks mod 64) and the ds_read_b128
		 * (consecutive slots) free of bank conflicts */
#pragma unroll
		for (int q = 0; q < 8; q++) xc4[66 * lo3 + hi3 + 8 * q] = make_float4(re[q].x, re[q].y, im[q].x, im[q].y); /* pairs stay pairs */
		ed_wave_sync();
#pragma unroll
		for (int c = 0; c < 8; c++)
		{
			const float4 v = xc4[66 * c + lane];
			re[c] = ed_mk2(v.x, v.y); im[c] = ed_mk2(v.z, v.w);
		}
		ed_wave_sync();

		ED2_ST(4)
		/* ---- 2c. pass 3: reg r holds Z[k0 + 64r] of both frames */
		ed_radix8_2(re, im);

		ED2_ST(5)
		if (!staged)
		{
			/* first use of the LDS tables (split twiddles, then mel, DCT): all four staging waves must have counted */
			while ((uint32_t)__builtin_amdgcn_readfirstlane(__hip_atomic_load(queue + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < 4u)
				__builtin_amdgcn_s_sleep(1);
			staged = true;
		}
		/* ---- 3. real-FFT split (see ed_mfcc_kernel); the partner values come per frame through ds_bpermute */
		ed_f2 slo[4], shi[4];
		ed_f2 pzr_[4], pzi_[4];
#pragma unroll
		for (int m = 0; m < 4; m++)
		{
			pzr_[m].x = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(re[7 - m].x)));
			pzr_[m].y = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(re[7 - m].y)));
			pzi_[m].x = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(im[7 - m].x)));
			pzi_[m].y = __int_as_float(__builtin_amdgcn_ds_bpermute(pull, __float_as_int(im[7 - m].y)));
		}
		/* lane 0 is its own partner, one register further up: ONE exec-masked block of 16 v_mov_b32 (2.5 cycles each)
		 * instead of 16 v_cndmask_b32_e64 (4.4 each) -- the empty asm keeps the compiler from if-converting it back */
		if (lane == 0)
		{
			asm volatile("");
#pragma unroll
			for (int m = 0; m < 4; m++) { pzr_[m] = re[(8 - m) & 7]; pzi_[m] = im[(8 - m) & 7]; }
		}
#pragma unroll
		for (int m = 0; m < 4; m++)
		{
			ed_f2 pzr = pzr_[m], pzi = pzi_[m];
			const float2 tw = tpl[64 * m + lane];
			const ed_f2 twx = ed_splat(tw.x), twy = ed_splat(tw.y);
			const ed_f2 ar = re[m] + pzr, ai = im[m] - pzi;
			const ed_f2 br = re[m] - pzr, bi = im[m] + pzi;
			const ed_f2 tr = twx * bi + twy * br;
			const ed_f2 ti = twy * bi - twx * br;
			const ed_f2 xr = ar + tr, xi = ai + ti;
			const ed_f2 yr = ar - tr, yi = ai - ti;
			const ed_f2 e0 = xr * xr + xi * xi, e1 = yr * yr + yi * yi;
			slo[m] = ed_mk2(__builtin_amdgcn_sqrtf(e0.x), __builtin_amdgcn_sqrtf(e0.y));
			shi[m] = ed_mk2(__builtin_amdgcn_sqrtf(e1.x), __builtin_amdgcn_sqrtf(e1.y));
		}
		const ed_f2 e256 = re[4] * re[4] + im[4] * im[4];
		const ed_f2 s256 = ed_mk2(2.0f * __builtin_amdgcn_sqrtf(e256.x), 2.0f * __builtin_amdgcn_sqrtf(e256.y));

		ED2_ST(6)
		/* ---- 4. both spectra to LDS, interleaved: S2[k] = (|2X_A[k]|, |2X_B[k]|) */
#pragma unroll
		for (int m = 0; m < 4; m++)
		{
			S2[k0 + 64 * m] = slo[m];
			S2[512 - k0 - 64 * m] = shi[m];
		}
		if (lane == 0) S2[256] = s256;
		ed_wave_sync();

		ED2_ST(7)
		/* ---- 5. mel filterbank, balanced as above; a spectrum quad of both frames is two 16-byte reads */
		const float4 *S4 = reinterpret_cast<const float4 *>(S2);
		/* Every ds_read_b128 serves 16 lanes from 16 slots of 16 bytes; if all lanes took the first half of their quad,
		 * only the even slots would be used. Half of the lanes therefore start with the second half (their weights are
		 * swapped to match); tables.c picks them, and the column order, to minimise the passes (mel_half, mel_band). */
		const int half = mel_half;
		const int qlo_a = 2 * mel_slo4 + half, qlo_b = 2 * mel_slo4 + 1 - half;
		const int qhi_a = 2 * mel_shi4 + half, qhi_b = 2 * mel_shi4 + 1 - half;
		ed_f2 alo0 = ed_splat(0.0f), alo1 = alo0, ahi0 = alo0, ahi1 = alo0;
#pragma unroll
		for (int t = 0; t < NLO; t++)
		{
			const float4 sa = S4[qlo_a + 2 * t], sb = S4[qlo_b + 2 * t], w = melw4[t * 64 + lane];
			alo0 = ed_fma2(ed_mk2(sa.x, sa.y), ed_splat(w.x), alo0); alo1 = ed_fma2(ed_mk2(sa.z, sa.w), ed_splat(w.y), alo1);
			alo0 = ed_fma2(ed_mk2(sb.x, sb.y), ed_splat(w.z), alo0); alo1 = ed_fma2(ed_mk2(sb.z, sb.w), ed_splat(w.w), alo1);
		}
#pragma unroll
		for (int t = 0; t < NHI; t++)
		{
			if (t % 2 == 0) __builtin_amdgcn_sched_barrier(0); /* bounds the registers this stage holds in flight */
			const float4 sa = S4[qhi_a + 2 * t], sb = S4[qhi_b + 2 * t], w = melw4[(NLO + t) * 64 + lane];
			ahi0 = ed_fma2(ed_mk2(sa.x, sa.y), ed_splat(w.x), ahi0); ahi1 = ed_fma2(ed_mk2(sa.z, sa.w), ed_splat(w.y), ahi1);
			ahi0 = ed_fma2(ed_mk2(sb.x, sb.y), ed_splat(w.z), ahi0); ahi1 = ed_fma2(ed_mk2(sb.z, sb.w), ed_splat(w.w), ahi1);
		}
		__builtin_amdgcn_sched_barrier(0);
		/* The four row-quarters of a band are summed with the swap instructions, and because a swap exchanges halves
		 * of TWO registers, two sums are folded at once: first frame A's and frame B's partial sums over the wave
		 * halves (lanes 0..31 then hold A, 32..63 B), then the narrow and the wide band over the row pairs. One
		 * register ends up with  row 0: band b of A,  row 1: band 31-b of A,  row 2: b of B,  row 3: 31-b of B. */
		ED2_ST(8)
		const ed_f2 plo = alo0 + alo1, phi = ahi0 + ahi1;
		float t = ed_fold_rows(ed_fold_halves(plo.x, plo.y), ed_fold_halves(phi.x, phi.y));
		if (do_log) { asm volatile(""); t = __logf(t + log_offset); } /* the empty asm keeps this a branch (wave-uniform) */

		ED2_ST(9)
		/* WINDOW: the next pair's window values are asked for here, behind the mel stage and its folds, where the fewest registers are live; the lane number is
		 * made opaque so that the sixteen values are not kept in registers across the loop (the kernel sits at the 160 that three waves per SIMD allow) */
		float2 wv[8];
		if (WINDOW)
		{
			int wl = lane;
			asm volatile("" : "+v"(wl));
#pragma unroll
			for (int a = 0; a < 8; a++) wv[a] = *reinterpret_cast<const float2 *>(&tab->window2[wl + 64 * a][0]);
		}
		/* ---- 6. DCT-II through cos symmetry, both frames: u = L[b] + L[31-b] (even rows), v = L[b] - L[31-b] (odd) */
		ed_f2 *Lb2 = reinterpret_cast<ed_f2 *>(xbuf + ED2_L_OFF); /* u[16] | v[16] as float2 (A, B) */
		{
			const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
			const float tb = __uint_as_float(r[0]), to = __uint_as_float(r[1]); /* [r0 r0 r2 r2], [r1 r1 r3 r3] */
			reinterpret_cast<float *>(Lb2)[l_idx] = (lane & 16) ? tb - to : tb + to;
		}
		ed_wave_sync();
		const float4 *L4 = reinterpret_cast<const float4 *>(Lb2 + 16 * (lane & 1) + 8 * (lane >> 5));
		const float4 v0 = L4[0], v1 = L4[1], v2 = L4[2], v3 = L4[3];
		const float4 w0 = dctl[lane], w1 = dctl[64 + lane];
		ed_f2 d = ed_mk2(v0.x, v0.y) * ed_splat(w0.x), d1 = ed_mk2(v2.x, v2.y) * ed_splat(w1.x);
		d = ed_fma2(ed_mk2(v0.z, v0.w), ed_splat(w0.y), d); d1 = ed_fma2(ed_mk2(v2.z, v2.w), ed_splat(w1.y), d1);
		d = ed_fma2(ed_mk2(v1.x, v1.y), ed_splat(w0.z), d); d1 = ed_fma2(ed_mk2(v3.x, v3.y), ed_splat(w1.z), d1);
		d = ed_fma2(ed_mk2(v1.z, v1.w), ed_splat(w0.w), d); d1 = ed_fma2(ed_mk2(v3.z, v3.w), ed_splat(w1.w), d1);
		d = d + d1;
		/* the two halves of the sum, again for both frames with one swap: lanes 0..31 get coefficient `lane` of
		 * frame A, lanes 32..63 coefficient `lane - 32` of frame B */
		const float coef = ed_fold_halves(d.x, d.y);
		ed_wave_sync(); /* Lb2 / S2 are rewritten by the next pair */

		ED2_ST(10)
		/* ---- 7. store: A's and B's rows are adjacent in memory, one instruction writes both */
		const int c = lane & 31;
		if (c < args.n_coef && (lane < 32 || haveB))
		{
			if (LIST)
			{
				/* frame A is row gi_cur of batch g_cur; frame B the next row, or row 0 of the next batch */
				const uint32_t fpg = (uint32_t)args.frames_per_group;
				const bool wrap = gi_cur + 1 >= fpg;
				const uint32_t gB = wrap ? (g_cur + 1 < ED_MFCC_LIST_MAX ? g_cur + 1 : g_cur) : g_cur, iB = wrap ? 0u : gi_cur + 1; /* (never past the table: a last odd frame has no B) */
				const bool isB = lane >= 32;
				const int64_t at = (int64_t)(isB ? iB : gi_cur) * args.n_coef + c;
				if (args.mfcc) { float *o = isB ? list->mfcc[gB] : list->mfcc[g_cur]; o[at] = coef; }
				if (args.feat) { int8_t *o = isB ? list->feat[gB] : list->feat[g_cur]; o[at] = (int8_t)__float2int_rn(fminf(fmaxf(coef * args.feat_scale, -128.0f), 127.0f)); }
			}
			else
			{
				const int64_t at = (int64_t)(fA + (lane >> 5)) * args.n_coef + c;
				if (args.mfcc) args.mfcc[at] = coef;
				if (args.feat) args.feat[at] = (int8_t)__float2int_rn(fminf(fmaxf(coef * args.feat_scale, -128.0f), 127.0f));
			}
		}
		ED2_ST(11)
		/* the next pair's samples (requested at the top of this iteration) become the floats the next iteration starts from */
#pragma unroll
		for (int a = 0; a < 8; a++)
		{
			re[a] = ed_mk2((float)(int16_t)(rawA[a] & 0xffffu), (float)(int16_t)(rawB[a] & 0xffffu));
			im[a] = ed_mk2((float)(int16_t)(rawA[a] >> 16), (float)(int16_t)(rawB[a] >> 16));
			if (WINDOW)
			{
				re[a] = re[a] * ed_splat(wv[a].x);
				im[a] = im[a] * ed_splat(wv[a].y);
			}
		}
		i_cur = i_next;
		if (LIST) { g_cur = g_nxt; gi_cur = gi_nxt; }
	}
#if ED2_STAMP
	asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1) :: "memory");
	ph[12] = ed2_now() - tfirst; ph[13] = rt1 - rt0; ph[14] = rt_entry; ph[15] = rt0; ph[16] = rt1;
	if (g_ed2_dbg && lane == 0)
	{
		size_t slot_off = 0;
		if (g_ed2_slot[1])
			slot_off = (size_t)((((unsigned long long)args.mfcc - g_ed2_slot[0]) / g_ed2_slot[1]) % g_ed2_slot[2]) * g_ed2_slot[3];
		unsigned long long *dbg = g_ed2_dbg + (slot_off + (size_t)(blockIdx.x * ED2_WPB + wave)) * ED2_NPH;
		ph[11] = i_cur; /* not a time: how far this wave got in the slice */
		{
			unsigned hw, xcc; /* which CU this wave ran on: HW_ID = wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ..., XCC_ID[3:0] */
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
			ph[17] = hw; ph[18] = xcc;
		}
		for (int i_ = 0; i_ < ED2_NPH; i_++) dbg[i_] = ph[i_];
	}
#endif
}

template <bool ALIGNED, bool PLAIN, int NLO, int NHI>
__global__ __launch_bounds__(64 * ED2_WPB) void ed_mfcc2_kernel(ed_mfcc_args_t args, const ed_mfcc_tables_t *__restrict__ tab)
{
	ed_mfcc2_body<ALIGNED, PLAIN, NLO, NHI, false>(args, tab, nullptr);
}

/* variant TF (windowed frames, mfcc_utils.py:201-253) on the same loop: the window multiplies the samples at the unpack, nothing else differs */
template <bool ALIGNED, bool PLAIN, int NLO, int NHI>
__global__ __launch_bounds__(64 * ED2_WPB) void ed_mfcc2_window_kernel(ed_mfcc_args_t args, const ed_mfcc_tables_t *__restrict__ tab)
{
	ed_mfcc2_body<ALIGNED, PLAIN, NLO, NHI, false, true>(args, tab, nullptr);
}

/* the same loop over a LIST of independent batches (edison_mfcc_batches_dev): one launch keeps the chip busy across them (what the
 * launch interface cannot do for separate launches: profiles/r05_mfcc_two_queues_notes.txt) */
template <bool ALIGNED, int NLO, int NHI>
__global__ __launch_bounds__(64 * ED2_WPB) void ed_mfcc2_list_kernel(ed_mfcc_args_t args, const ed_mfcc_tables_t *__restrict__ tab, ed_mfcc_list_t list)
{
	ed_mfcc2_body<ALIGNED, false, NLO, NHI, true>(args, tab, &list);
}

#if defined(ED_LAB)
static int g_ed_lab_launch_flags = 0; /* lab: hipExtLaunchKernel flags of the fast path (hipExtAnyOrderLaunch = 1) */
extern "C" void ed_lab_set_launch_flags(int f) { g_ed_lab_launch_flags = f; }
#endif
/* occupancy-derived grid sizes and "dynamic-LDS limit raised" flags, per DEVICE (0 = not asked yet; the attribute belongs to
 * the function on the current device, so two contexts on different GPUs of one process must each set it) */
static int g_mfcc_blocks_per_cu[16][2];
static int g_mfcc2_blocks_per_cu[16][16];

template <int NLO, int NHI>
static int ed_launch_mfcc_shape(const ed_mfcc_args_t *args, const ed_mfcc_tables_t *dev_tab, int stages, int n_cu,
                                hipStream_t stream, int *blocks_per_cu)
{
	const size_t lds = sizeof(float) * (ED_FIXTAB_FLOATS + (NLO + NHI) * 256 + ED_WPB * ED_XBUF_FLOATS);
	if (*blocks_per_cu <= 0)
	{
		/* persistent grid = exactly what is resident; sized once from the fast kernel's occupancy */
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ed_mfcc_kernel<false, true, NLO, NHI>, 64 * ED_WPB, lds) != hipSuccess || nb < 1)
			nb = 1;
		const char *env = getenv("ED_MFCC_BLOCKS_PER_CU"); /* tuning knob: cap the persistent grid */
		if (env && atoi(env) > 0 && atoi(env) < nb) nb = atoi(env);
		*blocks_per_cu = nb;
	}
	int64_t blocks = (args->n_frames + ED_WPB - 1) / ED_WPB;
	const int64_t cap = (int64_t)n_cu * *blocks_per_cu;
	if (blocks > cap) blocks = cap;
	/* 4-byte loads need every frame start 4-byte aligned */
	const bool aligned = ((reinterpret_cast<uintptr_t>(args->audio) & 3) == 0) && (args->frame_step % 2 == 0) &&
	                     (args->group_stride % 2 == 0);
	static const int one_frame = getenv("ED_MFCC_ONE_FRAME") ? atoi(getenv("ED_MFCC_ONE_FRAME")) : 0; /* A/B knob */
	if (args->window && (stages || one_frame))
	{
		/* variant TF (windowed frames) with stage dumps: the one-frame kernel's WINDOW instances */
		dim3 grid((unsigned)blocks), block(64 * ED_WPB);
		if (stages)
		{
			if (aligned) hipLaunchKernelGGL((ed_mfcc_kernel<true, true, NLO, NHI, true>), grid, block, lds, stream, *args, dev_tab);
			else hipLaunchKernelGGL((ed_mfcc_kernel<true, false, NLO, NHI, true>), grid, block, lds, stream, *args, dev_tab);
		}
		else
		{
			if (aligned) hipLaunchKernelGGL((ed_mfcc_kernel<false, true, NLO, NHI, true>), grid, block, lds, stream, *args, dev_tab);
			else hipLaunchKernelGGL((ed_mfcc_kernel<false, false, NLO, NHI, true>), grid, block, lds, stream, *args, dev_tab);
		}
		return (int)hipGetLastError();
	}
	if (!stages && !one_frame)
	{
		/* the fast path: two frames per wavefront in packed fp32, one ED2_WPB-wave workgroup per CU */
		const size_t lds2 = sizeof(float) * (ED_FIXTAB_FLOATS + (NLO + NHI) * 256 + ED2_WPB * ED2_XBUF_FLOATS) + 16 /* queue */;
		const bool plain = args->frames_per_group >= args->n_frames;
		const void *fn = aligned ? (plain ? (const void *)ed_mfcc2_kernel<true, true, NLO, NHI> : (const void *)ed_mfcc2_kernel<true, false, NLO, NHI>)
		                         : (plain ? (const void *)ed_mfcc2_kernel<false, true, NLO, NHI> : (const void *)ed_mfcc2_kernel<false, false, NLO, NHI>);
		if (args->window)
			fn = aligned ? (plain ? (const void *)ed_mfcc2_window_kernel<true, true, NLO, NHI> : (const void *)ed_mfcc2_window_kernel<true, false, NLO, NHI>)
			             : (plain ? (const void *)ed_mfcc2_window_kernel<false, true, NLO, NHI> : (const void *)ed_mfcc2_window_kernel<false, false, NLO, NHI>);
		int dev_ = 0;
		(void)hipGetDevice(&dev_);
		int *bpc2 = &g_mfcc2_blocks_per_cu[dev_ & 15][(args->window ? 8 : 0) + (NLO == 2 ? 0 : 4) + (aligned ? 2 : 0) + (plain ? 1 : 0)];
		if (*bpc2 <= 0)
		{
			/* more than 64 KB of dynamic LDS has to be asked for, once per kernel instance */
			if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) return (int)hipGetLastError();
			int nb = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * ED2_WPB, lds2) != hipSuccess || nb < 1) nb = 1;
			const char *env = getenv("ED_MFCC_BLOCKS_PER_CU");
			if (env && atoi(env) > 0 && atoi(env) < nb) nb = atoi(env);
			*bpc2 = nb;
		}
		const int64_t n_pairs = (args->n_frames + 1) / 2;
		int64_t blocks2 = (n_pairs + ED2_WPB - 1) / ED2_WPB;
		if (blocks2 > (int64_t)n_cu * *bpc2) blocks2 = (int64_t)n_cu * *bpc2;
		void *kargs[] = {(void *)args, (void *)&dev_tab};
#if defined(ED_LAB)
		if (g_ed_lab_launch_flags)
			return (int)hipExtLaunchKernel(fn, dim3((unsigned)blocks2), dim3(64 * ED2_WPB), kargs, lds2, stream, nullptr, nullptr, g_ed_lab_launch_flags);
#endif
		return (int)hipLaunchKernel(fn, dim3((unsigned)blocks2), dim3(64 * ED2_WPB), kargs, lds2, stream);
	}
	dim3 grid((unsigned)blocks), block(64 * ED_WPB);
	if (stages)
	{
		if (aligned) hipLaunchKernelGGL((ed_mfcc_kernel<true, true, NLO, NHI>), grid, block, lds, stream, *args, dev_tab);
		else hipLaunchKernelGGL((ed_mfcc_kernel<true, false, NLO, NHI>), grid, block, lds, stream, *args, dev_tab);
	}
	else
	{
		if (aligned) hipLaunchKernelGGL((ed_mfcc_kernel<false, true, NLO, NHI>), grid, block, lds, stream, *args, dev_tab);
		else hipLaunchKernelGGL((ed_mfcc_kernel<false, false, NLO, NHI>), grid, block, lds, stream, *args, dev_tab);
	}
	return (int)hipGetLastError();
}

static int g_mfcc2_list_blocks_per_cu[16][4];

template <int NLO, int NHI>
static int ed_launch_mfcc_list_shape(const ed_mfcc_args_t *args, const ed_mfcc_list_t *list, int n_batches, const ed_mfcc_tables_t *dev_tab, int n_cu, hipStream_t stream)
{
	const size_t lds2 = sizeof(float) * (ED_FIXTAB_FLOATS + (NLO + NHI) * 256 + ED2_WPB * ED2_XBUF_FLOATS) + 16 /* queue */;
	bool aligned = args->frame_step % 2 == 0;
	for (int b = 0; b < n_batches; b++) aligned = aligned && (reinterpret_cast<uintptr_t>(list->audio[b]) & 3) == 0;
	const void *fn = aligned ? (const void *)ed_mfcc2_list_kernel<true, NLO, NHI> : (const void *)ed_mfcc2_list_kernel<false, NLO, NHI>;
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	int *bpc = &g_mfcc2_list_blocks_per_cu[dev_ & 15][(NLO == 2 ? 0 : 2) + (aligned ? 1 : 0)];
	if (*bpc <= 0)
	{
		if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) return (int)hipGetLastError();
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * ED2_WPB, lds2) != hipSuccess || nb < 1) nb = 1;
		*bpc = nb;
	}
	const int64_t n_pairs = (args->n_frames + 1) / 2;
	int64_t blocks = (n_pairs + ED2_WPB - 1) / ED2_WPB;
	if (blocks > (int64_t)n_cu * *bpc) blocks = (int64_t)n_cu * *bpc;
	void *kargs[] = {(void *)args, (void *)&dev_tab, (void *)list};
	return (int)hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(64 * ED2_WPB), kargs, lds2, stream);
}

/* n_batches (1 .. ED_MFCC_LIST_MAX) batches of args->frames_per_group frames each as the groups of one launch; args->n_frames =
 * n_batches * frames_per_group; args->audio / mfcc / feat are used as flags only (NULL: that output is not written) */
extern "C" int ed_launch_mfcc_list(const ed_mfcc_args_t *args, const ed_mfcc_list_t *list, int n_batches, const ed_mfcc_tables_t *dev_tab, int n_cu,
                                   hipStream_t stream)
{
	if (args->n_frames <= 0) return 0;
	if (n_batches < 1 || n_batches > ED_MFCC_LIST_MAX || args->frames_per_group < 1 || args->n_frames != (int64_t)n_batches * args->frames_per_group)
		return (int)hipErrorInvalidValue;
	if (args->mel_NLO == 2 && args->mel_NHI == 5) return ed_launch_mfcc_list_shape<2, 5>(args, list, n_batches, dev_tab, n_cu, stream);
	if (args->mel_NLO == ED_MEL_NLO_MAX && args->mel_NHI == ED_MEL_NHI_MAX)
		return ed_launch_mfcc_list_shape<ED_MEL_NLO_MAX, ED_MEL_NHI_MAX>(args, list, n_batches, dev_tab, n_cu, stream);
	return (int)hipErrorInvalidValue;
}

extern "C" int ed_launch_mfcc(const ed_mfcc_args_t *args, const ed_mfcc_tables_t *dev_tab, int stages, int n_cu,
                              hipStream_t stream)
{
	if (args->n_frames <= 0) return 0;
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	dev_ &= 15;
	/* the two table shapes tables.c produces */
	if (args->mel_NLO == 2 && args->mel_NHI == 5)
		return ed_launch_mfcc_shape<2, 5>(args, dev_tab, stages, n_cu, stream, &g_mfcc_blocks_per_cu[dev_][0]);
	if (args->mel_NLO == ED_MEL_NLO_MAX && args->mel_NHI == ED_MEL_NHI_MAX)
		return ed_launch_mfcc_shape<ED_MEL_NLO_MAX, ED_MEL_NHI_MAX>(args, dev_tab, stages, n_cu, stream, &g_mfcc_blocks_per_cu[dev_][1]);
	return (int)hipErrorInvalidValue;
}
