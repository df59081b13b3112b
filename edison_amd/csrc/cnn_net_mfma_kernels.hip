/*
 * cnn_net_mfma_kernels.hip -- ANY sequential NNoM int8 graph the planner accepts, with every Conv2D / Dense layer on the
 * gfx950 matrix cores (v_mfma_i32_32x32x32_i8, v_mfma_i32_16x16x64_i8). The GPU's model_run() (nnom.c:975-1040) for batch scoring; the
 * layer-by-layer kernel of cnn_net_kernels.hip stays for per-layer dumps and for graphs whose plan does not fit here.
 *
 * Arithmetic: exactly the reference's -- out = sat8((sum x*w + (bias << BL) + NN_ROUND(RS)) >> RS), ReLU as a tail
 * activation, max-pool over the part of the window inside the image, arm_softmax_q7's portable branch, first-maximum
 * argmax (citations in cnn_net_kernels.hip); integer sums are exact in any order, so the results are bit-identical.
 *
 * Scheme (ed_mm_plan_t, model_net_mm.c): implicit GEMM D[out_channel][pixel] with k = (kernel row, 16-byte chunk of the
 * row's contiguous kw * C_in input bytes). The consumer layer dictates how its input lies in LDS: zero-padded so that no
 * tap test is needed (and, where C_in is an even multiple of 16, with a 16-byte gap behind every pixel that takes neighbouring
 * pixels off the same LDS banks); the producing layer's epilogue writes straight into that layout. Layers whose C_in is not a
 * multiple of 16 read from an expanded copy with one aligned record per (input row, output x) -- unless the image is narrow
 * (a spectrogram, C_in = 1): then the layer runs in ROW-TOEPLITZ form, the GEMM's rows being (output x, output channel)
 * pairs and k running over whole zero-padded input rows, with no copy at all. A WAVEFRONT takes `batch` inputs through the
 * whole layer list by itself in its own slice of LDS (a layer's input images at one end of the activation region, its output
 * images at the other) -- no workgroup barrier in the loop (a wave's DS instructions are serviced in order); the first
 * version ran the workgroup in lockstep phases and spent a third of its time in barriers. The weight fragments stay in LDS
 * for the whole launch, shared by the waves, when they fit beside the activation slices (mode 2), else they stream from L2 (0).
 *
 * Two builds of this text. The GENERAL kernel (ed_net_mfma_kernel) reads all of the above from the plans at run time. A
 * graph's OWN kernel (ed_net_mfma_spec: -DEMM_SPEC, compiled at run time by edison_net_specialize with the plan in front of
 * the text as constants, net_spec.c) has the layer loop and the k-loops unrolled, every choice below made by the compiler,
 * fragments that several tile groups share held in registers: 2-3 x the general kernel, bit-identical.
 *
 * What the wave does NOT work out itself (the kernel is bound by vector-instruction issue, so every index calculation
 * counts): the planner ships one run record per layer (ed_mm_run_t, two scalar loads), the place of every input byte in
 * layer 0's layout, source / destination / valid bytes of every expansion record and the operand / output offsets of
 * every stored pixel; they are copied into LDS once per workgroup. Tiles: 32 x 32 x 32 with up to four MFMA chains at
 * once (groups of row tiles x column tiles x the positions of a fused pooling window, sharing A and B fragments), or 16 x 16 x 64
 * for layers with at most 16 columns per wave, where a 32-column tile would be mostly padding.
 */
/* -DEMM_JIT=1: this text is being compiled by hipRTC at run time (edison_net_specialize, edison_net_jit.hip) for one graph:
 * device code only, the headers come from the library's own copy of them, the kernel gets a C name */
/* ---- lab knobs: only a lab build (ED_LAB, tools/lab/mkvariant.py) may set them; the product build has none, and
 * tests/test_host_cpu.py checks the values below against what edison_amd/build.py compiles. (The run-time compiler of
 * edison_net_specialize defines EMM_JIT / EMM_SPEC / EMM_SPEC_HEADER -- modes, not knobs -- and never ED_LAB.) */
#if !defined(ED_LAB) && (defined(EMM_STAMP) || defined(EMM_SKIP) || defined(EMM_PB) || defined(EMM_PRIO) || defined(EMM_NO_HI) || defined(EMM_NO_OPAQUE))
#error "EMM_* lab knob defined without ED_LAB (tools/lab/mkvariant.py builds lab variants)"
#endif
#ifndef EMM_JIT
#define EMM_JIT 0
#endif
#if EMM_JIT
#include "edison_hip.h"
#else
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/edison_hip.h"
#endif
#include "edison_internal.h"
#if defined(ED_LAB) && !EMM_JIT
/* a lab build says so: the product library exports no ed_lab_build_* symbol (tests/test_host_cpu.py) */
extern "C" { extern const int ed_lab_build_cnn_net_mfma; const int ed_lab_build_cnn_net_mfma = 1; }
#endif

/* diagnostic build only (-DEMM_STAMP=1, tools/lab): workgroup-level cycle stamps per phase into a debug buffer */
#ifndef EMM_STAMP
#define EMM_STAMP 0
#endif
#if EMM_STAMP && !EMM_JIT
__device__ unsigned long long *g_emm_dbg = nullptr;
extern "C" void ed_set_net_debug_buffer(void *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_emm_dbg), &p, sizeof(p)); }
#define EMM_ST(i) { unsigned long long n_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_) :: "memory"); stamp_[i] += n_ - tl_; tl_ = n_; }
#else
#define EMM_ST(i)
#endif
/* timing-only ablations for A/B work (tools/lab; results are WRONG when non-zero): 1 no epilogue, 2 no k-loop (the
 * accumulators stay the seeds), 4 no expansion, 8 no input load, 16 no softmax / outputs, 32 zero seeds instead of the seed reads (32 x 32 tiles) */
#ifndef EMM_SKIP
#define EMM_SKIP 0
#endif
/* -DEMM_SPEC=1 -DEMM_SPEC_HEADER='"file"': a kernel for ONE graph -- the plans' scalars and layer records come from the
 * generated header (net_spec.c) as C++ constants, the layer loop is unrolled and the per-layer bookkeeping folds away */
/* lab: 1 = every layer requantises with a shift and two clamps per value (A/B of the high-byte path, ED_RUN_RS_HI) */
#ifndef EMM_NO_HI
#define EMM_NO_HI 0
#endif
#ifndef EMM_SPEC
#define EMM_SPEC 0
#endif
/* 1 = the lane number is left transparent, 0 = opaque per batch and per layer (emm_net_body): the general kernel needs the
 * opaque copies (+10.7 %: no register spills), a graph's own kernel is better off without them (+6.1 %: nothing spills there, and
 * the hoisted per-lane values are constants' worth of work saved per layer) */
#ifndef EMM_NO_OPAQUE
#define EMM_NO_OPAQUE EMM_SPEC
#endif
#if EMM_SPEC
#include EMM_SPEC_HEADER
#define EMM_PF(f) (EMM_SP_##f)
#define EMM_MF(f) (EMM_SM_##f)
#define EMM_RUN(li) (EMM_SR[li])
#define EMM_NETL(li) (EMM_SPL[li])
#define EMM_MML(li) (EMM_SML[li])
#define EMM_UNROLL_LAYERS _Pragma("unroll")
#define EMM_CONST constexpr
#else
#define EMM_PF(f) (P->f)
#define EMM_MF(f) (M->f)
#define EMM_RUN(li) (M->R[li])
#define EMM_NETL(li) (P->L[li])
#define EMM_MML(li) (M->L[li])
#define EMM_UNROLL_LAYERS
#define EMM_CONST const
#endif
#if EMM_SPEC
#define EMM_NL_EXPR EMM_SPEC_NL
#else
#define EMM_NL_EXPR (P->n_layers)
#endif
#define EMM_MAX_THREADS 768 /* 12 waves: 168 VGPRs each (four accumulator tiles + the pipeline's operands need ~150) */

template <int V> struct emm_int { static constexpr int value = V; }; /* a compile-time number as a generic lambda's argument */

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

/* Everything a wave touches in its layer loop lives in LDS, and the pointers say so: a generic pointer makes the
 * compiler emit flat_load / flat_store (64-bit addresses, both wait counters, a full drain before every MFMA). */
#define EMM_LDS __attribute__((address_space(3)))
typedef EMM_LDS int8_t lds8;
typedef EMM_LDS int lds32;
#define EMM_LD128(p) (*reinterpret_cast<const EMM_LDS v4i *>(p))
#define EMM_ST128(p, v) (*reinterpret_cast<EMM_LDS v4i *>(p) = (v))
#define EMM_LD32(p) (*reinterpret_cast<const EMM_LDS int *>(p))
#define EMM_ST32(p, v) (*reinterpret_cast<EMM_LDS uint32_t *>(p) = (v))

/* lo <= hi: v_min_i32 + v_max_i32 (the nested-ternary form compiles to a compare and a v_cndmask on top of the v_min) */
__device__ __forceinline__ int emm_med3(int v, int lo, int hi) { const int t = v > hi ? hi : v; return t < lo ? lo : t; }

/* Four accumulators -> four requantised bytes in a dword: sat8(v >> rs) with the ReLU folded in as the lower clamp, packed
 * with three v_perm_b32. (The v_cvt_pk_i16_i32 + v_sat_pk_u8_i16 sequence of cnn_mfma_kernels.hip measured 8.5 % SLOWER in
 * the graph's own kernel and 0.5 % slower here: with constant bounds the clamp is one v_med3_i32 and the asm statements of
 * the other form pin the schedule.) */
__device__ __forceinline__ uint32_t emm_pack4(int a0, int a1, int a2, int a3, int rs, int lo_clamp)
{
	const int v0 = emm_med3(a0 >> rs, lo_clamp, 127), v1 = emm_med3(a1 >> rs, lo_clamp, 127);
	const int v2 = emm_med3(a2 >> rs, lo_clamp, 127), v3 = emm_med3(a3 >> rs, lo_clamp, 127);
	const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)v1, (uint32_t)v0, 0x0c0c0400u), p23 = __builtin_amdgcn_perm((uint32_t)v3, (uint32_t)v2, 0x0c0c0400u);
	return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

/* The same four bytes where the planner allows it (ED_RUN_RS_HI; a0..a3 already shifted by rs - 8): sat8(v >> 8) is the high
 * byte of sat16(v) -- two v_cvt_pk_i16_i32, the lower clamp as two v_pk_max_i16 (lo2: 0 | 0 with a ReLU, -32768 | -32768
 * without), one v_perm_b32 that picks bytes 1 and 3 of both pairs: 5 instructions for four values instead of 15. */
__device__ __forceinline__ uint32_t emm_pack4_hi(int a0, int a1, int a2, int a3, uint32_t lo2)
{
	typedef short s2 __attribute__((ext_vector_type(2)));
	s2 p01 = __builtin_amdgcn_cvt_pk_i16(a0, a1), p23 = __builtin_amdgcn_cvt_pk_i16(a2, a3), lo;
	__builtin_memcpy(&lo, &lo2, 4);
	p01 = __builtin_elementwise_max(p01, lo); p23 = __builtin_elementwise_max(p23, lo);
	uint32_t u01, u23;
	__builtin_memcpy(&u01, &p01, 4); __builtin_memcpy(&u23, &p23, 4);
	return __builtin_amdgcn_perm(u23, u01, 0x07050301u);
}

/* i / d and i % d for 0 <= i < 2^21, 0 < d: one float multiply and a one-step correction instead of the ~20-instruction
 * integer division sequence (gfx950 has no integer divide); inv ~ 1 / d (v_rcp_f32, 1 ulp) is computed once per loop. Every
 * index here counts bytes or records of one wave's LDS slice, < 2^18. */
__device__ __forceinline__ void emm_divmod(int i, int d, float inv, int &q, int &r)
{
	q = (int)((float)i * inv);
	r = i - q * d;
	if (r < 0) { q--; r += d; }
	else if (r >= d) { q++; r -= d; }
}

/* wave-wide maximum / sum of one int per lane, the same (uniform) value in every lane: four DPP steps inside the rows of
 * 16, then the four row results through v_readlane */
#define EMM_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xf, 0xf, true)
__device__ __forceinline__ int emm_wave_max(int v)
{
	int t;
	t = EMM_DPP(v, 0xB1); v = t > v ? t : v;   /* quad_perm [1,0,3,2] */
	t = EMM_DPP(v, 0x4E); v = t > v ? t : v;   /* quad_perm [2,3,0,1] */
	t = EMM_DPP(v, 0x141); v = t > v ? t : v;  /* row_half_mirror      */
	t = EMM_DPP(v, 0x140); v = t > v ? t : v;  /* row_mirror           */
	const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
	const int m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
	return m01 > m23 ? m01 : m23;
}
/* the same inside every row of 16 lanes (four DPP steps, every lane of a row ends with its row's result): up to four images side by side */
__device__ __forceinline__ int emm_row_max(int v)
{
	int t;
	t = EMM_DPP(v, 0xB1); v = t > v ? t : v;
	t = EMM_DPP(v, 0x4E); v = t > v ? t : v;
	t = EMM_DPP(v, 0x141); v = t > v ? t : v;
	t = EMM_DPP(v, 0x140); v = t > v ? t : v;
	return v;
}
__device__ __forceinline__ int emm_row_add(int v)
{
	v += EMM_DPP(v, 0xB1); v += EMM_DPP(v, 0x4E); v += EMM_DPP(v, 0x141); v += EMM_DPP(v, 0x140);
	return v;
}
__device__ __forceinline__ int emm_wave_add(int v)
{
	v += EMM_DPP(v, 0xB1); v += EMM_DPP(v, 0x4E); v += EMM_DPP(v, 0x141); v += EMM_DPP(v, 0x140);
	return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

/* 16 bytes from byte offset soff of an LDS image, the bytes from `keep` on zeroed (keep >= 1) */
__device__ __forceinline__ v4i emm_gather16(const lds8 *a, int soff, int keep)
{
	const lds8 *s4 = a + (soff & ~3);
	const uint32_t sh = (uint32_t)(soff & 3) * 8;
	const uint32_t w0 = (uint32_t)EMM_LD32(s4), w1 = (uint32_t)EMM_LD32(s4 + 4), w2 = (uint32_t)EMM_LD32(s4 + 8), w3 = (uint32_t)EMM_LD32(s4 + 12), w4 = (uint32_t)EMM_LD32(s4 + 16);
	uint32_t d[4] = {__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh),
	                 __builtin_amdgcn_alignbit(w3, w2, sh), __builtin_amdgcn_alignbit(w4, w3, sh)};
#pragma unroll
	for (int t = 0; t < 4; t++)
	{
		const int kb = keep - 4 * t;
		d[t] = kb >= 4 ? d[t] : (kb <= 0 ? 0u : d[t] & (0xffffffffu >> (8 * (4 - kb))));
	}
	return (v4i){(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
}

/* elements 4 lane + 256 k .. +3 of one input image (in_n >= 4) into x[k], k < EMM_PRE, zero past the image. Branch-free:
 * a dword that would overrun the image's last byte (the batch may end there) is fetched from in_n - 4 instead and shifted
 * down; lanes past the image fetch that same dword and drop it. (With byte loads under divergent branches for the tail
 * the compiler put a full vmcnt(0) behind every one of them -- inside the code that was meant to PREFETCH.) */
#define EMM_PRE 2
/* images per wave the prefetch covers: 4 in a graph's own kernel (unused slots fold away), 2 here -- the general kernel sits at
 * its 168-register limit, two more images' registers go to scratch (+3 % at 2, measured on the planner's usual batch of 2) */
#ifndef EMM_PB
#define EMM_PB (EMM_SPEC ? 4 : 2)
#endif
/* A wave's priority rises with the layer it is in (the waves of a workgroup walk their own batches; see ED2_PRIO in
 * mfcc_kernels.hip): +9.5 % on the general kernel, +7.9 % on a graph's own (kws_conv, interleaved A/B). 0 = none; 3..6: lab */
#ifndef EMM_PRIO
#define EMM_PRIO (EMM_SPEC ? 1 : 2) /* own kernel: (4 li) / n (+7.9 %, min(li, 3): +6.9 %); general kernel: min(li, 3) (+9.5 %, (4 li) / n: +8.6 %) */
#endif
#if EMM_PRIO
#define EMM_PRIO_OF(li, n) (EMM_PRIO == 1 ? ((li) * 4) / (n) : EMM_PRIO == 2 ? (li) : EMM_PRIO == 3 ? (li) - ((n) - 4) : EMM_PRIO == 4 ? ((li) * 3) / (n) + 1 : ((li) * 8) / (n) - 2)
#define EMM_PR(li, n) if (EMM_PRIO != 6) { const int p_ = EMM_PRIO_OF(li, n); if (p_ <= 0) __builtin_amdgcn_s_setprio(0); else if (p_ == 1) __builtin_amdgcn_s_setprio(1); \
	else if (p_ == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
#else
#define EMM_PR(li, n)
#endif
__device__ __forceinline__ void emm_load_image(const int8_t *src, int in_n, int lane, uint32_t (&x)[EMM_PRE])
{
#pragma unroll
	for (int k = 0; k < EMM_PRE; k++)
	{
		const int e = 4 * lane + 256 * k;
		const int at = e + 4 <= in_n ? e : in_n - 4;
		x[k] = *reinterpret_cast<const uint32_t *>(src + at); /* raw: shifted where it is USED (emm_image_dword), an iteration later */
	}
}
__device__ __forceinline__ uint32_t emm_image_dword(uint32_t raw, int in_n, int lane, int k)
{
	const int e = 4 * lane + 256 * k;
	const int at = e + 4 <= in_n ? e : in_n - 4, sh = 8 * (e - at);
	return sh < 32 ? raw >> sh : 0u; /* e - at >= 4: nothing of this lane's four bytes lies inside the image */
}

struct emm_layout { int hp, wp, py, px, img; }; /* how an activation tensor lies in LDS: padded dims, origin, bytes per image */

__device__ __forceinline__ void emm_zero(lds8 *buf, int bytes, int lane)
{
	for (int i = lane * 16; i < bytes; i += 64 * 16) EMM_ST128(buf + i, ((v4i){0, 0, 0, 0}));
}

/* Order this wave's LDS writes before its following LDS reads: DS instructions of a wave are issued and serviced in
 * order; the (code-less) wave barrier keeps the compiler from moving memory operations across. */
__device__ __forceinline__ void emm_sync() { __builtin_amdgcn_wave_barrier(); }

template <bool FRAG_LDS>
__device__ __forceinline__ v4i emm_load_a(const lds8 *fl, const int8_t *fg, int s)
{
	if (FRAG_LDS) return EMM_LD128(fl + s * 1024);
	return *reinterpret_cast<const v4i *>(fg + (size_t)s * 1024);
}

/*
 * The k-loop of a GROUP of R row tiles x C column tiles with NW accumulator tiles each (the positions of a fused pooling
 * window; 1 when nothing is fused) -- R * C * NW <= 4 independent MFMA chains at once. A k-step fetches R A fragments (one
 * per row tile, shared by its C * NW chains) and C * NW B fragments (one per column tile and window, shared by the R row
 * tiles): the kernel moves 1 KB of LDS per fragment and the LDS pipe is two thirds busy, so a 2 x 2 group (4 KB for four
 * MFMAs) is worth twice a pair of independent tiles (4 KB for two). Software pipeline in program order, so that every wait
 * covers loads issued a whole step earlier (LDS returns in order):
 *   step s:  chunk offset of step s+2  |  A_u(s+1), B_c(s+1) at the offset read one step ago  |  MFMAs of step s
 * The accumulators start as the seeds (read straight into them). acc[u] returns the element-wise maximum over the
 * unit's windows (max before the one requantisation is exact: the requantisation is monotone).
 */
/* The operands carried into the next k-step pass through an empty asm: without it the optimiser notices that "load for
 * step s+1, use one iteration later" equals "load at the top of step s+1", rotates the loop back and the pipeline is gone
 * (every MFMA then waits for two dependent LDS round trips). */
template <int N>
__device__ __forceinline__ void emm_keep(v4i (&x)[N])
{
#pragma unroll
	for (int i = 0; i < N; i++) asm volatile("" : "+v"(x[i]));
}

/* MODE (a graph's own kernel only, where n_ks is a constant and the k-loop is unrolled): 1 = the B fragments of all k-steps are
 * already in registers (res[s * C * NW + c]: a layer with one column-tile group and several row-tile groups fetched them once
 * for all of those), 2 = the A fragments are (res[s * R + r]: one row-tile group, several column-tile groups). */
#define EMM_RES_MAX 8
/* AC: the caller's acc[] has AC tiles per row tile (a narrow chain for the last group of an odd tile count fills the first R x C
 * of the group's array) */
template <int NW, int R, int C, bool FRAG_LDS, int MODE, int AC>
__device__ __forceinline__ void emm_chain(const lds8 *const *fl, const int8_t *const *fg, const lds8 *kp /* &koff[h] */, const lds8 *const *bw,
                                          int n_ks, const lds8 *const *sp, v16i *acc, const v4i (&res)[EMM_RES_MAX])
{
	constexpr int NT = R * C * NW, NB = C * NW; /* accumulator tile (r, c, w) is aw[(r * C + c) * NW + w] */
	v16i aw[NT];
	const int last = n_ks - 1;
#if !EMM_SPEC
	/* The general kernel's n_ks is a run-time number. Its loop is two k-steps long and the operands PING-PONG between two
	 * register sets, so that "carry the operands into the next step" is no instruction at all (the one-step form below copied
	 * (R + C NW) fragments per step with v_mov: ~30 moves beside four MFMAs in this kernel). Step 0 stands in front of the
	 * loop: its MFMAs take a row tile's seeds as their C operand and write the tiles' own registers, so the C NW tiles of
	 * a row tile share ONE copy of the seeds (reading them "straight into the accumulators" cost 16 v_mov per further tile).
	 * Same program order as below: the chunk offset of step s+2, the operands of step s+1, the MFMAs of step s. */
	v16i sd[R];
#pragma unroll
	for (int r = 0; r < R; r++)
#pragma unroll
		for (int g = 0; g < 4; g++)
		{
			const v4i s4 = (EMM_SKIP & 32) ? (v4i){0, 0, 0, 0} : EMM_LD128(sp[r] + 32 * g);
			sd[r][4 * g] = s4.x; sd[r][4 * g + 1] = s4.y; sd[r][4 * g + 2] = s4.z; sd[r][4 * g + 3] = s4.w;
		}
	int ke = EMM_LD32(kp), ko = EMM_LD32(kp + 8 * (last < 1 ? last : 1)); /* offsets of the next even / odd step to fetch */
	v4i ae[R], be[NB], ao[R], bo[NB];                                       /* operands of an even / odd step              */
#pragma unroll
	for (int r = 0; r < R; r++) ae[r] = emm_load_a<FRAG_LDS>(fl[r], fg[r], 0);
#pragma unroll
	for (int c = 0; c < NB; c++) be[c] = EMM_LD128(bw[c] + ke);
	if (EMM_SKIP & 2)
	{
#pragma unroll
		for (int t = 0; t < NT; t++) aw[t] = sd[t / (C * NW)];
	}
	else
	{
		ke = EMM_LD32(kp + 8 * (2 < n_ks ? 2 : last));
		if (1 < n_ks) /* uniform */
		{
#pragma unroll
			for (int r = 0; r < R; r++) ao[r] = emm_load_a<FRAG_LDS>(fl[r], fg[r], 1);
#pragma unroll
			for (int c = 0; c < NB; c++) bo[c] = EMM_LD128(bw[c] + ko);
		}
#pragma unroll
		for (int t = 0; t < NT; t++) aw[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ae[t / (C * NW)], be[t % (C * NW)], sd[t / (C * NW)], 0, 0, 0);
		int s = 1;
		for (; s + 1 < n_ks; s += 2) /* steps s (odd set) and s + 1 (even set) */
		{
			ko = EMM_LD32(kp + 8 * (s + 2 < n_ks ? s + 2 : last));
#pragma unroll
			for (int r = 0; r < R; r++) ae[r] = emm_load_a<FRAG_LDS>(fl[r], fg[r], s + 1);
#pragma unroll
			for (int c = 0; c < NB; c++) be[c] = EMM_LD128(bw[c] + ke);
#pragma unroll
			for (int t = 0; t < NT; t++) aw[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ao[t / (C * NW)], bo[t % (C * NW)], aw[t], 0, 0, 0);
			ke = EMM_LD32(kp + 8 * (s + 3 < n_ks ? s + 3 : last));
			if (s + 2 < n_ks) /* uniform: the last step has nothing to fetch (a k-step's operands are R + C * NW KB of LDS traffic) */
			{
#pragma unroll
				for (int r = 0; r < R; r++) ao[r] = emm_load_a<FRAG_LDS>(fl[r], fg[r], s + 2);
#pragma unroll
				for (int c = 0; c < NB; c++) bo[c] = EMM_LD128(bw[c] + ko);
			}
#pragma unroll
			for (int t = 0; t < NT; t++) aw[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ae[t / (C * NW)], be[t % (C * NW)], aw[t], 0, 0, 0);
			/* the odd set and the even offset cross the back edge: see emm_keep */
			asm volatile("" : "+v"(ke));
			emm_keep(ao);
			emm_keep(bo);
		}
		if (s < n_ks) /* an even count: the last step's operands are in the odd set */
		{
#pragma unroll
			for (int t = 0; t < NT; t++) aw[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ao[t / (C * NW)], bo[t % (C * NW)], aw[t], 0, 0, 0);
		}
	}
#else
#pragma unroll
	for (int t = 0; t < NT; t++)
#pragma unroll
		for (int g = 0; g < 4; g++)
		{
			const v4i s4 = EMM_LD128(sp[t / (C * NW)] + 32 * g);
			aw[t][4 * g] = s4.x; aw[t][4 * g + 1] = s4.y; aw[t][4 * g + 2] = s4.z; aw[t][4 * g + 3] = s4.w;
		}
	int k_cur = EMM_LD32(kp), k_nxt = EMM_LD32(kp + 8 * (last < 1 ? last : 1));
	v4i a[R], b[NB];
#pragma unroll
	for (int r = 0; r < R; r++) a[r] = MODE == 2 ? res[r] : emm_load_a<FRAG_LDS>(fl[r], fg[r], 0);
#pragma unroll
	for (int c = 0; c < NB; c++) b[c] = MODE == 1 ? res[c] : EMM_LD128(bw[c] + k_cur);
	for (int s = 0; s < ((EMM_SKIP & 2) ? 0 : n_ks); s++)
	{
		const int s2 = s + 2 < n_ks ? s + 2 : last;
		const int k3 = MODE == 1 ? 0 : EMM_LD32(kp + 8 * s2);
		v4i an[R], bn[NB];
#pragma unroll
		for (int r = 0; r < R; r++) an[r] = a[r];
#pragma unroll
		for (int c = 0; c < NB; c++) bn[c] = b[c];
		if (s + 1 < n_ks) /* uniform: the last step has nothing to fetch (a k-step's operands are R + C * NW KB of LDS traffic) */
		{
#pragma unroll
			for (int r = 0; r < R; r++) an[r] = MODE == 2 ? res[(s + 1) * R + r] : emm_load_a<FRAG_LDS>(fl[r], fg[r], s + 1);
#pragma unroll
			for (int c = 0; c < NB; c++) bn[c] = MODE == 1 ? res[(s + 1) * NB + c] : EMM_LD128(bw[c] + k_nxt);
		}
#pragma unroll
		for (int t = 0; t < NT; t++) aw[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t / (C * NW)], b[t % (C * NW)], aw[t], 0, 0, 0);
		k_nxt = k3;
#pragma unroll
		for (int r = 0; r < R; r++) a[r] = an[r];
#pragma unroll
		for (int c = 0; c < NB; c++) b[c] = bn[c];
		asm volatile("" : "+v"(k_nxt));
		emm_keep(a);
		emm_keep(b);
	}
#endif
#pragma unroll
	for (int u = 0; u < R * C; u++)
	{
		v16i &m = acc[(u / C) * AC + u % C];
		m = aw[u * NW];
#pragma unroll
		for (int w = 1; w < NW; w++)
#pragma unroll
			for (int i = 0; i < 16; i++) m[i] = aw[u * NW + w][i] > m[i] ? aw[u * NW + w][i] : m[i];
	}
}

/* what the tile loop of a layer needs, gathered once per layer (all wave-uniform) */
struct emm_mm_args
{
	const lds8 *bsrc;       /* B source: the layer's input image(s) or their expanded copy               */
	lds8 *o;                /* where the epilogue stores                                                  */
	const lds8 *fragl; const int8_t *fragg; /* this layer's fragments (LDS resident / global)             */
	const lds8 *koff;       /* chunk offsets [2 s + h]                                                    */
	const lds8 *seeds;      /* 32 * n_rt accumulator seeds                                                */
	const lds8 *coltab;     /* (B offset, output offset) per stored pixel, or null                        */
	int img, o_img;         /* bytes per image of the B source / of the output layout                     */
	int n_ks, n_rt, n_cols, pix_per_img, col_w;
	int pitch_x, pitch_y, sh, ph, pw;
	int o_origin, o_row, oc_pitch, out_c, rs, lo_clamp;
	int hi, qsh;            /* ED_RUN_RS_HI: requantise through the high byte of sat16(v >> qsh), qsh = rs - 8 (< 0: a left shift) */
	int small;              /* 16 x 16 x 64 tiles (n_ks / n_rt count those) */
#if EMM_STAMP
	unsigned long long *st_, *tl_p;
#endif
};
#if EMM_STAMP
#define EMM_ST_T(i) { unsigned long long n_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_) :: "memory"); A.st_[i] += n_ - *A.tl_p; *A.tl_p = n_; }
#else
#define EMM_ST_T(i)
#endif

template <int NW, int R, int C, bool FRAG_LDS>
__device__ __forceinline__ void emm_layer_tiles(const emm_mm_args &A, int lane)
{
	const int col = lane & 31, h = lane >> 5;
	const int n_ct = (A.n_cols + 31) >> 5;
	const float inv_ppi = __builtin_amdgcn_rcpf((float)A.pix_per_img), inv_ow = __builtin_amdgcn_rcpf((float)A.col_w);
	/* the windows of a column start (wy * sh) input rows / wx output columns after its first one; the fused windows are
	 * 2x1, 1x2 or 2x2 */
	int wdelta[NW];
#pragma unroll
	for (int w = 0; w < NW; w++)
	{
		const int wy = A.pw == 2 ? w >> 1 : w, wx = A.pw == 2 ? w & 1 : 0;
		wdelta[w] = (wy * A.sh) * A.pitch_y + wx * A.pitch_x;
	}
	/* A graph's own kernel keeps what every group of a layer would fetch again in registers (all uniform, constants there):
	 * the A fragments when the layer is ONE group of row tiles walked over several column-tile groups, the B fragments when it
	 * is one column-tile group under several row-tile groups (a Toeplitz first layer: 3 of its 4.5 KB of operands per group). */
	const bool a_res = EMM_SPEC && !(EMM_SKIP & 2) && A.n_rt <= R && n_ct > C && A.n_ks * R <= EMM_RES_MAX;
	const bool b_res = EMM_SPEC && !(EMM_SKIP & 2) && !a_res && A.n_rt > R && n_ct <= C && A.n_ks * C * NW <= EMM_RES_MAX;
	v4i res[EMM_RES_MAX];
	if (a_res)
	{
#pragma unroll
		for (int r = 0; r < R; r++)
		{
			const int rt = r < A.n_rt ? r : A.n_rt - 1;
			for (int s_ = 0; s_ < A.n_ks; s_++)
				res[s_ * R + r] = emm_load_a<FRAG_LDS>(A.fragl + rt * A.n_ks * 1024 + lane * 16, A.fragg + (size_t)rt * A.n_ks * 1024 + lane * 16, s_);
		}
	}
	/* groups of R row tiles x C column tiles; a group's spare slots (past the last row / column tile) repeat the last tile
	 * and store nothing */
	for (int ct0 = 0; ct0 < n_ct; ct0 += C)
	{
		/* this lane's column in each of the C column tiles: its B source and where its outputs go */
		const lds8 *bw[NW * C];
		lds8 *op[C];
		bool live[C];
#pragma unroll
		for (int c = 0; c < C; c++)
		{
			const int ct = ct0 + c < n_ct ? ct0 + c : n_ct - 1;
			const int q = ct * 32 + col;
			live[c] = ct0 + c < n_ct && q < A.n_cols;
			const int qq = q < A.n_cols ? q : A.n_cols - 1;
			int b = 0, pp = qq;
			if (A.n_cols > A.pix_per_img) emm_divmod(qq, A.pix_per_img, inv_ppi, b, pp); /* uniform: more than one image per wave */
			int boff, ooff;
			if (A.coltab)
			{
				boff = EMM_LD32(A.coltab + 8 * pp); ooff = EMM_LD32(A.coltab + 8 * pp + 4);
			}
			else
			{
				int y, x;
				emm_divmod(pp, A.col_w, inv_ow, y, x);
				boff = (y * A.ph * A.sh) * A.pitch_y + (x * A.pw) * A.pitch_x;
				ooff = A.o_origin + y * A.o_row + x * A.oc_pitch;
			}
#pragma unroll
			for (int w = 0; w < NW; w++) bw[c * NW + w] = A.bsrc + b * A.img + boff + wdelta[w];
			op[c] = A.o + b * A.o_img + ooff;
		}
		if (b_res)
		{
			for (int s_ = 0; s_ < A.n_ks; s_++)
			{
				const int k_ = EMM_LD32(A.koff + 4 * h + 8 * s_);
#pragma unroll
				for (int c = 0; c < NW * C; c++) res[s_ * NW * C + c] = EMM_LD128(bw[c] + k_);
			}
		}
		for (int rt0 = 0; rt0 < A.n_rt; rt0 += R)
		{
			EMM_ST_T(43)
			const lds8 *fl[R], *sp[R];
			const int8_t *fg[R];
			int rts[R];
#pragma unroll
			for (int r = 0; r < R; r++)
			{
				rts[r] = rt0 + r < A.n_rt ? rt0 + r : A.n_rt - 1;
				fl[r] = A.fragl + rts[r] * A.n_ks * 1024 + lane * 16;
				fg[r] = A.fragg + (size_t)rts[r] * A.n_ks * 1024 + lane * 16;
				sp[r] = A.seeds + 4 * (32 * rts[r] + 4 * h);
			}
			v16i acc[R * C];
			EMM_ST_T(40)
			if (a_res) emm_chain<NW, R, C, FRAG_LDS, EMM_SPEC ? 2 : 0, C>(fl, fg, A.koff + 4 * h, bw, A.n_ks, sp, acc, res);
			else if (b_res) emm_chain<NW, R, C, FRAG_LDS, EMM_SPEC ? 1 : 0, C>(fl, fg, A.koff + 4 * h, bw, A.n_ks, sp, acc, res);
			else emm_chain<NW, R, C, FRAG_LDS, 0, C>(fl, fg, A.koff + 4 * h, bw, A.n_ks, sp, acc, res);
			EMM_ST_T(41)
			/* lane (column, h) holds rows 32 rt + 8 g + 4 h .. +3 in registers 4g..4g+3. Requantise, clamp, pack four rows
			 * into a dword with three v_perm_b32; whole groups of 8 rows past C_out are skipped under a uniform branch, the
			 * store alone is predicated. */
			const uint32_t lo2 = A.lo_clamp == 0 ? 0u : 0x80008000u;
			const bool full8 = (A.out_c & 7) == 0; /* uniform: a live group of 8 rows is then wholly below C_out */
			/* QM: 0 (and 4, with byte stores) shift + clamp per value (emm_pack4), 1 / 2 / 3 through the high byte of sat16 (emm_pack4_hi) with no / a right / a
			 * left shift in front of it -- five copies of the code under one uniform branch per group, so that none of them
			 * carries the others' selects and register copies */
			auto epilogue = [&](auto qm_) __attribute__((always_inline))
			{
				constexpr int QM = decltype(qm_)::value;
#pragma unroll
				for (int r = 0; r < R; r++)
#pragma unroll
					for (int c = 0; c < C; c++)
					{
						if ((EMM_SKIP & 1) || rt0 + r >= A.n_rt) continue; /* uniform */
						const v16i &t = acc[r * C + c];
						uint32_t d[4];
						/* the arithmetic of the whole tile under uniform branches only, then ONE predicated region for its stores (a
						 * predicate per store cost five scalar instructions each: exec saved, masked, branched over, restored) */
#pragma unroll
						for (int g = 0; g < 4; g++)
						{
							int a0 = t[4 * g], a1 = t[4 * g + 1], a2 = t[4 * g + 2], a3 = t[4 * g + 3];
							if (QM == 2) { a0 >>= A.qsh; a1 >>= A.qsh; a2 >>= A.qsh; a3 >>= A.qsh; }
							if (QM == 3) { a0 = (int)((uint32_t)a0 << -A.qsh); a1 = (int)((uint32_t)a1 << -A.qsh); a2 = (int)((uint32_t)a2 << -A.qsh); a3 = (int)((uint32_t)a3 << -A.qsh); }
							d[g] = (QM == 0 || QM == 4) ? emm_pack4(a0, a1, a2, a3, A.rs, A.lo_clamp) : emm_pack4_hi(a0, a1, a2, a3, lo2);
						}
						if (!live[c]) continue;
#pragma unroll
						for (int g = 0; g < 4; g++)
						{
							if (32 * rts[r] + 8 * g >= A.out_c) continue; /* uniform */
							const int r0 = 32 * rts[r] + 8 * g + 4 * h;
							if (QM != 4)
							{
								if (full8 || r0 < A.out_c) EMM_ST32(op[c] + r0, d[g]);
							}
							else /* a C_out that is no multiple of 4: byte stores, in a copy of their own (they are the larger half of it) */
							{
								if (r0 < A.out_c) op[c][r0] = (int8_t)d[g];
								if (r0 + 1 < A.out_c) op[c][r0 + 1] = (int8_t)(d[g] >> 8);
								if (r0 + 2 < A.out_c) op[c][r0 + 2] = (int8_t)(d[g] >> 16);
								if (r0 + 3 < A.out_c) op[c][r0 + 3] = (int8_t)(d[g] >> 24);
							}
						}
					}
			};
			if ((A.out_c & 3) != 0) epilogue(emm_int<4>());
			else if (!A.hi) epilogue(emm_int<0>());
			else if (A.qsh == 0) epilogue(emm_int<1>());
			else if (A.qsh > 0) epilogue(emm_int<2>());
			else epilogue(emm_int<3>());
			EMM_ST_T(42)
		}
	}
}

/*
 * Layers with at most 16 columns per wave and no fused pooling (model_net_mm.c: `small`): v_mfma_i32_16x16x64_i8 tiles --
 * 16 output channels x 16 columns, four 16-byte chunks per k-step (lane l: row / column l & 15, chunk l >> 4). A 32 x 32
 * tile there would be mostly padding that the epilogue requantises all the same (conv4 of kws_conv: 3 live columns, the
 * dense layer: 1); here a tile is 4 accumulator registers instead of 16 and the k-loop half as long. Up to four row
 * tiles run at once and share the B fragment of a k-step; same one-deep pipeline as emm_chain.
 */
template <int U, bool FRAG_LDS>
__device__ __forceinline__ void emm_layer_small(const emm_mm_args &A, int lane)
{
	const int col = lane & 15, kq = lane >> 4;
	const bool live = col < A.n_cols;
	const int qq = live ? col : A.n_cols - 1;
	int b = 0, pp = qq;
	if (A.n_cols > A.pix_per_img) emm_divmod(qq, A.pix_per_img, __builtin_amdgcn_rcpf((float)A.pix_per_img), b, pp);
	int boff, ooff;
	if (A.coltab)
	{
		boff = EMM_LD32(A.coltab + 8 * pp); ooff = EMM_LD32(A.coltab + 8 * pp + 4);
	}
	else
	{
		int y, x;
		emm_divmod(pp, A.col_w, __builtin_amdgcn_rcpf((float)A.col_w), y, x);
		boff = (y * A.sh) * A.pitch_y + x * A.pitch_x;
		ooff = A.o_origin + y * A.o_row + x * A.oc_pitch;
	}
	const lds8 *bp = A.bsrc + b * A.img + boff;
	lds8 *op = A.o + b * A.o_img + ooff;
	const lds8 *kp = A.koff + 4 * kq;
	const int n_ks = A.n_ks, last = n_ks - 1;
	for (int rt0 = 0; rt0 < A.n_rt; rt0 += U)
	{
		v4i aw[U], a[U];
		int rts[U];
		const lds8 *fl[U];
		const int8_t *fg[U];
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			rts[u] = rt0 + u < A.n_rt ? rt0 + u : A.n_rt - 1; /* spare slots repeat the last tile and store nothing */
			aw[u] = EMM_LD128(A.seeds + 4 * (16 * rts[u] + 4 * kq));
			fl[u] = A.fragl + rts[u] * n_ks * 1024 + lane * 16;
			fg[u] = A.fragg + (size_t)rts[u] * n_ks * 1024 + lane * 16;
			a[u] = emm_load_a<FRAG_LDS>(fl[u], fg[u], 0);
		}
#if !EMM_SPEC
		/* two k-steps per iteration, the operands ping-pong between two register sets (see emm_chain) */
		int k0 = EMM_LD32(kp), k1 = EMM_LD32(kp + 16 * (last < 1 ? last : 1));
		v4i bq = EMM_LD128(bp + k0), a1[U], b1;
		int s = 0;
		for (; s + 1 < ((EMM_SKIP & 2) ? 0 : n_ks); s += 2)
		{
			k0 = EMM_LD32(kp + 16 * (s + 2 < n_ks ? s + 2 : last));
#pragma unroll
			for (int u = 0; u < U; u++) a1[u] = emm_load_a<FRAG_LDS>(fl[u], fg[u], s + 1);
			b1 = EMM_LD128(bp + k1);
#pragma unroll
			for (int u = 0; u < U; u++) aw[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u], bq, aw[u], 0, 0, 0);
			k1 = EMM_LD32(kp + 16 * (s + 3 < n_ks ? s + 3 : last));
			if (s + 2 < n_ks) /* uniform: the last step has nothing to fetch */
			{
#pragma unroll
				for (int u = 0; u < U; u++) a[u] = emm_load_a<FRAG_LDS>(fl[u], fg[u], s + 2);
				bq = EMM_LD128(bp + k0);
			}
#pragma unroll
			for (int u = 0; u < U; u++) aw[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[u], b1, aw[u], 0, 0, 0);
			asm volatile("" : "+v"(k1), "+v"(bq));
			emm_keep(a);
		}
		if (s < ((EMM_SKIP & 2) ? 0 : n_ks))
		{
#pragma unroll
			for (int u = 0; u < U; u++) aw[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u], bq, aw[u], 0, 0, 0);
		}
#else
		int k_cur = EMM_LD32(kp), k_nxt = EMM_LD32(kp + 16 * (last < 1 ? last : 1));
		v4i bq = EMM_LD128(bp + k_cur);
		for (int s = 0; s < ((EMM_SKIP & 2) ? 0 : n_ks); s++)
		{
			const int s2 = s + 2 < n_ks ? s + 2 : last;
			const int k3 = EMM_LD32(kp + 16 * s2);
			v4i an[U], bn = bq;
#pragma unroll
			for (int u = 0; u < U; u++) an[u] = a[u];
			if (s + 1 < n_ks) /* uniform: the last step has nothing to fetch */
			{
#pragma unroll
				for (int u = 0; u < U; u++) an[u] = emm_load_a<FRAG_LDS>(fl[u], fg[u], s + 1);
				bn = EMM_LD128(bp + k_nxt);
			}
#pragma unroll
			for (int u = 0; u < U; u++) aw[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u], bq, aw[u], 0, 0, 0);
			k_nxt = k3; bq = bn;
#pragma unroll
			for (int u = 0; u < U; u++) a[u] = an[u];
			asm volatile("" : "+v"(k_nxt), "+v"(bq));
			emm_keep(a);
		}
#endif
		/* lane (column, kq) holds rows 16 rt + 4 kq .. +3 */
#pragma unroll
		for (int u = 0; u < U; u++)
		{
			const int r0 = 16 * rts[u] + 4 * kq;
			if ((EMM_SKIP & 1) || rt0 + u >= A.n_rt) continue; /* uniform */
			uint32_t d;
			if (A.hi) /* uniform */
			{
				v4i t = aw[u];
				if (A.qsh > 0) { t.x >>= A.qsh; t.y >>= A.qsh; t.z >>= A.qsh; t.w >>= A.qsh; }
				else if (A.qsh < 0) { t.x = (int)((uint32_t)t.x << -A.qsh); t.y = (int)((uint32_t)t.y << -A.qsh); t.z = (int)((uint32_t)t.z << -A.qsh); t.w = (int)((uint32_t)t.w << -A.qsh); }
				d = emm_pack4_hi(t.x, t.y, t.z, t.w, A.lo_clamp == 0 ? 0u : 0x80008000u);
			}
			else d = emm_pack4(aw[u].x, aw[u].y, aw[u].z, aw[u].w, A.rs, A.lo_clamp);
			if ((A.out_c & 3) == 0)
			{
				if (live && r0 < A.out_c) EMM_ST32(op + r0, d);
			}
			else if (live)
			{
				if (r0 < A.out_c) op[r0] = (int8_t)d;
				if (r0 + 1 < A.out_c) op[r0 + 1] = (int8_t)(d >> 8);
				if (r0 + 2 < A.out_c) op[r0 + 2] = (int8_t)(d >> 16);
				if (r0 + 3 < A.out_c) op[r0 + 3] = (int8_t)(d >> 24);
			}
		}
	}
}

/* NW windows per tile (1, 2 or 4) and the group shape: R row tiles x C column tiles, R * C * NW <= 4 accumulator tiles, rows
 * first (a row tile more costs one A fragment per k-step, a column tile more costs NW B fragments) */
template <bool FRAG_LDS>
__device__ __forceinline__ void emm_layer_dispatch(const emm_mm_args &A, int lane)
{
	if (A.small)
	{
		/* row tiles at a time: as many as the layer has, up to four (a spare slot would fetch its A fragments all the same) */
		if (A.n_rt >= 3) emm_layer_small<4, FRAG_LDS>(A, lane);
		else if (A.n_rt == 2) emm_layer_small<2, FRAG_LDS>(A, lane);
		else emm_layer_small<1, FRAG_LDS>(A, lane);
		return;
	}
	const int nwin = A.ph * A.pw, n_ct = (A.n_cols + 31) >> 5;
	if (nwin == 1)
	{
		if (A.n_rt >= 2 && n_ct >= 2) emm_layer_tiles<1, 2, 2, FRAG_LDS>(A, lane);
		else if (A.n_rt >= 2) emm_layer_tiles<1, 2, 1, FRAG_LDS>(A, lane);
		else if (n_ct >= 2) emm_layer_tiles<1, 1, 2, FRAG_LDS>(A, lane);
		else emm_layer_tiles<1, 1, 1, FRAG_LDS>(A, lane);
	}
	else if (nwin == 2)
	{
		if (A.n_rt >= 2) emm_layer_tiles<2, 2, 1, FRAG_LDS>(A, lane);
		else if (n_ct >= 2) emm_layer_tiles<2, 1, 2, FRAG_LDS>(A, lane);
		else emm_layer_tiles<2, 1, 1, FRAG_LDS>(A, lane);
	}
	else emm_layer_tiles<4, 1, 1, FRAG_LDS>(A, lane);
}

template <bool FRAG_LDS>
__device__ __forceinline__ void emm_net_body(const ed_net_plan_t *__restrict__ P, const ed_mm_plan_t *__restrict__ M,
                                             const int8_t *__restrict__ frag, const int32_t *__restrict__ seeds,
                                             const int8_t *__restrict__ in, int64_t n, int64_t in_stride,
                                             int8_t *__restrict__ logits, int8_t *__restrict__ softmax,
                                             int32_t *__restrict__ argmax, unsigned *done_flag, unsigned done_seq)
{
	extern __shared__ __attribute__((aligned(16))) int8_t emm_lds_generic[];
	lds8 *emm_lds = (lds8 *)emm_lds_generic;
	EMM_CONST int n_layers = EMM_NL_EXPR, batch = EMM_MF(batch), buf_bytes = EMM_MF(buf_bytes);
	const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if EMM_PRIO == 6 /* lab: a fixed priority per wave of a SIMD */
	if ((wave >> 2) == 1) __builtin_amdgcn_s_setprio(1); else if ((wave >> 2) >= 2) __builtin_amdgcn_s_setprio(2);
#endif
	const int n_threads = blockDim.x, n_waves = n_threads >> 6;
	/* small tables, copied once per workgroup: chunk offsets of every layer | seeds | column, expansion, input tables; then
	 * the weight fragments (when resident); then one slice per wave: the activation region and the expansion
	 * buffer */
	lds8 *tbl = emm_lds;
	EMM_CONST int n_koff = EMM_MF(n_koff), n_seeds = EMM_MF(n_seeds), n_coltab = EMM_MF(n_cols);
	lds32 *koff_all = reinterpret_cast<lds32 *>(tbl);
	lds32 *seeds_l = reinterpret_cast<lds32 *>(tbl + ((4 * n_koff + 15) & ~15));
	lds32 *coltab_l = reinterpret_cast<lds32 *>(reinterpret_cast<lds8 *>(seeds_l) + ((4 * n_seeds + 15) & ~15));
	EMM_CONST int n_xtab = EMM_MF(n_xtab), n_intab = EMM_MF(n_intab);
	lds32 *xtab_l = reinterpret_cast<lds32 *>(reinterpret_cast<lds8 *>(coltab_l) + ((8 * n_coltab + 15) & ~15));
	EMM_LDS uint16_t *intab_l = reinterpret_cast<EMM_LDS uint16_t *>(reinterpret_cast<lds8 *>(xtab_l) + ((8 * n_xtab + 15) & ~15));
	lds8 *fragl = tbl + EMM_MF(tbl_bytes);
	lds8 *slice = fragl + EMM_MF(frag_lds) + wave * (2 * buf_bytes + EMM_MF(x_bytes));
	lds8 *xbuf = slice + 2 * buf_bytes; /* the activation region in front of it: a layer's input at one end, its output at the other */
	lds8 *in0 = slice + EMM_RUN(0).in_off; /* (layer 0 always runs: a MaxPool is only skipped behind a convolution) */
	{
		for (int i = threadIdx.x; i < n_koff; i += n_threads) koff_all[i] = M->koff[i];
		for (int i = threadIdx.x; i < n_seeds; i += n_threads) seeds_l[i] = seeds[i];
		for (int i = threadIdx.x; i < 2 * n_coltab; i += n_threads) coltab_l[i] = M->coltab[i];
		for (int i = threadIdx.x; i < 2 * n_xtab; i += n_threads) xtab_l[i] = M->xtab[i];
		for (int i = threadIdx.x; i < n_intab; i += n_threads) intab_l[i] = M->intab[i];
		if (FRAG_LDS)
		{
			const v4i *src = reinterpret_cast<const v4i *>(frag);
			for (int i = threadIdx.x; i < EMM_MF(frag_bytes) / 16; i += n_threads) EMM_ST128(fragl + 16 * i, src[i]);
		}
	}
	__syncthreads(); /* the only workgroup barrier: from here on every wave is on its own */
	EMM_CONST int out_n = EMM_PF(out_n), logits_layer = EMM_PF(logits_layer), has_softmax = EMM_PF(has_softmax);

#if EMM_STAMP
	unsigned long long stamp_[48], tl_;
	for (int i = 0; i < 48; i++) stamp_[i] = 0;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_) :: "memory");
#endif
	/* Images of at most 512 bytes, up to EMM_PB per wave: the next batch's dwords are requested from HBM before this one's
	 * layers run (two registers per image), so that the wave never waits a memory latency per batch. */
	const int64_t u_first = ((int64_t)blockIdx.x * n_waves + wave) * batch, u_step = (int64_t)gridDim.x * n_waves * batch;
	static_assert(EMM_PRE * 256 == ED_MM_INTAB_PAD, "the planner pads the input table to what the prefetch covers");
	EMM_CONST bool prefetch = n_intab >= ED_MM_INTAB_PAD && batch <= EMM_PB && EMM_PF(in_n) <= EMM_PRE * 256 && EMM_PF(in_n) >= 4;
	uint32_t pre[EMM_PB][EMM_PRE];
	if (prefetch && u_first < n)
	{
#pragma unroll
		for (int b = 0; b < EMM_PB; b++) /* a slot past the end of the input fetches the last image again: it is never written out */
			if (b < batch) emm_load_image(in + (u_first + b < n ? u_first + b : n - 1) * in_stride, EMM_PF(in_n), lane0, pre[b]);
	}
	for (int64_t u0 = u_first; u0 < n; u0 += u_step)
	{
		/* The lane number is made opaque once per batch and once per layer: everything a stage derives from it (addresses into
		 * its tables and images, column numbers, predicates) is then worked out where it is used. Left to itself the optimiser
		 * hoists those loop-invariant per-lane values of EVERY stage and tile shape in front of the batch loop, keeps them
		 * alive across it, runs out of registers and spills -- and a scratch reload waits on vmcnt, i.e. for the NEXT batch's
		 * prefetch that was put in flight just before it: the wave then sits out a memory latency per batch after all. */
		int lane = lane0;
		if (!EMM_NO_OPAQUE) asm volatile("" : "+v"(lane));
		const int nb = batch == 1 ? 1 : (int)((n - u0) < batch ? (n - u0) : batch);
		EMM_ST(47)
		/* ---- the inputs into layer 0's layout */
		{
			const ed_mm_layer_t ml0 = EMM_MML(0);
			const emm_layout l0 = {ml0.in_hp, ml0.in_wp, ml0.in_py, ml0.in_px, ml0.in_img};
			EMM_CONST int in_h = EMM_PF(in_h), in_w = EMM_PF(in_w), in_c = EMM_PF(in_c), in_n = EMM_PF(in_n);
			if (l0.hp != in_h || l0.wp != in_w) /* uniform: a zero border to keep */
			{
				emm_zero(in0, batch * l0.img, lane);
				emm_sync();
			}
			if (EMM_SKIP & 8) {}
			else if (prefetch)
			{
				uint32_t v[EMM_PB][EMM_PRE];
#pragma unroll
				for (int b = 0; b < EMM_PB; b++)
#pragma unroll
					for (int k = 0; k < EMM_PRE; k++) v[b][k] = b < batch ? emm_image_dword(pre[b][k], in_n, lane, k) : 0u;
				/* unconditional, like the MFCC kernels' prefetch: a wave's last pass re-reads its own images (an L2 hit) */
				const int64_t u_nxt = u0 + u_step < n ? u0 + u_step : u0;
#pragma unroll
				for (int b = 0; b < EMM_PB; b++)
					if (b < batch) emm_load_image(in + (u_nxt + b < n ? u_nxt + b : n - 1) * in_stride, in_n, lane, pre[b]);
#pragma unroll
				for (int k = 0; k < EMM_PRE; k++)
				{
					/* the four places of this lane's dword k in one 8-byte read; no test: the table is padded (ED_MM_INTAB_PAD), a byte
					 * past the image lands in the image's slack */
					const int e = 4 * lane + 256 * k;
					const uint64_t at4 = *reinterpret_cast<const EMM_LDS uint64_t *>(intab_l + e);
#pragma unroll
					for (int t = 0; t < 4; t++)
					{
						const int at = (int)((at4 >> (16 * t)) & 0xffffu); /* the same place in every image of the batch */
#pragma unroll
						for (int b = 0; b < EMM_PB; b++)
							if (b < batch) in0[b * l0.img + at] = (int8_t)(v[b][k] >> (8 * t));
					}
				}
			}
			else if (n_intab)
			{
				/* four elements per lane and step: one (unaligned) dword from HBM, their four places from the table */
				for (int b = 0; b < nb; b++)
				{
					const int8_t *src = in + (u0 + b) * in_stride;
					lds8 *dst = in0 + b * l0.img;
					for (int e = 4 * lane; e < in_n; e += 256)
					{
						uint32_t v;
						if (e + 4 <= in_n) v = *reinterpret_cast<const uint32_t *>(src + e); /* the batch's last bytes are not overrun */
						else
						{
							v = (uint32_t)(uint8_t)src[e];
							if (e + 1 < in_n) v |= (uint32_t)(uint8_t)src[e + 1] << 8;
							if (e + 2 < in_n) v |= (uint32_t)(uint8_t)src[e + 2] << 16;
						}
#pragma unroll
						for (int t = 0; t < 4; t++)
							if (e + t < in_n) dst[intab_l[e + t]] = (int8_t)(v >> (8 * t));
					}
				}
			}
			else
			{
				const float inv_n = __builtin_amdgcn_rcpf((float)in_n), inv_c = __builtin_amdgcn_rcpf((float)in_c), inv_w = __builtin_amdgcn_rcpf((float)in_w);
				for (int i0 = 0; i0 < nb * in_n; i0 += 8 * 64)
				{
					int8_t v[8];
#pragma unroll
					for (int k = 0; k < 8; k++)
					{
						const int i = i0 + k * 64 + lane;
						int b, e;
						emm_divmod(i, in_n, inv_n, b, e);
						v[k] = i < nb * in_n ? in[(u0 + b) * in_stride + e] : 0;
					}
#pragma unroll
					for (int k = 0; k < 8; k++)
					{
						const int i = i0 + k * 64 + lane;
						if (i >= nb * in_n) continue;
						int b, e, pix, c, y, x;
						emm_divmod(i, in_n, inv_n, b, e); emm_divmod(e, in_c, inv_c, pix, c); emm_divmod(pix, in_w, inv_w, y, x);
						in0[b * l0.img + ((y + l0.py) * l0.wp + x + l0.px) * in_c + c] = v[k];
					}
				}
			}
			emm_sync();
		}
		EMM_ST(0)
		EMM_UNROLL_LAYERS
		for (int li = 0; li < n_layers; li++)
		{
			/* one wave-uniform run record per layer, worked out by the planner: two scalar loads (copies in LDS cost a
			 * ds_read + v_readfirstlane per field; deriving it here from the layer records took a chain of dependent
			 * scalar loads and ~100 scalar instructions per layer and input) */
			EMM_PR(li, n_layers)
			EMM_ST(4 + 5 * li)
			const ed_mm_run_t R = EMM_RUN(li);
			int lane_l = lane0; /* opaque per layer: see the top of the batch loop */
			if (!EMM_NO_OPAQUE) asm volatile("" : "+v"(lane_l));
			if (R.kind == ED_RUN_SKIP) continue; /* a MaxPool taken in the epilogue of the layer in front of it */
			EMM_ST(3 + 5 * li)
			const lds8 *a = slice + R.in_off;
			lds8 *o = slice + R.o_off;
			if (R.zero_border) /* uniform: the consumer wants a zero border */
			{
				emm_zero(o, batch * R.o_img, lane_l);
				emm_sync();
			}
			if (R.kind == ED_RUN_MM)
			{
				emm_mm_args A;
				A.bsrc = a; A.img = R.in_img;
				if (R.expand && !(EMM_SKIP & 4))
				{
					/* one aligned record of 16 * cpr bytes per (input row, output x): the kw * C_in bytes under a kernel row.
					 * 16 bytes from an arbitrary byte offset: five aligned dwords around them, funnel-shifted (v_alignbit), the
					 * bytes past the end of the kernel-row segment zeroed (the image buffers carry 16 bytes of slack) */
					if (R.xtab_off >= 0)
					{
						const lds8 *xt = reinterpret_cast<const lds8 *>(xtab_l + 2 * R.xtab_off);
						for (int i = lane_l; i < R.rec_per_img; i += 64)
						{
							const int w0_ = EMM_LD32(xt + 8 * i), doff = EMM_LD32(xt + 8 * i + 4);
							const int soff0 = w0_ & 0xffffff, keep = w0_ >> 24;
							for (int b = 0; b < nb; b++)
								EMM_ST128(xbuf + b * R.x_img + doff, emm_gather16(a, b * R.in_img + soff0, keep));
						}
					}
					else
					{
						/* no table (ED_MM_MAX_XTAB): the rare path reads the layer records and divides */
						const ed_net_layer_t L = EMM_NETL(li);
						const ed_mm_layer_t ML = EMM_MML(li);
						const int dense = L.type == ED_NET_DENSE, out_w = dense ? 1 : L.out_w;
						const int in_c = dense ? L.in_n : L.in_c, seg = (dense ? 1 : L.kw) * in_c, sw = dense ? 1 : L.sw;
						const float inv_rec = __builtin_amdgcn_rcpf((float)R.rec_per_img), inv_row = __builtin_amdgcn_rcpf((float)(out_w * ML.cpr)), inv_cpr = __builtin_amdgcn_rcpf((float)ML.cpr);
						for (int i = lane_l; i < nb * R.rec_per_img; i += 64)
						{
							int b, e, r, e2, xo, j;
							emm_divmod(i, R.rec_per_img, inv_rec, b, e); emm_divmod(e, out_w * ML.cpr, inv_row, r, e2); emm_divmod(e2, ML.cpr, inv_cpr, xo, j);
							const int soff = b * R.in_img + (r * ML.in_wp + xo * sw) * in_c + 16 * j;
							EMM_ST128(xbuf + b * R.x_img + r * R.pitch_y + xo * R.pitch_x + 16 * j, emm_gather16(a, soff, seg - 16 * j));
						}
					}
					A.bsrc = xbuf;
					A.img = R.x_img;
				}
				EMM_ST(1 + 5 * li)
				emm_sync();
				A.o = o; A.o_img = R.o_img;
				A.fragl = fragl + R.frag_off; A.fragg = frag + R.frag_off;
				A.koff = reinterpret_cast<const lds8 *>(koff_all + R.koff_off);
				A.seeds = reinterpret_cast<const lds8 *>(seeds_l + R.seed_off);
				A.coltab = (R.col_off >= 0 && !(EMM_SKIP & 64)) ? reinterpret_cast<const lds8 *>(coltab_l + 2 * R.col_off) : nullptr; /* lab, 64: no column tables (results stay right) */
				A.n_ks = R.n_ks; A.n_rt = R.n_rt;
				A.col_w = R.col_w;
				A.pix_per_img = R.pix_per_img;
				A.n_cols = (EMM_SPEC ? batch : nb) * R.pix_per_img; /* the graph's own kernel keeps its tile counts constant: the slots past
				                                                      * a ragged last batch compute on stale LDS and are never written out */
				A.pitch_x = R.pitch_x; A.pitch_y = R.pitch_y; A.sh = R.sh;
				A.ph = R.ph; A.pw = R.pw;
				A.small = R.small;
				A.o_origin = R.o_origin; A.o_row = R.o_row; A.oc_pitch = R.oc_pitch; A.out_c = R.out_c; A.rs = R.rs & ED_RUN_RS_MASK; A.lo_clamp = R.lo_clamp;
				A.hi = !EMM_NO_HI && (R.rs & ED_RUN_RS_HI) != 0; A.qsh = (R.rs & ED_RUN_RS_MASK) - 8;
#if EMM_STAMP
				A.st_ = stamp_; A.tl_p = &tl_;
#endif
				emm_layer_dispatch<FRAG_LDS>(A, lane_l); /* windows: 1, 2 or 4 (model_net_mm.c fuses nothing else) */
			}
			else if (R.kind == ED_RUN_POOL4)
			{
				const ed_net_layer_t L = EMM_NETL(li);
				/* four channels per thread: byte-wise signed maximum of dwords */
				const int c4n = L.in_c >> 2, per_img = L.out_h * L.out_w * c4n;
				const float inv_img = __builtin_amdgcn_rcpf((float)per_img), inv_c4 = __builtin_amdgcn_rcpf((float)c4n), inv_ow = __builtin_amdgcn_rcpf((float)L.out_w);
				for (int i = lane_l; i < nb * per_img; i += 64)
				{
					int b, e, pix, c4, y, x;
					emm_divmod(i, per_img, inv_img, b, e); emm_divmod(e, c4n, inv_c4, pix, c4); emm_divmod(pix, L.out_w, inv_ow, y, x);
					int m0 = -129, m1 = -129, m2 = -129, m3 = -129;
					for (int ky = 0; ky < L.kh; ky++)
					{
						const int iy = y * L.sh - L.pad_h + ky;
						if ((unsigned)iy >= (unsigned)L.in_h) continue;
						for (int kx = 0; kx < L.kw; kx++)
						{
							const int ix = x * L.sw - L.pad_w + kx;
							if ((unsigned)ix >= (unsigned)L.in_w) continue;
							const int v = EMM_LD32(a + b * R.in_img + (iy * L.in_w + ix) * L.in_c + 4 * c4);
							const int v0 = (int)(int8_t)v, v1 = (int)(int8_t)(v >> 8), v2 = (int)(int8_t)(v >> 16), v3 = v >> 24;
							m0 = v0 > m0 ? v0 : m0; m1 = v1 > m1 ? v1 : m1; m2 = v2 > m2 ? v2 : m2; m3 = v3 > m3 ? v3 : m3;
						}
					}
					EMM_ST32(o + b * R.o_img + R.o_origin + y * R.o_row + x * R.oc_pitch + 4 * c4,
					         (uint32_t)(uint8_t)m0 | ((uint32_t)(uint8_t)m1 << 8) | ((uint32_t)(uint8_t)m2 << 16) | ((uint32_t)(uint8_t)m3 << 24));
				}
			}
			else if (R.kind == ED_RUN_POOL1)
			{
				const ed_net_layer_t L = EMM_NETL(li);
				const int per_img = L.out_n;
				for (int i = lane_l; i < nb * per_img; i += 64)
				{
					const int b = i / per_img, e = i - b * per_img;
					const int pix = e / L.in_c, c = e - pix * L.in_c, y = pix / L.out_w, x = pix - y * L.out_w;
					int mx = -129;
					for (int ky = 0; ky < L.kh; ky++)
					{
						const int iy = y * L.sh - L.pad_h + ky;
						if ((unsigned)iy >= (unsigned)L.in_h) continue;
						for (int kx = 0; kx < L.kw; kx++)
						{
							const int ix = x * L.sw - L.pad_w + kx;
							if ((unsigned)ix >= (unsigned)L.in_w) continue;
							const int v = a[b * R.in_img + (iy * L.in_w + ix) * L.in_c + c];
							mx = v > mx ? v : mx;
						}
					}
					o[b * R.o_img + R.o_origin + y * R.o_row + x * R.oc_pitch + c] = (int8_t)mx;
				}
			}
			else /* softmax: arm_softmax_q7.c:215-260 */
			{
				const int in_n = (EMM_SKIP & 16) ? 0 : R.in_n;
				if (in_n <= 16)
				{
					/* at most 16 classes: image b in lane_l row b (a wave takes at most four images), one lane_l per class; maximum and
					 * sum are reductions inside the rows (DPP only), all images of the batch in one pass */
					const int b = lane_l >> 4, i = lane_l & 15;
					const bool in = i < in_n && b < nb;
					const int x = in ? (int)a[b * R.in_img + i] : -128;
					const int base = emm_row_max(x) - 8;
					int sum = emm_row_add(in ? 1 << emm_med3(x - base, 0, 7) : 0);
					sum = sum < 1 ? 1 : sum; /* a row without an image */
					int output_base, rem;
					emm_divmod(1 << 20, sum, __builtin_amdgcn_rcpf((float)sum), output_base, rem);
					if (in) o[b * R.o_img + i] = (int8_t)emm_med3(output_base >> emm_med3(13 + base - x, 0, 31), -128, 127);
				}
				else if (in_n <= 64)
				{
					/* one lane_l per class, one image after the other: maximum and sum are wave reductions, the division
					 * happens once, in float with a one-step correction (2^20 < 2^24) */
					for (int b = 0; b < nb; b++)
					{
						const bool in = lane_l < in_n;
						const int x = in ? (int)a[b * R.in_img + lane_l] : -128;
						const int base = emm_wave_max(x) - 8;
						const int sum = emm_wave_add(in ? 1 << emm_med3(x - base, 0, 7) : 0);
						int output_base, rem;
						emm_divmod(1 << 20, sum, __builtin_amdgcn_rcpf((float)sum), output_base, rem);
						if (in) o[b * R.o_img + lane_l] = (int8_t)emm_med3(output_base >> emm_med3(13 + base - x, 0, 31), -128, 127);
					}
				}
				else if (lane_l < nb)
				{
					const lds8 *v = a + lane_l * R.in_img;
					lds8 *w = o + lane_l * R.o_img;
					int base = -128;
					for (int i = 0; i < in_n; i++) base = v[i] > base ? v[i] : base;
					base -= 8;
					int sum = 0;
					for (int i = 0; i < in_n; i++) sum += 1 << emm_med3(v[i] - base, 0, 7);
					const int output_base = (1 << 20) / sum;
					for (int i = 0; i < in_n; i++) w[i] = (int8_t)emm_med3(output_base >> emm_med3(13 + base - v[i], 0, 31), -128, 127);
				}
			}
			EMM_ST(2 + 5 * li)
			emm_sync();
			/* outputs (the layouts of the logits layer's and the last layer's outputs are compact); what this pass stored is
			 * the output of layer li_out: the fused MaxPool's when there is one */
			const int li_out = (EMM_SKIP & 16) ? -1 : R.li_out;
			if (out_n <= 16)
			{
				/* at most 16 outputs: image b in lane_l row b, all images of the batch in one pass (see the Softmax above) */
				const int b = lane_l >> 4, i = lane_l & 15;
				const bool in = i < out_n && b < nb;
				if (li_out == logits_layer && logits && in) logits[(u0 + b) * out_n + i] = o[b * R.o_img + i];
				if (li_out == n_layers - 1)
				{
					if (has_softmax && softmax && in) softmax[(u0 + b) * out_n + i] = o[b * R.o_img + i];
					if (argmax)
					{
						/* first maximum: the largest (value, 63 - index) pair of the row */
						const int key = in ? (((int)o[b * R.o_img + i] + 128) << 6) | (63 - i) : -1;
						const int best = 63 - (emm_row_max(key) & 63);
						if (i == 0 && b < nb) argmax[u0 + b] = best;
					}
				}
			}
			else
			{
				if (li_out == logits_layer && logits)
					for (int b = 0; b < nb; b++)
						for (int i = lane_l; i < out_n; i += 64) logits[(u0 + b) * out_n + i] = o[b * R.o_img + i];
				if (li_out == n_layers - 1)
				{
					if (has_softmax && softmax)
						for (int b = 0; b < nb; b++)
							for (int i = lane_l; i < out_n; i += 64) softmax[(u0 + b) * out_n + i] = o[b * R.o_img + i];
					if (argmax && out_n <= 64)
					{
						/* first maximum: the largest (value, 63 - index) pair of the wave */
						for (int b = 0; b < nb; b++)
						{
							const int key = lane_l < out_n ? (((int)o[b * R.o_img + lane_l] + 128) << 6) | (63 - lane_l) : -1;
							const int best = 63 - (emm_wave_max(key) & 63);
							if (lane_l == 0) argmax[u0 + b] = best;
						}
					}
					else if (argmax && lane_l < nb)
					{
						const lds8 *v = o + lane_l * R.o_img;
						int best = 0, mx = -129;
						for (int i = 0; i < out_n; i++)
							if (v[i] > mx) { mx = v[i]; best = i; }
						argmax[u0 + lane_l] = best;
					}
				}
			}
		}
		EMM_ST(46)
		emm_sync(); /* the next batch overwrites both buffers */
	}
	/* one-window launches of the microphone path (edison_stream.hip): the only wave that had work tells the host itself, in
	 * host-mapped memory behind a system-scope fence, as ed_cnn_mfma_kernel does (a command-processor write behind the kernel
	 * costs ~2 us more). The launcher passes a flag only when the launch is one workgroup whose first wave takes every input. */
	if (done_flag && blockIdx.x == 0 && wave == 0)
	{
		__threadfence_system();
		if (lane0 == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
#if EMM_STAMP
	if (g_emm_dbg && threadIdx.x == 0 && blockIdx.x == 0) for (int i = 0; i < 48; i++) g_emm_dbg[i] = stamp_[i];
#endif
}

#if EMM_JIT
/* the kernel of ONE graph: same arguments as the general one (P is not read, of M only the tables) */
extern "C" __global__ __launch_bounds__(64 * EMM_SM_waves) void ed_net_mfma_spec(const ed_net_plan_t *__restrict__ P, const ed_mm_plan_t *__restrict__ M,
                                                                             const int8_t *__restrict__ frag, const int32_t *__restrict__ seeds,
                                                                             const int8_t *__restrict__ in, int64_t n, int64_t in_stride,
                                                                             int8_t *__restrict__ logits, int8_t *__restrict__ softmax,
                                                                             int32_t *__restrict__ argmax, unsigned *done_flag, unsigned done_seq)
{
	emm_net_body<EMM_SM_frag_mode == 2>(P, M, frag, seeds, in, n, in_stride, logits, softmax, argmax, done_flag, done_seq);
}
#else
template <bool FRAG_LDS>
__global__ __launch_bounds__(EMM_MAX_THREADS) void ed_net_mfma_kernel(const ed_net_plan_t *__restrict__ P, const ed_mm_plan_t *__restrict__ M,
                                                                 const int8_t *__restrict__ frag, const int32_t *__restrict__ seeds,
                                                                 const int8_t *__restrict__ in, int64_t n, int64_t in_stride,
                                                                 int8_t *__restrict__ logits, int8_t *__restrict__ softmax,
                                                                 int32_t *__restrict__ argmax, unsigned *done_flag, unsigned done_seq)
{
	emm_net_body<FRAG_LDS>(P, M, frag, seeds, in, n, in_stride, logits, softmax, argmax, done_flag, done_seq);
}

extern "C" int ed_launch_net_mfma(const ed_net_plan_t *dev_plan, const ed_mm_plan_t *dev_mm, const int8_t *dev_frag,
                                  const int32_t *dev_seeds, int lds_bytes, int batch, int waves, int frag_mode, const int8_t *in, int64_t n,
                                  int64_t in_stride, int8_t *logits, int8_t *softmax, int32_t *argmax, int n_cu, hipStream_t stream,
                                  unsigned *done_flag, unsigned done_seq, int *flag_written)
{
	if (flag_written) *flag_written = 0;
	if (n <= 0) return 0;
	if (waves < 1 || waves > EMM_MAX_THREADS / 64) return (int)hipErrorInvalidValue;
	int per_cu = (160 * 1024) / (lds_bytes + 256);
	if (per_cu > 32 / waves) per_cu = 32 / waves;
	if (per_cu < 1) per_cu = 1;
	const int64_t per_block = (int64_t)batch * waves;
	int64_t blocks = (n + per_block - 1) / per_block;
	if (blocks > (int64_t)n_cu * per_cu) blocks = (int64_t)n_cu * per_cu;
	const int resident = frag_mode == 2;
	const void *fn = resident ? (const void *)ed_net_mfma_kernel<true> : (const void *)ed_net_mfma_kernel<false>;
	static int max_lds_set_dev[16][2]; /* per device: the attribute belongs to the function on the current device */
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	int *max_lds_set = max_lds_set_dev[dev_ & 15];
	if (lds_bytes > max_lds_set[resident])
	{
		/* more than 64 KB of dynamic LDS has to be asked for */
		hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
		if (e != hipSuccess) return (int)e;
		max_lds_set[resident] = lds_bytes;
	}
	/* done_flag: written by the kernel itself when the launch is one workgroup whose first wave takes every input */
	unsigned *flag = (done_flag && n <= batch) ? done_flag : nullptr;
	void *kargs[] = {(void *)&dev_plan, (void *)&dev_mm, (void *)&dev_frag, (void *)&dev_seeds, (void *)&in, (void *)&n, (void *)&in_stride,
	                 (void *)&logits, (void *)&softmax, (void *)&argmax, (void *)&flag, (void *)&done_seq};
	const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(64 * waves), kargs, (size_t)lds_bytes, stream);
	if (e == hipSuccess && flag && flag_written) *flag_written = 1;
	return (int)e;
}
#endif /* !EMM_JIT */
