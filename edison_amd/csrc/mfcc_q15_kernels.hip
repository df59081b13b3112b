/*
 * mfcc_q15_kernels.hip -- MFCC variant C on gfx950: the firmware's fixed-point audioCalcMFCCs
 * (firmware/src/audioprocessing.c:116-215), bit for bit, one wavefront per 1024-sample frame.
 *
 * Reference arithmetic restated here (all integer, so the result is exact, not "within tolerance"):
 *   arm_cfft_q15 len 1024   = arm_radix4_butterfly_q15, ARM_MATH_DSP branch (arm_cfft_radix4_q15.c:147-563) + bit reversal
 *   cmpl_mag_q15            = arm_sqrt_q31(re^2 + im^2) >> 16            (audioprocessing.c:299-312, arm_sqrt_q31.c:50-139)
 *   compact mel matrix      = 32-bit dot product / MEL_MTX_SCALE -> q15   (audioprocessing.c:158-172)
 *   dct2_q15                = even/odd reorder, 32-point arm_rfft_q15, real parts (audioprocessing.c:330-436,
 *                             arm_rfft_q15.c:76-123,241-325)
 *
 * Mapping onto a wavefront. A complex Q15 value is one dword (re in the low halfword, like the firmware's
 * read_q15x2), so the Cortex-M4 SIMD instructions of the DSP branch have direct CDNA4 counterparts:
 *   __QADD16/__QSUB16 -> v_pk_add_i16/v_pk_sub_i16 clamp      __SHADD16(x,0) -> v_pk_ashrrev_i16
 *   __SHADD16/__SHSUB16 -> and/xor + v_pk_ashrrev_i16 + v_pk_add/sub_u16 (overflow-free floor average)
 *   __SMUAD/__SMUSDX -> v_dot2_i32_i16 against the two pre-packed forms of the twiddle, v_perm_b32 packs bits 31..16
 * The five radix-4 stages of the 1024-point transform are grouped so that a lane always owns whole butterflies:
 *   stage 1        lane l owns butterflies j = l + 64u (elements j + 256q), samples straight from HBM
 *   stages 2 + 3   lane (U = l>>4, j3 = l&15) owns the 16 elements 256U + j3 + 16a + 64b: four stage-2 butterflies
 *                  over b, then four stage-3 butterflies over a, all in registers
 *   stages 4 + 5   lane l owns the 16 consecutive elements 16l .. 16l+15
 * with two exchanges through LDS in between; element p lives at dword p + (p>>4), which makes every one of the
 * access patterns above bank-conflict free. The output of the radix-4 routine is in bit-reversed order, so after
 * stage 5 register m of lane l is X[64*bitrev4(m) + bitrev6(l)]: the even registers are exactly the bins below 512.
 *
 * arm_sqrt_q31's 64-bit products (a*b)>>31 are single v_mul_hi_u32 with one operand pre-doubled; that this is the
 * same function on all 2^31-1 positive inputs is checked by enumeration (tools/verify/sqrt_q31_equiv.c).
 * The mel sums are spread over the wavefront (see edison_internal.h) and the small DCT stage -- a 16-point complex
 * FFT that would keep 4 of 64 lanes busy -- is deferred: a wavefront parks the 32 mel values of each frame in LDS and
 * runs the DCT stage for 16 frames at once.
 *
 * HBM traffic per frame: 2048 B of samples in, n_coef * (2 + 4 + 1) B out at most; the tables (15 KB) stay in L2.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

/* ---- lab knobs: only a lab build (ED_LAB, tools/lab/mkvariant.py) may set them; the product build has none, and
 * tests/test_host_cpu.py checks the values below against what edison_amd/build.py compiles */
#if !defined(ED_LAB) && (defined(EQ_WPB) || defined(EQ_PRIO) || defined(EQ_ABLATE) || defined(EQ_WAVES_PER_EU))
#error "EQ_* lab knob defined without ED_LAB (tools/lab/mkvariant.py builds lab variants)"
#endif
#if defined(ED_LAB)
/* a lab build says so: the product library exports no ed_lab_build_* symbol (tests/test_host_cpu.py) */
extern "C" { extern const int ed_lab_build_mfcc_q15; const int ed_lab_build_mfcc_q15 = 1; }
#endif
#ifndef EQ_WPB
#define EQ_WPB 16                /* wavefronts (= frames in flight) per workgroup: one workgroup per CU, 4 waves per SIMD */
#endif
#define EQ_BUF 1088              /* 1024 complex values + one pad dword per 16                        */
#define EQ_P(p) ((p) + ((p) >> 4))
#define EQ_NB 16                 /* frames whose DCT stage a wavefront runs together                  */
/* Wave priorities rising with the progress through a frame (see ED2_PRIO in mfcc_kernels.hip; +2.9 ... +3.2 % here), two bits per
 * point (product table 0,1,1,2,2,3): 0 top of the loop (global loads + reads of stages 2+3), 1 stage 2+3 arithmetic, 2 its write-back + reads of 4+5,
 * 3 stage 4+5 + magnitudes, 4 spectrum store + mel, 5 the next frame's stage 1 (at priority 0 the gain is gone). 0 = none. */
#ifndef EQ_PRIO
#define EQ_PRIO 0xe94
#endif
#if EQ_PRIO
#define EQ_PR(pt) __builtin_amdgcn_s_setprio((EQ_PRIO >> (2 * (pt))) & 3)
#else
#define EQ_PR(pt) ((void)0)
#endif
/* timing-only ablations for A/B work (results are WRONG when non-zero): 1 no sqrt, 2 no DCT stage, 4 no mel taps,
 * 8 no FFT stages 2-5, 16 no stage-1 butterflies */
#ifndef EQ_ABLATE
#define EQ_ABLATE 0
#endif

typedef unsigned int u32;
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 eq_s(u32 x) { return __builtin_bit_cast(s16x2, x); }
__device__ __forceinline__ u32 eq_u(s16x2 x) { return __builtin_bit_cast(u32, x); }

/* both halfwords at once */
__device__ __forceinline__ u32 eq_qadd(u32 a, u32 b) { return eq_u(__builtin_elementwise_add_sat(eq_s(a), eq_s(b))); }
__device__ __forceinline__ u32 eq_qsub(u32 a, u32 b) { return eq_u(__builtin_elementwise_sub_sat(eq_s(a), eq_s(b))); }
__device__ __forceinline__ u32 eq_asr1(u32 a) { return eq_u(eq_s(a) >> (s16x2)(1)); }
__device__ __forceinline__ u32 eq_asr2(u32 a) { return eq_u(eq_s(a) >> (s16x2)(2)); }
/* floor((a+b)/2) = (a&b) + ((a^b)>>1); floor((a-b)/2) = ((a^b)>>1) - (~a&b): no 17th bit needed */
__device__ __forceinline__ u32 eq_hadd(u32 a, u32 b) { return eq_u(eq_s(a & b) + (eq_s(a ^ b) >> (s16x2)(1))); }
__device__ __forceinline__ u32 eq_hsub(u32 a, u32 b) { return eq_u((eq_s(a ^ b) >> (s16x2)(1)) - eq_s((a ^ b) & b)); } /* (a^b)&b == ~a&b */
/* floor((a-b)/2) from h = floor((a+b)/2): h - b, exactly (the value fits 16 bits, so the wrap-around subtract is it) */
__device__ __forceinline__ u32 eq_hsub_from_hadd(u32 h, u32 b) { return eq_u(eq_s(h) - eq_s(b)); }
__device__ __forceinline__ u32 eq_swap(u32 a) { return __builtin_amdgcn_alignbit(a, a, 16); }
__device__ __forceinline__ u32 eq_lohi(u32 lo_from, u32 hi_from) { return (lo_from & 0xffffu) | (hi_from & 0xffff0000u); }

/* a.lo*b.lo + a.hi*b.hi, 32-bit wrap-around. The VOP3P form with an inline 0 accumulator: the builtin selects the
 * accumulating VOP2 form and spends a v_mov on the zero. UNIFORM = the coefficient lives in an SGPR. */
template <bool UNIFORM>
__device__ __forceinline__ int eq_dot2(u32 coef, u32 x)
{
	int r;
	if (UNIFORM) asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "s"(coef), "v"(x));
	else asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(coef), "v"(x));
	return r;
}

/* x * conj(w): bits 31..16 of the two dual 16x16 multiply-accumulates */
template <bool UNIFORM>
__device__ __forceinline__ u32 eq_twiddle(u32 x, u32 w, u32 wx)
{
	const u32 re = (u32)eq_dot2<UNIFORM>(w, x), im = (u32)eq_dot2<UNIFORM>(wx, x);
	return __builtin_amdgcn_perm(im, re, 0x07060302u); /* (re >> 16) | (im & 0xffff0000) */
}

struct eq_tw3 { u32 w[3], wx[3]; }; /* pairs ic, 2ic, 3ic */

__device__ __forceinline__ eq_tw3 eq_load_tw(const u32 *w, const u32 *wx, int ic)
{
	eq_tw3 t;
#pragma unroll
	for (int i = 0; i < 3; i++) { t.w[i] = w[(i + 1) * ic]; t.wx[i] = wx[(i + 1) * ic]; }
	return t;
}

/* s + i*t and s - i*t from the packed sum and difference with the halfword-swapped t */
#define EQ_PLUS_MINUS_I(sum, dif, plus, minus) \
	do { (plus) = eq_lohi((dif), (sum)); (minus) = eq_lohi((sum), (dif)); } while (0)

/* first stage: inputs >> 2 (arm_cfft_radix4_q15.c:181-321) */
__device__ __forceinline__ void eq_bf_first(u32 &a, u32 &b, u32 &c, u32 &d, const eq_tw3 &t)
{
	a = eq_asr2(a); b = eq_asr2(b); c = eq_asr2(c); d = eq_asr2(d);
	const u32 r = eq_qadd(a, c), s = eq_qsub(a, c), tt = eq_qadd(b, d);
	const u32 x0 = eq_hadd(r, tt);
	const u32 x1 = eq_twiddle<false>(eq_qsub(r, tt), t.w[1], t.wx[1]);
	const u32 rt = eq_swap(eq_qsub(b, d));
	u32 plus, minus;
	EQ_PLUS_MINUS_I(eq_qadd(s, rt), eq_qsub(s, rt), plus, minus);
	a = x0; b = x1;
	c = eq_twiddle<false>(minus, t.w[0], t.wx[0]);
	d = eq_twiddle<false>(plus, t.w[2], t.wx[2]);
}

/*
 * The same first stage on the REAL samples of a frame, two butterflies at a time: with every imaginary part zero the
 * saturating adds and the halving run on (sample of butterfly A, sample of butterfly B) pairs -- a dword of two
 * neighbouring samples as it comes from memory. What the general routine computes with rt = (0, b-d), s = (a-c, 0):
 *   x0 = ((r + tt) >> 1, 0)          x1 = (r - tt, 0) * conj(w2)
 *   plus = (s, q)                    minus = (s, -q)                    q = b - d, in [-16383, 16383] after the >> 2
 * so x1 needs one product per part (the coefficient of butterfly A sits in the low half of its dword with a zero above
 * it, B's in the high half: v_dot2 then picks the right sample of the pair by itself) and plus / minus are two
 * v_perm_b32 each. tA / tB: coefficient sets of the two butterflies, index 1 (the 2 ic pair) in that one-sided form.
 */
__device__ __forceinline__ void eq_bf_first_real2(u32 a0, u32 a1, u32 a2, u32 a3, const eq_tw3 &tA, const eq_tw3 &tB,
                                                  u32 (&oA)[4], u32 (&oB)[4])
{
	a0 = eq_asr2(a0); a1 = eq_asr2(a1); a2 = eq_asr2(a2); a3 = eq_asr2(a3);
	const u32 r = eq_qadd(a0, a2), s = eq_qsub(a0, a2), tt = eq_qadd(a1, a3), q = eq_qsub(a1, a3);
	const u32 x0 = eq_hadd(r, tt), x1 = eq_qsub(r, tt);
	const u32 nq = eq_u((s16x2)(0) - eq_s(q));
	oA[0] = x0 & 0xffffu; oB[0] = x0 >> 16;
	oA[1] = __builtin_amdgcn_perm((u32)eq_dot2<false>(tA.wx[1], x1), (u32)eq_dot2<false>(tA.w[1], x1), 0x07060302u);
	oB[1] = __builtin_amdgcn_perm((u32)eq_dot2<false>(tB.wx[1], x1), (u32)eq_dot2<false>(tB.w[1], x1), 0x07060302u);
	oA[2] = eq_twiddle<false>(__builtin_amdgcn_perm(nq, s, 0x05040100u), tA.w[0], tA.wx[0]);
	oB[2] = eq_twiddle<false>(__builtin_amdgcn_perm(nq, s, 0x07060302u), tB.w[0], tB.wx[0]);
	oA[3] = eq_twiddle<false>(__builtin_amdgcn_perm(q, s, 0x05040100u), tA.w[2], tA.wx[2]);
	oB[3] = eq_twiddle<false>(__builtin_amdgcn_perm(q, s, 0x07060302u), tB.w[2], tB.wx[2]);
}

/* the samples of one frame as stage 1 wants them: x[4 v + q] = samples (2 lane + 128 v + 256 q, and the next one) */
template <bool ALIGNED>
__device__ __forceinline__ void eq_load_frame(const int16_t *src, int lane, u32 (&x)[8])
{
#pragma unroll
	for (int i = 0; i < 8; i++)
	{
		const int at = 2 * lane + 128 * (i >> 2) + 256 * (i & 3);
		/* read once: non-temporal (see ed_load_frame in mfcc_one_frame.h; +1.0 % here). */
		if (ALIGNED) x[i] = __builtin_nontemporal_load(reinterpret_cast<const u32 *>(src + at));
		else x[i] = (u32)(unsigned short)src[at] | ((u32)(unsigned short)src[at + 1] << 16);
	}
}

/* middle stages (:335-455) */
template <bool UNIFORM>
__device__ __forceinline__ void eq_bf_mid(u32 &a, u32 &b, u32 &c, u32 &d, const eq_tw3 &t)
{
	const u32 r = eq_qadd(a, c), s = eq_qsub(a, c), tt = eq_qadd(b, d);
	const u32 h = eq_hadd(r, tt);
	const u32 x0 = eq_asr1(h);
	const u32 x1 = eq_twiddle<UNIFORM>(eq_hsub_from_hadd(h, tt), t.w[1], t.wx[1]);
	const u32 rt = eq_swap(eq_qsub(b, d));
	const u32 hs = eq_hadd(s, rt);
	u32 plus, minus;
	EQ_PLUS_MINUS_I(hs, eq_hsub_from_hadd(hs, rt), plus, minus);
	a = x0; b = x1;
	c = eq_twiddle<UNIFORM>(minus, t.w[0], t.wx[0]);
	d = eq_twiddle<UNIFORM>(plus, t.w[2], t.wx[2]);
}

/* last stage, no twiddles (:470-561) */
__device__ __forceinline__ void eq_bf_last(u32 &a, u32 &b, u32 &c, u32 &d)
{
	const u32 r = eq_qadd(a, c), tt = eq_qadd(b, d), s = eq_qsub(a, c), ru = eq_swap(eq_qsub(b, d));
	const u32 hs = eq_hadd(s, ru), h = eq_hadd(r, tt);
	u32 plus, minus;
	EQ_PLUS_MINUS_I(hs, eq_hsub_from_hadd(hs, ru), plus, minus);
	a = h; b = eq_hsub_from_hadd(h, tt); c = minus; d = plus;
}

__device__ __forceinline__ u32 eq_mulhi(u32 a, u32 b)
{
	u32 r;
	asm("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}

/* arm_sqrt_q31: float seed from the exponent trick, three Newton steps on 1/sqrt, one multiply back. Every
 * intermediate stays in [0, 2^31), so (a*b)>>31 == mulhi(2a, b) (tools/verify/sqrt_q31_equiv.c enumerates it). */
__device__ __forceinline__ int eq_sqrt_q31(int in_raw)
{
	/* branch-free (the eight square roots of a lane are independent chains the scheduler can interleave):
	 * non-positive inputs run the sequence on 1 and select 0 at the end */
	const int in = in_raw > 0 ? in_raw : 1;
	const int sh = (__builtin_clz((u32)in) - 1) & ~1;
	const u32 number = (u32)in << sh, number2 = number & ~1u;
	const float seed = (float)(int)number * 4.6566128731e-010f;
	const float guess = __int_as_float(0x5f3759df - (__float_as_int(seed) >> 1)) * 1073741824.0f;
	u32 v = (u32)(int)guess;
	/* the multiplies are opaque to the optimiser on purpose: left alone it rewrites "mulhi << 2" into a 64-bit
	 * multiply plus funnel shifts and masks, twice the instructions */
#pragma unroll
	for (int it = 0; it < 3; it++)
	{
		const u32 vv = eq_mulhi(v + v, v);
		const u32 hv = eq_mulhi(vv, number2);
		v = eq_mulhi(v, 0x60000000u - (hv + hv)) << 2; /* (0x30000000 - hv) << 1 */
	}
	v = eq_mulhi(number + number, v) << 1;
	return in_raw > 0 ? (int)v >> (sh >> 1) : 0;
}

/* The firmware keeps mag = arm_sqrt_q31(x) >> 16 of x = re^2 + im^2. The routine approximates sqrt(x * 2^31), so
 * mag is c = floor(sqrt(x / 2)) unless its few-LSB error crosses a multiple of 2^16. tools/verify/sqrt_q31_floor.c
 * walks all 2^31 inputs: with d = x - 2 c^2 and t = c >> 11, t < d < 4c + 2 - t implies mag == c. The test also
 * certifies c itself (0 <= d < 4c + 2), so the 1-ulp v_sqrt_f32 only has to be right almost always: whatever fails
 * the test goes to eq_mag_fix. Returns true when c is proven. */
__device__ __forceinline__ bool eq_mag_fast(u32 x, int &c)
{
	c = (int)__builtin_amdgcn_sqrtf((float)x * 0.5f);
	const u32 c2 = (u32)c + (u32)c;                       /* c <= 46340: both factors fit 24 bits, the product 32 */
	const u32 d = x - __umul24(c2, (u32)c);
	const u32 t = (u32)c >> 11;
	return d + ~t <= ((c2 - t) << 1); /* t + 1 <= d <= 4c + 1 - t in one unsigned compare */
}

/* The rest: x == 2 c^2 (mag is c or c - 1: one bit per c from tables_q15.c) and the rare near-boundary inputs, the
 * wrapped sum 0x80000000 and a mis-rounded c, which run the routine itself. */
__device__ __forceinline__ int eq_mag_fix(u32 x, int c, const u32 *sqbit)
{
	if (x == 2u * (u32)__mul24(c, c) && c < 32768) return c - (int)((sqbit[c >> 5] >> (c & 31)) & 1u);
	return eq_sqrt_q31((int)x) >> 16;
}

#define EQ_BR4(kk) ((((kk) & 1) << 3) | (((kk) & 2) << 1) | (((kk) & 4) >> 1)) /* bitrev4(kk), kk < 8 */

__device__ __forceinline__ int eq_re(u32 x) { return (int)(short)(x & 0xffffu); }
__device__ __forceinline__ int eq_im(u32 x) { return (int)x >> 16; }
__device__ __forceinline__ int eq_bitrev(int v, int bits) { return (int)(__builtin_bitreverse32((u32)v) >> (32 - bits)); }

/* Orders this wave's LDS writes before its following LDS reads for the compiler; the hardware services a wave's DS
 * instructions in order. */
__device__ __forceinline__ void eq_wave_sync() { __builtin_amdgcn_wave_barrier(); }

/* sum over the four 16-lane rows in every lane (wrap-around): v_permlane16_swap, then v_permlane32_swap */
/* lanes 0..31: a[l] + a[l+32], lanes 32..63: b[l-32] + b[l] (a's upper half is exchanged with b's lower half) */
__device__ __forceinline__ u32 eq_fold_halves(u32 a, u32 b)
{
	const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
	return q[0] + q[1];
}
/* every row gets the sum of its row pair (rows 0+1, rows 2+3) */
__device__ __forceinline__ u32 eq_sum_row_pairs(u32 x)
{
	const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
	return r[0] + r[1];
}

/* First sample of frame f: (f / fpg) * group_stride + (f % fpg) * frame_step (f is wave-uniform, < 2^31). */
__device__ __forceinline__ const int16_t *eq_frame_ptr(const ed_mfcc_q15_args_t &a, uint32_t f)
{
	uint32_t g = 0, i = f;
	if (a.frames_per_group < a.n_frames)
	{
		g = f / (uint32_t)a.frames_per_group;
		i = f - g * (uint32_t)a.frames_per_group;
	}
	return a.audio + ((int64_t)g * a.group_stride + (int64_t)i * a.frame_step);
}

/*
 * dct2_q15 for nb <= 16 parked frames of this wavefront (mel rows in melb[s][32]): v[i] = mel[2i], v[31-i] = mel[2i+1];
 * z[n] = (v[2n], v[2n+1]); 16-point radix-4 transform (first + last stage); real-FFT split; real parts. Frame s of the
 * batch is frame fid[s] of the launch (the work queue hands a wave frames in no fixed pattern). The output rows
 * overwrite the mel rows.
 */
__device__ __forceinline__ void eq_dct_batch(const ed_mfcc_q15_args_t &a, int *melb, u32 *zb, int nb, const u32 *fid,
                                             int lane, const ed_q15_tables_t *__restrict__ T)
{
	/* the lane number is made opaque here: everything this stage derives from it (a dozen and a half addresses and indices) would
	 * otherwise be hoisted out of the frame loop and, at 128 registers, spilled -- 20 MB of scratch writes per launch */
	asm volatile("" : "+v"(lane));
	/* once per 16 frames: the constants of this stage are fetched here (L2 hits) instead of holding 8 registers */
	const eq_tw3 t16 = eq_load_tw(T->tw16, T->tw16x, lane & 3);
	const u32 rfa_l = T->rfa[lane & 15], rfb_l = T->rfb[lane & 15];
	{
		const int s = lane >> 2, j = lane & 3;
		const int *mel = melb + 32 * s;
		u32 *zs = zb + 16 * s;
		if (s < nb)
		{
			u32 z[4];
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				const int n = j + 4 * q;
				const int re = n < 8 ? mel[4 * n] : mel[63 - 4 * n];
				const int im = n < 8 ? mel[4 * n + 2] : mel[61 - 4 * n];
				z[q] = ((u32)re & 0xffffu) | ((u32)im << 16);
			}
			eq_bf_first(z[0], z[1], z[2], z[3], t16);
#pragma unroll
			for (int q = 0; q < 4; q++) zs[j + 4 * q] = z[q];
		}
		eq_wave_sync();
		if (s < nb)
		{
			u32 z0 = zs[4 * j], z1 = zs[4 * j + 1], z2 = zs[4 * j + 2], z3 = zs[4 * j + 3];
			eq_bf_last(z0, z1, z2, z3);
			zs[4 * j] = z0; zs[4 * j + 1] = z1; zs[4 * j + 2] = z2; zs[4 * j + 3] = z3;
		}
		eq_wave_sync();
	}
	{
		const int i = lane & 15;
#pragma unroll
		for (int p = 0; p < EQ_NB / 4; p++)
		{
			const int s = 4 * p + (lane >> 4);
			if (s < nb)
			{
				const u32 *zs = zb + 16 * s;
				int *out = melb + 32 * s;
				if (i == 0)
				{
					const u32 z0 = zs[0];
					out[0] = (int)(short)((eq_re(z0) + eq_im(z0)) >> 1);
					out[16] = (int)(short)((eq_re(z0) - eq_im(z0)) >> 1);
				}
				else
				{
					const u32 pz = zs[eq_bitrev(i, 4)], qz = zs[eq_bitrev(16 - i, 4)];
					const u32 r = (u32)eq_dot2<false>(rfa_l, pz) + (u32)eq_dot2<false>(rfb_l, qz);
					const int o = (int)(short)(r >> 16);
					out[i] = o;
					out[32 - i] = o;
				}
			}
		}
		eq_wave_sync();
	}
	{
		const int c = lane & 31;
#pragma unroll
		for (int p = 0; p < EQ_NB / 2; p++)
		{
			const int s = 2 * p + (lane >> 5);
			if (s < nb && c < a.n_coef)
			{
				const int o = melb[32 * s + c];
				const int64_t at = (int64_t)fid[s] * a.n_coef + c;
				if (a.mfcc_i16) a.mfcc_i16[at] = (int16_t)o;
				if (a.mfcc_f32) a.mfcc_f32[at] = (float)o;
				if (a.feat) a.feat[at] = (int8_t)(o > 127 ? 127 : (o < -128 ? -128 : o));
			}
		}
		eq_wave_sync();
	}
}

/* 1: the per-lane coefficients of stages 1 and 2 are read from LDS in every frame instead of living in 48 VGPRs:
 * 216 -> 167 registers, 3 waves per SIMD instead of 2, +6.7 % (48 more conflict-free ds_read_b32 per frame) */
/* 2 (round 3): stage 3's six as well -- 106 registers, 4 waves per SIMD. Without wave priorities the fourth wave bought nothing
 * (89.8 us at 12 and at 16 waves); with them 89.9 -> 87.1 us (+3.3 %, profiles/r03_wave_priorities.txt) */
#define EQ_TW12(stage, u, regs) eq_tw_from_lds(s_tw12 + ((stage) * 4 + (u)) * 6 * 64, lane)
#define EQ_TW3(regs) eq_tw_from_lds(s_tw12 + 8 * 6 * 64, lane)
__device__ __forceinline__ eq_tw3 eq_tw_from_lds(const u32 *t, int lane)
{
	eq_tw3 r;
#pragma unroll
	for (int i = 0; i < 3; i++) { r.w[i] = t[64 * i + lane]; r.wx[i] = t[64 * (3 + i) + lane]; }
	return r;
}

#ifndef EQ_WAVES_PER_EU /* the occupancy the register allocator is asked for */
#define EQ_WAVES_PER_EU (EQ_WPB / 4)
#endif
#if EQ_WAVES_PER_EU
#define EQ_OCCUPANCY __attribute__((amdgpu_waves_per_eu(EQ_WAVES_PER_EU, EQ_WAVES_PER_EU)))
#else
#define EQ_OCCUPANCY
#endif

/* LDS of one workgroup, in dwords: mel taps, the x == 2c^2 bits, the stage-1/2 coefficients, then per wave the transform
 * buffer, 16 parked mel rows, the DCT scratch and the parked frames' numbers; one dword of work queue at the end */
#define EQ_TW_DWORDS ((2 * 4 + 1) * 6 * 64) /* coefficient sets of stages 1, 2 (four each) and 3 (one), 6 x 64 dwords a set */
#define EQ_LDS_DWORDS(nlop, nhip) (((nlop) + (nhip)) * 64 + 1024 + EQ_TW_DWORDS + EQ_WPB * (EQ_BUF + EQ_NB * 32 + EQ_NB * 16 + EQ_NB) + 4)

/*
 * One workgroup per CU (EQ_WPB wavefronts, 4 per SIMD at 106 VGPRs) owns a contiguous slice of the launch's frames and
 * hands them to its wavefronts through a counter in LDS: the vector ALU arbitrates oldest-first, so with a fixed
 * frame -> wave assignment the old waves of a SIMD finish long before the young ones and the tail of the launch runs
 * at one or two waves per SIMD (SQ_WAVE_CYCLES / SQ_WAVES was 78 % of the kernel's busy time); drawn from a queue, all
 * waves of a CU finish within one frame of each other. The tables are staged once per CU instead of once per 4 waves.
 */
template <bool STAGES, bool ALIGNED, int NLO, int NHI>
__global__ __launch_bounds__(64 * EQ_WPB) EQ_OCCUPANCY void ed_mfcc_q15_kernel(ed_mfcc_q15_args_t a, const ed_q15_tables_t *__restrict__ T)
{
	extern __shared__ __align__(16) u32 eq_smem[];
	constexpr int NLOP = ED_Q15_PAIRS(NLO), NHIP = ED_Q15_PAIRS(NHI); /* dword reads of the int16 spectrum */
	u32 *s_tap = eq_smem;                        /* [NLOP + NHIP][64] */
	u32 *s_sqbit = s_tap + (NLOP + NHIP) * 64;   /* [1024]            */
	u32 *s_tw12 = s_sqbit + 1024;                /* [2][4][6][64]     */
	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	u32 *buf = s_tw12 + EQ_TW_DWORDS + w * (EQ_BUF + EQ_NB * 32 + EQ_NB * 16 + EQ_NB);
	int *melb = reinterpret_cast<int *>(buf + EQ_BUF);
	u32 *zb = buf + EQ_BUF + EQ_NB * 32;
	u32 *fid = zb + EQ_NB * 16;
	unsigned *queue = s_tw12 + EQ_TW_DWORDS + EQ_WPB * (EQ_BUF + EQ_NB * 32 + EQ_NB * 16 + EQ_NB);
	/* its LDS byte address (the kernel has no static LDS: dynamic LDS starts at 0), for the asm draw */
	const uint32_t queue_addr = (uint32_t)(sizeof(u32) * ((NLOP + NHIP) * 64 + 1024 + EQ_TW_DWORDS + EQ_WPB * (EQ_BUF + EQ_NB * 32 + EQ_NB * 16 + EQ_NB)));

	for (int i = threadIdx.x; i < (NLOP + NHIP) * 64; i += 64 * EQ_WPB) s_tap[i] = (&T->mel_tap2[0][0])[i];
	for (int i = threadIdx.x; i < 1024; i += 64 * EQ_WPB) s_sqbit[i] = T->sqbit[i];
	if (threadIdx.x == 0) *queue = 2 * EQ_WPB; /* the first two frames of every wave are handed out by position */
	__syncthreads();

	/* per-lane constants of the whole run */
	const int U = lane >> 4, j3 = lane & 15;
	eq_tw3 t1[4], t2[4], t4[4];
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		/* stage 1: ic = j, lane l owns butterflies j = 2 l + (u & 1) + 128 (u >> 1); the 2 ic pair in its one-sided form */
		t1[u] = eq_load_tw(T->tw1024, T->tw1024x, 2 * lane + (u & 1) + 128 * (u >> 1));
		t1[u].w[1] = (u & 1) ? t1[u].w[1] << 16 : t1[u].w[1] & 0xffffu;
		t1[u].wx[1] = (u & 1) ? t1[u].wx[1] << 16 : t1[u].wx[1] & 0xffffu;
		t2[u] = eq_load_tw(T->tw1024, T->tw1024x, 4 * (j3 + 16 * u)); /* stage 2: ic = 4 j, j = j3 + 16a  */
		/* the per-lane coefficients of stages 1 and 2 (48 registers) live in LDS instead, one conflict-free
		 * ds_read_b32 each per frame: every wave of the workgroup needs the same values in the same lanes */
		if (w == 0)
		{
#pragma unroll
			for (int i = 0; i < 3; i++)
			{
				s_tw12[((0 * 4 + u) * 6 + i) * 64 + lane] = t1[u].w[i]; s_tw12[((0 * 4 + u) * 6 + 3 + i) * 64 + lane] = t1[u].wx[i];
				s_tw12[((1 * 4 + u) * 6 + i) * 64 + lane] = t2[u].w[i]; s_tw12[((1 * 4 + u) * 6 + 3 + i) * 64 + lane] = t2[u].wx[i];
			}
		}
		t4[u] = eq_load_tw(T->tw1024, T->tw1024x, 64 * u);            /* stage 4: ic = 64 j (uniform)     */
#pragma unroll
		for (int i = 0; i < 3; i++)
		{
			t4[u].w[i] = __builtin_amdgcn_readfirstlane(t4[u].w[i]);
			t4[u].wx[i] = __builtin_amdgcn_readfirstlane(t4[u].wx[i]);
		}
	}
	const eq_tw3 t3 = eq_load_tw(T->tw1024, T->tw1024x, 16 * j3);     /* stage 3: ic = 16 j               */
	if (w == 1 % EQ_WPB)
	{
#pragma unroll
		for (int i = 0; i < 3; i++) { s_tw12[(8 * 6 + i) * 64 + lane] = t3.w[i]; s_tw12[(8 * 6 + 3 + i) * 64 + lane] = t3.wx[i]; }
	}
	__syncthreads(); /* s_tw12 was written by waves 0 and 1 */
	const int mel_lo_pair = T->mel_lo_pair[lane], mel_hi_pair = T->mel_hi_pair[lane];
	const int mel_scale = T->mel_scale;
	/* the firmware divides by MEL_MTX_SCALE = 128 (mel_constants.h:7): a power of two is an add and a shift (C's division
	 * truncates towards zero), anything else the compiler's 30-instruction expansion */
	const bool scale_pow2 = mel_scale > 0 && (mel_scale & (mel_scale - 1)) == 0;
	const int scale_sh = __builtin_ctz((unsigned)mel_scale | 0x80000000u);
	const bool need_nyquist = STAGES || T->need_nyquist != 0; /* a band that reaches bin 512 (not the shipped filterbank) */
	const int rev6 = eq_bitrev(lane, 6);

	/* this workgroup's slice of the launch, and this wave's first two frames of it */
	const uint32_t n_frames = (uint32_t)a.n_frames;
	const uint32_t s0 = (uint32_t)(((uint64_t)blockIdx.x * n_frames) / gridDim.x);
	const uint32_t cnt = (uint32_t)(((uint64_t)(blockIdx.x + 1) * n_frames) / gridDim.x) - s0;
	uint32_t i_cur = (uint32_t)w, i_next = (uint32_t)w + EQ_WPB;
	int slot = 0;
	/* software prefetch: the samples of the wave's next frame are requested at the top of an iteration and go through STAGE 1
	 * at its bottom (round 3), into the wave's transform buffer, which the mel stage has released by then: what an iteration
	 * hands to the next one lies in LDS, not in registers. (With stage 1 at the top, the prefetch wrote a second set of 8
	 * registers that the loop latch copied back: 8 v_mov_b32 per frame.) */
	u32 raw[8];
	/* ---- stage 1 on sample pairs: butterflies j = 2 lane + 128 v (low halves) and j + 1 (high halves) */
	auto stage1 = [&](const u32 (&x)[8]) {
#pragma unroll
		for (int v = 0; v < 2; v++)
		{
			u32 oA[4], oB[4];
			if (!(EQ_ABLATE & 16))
				eq_bf_first_real2(x[4 * v], x[4 * v + 1], x[4 * v + 2], x[4 * v + 3], EQ_TW12(0, 2 * v, t1[2 * v]), EQ_TW12(0, 2 * v + 1, t1[2 * v + 1]), oA, oB);
			else
			{
#pragma unroll
				for (int q = 0; q < 4; q++) { oA[q] = x[4 * v + q] & 0xffffu; oB[q] = x[4 * v + q] >> 16; }
			}
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				const int p = 2 * lane + 128 * v + 256 * q; /* even: p and p + 1 share their pad */
				buf[EQ_P(p)] = oA[q]; buf[EQ_P(p) + 1] = oB[q];
			}
		}
	};
	if (i_cur < cnt)
	{
		eq_load_frame<ALIGNED>(eq_frame_ptr(a, s0 + i_cur), lane, raw);
		stage1(raw);
	}
	while (i_cur < cnt)
	{
		const uint32_t f = s0 + i_cur;
		u32 e[16];
		EQ_PR(0);
		/* the next frame's samples: unconditional (a wave's last iteration re-reads the slice's last frame, an L2 hit, and
		 * never uses it) */
		eq_load_frame<ALIGNED>(eq_frame_ptr(a, s0 + (i_next < cnt ? i_next : cnt - 1)), lane, raw);
		eq_wave_sync();

		if (!(EQ_ABLATE & 8))
		{
			/* ---- stages 2 + 3 on the 4x4 block e[a][b] = element 256U + j3 + 16a + 64b */
#pragma unroll
			for (int aa = 0; aa < 4; aa++)
#pragma unroll
				for (int b = 0; b < 4; b++) e[4 * aa + b] = buf[EQ_P(256 * U + j3 + 16 * aa + 64 * b)];
			EQ_PR(1);
#pragma unroll
			for (int aa = 0; aa < 4; aa++) eq_bf_mid<false>(e[4 * aa], e[4 * aa + 1], e[4 * aa + 2], e[4 * aa + 3], EQ_TW12(1, aa, t2[aa]));
#pragma unroll
			for (int b = 0; b < 4; b++) eq_bf_mid<false>(e[b], e[4 + b], e[8 + b], e[12 + b], EQ_TW3(t3));
			EQ_PR(2);
#pragma unroll
			for (int aa = 0; aa < 4; aa++)
#pragma unroll
				for (int b = 0; b < 4; b++) buf[EQ_P(256 * U + j3 + 16 * aa + 64 * b)] = e[4 * aa + b];
			eq_wave_sync();

			/* ---- stages 4 + 5 on the 16 consecutive elements of this lane */
#pragma unroll
			for (int m = 0; m < 16; m++) e[m] = buf[17 * lane + m]; /* EQ_P(16 lane + m) */
			EQ_PR(3);
#pragma unroll
			for (int j = 0; j < 4; j++) eq_bf_mid<true>(e[j], e[4 + j], e[8 + j], e[12 + j], t4[j]);
#pragma unroll
			for (int g = 0; g < 4; g++) eq_bf_last(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]);
		}
		eq_wave_sync();

		/* ---- magnitudes: register m = X[64 bitrev4(m) + bitrev6(lane)]; bins 0..511 are the even registers */
		short *spec = (short *)buf; /* int16: the mel stage reads two bins per dword */
		int mag[8];
		if (EQ_ABLATE & 1)
		{
#pragma unroll
			for (int kk = 0; kk < 8; kk++) mag[kk] = (int)((e[EQ_BR4(kk)] ^ (e[EQ_BR4(kk)] >> 16)) & 0x7fff);
		}
		else
		{
			/* re^2 + im^2 (32-bit wrap-around, like the firmware's q31 sum) is one dot2 of the value with itself */
			u32 pw[8];
			bool ok[8], all_ok = true;
#pragma unroll
			for (int kk = 0; kk < 8; kk++)
			{
				pw[kk] = (u32)eq_dot2<false>(e[EQ_BR4(kk)], e[EQ_BR4(kk)]);
				ok[kk] = eq_mag_fast(pw[kk], mag[kk]);
				all_ok &= ok[kk];
			}
			if (__builtin_amdgcn_ballot_w64(!all_ok) != 0)
			{
#pragma unroll
				for (int kk = 0; kk < 8; kk++)
					if (!ok[kk]) mag[kk] = eq_mag_fix(pw[kk], mag[kk], s_sqbit);
			}
		}
		EQ_PR(4);
#pragma unroll
		for (int kk = 0; kk < 8; kk++)
		{
			spec[64 * kk + rev6] = (short)mag[kk];
			if (STAGES)
			{
				const int64_t k = 64 * kk + rev6;
				const u32 v = e[EQ_BR4(kk)];
				if (a.fft) { a.fft[((int64_t)f * 513 + k) * 2] = (int16_t)eq_re(v); a.fft[((int64_t)f * 513 + k) * 2 + 1] = (int16_t)eq_im(v); }
				if (a.spec) a.spec[(int64_t)f * 513 + k] = (int16_t)mag[kk];
			}
		}
		if (need_nyquist && lane == 0)
		{
			const int re = eq_re(e[1]), im = eq_im(e[1]); /* X[512] */
			const u32 pw512 = (u32)eq_dot2<false>(e[1], e[1]);
			int mag512;
			if (!eq_mag_fast(pw512, mag512)) mag512 = eq_mag_fix(pw512, mag512, s_sqbit);
			spec[512] = (short)mag512;
			if (STAGES && a.fft) { a.fft[((int64_t)f * 513 + 512) * 2] = (int16_t)re; a.fft[((int64_t)f * 513 + 512) * 2 + 1] = (int16_t)im; }
			if (STAGES && a.spec) a.spec[(int64_t)f * 513 + 512] = (int16_t)mag512;
		}
		eq_wave_sync();

		/* the frame after the next one is drawn from the queue here, where few registers are live; the answer is needed
		 * at the bottom of the loop, a mel stage later */
		uint32_t drawn = 0;
		if (lane == 0)
		{
			/* one exec-masked ds_add_rtn_u32 (the builtin goes through the compiler's wave-aggregation code and waits on the spot) */
			asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(drawn) : "v"(queue_addr), "v"(1u) : "memory");
			fid[slot] = f;
		}

		/* ---- compact mel matrix: lane (b, r) sums quarter r of band b and of band 31-b; 32-bit wrap-around like the MCU */
		u32 acc_lo = 0, acc_hi = 0;
		const u32 *spec2 = buf; /* the int16 spectrum, two bins per dword */
		if (EQ_ABLATE & 4) { acc_lo = spec2[mel_lo_pair]; acc_hi = spec2[mel_hi_pair]; }
		else
		{
			/* two taps per v_dot2_i32_i16 (no clamp: the same wrap-around sum); magnitudes and taps fit 16 bits */
#pragma unroll
			for (int t = 0; t < NLOP; t++)
				acc_lo = (u32)__builtin_amdgcn_sdot2(eq_s(spec2[mel_lo_pair + t]), eq_s(s_tap[64 * t + lane]), (int)acc_lo, false);
#pragma unroll
			for (int t = 0; t < NHIP; t++)
				acc_hi = (u32)__builtin_amdgcn_sdot2(eq_s(spec2[mel_hi_pair + t]), eq_s(s_tap[64 * (NLOP + t) + lane]), (int)acc_hi, false);
		}
		/* both sums with one v_permlane32_swap (it exchanges halves of TWO registers: lanes 0..31 then hold the
		 * narrow band's half sums, 32..63 the wide band's), then the row pairs: rows 0,1 = band b, rows 2,3 = 31-b */
		const u32 acc = eq_sum_row_pairs(eq_fold_halves(acc_lo, acc_hi));
		if (!(lane & 16))
		{
			const int band = lane < 32 ? lane : 63 - lane; /* row 0: band b = lane, row 2: band 31 - (lane & 15) */
			const int quot = scale_pow2 ? ((int)acc + (((int)acc >> 31) & (mel_scale - 1))) >> scale_sh : (int)acc / mel_scale;
			const int melv = (int)(short)quot;
			melb[32 * slot + band] = melv;
			if (STAGES && a.mel) a.mel[(int64_t)f * 32 + band] = (int16_t)melv;
		}
		eq_wave_sync();

		/* ---- the next frame's stage 1 (see the prologue): the transform buffer is free from here on */
		EQ_PR(5);
		if (i_next < cnt) stage1(raw);

		/* ---- dct2_q15, deferred: run it when 16 frames are parked or the wave has no frame left */
		const bool last = i_next >= cnt;
		if (slot == EQ_NB - 1 || last)
		{
			if (EQ_ABLATE & 2)
			{
				if (lane < a.n_coef && a.mfcc_i16) a.mfcc_i16[(int64_t)f * a.n_coef + lane] = (int16_t)melb[32 * slot + lane];
			}
			else
				eq_dct_batch(a, melb, zb, slot + 1, fid, lane, T);
			slot = 0;
		}
		else
			slot++;
		/* the draw was issued a mel stage ago; a wave's DS operations complete in order and the mel stage's reads came
		 * behind it, but the compiler does not know about the asm's result being in flight */
		asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(drawn));
		i_cur = i_next; i_next = __builtin_amdgcn_readfirstlane(drawn);
	}
}

static int g_q15_blocks_per_cu[16][8]; /* per device (0 = not asked yet): the LDS attribute belongs to the function on the current device */

template <int NLO, int NHI>
static int ed_launch_q15_shape(const ed_mfcc_q15_args_t *args, const ed_q15_tables_t *dev_tab, int stages, int n_cu,
                               hipStream_t stream, int *blocks_per_cu /* [4]: stages x aligned */)
{
	/* 4-byte loads of sample pairs need every frame start 4-byte aligned */
	const bool aligned = ((reinterpret_cast<uintptr_t>(args->audio) & 3) == 0) && (args->frame_step % 2 == 0) && (args->group_stride % 2 == 0);
	const void *fn = stages ? (aligned ? (const void *)ed_mfcc_q15_kernel<true, true, NLO, NHI> : (const void *)ed_mfcc_q15_kernel<true, false, NLO, NHI>)
	                        : (aligned ? (const void *)ed_mfcc_q15_kernel<false, true, NLO, NHI> : (const void *)ed_mfcc_q15_kernel<false, false, NLO, NHI>);
	const size_t lds = sizeof(u32) * EQ_LDS_DWORDS(ED_Q15_PAIRS(NLO), ED_Q15_PAIRS(NHI));
	int *bpc = &blocks_per_cu[(stages ? 2 : 0) + (aligned ? 1 : 0)];
	if (*bpc <= 0)
	{
		/* more than 64 KB of dynamic LDS has to be asked for, once per kernel instance and device */
		if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return (int)hipGetLastError();
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * EQ_WPB, lds) != hipSuccess || nb < 1) nb = 1;
		const char *env = getenv("ED_Q15_BLOCKS_PER_CU"); /* tuning knob: cap the persistent grid */
		if (env && atoi(env) > 0 && atoi(env) < nb) nb = atoi(env);
		*bpc = nb;
	}
	int64_t blocks = (args->n_frames + EQ_WPB - 1) / EQ_WPB;
	const int64_t cap = (int64_t)n_cu * *bpc;
	if (blocks > cap) blocks = cap;
	if (blocks < 1) return 0;
	void *kargs[] = {(void *)args, (void *)&dev_tab};
	return (int)hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(64 * EQ_WPB), kargs, lds, stream);
}

/* mel_nlo / mel_nhi: the host's copy of the table shape (tables_q15.c picks 6+18 or 8+24) */
extern "C" int ed_launch_mfcc_q15(const ed_mfcc_q15_args_t *args, const ed_q15_tables_t *dev_tab, int mel_nlo, int mel_nhi,
                                  int stages, int n_cu, hipStream_t stream)
{
	int dev_ = 0;
	(void)hipGetDevice(&dev_);
	dev_ &= 15;
	if (mel_nlo == 6 && mel_nhi == 18)
		return ed_launch_q15_shape<6, 18>(args, dev_tab, stages, n_cu, stream, &g_q15_blocks_per_cu[dev_][0]);
	if (mel_nlo == ED_Q15_NLO_MAX && mel_nhi == ED_Q15_NHI_MAX)
		return ed_launch_q15_shape<ED_Q15_NLO_MAX, ED_Q15_NHI_MAX>(args, dev_tab, stages, n_cu, stream, &g_q15_blocks_per_cu[dev_][4]);
	return (int)hipErrorInvalidValue;
}
