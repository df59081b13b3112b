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

#define EQ_WPB 4                 /* wavefronts (= frames in flight) per workgroup                     */
#define EQ_BUF 1088              /* 1024 complex values + one pad dword per 16                        */
#define EQ_P(p) ((p) + ((p) >> 4))
#define EQ_NB 16                 /* frames whose DCT stage a wavefront runs together                  */
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

/* middle stages (:335-455) */
template <bool UNIFORM>
__device__ __forceinline__ void eq_bf_mid(u32 &a, u32 &b, u32 &c, u32 &d, const eq_tw3 &t)
{
	const u32 r = eq_qadd(a, c), s = eq_qsub(a, c), tt = eq_qadd(b, d);
	const u32 x0 = eq_asr1(eq_hadd(r, tt));
	const u32 x1 = eq_twiddle<UNIFORM>(eq_hsub(r, tt), t.w[1], t.wx[1]);
	const u32 rt = eq_swap(eq_qsub(b, d));
	u32 plus, minus;
	EQ_PLUS_MINUS_I(eq_hadd(s, rt), eq_hsub(s, rt), plus, minus);
	a = x0; b = x1;
	c = eq_twiddle<UNIFORM>(minus, t.w[0], t.wx[0]);
	d = eq_twiddle<UNIFORM>(plus, t.w[2], t.wx[2]);
}

/* last stage, no twiddles (:470-561) */
__device__ __forceinline__ void eq_bf_last(u32 &a, u32 &b, u32 &c, u32 &d)
{
	const u32 r = eq_qadd(a, c), tt = eq_qadd(b, d), s = eq_qsub(a, c), ru = eq_swap(eq_qsub(b, d));
	u32 plus, minus;
	EQ_PLUS_MINUS_I(eq_hadd(s, ru), eq_hsub(s, ru), plus, minus);
	a = eq_hadd(r, tt); b = eq_hsub(r, tt); c = minus; d = plus;
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
	const int d = (int)x + __mul24(c, __mul24(c, -2));
	const u32 t = (u32)c >> 11;
	return (u32)d + ~t < 4u * (u32)c + 1u - 2u * t; /* t + 1 <= d <= 4c + 1 - t in one unsigned compare */
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
 * batch is frame f0 + s * fstride of the launch. The output rows overwrite the mel rows.
 */
__device__ __forceinline__ void eq_dct_batch(const ed_mfcc_q15_args_t &a, int *melb, u32 *zb, int nb, uint32_t f0,
                                             uint32_t fstride, int lane, const eq_tw3 &t16, u32 rfa_l, u32 rfb_l)
{
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
				const int64_t at = (int64_t)(f0 + (uint32_t)s * fstride) * a.n_coef + c;
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
#ifndef EQ_TW_LDS
#define EQ_TW_LDS 1
#endif
#if EQ_TW_LDS
#define EQ_TW12(stage, u, regs) eq_tw_from_lds(s_tw12[stage][u], lane)
#else
#define EQ_TW12(stage, u, regs) (regs)
#endif
__device__ __forceinline__ eq_tw3 eq_tw_from_lds(const u32 (*t)[64], int lane)
{
	eq_tw3 r;
#pragma unroll
	for (int i = 0; i < 3; i++) { r.w[i] = t[i][lane]; r.wx[i] = t[3 + i][lane]; }
	return r;
}

#ifdef EQ_WAVES_PER_EU /* tuning knob: ask the register allocator for this occupancy */
#define EQ_OCCUPANCY __attribute__((amdgpu_waves_per_eu(EQ_WAVES_PER_EU, EQ_WAVES_PER_EU)))
#else
#define EQ_OCCUPANCY
#endif

template <bool STAGES, int NLO, int NHI>
__global__ __launch_bounds__(64 * EQ_WPB) EQ_OCCUPANCY void ed_mfcc_q15_kernel(ed_mfcc_q15_args_t a, const ed_q15_tables_t *__restrict__ T)
{
	__shared__ u32 s_buf[EQ_WPB][EQ_BUF];
	constexpr int NLOP = ED_Q15_PAIRS(NLO), NHIP = ED_Q15_PAIRS(NHI); /* dword reads of the int16 spectrum */
	__shared__ u32 s_tap[NLOP + NHIP][64];
	__shared__ int s_melb[EQ_WPB][EQ_NB * 32];
	__shared__ u32 s_zb[EQ_WPB][EQ_NB * 16];
	__shared__ u32 s_sqbit[1024];
#if EQ_TW_LDS
	__shared__ u32 s_tw12[2][4][6][64];
#endif
	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	u32 *buf = s_buf[w];
	int *melb = s_melb[w];
	u32 *zb = s_zb[w];

	for (int i = threadIdx.x; i < (NLOP + NHIP) * 64; i += 64 * EQ_WPB) (&s_tap[0][0])[i] = (&T->mel_tap2[0][0])[i];
	for (int i = threadIdx.x; i < 1024; i += 64 * EQ_WPB) s_sqbit[i] = T->sqbit[i];
	__syncthreads();

	/* per-lane constants of the whole run */
	const int U = lane >> 4, j3 = lane & 15;
	eq_tw3 t1[4], t2[4], t4[4];
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		t1[u] = eq_load_tw(T->tw1024, T->tw1024x, lane + 64 * u);     /* stage 1: ic = j                  */
		t2[u] = eq_load_tw(T->tw1024, T->tw1024x, 4 * (j3 + 16 * u)); /* stage 2: ic = 4 j, j = j3 + 16a  */
#if EQ_TW_LDS
		/* the per-lane coefficients of stages 1 and 2 (48 registers) live in LDS instead, one conflict-free
		 * ds_read_b32 each per frame: every wave of the workgroup needs the same values in the same lanes */
		if (w == 0)
		{
#pragma unroll
			for (int i = 0; i < 3; i++)
			{
				s_tw12[0][u][i][lane] = t1[u].w[i]; s_tw12[0][u][3 + i][lane] = t1[u].wx[i];
				s_tw12[1][u][i][lane] = t2[u].w[i]; s_tw12[1][u][3 + i][lane] = t2[u].wx[i];
			}
		}
#endif
		t4[u] = eq_load_tw(T->tw1024, T->tw1024x, 64 * u);            /* stage 4: ic = 64 j (uniform)     */
#pragma unroll
		for (int i = 0; i < 3; i++)
		{
			t4[u].w[i] = __builtin_amdgcn_readfirstlane(t4[u].w[i]);
			t4[u].wx[i] = __builtin_amdgcn_readfirstlane(t4[u].wx[i]);
		}
	}
#if EQ_TW_LDS
	__syncthreads(); /* s_tw12 was written by wave 0 */
#endif
	const eq_tw3 t3 = eq_load_tw(T->tw1024, T->tw1024x, 16 * j3);     /* stage 3: ic = 16 j               */
	const int mel_lo_pair = T->mel_lo_pair[lane], mel_hi_pair = T->mel_hi_pair[lane];
	const int mel_scale = T->mel_scale;
	const bool need_nyquist = STAGES || T->need_nyquist != 0; /* a band that reaches bin 512 (not the shipped filterbank) */
	const int rev6 = eq_bitrev(lane, 6);
	const eq_tw3 t16 = eq_load_tw(T->tw16, T->tw16x, lane & 3);
	const u32 rfa_l = T->rfa[lane & 15], rfb_l = T->rfb[lane & 15];

	const uint32_t n_frames = (uint32_t)a.n_frames, fstride = gridDim.x * EQ_WPB;
	const uint32_t f_first = blockIdx.x * EQ_WPB + (uint32_t)w;
	int slot = 0;
	/* software prefetch: the samples of frame f + fstride are requested while frame f is being transformed */
	unsigned short raw[16];
	if (f_first < n_frames)
	{
		const int16_t *src = eq_frame_ptr(a, f_first);
#pragma unroll
		for (int i = 0; i < 16; i++) raw[i] = (unsigned short)src[lane + 64 * (i >> 2) + 256 * (i & 3)];
	}
	for (uint32_t f = f_first; f < n_frames; f += fstride)
	{
		u32 e[16];

		/* ---- stage 1: real samples become (re, 0) */
#pragma unroll
		for (int i = 0; i < 16; i++) e[i] = (u32)raw[i];
		{
			/* unconditional (a conditional load would make raw[] a merge of two definitions: 16 copies per frame at the
			 * loop latch); a wave's last iteration re-reads the batch's last frame, an L2 hit */
			const uint32_t fn = (f + fstride < n_frames && f + fstride > f) ? f + fstride : n_frames - 1;
			const int16_t *src = eq_frame_ptr(a, fn);
#pragma unroll
			for (int i = 0; i < 16; i++) raw[i] = (unsigned short)src[lane + 64 * (i >> 2) + 256 * (i & 3)];
		}
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			if (!(EQ_ABLATE & 16)) eq_bf_first(e[4 * u], e[4 * u + 1], e[4 * u + 2], e[4 * u + 3], EQ_TW12(0, u, t1[u]));
#pragma unroll
			for (int q = 0; q < 4; q++) buf[EQ_P(lane + 64 * u + 256 * q)] = e[4 * u + q];
		}
		eq_wave_sync();

		if (!(EQ_ABLATE & 8))
		{
			/* ---- stages 2 + 3 on the 4x4 block e[a][b] = element 256U + j3 + 16a + 64b */
#pragma unroll
			for (int aa = 0; aa < 4; aa++)
#pragma unroll
				for (int b = 0; b < 4; b++) e[4 * aa + b] = buf[EQ_P(256 * U + j3 + 16 * aa + 64 * b)];
#pragma unroll
			for (int aa = 0; aa < 4; aa++) eq_bf_mid<false>(e[4 * aa], e[4 * aa + 1], e[4 * aa + 2], e[4 * aa + 3], EQ_TW12(1, aa, t2[aa]));
#pragma unroll
			for (int b = 0; b < 4; b++) eq_bf_mid<false>(e[b], e[4 + b], e[8 + b], e[12 + b], t3);
#pragma unroll
			for (int aa = 0; aa < 4; aa++)
#pragma unroll
				for (int b = 0; b < 4; b++) buf[EQ_P(256 * U + j3 + 16 * aa + 64 * b)] = e[4 * aa + b];
			eq_wave_sync();

			/* ---- stages 4 + 5 on the 16 consecutive elements of this lane */
#pragma unroll
			for (int m = 0; m < 16; m++) e[m] = buf[17 * lane + m]; /* EQ_P(16 lane + m) */
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

		/* ---- compact mel matrix: lane (b, r) sums quarter r of band b and of band 31-b; 32-bit wrap-around like the MCU */
		u32 acc_lo = 0, acc_hi = 0;
		const u32 *spec2 = buf; /* the int16 spectrum, two bins per dword */
		if (EQ_ABLATE & 4) { acc_lo = spec2[mel_lo_pair]; acc_hi = spec2[mel_hi_pair]; }
		else
		{
			/* two taps per v_dot2_i32_i16 (no clamp: the same wrap-around sum); magnitudes and taps fit 16 bits */
#pragma unroll
			for (int t = 0; t < NLOP; t++)
				acc_lo = (u32)__builtin_amdgcn_sdot2(eq_s(spec2[mel_lo_pair + t]), eq_s(s_tap[t][lane]), (int)acc_lo, false);
#pragma unroll
			for (int t = 0; t < NHIP; t++)
				acc_hi = (u32)__builtin_amdgcn_sdot2(eq_s(spec2[mel_hi_pair + t]), eq_s(s_tap[NLOP + t][lane]), (int)acc_hi, false);
		}
		/* both sums with one v_permlane32_swap (it exchanges halves of TWO registers: lanes 0..31 then hold the
		 * narrow band's half sums, 32..63 the wide band's), then the row pairs: rows 0,1 = band b, rows 2,3 = 31-b */
		const u32 acc = eq_sum_row_pairs(eq_fold_halves(acc_lo, acc_hi));
		if (!(lane & 16))
		{
			const int band = lane < 32 ? lane : 63 - lane; /* row 0: band b = lane, row 2: band 31 - (lane & 15) */
			const int melv = (int)(short)((int)acc / mel_scale);
			melb[32 * slot + band] = melv;
			if (STAGES && a.mel) a.mel[(int64_t)f * 32 + band] = (int16_t)melv;
		}
		eq_wave_sync();

		/* ---- dct2_q15, deferred: run it when 16 frames are parked or the wave has no frame left */
		const bool last = f + fstride >= n_frames || f + fstride < f;
		if (slot == EQ_NB - 1 || last)
		{
			if (EQ_ABLATE & 2)
			{
				if (lane < a.n_coef && a.mfcc_i16) a.mfcc_i16[(int64_t)f * a.n_coef + lane] = (int16_t)melb[32 * slot + lane];
			}
			else
				eq_dct_batch(a, melb, zb, slot + 1, f - (uint32_t)slot * fstride, fstride, lane, t16, rfa_l, rfb_l);
			slot = 0;
		}
		else
			slot++;
	}
}

static int g_q15_blocks_per_cu[2] = {-1, -1};

template <int NLO, int NHI>
static int ed_launch_q15_shape(const ed_mfcc_q15_args_t *args, const ed_q15_tables_t *dev_tab, int stages, int n_cu,
                               hipStream_t stream, int *blocks_per_cu)
{
	if (*blocks_per_cu < 0)
	{
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ed_mfcc_q15_kernel<false, NLO, NHI>, 64 * EQ_WPB, 0) != hipSuccess || nb < 1)
			nb = 2;
		const char *env = getenv("ED_Q15_BLOCKS_PER_CU"); /* tuning knob: cap the persistent grid */
		if (env && atoi(env) > 0 && atoi(env) < nb) nb = atoi(env);
		*blocks_per_cu = nb;
	}
	int64_t blocks = (args->n_frames + EQ_WPB - 1) / EQ_WPB;
	const int64_t cap = (int64_t)n_cu * *blocks_per_cu;
	if (blocks > cap) blocks = cap;
	if (blocks < 1) return 0;
	dim3 grid((unsigned)blocks), block(64 * EQ_WPB);
	if (stages) hipLaunchKernelGGL((ed_mfcc_q15_kernel<true, NLO, NHI>), grid, block, 0, stream, *args, dev_tab);
	else hipLaunchKernelGGL((ed_mfcc_q15_kernel<false, NLO, NHI>), grid, block, 0, stream, *args, dev_tab);
	return (int)hipGetLastError();
}

/* mel_nlo / mel_nhi: the host's copy of the table shape (tables_q15.c picks 6+18 or 8+24) */
extern "C" int ed_launch_mfcc_q15(const ed_mfcc_q15_args_t *args, const ed_q15_tables_t *dev_tab, int mel_nlo, int mel_nhi,
                                  int stages, int n_cu, hipStream_t stream)
{
	if (mel_nlo == 6 && mel_nhi == 18)
		return ed_launch_q15_shape<6, 18>(args, dev_tab, stages, n_cu, stream, &g_q15_blocks_per_cu[0]);
	if (mel_nlo == ED_Q15_NLO_MAX && mel_nhi == ED_Q15_NHI_MAX)
		return ed_launch_q15_shape<ED_Q15_NLO_MAX, ED_Q15_NHI_MAX>(args, dev_tab, stages, n_cu, stream, &g_q15_blocks_per_cu[1]);
	return (int)hipErrorInvalidValue;
}
