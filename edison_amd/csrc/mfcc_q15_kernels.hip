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
 *   __SMUAD/__SMUSDX -> v_dot2_i32_i16 against the two pre-packed forms of the twiddle
 * The five radix-4 stages of the 1024-point transform are grouped so that a lane always owns whole butterflies:
 *   stage 1        lane l owns butterflies j = l + 64u (elements j + 256q), samples straight from HBM
 *   stages 2 + 3   lane (U = l>>4, j3 = l&15) owns the 16 elements 256U + j3 + 16a + 64b: four stage-2 butterflies
 *                  over b, then four stage-3 butterflies over a, all in registers
 *   stages 4 + 5   lane l owns the 16 consecutive elements 16l .. 16l+15
 * with two exchanges through LDS in between; element p lives at dword p + (p>>4), which makes every one of the
 * access patterns above bank-conflict free. The output of the radix-4 routine is in bit-reversed order, so after
 * stage 5 register m of lane l is X[64*bitrev4(m) + bitrev6(l)]: the even registers are exactly the bins below 512.
 *
 * HBM traffic per frame: 2048 B of samples in, n_coef * (2 + 4 + 1) B out at most; the tables (8 KB) stay in L2.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/edison_hip.h"
#include "edison_internal.h"

#define EQ_WPB 4                 /* wavefronts (= frames in flight) per workgroup                     */
#define EQ_BUF 1088              /* 1024 complex values + one pad dword per 16                        */
#define EQ_P(p) ((p) + ((p) >> 4))

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
__device__ __forceinline__ u32 eq_hsub(u32 a, u32 b) { return eq_u((eq_s(a ^ b) >> (s16x2)(1)) - eq_s(~a & b)); }
__device__ __forceinline__ u32 eq_swap(u32 a) { return __builtin_amdgcn_alignbit(a, a, 16); }
__device__ __forceinline__ u32 eq_lohi(u32 lo_from, u32 hi_from) { return (lo_from & 0xffffu) | (hi_from & 0xffff0000u); }

/* x * conj(w): bits 31..16 of the two dual 16x16 multiply-accumulates */
__device__ __forceinline__ u32 eq_twiddle(u32 x, u32 w, u32 wx)
{
	const int re = __builtin_amdgcn_sdot2(eq_s(w), eq_s(x), 0, false);
	const int im = __builtin_amdgcn_sdot2(eq_s(wx), eq_s(x), 0, false);
	return ((u32)re >> 16) | ((u32)im & 0xffff0000u);
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
	const u32 x1 = eq_twiddle(eq_qsub(r, tt), t.w[1], t.wx[1]);
	const u32 rt = eq_swap(eq_qsub(b, d));
	u32 plus, minus;
	EQ_PLUS_MINUS_I(eq_qadd(s, rt), eq_qsub(s, rt), plus, minus);
	a = x0; b = x1;
	c = eq_twiddle(minus, t.w[0], t.wx[0]);
	d = eq_twiddle(plus, t.w[2], t.wx[2]);
}

/* middle stages (:335-455) */
__device__ __forceinline__ void eq_bf_mid(u32 &a, u32 &b, u32 &c, u32 &d, const eq_tw3 &t)
{
	const u32 r = eq_qadd(a, c), s = eq_qsub(a, c), tt = eq_qadd(b, d);
	const u32 x0 = eq_asr1(eq_hadd(r, tt));
	const u32 x1 = eq_twiddle(eq_hsub(r, tt), t.w[1], t.wx[1]);
	const u32 rt = eq_swap(eq_qsub(b, d));
	u32 plus, minus;
	EQ_PLUS_MINUS_I(eq_hadd(s, rt), eq_hsub(s, rt), plus, minus);
	a = x0; b = x1;
	c = eq_twiddle(minus, t.w[0], t.wx[0]);
	d = eq_twiddle(plus, t.w[2], t.wx[2]);
}

/* last stage, no twiddles (:470-561) */
__device__ __forceinline__ void eq_bf_last(u32 &a, u32 &b, u32 &c, u32 &d)
{
	const u32 r = eq_qadd(a, c), tt = eq_qadd(b, d), s = eq_qsub(a, c), ru = eq_swap(eq_qsub(b, d));
	u32 plus, minus;
	EQ_PLUS_MINUS_I(eq_hadd(s, ru), eq_hsub(s, ru), plus, minus);
	a = eq_hadd(r, tt); b = eq_hsub(r, tt); c = minus; d = plus;
}

/* arm_sqrt_q31: float seed from the exponent trick, three Newton steps on 1/sqrt, one multiply back */
__device__ __forceinline__ int eq_sqrt_q31(int in)
{
	if (in <= 0) return 0;
	const int sh = (__builtin_clz((u32)in) - 1) & ~1;
	const int number = (int)((u32)in << sh), half = number >> 1;
	const float seed = (float)number * 4.6566128731e-010f;
	const float guess = __int_as_float(0x5f3759df - (__float_as_int(seed) >> 1)) * 1073741824.0f;
	int v = (int)guess;
#pragma unroll
	for (int it = 0; it < 3; it++)
	{
		const int vv = (int)(((long long)v * v) >> 31);
		const int hv = (int)(((long long)vv * half) >> 31);
		v = (int)((u32)(int)(((long long)v * (0x30000000 - hv)) >> 31) << 2);
	}
	v = (int)((u32)(int)(((long long)number * v) >> 31) << 1);
	return v >> (sh >> 1);
}

__device__ __forceinline__ int eq_re(u32 x) { return (int)(short)(x & 0xffffu); }
__device__ __forceinline__ int eq_im(u32 x) { return (int)x >> 16; }
__device__ __forceinline__ int eq_bitrev(int v, int bits) { return (int)(__builtin_bitreverse32((u32)v) >> (32 - bits)); }

__device__ __forceinline__ void eq_wave_sync() { __builtin_amdgcn_wave_barrier(); }

template <bool STAGES>
__global__ __launch_bounds__(64 * EQ_WPB) void ed_mfcc_q15_kernel(ed_mfcc_q15_args_t a, const ed_q15_tables_t *__restrict__ T)
{
	__shared__ u32 s_buf[EQ_WPB][EQ_BUF];
	__shared__ int s_coef[ED_Q15_MEL_COEF_MAX];
	__shared__ int s_small[EQ_WPB][96]; /* mel[32] | z[16] (packed) | out[32] */
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	u32 *buf = s_buf[w];
	int *sm_mel = s_small[w], *sm_out = s_small[w] + 64;
	u32 *sm_z = (u32 *)(s_small[w] + 32);

	for (int i = threadIdx.x; i < ED_Q15_MEL_COEF_MAX; i += 64 * EQ_WPB) s_coef[i] = T->mel_coef[i];
	__syncthreads();

	/* per-lane constants of the whole run */
	const int U = lane >> 4, j3 = lane & 15;
	eq_tw3 t1[4], t2[4], t4[4];
#pragma unroll
	for (int u = 0; u < 4; u++)
	{
		t1[u] = eq_load_tw(T->tw1024, T->tw1024x, lane + 64 * u);     /* stage 1: ic = j                  */
		t2[u] = eq_load_tw(T->tw1024, T->tw1024x, 4 * (j3 + 16 * u)); /* stage 2: ic = 4 j, j = j3 + 16a  */
		t4[u] = eq_load_tw(T->tw1024, T->tw1024x, 64 * u);            /* stage 4: ic = 64 j (uniform)     */
	}
	const eq_tw3 t3 = eq_load_tw(T->tw1024, T->tw1024x, 16 * j3);     /* stage 3: ic = 16 j               */
	const int band = lane & 31, bhalf = lane >> 5;
	const int m_cnt = T->mel_count[band], m_first = (m_cnt + 1) >> 1;
	const int m_lo = bhalf ? m_first : 0, m_hi = bhalf ? m_cnt : m_first;
	const int m_spec = T->mel_start[band], m_off = T->mel_off[band];
	const int mel_scale = T->mel_scale;
	const bool need_nyquist = STAGES || T->need_nyquist != 0; /* a band that reaches bin 512 (not the shipped filterbank) */
	const int rev6 = eq_bitrev(lane, 6);
	const eq_tw3 t16 = eq_load_tw(T->tw16, T->tw16x, lane & 3);
	const u32 rfa_l = T->rfa[lane & 15], rfb_l = T->rfb[lane & 15];

	for (int64_t f = (int64_t)blockIdx.x * EQ_WPB + w; f < a.n_frames; f += (int64_t)gridDim.x * EQ_WPB)
	{
		const int16_t *src = a.audio + (f / a.frames_per_group) * a.group_stride + (f % a.frames_per_group) * a.frame_step;
		u32 e[16];

		/* ---- stage 1: real samples become (re, 0) */
#pragma unroll
		for (int u = 0; u < 4; u++)
#pragma unroll
			for (int q = 0; q < 4; q++) e[4 * u + q] = (u32)(unsigned short)src[lane + 64 * u + 256 * q];
#pragma unroll
		for (int u = 0; u < 4; u++)
		{
			eq_bf_first(e[4 * u], e[4 * u + 1], e[4 * u + 2], e[4 * u + 3], t1[u]);
#pragma unroll
			for (int q = 0; q < 4; q++) buf[EQ_P(lane + 64 * u + 256 * q)] = e[4 * u + q];
		}
		eq_wave_sync();

		/* ---- stages 2 + 3 on the 4x4 block e[a][b] = element 256U + j3 + 16a + 64b */
#pragma unroll
		for (int aa = 0; aa < 4; aa++)
#pragma unroll
			for (int b = 0; b < 4; b++) e[4 * aa + b] = buf[EQ_P(256 * U + j3 + 16 * aa + 64 * b)];
#pragma unroll
		for (int aa = 0; aa < 4; aa++) eq_bf_mid(e[4 * aa], e[4 * aa + 1], e[4 * aa + 2], e[4 * aa + 3], t2[aa]);
#pragma unroll
		for (int b = 0; b < 4; b++) eq_bf_mid(e[b], e[4 + b], e[8 + b], e[12 + b], t3);
		eq_wave_sync();
#pragma unroll
		for (int aa = 0; aa < 4; aa++)
#pragma unroll
			for (int b = 0; b < 4; b++) buf[EQ_P(256 * U + j3 + 16 * aa + 64 * b)] = e[4 * aa + b];
		eq_wave_sync();

		/* ---- stages 4 + 5 on the 16 consecutive elements of this lane */
#pragma unroll
		for (int m = 0; m < 16; m++) e[m] = buf[17 * lane + m]; /* EQ_P(16 lane + m) */
#pragma unroll
		for (int j = 0; j < 4; j++) eq_bf_mid(e[j], e[4 + j], e[8 + j], e[12 + j], t4[j]);
#pragma unroll
		for (int g = 0; g < 4; g++) eq_bf_last(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]);
		eq_wave_sync();

		/* ---- magnitudes: register m = X[64 bitrev4(m) + bitrev6(lane)]; bins 0..511 are the even registers */
		int *spec = (int *)buf;
#pragma unroll
		for (int kk = 0; kk < 8; kk++)
		{
			const int m = ((kk & 1) << 3) | ((kk & 2) << 1) | ((kk & 4) >> 1); /* bitrev4(kk), kk < 8 */
			const int re = eq_re(e[m]), im = eq_im(e[m]);
			const int mag = (int)(short)(eq_sqrt_q31((int)((u32)(re * re) + (u32)(im * im))) >> 16);
			spec[64 * kk + rev6] = mag;
			if (STAGES)
			{
				const int64_t k = 64 * kk + rev6;
				if (a.fft) { a.fft[(f * 513 + k) * 2] = (int16_t)re; a.fft[(f * 513 + k) * 2 + 1] = (int16_t)im; }
				if (a.spec) a.spec[f * 513 + k] = (int16_t)mag;
			}
		}
		if (need_nyquist && lane == 0)
		{
			const int re = eq_re(e[1]), im = eq_im(e[1]); /* X[512] */
			const int mag = (int)(short)(eq_sqrt_q31((int)((u32)(re * re) + (u32)(im * im))) >> 16);
			spec[512] = mag;
			if (STAGES && a.fft) { a.fft[(f * 513 + 512) * 2] = (int16_t)re; a.fft[(f * 513 + 512) * 2 + 1] = (int16_t)im; }
			if (STAGES && a.spec) a.spec[f * 513 + 512] = (int16_t)mag;
		}
		eq_wave_sync();

		/* ---- compact mel matrix: lane (band, half) sums half of the band's run; 32-bit wrap-around like the MCU */
		u32 acc = 0;
		for (int i = m_lo; i < m_hi; i++) acc += (u32)(spec[m_spec + i] * s_coef[m_off + i]);
		acc += (u32)__shfl_xor((int)acc, 32);
		const int melv = (int)(short)((int)acc / mel_scale);
		if (lane < 32)
		{
			sm_mel[lane] = melv;
			if (STAGES && a.mel) a.mel[f * 32 + lane] = (int16_t)melv;
		}
		eq_wave_sync();

		/* ---- dct2_q15: v[i] = mel[2i], v[31-i] = mel[2i+1]; z[n] = (v[2n], v[2n+1]); 16-point radix-4; split */
		if (lane < 4)
		{
			u32 z[4];
#pragma unroll
			for (int q = 0; q < 4; q++)
			{
				const int n = lane + 4 * q;
				const int re = n < 8 ? sm_mel[4 * n] : sm_mel[63 - 4 * n];
				const int im = n < 8 ? sm_mel[4 * n + 2] : sm_mel[61 - 4 * n];
				z[q] = ((u32)re & 0xffffu) | ((u32)im << 16);
			}
			eq_bf_first(z[0], z[1], z[2], z[3], t16);
#pragma unroll
			for (int q = 0; q < 4; q++) sm_z[lane + 4 * q] = z[q];
		}
		eq_wave_sync();
		if (lane < 4)
		{
			u32 z0 = sm_z[4 * lane], z1 = sm_z[4 * lane + 1], z2 = sm_z[4 * lane + 2], z3 = sm_z[4 * lane + 3];
			eq_bf_last(z0, z1, z2, z3);
			sm_z[4 * lane] = z0; sm_z[4 * lane + 1] = z1; sm_z[4 * lane + 2] = z2; sm_z[4 * lane + 3] = z3;
		}
		eq_wave_sync();
		if (lane < 16)
		{
			if (lane == 0)
			{
				const u32 z0 = sm_z[0];
				sm_out[0] = (int)(short)((eq_re(z0) + eq_im(z0)) >> 1);
				sm_out[16] = (int)(short)((eq_re(z0) - eq_im(z0)) >> 1);
			}
			else
			{
				const u32 p = sm_z[eq_bitrev(lane, 4)], q = sm_z[eq_bitrev(16 - lane, 4)];
				const u32 r = (u32)__builtin_amdgcn_sdot2(eq_s(rfa_l), eq_s(p), 0, false) +
				              (u32)__builtin_amdgcn_sdot2(eq_s(rfb_l), eq_s(q), 0, false);
				const int o = (int)(short)(r >> 16);
				sm_out[lane] = o;
				sm_out[32 - lane] = o;
			}
		}
		eq_wave_sync();
		if (lane < a.n_coef)
		{
			const int o = sm_out[lane];
			const int64_t at = f * a.n_coef + lane;
			if (a.mfcc_i16) a.mfcc_i16[at] = (int16_t)o;
			if (a.mfcc_f32) a.mfcc_f32[at] = (float)o;
			if (a.feat) a.feat[at] = (int8_t)(o > 127 ? 127 : (o < -128 ? -128 : o));
		}
		eq_wave_sync();
	}
}

static int g_q15_blocks_per_cu = -1;

extern "C" int ed_launch_mfcc_q15(const ed_mfcc_q15_args_t *args, const ed_q15_tables_t *dev_tab, int stages, int n_cu,
                                  hipStream_t stream)
{
	if (g_q15_blocks_per_cu < 0)
	{
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ed_mfcc_q15_kernel<false>, 64 * EQ_WPB, 0) != hipSuccess || nb < 1)
			nb = 2;
		const char *env = getenv("ED_Q15_BLOCKS_PER_CU"); /* tuning knob: cap the persistent grid */
		if (env && atoi(env) > 0 && atoi(env) < nb) nb = atoi(env);
		g_q15_blocks_per_cu = nb;
	}
	int64_t blocks = (args->n_frames + EQ_WPB - 1) / EQ_WPB;
	const int64_t cap = (int64_t)n_cu * g_q15_blocks_per_cu;
	if (blocks > cap) blocks = cap;
	if (blocks < 1) return 0;
	dim3 grid((unsigned)blocks), block(64 * EQ_WPB);
	if (stages) hipLaunchKernelGGL(ed_mfcc_q15_kernel<true>, grid, block, 0, stream, *args, dev_tab);
	else hipLaunchKernelGGL(ed_mfcc_q15_kernel<false>, grid, block, 0, stream, *args, dev_tab);
	return (int)hipGetLastError();
}
