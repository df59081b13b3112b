/*
 * mfcc_fft.h -- the register-level 512-point complex FFT machinery shared by the float MFCC kernels (mfcc_kernels.hip:
 * variants A / B, mfcc_f32_kernels.hip: variant D): radix-8 butterflies (scalar and two-frames-packed), the VALU digit
 * transpose (v_permlane32/16_swap + DPP) and the swap-based folds. Device code only; see mfcc_kernels.hip for the
 * design notes.
 */
#ifndef EDISON_MFCC_FFT_H
#define EDISON_MFCC_FFT_H
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void ed_wave_sync()
{
	/* Order this wave's LDS writes before its following LDS reads. A wave's DS instructions are issued and
	 * serviced in order, so no wait is needed; the (code-less) wave barrier only stops the compiler from
	 * moving memory operations across this point. */
	__builtin_amdgcn_wave_barrier();
}

/* x[lane] + x[lane ^ 32] in every lane: v_permlane32_swap exchanges the upper half of one register with the
 * lower half of another, so swapping a register with a copy of itself leaves (lo,lo) and (hi,hi). */
__device__ __forceinline__ float ed_sum_halves(float x)
{
	const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
	return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

/* sum over the four 16-lane rows, result in every lane: v_permlane16_swap (odd rows of one register <-> even rows
 * of the other) gives the row-pair sums, ed_sum_halves finishes. */
__device__ __forceinline__ float ed_sum_rows(float x)
{
	const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
	return ed_sum_halves(__uint_as_float(r[0]) + __uint_as_float(r[1]));
}

/*
 * One butterfly stage of an in-register transpose: for the register pair (A: register-index bit 0, B: bit 1) and lane
 * bit LB, exchange A at lanes whose bit LB is 1 with B at the partner lanes (lane ^ (1 << LB)) whose bit LB is 0.
 * Three such stages (three register-index bits against three lane bits) swap a 3-bit register index with a 3-bit
 * lane-index field: element (lane field = u, register = v) moves to (lane field = v, register = u).
 */
template <int LB>
__device__ __forceinline__ void ed_xchg(float &A, float &B, int lane)
{
	const unsigned a = __float_as_uint(A), b = __float_as_uint(B);
	if (LB == 5)
	{
		const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); /* A.upper half <-> B.lower half */
		A = __uint_as_float(r[0]); B = __uint_as_float(r[1]);
	}
	else if (LB == 4)
	{
		const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); /* A.odd rows <-> B.even rows */
		A = __uint_as_float(r[0]); B = __uint_as_float(r[1]);
	}
	else if (LB == 3)
	{
		/* row_shr:8 -> lanes 8..15 of a row (banks 2,3) read lane-8; row_shl:8 -> lanes 0..7 (banks 0,1) read lane+8.
		 * (Whole-register shifts + a select on lane bit 3 cost one instruction more: the select does not fold into a
		 * v_cndmask DPP form.) */
		const unsigned na = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x118, 0xf, 0xc, false);
		const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x108, 0xf, 0x3, false);
		A = __uint_as_float(na); B = __uint_as_float(nb);
	}
	else if (LB == 2)
	{
		/* row_shr:4 into banks 1,3 (lane bit 2 set); row_shl:4 into banks 0,2 */
		const unsigned na = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x114, 0xf, 0xa, false);
		const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x104, 0xf, 0x5, false);
		A = __uint_as_float(na); B = __uint_as_float(nb);
	}
	else
	{
		/* inside a quad no mask applies: pull the partner lane with quad_perm, select on the lane bit */
		constexpr int ctrl = (LB == 1) ? 0x4E /* [2,3,0,1] */ : 0xB1 /* [1,0,3,2] */;
		const unsigned pb = (unsigned)__builtin_amdgcn_mov_dpp((int)b, ctrl, 0xf, 0xf, false);
		const unsigned pa = (unsigned)__builtin_amdgcn_mov_dpp((int)a, ctrl, 0xf, 0xf, false);
		const bool up = (lane >> LB) & 1;
		A = __uint_as_float(up ? pb : a);
		B = __uint_as_float(up ? b : pa);
	}
}

/* swap the 3-bit register index of x[0..7] with lane bits LB0 (register bit 0), LB1 (bit 1), LB2 (bit 2) */
template <int LB0, int LB1, int LB2>
__device__ __forceinline__ void ed_transpose8(float (&x)[8], int lane)
{
#pragma unroll
	for (int i = 0; i < 8; i += 2) ed_xchg<LB0>(x[i], x[i + 1], lane);
#pragma unroll
	for (int i = 0; i < 8; i++) if (!(i & 2)) ed_xchg<LB1>(x[i], x[i + 2], lane);
#pragma unroll
	for (int i = 0; i < 4; i++) ed_xchg<LB2>(x[i], x[i + 4], lane);
}

__device__ __forceinline__ void ed_dft4(float y0r, float y0i, float y1r, float y1i, float y2r, float y2i, float y3r,
                                        float y3i, float &o0r, float &o0i, float &o1r, float &o1i, float &o2r,
                                        float &o2i, float &o3r, float &o3i)
{
	float a0r = y0r + y2r, a0i = y0i + y2i;
	float a1r = y0r - y2r, a1i = y0i - y2i;
	float a2r = y1r + y3r, a2i = y1i + y3i;
	float a3r = y1i - y3i, a3i = y3r - y1r; /* (y1 - y3) * (-i) */
	o0r = a0r + a2r; o0i = a0i + a2i;
	o2r = a0r - a2r; o2i = a0i - a2i;
	o1r = a1r + a3r; o1i = a1i + a3i;
	o3r = a1r - a3r; o3i = a1i - a3i;
}

/* In-place 8-point forward DFT, natural order in and out. */
__device__ __forceinline__ void ed_radix8(float (&r)[8], float (&i)[8])
{
	const float h = 0.70710678118654752440f;
	float ur[4], ui[4], vr[4], vi[4];
#pragma unroll
	for (int a = 0; a < 4; a++)
	{
		ur[a] = r[a] + r[a + 4]; ui[a] = i[a] + i[a + 4];
		vr[a] = r[a] - r[a + 4]; vi[a] = i[a] - i[a + 4];
	}
	ed_dft4(ur[0], ui[0], ur[1], ui[1], ur[2], ui[2], ur[3], ui[3], r[0], i[0], r[2], i[2], r[4], i[4], r[6], i[6]);
	/* odd outputs: DFT4 of (v0, v1*(1-i)/sqrt2, v2*(-i), v3*(-1-i)/sqrt2). The 1/sqrt2 of the two rotated
	 * inputs is not applied to them but carried into the last butterfly as an FMA coefficient:
	 *   y1 = h*t1, t1 = (v1r+v1i, v1i-v1r);   y3 = h*t3, t3 = (v3i-v3r, -(v3i+v3r));   y2 = (v2i, -v2r)
	 *   with u = t1+t3, w = -i*(t1-t3), a0 = v0+y2, a1 = v0-y2:  X1 = a0 + h*u, X5 = a0 - h*u, X3 = a1 + h*w,
	 *   X7 = a1 - h*w                                                                                          */
	const float t1r = vr[1] + vi[1], t1i = vi[1] - vr[1];
	const float t3r = vi[3] - vr[3], t3i = -(vi[3] + vr[3]);
	const float a0r = vr[0] + vi[2], a0i = vi[0] - vr[2];
	const float a1r = vr[0] - vi[2], a1i = vi[0] + vr[2];
	const float u_r = t1r + t3r, u_i = t1i + t3i;
	const float w_r = t1i - t3i, w_i = t3r - t1r;
	r[1] = fmaf(h, u_r, a0r);  i[1] = fmaf(h, u_i, a0i);   /* Y0 = a0 + a2 */
	r[5] = fmaf(-h, u_r, a0r); i[5] = fmaf(-h, u_i, a0i);  /* Y2 = a0 - a2 */
	r[3] = fmaf(h, w_r, a1r);  i[3] = fmaf(h, w_i, a1i);   /* Y1 = a1 + a3 */
	r[7] = fmaf(-h, w_r, a1r); i[7] = fmaf(-h, w_i, a1i);  /* Y3 = a1 - a3 */
}

typedef float ed_f2 __attribute__((ext_vector_type(2)));

/* NOT (ed_f2)(a, b): in C++ that is a cast of the comma expression, i.e. a splat of b */
__device__ __forceinline__ ed_f2 ed_mk2(float a, float b) { ed_f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ ed_f2 ed_splat(float x) { return ed_mk2(x, x); }
__device__ __forceinline__ ed_f2 ed_fma2(ed_f2 a, ed_f2 b, ed_f2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ void ed_dft4_2(ed_f2 y0r, ed_f2 y0i, ed_f2 y1r, ed_f2 y1i, ed_f2 y2r, ed_f2 y2i, ed_f2 y3r, ed_f2 y3i,
                                          ed_f2 &o0r, ed_f2 &o0i, ed_f2 &o1r, ed_f2 &o1i, ed_f2 &o2r, ed_f2 &o2i, ed_f2 &o3r, ed_f2 &o3i)
{
	ed_f2 a0r = y0r + y2r, a0i = y0i + y2i;
	ed_f2 a1r = y0r - y2r, a1i = y0i - y2i;
	ed_f2 a2r = y1r + y3r, a2i = y1i + y3i;
	ed_f2 a3r = y1i - y3i, a3i = y3r - y1r;
	o0r = a0r + a2r; o0i = a0i + a2i;
	o2r = a0r - a2r; o2i = a0i - a2i;
	o1r = a1r + a3r; o1i = a1i + a3i;
	o3r = a1r - a3r; o3i = a1i - a3i;
}

/* ed_radix8 on two frames at once */
__device__ __forceinline__ void ed_radix8_2(ed_f2 (&r)[8], ed_f2 (&i)[8])
{
	const ed_f2 h = ed_splat(0.70710678118654752440f), nh = ed_splat(-0.70710678118654752440f);
	ed_f2 ur[4], ui[4], vr[4], vi[4];
#pragma unroll
	for (int a = 0; a < 4; a++)
	{
		ur[a] = r[a] + r[a + 4]; ui[a] = i[a] + i[a + 4];
		vr[a] = r[a] - r[a + 4]; vi[a] = i[a] - i[a + 4];
	}
	ed_dft4_2(ur[0], ui[0], ur[1], ui[1], ur[2], ui[2], ur[3], ui[3], r[0], i[0], r[2], i[2], r[4], i[4], r[6], i[6]);
	const ed_f2 t1r = vr[1] + vi[1], t1i = vi[1] - vr[1];
	const ed_f2 t3r = vi[3] - vr[3], t3i = -(vi[3] + vr[3]);
	const ed_f2 a0r = vr[0] + vi[2], a0i = vi[0] - vr[2];
	const ed_f2 a1r = vr[0] - vi[2], a1i = vi[0] + vr[2];
	const ed_f2 u_r = t1r + t3r, u_i = t1i + t3i;
	const ed_f2 w_r = t1i - t3i, w_i = t3r - t1r;
	r[1] = ed_fma2(h, u_r, a0r);  i[1] = ed_fma2(h, u_i, a0i);
	r[5] = ed_fma2(nh, u_r, a0r); i[5] = ed_fma2(nh, u_i, a0i);
	r[3] = ed_fma2(h, w_r, a1r);  i[3] = ed_fma2(h, w_i, a1i);
	r[7] = ed_fma2(nh, w_r, a1r); i[7] = ed_fma2(nh, w_i, a1i);
}

/* the VALU transpose works on 32-bit registers: once per frame */
template <int LB0, int LB1, int LB2>
__device__ __forceinline__ void ed_transpose8_2(ed_f2 (&x)[8], int lane)
{
	float a[8], b[8];
#pragma unroll
	for (int i = 0; i < 8; i++) { a[i] = x[i].x; b[i] = x[i].y; }
	ed_transpose8<LB0, LB1, LB2>(a, lane);
	ed_transpose8<LB0, LB1, LB2>(b, lane);
#pragma unroll
	for (int i = 0; i < 8; i++) x[i] = ed_mk2(a[i], b[i]);
}

/* v_permlane32_swap exchanges a's upper half with b's lower half; the sum of the two results is
 * lanes 0..31: a[l] + a[l+32], lanes 32..63: b[l-32] + b[l] -- both registers folded by one swap and one add */
__device__ __forceinline__ float ed_fold_halves(float a, float b)
{
	const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
	return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
/* v_permlane16_swap exchanges a's odd rows with b's even rows; the sum is
 * even rows R: a[R] + a[R+1], odd rows R: b[R-1] + b[R] */
__device__ __forceinline__ float ed_fold_rows(float a, float b)
{
	const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
	return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#endif
