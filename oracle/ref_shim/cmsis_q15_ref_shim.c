/*
 * oracle/ref_shim/cmsis_q15_ref_shim.c -- TEST INFRASTRUCTURE, not product code.
 *
 * The reference's OWN fixed-point transform and square-root routines (CMSIS-DSP as vendored under
 * firmware/src/lib/CMSIS/DSP), compiled from where the sources lie into oracle/_ref/libcmsis_q15_ref.so by
 * `make -C oracle q15ref`, in the configuration the firmware builds them in (ARM_MATH_DSP: firmware/Makefile:30):
 *
 *   arm_cfft_q15                 Source/TransformFunctions/arm_cfft_q15.c:695-745
 *   arm_radix4_butterfly_q15     Source/TransformFunctions/arm_cfft_radix4_q15.c:147-563   (DSP branch)
 *   arm_split_rfft_q15           Source/TransformFunctions/arm_rfft_q15.c:241-325          (DSP branch)
 *   arm_sqrt_q31                 Source/FastMathFunctions/arm_sqrt_q31.c:50-139
 *
 * How the DSP branches compile on x86: arm_math.h, included FIRST with CMSIS's own host switch __GNUC_PYTHON__
 * (arm_math.h:373-379) and without ARM_MATH_DSP, supplies its own plain-C definitions of the Cortex-M4 DSP
 * intrinsics (__QADD16, __SHADD16, __SMUAD, __SMUSDX ...: arm_math.h:1368-1700); ARM_MATH_DSP is defined only
 * afterwards, so that the .c files included below take the branches the firmware takes, on top of the reference's
 * own C model of the instructions. No header, table or library of the reference is replaced by a stand-in: the
 * coefficient tables these routines use are passed BY POINTER by the caller (arm_common_tables.c is absent from the
 * snapshot; the tests pass the regenerated tables whose values are pinned separately), and the
 * one routine that would need an absent table -- arm_bitreversal_16 with armBitRevIndexTable_fixed_* -- is never
 * reached: the transforms are run with bitReverseFlag = 0 and the caller undoes the (plain) bit reversal.
 *
 * Nothing of the reference is copied here; this file includes the reference's sources and calls their entry points.
 */
#include "arm_math.h"

#define ARM_MATH_DSP
#include "TransformFunctions/arm_cfft_radix4_q15.c"
#include "TransformFunctions/arm_cfft_q15.c"
#include "TransformFunctions/arm_rfft_q15.c"
#include "FastMathFunctions/arm_sqrt_q31.c"

/* n complex Q15 values per frame in `buf` ([n_frames][2 * n] int16, in place), forward, no bit reversal: exactly the
 * call audioCalcMFCCs makes (arm_cfft_q15(&arm_cfft_sR_q15_len1024, bufFft, 0, 1), audioprocessing.c:139) minus the
 * table-driven reordering. tw = twiddleCoef_<n>_q15 (3n/4 complex entries). */
int q15ref_cfft(int16_t *buf, int n, const int16_t *tw, long n_frames)
{
	if (n != 16 && n != 64 && n != 256 && n != 1024 && n != 4096) return -1;
	arm_cfft_instance_q15 S;
	S.fftLen = (uint16_t)n;
	S.pTwiddle = tw;
	S.pBitRevTable = 0;
	S.bitRevLength = 0;
	for (long f = 0; f < n_frames; f++) arm_cfft_q15(&S, buf + f * 2 * n, 0, 0);
	return 0;
}

/* the real-FFT split stage of arm_rfft_q15 (arm_rfft_q15.c:111): src = n_cplx complex values in natural order per frame,
 * dst = 4 * n_cplx int16 per frame; A / B = realCoefAQ15 / realCoefBQ15 read with stride 2 * modifier */
int q15ref_split_rfft(int16_t *src, int n_cplx, const int16_t *A, const int16_t *B, int16_t *dst, int modifier, long n_frames)
{
	for (long f = 0; f < n_frames; f++)
		arm_split_rfft_q15(src + f * 2 * n_cplx, (uint32_t)n_cplx, A, B, dst + f * 4 * n_cplx, (uint32_t)modifier);
	return 0;
}

/* out[i] = arm_sqrt_q31(in[i]) (the status is dropped: negative inputs give 0 like the routine's own *pOut = 0) */
int q15ref_sqrt_q31(const int32_t *in, int32_t *out, long n)
{
	for (long i = 0; i < n; i++) (void)arm_sqrt_q31(in[i], &out[i]);
	return 0;
}

/* one value, for the enumeration tools (tools/verify/sqrt_q31_*.c) */
int32_t q15ref_sqrt_q31_one(int32_t in)
{
	q31_t o = 0;
	(void)arm_sqrt_q31(in, &o);
	return o;
}
