/*
 * Exhaustive equivalence proof (by enumeration) for the GPU's formulation of arm_sqrt_q31.
 *
 * ref():  the sequence of CMSIS-DSP arm_sqrt_q31 (FastMathFunctions/arm_sqrt_q31.c:50-139) with its 64-bit products.
 * fast(): what mfcc_q15_kernels.hip executes: every (a*b)>>31 becomes one unsigned high multiply with one operand
 *         pre-doubled, valid as long as every intermediate stays in [0, 2^31).
 * The program walks ALL 2^31-1 positive inputs (a superset of re^2+im^2) and compares the full 32-bit results.
 *   gcc -O2 -fopenmp -o sqrt_q31_equiv sqrt_q31_equiv.c -ldl && ./sqrt_q31_equiv [stride] [oracle/_ref/libcmsis_q15_ref.so]
 * With the library given (built by `make -C oracle q15ref` from the reference's CMSIS-DSP sources), the authority is the
 * REFERENCE'S COMPILED arm_sqrt_q31 itself (q15ref_sqrt_q31_one), and the restated ref() below is checked against it too.
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static inline int32_t ref(int32_t in)
{
	if (in <= 0) return 0;
	int sb = __builtin_clz((uint32_t)in) - 1;
	int sh = (sb & 1) ? sb - 1 : sb;
	int32_t number = (int32_t)((uint32_t)in << sh), half = number >> 1, keep = number;
	union { int32_t i; float f; } cv;
	cv.f = (float)number * 4.6566128731e-010f;
	cv.i = 0x5f3759df - (cv.i >> 1);
	int32_t v = (int32_t)(cv.f * 1073741824.0f);
	for (int it = 0; it < 3; it++)
	{
		int32_t vv = (int32_t)(((int64_t)v * v) >> 31);
		int32_t hv = (int32_t)(((int64_t)vv * (int64_t)half) >> 31);
		v = (int32_t)((uint32_t)(int32_t)(((int64_t)v * (int64_t)(0x30000000 - hv)) >> 31) << 2);
	}
	v = (int32_t)((uint32_t)(int32_t)(((int64_t)keep * v) >> 31) << 1);
	return v >> (sh / 2);
}

static inline uint32_t mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

static inline int32_t fast(int32_t in)
{
	if (in <= 0) return 0;
	const int sh = (__builtin_clz((uint32_t)in) - 1) & ~1;
	const uint32_t number = (uint32_t)in << sh, number2 = number & ~1u; /* 2 * (number >> 1) */
	union { int32_t i; float f; } cv;
	cv.f = (float)(int32_t)number * 4.6566128731e-010f;
	cv.i = 0x5f3759df - (cv.i >> 1);
	uint32_t v = (uint32_t)(int32_t)(cv.f * 1073741824.0f);
	for (int it = 0; it < 3; it++)
	{
		const uint32_t vv = mulhi(v << 1, v);
		const uint32_t hv = mulhi(vv, number2);
		v = mulhi(v, (0x30000000u - hv) << 1) << 2;
	}
	v = mulhi(number << 1, v) << 1;
	return (int32_t)v >> (sh >> 1);
}

typedef int32_t (*sqrt_fn)(int32_t);

int main(int argc, char **argv)
{
	const int64_t stride = argc > 1 ? atoll(argv[1]) : 1;
	sqrt_fn cmsis = NULL;
	if (argc > 2)
	{
		void *h = dlopen(argv[2], RTLD_LAZY | RTLD_LOCAL);
		cmsis = h ? (sqrt_fn)dlsym(h, "q15ref_sqrt_q31_one") : NULL;
		if (!cmsis) { fprintf(stderr, "cannot load q15ref_sqrt_q31_one from %s: %s\n", argv[2], dlerror()); return 2; }
	}
	int64_t bad = 0, bad_restatement = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad, bad_restatement)
	for (int64_t x = 1; x < ((int64_t)1 << 31); x += stride)
	{
		const int32_t want = cmsis ? cmsis((int32_t)x) : ref((int32_t)x);
		if (want != fast((int32_t)x)) bad++;
		if (cmsis && want != ref((int32_t)x)) bad_restatement++;
	}
	printf("stride %lld, authority = %s: %lld mismatches of the GPU formulation", (long long)stride,
	       cmsis ? "reference object code (arm_sqrt_q31 compiled from the reference)" : "the restated sequence", (long long)bad);
	if (cmsis) printf(", %lld of the restatement", (long long)bad_restatement);
	printf("\n");
	return (bad != 0) | (bad_restatement != 0);
}
