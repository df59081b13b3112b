/*
 * Exhaustive check (by enumeration) of the magnitude shortcut of mfcc_q15_kernels.hip (eq_mag_fast / eq_mag_fix).
 *
 * The firmware keeps bits 30..16 of arm_sqrt_q31(x), x = re^2 + im^2: mag(x) = arm_sqrt_q31(x) >> 16. The exact value
 * of that square root is sqrt(x * 2^31), so mag(x) is c = floor(sqrt(x / 2)) unless the routine's few-LSB error
 * carries across a multiple of 2^16. With d = x - 2 c^2 (0 <= d <= 4c + 1) and t = c >> 11 this program walks ALL
 * inputs 1 <= x < 2^31 and proves:
 *   1. t < d < 4c + 2 - t           =>  mag(x) == c                       (the kernel's fast path)
 *   2. d == 0                       =>  mag(x) is c or c - 1              (the kernel's bitmap, built by tables_q15.c)
 * Everything else goes through the full routine in the kernel. It also prints how many inputs each class holds.
 *   gcc -O2 -fopenmp -o sqrt_q31_floor sqrt_q31_floor.c -lm -ldl && ./sqrt_q31_floor [stride] [oracle/_ref/libcmsis_q15_ref.so]
 * With the library given (`make -C oracle q15ref`), arm_sqrt_q31 is the REFERENCE'S COMPILED routine (q15ref_sqrt_q31_one)
 * instead of the restated sequence below.
 */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

/* the sequence of CMSIS-DSP arm_sqrt_q31 (FastMathFunctions/arm_sqrt_q31.c:50-139) with its 64-bit products */
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
	int64_t n_fast = 0, n_bitmap = 0, n_bitmap_minus = 0, n_slow = 0, bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_fast, n_bitmap, n_bitmap_minus, n_slow, bad)
	for (int64_t x = 1; x < ((int64_t)1 << 31); x += stride)
	{
		int64_t c = (int64_t)floor(sqrt((double)x * 0.5));
		while (2 * c * c > x) c--;
		while (2 * (c + 1) * (c + 1) <= x) c++;
		const int64_t d = x - 2 * c * c, t = c >> 11;
		const int mag = (cmsis ? cmsis((int32_t)x) : ref((int32_t)x)) >> 16;
		/* the kernel's unsigned form of claim 1 */
		const uint32_t u = (uint32_t)d + ~(uint32_t)t, bound = 4u * (uint32_t)c + 1u - 2u * (uint32_t)t;
		const int fast = u < bound;
		if (fast != (d > t && d < 4 * c + 2 - t)) bad++;
		if (fast) { n_fast++; if (mag != c) bad++; }
		else if (d == 0) { n_bitmap++; if (mag == c - 1) n_bitmap_minus++; else if (mag != c) bad++; }
		else n_slow++;
	}
	printf("stride %lld, authority = %s: ", (long long)stride,
	       cmsis ? "reference object code (arm_sqrt_q31 compiled from the reference)" : "the restated sequence");
	printf("fast path %lld, bitmap %lld (of which c-1: %lld), full routine %lld, violations %lld\n", (long long)n_fast,
	       (long long)n_bitmap, (long long)n_bitmap_minus, (long long)n_slow, (long long)bad);
	return bad != 0;
}
