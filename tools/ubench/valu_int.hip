// Integer / packed-16 VALU issue cost on gfx950 (same harness as valu_ops.hip): cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(ASM)                                                                             \
	for (int it = 0; it < iters; it++)                                                        \
	{                                                                                         \
		_Pragma("unroll") for (int r = 0; r < 4; r++)                                         \
		{                                                                                     \
			asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s)); \
		}                                                                                     \
	}
#define OP8(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP8S(op, suf) op " %0, %0, %8 " suf "\n" op " %1, %1, %8 " suf "\n" op " %2, %2, %8 " suf "\n" op " %3, %3, %8 " suf "\n" op " %4, %4, %8 " suf "\n" op " %5, %5, %8 " suf "\n" op " %6, %6, %8 " suf "\n" op " %7, %7, %8 " suf "\n"
#define OP8T(op) op " %0, %0, %8, %8\n" op " %1, %1, %8, %8\n" op " %2, %2, %8, %8\n" op " %3, %3, %8, %8\n" op " %4, %4, %8, %8\n" op " %5, %5, %8, %8\n" op " %6, %6, %8, %8\n" op " %7, %7, %8, %8\n"
#define OP8Z(op) op " %0, %0, %8, 0\n" op " %1, %1, %8, 0\n" op " %2, %2, %8, 0\n" op " %3, %3, %8, 0\n" op " %4, %4, %8, 0\n" op " %5, %5, %8, 0\n" op " %6, %6, %8, 0\n" op " %7, %7, %8, 0\n"
template <int MODE> __global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned s)
{
	unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	if (MODE == 0) BODY(OP8("v_add_u32"))
	if (MODE == 1) BODY(OP8("v_xor_b32"))
	if (MODE == 2) BODY(OP8S("v_pk_add_i16", "clamp"))
	if (MODE == 3) BODY(OP8("v_pk_sub_u16"))
	if (MODE == 4) BODY(OP8("v_pk_ashrrev_i16"))
	if (MODE == 5) BODY(OP8Z("v_dot2_i32_i16"))
	if (MODE == 6) BODY(OP8T("v_perm_b32"))
	if (MODE == 7) BODY(OP8T("v_bfi_b32"))
	if (MODE == 8) BODY(OP8T("v_alignbit_b32"))
	if (MODE == 9) BODY(OP8("v_mul_hi_u32"))
	if (MODE == 10) BODY(OP8("v_mul_lo_u32"))
	if (MODE == 11) BODY(OP8("v_mul_u32_u24"))
	if (MODE == 12) BODY(OP8T("v_mad_u32_u24"))
	if (MODE == 13) BODY(OP8T("v_and_or_b32"))
	if (MODE == 14) BODY(OP8("v_ashrrev_i32"))
	if (MODE == 15) BODY(OP8T("v_lshl_add_u32"))
	if (MODE == 16) BODY(OP8("v_mul_hi_u32_u24"))
	if (MODE == 17) BODY(OP8T("v_add3_u32"))
	if (MODE == 18) BODY(OP8T("v_mad_i32_i16"))
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE> void run(const char *name)
{
	unsigned *d; hipMalloc(&d, 256 * 2048 * 4);
	const int iters = 2000, blocks = 256 * 8;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<MODE><<<blocks, 256>>>(d, 10, 3u);
	hipEventRecord(e0);
	k<MODE><<<blocks, 256>>>(d, iters, 3u);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double per_simd = (double)iters * 32 * 8 /* waves per SIMD */;
	printf("%-28s %.3f ms  -> %.2f cycles @2.4GHz per wave-instr per SIMD\n", name, ms, ms * 1e6 / per_simd * 2.4);
	hipFree(d);
}
int main()
{
	run<0>("v_add_u32"); run<1>("v_xor_b32"); run<2>("v_pk_add_i16 clamp"); run<3>("v_pk_sub_u16"); run<4>("v_pk_ashrrev_i16");
	run<5>("v_dot2_i32_i16"); run<6>("v_perm_b32"); run<7>("v_bfi_b32"); run<8>("v_alignbit_b32"); run<9>("v_mul_hi_u32");
	run<10>("v_mul_lo_u32"); run<11>("v_mul_u32_u24"); run<12>("v_mad_u32_u24"); run<13>("v_and_or_b32"); run<14>("v_ashrrev_i32");
	run<15>("v_lshl_add_u32"); run<16>("v_mul_hi_u32_u24"); run<17>("v_add3_u32"); run<18>("v_mad_i32_i16");
	return 0;
}
