// VALU issue cost by instruction class on gfx950 (8 waves/SIMD, 8 independent chains per wave, inline asm so the
// compiler cannot fold anything). Prints cycles per wave-instruction per SIMD assuming 2.4 GHz.
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
#define OP8U(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define OP8S(op, suf) op " %0, %0 " suf "\n" op " %1, %1 " suf "\n" op " %2, %2 " suf "\n" op " %3, %3 " suf "\n" op " %4, %4 " suf "\n" op " %5, %5 " suf "\n" op " %6, %6 " suf "\n" op " %7, %7 " suf "\n"
#define OP8SW(op) op " %0, %1\n" op " %2, %3\n" op " %4, %5\n" op " %6, %7\n" op " %0, %2\n" op " %1, %3\n" op " %4, %6\n" op " %5, %7\n"
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	if (MODE == 0) BODY(OP8("v_add_f32"))
	if (MODE == 1) BODY(OP8("v_and_b32"))
	if (MODE == 2) BODY(OP8U("v_mov_b32"))
	if (MODE == 3) BODY(OP8U("v_cvt_f32_i32"))
	if (MODE == 4) BODY(OP8S("v_mov_b32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
	if (MODE == 5) BODY(OP8S("v_mov_b32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xa"))
	if (MODE == 6) BODY(OP8SW("v_permlane32_swap_b32"))
	if (MODE == 7) BODY(OP8SW("v_permlane16_swap_b32"))
	if (MODE == 8) BODY(OP8("v_cndmask_b32"))   // uses vcc implicitly via e32? use 3-operand form below
	if (MODE == 9) BODY(OP8U("v_sqrt_f32"))
	if (MODE == 10) BODY(OP8("v_lshlrev_b32"))
	if (MODE == 11) BODY(OP8("v_mul_f32"))
	if (MODE == 12) BODY(OP8S("v_add_f32_dpp", ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE> void run(const char *name)
{
	float *d; hipMalloc(&d, 256 * 2048 * 4);
	const int iters = 2000, blocks = 256 * 8;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<MODE><<<blocks, 256>>>(d, 10, 1.0f);
	hipEventRecord(e0);
	k<MODE><<<blocks, 256>>>(d, iters, 1.0f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double per_simd = (double)iters * 32 * 8 /* waves per SIMD */;
	printf("%-28s %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
	hipFree(d);
}
int main()
{
	run<0>("v_add_f32"); run<11>("v_mul_f32"); run<1>("v_and_b32"); run<10>("v_lshlrev_b32"); run<2>("v_mov_b32"); run<3>("v_cvt_f32_i32");
	run<8>("v_cndmask_b32 (vcc)"); run<4>("v_mov_b32_dpp quad_perm"); run<5>("v_mov_b32_dpp row_shr"); run<12>("v_add_f32_dpp quad_perm");
	run<6>("v_permlane32_swap"); run<7>("v_permlane16_swap"); run<9>("v_sqrt_f32");
	return 0;
}
