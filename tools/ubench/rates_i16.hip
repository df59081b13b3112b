// Issue rates of the integer / packed-int16 instructions the Q15 MFCC kernel is made of (mfcc_q15_kernels.hip), in
// SHADER CYCLES per wave-instruction per SIMD, at 2 and 3 waves per SIMD (the kernel runs 3).
//   hipcc --offload-arch=gfx950 -O3 rates_i16.hip -o rates_i16
// Same method as rates.hip: inline asm on 8 independent register chains per wave, s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define X8(f) f("%0") f("%1") f("%2") f("%3") f("%4") f("%5") f("%6") f("%7")

#define I_PKADD(r) "v_pk_add_i16 " r ", " r ", %8 clamp\n"
#define I_PKSUB(r) "v_pk_sub_i16 " r ", " r ", %8 clamp\n"
#define I_PKADDU(r) "v_pk_add_u16 " r ", " r ", %8\n"
#define I_PKASHR(r) "v_pk_ashrrev_i16 " r ", 1, " r " op_sel_hi:[0,1]\n"
#define I_DOT2(r) "v_dot2_i32_i16 " r ", " r ", %8, 0\n"
#define I_PERM(r) "v_perm_b32 " r ", " r ", %8, %9\n"
#define I_BFI(r) "v_bfi_b32 " r ", %9, " r ", %8\n"
#define I_AND(r) "v_and_b32 " r ", %8, " r "\n"
#define I_XOR(r) "v_xor_b32 " r ", %8, " r "\n"
#define I_ALIGN(r) "v_alignbit_b32 " r ", " r ", " r ", 16\n"
#define I_MULHI(r) "v_mul_hi_u32 " r ", " r ", %8\n"
#define I_MUL24(r) "v_mul_i32_i24 " r ", " r ", %8\n"
#define I_CVTPK(r) "v_cvt_pk_i16_i32 " r ", " r ", %8\n"
#define I_PKMAX(r) "v_pk_max_i16 " r ", " r ", %8\n"
#define I_ASHR(r) "v_ashrrev_i32 " r ", 3, " r "\n"
#define I_MAXI(r) "v_max_i32 " r ", %8, " r "\n"
#define I_MINI(r) "v_min_i32 " r ", %8, " r "\n"
#define I_MOV(r) "v_mov_b32 " r ", %8\n"
#define I_ADDU(r) "v_add_u32 " r ", %8, " r "\n"
#define I_MED3(r) "v_med3_i32 " r ", " r ", %8, %9\n"
#define I_MAX3(r) "v_max3_i32 " r ", " r ", %8, %9\n"
#define I_DSR(r) "ds_read_b32 " r ", %10\n"
/* one middle butterfly's instruction mix, scaled to 8 chains: 3 sat add/sub, 2 and, 1 xor, 1 ashr, 2 plain add/sub, 2 dot2, 1 perm (x3 ~ 9) ... kept as
 * a repeating 12-instruction pattern */
#define I_MIX(r) I_PKADD(r) I_AND(r) I_XOR(r) I_PKASHR(r) I_PKADDU(r) I_DOT2(r) I_DOT2(r) I_PERM(r) I_PKSUB(r) I_BFI(r) I_DOT2(r) I_PERM(r)

enum { PKADD, PKSUB, PKADDU, PKASHR, DOT2, PERM, BFI, AND, XOR, ALIGN, MULHI, MUL24, MIX, CVTPK, PKMAX, ASHR, MAXI, MED3, MAX3, MINI, MOV, ADDU, NMODES };
static const char *names[NMODES] = {"v_pk_add_i16 clamp", "v_pk_sub_i16 clamp", "v_pk_add_u16", "v_pk_ashrrev_i16", "v_dot2_i32_i16 (VOP3P, 0 acc)", "v_perm_b32", "v_bfi_b32", "v_and_b32 (VOP2)", "v_xor_b32 (VOP2)", "v_alignbit_b32", "v_mul_hi_u32", "v_mul_i32_i24", "butterfly mix (12 instr)", "v_cvt_pk_i16_i32", "v_pk_max_i16", "v_ashrrev_i32 (VOP2)", "v_max_i32 (VOP2)", "v_med3_i32", "v_max3_i32", "v_min_i32 (VOP2)", "v_mov_b32 (VOP1)", "v_add_u32 (VOP2)"};

template <int MODE> __global__ __launch_bounds__(256) void k(unsigned long long *stamps, unsigned *out, int iters, unsigned s, unsigned sel)
{
	extern __shared__ unsigned dyn[];
	if (s == 12345u) dyn[threadIdx.x] = s;
	unsigned a0 = threadIdx.x * 2654435761u, a1 = a0 + 0x10001, a2 = a0 + 0x20002, a3 = a0 + 0x30003, a4 = a0 + 0x40004, a5 = a0 + 0x50005, a6 = a0 + 0x60006, a7 = a0 + 0x70007;
	unsigned long long t0, r0, t1, r1;
	asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
			if (MODE == PKADD) asm volatile(X8(I_PKADD) : OPS : "v"(s), "v"(sel));
			if (MODE == PKSUB) asm volatile(X8(I_PKSUB) : OPS : "v"(s), "v"(sel));
			if (MODE == PKADDU) asm volatile(X8(I_PKADDU) : OPS : "v"(s), "v"(sel));
			if (MODE == PKASHR) asm volatile(X8(I_PKASHR) : OPS : "v"(s), "v"(sel));
			if (MODE == DOT2) asm volatile(X8(I_DOT2) : OPS : "v"(s), "v"(sel));
			if (MODE == PERM) asm volatile(X8(I_PERM) : OPS : "v"(s), "v"(sel));
			if (MODE == BFI) asm volatile(X8(I_BFI) : OPS : "v"(s), "v"(sel));
			if (MODE == AND) asm volatile(X8(I_AND) : OPS : "v"(s), "v"(sel));
			if (MODE == XOR) asm volatile(X8(I_XOR) : OPS : "v"(s), "v"(sel));
			if (MODE == ALIGN) asm volatile(X8(I_ALIGN) : OPS : "v"(s), "v"(sel));
			if (MODE == MULHI) asm volatile(X8(I_MULHI) : OPS : "v"(s), "v"(sel));
			if (MODE == MUL24) asm volatile(X8(I_MUL24) : OPS : "v"(s), "v"(sel));
			if (MODE == MIX) asm volatile(X8(I_MIX) : OPS : "v"(s), "v"(sel));
			if (MODE == CVTPK) asm volatile(X8(I_CVTPK) : OPS : "v"(s), "v"(sel));
			if (MODE == PKMAX) asm volatile(X8(I_PKMAX) : OPS : "v"(s), "v"(sel));
			if (MODE == ASHR) asm volatile(X8(I_ASHR) : OPS : "v"(s), "v"(sel));
			if (MODE == MAXI) asm volatile(X8(I_MAXI) : OPS : "v"(s), "v"(sel));
			if (MODE == MED3) asm volatile(X8(I_MED3) : OPS : "v"(s), "v"(sel));
			if (MODE == MAX3) asm volatile(X8(I_MAX3) : OPS : "v"(s), "v"(sel));
			if (MODE == MINI) asm volatile(X8(I_MINI) : OPS : "v"(s), "v"(sel));
			if (MODE == MOV) asm volatile(X8(I_MOV) : OPS : "v"(s), "v"(sel));
			if (MODE == ADDU) asm volatile(X8(I_ADDU) : OPS : "v"(s), "v"(sel));
		}
	}
	asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if ((threadIdx.x & 63) == 0)
	{
		const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
		stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
	}
}
template <int MODE> void run(int wps, int per_chain)
{
	const int iters = 2000, blocks = 256 * wps;
	const size_t lds = ((160 * 1024 / wps) & ~(size_t)1023) - 2048; /* exactly wps workgroups fit a CU */
	(void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	unsigned long long *st; unsigned *d;
	hipMalloc(&st, sizeof(unsigned long long) * 2 * blocks * 4); hipMalloc(&d, sizeof(unsigned) * blocks * 256);
	k<MODE><<<blocks, 256, lds>>>(st, d, 200, 0x00030005u, 0x07060302u);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<MODE><<<blocks, 256, lds>>>(st, d, iters, 0x00030005u, 0x07060302u);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> h(2 * blocks * 4);
	hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
	std::vector<double> cyc, clk;
	for (int w = 0; w < blocks * 4; w++) { cyc.push_back((double)h[2 * w]); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1); }
	std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
	const double instr = (double)iters * 4 * 8 * per_chain;
	/* stamps = median wave's own cycles (with 3 waves the oldest-first arbiter lets two waves run at the per-wave limit
	 * and the third wait, so the median under-states the SIMD's cost); wall = launch time x measured clock: read that one */
	const double ck = clk[clk.size() / 2];
	printf("%-32s waves/SIMD %d: %8.3f ms  clock %.2f GHz  => cycles per instruction per SIMD: %.2f (stamps) %.2f (wall)\n", names[MODE], wps, ms, ck,
	       cyc[cyc.size() / 2] / (instr * wps), ms * 1e6 * ck / (instr * wps));
	hipFree(st); hipFree(d);
}
int main()
{
	for (int w : {2, 3})
	{
		run<PKADD>(w, 1); run<PKSUB>(w, 1); run<PKADDU>(w, 1); run<PKASHR>(w, 1); run<DOT2>(w, 1); run<PERM>(w, 1); run<BFI>(w, 1); run<AND>(w, 1); run<XOR>(w, 1);
		run<ALIGN>(w, 1); run<MULHI>(w, 1); run<MUL24>(w, 1); run<MIX>(w, 12);
		run<CVTPK>(w, 1); run<PKMAX>(w, 1); run<ASHR>(w, 1); run<MAXI>(w, 1); run<MED3>(w, 1); run<MAX3>(w, 1); run<MINI>(w, 1); run<MOV>(w, 1); run<ADDU>(w, 1);
	}
	return 0;
}
