// Does the issue rate of packed fp32 instructions on gfx950 depend on WHICH registers their operands are (VGPR banks)?
// Explicit register numbers in inline asm, 8 independent instructions per group, 4 waves per SIMD (launch_bounds 1024), shader cycles
// per instruction per SIMD. Operand "bank pair" of an aligned 64-bit register pair v[2k:2k+1] is k & 1 if there are four banks (index mod 4).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59"
// destinations v[40:55] (8 pairs); sources chosen per mode
#define G8(op, s0a, s1a, s0b, s1b) \
	op " v[40:41], " s0a ", " s1a "\n" op " v[42:43], " s0b ", " s1b "\n" op " v[44:45], " s0a ", " s1a "\n" op " v[46:47], " s0b ", " s1b "\n" \
	op " v[48:49], " s0a ", " s1a "\n" op " v[50:51], " s0b ", " s1b "\n" op " v[52:53], " s0a ", " s1a "\n" op " v[54:55], " s0b ", " s1b "\n"
#define G8F(s0a, s1a, s2a, s0b, s1b, s2b) \
	"v_pk_fma_f32 v[40:41], " s0a ", " s1a ", " s2a "\nv_pk_fma_f32 v[42:43], " s0b ", " s1b ", " s2b "\nv_pk_fma_f32 v[44:45], " s0a ", " s1a ", " s2a "\nv_pk_fma_f32 v[46:47], " s0b ", " s1b ", " s2b "\n" \
	"v_pk_fma_f32 v[48:49], " s0a ", " s1a ", " s2a "\nv_pk_fma_f32 v[50:51], " s0b ", " s1b ", " s2b "\nv_pk_fma_f32 v[52:53], " s0a ", " s1a ", " s2a "\nv_pk_fma_f32 v[54:55], " s0b ", " s1b ", " s2b "\n"
enum { ADD_SAME, ADD_DIFF, ADD_SAMEREG, MUL_SAME, MUL_DIFF, FMA_000, FMA_001, FMA_011_DIFFSRC, FMA_SAME2, ADD32_SAME, ADD32_DIFF, NM };
static const char *nm[NM] = {"v_pk_add_f32  src pairs v[12:13], v[16:17]  (same bank pair)", "v_pk_add_f32  src pairs v[12:13], v[18:19]  (different bank pairs)",
	"v_pk_add_f32  src0 = src1 = v[12:13]", "v_pk_mul_f32  same bank pair", "v_pk_mul_f32  different bank pairs",
	"v_pk_fma_f32  v[12:13], v[16:17], v[20:21]  (all one bank pair)", "v_pk_fma_f32  v[12:13], v[16:17], v[22:23]  (two + one)", "v_pk_fma_f32  v[12:13], v[18:19], v[22:23]  (one + two)",
	"v_pk_fma_f32  v[12:13], v[12:13], v[18:19]  (a register twice)", "v_add_f32     v12, v16 (same bank)", "v_add_f32     v12, v17 (different banks)"};
template <int MODE> __global__ __launch_bounds__(1024) void k(unsigned long long *stamps, float *out, int iters)
{
	unsigned long long t0, t1;
	asm volatile("v_mov_b32 v12, 1.0\nv_mov_b32 v13, 1.0\nv_mov_b32 v16, 0.5\nv_mov_b32 v17, 0.5\nv_mov_b32 v18, 0.25\nv_mov_b32 v19, 0.25\nv_mov_b32 v20, 2.0\nv_mov_b32 v21, 2.0\nv_mov_b32 v22, 4.0\nv_mov_b32 v23, 4.0\n" ::: CLOB);
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
			if (MODE == ADD_SAME) asm volatile(G8("v_pk_add_f32", "v[12:13]", "v[16:17]", "v[16:17]", "v[20:21]") ::: CLOB);
			if (MODE == ADD_DIFF) asm volatile(G8("v_pk_add_f32", "v[12:13]", "v[18:19]", "v[16:17]", "v[22:23]") ::: CLOB);
			if (MODE == ADD_SAMEREG) asm volatile(G8("v_pk_add_f32", "v[12:13]", "v[12:13]", "v[18:19]", "v[18:19]") ::: CLOB);
			if (MODE == MUL_SAME) asm volatile(G8("v_pk_mul_f32", "v[12:13]", "v[16:17]", "v[16:17]", "v[20:21]") ::: CLOB);
			if (MODE == MUL_DIFF) asm volatile(G8("v_pk_mul_f32", "v[12:13]", "v[18:19]", "v[16:17]", "v[22:23]") ::: CLOB);
			if (MODE == FMA_000) asm volatile(G8F("v[12:13]", "v[16:17]", "v[20:21]", "v[16:17]", "v[20:21]", "v[12:13]") ::: CLOB);
			if (MODE == FMA_001) asm volatile(G8F("v[12:13]", "v[16:17]", "v[22:23]", "v[16:17]", "v[20:21]", "v[18:19]") ::: CLOB);
			if (MODE == FMA_011_DIFFSRC) asm volatile(G8F("v[12:13]", "v[18:19]", "v[22:23]", "v[16:17]", "v[22:23]", "v[18:19]") ::: CLOB);
			if (MODE == FMA_SAME2) asm volatile(G8F("v[12:13]", "v[12:13]", "v[18:19]", "v[16:17]", "v[16:17]", "v[22:23]") ::: CLOB);
			if (MODE == ADD32_SAME) asm volatile("v_add_f32 v40, v12, v16\nv_add_f32 v41, v16, v20\nv_add_f32 v42, v12, v16\nv_add_f32 v43, v16, v20\nv_add_f32 v44, v12, v16\nv_add_f32 v45, v16, v20\nv_add_f32 v46, v12, v16\nv_add_f32 v47, v16, v20\n" ::: CLOB);
			if (MODE == ADD32_DIFF) asm volatile("v_add_f32 v40, v12, v17\nv_add_f32 v41, v16, v21\nv_add_f32 v42, v12, v17\nv_add_f32 v43, v16, v21\nv_add_f32 v44, v12, v17\nv_add_f32 v45, v16, v21\nv_add_f32 v46, v12, v17\nv_add_f32 v47, v16, v21\n" ::: CLOB);
		}
	}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
	float s;
	asm volatile("v_add_f32 %0, v40, v47" : "=v"(s) :: CLOB);
	out[blockIdx.x * 1024 + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> void run()
{
	const int iters = 2000, blocks = 256;
	unsigned long long *st; float *o;
	(void)hipMalloc(&st, 8 * blocks * 16); (void)hipMalloc(&o, blocks * 1024 * 4);
	k<MODE><<<blocks, 1024>>>(st, o, 50);
	k<MODE><<<blocks, 1024>>>(st, o, iters);
	(void)hipDeviceSynchronize();
	std::vector<unsigned long long> h(blocks * 16);
	(void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
	std::sort(h.begin(), h.end());
	// 16 waves per workgroup = 4 per SIMD; each wave issues iters * 32 instructions
	printf("%-72s %5.2f cycles per instruction per SIMD (4 waves per SIMD)\n", nm[MODE], (double)h[h.size() / 2] / (iters * 32.0 * 4.0));
	(void)hipFree(st); (void)hipFree(o);
}
int main()
{
	run<ADD_SAME>(); run<ADD_DIFF>(); run<ADD_SAMEREG>(); run<MUL_SAME>(); run<MUL_DIFF>(); run<FMA_000>(); run<FMA_001>(); run<FMA_011_DIFFSRC>(); run<FMA_SAME2>();
	run<ADD32_SAME>(); run<ADD32_DIFF>();
	return 0;
}
