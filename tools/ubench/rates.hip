// VALU / cross-lane / LDS-crossbar issue rates on gfx950 in SHADER CYCLES (s_memtime), with the clock the chip
// actually held (s_memtime / s_memrealtime), at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 rates.hip -o rates
// Every mode is inline asm on 8 independent register chains per wave, so nothing is folded, fused or SLP-packed.
// Output columns: cycles per wave-instruction per SIMD  (= wave cycles / (instructions per wave * waves per SIMD)).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define A3(op) #op " %0, %0, %8\n" #op " %1, %1, %8\n" #op " %2, %2, %8\n" #op " %3, %3, %8\n" #op " %4, %4, %8\n" #op " %5, %5, %8\n" #op " %6, %6, %8\n" #op " %7, %7, %8\n"
#define A4(op) #op " %0, %0, %8, %8\n" #op " %1, %1, %8, %8\n" #op " %2, %2, %8, %8\n" #op " %3, %3, %8, %8\n" #op " %4, %4, %8, %8\n" #op " %5, %5, %8, %8\n" #op " %6, %6, %8, %8\n" #op " %7, %7, %8, %8\n"
#define A2(op, suf) #op " %0, %0 " suf "\n" #op " %1, %1 " suf "\n" #op " %2, %2 " suf "\n" #op " %3, %3 " suf "\n" #op " %4, %4 " suf "\n" #op " %5, %5 " suf "\n" #op " %6, %6 " suf "\n" #op " %7, %7 " suf "\n"
#define ASW(op) #op " %0, %1\n" #op " %2, %3\n" #op " %4, %5\n" #op " %6, %7\n" #op " %0, %2\n" #op " %1, %3\n" #op " %4, %6\n" #op " %5, %7\n"
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

enum { FMA, PKFMA, PKADD, PKMUL, ADD, MOV, CVTSDWA, DPPSHR, DPPQUAD, ADDDPP, SWAP32, SWAP16, CNDMASK, SQRT, LOG, BPERM, MIX_PK_S, MIX_PK_DPP, FMA_DEP, FMA_SALU, PKFMA_SALU, CND64, ADD64, FMAC32, MUL32, MIX_SQRT_PK, MIX_SWAP_PK, MIX_SQRT_ADD, MIX_PK_ADD32, MIX_LDSR_PK, LDSR128, NMODES };
static const char *names[NMODES] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_add_f32", "v_mov_b32", "v_cvt_f32_i32 sdwa", "v_mov_b32_dpp row_shr:8", "v_mov_b32_dpp quad_perm", "v_add_f32_dpp quad_perm", "v_permlane32_swap", "v_permlane16_swap", "v_cndmask_b32", "v_sqrt_f32", "v_log_f32", "ds_bpermute_b32", "4 pk_fma + 4 fma", "4 pk_fma + 4 mov_dpp", "v_fma_f32 one dependent chain", "8 fma + 8 s_add_u32", "8 pk_fma + 8 s_add_u32", "v_cndmask_b32_e64 (sgpr pair)", "v_add_f32_e64 (8-byte encoding)", "v_fmac_f32_e32 (4-byte)", "v_mul_f32_e32 (4-byte)", "4 v_sqrt + 4 pk_fma", "4 swap32 + 4 pk_fma", "4 v_sqrt + 4 v_add_f32", "4 pk_fma + 4 v_add_f32", "4 ds_read_b128 + 4 pk_fma", "8 ds_read_b128"};

template <int MODE> __global__ __launch_bounds__(256) void k(unsigned long long *stamps, float *out, int iters, float s)
{
	extern __shared__ float dyn[];
	if (s == 12345.0f) dyn[threadIdx.x] = s; /* never true: keeps the dynamic LDS allocation */
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	f2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
	f2 sv = {s, s};
	int addr = ((threadIdx.x * 7) & 63) << 2;
	int ldsaddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
	unsigned sacc = 0;
	unsigned long long msk = 0x5555555555555555ull ^ (unsigned long long)iters;
	unsigned long long t0, r0, t1, r1;
	asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
			if (MODE == FMA) asm volatile(A4(v_fma_f32) : OPS : "v"(s));
			if (MODE == ADD) asm volatile(A3(v_add_f32) : OPS : "v"(s));
			if (MODE == MOV) asm volatile(A2(v_mov_b32, "") : OPS : "v"(s));
			if (MODE == CVTSDWA) asm volatile(A2(v_cvt_f32_i32_sdwa, "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") : OPS : "v"(s));
			if (MODE == DPPSHR) asm volatile(A2(v_mov_b32_dpp, "row_shr:8 row_mask:0xf bank_mask:0xc") : OPS : "v"(s));
			if (MODE == DPPQUAD) asm volatile(A2(v_mov_b32_dpp, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") : OPS : "v"(s));
			if (MODE == ADDDPP) asm volatile(A2(v_add_f32_dpp, ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") : OPS : "v"(s));
			if (MODE == SWAP32) asm volatile(ASW(v_permlane32_swap_b32) : OPS : "v"(s));
			if (MODE == SWAP16) asm volatile(ASW(v_permlane16_swap_b32) : OPS : "v"(s));
			if (MODE == CNDMASK) asm volatile(A3(v_cndmask_b32) : OPS : "v"(s) : "vcc");
			if (MODE == CND64) asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\nv_cndmask_b32_e64 %1, %1, %8, %9\nv_cndmask_b32_e64 %2, %2, %8, %9\nv_cndmask_b32_e64 %3, %3, %8, %9\nv_cndmask_b32_e64 %4, %4, %8, %9\nv_cndmask_b32_e64 %5, %5, %8, %9\nv_cndmask_b32_e64 %6, %6, %8, %9\nv_cndmask_b32_e64 %7, %7, %8, %9\n" : OPS : "v"(s), "s"(msk));
			if (MODE == ADD64) asm volatile(A3(v_add_f32_e64) : OPS : "v"(s));
			if (MODE == FMAC32) asm volatile(A3(v_fmac_f32_e32) : OPS : "v"(s));
			if (MODE == MUL32) asm volatile(A3(v_mul_f32_e32) : OPS : "v"(s));
			if (MODE == SQRT) asm volatile(A2(v_sqrt_f32, "") : OPS : "v"(s));
			if (MODE == LOG) asm volatile(A2(v_log_f32, "") : OPS : "v"(s));
			if (MODE == BPERM) asm volatile(A3(ds_bpermute_b32) "s_waitcnt lgkmcnt(0)\n" : OPS : "v"(addr));
			if (MODE == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\n" : OPS : "v"(s));
			if (MODE == FMA_SALU)
				asm volatile("v_fma_f32 %0, %0, %8, %8\ns_add_u32 %9, %9, 1\nv_fma_f32 %1, %1, %8, %8\ns_add_u32 %9, %9, 1\nv_fma_f32 %2, %2, %8, %8\ns_add_u32 %9, %9, 1\nv_fma_f32 %3, %3, %8, %8\ns_add_u32 %9, %9, 1\nv_fma_f32 %4, %4, %8, %8\ns_add_u32 %9, %9, 1\nv_fma_f32 %5, %5, %8, %8\ns_add_u32 %9, %9, 1\nv_fma_f32 %6, %6, %8, %8\ns_add_u32 %9, %9, 1\nv_fma_f32 %7, %7, %8, %8\ns_add_u32 %9, %9, 1\n" : OPS : "v"(s), "s"(sacc) : "scc");
		}
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
#define PO "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
			if (MODE == PKFMA) asm volatile(A4(v_pk_fma_f32) : PO : "v"(sv));
			if (MODE == PKADD) asm volatile(A3(v_pk_add_f32) : PO : "v"(sv));
			if (MODE == PKMUL) asm volatile(A3(v_pk_mul_f32) : PO : "v"(sv));
			if (MODE == PKFMA_SALU)
				asm volatile("v_pk_fma_f32 %0, %0, %8, %8\ns_add_u32 %9, %9, 1\nv_pk_fma_f32 %1, %1, %8, %8\ns_add_u32 %9, %9, 1\nv_pk_fma_f32 %2, %2, %8, %8\ns_add_u32 %9, %9, 1\nv_pk_fma_f32 %3, %3, %8, %8\ns_add_u32 %9, %9, 1\nv_pk_fma_f32 %4, %4, %8, %8\ns_add_u32 %9, %9, 1\nv_pk_fma_f32 %5, %5, %8, %8\ns_add_u32 %9, %9, 1\nv_pk_fma_f32 %6, %6, %8, %8\ns_add_u32 %9, %9, 1\nv_pk_fma_f32 %7, %7, %8, %8\ns_add_u32 %9, %9, 1\n" : PO : "v"(sv), "s"(sacc) : "scc");
			if (MODE == MIX_PK_S)
				asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_fma_f32 %4, %4, %9, %9\nv_pk_fma_f32 %1, %1, %8, %8\nv_fma_f32 %5, %5, %9, %9\nv_pk_fma_f32 %2, %2, %8, %8\nv_fma_f32 %6, %6, %9, %9\nv_pk_fma_f32 %3, %3, %8, %8\nv_fma_f32 %7, %7, %9, %9\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sv), "v"(s));
			if (MODE == MIX_SQRT_PK)
				asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_sqrt_f32 %4, %4\nv_pk_fma_f32 %1, %1, %8, %8\nv_sqrt_f32 %5, %5\nv_pk_fma_f32 %2, %2, %8, %8\nv_sqrt_f32 %6, %6\nv_pk_fma_f32 %3, %3, %8, %8\nv_sqrt_f32 %7, %7\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sv), "v"(s));
			if (MODE == MIX_SWAP_PK)
				asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_permlane32_swap_b32 %4, %5\nv_pk_fma_f32 %1, %1, %8, %8\nv_permlane32_swap_b32 %6, %7\nv_pk_fma_f32 %2, %2, %8, %8\nv_permlane32_swap_b32 %4, %6\nv_pk_fma_f32 %3, %3, %8, %8\nv_permlane32_swap_b32 %5, %7\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sv), "v"(s));
			if (MODE == MIX_SQRT_ADD)
				asm volatile("v_add_f32 %0, %0, %8\nv_sqrt_f32 %4, %4\nv_add_f32 %1, %1, %8\nv_sqrt_f32 %5, %5\nv_add_f32 %2, %2, %8\nv_sqrt_f32 %6, %6\nv_add_f32 %3, %3, %8\nv_sqrt_f32 %7, %7\n" : OPS : "v"(s));
			if (MODE == MIX_PK_ADD32)
				asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_add_f32 %4, %4, %9\nv_pk_fma_f32 %1, %1, %8, %8\nv_add_f32 %5, %5, %9\nv_pk_fma_f32 %2, %2, %8, %8\nv_add_f32 %6, %6, %9\nv_pk_fma_f32 %3, %3, %8, %8\nv_add_f32 %7, %7, %9\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sv), "v"(s));
			if (MODE == MIX_LDSR_PK)
			{
				float4 q0, q1, q2, q3;
				asm volatile("ds_read_b128 %4, %9\nv_pk_fma_f32 %0, %0, %8, %8\nds_read_b128 %5, %9 offset:1024\nv_pk_fma_f32 %1, %1, %8, %8\nds_read_b128 %6, %9 offset:2048\nv_pk_fma_f32 %2, %2, %8, %8\nds_read_b128 %7, %9 offset:3072\nv_pk_fma_f32 %3, %3, %8, %8\ns_waitcnt lgkmcnt(0)\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(sv), "v"(ldsaddr));
				a4 += q0.x + q1.y + q2.z + q3.w;
			}
			if (MODE == LDSR128)
			{
				float4 q0, q1, q2, q3, q4, q5, q6, q7;
				asm volatile("ds_read_b128 %0, %8\nds_read_b128 %1, %8 offset:1024\nds_read_b128 %2, %8 offset:2048\nds_read_b128 %3, %8 offset:3072\nds_read_b128 %4, %8 offset:4096\nds_read_b128 %5, %8 offset:5120\nds_read_b128 %6, %8 offset:6144\nds_read_b128 %7, %8 offset:7168\ns_waitcnt lgkmcnt(0)\n" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(ldsaddr));
				a4 += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w;
			}
			if (MODE == MIX_PK_DPP)
				asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_mov_b32_dpp %4, %4 row_shr:8 row_mask:0xf bank_mask:0xc\nv_pk_fma_f32 %1, %1, %8, %8\nv_mov_b32_dpp %5, %5 row_shr:8 row_mask:0xf bank_mask:0xc\nv_pk_fma_f32 %2, %2, %8, %8\nv_mov_b32_dpp %6, %6 row_shr:8 row_mask:0xf bank_mask:0xc\nv_pk_fma_f32 %3, %3, %8, %8\nv_mov_b32_dpp %7, %7 row_shr:8 row_mask:0xf bank_mask:0xc\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sv), "v"(s));
		}
	}
	asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
	f2 ps = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + ps.x + ps.y + (float)sacc;
	if ((threadIdx.x & 63) == 0)
	{
		const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
		stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
	}
}
template <int MODE> void run(int wps, int n_instr_per_group /* VALU (or DS) instructions per 8-group */)
{
	const int iters = 3000, blocks = 256 * wps;
	const size_t lds = ((160 * 1024 / wps) & ~(size_t)1023) - 2048; /* exactly wps workgroups fit a CU: even placement */
	(void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	unsigned long long *st; float *d;
	hipMalloc(&st, sizeof(unsigned long long) * 2 * blocks * 4); hipMalloc(&d, sizeof(float) * blocks * 256);
	k<MODE><<<blocks, 256, lds>>>(st, d, 300, 1.0001f);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<MODE><<<blocks, 256, lds>>>(st, d, iters, 1.0001f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> h(2 * blocks * 4);
	hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
	std::vector<double> cyc, clk;
	for (int w = 0; w < blocks * 4; w++) { cyc.push_back((double)h[2 * w]); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1); }
	std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
	const double instr = (double)iters * 4 * n_instr_per_group;
	const double ck = clk[clk.size() / 2];
	printf("%-30s waves/SIMD %d: %8.3f ms  wave cycles median %9.0f max %9.0f  clock %.2f GHz  => cycles per instruction per SIMD: %.2f (stamps) %.2f (wall)\n", names[MODE], wps, ms,
	       cyc[cyc.size() / 2], cyc.back(), ck, cyc[cyc.size() / 2] / (instr * wps), ms * 1e6 * ck / (instr * wps));
	hipFree(st); hipFree(d);
}
int main()
{
	for (int w : {1, 2, 4})
	{
		run<FMA>(w, 8); run<PKFMA>(w, 8); run<PKADD>(w, 8); run<PKMUL>(w, 8); run<ADD>(w, 8); run<MIX_PK_S>(w, 8); run<MIX_PK_DPP>(w, 8);
		run<FMA_DEP>(w, 8); run<FMA_SALU>(w, 8); run<PKFMA_SALU>(w, 8); run<ADD64>(w, 8); run<FMAC32>(w, 8); run<MUL32>(w, 8); run<MIX_SQRT_PK>(w, 8); run<MIX_SWAP_PK>(w, 8); run<MIX_SQRT_ADD>(w, 8); run<MIX_PK_ADD32>(w, 8); run<MIX_LDSR_PK>(w, 8); run<LDSR128>(w, 8);
	}
	for (int w : {2, 4})
	{
		run<MOV>(w, 8); run<CVTSDWA>(w, 8); run<DPPSHR>(w, 8); run<DPPQUAD>(w, 8); run<ADDDPP>(w, 8); run<SWAP32>(w, 8); run<SWAP16>(w, 8);
		run<CNDMASK>(w, 8); run<CND64>(w, 8); run<SQRT>(w, 8); run<LOG>(w, 8); run<BPERM>(w, 8);
	}
	return 0;
}
