// Do the matrix pipe and the VALU of one SIMD overlap when they belong to DIFFERENT waves (two waves per SIMD)?
// 512-thread workgroup, one per CU: waves 0-3 and waves 4-7 share the four SIMDs pairwise. Roles: M = a chain-free
// stream of v_mfma_i32_32x32x32_i8, V = a stream of independent v_med3_i32 / v_ashrrev_i32 (the CNN epilogue's mix).
// Prints shader cycles per role mix: MM, VV, MV, and M / V alone (the partner wave idles).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int ROLE_LO, int ROLE_HI> // 0 idle, 1 MFMA, 2 VALU
__global__ __launch_bounds__(512) void k(unsigned long long *stamps, int *out, int iters)
{
	const int wave = threadIdx.x >> 6;
	const int role = wave < 4 ? ROLE_LO : ROLE_HI;
	v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
	v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
	int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	unsigned long long t0, t1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
	if (role == 1)
		for (int it = 0; it < iters; it++)
		{
			c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
			c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
		}
	if (role == 2)
		for (int it = 0; it < iters; it++)
		{
#pragma unroll
			for (int r = 0; r < 4; r++)
				asm volatile("v_ashrrev_i32 %0, 1, %0\nv_med3_i32 %1, %1, 0, %8\nv_ashrrev_i32 %2, 1, %2\nv_med3_i32 %3, %3, 0, %8\nv_ashrrev_i32 %4, 1, %4\nv_med3_i32 %5, %5, 0, %8\nv_ashrrev_i32 %6, 1, %6\nv_med3_i32 %7, %7, 0, %8\n"
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(127));
		}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
	int s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
	for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
	out[blockIdx.x * 512 + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int LO, int HI> void run(const char *name)
{
	const int iters = 4000, blocks = 256;
	unsigned long long *st; int *o;
	(void)hipMalloc(&st, 8 * blocks * 8); (void)hipMalloc(&o, blocks * 512 * 4);
	k<LO, HI><<<blocks, 512>>>(st, o, 100);
	k<LO, HI><<<blocks, 512>>>(st, o, iters);
	(void)hipDeviceSynchronize();
	std::vector<unsigned long long> h(blocks * 8);
	(void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
	std::vector<double> lo, hi;
	for (int b = 0; b < blocks; b++) for (int w = 0; w < 8; w++) (w < 4 ? lo : hi).push_back((double)h[b * 8 + w]);
	std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end());
	// per iteration: 4 MFMAs (role 1) or 32 VALU instructions (role 2)
	printf("%-34s waves 0-3: %7.1f cycles/iteration   waves 4-7: %7.1f cycles/iteration\n", name, lo[lo.size() / 2] / iters, hi[hi.size() / 2] / iters);
	(void)hipFree(st); (void)hipFree(o);
}
int main()
{
	run<1, 0>("MFMA alone (4 per iteration)");
	run<2, 0>("VALU alone (32 per iteration)");
	run<1, 1>("MFMA + MFMA on each SIMD");
	run<2, 2>("VALU + VALU on each SIMD");
	run<1, 2>("MFMA (0-3) + VALU (4-7)");
	run<2, 1>("VALU (0-3) + MFMA (4-7)");
	return 0;
}
