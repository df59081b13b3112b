// What does an fp32 MFMA cost the vector pipe of its SIMD? (round-4 review item 4b: pass 2 of the MFCC's FFT as arithmetic on the idle matrix pipe.)
// As coexec.hip, with the float kernel's own instruction kinds: M = a chain-free stream of v_mfma_f32_16x16x4_f32 (a constant 16 x 16 real matrix
// times 16-row blocks is what a radix-8 butterfly would be), V = a stream of independent v_pk_fma_f32. 512-thread workgroup, one per CU: waves 0-3
// and waves 4-7 share the four SIMDs pairwise; also both roles in ONE wave (interleaved 1 : 8), which is how a fused kernel would issue them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int ROLE_LO, int ROLE_HI> // 0 idle, 1 MFMA, 2 VALU (packed fma), 3 both in one wave (1 MFMA : 8 packed fma)
__global__ __launch_bounds__(512) void k(unsigned long long *stamps, float *out, int iters)
{
	const int wave = threadIdx.x >> 6;
	const int role = wave < 4 ? ROLE_LO : ROLE_HI;
	float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)threadIdx.x * 1e-4f;
	v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
	v2f x0 = {a, b}, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
	const v2f m = {0.999f, 1.001f}, ad = {1e-3f, -1e-3f};
	unsigned long long t0, t1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
	if (role == 1)
		for (int it = 0; it < iters; it++)
		{
			c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
			c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
		}
	if (role == 2)
		for (int it = 0; it < iters; it++)
		{
#pragma unroll
			for (int r = 0; r < 4; r++)
				asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
				             "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(ad));
		}
	if (role == 3)
		for (int it = 0; it < iters; it++)
		{
#pragma unroll
			for (int r = 0; r < 4; r++)
			{
				asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(r == 0 ? c0 : r == 1 ? c1 : r == 2 ? c2 : c3) : "v"(a), "v"(b));
				asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
				             "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(ad));
			}
		}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
	float s = x0.x + x1.y + x2.x + x3.y + x4.x + x5.y + x6.x + x7.y;
	for (int i = 0; i < 4; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
	out[blockIdx.x * 512 + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int LO, int HI> void run(const char *name)
{
	const int iters = 4000, blocks = 256;
	unsigned long long *st; float *o;
	(void)hipMalloc(&st, 8 * blocks * 8); (void)hipMalloc(&o, blocks * 512 * 4);
	k<LO, HI><<<blocks, 512>>>(st, o, 100);
	k<LO, HI><<<blocks, 512>>>(st, o, iters);
	(void)hipDeviceSynchronize();
	std::vector<unsigned long long> h(blocks * 8);
	(void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
	std::vector<double> lo, hi;
	for (int b = 0; b < blocks; b++) for (int w = 0; w < 8; w++) (w < 4 ? lo : hi).push_back((double)h[b * 8 + w]);
	std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end());
	// per iteration: 4 MFMAs (role 1), 32 packed FMAs (role 2), 4 MFMAs + 32 packed FMAs (role 3)
	printf("%-44s waves 0-3: %7.1f cycles/iteration   waves 4-7: %7.1f cycles/iteration\n", name, lo[lo.size() / 2] / iters, hi[hi.size() / 2] / iters);
	(void)hipFree(st); (void)hipFree(o);
}
int main()
{
	run<1, 0>("MFMA f32 16x16x4 alone (4 per iteration)");
	run<2, 0>("v_pk_fma_f32 alone (32 per iteration)");
	run<2, 2>("pk_fma + pk_fma on each SIMD");
	run<1, 2>("MFMA (0-3) + pk_fma (4-7)");
	run<3, 0>("4 MFMA + 32 pk_fma in ONE wave");
	run<3, 3>("the same, two such waves per SIMD");
	return 0;
}
