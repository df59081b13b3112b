// Do VALU issue and LDS traffic overlap on gfx950? Each wave runs, per iteration, NV independent v_fma and NL LDS
// ops (wave-private addresses, conflict-free). Compare time(NV,NL) with time(NV,0) and time(0,NL).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NV, int NLW, int NLR>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
	__shared__ float2 lds[256 * 8];
	float a[8];
	for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
	float2 v = make_float2(a[0], a[1]);
	float2 acc = make_float2(0, 0);
	float2 *mine = lds + (threadIdx.x >> 6) * 512 + (threadIdx.x & 63);
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int i = 0; i < NV; i++) a[i & 7] = __builtin_fmaf(a[i & 7], 1.0001f, 0.5f);
#pragma unroll
		for (int i = 0; i < NLW; i++) mine[64 * (i & 7)] = v;
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int i = 0; i < NLR; i++) { float2 t = mine[64 * (i & 7)]; acc.x += t.x; acc.y += t.y; }
		__builtin_amdgcn_wave_barrier();
	}
	float r = acc.x + acc.y; for (int i = 0; i < 8; i++) r += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NV, int NLW, int NLR> void run(int wps)
{
	float *d; hipMalloc(&d, 256 * 8192 * 4);
	const int iters = 2000, blocks = 256 * wps;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<NV, NLW, NLR><<<blocks, 256>>>(d, 10);
	hipEventRecord(e0);
	k<NV, NLW, NLR><<<blocks, 256>>>(d, iters);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	// per CU per iteration: time in ns
	double ns_per_iter_per_wave = ms * 1e6 / iters;              // all waves run concurrently (one round)
	printf("NV=%3d NLW=%2d NLR=%2d waves/SIMD=%d: %.3f ms  -> %.1f ns per iteration (all %d waves of a CU in parallel)\n", NV, NLW, NLR, wps, ms, ns_per_iter_per_wave, 4 * wps);
	hipFree(d);
}
int main()
{
	for (int w : {1, 4})
	{
		run<64, 0, 0>(w); run<0, 8, 8>(w); run<64, 8, 8>(w);
		run<0, 16, 0>(w); run<64, 16, 0>(w); run<0, 0, 16>(w); run<64, 0, 16>(w);
		run<128, 8, 8>(w);
	}
	return 0;
}
