// Micro-benchmark: VALU issue rate on gfx950 for v_fma_f32, v_pk_fma_f32, v_add_f32, v_pk_add_f32 (hipcc valu.hip -o valu)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float s)
{
	float a[8]; v2f p[8];
	for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = v2f{a[i], a[i] + 1.0f}; }
	const v2f sv = {s, s * 0.5f};
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int r = 0; r < 4; r++)
#pragma unroll
			for (int i = 0; i < 8; i++)
			{
				if (MODE == 0) a[i] = __builtin_fmaf(a[i], s, 0.5f);
				else if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], sv, sv);
				else if (MODE == 2) a[i] = a[i] + s;
				else p[i] = p[i] + sv;
			}
	}
	float r = 0; for (int i = 0; i < 8; i++) r += a[i] + p[i].x + p[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, int wavesPerSimd)
{
	float *d; hipMalloc(&d, 256 * 4096 * 4);
	const int iters = 4000, blocks = 256 * wavesPerSimd; // 256 threads = 4 waves = 1 per SIMD
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<MODE><<<blocks, 256>>>(d, 10, 1.0001f);
	hipEventRecord(e0);
	k<MODE><<<blocks, 256>>>(d, iters, 1.0001f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double instr_per_wave = (double)iters * 32;
	double wave_instr = instr_per_wave * blocks * 4;
	printf("%-12s waves/SIMD=%d: %.3f ms, %.2f wave-instr/ns chip, per SIMD %.3f instr/ns (=> %.2f cycles/instr @2.4GHz)\n", name, wavesPerSimd, ms,
	       wave_instr / ms / 1e6, wave_instr / ms / 1e6 / 1024, 2.4 / (wave_instr / ms / 1e6 / 1024));
	hipFree(d);
}
int main()
{
	for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_add_f32", w); run<3>("v_pk_add_f32", w); }
	return 0;
}
