// How fast can one wavefront-per-2KB-frame streaming read go on gfx950, by load shape and waves per CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB) void k(const uint32_t *in, uint32_t *out, uint32_t n_frames, int extra)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t stride = gridDim.x * WPB;
	uint32_t acc = 0;
	uint32_t f = blockIdx.x * WPB + wave;
	if (MODE == 0)
	{
		uint32_t v[8];
		if (f < n_frames) for (int a = 0; a < 8; a++) v[a] = in[(size_t)f * 512 + lane + 64 * a];
		for (; f < n_frames; f += stride)
		{
			uint32_t c[8];
			for (int a = 0; a < 8; a++) c[a] = v[a];
			if (f + stride < n_frames) for (int a = 0; a < 8; a++) v[a] = in[(size_t)(f + stride) * 512 + lane + 64 * a];
			for (int a = 0; a < 8; a++) acc += c[a];
			for (int e = 0; e < extra; e++) acc = acc * 1664525u + 1013904223u; // dependent VALU filler
		}
	}
	else
	{
		uint4 v[2];
		const uint4 *in4 = reinterpret_cast<const uint4 *>(in);
		if (f < n_frames) for (int a = 0; a < 2; a++) v[a] = in4[(size_t)f * 128 + lane + 64 * a];
		for (; f < n_frames; f += stride)
		{
			uint4 c[2];
			for (int a = 0; a < 2; a++) c[a] = v[a];
			if (f + stride < n_frames) for (int a = 0; a < 2; a++) v[a] = in4[(size_t)(f + stride) * 128 + lane + 64 * a];
			for (int a = 0; a < 2; a++) acc += c[a].x + c[a].y + c[a].z + c[a].w;
			for (int e = 0; e < extra; e++) acc = acc * 1664525u + 1013904223u;
		}
	}
	if (acc == 0x12345) out[0] = acc;
}
template <int MODE, int WPB> void run(const uint32_t *d, uint32_t *o, uint32_t n, int blocks_per_cu, int extra)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<MODE, WPB><<<256 * blocks_per_cu, 64 * WPB>>>(d, o, n, extra);
	hipEventRecord(e0);
	for (int i = 0; i < 5; i++) k<MODE, WPB><<<256 * blocks_per_cu, 64 * WPB>>>(d, o, n, extra);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
	printf("mode %s waves/CU %2d extra %4d: %.3f ms  %.0f GB/s  %.0f Mframes/s\n", MODE ? "dwordx4x2" : "dword x8 ", WPB * blocks_per_cu, extra, ms,
	       n * 2048.0 / ms / 1e6, n / ms / 1e3);
}
int main()
{
	const uint32_t n = 1048576; uint32_t *d, *o;
	hipMalloc(&d, (size_t)n * 2048); hipMalloc(&o, 64); hipMemset(d, 1, (size_t)n * 2048);
	for (int extra : {0, 400, 1500})
	{
		run<0, 8>(d, o, n, 1, extra); run<0, 8>(d, o, n, 2, extra); run<0, 8>(d, o, n, 4, extra);
		run<1, 8>(d, o, n, 1, extra); run<1, 8>(d, o, n, 2, extra); run<1, 8>(d, o, n, 4, extra);
	}
	return 0;
}
