// Does packed fp32 (v_pk_add/mul/fma_f32: two frames per wave, one in .x and one in .y) pay off for a radix-8 FFT pass?
// Same butterfly + twiddle code on T = float (one frame per lane set) and T = float2 (two frames): time per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T bc(float x);
template <> __device__ __forceinline__ float bc<float>(float x) { return x; }
template <> __device__ __forceinline__ f2 bc<f2>(float x) { return (f2)(x, x); }
template <typename T> __device__ __forceinline__ T fma_(T a, T b, T c);
template <> __device__ __forceinline__ float fma_<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ f2 fma_<f2>(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <typename T>
__device__ __forceinline__ void dft4(T y0r, T y0i, T y1r, T y1i, T y2r, T y2i, T y3r, T y3i, T &o0r, T &o0i, T &o1r, T &o1i, T &o2r, T &o2i, T &o3r, T &o3i)
{
	T a0r = y0r + y2r, a0i = y0i + y2i, a1r = y0r - y2r, a1i = y0i - y2i;
	T a2r = y1r + y3r, a2i = y1i + y3i, a3r = y1i - y3i, a3i = y3r - y1r;
	o0r = a0r + a2r; o0i = a0i + a2i; o2r = a0r - a2r; o2i = a0i - a2i;
	o1r = a1r + a3r; o1i = a1i + a3i; o3r = a1r - a3r; o3i = a1i - a3i;
}
template <typename T>
__device__ __forceinline__ void radix8(T (&r)[8], T (&i)[8])
{
	const T h = bc<T>(0.70710678118654752440f), nh = bc<T>(-0.70710678118654752440f);
	T ur[4], ui[4], vr[4], vi[4];
#pragma unroll
	for (int a = 0; a < 4; a++) { ur[a] = r[a] + r[a + 4]; ui[a] = i[a] + i[a + 4]; vr[a] = r[a] - r[a + 4]; vi[a] = i[a] - i[a + 4]; }
	dft4<T>(ur[0], ui[0], ur[1], ui[1], ur[2], ui[2], ur[3], ui[3], r[0], i[0], r[2], i[2], r[4], i[4], r[6], i[6]);
	const T t1r = vr[1] + vi[1], t1i = vi[1] - vr[1], t3r = vi[3] - vr[3], t3i = -(vi[3] + vr[3]);
	const T a0r = vr[0] + vi[2], a0i = vi[0] - vr[2], a1r = vr[0] - vi[2], a1i = vi[0] + vr[2];
	const T u_r = t1r + t3r, u_i = t1i + t3i, w_r = t1i - t3i, w_i = t3r - t1r;
	r[1] = fma_<T>(h, u_r, a0r); i[1] = fma_<T>(h, u_i, a0i); r[5] = fma_<T>(nh, u_r, a0r); i[5] = fma_<T>(nh, u_i, a0i);
	r[3] = fma_<T>(h, w_r, a1r); i[3] = fma_<T>(h, w_i, a1i); r[7] = fma_<T>(nh, w_r, a1r); i[7] = fma_<T>(nh, w_i, a1i);
}
template <typename T>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
	T re[8], im[8], wr[8], wi[8];
#pragma unroll
	for (int a = 0; a < 8; a++)
	{
		re[a] = bc<T>(seed * (threadIdx.x + a)); im[a] = bc<T>(seed * (a + 1));
		wr[a] = bc<T>(__cosf(0.01f * threadIdx.x * a)); wi[a] = bc<T>(__sinf(0.01f * threadIdx.x * a));
	}
	for (int it = 0; it < iters; it++)
	{
		radix8<T>(re, im);
#pragma unroll
		for (int p = 1; p < 8; p++)
		{
			T xr = re[p], xi = im[p];
			re[p] = xr * wr[p] - xi * wi[p];
			im[p] = fma_<T>(xr, wi[p], xi * wr[p]);
		}
	}
	T s = bc<T>(0.0f);
#pragma unroll
	for (int a = 0; a < 8; a++) s += re[a] + im[a];
	out[blockIdx.x * blockDim.x + threadIdx.x] = *reinterpret_cast<float *>(&s);
}
template <typename T> float run(int waves_per_simd, int iters)
{
	float *d; hipMalloc(&d, 256 * 4096 * 4);
	const int blocks = 256 * waves_per_simd; // 256 threads = 4 waves per block -> waves_per_simd blocks per CU
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<T><<<blocks, 256>>>(d, 10, 1e-3f);
	hipEventRecord(e0);
	k<T><<<blocks, 256>>>(d, iters, 1e-3f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	hipFree(d);
	return ms;
}
int main()
{
	const int iters = 20000;
	for (int w = 1; w <= 8; w *= 2)
	{
		float a = run<float>(w, iters), b = run<f2>(w, iters);
		// per SIMD: w waves, each does `iters` passes; a packed pass carries two frames
		printf("waves/SIMD %d: scalar %.3f ms (%.1f cycles/pass/wave-slot), packed %.3f ms (%.1f cycles per 2 passes) -> packed/scalar work ratio %.2f\n", w, a,
		       a * 1e-3 * 2.4e9 / iters / w, b, b * 1e-3 * 2.4e9 / iters / w, 2.0 * a / b);
	}
	return 0;
}
