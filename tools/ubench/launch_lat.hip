// Launch-to-answer latency of a trivial kernel on this platform, two ways of waiting:
//   (a) hipStreamSynchronize after the launch;  (b) the kernel writes a flag into host-mapped memory, the host spins on it.
// And the same for two dependent kernels (the one-frame streaming push is MFCC -> CNN).
//   hipcc --offload-arch=gfx950 -O3 launch_lat.hip -o launch_lat
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_flag(volatile unsigned *flag, unsigned v, unsigned *sink)
{
	if (threadIdx.x == 0) { sink[0] = v; __threadfence_system(); *flag = v; }
}
__global__ void k_plain(unsigned *sink, unsigned v) { if (threadIdx.x == 0) sink[0] = v; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main()
{
	unsigned *flag, *sink, *dflag;
	(void)hipHostMalloc((void **)&flag, 64, hipHostMallocMapped);
	(void)hipHostGetDevicePointer((void **)&dflag, flag, 0);
	(void)hipMalloc(&sink, 64);
	hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	const int n = 2000;
	std::vector<double> a, b, c, d;
	for (int i = 0; i < n + 100; i++)
	{
		double t0 = now();
		hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, sink, (unsigned)i);
		(void)hipStreamSynchronize(st);
		if (i >= 100) a.push_back(now() - t0);
	}
	*flag = 0;
	for (int i = 1; i <= n + 100; i++)
	{
		double t0 = now();
		hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, dflag, (unsigned)i, sink);
		while (*(volatile unsigned *)flag != (unsigned)i) {}
		if (i > 100) b.push_back(now() - t0);
	}
	(void)hipStreamSynchronize(st);
	for (int i = 0; i < n + 100; i++)
	{
		double t0 = now();
		hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, sink, (unsigned)i);
		hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, sink + 1, (unsigned)i);
		(void)hipStreamSynchronize(st);
		if (i >= 100) c.push_back(now() - t0);
	}
	*flag = 0;
	for (int i = 1; i <= n + 100; i++)
	{
		double t0 = now();
		hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, sink, (unsigned)i);
		hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, dflag, (unsigned)i, sink + 1);
		while (*(volatile unsigned *)flag != (unsigned)i) {}
		if (i > 100) d.push_back(now() - t0);
	}
	(void)hipStreamSynchronize(st);
	std::vector<double> e;
	*flag = 0;
	for (int i = 1; i <= n + 100; i++)
	{
		double t0 = now();
		hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, sink, (unsigned)i);
		hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, st, sink + 1, (unsigned)i);
		hipError_t er = hipStreamWriteValue32(st, dflag, (unsigned)i, 0);
		if (er != hipSuccess) { printf("hipStreamWriteValue32: %s\n", hipGetErrorString(er)); break; }
		while (*(volatile unsigned *)flag != (unsigned)i) {}
		if (i > 100) e.push_back(now() - t0);
	}
	(void)hipStreamSynchronize(st);
	if (!e.empty()) printf("two kernels + hipStreamWriteValue32 of the flag, spin: %.1f us\n", med(e));
	printf("one kernel:  launch + hipStreamSynchronize %.1f us   launch + spin on a host-mapped flag %.1f us\n", med(a), med(b));
	printf("two kernels: launch + hipStreamSynchronize %.1f us   launch + spin on a host-mapped flag %.1f us\n", med(c), med(d));
	return 0;
}
