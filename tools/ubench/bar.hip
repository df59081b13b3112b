// Can the host write straight into device memory (fine-grained allocation over the BAR) and how long does a tiny kernel take to read
// 2 KB from there vs from host-mapped memory? hipcc --offload-arch=gfx950 -O3 bar.hip -o bar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <vector>
#include <algorithm>
__global__ void rd(const int *src, int *dst_host, volatile unsigned *flag, unsigned seq)
{
    int v = src[threadIdx.x] + src[threadIdx.x + 256];
    dst_host[threadIdx.x] = v;
    __threadfence_system();
    if (threadIdx.x == 0) *flag = seq;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    int *dev_fg = nullptr, *host_mapped = nullptr, *out = nullptr; unsigned *flag = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&dev_fg, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    hipHostMalloc((void **)&host_mapped, 4096, hipHostMallocMapped);
    hipHostMalloc((void **)&out, 4096, hipHostMallocMapped);
    hipHostMalloc((void **)&flag, 64, hipHostMallocMapped);
    int *d_mapped, *d_out; unsigned *d_flag;
    hipHostGetDevicePointer((void **)&d_mapped, host_mapped, 0); hipHostGetDevicePointer((void **)&d_out, out, 0); hipHostGetDevicePointer((void **)&d_flag, flag, 0);
    hipPointerAttribute_t at; memset(&at, 0, sizeof(at));
    e = hipPointerGetAttributes(&at, dev_fg);
    printf("attributes: %s type %d hostPointer %p devicePointer %p\n", hipGetErrorString(e), (int)at.type, at.hostPointer, at.devicePointer);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned seq = 0;
    for (int mode = 0; mode < 2; mode++)
    {
        int *hp = mode == 0 ? host_mapped : dev_fg;     // host writes here
        int *gp = mode == 0 ? d_mapped : dev_fg;        // kernel reads here
        std::vector<double> lat;
        for (int it = 0; it < 3000; it++)
        {
            double t0 = now();
            for (int i = 0; i < 512; i++) hp[i] = it + i;   // 2 KB of "samples" written by the host
            seq++;
            hipLaunchKernelGGL(rd, dim3(1), dim3(256), 0, st, gp, d_out, d_flag, seq);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
            double t1 = now();
            if (out[5] != (it + 5) + (it + 5 + 256)) { printf("WRONG DATA mode %d it %d: %d\n", mode, it, out[5]); return 1; }
            if (it >= 200) lat.push_back(t1 - t0);
        }
        std::sort(lat.begin(), lat.end());
        printf("%s: p50 %.2f us p90 %.2f us (host writes 2 KB, launch, kernel reads it, writes 1 KB + flag to host-mapped memory)\n",
               mode == 0 ? "input in host-mapped memory (GPU reads over the bus)" : "input in fine-grained DEVICE memory (host writes over the bus)", lat[lat.size() / 2], lat[lat.size() * 9 / 10]);
    }
    return 0;
}
