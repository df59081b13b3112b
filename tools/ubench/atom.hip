// Cost of a returning atomic add on ONE address per XCD / per chip, by memory scope (gfx950). hipcc --offload-arch=gfx950 -O3 atom.hip -o atom
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
template <int MODE> __global__ __launch_bounds__(768) void k(unsigned *ctr, unsigned *out, unsigned *xcc_of_block, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = x;
    unsigned acc = 0;
    if (lane == 0)
        for (int i = 0; i < per_wave; i++)
        {
            unsigned v;
            if (MODE == 0) v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1) v = __hip_atomic_fetch_add(ctr + 64 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = __hip_atomic_fetch_add(ctr + 64 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            acc += v;
        }
    if (lane == 0) out[blockIdx.x * 12 + (threadIdx.x >> 6)] = acc;
}
int main()
{
    unsigned *ctr, *out, *xb; hipMalloc(&ctr, 64 * 4 * 16); hipMalloc(&out, 256 * 12 * 4); hipMalloc(&xb, 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int per_wave = 11;
    for (int mode = 0; mode < 3; mode++)
    {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++)
        {
            hipMemset(ctr, 0, 64 * 4 * 16);
            hipEventRecord(e0);
            if (mode == 0) k<0><<<256, 768>>>(ctr, out, xb, per_wave);
            else if (mode == 1) k<1><<<256, 768>>>(ctr, out, xb, per_wave);
            else k<2><<<256, 768>>>(ctr, out, xb, per_wave);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        std::vector<unsigned> h(64 * 16), hx(256);
        hipMemcpy(h.data(), ctr, 64 * 16 * 4, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xb, 1024, hipMemcpyDeviceToHost);
        unsigned total = 0; for (int x = 0; x < 16; x++) total += h[64 * x];
        int per_x[16] = {0}; for (int b = 0; b < 256; b++) per_x[hx[b] & 15]++;
        printf("mode %d (%s): %.1f us for %d atomics (%.2f ns each), counter total %u (expect %d); blocks per XCC:", mode,
               mode == 0 ? "one address, agent scope" : mode == 1 ? "one address per XCC, agent scope" : "one address per XCC, workgroup scope",
               best * 1e3, 256 * 12 * per_wave, best * 1e6 / (256 * 12 * per_wave), total, 256 * 12 * per_wave);
        for (int x = 0; x < 8; x++) printf(" %d", per_x[x]);
        printf("; blockIdx%%8 == xcc for all: %s\n", [&]{ for (int b = 0; b < 256; b++) if ((hx[b] & 15) != (unsigned)(b % 8)) return "no"; return "yes"; }());
    }
    return 0;
}
