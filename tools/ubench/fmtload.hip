#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ v2f ed_buffer_load_format_xy(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v2f32");
__device__ v4f ed_buffer_load_format_xyzw(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");

__device__ __forceinline__ v4i make_rsrc(const void *p, uint32_t bytes, int data_format)
{
    uint64_t base = (uint64_t)p;
    v4i r;
    r.x = (int)(uint32_t)base;
    r.y = (int)((uint32_t)(base >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = (4) | (5 << 3) | (6 << 6) | (7 << 9) | (3 << 12) | (data_format << 15);
    return r;
}
__global__ void conv_xy(const int16_t *x, float *out, int n)   // n int16 values
{
    v4i r = make_rsrc(x, n * 2, 5);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i < n) { v2f v = ed_buffer_load_format_xy(r, i * 4, 0, 0); out[2 * i] = v.x; out[2 * i + 1] = v.y; }
}
__global__ void conv_xyzw(const int16_t *x, float *out, int n)
{
    v4i r = make_rsrc(x, n * 2, 12);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * i < n) { v4f v = ed_buffer_load_format_xyzw(r, i * 8, 0, 0); out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4*i+2]=v.z; out[4*i+3]=v.w; }
}
// bandwidth: each wave reads frames of 2048 B like the MFCC kernel: 8 loads of 4 B per lane; sum -> one store per wave per frame
template <int MODE>
__global__ __launch_bounds__(768) void bw(const int16_t *x, float *out, int n_frames)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 12 + wave, nw = gridDim.x * 12;
    float acc = 0.f;
    for (int f = gw; f < n_frames; f += nw)
    {
        const int16_t *fp = x + (size_t)f * 1024;
        if (MODE == 0)
        {
            const uint32_t *p = (const uint32_t *)fp;
#pragma unroll
            for (int a = 0; a < 8; a++) { uint32_t v = p[lane + 64 * a]; acc += (float)(int16_t)(v & 0xffff) + (float)(int16_t)(v >> 16); }
        }
        else if (MODE == 1)
        {
            v4i r = make_rsrc(fp, 2048, 5);
#pragma unroll
            for (int a = 0; a < 8; a++) { v2f v = ed_buffer_load_format_xy(r, (lane + 64 * a) * 4, 0, 0); acc += v.x + v.y; }
        }
        else
        {
            v4i r = make_rsrc(fp, 2048, 12);
#pragma unroll
            for (int a = 0; a < 4; a++) { v4f v = ed_buffer_load_format_xyzw(r, (lane + 64 * a) * 8, 0, 0); acc += (v.x + v.y) + (v.z + v.w); }
        }
    }
    out[(size_t)gw * 64 + lane] = acc;
}
int main()
{
    const int n = 65536;
    std::vector<int16_t> h(n);
    for (int i = 0; i < n; i++) h[i] = (int16_t)(i - 32768);
    int16_t *dx; float *dout;
    hipMalloc(&dx, n * 2); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, h.data(), n * 2, hipMemcpyHostToDevice);
    std::vector<float> o(n);
    for (int mode = 0; mode < 2; mode++)
    {
        hipMemset(dout, 0xff, n * 4);
        if (mode == 0) conv_xy<<<n / 2 / 256, 256>>>(dx, dout, n); else conv_xyzw<<<n / 4 / 256, 256>>>(dx, dout, n);
        hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; i++) if (o[i] != (float)h[i]) { if (bad < 5) printf("mode %d i %d got %g want %g\n", mode, i, o[i], (float)h[i]); bad++; }
        printf("%s: %ld wrong of %d\n", mode == 0 ? "buffer_load_format_xy 16_16 SSCALED" : "buffer_load_format_xyzw 16_16_16_16 SSCALED", bad, n);
    }
    // bandwidth
    const int nf = 65536 * 4;
    int16_t *big; hipMalloc(&big, (size_t)nf * 2048);
    hipMemset(big, 1, (size_t)nf * 2048);
    float *o2; hipMalloc(&o2, 256 * 12 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++)
    {
        for (int rep = 0; rep < 3; rep++)
        {
            hipEventRecord(e0);
            for (int it = 0; it < 10; it++)
            {
                if (mode == 0) bw<0><<<256, 768>>>(big, o2, nf);
                else if (mode == 1) bw<1><<<256, 768>>>(big, o2, nf);
                else bw<2><<<256, 768>>>(big, o2, nf);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("mode %d (%s): %.1f us per %d frames = %.2f TB/s\n", mode, mode == 0 ? "global_load_dword + cvt" : mode == 1 ? "buffer_load_format_xy" : "buffer_load_format_xyzw", ms * 100, nf, (double)nf * 2048 / (ms * 1e-4) / 1e12 * 1e-0 / 1e0 / 1e0 * 1e-0);
        }
    }
    return 0;
}
