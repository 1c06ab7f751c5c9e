// LDS-DMA fill rate per CU: 8 waves per workgroup, one workgroup per CU, each wave keeps two "tiles" of
// 5 x 1 KiB global_load_lds_dwordx4 in flight (the sense_mix_dma pattern) and does nothing else.
// mode 0: source re-read from a per-WG 480 KiB window (L2 / Infinity-Cache resident after the first pass)
// mode 1: source streamed from a large buffer (HBM)
// mode 2/3: same two sources with plain global_load_dwordx4 into registers (no LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void dma16(const void *g, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %1, off\n s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512) void fill(const char *src, size_t window, int iters, float *out) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 40960];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
    const char *base = src + (size_t)blockIdx.x * window;
    float accum = 0.f;
    for (int it = 0; it < iters; ++it) {
        const size_t off = ((size_t)it * 40960) % window;
        if (MODE < 2) {
#pragma unroll
            for (int j = 0; j < 5; ++j)
                dma16(base + off + (wave * 5 + j) * 1024 + lane * 16,
                      __builtin_amdgcn_readfirstlane(lds0 + (it % 3) * 40960 + (wave * 5 + j) * 1024));
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            uint4 v[5];
#pragma unroll
            for (int j = 0; j < 5; ++j)
                v[j] = *reinterpret_cast<const uint4 *>(base + off + (wave * 5 + j) * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < 5; ++j) accum += __uint_as_float(v[j].x ^ v[j].w);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (accum == 12345.f || smem[threadIdx.x] == 77) out[0] = accum;
}

template <int MODE>
void run(const char *name, const char *d, size_t window, float *out) {
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(fill<MODE>, dim3(256), dim3(512), 0, 0, d, window, 50, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(fill<MODE>, dim3(256), dim3(512), 0, 0, d, window, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = 256.0 * iters * 40960;
    printf("%-58s %7.3f ms  %7.2f TB/s chip  %6.1f GB/s per CU  %6.2f us per 40-KiB tile\n", name, ms, bytes / ms / 1e9,
           bytes / ms / 1e6 / 256, ms * 1e3 / iters);
}

int main() {
    char *d; float *out;
    const size_t big = (size_t)256 * 40960 * 2000;   // 21 GB
    hipMalloc(&d, big); hipMalloc(&out, 4);
    hipMemset(d, 1, big);
    run<0>("LDS-DMA, 480 KiB window per WG (123 MB total: L2/MALL)", d, 491520, out);
    run<0>("LDS-DMA, 40 KiB window per WG (10 MB total: L2)", d, 40960, out);
    run<1>("LDS-DMA, streamed from HBM", d, (size_t)40960 * 2000, out);
    run<2>("global_load_dwordx4 -> VGPR, 480 KiB window per WG", d, 491520, out);
    run<2>("global_load_dwordx4 -> VGPR, 40 KiB window per WG", d, 40960, out);
    run<3>("global_load_dwordx4 -> VGPR, streamed from HBM", d, (size_t)40960 * 2000, out);
    return 0;
}
