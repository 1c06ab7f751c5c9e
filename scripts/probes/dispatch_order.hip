// Where and when does block L of a 1-D grid run?  Each workgroup records its XCC id, HW_ID (SE / CU) and start /
// end time; the host checks the two assumptions the kernels' work orders rest on:
//   (1) block L is placed on XCD L % 8;   (2) within an XCD, blocks start in increasing L as slots free up.
// LDS per workgroup is a parameter so that the CU holds exactly `1` workgroup (as sense_mix_dma does).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <cstdlib>

struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };

__global__ __launch_bounds__(512) void probe(Rec *out, int units_base, int period) {
    extern __shared__ char lds[];
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned long long t0 = (unsigned long long)wall_clock64();
    // job length: (3 - (blockIdx / 8) % 4 + 1) units, like the 4 causal query tiles of a group (heaviest first)
    const int units = units_base * (period - (int)((blockIdx.x >> 3) % period));
    float x = threadIdx.x;
    for (int i = 0; i < units * 2000; ++i) x = x * 1.0001f + 0.5f;
    if (x == 1.2345f) lds[threadIdx.x] = 1;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{xcc, hwid, t0, (unsigned long long)wall_clock64()};
}

int main(int argc, char **argv) {
    // argv: lds_kib threads grid period   (defaults: the sense_mix_dma configuration)
    const int lds_kib = argc > 1 ? atoi(argv[1]) : 120, threads = argc > 2 ? atoi(argv[2]) : 512;
    const int grid = argc > 3 ? atoi(argv[3]) : 768, period = argc > 4 ? atoi(argv[4]) : 4;
    const int units_base = argc > 5 ? atoi(argv[5]) : 4;
    printf("== LDS %d KiB, %d threads, grid %d, job lengths %d..1 units\n", lds_kib, threads, grid, period);
    Rec *d; hipMalloc(&d, grid * sizeof(Rec));
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(threads), lds_kib * 1024, 0, d, units_base, period);
    hipDeviceSynchronize();
    std::vector<Rec> r(grid);
    hipMemcpy(r.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto &x : r) { tmin = std::min(tmin, x.t0); tmax = std::max(tmax, x.t1); }
    int xcc_ok = 0;
    for (int i = 0; i < grid; ++i) xcc_ok += ((r[i].xcc & 0xf) == (unsigned)(i & 7));
    printf("blocks on XCD (L %% 8): %d of %d;  kernel span %.1f us (wall_clock64 ticks = 10 ns)\n", xcc_ok, grid, (tmax - tmin) * 0.01);
    // per XCD 0: start order vs block index
    printf("XCD 0: block(slot) start_us end_us cu   -- in start order\n");
    std::vector<int> idx;
    for (int i = 0; i < grid; i += 8) idx.push_back(i);
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return r[a].t0 < r[b].t0; });
    int inversions = 0;
    for (size_t i = 1; i < idx.size(); ++i) inversions += idx[i] < idx[i - 1];
    for (size_t i = 0; i < idx.size(); ++i)
        if (getenv("VERBOSE") && (i < 40 || i + 6 > idx.size()))
            printf("  %4d(%3d) %8.1f %8.1f  se%u cu%u\n", idx[i], idx[i] >> 3, (r[idx[i]].t0 - tmin) * 0.01,
                   (r[idx[i]].t1 - tmin) * 0.01, (r[idx[i]].hwid >> 13) & 7, (r[idx[i]].hwid >> 8) & 15);
    printf("start-order inversions on XCD 0: %d of %zu\n", inversions, idx.size());
    // how many blocks of XCD 0 start while the first-round blocks are still running?
    {
        unsigned long long first_end = 0;
        std::vector<unsigned long long> ends;
        for (int i = 0; i < grid; i += 8) if ((r[i].t0 - tmin) * 0.01 < 1.0) first_end = std::max(first_end, r[i].t1);
        int early = 0, late = 0;
        for (int i = 0; i < grid; i += 8) {
            if ((r[i].t0 - tmin) * 0.01 < 1.0) continue;
            (r[i].t0 + 100 < first_end ? early : late)++;
        }
        printf("XCD 0: blocks started before the slowest first-round block ended: %d, after: %d\n", early, late);
    }
    // concurrently running blocks per XCD at t = 1 us after start
    for (int x = 0; x < 8; ++x) {
        int live = 0;
        for (int i = x; i < grid; i += 8) live += (r[i].t0 - tmin) * 0.01 < 1.0;
        printf("XCD %d: %d blocks started within the first microsecond\n", x, live);
    }
    return 0;
}
