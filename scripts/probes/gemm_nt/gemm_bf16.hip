// Dense layer  C[M,N] = X[M,K] * W[N,K]^T (+ bias[N]) (+ tanh-GELU)  for gfx950, bf16 / fp16 in and out, fp32
// accumulation -- the shape of every nn.Linear of the Backpack forward (M = batch*seq rows of activations, W in
// torch's (out_features, in_features) layout, so both operands are K-contiguous and no transposition is needed).
// EXPERIMENTAL (csrc/README in DESIGN.md): the model's dense layers run on hipBLASLt; this kernel exists to
// measure what the LDS-DMA + MFMA toolkit of this repo reaches on those shapes next to the library.
//
// Workgroup = 8 waves = 2 (m) x 4 (n); workgroup tile 256 x 256, wave tile 128 (m) x 64 (n) = 4 x 2 MFMA blocks
// (128 accumulator VGPRs), K in chunks of 64 through a 2-slot LDS ring (2 x 64 KiB), both tiles as K-contiguous
// row images with the XOR swizzle of bp_dma.h (conflict-free ds_read_b128 for 32 lanes on 32 rows).
// Per 16-wide K step a wave reads 2 W fragments + 4 X fragments (6 x 1 KiB) for 8 MFMAs.
// D = A B with A = W rows (n), B = X rows (m): lane = m, registers = 4-runs of consecutive n -> 8-byte stores.
#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"

namespace bp {

namespace {
constexpr int GM = 256, GN = 256, GK = 64, GROW = 128;   // GROW: bytes per tile row (64 x 2 B)
constexpr int GTILE = 256 * GROW;                         // 32 KiB per operand tile
constexpr int GSTAGE = 2 * GTILE;

BP_DEV float gelu_tanh(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  ==  x * sigmoid(2 u),  u = sqrt(2/pi) (x + 0.044715 x^3)
    const float u = 0.7978845608028654f * x * fmaf(0.044715f, x * x, 1.f);
    return x / (1.f + fast_exp2(-2.f * kLog2e * u));
}
}  // namespace

template <class ET, bool GELU>
__global__ __launch_bounds__(512) void gemm_nt_kernel(const GemmParams p) {
    using E = Elem<ET>;
    __shared__ __attribute__((aligned(16))) char smem[2 * GSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    const int n_tiles = (p.n + GN - 1) / GN, m_tiles = (p.m + GM - 1) / GM;
    int mt, nt;
    if (!xcd_map(blockIdx.x, m_tiles, n_tiles, mt, nt)) return;   // the n tiles of one m tile share an XCD's L2
    const int m0 = mt * GM, n0 = nt * GN;

    const uint16_t *xg = reinterpret_cast<const uint16_t *>(p.x);
    const uint16_t *wg = reinterpret_cast<const uint16_t *>(p.w);

    // DMA: piece = 8 rows x 128 B; 32 pieces per operand tile, 4 + 4 per wave.  Per-lane source = clamped row
    // (rows past M / N are masked at the store) x leading dimension + swizzled 16-byte slot.
    uint32_t xoff[4], woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave * 4 + j) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ k_swz<GROW>(row);
        xoff[j] = (uint32_t)((int64_t)min(m0 + row, p.m - 1) * p.ldx + slot * 8) * 2u;
        woff[j] = (uint32_t)((int64_t)min(n0 + row, p.n - 1) * p.ldw + slot * 8) * 2u;
    }
    const uint32_t lds0 = lds_base_addr(smem);
    const int nchunks = p.k / GK;
    auto issue = [&](int c) {
        const uint32_t st = lds0 + (c & 1) * GSTAGE;
        const uint16_t *xb = xg + (int64_t)c * GK, *wb = wg + (int64_t)c * GK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dma16_s(wb, woff[j], st + (wave * 4 + j) * 1024);
            dma16_s(xb, xoff[j], st + GTILE + (wave * 4 + j) * 1024);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // lane-constant read offsets: row r of a tile, K step ks: slot (2 ks + hh) ^ swz(r)
    int w_off[2], x_off[4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) w_off[ni] = (wn * 64 + ni * 32 + l31) * GROW;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) x_off[mi] = GTILE + (wm * 128 + mi * 32 + l31) * GROW;
    const int sw = k_swz<GROW>(l31);   // depends on row bits 0..3 only; all rows here are l31 + multiple of 32

    issue(0);
    for (int c = 0; c < nchunks; ++c) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (c + 1 < nchunks) issue(c + 1);
        const char *st = smem + (c & 1) * GSTAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int so = ((2 * ks + hh) ^ sw) * 16;
            u32x4 a[2], b[4];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) a[ni] = lds_read_16B(st, w_off[ni] + so);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) b[mi] = lds_read_16B(st, x_off[mi] + so);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = E::mfma(a[ni], b[mi], acc[mi][ni]);
        }
    }

    // epilogue: bias, activation, 16-bit stores (lane = row m, 4 consecutive n per store)
    uint16_t *cg = reinterpret_cast<uint16_t *>(p.c);
    const uint16_t *bias = reinterpret_cast<const uint16_t *>(p.bias);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * hh;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr && n + 3 < p.n) {
                const u32x2 bw = *reinterpret_cast<const u32x2 *>(bias + n);
                const uint32_t b0 = bw[0], b1 = bw[1];
                bv[0] = E::lo_f32(b0); bv[1] = E::hi_f32(b0); bv[2] = E::lo_f32(b1); bv[3] = E::hi_f32(b1);
            } else if (bias != nullptr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (n + i < p.n) bv[i] = E::lo_f32(bias[n + i]);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = m0 + wm * 128 + mi * 32 + l31;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[mi][ni][4 * g + i] + bv[i];
                    if (GELU) v[i] = gelu_tanh(v[i]);
                }
                if (m < p.m) {
                    uint16_t *dst = cg + (int64_t)m * p.ldc + n;
                    if (n + 3 < p.n) {
                        *reinterpret_cast<u32x2 *>(dst) = u32x2{E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (n + i < p.n) dst[i] = E::from_float(v[i]);
                    }
                }
            }
        }
    }
}

// K % 64 == 0, leading dimensions multiples of 8, 16-byte aligned bases
hipError_t launch_gemm_nt(const GemmParams &p, int dtype, hipStream_t stream) {
    const int grid = xcd_grid((p.m + GM - 1) / GM, (p.n + GN - 1) / GN);
    dim3 g(grid), t(512);
    if (dtype == 1) {
        if (p.gelu) hipLaunchKernelGGL((gemm_nt_kernel<BF16, true>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<BF16, false>), g, t, 0, stream, p);
    } else {
        if (p.gelu) hipLaunchKernelGGL((gemm_nt_kernel<F16, true>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<F16, false>), g, t, 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace bp
