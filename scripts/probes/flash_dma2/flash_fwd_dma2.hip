// Fused attention forward for gfx950, LDS-DMA ring + TWO 32-query blocks per wave.
//
// Why a second variant: the phase stamps of flash_fwd_dma.hip (BP_PROFILE_PHASES build,
// scripts/probes/flash_phases.py) show a wave spending its time in strictly alternating phases --
// S^T MFMAs, then softmax VALU, then PV MFMAs -- so the matrix pipe idles while the VALU works and vice
// versa (MFMA busy 18 %, VALU busy ~50 %, rocprof r01_b), and with ~2 resident waves per SIMD the other
// waves do not fill the gaps.  Here one wave owns 64 query rows as two independent 32-row blocks A and
// B whose instruction streams the scheduler can interleave:
//        S_A = K Q_A^T | S_B = K Q_B^T  ||  softmax(A)       (MFMA || VALU)
//        O_A += V^T P_A                  ||  softmax(B)
//        O_B += V^T P_B
// Each K / V fragment read from LDS now feeds two MFMAs (half the LDS reads per MFMA).
// Everything else (tile algebra, LDS image, DMA ring, masks) is flash_fwd_dma.hip's.
#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"

namespace bp {

template <int KD, int NV, int NWAVE>
struct Flash2Cfg {
    static constexpr int QW = 64;                 // queries per wave (2 x 32)
    static constexpr int BM = NWAVE * QW, BN = 64, NT = NWAVE * 64, NSTAGE = 2;
    static constexpr int KROW = KD <= 4 ? 128 : 256;
    static constexpr int KSLOTS = KROW / 16;
    static constexpr int VROW = NV * 64;
    static constexpr int VCH = NV * 4;
    static constexpr int KTILE = BN * KROW;
    static constexpr int VTILE = BN * VROW;
    static constexpr int STAGE = KTILE + VTILE;
    static constexpr int K_DMA = KTILE / 1024 / NWAVE;
    static constexpr int V_DMA = VTILE / 1024 / NWAVE;
    static constexpr int DMA_PER_STAGE = K_DMA + V_DMA;
    static constexpr int K_ROWS_PER_DMA = 1024 / KROW;
};

template <class ET, int KD, int NV, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void flash_fwd_dma2_kernel(const FlashParams p) {
    using C = Flash2Cfg<KD, NV, NWAVE>;
    using E = Elem<ET>;
    __shared__ __attribute__((aligned(16))) char smem[C::NSTAGE * C::STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    const int n_qtiles = (p.max_sq + C::BM - 1) / C::BM;
    int bh, slot;
    if (!xcd_map(blockIdx.x, p.b * p.h, n_qtiles, bh, slot)) return;
    const int qt = n_qtiles - 1 - slot;
    const int batch = bh / p.h;
    const int head = bh - batch * p.h;

    int seq_q, seq_k;
    int64_t q_off, k_off, v_off, o_off;
    if (p.cu_q != nullptr) {
        const int a = p.cu_q[batch], b = p.cu_q[batch + 1];
        const int c = p.cu_k[batch], d = p.cu_k[batch + 1];
        seq_q = b - a; seq_k = d - c;
        q_off = a * p.q_rs; o_off = a * p.o_rs; k_off = c * p.k_rs; v_off = c * p.v_rs;
    } else {
        seq_q = p.max_sq; seq_k = p.max_sk;
        q_off = batch * p.q_bs; o_off = batch * p.o_bs; k_off = batch * p.k_bs; v_off = batch * p.v_bs;
    }
    if (qt * C::BM >= seq_q) return;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + q_off + (int64_t)head * p.q_hs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + k_off + (int64_t)head * p.k_hs;
    const uint16_t *vg = reinterpret_cast<const uint16_t *>(p.v) + v_off + (int64_t)head * p.v_hs;

    int k_end = seq_k;
    if (p.causal) k_end = min(seq_k, qt * C::BM + C::BM);
    const int nkb = (k_end + C::BN - 1) / C::BN;

    const int q0 = qt * C::BM + wave * C::QW;     // first query row of this wave (multiple of 64)
    const bool wave_has_rows = q0 < seq_q;
    const float c2 = p.scale_log2e;

    if (p.d * 2 != C::KROW) {   // K pad slots are never written by the DMA: zero them once
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < C::NSTAGE * C::STAGE; off += C::NT * 16) lds_write_16B(smem, off, z);
        __syncthreads();
    }

    // ---- Q fragments of both query blocks ---------------------------------------------------------
    u32x4 qf[2][KD];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const uint16_t *row = qg + (int64_t)min(q0 + 32 * qb + l31, seq_q - 1) * p.q_rs;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (col < p.d) v = ld_global_16B(row + col);
            qf[qb][s] = v;
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int s = 0; s < KD; ++s) settle(qf[qb][s]);   // see bp_common.h: no vmcnt(0) in the loop

    // ---- per-lane DMA source descriptors ------------------------------------------------------------
    int k_row[C::K_DMA], k_col[C::K_DMA];
    uint32_t k_voff[C::K_DMA];
#pragma unroll
    for (int j = 0; j < C::K_DMA; ++j) {
        const int row = (wave * C::K_DMA + j) * C::K_ROWS_PER_DMA + lane / C::KSLOTS;
        k_row[j] = row;
        k_col[j] = ((lane % C::KSLOTS) ^ k_swz<C::KROW>(row)) * 8;
        k_voff[j] = (uint32_t)(row * p.k_rs + k_col[j]) * 2u;
    }
    int v_row[C::V_DMA], v_col[C::V_DMA];
    uint32_t v_voff[C::V_DMA];
#pragma unroll
    for (int j = 0; j < C::V_DMA; ++j) {
        const int c = (wave * C::V_DMA + j) * 64 + lane;
        const int row = c / C::VCH, stored = c - row * C::VCH;
        int c64 = stored >> 2;
        if (NV == 2) c64 ^= (row >> 1) & 1;
        if (NV == 4) c64 ^= row & 3;
        v_row[j] = row;
        v_col[j] = ((c64 << 2) | (stored & 3)) * 8;
        v_voff[j] = (uint32_t)(row * p.v_rs + v_col[j]) * 2u;
    }
    const uint32_t lds0 = lds_base_addr(smem);
    auto issue = [&](int kb) {
        const uint32_t stage = lds0 + (kb % C::NSTAGE) * C::STAGE;
        const uint16_t *kt = kg + (int64_t)kb * C::BN * p.k_rs;
        const uint16_t *vt = vg + (int64_t)kb * C::BN * p.v_rs;
        const bool full = kb * C::BN + C::BN <= seq_k;
#pragma unroll
        for (int j = 0; j < C::K_DMA; ++j) {
            uint32_t off = k_voff[j];
            if (!full) off = (uint32_t)(min(k_row[j], seq_k - 1 - kb * C::BN) * p.k_rs + k_col[j]) * 2u;
            if (k_col[j] < p.d) dma16_s(kt, off, stage + (wave * C::K_DMA + j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < C::V_DMA; ++j) {
            uint32_t off = v_voff[j];
            if (!full) off = (uint32_t)(min(v_row[j], seq_k - 1 - kb * C::BN) * p.v_rs + v_col[j]) * 2u;
            if (v_col[j] < p.d) dma16_s(vt, off, stage + C::KTILE + (wave * C::V_DMA + j) * 1024);
        }
    };

    f32x16 acc[2][NV];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int n = 0; n < NV; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qb][n][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY};
    float l_run[2] = {0.f, 0.f};

    int k_read_off[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) k_read_off[s] = l31 * C::KROW + (((2 * s + hh) ^ k_swz<C::KROW>(l31)) * 16);
    int v_read_off[NV];
    {
        const int v_row_lane = 4 * hh + ((lane & 15) >> 2);
        const int v_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
#pragma unroll
        for (int n = 0; n < NV; ++n) v_read_off[n] = v_lds_off<NV>(v_row_lane, n * 4 + v_ch_lane) + (lane & 1) * 8;
    }

    // ---- one 64-key tile, hand-interleaved so the matrix pipe and the VALU work at the same time -------
    //   phase 1   S_A = K Q_A^T                      (2*KD MFMAs)
    //   (row max of A, rescale factors)
    //   phase 2   S_B = K Q_B^T      ||  P_A = exp2(S_A*c - m*c), row sums, 16-bit packing, O_A rescale
    //   (row max of B)
    //   phase 3   O_A += V^T P_A^T   ||  P_B = ..., O_B rescale
    //   phase 4   O_B += V^T P_B^T
    // `sched_barrier(0)` pins each MFMA next to its slice of the other block's element-wise work; hipcc
    // on its own clusters all MFMAs, then all 66 exps, then all MFMAs (seen in the ISA).
    constexpr int NQK = 2 * KD;      // S^T MFMAs per query block
    constexpr int NPV = 4 * NV;      // PV MFMAs per query block
    struct RowState { float mc, alpha; };

    auto row_max = [&](f32x16 (&st)[2], int qb) -> RowState {
        float mxa = st[0][0], mxb = st[0][8], mxc = st[1][0], mxd = st[1][8];
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            mxa = fmaxf(mxa, st[0][r]);
            mxb = fmaxf(mxb, st[0][8 + r]);
            mxc = fmaxf(mxc, st[1][r]);
            mxd = fmaxf(mxd, st[1][8 + r]);
        }
        const float m_new = xhalf_max(fmaxf(fmaxf(fmaxf(mxa, mxb), fmaxf(mxc, mxd)), m_run[qb]));
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        RowState rs;
        rs.mc = m_use * c2;
        rs.alpha = fast_exp2(m_run[qb] * c2 - rs.mc);
        m_run[qb] = m_new;
        return rs;
    };
    // element-wise slice `i` of `n` for one query block: pairs [16*i/n, 16*(i+1)/n) of each 32-key half
    // -> exp2, row-sum, pack to 16 bit (pf[kk][ks] = B operand of PV K-step ks of half kk), plus the
    // matching share of the O rescale
    auto ew_slice = [&](f32x16 (&st)[2], u32x4 (&pf)[2][2], f32x2 &rs2, const RowState &rw, int qb, int i, int n) {
        const f32x2 c2v = {c2, c2}, mcv = {-rw.mc, -rw.mc};
        const int p0 = (16 * i) / n, p1 = (16 * (i + 1)) / n;   // pair indices over 2 halves x 8 pairs
#pragma unroll
        for (int pi = 0; pi < 16; ++pi) {
            if (pi >= p0 && pi < p1) {
                const int kk = pi >> 3, r = (pi & 7) * 2;
                f32x2 x = {st[kk][r], st[kk][r + 1]};
                x = __builtin_elementwise_fma(x, c2v, mcv);
                x[0] = fast_exp2(x[0]);
                x[1] = fast_exp2(x[1]);
                rs2 += x;
                pf[kk][r >> 3][(r & 7) >> 1] = E::pack2(x[0], x[1]);
            }
        }
        // O rescale: 16*NV registers of this query block, spread over the n slices
        const int a0 = (16 * NV * i) / n, a1 = (16 * NV * (i + 1)) / n;
#pragma unroll
        for (int t = 0; t < 16 * NV; ++t)
            if (t >= a0 && t < a1) acc[qb][t >> 4][t & 15] *= rw.alpha;
    };
    auto apply_mask = [&](f32x16 (&st)[2], int kb, int qb) {
        const int my_q = q0 + 32 * qb + l31;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * C::BN + kk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (key >= seq_k || (p.causal && key > my_q)) st[kk][r] = -INFINITY;
            }
    };

    auto block = [&](int kb, const char *kbuf, const char *vbuf, auto MASKED) {
        constexpr bool kMasked = decltype(MASKED)::value;
        f32x16 sA[2], sB[2];
        u32x4 pfA[2][2], pfB[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sA[kk][r] = 0.f; sB[kk][r] = 0.f; }
        // ---- phase 1 ------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NQK; ++i) {
            const int kk = i / KD, s = i - kk * KD;
            const u32x4 a = lds_read_16B(kbuf, k_read_off[s] + kk * 32 * C::KROW);
            sA[kk] = E::mfma(a, qf[0][s], sA[kk]);
        }
        if (kMasked) apply_mask(sA, kb, 0);
        const RowState rwA = row_max(sA, 0);
        f32x2 rsA = {0.f, 0.f}, rsB = {0.f, 0.f};
        // ---- phase 2: S_B MFMAs || element-wise work of A ------------------------------------------
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NQK; ++i) {
            const int kk = i / KD, s = i - kk * KD;
            const u32x4 a = lds_read_16B(kbuf, k_read_off[s] + kk * 32 * C::KROW);
            sB[kk] = E::mfma(a, qf[1][s], sB[kk]);
            ew_slice(sA, pfA, rsA, rwA, 0, i, NQK);
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run[0] = l_run[0] * rwA.alpha + (rsA[0] + rsA[1]);
        if (kMasked) apply_mask(sB, kb, 1);
        const RowState rwB = row_max(sB, 1);
        // ---- phase 3: PV_A MFMAs || element-wise work of B -----------------------------------------
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            const int n = i % NV, ks = (i / NV) & 1, kk = i / (2 * NV);
            const int rows = (kk * 32 + ks * 16) * C::VROW;
            const u32x2 lo = lds_read_tr16_8B(vbuf, v_read_off[n] + rows);
            const u32x2 hi = lds_read_tr16_8B(vbuf, v_read_off[n] + rows + 8 * C::VROW);
            const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
            acc[0][n] = E::mfma(a, pfA[kk][ks], acc[0][n]);
            ew_slice(sB, pfB, rsB, rwB, 1, i, NPV);
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run[1] = l_run[1] * rwB.alpha + (rsB[0] + rsB[1]);
        // ---- phase 4: PV_B ---------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            const int n = i % NV, ks = (i / NV) & 1, kk = i / (2 * NV);
            const int rows = (kk * 32 + ks * 16) * C::VROW;
            const u32x2 lo = lds_read_tr16_8B(vbuf, v_read_off[n] + rows);
            const u32x2 hi = lds_read_tr16_8B(vbuf, v_read_off[n] + rows + 8 * C::VROW);
            const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
            acc[1][n] = E::mfma(a, pfB[kk][ks], acc[1][n]);
        }
    };

#pragma unroll
    for (int t = 0; t < C::NSTAGE - 1; ++t)
        if (t < nkb) issue(t);
    for (int kb = 0; kb < nkb; ++kb) {
        wait_vmcnt<0>();                       // NSTAGE == 2: only tile kb can be in flight here
        __builtin_amdgcn_s_barrier();
        if (kb + 1 < nkb) issue(kb + 1);
        const bool active = wave_has_rows && !(p.causal && kb * C::BN > q0 + C::QW - 1);
        if (active) {
            const char *kbuf = smem + (kb % C::NSTAGE) * C::STAGE;
            const char *vbuf = kbuf + C::KTILE;
            const bool need_mask = (kb * C::BN + C::BN > seq_k) || (p.causal && kb * C::BN + C::BN - 1 > q0);
            if (need_mask) block(kb, kbuf, vbuf, std::true_type{});
            else block(kb, kbuf, vbuf, std::false_type{});
        }
    }

    if (!wave_has_rows) return;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int my_q = q0 + 32 * qb + l31;
        const float l_tot = xhalf_sum(l_run[qb]);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        if (my_q < seq_q) {
            if (hh == 0 && p.lse != nullptr) {
                const float lse = l_tot > 0.f ? (m_run[qb] * c2 + fast_log2(l_tot)) * kLn2 : -INFINITY;
                p.lse[((int64_t)batch * p.h + head) * p.lse_stride + my_q] = lse;
            }
            uint16_t *og = reinterpret_cast<uint16_t *>(p.o) + o_off + (int64_t)my_q * p.o_rs + (int64_t)head * p.o_hs;
#pragma unroll
            for (int n = 0; n < NV; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = n * 32 + 8 * g + 4 * hh;
                    if (d0 < p.d) {
                        u32x2 w = {E::pack2(acc[qb][n][4 * g + 0] * inv, acc[qb][n][4 * g + 1] * inv),
                                   E::pack2(acc[qb][n][4 * g + 2] * inv, acc[qb][n][4 * g + 3] * inv)};
                        *reinterpret_cast<u32x2 *>(og + d0) = w;
                    }
                }
        }
    }
}

#ifndef BP_FLASH2_NWAVE
#define BP_FLASH2_NWAVE 4
#endif

template <class ET, int KD, int NV>
static hipError_t launch_one(const FlashParams &p, hipStream_t stream) {
    constexpr int NW = BP_FLASH2_NWAVE;
    const int n_qtiles = (p.max_sq + NW * 64 - 1) / (NW * 64);
    const int grid = xcd_grid(p.b * p.h, n_qtiles);
    hipLaunchKernelGGL((flash_fwd_dma2_kernel<ET, KD, NV, NW>), dim3(grid), dim3(NW * 64), 0, stream, p);
    return hipGetLastError();
}

// Two-query-block variant, head dims up to 64 (acc + scores for 64 rows must fit 256 VGPRs).
// Returns hipErrorNotSupported for other shapes: the caller then uses flash_fwd_dma.hip.
hipError_t launch_flash_fwd_dma2(const FlashParams &p, int dtype, hipStream_t stream) {
    if (p.v == nullptr || p.d > 64) return hipErrorNotSupported;
    const int kd = (p.d + 15) / 16;
    if (dtype == 1) {
        switch (kd) {
            case 1: return launch_one<BF16, 1, 1>(p, stream);
            case 2: return launch_one<BF16, 2, 1>(p, stream);
            case 3: return launch_one<BF16, 3, 2>(p, stream);
            default: return launch_one<BF16, 4, 2>(p, stream);
        }
    }
    switch (kd) {
        case 1: return launch_one<F16, 1, 1>(p, stream);
        case 2: return launch_one<F16, 2, 1>(p, stream);
        case 3: return launch_one<F16, 3, 2>(p, stream);
        default: return launch_one<F16, 4, 2>(p, stream);
    }
}

}  // namespace bp
