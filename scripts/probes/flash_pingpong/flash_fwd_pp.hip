// Fused attention forward for gfx950, "ping-pong" version for head dims 33..64 (the trunk shapes): the same math,
// tile algebra, LDS images and two tile bodies as flash_fwd_dma.hip, with the two pipes of a SIMD kept busy BY
// CONSTRUCTION instead of by chance.
//
// Why.  For d = 64 a 64-key tile costs a wave 16 MFMAs (512 matrix-pipe cycles) and about 110 VALU + 32 v_exp_f32
// (about 800 VALU cycles).  With four independent waves per SIMD, each running S-MFMAs -> softmax -> PV-MFMAs in
// turn, the measured time per tile and wave is 0.88 x (MFMA + VALU): the pipes hardly overlap, because nothing
// keeps one wave in its matrix phase while another is in its softmax (a wave whose MFMA
// stream meets another wave's VALU stream on the same SIMD does overlap, scripts/probes/pipe_probe.hip).
// Here a workgroup is EIGHT waves in two groups, one per query tile (two adjacent 128-query tiles of one
// (sample, head), so they also share one K/V stream), and the groups alternate phases under a workgroup
// barrier:
//
//     phase 2j   : group 0   P.V of tile j-1, S of tile j   (matrix pipe)   | group 1   softmax of tile j-1   (VALU)
//     phase 2j+1 : group 0   softmax of tile j              (VALU)          | group 1   P.V of j-1, S of j    (matrix pipe)
//
// Every SIMD hosts one wave of each group, so at any time one of them feeds the matrix pipe and the other the
// VALU.  The K/V ring has four 64-key slots (a tile is last read three phases after it was first needed) and is
// filled by all eight waves; the phase barrier doubles as the ring's visibility / reuse barrier.
// Causal work balance: a workgroup takes the tile pairs u and npairs-1-u one after the other.
#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"

#ifndef BP_PP_PREFETCH
#define BP_PP_PREFETCH 2
#endif

namespace bp {

template <int KD, int NV>
struct FlashPpCfg {
    static constexpr int BM = 128, BN = 64, NT = 512, NWAVE = 8, NS = 4;
    static constexpr int KROW = 128;              // KD <= 4
    static constexpr int KSLOTS = KROW / 16;
    static constexpr int VROW = NV * 64;
    static constexpr int VCH = NV * 4;
    static constexpr int KTILE = BN * KROW;
    static constexpr int VTILE = BN * VROW;
    static constexpr int STAGE = KTILE + VTILE;
    static constexpr int K_DMA = KTILE / 1024 / NWAVE;   // 1
    static constexpr int V_DMA = VTILE / 1024 / NWAVE;   // 1
    static constexpr int K_ROWS_PER_DMA = 1024 / KROW;
    static constexpr int TILE_DMA = K_DMA + V_DMA;
    static_assert(KD >= 3 && KD <= 4 && NV == 2 && K_DMA == 1 && V_DMA == 1, "ping-pong kernel: head dims 33..64");
};

template <class ET> struct PpProbLimit;   // largest tile row sum the steady-state body accepts (flash_fwd_dma.hip)
template <> struct PpProbLimit<BF16> { static constexpr float value = 1073741824.f; };
template <> struct PpProbLimit<F16> { static constexpr float value = 16384.f; };

BP_DEV void wait_vmcnt_tiles(int tiles_in_flight, int per_tile) {   // per_tile == 2
    switch (tiles_in_flight) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<2>(); break;
        case 2: wait_vmcnt<4>(); break;
        default: wait_vmcnt<6>(); break;
    }
}

// Query tiles qt_a (group 0) and qt_b (group 1, -1: none) of (sample, head) bh.
template <class ET, int KD, int NV, bool FULLD>
BP_DEV void flash_pp_job(const FlashParams p, char *smem, const uint32_t lds0, const int bh, const int qt_a, const int qt_b) {
    using C = FlashPpCfg<KD, NV>;
    using E = Elem<ET>;
    constexpr float kLimit = PpProbLimit<ET>::value;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;        // 0: leads (matrix phase first), 1: lags by one phase
    const int wg = wave & 3;          // 32-row block of the group's query tile
    const int l31 = lane & 31;
    const int hh = lane >> 5;

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
    const int qt_hi = max(qt_a, qt_b);
    if (min(qt_a, qt_b < 0 ? qt_a : qt_b) * C::BM >= seq_q) return;   // neither tile has rows (workgroup-uniform)

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + q_off + (int64_t)head * p.q_hs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + k_off + (int64_t)head * p.k_hs;
    const uint16_t *vg = reinterpret_cast<const uint16_t *>(p.v) + v_off + (int64_t)head * p.v_hs;

    // the K/V stream covers what the later tile needs
    int k_end = seq_k;
    if (p.causal) k_end = min(seq_k, qt_hi * C::BM + C::BM);
    const int nkb = (k_end + C::BN - 1) / C::BN;

    const int qt = grp == 0 ? qt_a : qt_b;
    const int q0 = qt * C::BM + wg * 32;
    const int my_q = q0 + l31;
    const bool wave_has_rows = qt >= 0 && q0 < seq_q;
    const float c2 = p.scale_log2e;
    const int my_nkb = !wave_has_rows ? 0 : p.causal ? min(nkb, (q0 + 31) / C::BN + 1) : nkb;
    const int my_clean_end = p.causal ? min(seq_k / C::BN, (q0 + 1) / C::BN) : seq_k / C::BN;

    if (!FULLD) {   // K pad slots are never written by the DMA and meet zero Q columns: they must be finite
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < C::NS * C::STAGE; off += C::NT * 16) lds_write_16B(smem, off, z);
        __syncthreads();
    }

    // ---- Q fragments (B operand of S^T = K Q^T) ----------------------------------------------------
    u32x4 qf[KD];
    {
        const uint16_t *row = qg + (int64_t)max(min(my_q, seq_q - 1), 0) * p.q_rs;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (FULLD || col < p.d) v = ld_global_16B(row + col);
            qf[s] = v;
        }
    }

    // ---- DMA descriptors: every wave moves one 1-KB piece of K and one of V per tile ------------------
    const int k_row = wave * C::K_ROWS_PER_DMA + lane / C::KSLOTS;
    const int k_col = ((lane % C::KSLOTS) ^ k_swz<C::KROW>(k_row)) * 8;
    const int vc = wave * 64 + lane;                       // linear 16-B chunk of the V tile
    const int v_row = vc / C::VCH, v_stored = vc - v_row * C::VCH;
    const int v_col = ((((v_stored >> 2) ^ ((v_row >> 1) & 1)) << 2) | (v_stored & 3)) * 8;   // NV == 2 swizzle
    const int kb_partial = (seq_k % C::BN) != 0 ? seq_k / C::BN : -1;
    const int last_row = seq_k - 1 - (seq_k / C::BN) * C::BN;
    const uint32_t k_voff = (uint32_t)(k_row * p.k_rs + k_col) * 2u, v_voff = (uint32_t)(v_row * p.v_rs + v_col) * 2u;
    const uint32_t k_voff_p = (uint32_t)(min(k_row, last_row) * p.k_rs + k_col) * 2u;
    const uint32_t v_voff_p = (uint32_t)(min(v_row, last_row) * p.v_rs + v_col) * 2u;
    const int64_t k_tile_stride = (int64_t)C::BN * p.k_rs, v_tile_stride = (int64_t)C::BN * p.v_rs;
    const uint16_t *kt = kg, *vt = vg;   // tile of the NEXT issue (tiles are issued in order)
    auto issue = [&](int kb) {
        const uint32_t stage = __builtin_amdgcn_readfirstlane(lds0 + (kb & (C::NS - 1)) * C::STAGE);
        const bool partial = kb == kb_partial;
        if (FULLD || k_col < p.d)
            dma16_s(kt, partial ? k_voff_p : k_voff, __builtin_amdgcn_readfirstlane(stage + wave * 1024));
        if (FULLD || v_col < p.d)
            dma16_s(vt, partial ? v_voff_p : v_voff, __builtin_amdgcn_readfirstlane(stage + C::KTILE + wave * 1024));
        kt += k_tile_stride;
        vt += v_tile_stride;
    };

    f32x16 acc[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float m_run = -INFINITY, mc = 0.f, l_run = 0.f;

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

    f32x16 st[2];        // scores of the tile between its matrix phase and its softmax phase
    u32x4 pf[2][2];      // P (16 bit) of the tile between its softmax phase and the next matrix phase: [half][16-key step]

    auto scores = [&](int kb) {
        const char *kbuf = smem + (kb & (C::NS - 1)) * C::STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x16 s_;
#pragma unroll
            for (int r = 0; r < 16; ++r) s_[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const u32x4 a = lds_read_16B(kbuf, k_read_off[s] + kk * 32 * C::KROW);
                s_ = E::mfma(a, qf[s], s_);
            }
            st[kk] = s_;
        }
    };
    auto exponentiate = [&]() {
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float x0 = fast_exp2(fmaf(st[kk][r], c2, -mc));
                const float x1 = fast_exp2(fmaf(st[kk][r + 1], c2, -mc));
                st[kk][r] = x0;
                st[kk][r + 1] = x1;
                rs0 += x0;
                rs1 += x1;
            }
        return rs0 + rs1;
    };
    auto online_max_step = [&](int kb) {
        int last = seq_k - 1;
        if (p.causal) last = min(last, my_q);
        int lim = last - kb * C::BN - 4 * hh;
        asm volatile("" : "+v"(lim));   // (keeps the compares inside this rarely taken branch, see flash_fwd_dma.hip)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kk * 32 + (r & 3) + 8 * (r >> 2) > lim) st[kk][r] = -INFINITY;
        float mxa = st[0][0], mxb = st[0][8], mxc = st[1][0], mxd = st[1][8];
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            mxa = fmaxf(mxa, st[0][r]);
            mxb = fmaxf(mxb, st[0][8 + r]);
            mxc = fmaxf(mxc, st[1][r]);
            mxd = fmaxf(mxd, st[1][8 + r]);
        }
        const float mt = xhalf_max(fmaxf(fmaxf(mxa, mxb), fmaxf(mxc, mxd)));
        const float m_new = fmaxf(mt, m_run);
        const float mc_new = (m_new == -INFINITY) ? 0.f : m_new * c2;
        const float alpha = fast_exp2(m_run * c2 - mc_new);
        l_run *= alpha;
#pragma unroll
        for (int n = 0; n < NV; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] *= alpha;
        m_run = m_new;
        mc = mc_new;
    };

    // ---- softmax phase of tile kb (VALU): st -> pf, row sum; exact body for the first / masked tiles and as the retry
    auto softmax_phase = [&](int kb) {
        if (kb < 0 || kb >= my_nkb) return;
        bool exact = kb == 0 || kb >= my_clean_end;
        bool have = true;
        float rs;
        for (;;) {
            if (!have) scores(kb);   // (retry: the K tile is still resident)
            if (__builtin_expect(exact, 0)) online_max_step(kb);
            rs = exponentiate();
            if (__builtin_expect(exact || __all(rs <= kLimit), 1)) break;
            exact = true;
            have = false;
        }
        l_run += rs;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[kk][ks][i] = E::pack2(st[kk][ks * 8 + 2 * i], st[kk][ks * 8 + 2 * i + 1]);
    };
    // ---- matrix phase of step j: O^T += V^T P^T of tile j-1, then S^T of tile j.  Sixteen MFMAs whose LDS operands are
    //      requested BP_PP_PREFETCH MFMAs ahead, in program order (the empty asm pins keep that order: between them hipcc, short
    //      of registers at 128, would otherwise fetch each operand right in front of its MFMA and wait for it).
    constexpr int PD = BP_PP_PREFETCH;   // operand prefetch distance in MFMAs
    auto matrix_phase = [&](int j) {
        __builtin_amdgcn_s_setprio(2);
        const int kb = j - 1;
        if (kb >= 0 && kb < my_nkb) {
            const char *vbuf = smem + (kb & (C::NS - 1)) * C::STAGE + C::KTILE;
            const bool skip_hi = p.causal && (kb * C::BN + 32 > q0 + 31);   // second half entirely above my rows
            // MFMA i of the block: half kk = i / (2 NV), 16-key step ks = (i / NV) & 1, column block n = i % NV
            auto v_operand = [&](int i) {
                const int rows = ((i / (2 * NV)) * 32 + ((i / NV) & 1) * 16) * C::VROW;
                const u32x2 lo = lds_read_tr16_8B(vbuf, v_read_off[i % NV] + rows);
                const u32x2 hi = lds_read_tr16_8B(vbuf, v_read_off[i % NV] + rows + 8 * C::VROW);
                return u32x4{lo[0], lo[1], hi[0], hi[1]};
            };
            auto pv_half = [&](int kk) {
                u32x4 a[PD];
#pragma unroll
                for (int i = 0; i < PD; ++i) a[i] = v_operand(kk * 2 * NV + i);
#pragma unroll
                for (int i = 0; i < 2 * NV; ++i) {
                    u32x4 cur = a[i % PD];
                    asm volatile("" : "+v"(cur));
                    acc[i % NV] = E::mfma(cur, pf[kk][(i / NV) & 1], acc[i % NV]);
                    asm volatile("" : "+v"(acc[i % NV]));
                    if (i + PD < 2 * NV) a[i % PD] = v_operand(kk * 2 * NV + i + PD);
                }
            };
            pv_half(0);
            if (!skip_hi) pv_half(1);
        }
        if (j < my_nkb) {
            const char *kbuf = smem + (j & (C::NS - 1)) * C::STAGE;
            // MFMA i: half kk = i & 1 (two independent accumulation chains, interleaved), 16-wide k-step s = i >> 1
            auto k_operand = [&](int i) { return lds_read_16B(kbuf, k_read_off[i >> 1] + (i & 1) * 32 * C::KROW); };
            u32x4 a[PD];
#pragma unroll
            for (int i = 0; i < PD; ++i) a[i] = k_operand(i);
#pragma unroll
            for (int i = 0; i < 2 * KD; ++i) {
                u32x4 cur = a[i % PD];
                asm volatile("" : "+v"(cur));
                if (i < 2) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    st[i & 1] = E::mfma(cur, qf[0], z);
                } else {
                    st[i & 1] = E::mfma(cur, qf[i >> 1], st[i & 1]);
                }
                asm volatile("" : "+v"(st[i & 1]));
                if (i + PD < 2 * KD) a[i % PD] = k_operand(i + PD);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- the stream ------------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < C::NS; ++t)
        if (t < nkb) issue(t);
#pragma unroll
    for (int s = 0; s < KD; ++s) settle(qf[s]);

    // step j: [tile j landed] barrier A | phase 2j | barrier B | phase 2j+1
    auto step_head = [&](int j) {
        if (j < nkb) {
            const int issued_last = min(nkb - 1, j < 2 ? C::NS - 1 : j + C::NS - 3);
            wait_vmcnt_tiles(issued_last - j, C::TILE_DMA);   // my share of tile j has landed ...
        }
        __builtin_amdgcn_s_barrier();                          // ... everybody's; all waves are done with tile j-2
        if (j >= 2 && j + C::NS - 2 < nkb) issue(j + C::NS - 2);
    };
    if (grp == 0) {
        for (int j = 0; j <= nkb; ++j) {
            step_head(j);
            matrix_phase(j);
            __builtin_amdgcn_s_barrier();
            softmax_phase(j);
        }
    } else {
        for (int j = 0; j <= nkb; ++j) {
            step_head(j);
            softmax_phase(j - 1);
            __builtin_amdgcn_s_barrier();
            matrix_phase(j);
        }
    }

    if (!wave_has_rows) return;
    const float l_tot = xhalf_sum(l_run);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (my_q < seq_q) {
        if (hh == 0 && p.lse != nullptr) {
            const float lse = l_tot > 0.f ? (mc + fast_log2(l_tot)) * kLn2 : -INFINITY;
            p.lse[((int64_t)batch * p.h + head) * p.lse_stride + my_q] = lse;
        }
        uint16_t *og = reinterpret_cast<uint16_t *>(p.o) + o_off + (int64_t)my_q * p.o_rs + (int64_t)head * p.o_hs;
#pragma unroll
        for (int n = 0; n < NV; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = n * 32 + 8 * g + 4 * hh;
                if (d0 < p.d) {
                    u32x2 w = {E::pack2(acc[n][4 * g + 0] * inv, acc[n][4 * g + 1] * inv),
                               E::pack2(acc[n][4 * g + 2] * inv, acc[n][4 * g + 3] * inv)};
                    *reinterpret_cast<u32x2 *>(og + d0) = w;
                }
            }
    }
}

// Work list: query tiles are taken in adjacent pairs, the heaviest pair first: pair u = tiles (n-1-2u, n-2-2u).
// Causal: a workgroup runs pair u and then pair npairs-1-u (equal key-tile totals, cf. flash_fwd_dma.hip).
#ifndef BP_PP_MINWAVES
#define BP_PP_MINWAVES 4
#endif
template <class ET, int KD, int NV, bool FULLD>
__global__ __launch_bounds__(512, BP_PP_MINWAVES) void flash_fwd_pp_kernel(const FlashParams p) {
    using C = FlashPpCfg<KD, NV>;
    __shared__ __attribute__((aligned(16))) char smem[C::NS * C::STAGE];
    const uint32_t lds0 = lds_base_addr(smem);
    const int npairs = (p.n_qtiles + 1) / 2;
    const int per_group = p.pair ? (npairs + 1) / 2 : npairs;
    int bh, slot;
    if (!xcd_map(blockIdx.x, p.b * p.h, per_group, bh, slot)) return;
    const int n = p.n_qtiles;
    const int u0 = slot, u1 = npairs - 1 - slot;
    flash_pp_job<ET, KD, NV, FULLD>(p, smem, lds0, bh, n - 1 - 2 * u0, n - 2 - 2 * u0);   // (second tile is -1 for an odd last pair)
    if (p.pair && u1 != u0) {
        __syncthreads();   // every wave is done with the ring before the next job's DMA refills it
        flash_pp_job<ET, KD, NV, FULLD>(p, smem, lds0, bh, n - 1 - 2 * u1, n - 2 - 2 * u1);
    }
}

template <class ET, int KD, int NV>
static hipError_t launch_kd(const FlashParams &p, hipStream_t stream) {
    const int npairs = (p.n_qtiles + 1) / 2;
    const int grid = xcd_grid(p.b * p.h, p.pair ? (npairs + 1) / 2 : npairs);
    if (p.d == 64) hipLaunchKernelGGL((flash_fwd_pp_kernel<ET, KD, NV, true>), dim3(grid), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((flash_fwd_pp_kernel<ET, KD, NV, false>), dim3(grid), dim3(512), 0, stream, p);
    return hipGetLastError();
}

bool flash_fwd_pp_supported(const FlashParams &p) {
    return p.v != nullptr && p.drop_thr == 0u && p.d > 32 && p.d <= 64 && p.n_qtiles >= 2;
}

// Requires what launch_flash_fwd_dma requires (head_dim % 8 == 0, aligned bases and strides) and flash_fwd_pp_supported.
hipError_t launch_flash_fwd_pp(const FlashParams &p, int dtype, hipStream_t stream) {
    if (p.d <= 48) return dtype == 1 ? launch_kd<BF16, 3, 2>(p, stream) : launch_kd<F16, 3, 2>(p, stream);
    return dtype == 1 ? launch_kd<BF16, 4, 2>(p, stream) : launch_kd<F16, 4, 2>(p, stream);
}

}  // namespace bp
