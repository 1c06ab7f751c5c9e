// Fused attention forward for gfx950, LDS-DMA ring version: the fast path for 16-byte friendly
// shapes (head_dim % 8 == 0, aligned rows) -- every shape the reference kernel accepts
// (csrc/flash_attn/fmha_api.cpp:245).  Same math and tile algebra as flash_fwd.hip; what changes:
//   * K/V tiles (64 keys) go straight from global memory to a 3-slot LDS ring with
//     global_load_lds_dwordx4: no VGPR staging, no ds_write pass, two tiles always in flight,
//     counted s_waitcnt vmcnt + ONE raw s_barrier per tile (bp_dma.h);
//   * K rows: power-of-two pitch + XOR slot swizzle, V rows: 64-B chunk swizzle, both applied on the
//     DMA source address; every MFMA operand read is `lane base + immediate`;
//   * a 32-key half of a diagonal tile that is entirely above a wave's rows is skipped.
#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"

// tuning knobs (compile-time; defaults are the measured best)
#ifndef BP_FLASH_STAGES
#define BP_FLASH_STAGES 2
#endif
#ifndef BP_FLASH_DEFER
#define BP_FLASH_DEFER 8.f   // deferred-rescale threshold in exp2 units; negative = always rescale
#endif
#ifndef BP_FLASH_UNROLL
#define BP_FLASH_UNROLL 1   // unroll the key loop by the ring depth: static LDS slot addresses
#endif
#ifndef BP_FLASH_MINWAVES
#define BP_FLASH_MINWAVES 1
#endif

namespace bp {

template <int KD, int NV, bool HAS_V>
struct FlashDmaCfg {
    static constexpr int BM = 128, BN = 64, NT = 256, NWAVE = 4, NSTAGE = BP_FLASH_STAGES;
    static constexpr int KROW = KD <= 4 ? 128 : 256;
    static constexpr int KSLOTS = KROW / 16;
    static constexpr int VROW = NV * 64;
    static constexpr int VCH = NV * 4;
    static constexpr int KTILE = BN * KROW;
    static constexpr int VTILE = HAS_V ? BN * VROW : 0;
    static constexpr int STAGE = KTILE + VTILE;
    static constexpr int K_DMA = KTILE / 1024 / NWAVE;            // 2 or 4 per wave per tile
    static constexpr int V_DMA = HAS_V ? VTILE / 1024 / NWAVE : 0;  // 1..4
    static constexpr int DMA_PER_STAGE = K_DMA + V_DMA;
    static constexpr int K_ROWS_PER_DMA = 1024 / KROW;
};

// One 128-query tile `qt` of (sample, head) `bh`: the whole online-softmax sweep over its key tiles.
template <class ET, int KD, int NV, bool HAS_V>
BP_DEV void flash_fwd_tile(const FlashParams p, char *smem, const uint32_t lds0, const int bh, const int qt) {
    using C = FlashDmaCfg<KD, NV, HAS_V>;
    using E = Elem<ET>;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    if (qt * C::BM >= seq_q) return;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + q_off + (int64_t)head * p.q_hs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + k_off + (int64_t)head * p.k_hs;
    const uint16_t *vg = HAS_V ? reinterpret_cast<const uint16_t *>(p.v) + v_off + (int64_t)head * p.v_hs : nullptr;

    int k_end = seq_k;
    if (p.causal) k_end = min(seq_k, qt * C::BM + C::BM);
    const int nkb = (k_end + C::BN - 1) / C::BN;

    const int q0 = qt * C::BM + wave * 32;
    const int my_q = q0 + l31;
    const bool wave_has_rows = q0 < seq_q;
    const float c2 = p.scale_log2e;

    // K pad slots (head_dim not a multiple of 16, or pitch wider than the row) are never written by
    // the DMA and meet zero Q columns in the MFMA: they must hold finite values -> zero them once.
    if (p.d * 2 != C::KROW) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < C::NSTAGE * C::STAGE; off += C::NT * 16) lds_write_16B(smem, off, z);
        __syncthreads();
    }

    // ---- Q fragments (B operand of S^T = K Q^T) ----------------------------------------------------
    u32x4 qf[KD];
    {
        const uint16_t *row = qg + (int64_t)min(my_q, seq_q - 1) * p.q_rs;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (col < p.d) v = ld_global_16B(row + col);
            qf[s] = v;
        }
#pragma unroll
        for (int s = 0; s < KD; ++s) settle(qf[s]);
    }

    // ---- per-lane DMA source descriptors: tile row / column of the 16-B chunk this lane moves ----------
    int k_row[C::K_DMA], k_col[C::K_DMA];
    uint32_t k_voff[C::K_DMA];
#pragma unroll
    for (int j = 0; j < C::K_DMA; ++j) {
        const int row = (wave * C::K_DMA + j) * C::K_ROWS_PER_DMA + lane / C::KSLOTS;
        k_row[j] = row;
        k_col[j] = ((lane % C::KSLOTS) ^ k_swz<C::KROW>(row)) * 8;
        k_voff[j] = (uint32_t)(row * p.k_rs + k_col[j]) * 2u;
    }
    int v_row[HAS_V ? C::V_DMA : 1], v_col[HAS_V ? C::V_DMA : 1];
    uint32_t v_voff[HAS_V ? C::V_DMA : 1];
    if (HAS_V) {
#pragma unroll
        for (int j = 0; j < C::V_DMA; ++j) {
            const int c = (wave * C::V_DMA + j) * 64 + lane;   // linear 16-B chunk of the tile
            const int row = c / C::VCH, stored = c - row * C::VCH;
            int c64 = stored >> 2;
            if (NV == 2) c64 ^= (row >> 1) & 1;
            if (NV == 4) c64 ^= row & 3;
            v_row[j] = row;
            v_col[j] = ((c64 << 2) | (stored & 3)) * 8;
            v_voff[j] = (uint32_t)(row * p.v_rs + v_col[j]) * 2u;
        }
    }
    // Full tiles: scalar base (+= 64 rows per tile) + constant per-lane byte offset -> no VALU at all.
    // The last, partial tile clamps its rows to the final valid one (those keys are masked later).
    auto issue = [&](int kb) {
        const uint32_t stage = lds0 + (kb % C::NSTAGE) * C::STAGE;
        const uint16_t *kt = kg + (int64_t)kb * C::BN * p.k_rs;
        const uint16_t *vt = HAS_V ? vg + (int64_t)kb * C::BN * p.v_rs : nullptr;
        const bool full = kb * C::BN + C::BN <= seq_k;
#pragma unroll
        for (int j = 0; j < C::K_DMA; ++j) {
            uint32_t off = k_voff[j];
            if (!full) off = (uint32_t)(min(k_row[j], seq_k - 1 - kb * C::BN) * p.k_rs + k_col[j]) * 2u;
            if (k_col[j] < p.d) dma16_s(kt, off, stage + (wave * C::K_DMA + j) * 1024);
        }
        if (HAS_V) {
#pragma unroll
            for (int j = 0; j < C::V_DMA; ++j) {
                uint32_t off = v_voff[j];
                if (!full) off = (uint32_t)(min(v_row[j], seq_k - 1 - kb * C::BN) * p.v_rs + v_col[j]) * 2u;
                if (v_col[j] < p.d) dma16_s(vt, off, stage + C::KTILE + (wave * C::V_DMA + j) * 1024);
            }
        }
    };

    f32x16 acc[HAS_V ? NV : 1];
#pragma unroll
    for (int n = 0; n < (HAS_V ? NV : 1); ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float m_run = -INFINITY;
    float l_run = 0.f;

    int k_read_off[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) k_read_off[s] = l31 * C::KROW + (((2 * s + hh) ^ k_swz<C::KROW>(l31)) * 16);
    int v_read_off[HAS_V ? NV : 1];
    if (HAS_V) {
        const int v_row_lane = 4 * hh + ((lane & 15) >> 2);
        const int v_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
#pragma unroll
        for (int n = 0; n < NV; ++n) v_read_off[n] = v_lds_off<NV>(v_row_lane, n * 4 + v_ch_lane) + (lane & 1) * 8;
    }

#ifdef BP_PROFILE_PHASES
    // debug build: per-wave cycle stamps (s_memtime) summed per phase into p.lse as raw uint64 pairs
    unsigned long long ph[5] = {0, 0, 0, 0, 0};
    unsigned long long t_prev = __builtin_readcyclecounter();
    const unsigned long long t_start = t_prev;
#define BP_STAMP(i) { unsigned long long t_now = __builtin_readcyclecounter(); ph[i] += t_now - t_prev; t_prev = t_now; }
#else
#define BP_STAMP(i)
#endif
    auto block = [&](int kb, const char *kbuf, const char *vbuf, auto MASKED) {
        constexpr bool kMasked = decltype(MASKED)::value;
        // second 32-key half entirely above my rows?  (only possible on a masked tile)
        const bool skip_hi = kMasked && p.causal && (kb * C::BN + 32 > q0 + 31);
        f32x16 st[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 1 && skip_hi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kk][r] = -INFINITY;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kk][r] = 0.f;
#pragma unroll
                for (int s = 0; s < KD; ++s) {
#ifdef BP_ABL_NOLDS   // ablation: operand from registers instead of LDS (wrong results, timing only)
                    u32x4 a = qf[(s + 1) % KD];
                    asm volatile("" : "+v"(a));
#else
                    const u32x4 a = lds_read_16B(kbuf, k_read_off[s] + kk * 32 * C::KROW);
#endif
                    st[kk] = E::mfma(a, qf[s], st[kk]);
                }
                if (kMasked) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb * C::BN + kk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        const bool dead = key >= seq_k || (p.causal && key > my_q);
                        if (dead) st[kk][r] = -INFINITY;
                    }
                }
            }
        }
        BP_STAMP(1)
        // row max: four independent chains (short dependency depth), then the other half-wave
        // (plain fmaxf chains: hipcc fuses each pair into one v_max3_f32 and knows the MFMA->VALU
        //  read hazard, which an inline-asm v_max3 on fresh MFMA results would bypass)
        float mxa = st[0][0], mxb = st[0][8], mxc = st[1][0], mxd = st[1][8];
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            mxa = fmaxf(mxa, st[0][r]);
            mxb = fmaxf(mxb, st[0][8 + r]);
            mxc = fmaxf(mxc, st[1][r]);
            mxd = fmaxf(mxd, st[1][8 + r]);
        }
#ifdef BP_ABL_NOMAX   // ablation: no row-max reduction (timing only)
        const float mt = st[0][0];
#else
        const float mt = xhalf_max(fmaxf(fmaxf(mxa, mxb), fmaxf(mxc, mxd)));   // max of this tile's scores
#endif
        // Deferred rescale: while no row's maximum grows by more than BP_FLASH_DEFER (in exp2 units) the
        // old reference maximum is kept -- P <= 2^BP_FLASH_DEFER, harmless in fp32 / bf16 / fp16 -- and
        // the O / l rescale (32 multiplies + an exp2 per tile) is skipped for the whole wave.  A fresh row
        // (m_run = -inf) or a fully masked tile row (NaN difference) fails the test and takes the exact path.
        const bool defer = HAS_V && BP_FLASH_DEFER >= 0.f && __all((mt - m_run) * c2 <= BP_FLASH_DEFER);
        const float m_new = defer ? m_run : fmaxf(mt, m_run);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float mc = m_use * c2;
        float alpha = 1.f;
        if (!defer) alpha = fast_exp2(m_run * c2 - mc);
        m_run = m_new;
        // p = exp2(s*c2 - mc): packed fma on register pairs, packed row-sum accumulation
        const f32x2 c2v = {c2, c2}, mcv = {-mc, -mc};
        f32x2 rs2 = {0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 x = {st[kk][r], st[kk][r + 1]};
                x = __builtin_elementwise_fma(x, c2v, mcv);
#ifndef BP_ABL_NOEXP   // ablation builds only (timing experiments, results are wrong)
                x[0] = fast_exp2(x[0]);
                x[1] = fast_exp2(x[1]);
#endif
                st[kk][r] = x[0];
                st[kk][r + 1] = x[1];
#ifndef BP_ABL_NOSUM
                rs2 += x;
#endif
            }
        const float rs = rs2[0] + rs2[1];
        l_run = defer ? l_run + rs : l_run * alpha + rs;
        BP_STAMP(2)
#ifdef BP_ABL_NOPV
        asm volatile("" ::"v"(st[0][0]), "v"(st[0][15]), "v"(st[1][0]), "v"(st[1][15]), "v"(alpha));
        if (false) {
#else
        if (HAS_V) {
#endif
#ifndef BP_ABL_NORESCALE
            if (!defer) {
#pragma unroll
                for (int n = 0; n < NV; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[n][r] *= alpha;
            }
#endif
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk == 1 && skip_hi) continue;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 pf;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        pf[i] = E::pack2(st[kk][ks * 8 + 2 * i], st[kk][ks * 8 + 2 * i + 1]);
                    const int rows = (kk * 32 + ks * 16) * C::VROW;
#pragma unroll
                    for (int n = 0; n < NV; ++n) {
#ifdef BP_ABL_NOLDS
                        u32x4 a = qf[n % KD];
                        asm volatile("" : "+v"(a));
#else
                        const u32x2 lo = lds_read_tr16_8B(vbuf, v_read_off[n] + rows);
                        const u32x2 hi = lds_read_tr16_8B(vbuf, v_read_off[n] + rows + 8 * C::VROW);
                        const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
#endif
#ifdef BP_FLASH_SETPRIO
                        __builtin_amdgcn_s_setprio(1);
#endif
                        acc[n] = E::mfma(a, pf, acc[n]);
#ifdef BP_FLASH_SETPRIO
                        __builtin_amdgcn_s_setprio(0);
#endif
                    }
                }
            }
        }
    };

    // ring: NSTAGE-1 tiles are in flight before a tile is consumed
#pragma unroll
    for (int t = 0; t < C::NSTAGE - 1; ++t)
        if (t < nkb) issue(t);
    // One ring step; SLOT is the ring slot as a compile-time constant when the loop is unrolled by the ring depth
    // (BP_FLASH_UNROLL), so the LDS addresses of all operand reads fold into instruction offsets, or -1.
    auto ring_step = [&](int kb, auto SLOT) {
        constexpr int kSlot = decltype(SLOT)::value;
        // tiles kb+1 .. kb+NSTAGE-2 may still be in flight; tile kb must have landed
        const int later = min(nkb - 1 - kb, C::NSTAGE - 2);
#ifndef BP_ABL_NOVMWAIT
        if (later >= 2) wait_vmcnt<2 * C::DMA_PER_STAGE>();
        else if (later == 1) wait_vmcnt<C::DMA_PER_STAGE>();
        else wait_vmcnt<0>();
#endif
#ifndef BP_ABL_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
#ifndef BP_ABL_NODMA
        if (kb + C::NSTAGE - 1 < nkb) issue(kb + C::NSTAGE - 1);
#endif
        BP_STAMP(0)
        const bool active = wave_has_rows && !(p.causal && kb * C::BN > q0 + 31);
        if (active) {
            const char *kbuf = smem + (kSlot >= 0 ? kSlot : kb % C::NSTAGE) * C::STAGE;
            const char *vbuf = kbuf + C::KTILE;
            const bool need_mask = (kb * C::BN + C::BN > seq_k) || (p.causal && kb * C::BN + C::BN - 1 > q0);
            if (need_mask) block(kb, kbuf, vbuf, std::true_type{});
            else block(kb, kbuf, vbuf, std::false_type{});
        }
        BP_STAMP(4)
    };
#if BP_FLASH_UNROLL
    static_assert(C::NSTAGE == 2, "the unrolled loop assumes a 2-slot ring");
    for (int kb = 0; kb < nkb; kb += 2) {
        ring_step(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < nkb) ring_step(kb + 1, std::integral_constant<int, 1>{});
    }
#else
    for (int kb = 0; kb < nkb; ++kb) ring_step(kb, std::integral_constant<int, -1>{});
#endif
#ifdef BP_PROFILE_PHASES
    if (lane == 0 && p.o_bs == -12345) {   // never true: keeps the stamps alive without touching outputs
        p.lse[0] = (float)(ph[0] + ph[1] + ph[2] + ph[3] + ph[4]);
    }
    if (lane == 0 && p.prof != nullptr) {
        const int64_t w = ((int64_t)blockIdx.x * 4 + wave) * 8;
        p.prof[w + 0] = ph[0]; p.prof[w + 1] = ph[1]; p.prof[w + 2] = ph[2]; p.prof[w + 3] = ph[3];
        p.prof[w + 4] = ph[4]; p.prof[w + 5] = __builtin_readcyclecounter() - t_start; p.prof[w + 6] = nkb;
        p.prof[w + 7] = qt;
    }
#endif

    if (!wave_has_rows) return;
    const float l_tot = xhalf_sum(l_run);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (my_q < seq_q) {
        if (hh == 0 && p.lse != nullptr) {
            const float lse = l_tot > 0.f ? (m_run * c2 + fast_log2(l_tot)) * kLn2 : -INFINITY;
            p.lse[((int64_t)batch * p.h + head) * p.lse_stride + my_q] = lse;
        }
        if (HAS_V) {
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
}

// Work order.  The dispatcher hands workgroups to the CUs of an XCD strictly round-robin and IN ORDER: block
// i+1 is not placed before block i, and block i waits for a free slot on ITS CU even when other CUs idle
// (scripts/probes/dispatch_order.hip).  With causal tiles of 1..n key blocks in block order, a CU keeps getting
// the same tile length and the short ones wait for the long ones: 64 time units instead of 36 for S = 1024 in a
// model of that dispatcher, and 8.3 vs 5.9 ms in the probe.  So a causal workgroup takes TWO query tiles of its
// (sample, head), the heaviest remaining and the lightest (t and n-1-t): every workgroup then carries the same
// n+1 key blocks, nothing waits, and both tiles still belong to one group, i.e. one XCD's L2 holds their K/V.
template <class ET, int KD, int NV, bool HAS_V>
__global__ __launch_bounds__(256, BP_FLASH_MINWAVES) void flash_fwd_dma_kernel(const FlashParams p) {
    using C = FlashDmaCfg<KD, NV, HAS_V>;
    __shared__ __attribute__((aligned(16))) char smem[C::NSTAGE * C::STAGE];
    const uint32_t lds0 = lds_base_addr(smem);
    const int per_group = p.pair ? (p.n_qtiles + 1) / 2 : p.n_qtiles;
    int bh, slot;
    if (!xcd_map(blockIdx.x, p.b * p.h, per_group, bh, slot)) return;
    const int heavy = p.n_qtiles - 1 - slot;
    const int npass = (p.pair && slot != heavy) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass) __syncthreads();   // every wave is done with the ring before the next tile's DMA refills it
        flash_fwd_tile<ET, KD, NV, HAS_V>(p, smem, lds0, bh, pass ? slot : heavy);
    }
}

template <class ET, int KD, int NV, bool HAS_V>
static hipError_t launch_one(const FlashParams &p, hipStream_t stream) {
    const int grid = xcd_grid(p.b * p.h, p.pair ? (p.n_qtiles + 1) / 2 : p.n_qtiles);
    hipLaunchKernelGGL((flash_fwd_dma_kernel<ET, KD, NV, HAS_V>), dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <class ET, bool HAS_V>
static hipError_t launch_dim(const FlashParams &p, hipStream_t stream) {
    switch ((p.d + 15) / 16) {
        case 1: return launch_one<ET, 1, 1, HAS_V>(p, stream);
        case 2: return launch_one<ET, 2, 1, HAS_V>(p, stream);
        case 3: return launch_one<ET, 3, 2, HAS_V>(p, stream);
        case 4: return launch_one<ET, 4, 2, HAS_V>(p, stream);
        case 5: return launch_one<ET, 5, 3, HAS_V>(p, stream);
        case 6: return launch_one<ET, 6, 3, HAS_V>(p, stream);
        case 7: return launch_one<ET, 7, 4, HAS_V>(p, stream);
        default: return launch_one<ET, 8, 4, HAS_V>(p, stream);
    }
}

// Requires head_dim % 8 == 0, 16-byte aligned bases, strides multiples of 8, seq_k >= 1 per sequence
// handled inside (nkb == 0 issues nothing).
hipError_t launch_flash_fwd_dma(const FlashParams &p, int dtype, hipStream_t stream) {
    const bool has_v = p.v != nullptr;
    if (dtype == 1) return has_v ? launch_dim<BF16, true>(p, stream) : launch_dim<BF16, false>(p, stream);
    return has_v ? launch_dim<F16, true>(p, stream) : launch_dim<F16, false>(p, stream);
}

}  // namespace bp
