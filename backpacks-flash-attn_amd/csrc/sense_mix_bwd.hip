// Backward of the fused Backpack sense combination (second half) for gfx950.
//
// Forward (sense_mix_dma.hip):  out[t,:] = sum_l sum_{s<=t} P_l[t,s] C[s,l,:],  P_l[t,s] = exp(scale q_l[t].k_l[s] - lse_l[t])
// The reference leaves this to autograd through `torch.sum(contextualization @ content, dim=1)` and the softmax
// (training/src/models/backpack.py:116-122,313), which keeps three (B,k,S,S) fp32 tensors alive.  Here nothing of
// that size exists; P is recomputed from the saved log-sum-exp, as the attention backward does.
//
//   dC[s,l,:]  = sum_{t>=s} P_l[t,s] dout[t,:]                                     -> sense_mix_dc_kernel (this file)
//   dP_l[t,s]  = dout[t,:] . C[s,l,:]          (a 768-deep contraction: a plain GEMM, done by the BLAS library in
//                                               slabs of 128 queries into a (B, S*k, 128) buffer, see bp_hip/__init__.py)
//   D_l[t]     = sum_s P_l[t,s] dP_l[t,s]
//   dS_l[t,s]  = P_l[t,s] (dP_l[t,s] - D_l[t])
//   dq_l[t]    = scale sum_s dS_l[t,s] k_l[s]                                      -> sense_dq_kernel
//   dk_l[s]    = scale sum_t dS_l[t,s] q_l[t]                                      -> sense_dk_kernel
//
// sense_mix_dc_kernel is the forward kernel with the roles of queries and keys exchanged: a workgroup owns 256 KEYS
// x 256 output columns (8 waves x 32 keys, K_l fragments and the accumulators dC^T in registers), streams 64-query
// tiles of Q_l, dout and lse_l through the same 3-slot LDS-DMA ring, and flushes / clears its accumulators once per
// sense.  S = Q K^T comes out with lane = key and the queries along the registers, which is the B-operand layout of
// dC^T = dout^T P.  Same persistent job queues (heaviest key tile = tile 0 first), same pinned MFMA / softmax
// interleave in the steps that need no masking.
#include <atomic>

#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"

namespace bp {

template <int KD>
struct MixDcCfg {
    static constexpr int BM = 256, BQ = 64, NB = 8, BNC = 256, NT = 512, NWAVE = 8, NSTAGE = 3;
    static constexpr int QROW = KD <= 4 ? 128 : 256;   // bytes per Q row (power of two, XOR-swizzled)
    static constexpr int QSLOTS = QROW / 16;
    static constexpr int DROW = 512;                    // bytes per dout row (256 columns)
    static constexpr int QTILE = BQ * QROW;
    static constexpr int DTILE = BQ * DROW;
    static constexpr int STATS = NWAVE * 256;           // per wave: lse of the tile's 64 queries (fp32)
    static constexpr int STAGE = QTILE + DTILE + STATS;
    static constexpr int Q_DMA = QTILE / 1024 / NWAVE;  // 1 or 2
    static constexpr int D_DMA = DTILE / 1024 / NWAVE;  // 4
    static constexpr int DMA_PER_STAGE = Q_DMA + D_DMA + 1;
    static constexpr int Q_ROWS_PER_DMA = 1024 / QROW;
    static constexpr int JOB_OFF = NSTAGE * STAGE;
    static constexpr int SMEM = JOB_OFF + 16;
};

template <class ET, int KD, bool FULL>
__global__ __launch_bounds__(512) void sense_mix_dc_kernel(const MixBwdParams p) {
    using C = MixDcCfg<KD>;
    using E = Elem<ET>;
    __shared__ __attribute__((aligned(16))) char smem[C::SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;
    const int S = p.s;
    const float c2 = p.scale_log2e;
    const uint32_t lds0 = lds_base_addr(smem);

    int q_read_off[KD];   // Q fragment (A operand of S = Q K^T): row l31 (+32*kk), logical slot 2*s + hh
#pragma unroll
    for (int s = 0; s < KD; ++s) q_read_off[s] = l31 * C::QROW + (((2 * s + hh) ^ k_swz<C::QROW>(l31)) * 16);
    const int d_row_lane = 4 * hh + ((lane & 15) >> 2);
    const int d_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    static_assert(C::NB == 8, "d_read_off assumes 8 column blocks");
    int d_read_off[4];   // dout^T fragment; block n + 4 sits 256 bytes after block n (see sense_mix_dma.hip)
#pragma unroll
    for (int n = 0; n < 4; ++n) d_read_off[n] = v_lds_off<C::NB>(d_row_lane, n * 4 + d_ch_lane) + (lane & 1) * 8;

    // ---- job queues (as in sense_mix_dma.hip; slot 0 = key tile 0 = the most query tiles) ----------------------
    MixQueues *queues = p.queues;
    uint32_t exhausted = 0;
    const int my_xcd = blockIdx.x & 7;
    auto next_job = [&]() -> int {   // thread 0 only; returns grp * 256 + key tile, or -1
        for (int t = 0; t < 8; ++t) {
            const int q = (my_xcd + t) & 7;
            if (exhausted & (1u << q)) continue;
            const int groups = mix_queue_groups(p.b, p.n_chunks, q);
            const int njobs = groups * p.n_ktiles;
            const int idx = njobs > 0 ? (int)atomicAdd(&queues->ticket[q], 1u) : njobs;
            if (idx < njobs) {
                const int slot = idx / groups;
                const int grp = mix_queue_group(p.n_chunks, q, idx - slot * groups);
                return grp * 256 + slot;
            }
            exhausted |= 1u << q;
        }
        return -1;
    };

    for (;;) {
        __syncthreads();
        if (tid == 0) *reinterpret_cast<int *>(smem + C::JOB_OFF) = next_job();
        __syncthreads();
        const int job = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(smem + C::JOB_OFF));
        if (job < 0) break;
        const int grp = job >> 8, kt = job & 255;
        const int batch = grp / p.n_chunks;
        const int chunk = grp - batch * p.n_chunks;
        const int col_base = chunk * C::BNC;

        const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.qk_bs;
        const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.qk_bs;
        const uint16_t *dg = reinterpret_cast<const uint16_t *>(p.dout) + batch * p.do_bs;

        const int key0 = kt * C::BM + wave * 32;
        const int my_key = key0 + l31;
        const int my_key_clamped = min(my_key, S - 1);
        const bool wave_has_keys = key0 < S;
        const int my_key_sub = key0 >> 5;
        const int nb_live = FULL ? C::NB : min(C::NB, (p.dout_cols - col_base + 31) / 32);

        // query tiles (64 queries) of this job: [qb_begin, nqb); masks needed before qb_clean and in a partial last tile
        const int nqb = (S + C::BQ - 1) / C::BQ;
        const int qb_begin = (kt * C::BM) / C::BQ;
        const int qb_full_end = S / C::BQ;
        const int qb_clean = min(qb_begin + C::BM / C::BQ, nqb);
        const int nq = nqb - qb_begin;

        // Per-lane byte offsets of my DMA pieces inside a tile, rebuilt per job from an opaque copy of the lane index so
        // that the row / column tables are not hoisted to kernel entry and kept alive (and spilled) across the job loop;
        // the partial last tile clamps its rows inside issue(), in a cold branch (see sense_mix_dma.hip).
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int qb_partial = (S % C::BQ) != 0 ? nqb - 1 : -1;
        const int last_row = S - 1 - (nqb - 1) * C::BQ;
        auto q_piece_row = [&](int j) { return (wave * C::Q_DMA + j) * C::Q_ROWS_PER_DMA + lane_o / C::QSLOTS; };
        auto d_piece_row = [&](int j) { return (wave * C::D_DMA + j) * 2 + (lane_o >> 5); };
        uint32_t q_voff[C::Q_DMA], d_voff[C::D_DMA];
#pragma unroll
        for (int j = 0; j < C::Q_DMA; ++j) {
            const int row = q_piece_row(j);
            const int logical = (lane_o % C::QSLOTS) ^ k_swz<C::QROW>(row);
            const int col = logical * 8 < p.dk ? logical * 8 : 0;   // pad slot: a duplicate of column 0 (meets zero K columns)
            q_voff[j] = (uint32_t)(row * p.qk_rs + col) * 2u;
        }
#pragma unroll
        for (int j = 0; j < C::D_DMA; ++j) {
            const int row = d_piece_row(j);
            const int stored = lane_o & 31;
            const int logical = (((stored >> 2) ^ (row & 3)) << 2) | (stored & 3);
            const int col = (FULL || col_base + logical * 8 < p.dout_cols) ? col_base + logical * 8 : col_base;
            d_voff[j] = (uint32_t)(row * p.do_rs + col) * 2u;
        }

        // DMA pieces of the tile two steps ahead, (l2, qb2); its base pointers are carried and advanced on the scalar unit
        // once per step (advance2), as in sense_mix_dma.hip
        int l2 = 0, qb2 = qb_begin;
        const int64_t q_tile_step = (int64_t)C::BQ * p.qk_rs, d_tile_step = (int64_t)C::BQ * p.do_rs;
        const uint16_t *qs2 = qg + (int64_t)qb_begin * q_tile_step;   // first tile of sense l2
        const uint16_t *qt2 = qs2;
        const uint16_t *dt2 = dg + (int64_t)qb_begin * d_tile_step;
        auto issue = [&](int, int, int slot, uint32_t pieces) {
            const uint32_t stage_off = lds0 + slot * C::STAGE;
            if (__builtin_expect(qb2 == qb_partial, 0)) {
#pragma unroll
                for (int j = 0; j < C::Q_DMA; ++j)
                    if ((pieces >> j) & 1u) {
                        const uint32_t back = (uint32_t)(max(q_piece_row(j) - last_row, 0) * p.qk_rs) * 2u;
                        dma16_s(qt2, q_voff[j] - back, __builtin_amdgcn_readfirstlane(stage_off + (wave * C::Q_DMA + j) * 1024));
                    }
#pragma unroll
                for (int j = 0; j < C::D_DMA; ++j)
                    if ((pieces >> (C::Q_DMA + j)) & 1u) {
                        const uint32_t back = (uint32_t)(max(d_piece_row(j) - last_row, 0) * p.do_rs) * 2u;
                        dma16_s(dt2, d_voff[j] - back,
                                __builtin_amdgcn_readfirstlane(stage_off + C::QTILE + (wave * C::D_DMA + j) * 1024));
                    }
            } else {
#pragma unroll
                for (int j = 0; j < C::Q_DMA; ++j)
                    if ((pieces >> j) & 1u) dma16_s(qt2, q_voff[j], stage_off + (wave * C::Q_DMA + j) * 1024);
#pragma unroll
                for (int j = 0; j < C::D_DMA; ++j)
                    if ((pieces >> (C::Q_DMA + j)) & 1u)
                        dma16_s(dt2, d_voff[j], stage_off + C::QTILE + (wave * C::D_DMA + j) * 1024);
            }
            if ((pieces >> (C::Q_DMA + C::D_DMA)) & 1u) {
                // lse of the tile's 64 queries for sense l2: lane i fetches lse[q0 + i] into the wave's own 256-B slot
                const float *src = p.lse + ((int64_t)batch * p.nsenses + l2) * p.lse_stride + min(qb2 * C::BQ + lane, S - 1);
                dma4(src, stage_off + C::QTILE + C::DTILE + wave * 256);
            }
        };
        constexpr uint32_t kAllPieces = (1u << C::DMA_PER_STAGE) - 1u;

        f32x16 acc[C::NB];
        u32x4 kf[KD];
#pragma unroll
        for (int s = 0; s < KD; ++s) kf[s] = u32x4{0u, 0u, 0u, 0u};
        auto take_k = [&](int l) {
            const uint16_t *row = kg + (int64_t)my_key_clamped * p.qk_rs + (int64_t)l * p.qk_ss;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const int col = 16 * s + 8 * hh;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (col < p.dk) v = ld_global_16B(row + col);
                kf[s] = v;
            }
        };

        // S of the 32-query half kk of the tile: rows = queries (registers), column = my key
        auto scores = [&](int stage, int kk) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const u32x4 a = lds_read_16B(smem, q_read_off[s] + stage + kk * 32 * C::QROW);
                st = E::mfma(a, kf[s], st);
            }
            return st;
        };
        // lse (in log2 units) of the queries my registers hold: four runs of four consecutive queries
        auto row_lse = [&](int stage, int kk, float (&l2)[16]) {
            const int base = stage + C::QTILE + C::DTILE + wave * 256 + (kk * 32 + 4 * hh) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 w4 = lds_read_16B(smem, base + g * 32);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t wi = w4[i];   // by-value copy (bp_common.h, as_f32)
                    l2[4 * g + i] = as_f32(wi) * kLog2e;
                }
            }
        };
        auto pack = [&](const f32x16 &st, u32x4 (&pf)[2]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[ks][i] = E::pack2(st[ks * 8 + 2 * i], st[ks * 8 + 2 * i + 1]);
        };
        auto d_operand = [&](int rows, int n) {
            const u32x2 lo = lds_read_tr16_8B(smem, d_read_off[n & 3] + (n >> 2) * 256 + rows);
            const u32x2 hi = lds_read_tr16_8B(smem, d_read_off[n & 3] + (n >> 2) * 256 + rows + 8 * C::DROW);
            return u32x4{lo[0], lo[1], hi[0], hi[1]};
        };
        // dC^T += dout^T P over N consecutive 16-query steps (pk[0..N-1]) from query row `row0` of the tile at `stage`, as
        // one operand stream (mfma_stream, bp_common.h: the dout^T operand of MFMA i + 2 requested before MFMA i)
        auto pv_stream = [&](int stage, int row0, const auto &pk, auto &&mid) {
            constexpr int N = sizeof(pk) / sizeof(pk[0]) * C::NB;
            const int base = stage + C::QTILE + row0 * C::DROW;
            mfma_stream<N>([&](int i) { return d_operand(base + (i >> 3) * 16 * C::DROW, i & 7); },
                           [&](int i, const u32x4 &a) {
                               if (FULL || (i & 7) < nb_live) acc[i & 7] = E::mfma(a, pk[i >> 3], acc[i & 7]);
                               asm volatile("" : "+v"(acc[i & 7]));
                               mid(i);
                           });
        };

        // ---- steady-state step (every query of the tile exists and sees every key of the workgroup), cut as in
        // sense_mix_dma.hip into a vector-heavy half X (S of both query halves as one operand stream, softmax of half 0,
        // the first 8 MFMAs of half 0 with the exponentials of half 1 between them, pack) and a matrix-only half Y (the
        // other 24 MFMAs); the two waves of a SIMD (w, w + 4) run them in anti-phase, X at raised priority.
        u32x4 pfc[3];   // P of half 0 queries 16..31, of half 1 queries 0..15 and 16..31
        auto clean_x = [&](int stage, int slot2, bool dma) {
            __builtin_amdgcn_s_setprio(3);
            u32x4 pf0[2];
            f32x16 st1;
            {
                f32x16 st0;
#pragma unroll
                for (int r = 0; r < 16; ++r) st0[r] = st1[r] = 0.f;
                mfma_stream<2 * KD>(
                    [&](int i) { return lds_read_16B(smem, q_read_off[i >> 1] + stage + (i & 1) * 32 * C::QROW); },
                    [&](int i, const u32x4 &a) {
                        if (i & 1) { st1 = E::mfma(a, kf[i >> 1], st1); asm volatile("" : "+v"(st1)); }
                        else { st0 = E::mfma(a, kf[i >> 1], st0); asm volatile("" : "+v"(st0)); }
                    });
                float lq[16];
                row_lse(stage, 0, lq);
#pragma unroll
                for (int r = 0; r < 16; ++r) st0[r] = fast_exp2(fmaf(st0[r], c2, -lq[r]));
                pack(st0, pf0);
            }
            float lq1[16];
            row_lse(stage, 1, lq1);
            if (dma) issue(0, 0, slot2, 0x03u);
            {
                const int rows = stage + C::QTILE;
                u32x4 a = d_operand(rows, 0);
#pragma unroll
                for (int n = 0; n < C::NB; ++n) {
                    u32x4 a_next = a;
                    if (n + 1 < C::NB) a_next = d_operand(rows, n + 1);
                    asm volatile("" : "+v"(a));
                    acc[n] = E::mfma(a, pf0[0], acc[n]);
                    asm volatile("" : "+v"(acc[n]));
                    float x0 = st1[2 * n], x1 = st1[2 * n + 1];
                    asm volatile("" : "+v"(x0), "+v"(x1));
                    x0 = fast_exp2(fmaf(x0, c2, -lq1[2 * n]));
                    x1 = fast_exp2(fmaf(x1, c2, -lq1[2 * n + 1]));
                    asm volatile("" : "+v"(x0), "+v"(x1));
                    st1[2 * n] = x0;
                    st1[2 * n + 1] = x1;
                    a = a_next;
                }
            }
            if (dma) issue(0, 0, slot2, kAllPieces & ~0x03u);
            pfc[0] = pf0[1];
            u32x4 pf1[2];
            pack(st1, pf1);
            pfc[1] = pf1[0];
            pfc[2] = pf1[1];
            asm volatile("" : "+v"(pfc[0]), "+v"(pfc[1]), "+v"(pfc[2]));   // the packs belong to X, not behind the barrier
            __builtin_amdgcn_s_setprio(0);
        };
        auto clean_y = [&](int stage, int slot2, bool dma) {
            if (dma) issue(0, 0, slot2, 0x03u);
            pv_stream(stage, 16, pfc, [&](int i) {
                if (i == 11 && dma) issue(0, 0, slot2, kAllPieces & ~0x03u);
            });
        };

        // ---- a step in the diagonal region or on the partial last tile: per-half liveness, masking
        auto edge_step = [&](int stage, int qb, int l2, int qb2, int slot2) {
            issue(l2, qb2, slot2, 0x01u);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int qsub = qb * 2 + kk;
                const bool live = wave_has_keys && qsub >= my_key_sub && qsub * 32 < S;
                u32x4 pf[2];
                if (live) {
                    f32x16 st = scores(stage, kk);
                    float lq[16];
                    row_lse(stage, kk, lq);
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = fast_exp2(fmaf(st[r], c2, -lq[r]));
                    pack(st, pf);
                    // query of register r: qsub*32 + (r&3) + 8*(r>>2) + 4*hh.  It must exist (< S) and must not lie
                    // before my key: AND masks on the packed words (AND also kills an inf from an invisible pair).
                    const int q_lim = S - 1 - qsub * 32 - 4 * hh;                   // largest valid rel index (w/o 4hh)
                    const int k_rel = qsub == my_key_sub ? l31 - 4 * hh : -64;      // smallest visible rel index
                    if (qsub == my_key_sub || qsub * 32 + 31 >= S) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int r0 = ks * 8 + 2 * i;
                                const int rel0 = (r0 & 3) + 8 * (r0 >> 2);   // rel of r0 + 1 is rel0 + 1
                                const uint32_t keep = ((rel0 >= k_rel && rel0 <= q_lim) ? 0x0000ffffu : 0u) |
                                                      ((rel0 + 1 >= k_rel && rel0 + 1 <= q_lim) ? 0xffff0000u : 0u);
                                pf[ks][i] &= keep;
                            }
                    }
                    pv_stream(stage, kk * 32, pf, [](int) {});
                }
                issue(l2, qb2, slot2, kk == 0 ? 0x0eu : (kAllPieces & ~0x0fu));
            }
        };

        // ---- pipeline: two tiles in flight (steps past the end re-fetch the last tile, see sense_mix_dma.hip) -----
        auto advance2 = [&]() {
            if (qb2 + 1 < nqb) {
                ++qb2;
                qt2 += q_tile_step;
                dt2 += d_tile_step;
            } else if (l2 + 1 < p.nsenses) {
                ++l2;
                qb2 = qb_begin;
                qs2 += p.qk_ss;
                qt2 = qs2;
                dt2 = dg + (int64_t)qb_begin * d_tile_step;   // dout does not depend on the sense
            }
        };
        issue(0, qb_begin, 0, kAllPieces);
        advance2();
        issue(l2, qb2, 1, kAllPieces);
        advance2();

        int slot = 0;
        auto step_begin = [&]() {
            wait_vmcnt<C::DMA_PER_STAGE>();
            __builtin_amdgcn_s_barrier();
        };
        auto step_end = [&]() {
            slot = slot == 2 ? 0 : slot + 1;
            advance2();
        };
        uint16_t *dcg = reinterpret_cast<uint16_t *>(p.dc) + batch * p.c_bs + (int64_t)my_key * p.c_rs;
        for (int l = 0; l < p.nsenses; ++l) {
            if (wave_has_keys) take_k(l);
#pragma unroll
            for (int n = 0; n < C::NB; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
            for (int qb = qb_begin; qb < qb_clean; ++qb) {
                step_begin();
                edge_step(slot * C::STAGE, qb, l2, qb2, slot >= 1 ? slot - 1 : 2);
                step_end();
            }
            // clean steps, two barriers each; waves 4-7 one barrier late (see sense_mix_dma.hip for the ring argument)
            if (qb_clean < qb_full_end) {
                const bool late = wave >= C::NWAVE / 2;
                if (late) step_begin();
                for (int qb = qb_clean; qb < qb_full_end; ++qb) {
                    step_begin();
                    clean_x(slot * C::STAGE, slot >= 1 ? slot - 1 : 2, late);
                    step_begin();
                    clean_y(slot * C::STAGE, slot >= 1 ? slot - 1 : 2, !late);
                    step_end();
                }
                if (!late) __builtin_amdgcn_s_barrier();
            }
            for (int qb = max(qb_clean, qb_full_end); qb < nqb; ++qb) {
                step_begin();
                edge_step(slot * C::STAGE, qb, l2, qb2, slot >= 1 ? slot - 1 : 2);
                step_end();
            }
            // dC[my key, sense l, chunk columns]
            if (wave_has_keys && my_key < S) {
                uint16_t *og = dcg + (int64_t)l * p.c_ss;
#pragma unroll
                for (int n = 0; n < C::NB; ++n)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = col_base + n * 32 + 8 * g + 4 * hh;
                        if (col < p.dout_cols) {
                            u32x2 w = {E::pack2(acc[n][4 * g + 0], acc[n][4 * g + 1]),
                                       E::pack2(acc[n][4 * g + 2], acc[n][4 * g + 3])};
                            // (plain stores: 8-byte per-lane pieces marked non-temporal cost 1.6 -> 4.6 ms, r02_p)
                            *reinterpret_cast<u32x2 *>(og + col) = w;
                        }
                    }
            }
        }
        (void)nq;
        wait_vmcnt<0>();
    }

}

// =====================================================================================================================
// dq and dk from the precomputed dP^T slab.
//
// dpt: (B, N, TS) 16-bit, row index s*k + l (key s, sense l), TS = 128 queries of the slab [t0, t0 + TS): the output of
// the BLAS GEMM  C_flat[:, :N, :] @ dout[:, t0:t0+TS, :]^T  with N = (number of keys the slab can see) * k.
// Both kernels recompute P from q, k and the saved lse; one workgroup = one (batch, sense), 4 waves.
//   * sense_dq_kernel: wave = 32 queries of the slab, sweeps 64-key tiles (K_l rows + their dP^T rows through LDS;
//     dP^T is read with ds_read_b64_tr_b16 so that it lands in the register order of S^T = K Q^T).  D is not known
//     until the sweep ends, so the sweep accumulates  A1 = sum_s P dP k_s,  A2 = sum_s P k_s  and  D = sum_s P dP,
//     and the epilogue forms dq = scale (A1 - D A2).  D goes to memory for the dk kernel.
//   * sense_dk_kernel: wave = 32 keys (K_l fragments in registers), sweeps the slab's queries; lane = key, so its row
//     of dP^T is read directly; dS = P (dP - D) feeds dK^T += Q^T dS.  The result is ADDED to an fp32 accumulator:
//     slabs run one after the other on one stream and each (batch, sense, key) is owned by exactly one wave of one
//     launch, so the sum is deterministic.
// =====================================================================================================================
template <int KD>
struct SenseGradCfg {
    static constexpr int NT = 256, NWAVE = 4, TS = 128, BK = 64;
    static constexpr int KROW = KD <= 4 ? 128 : 256;   // K / Q row image pitch (b128 reads)
    static constexpr int KSLOTS = KROW / 16;
    static constexpr int NVK = KD <= 2 ? 1 : KD <= 4 ? 2 : 4;   // 64-byte chunks per row of the transposed-read image
    static constexpr int TROW = NVK * 64;
};

// ---- dq ----------------------------------------------------------------------------------------------------
template <class ET, int KD>
__global__ __launch_bounds__(256) void sense_dq_kernel(const SenseGradParams p) {
    using C = SenseGradCfg<KD>;
    using E = Elem<ET>;
    constexpr int NVK = C::NVK;
    // stage = K row image | K transposed-read image | dP^T tile (64 keys x 128 queries, 256-B rows)
    constexpr int RTILE = C::BK * C::KROW, TTILE = C::BK * C::TROW, PTILE = C::BK * 256;
    constexpr int STAGE = RTILE + TTILE + PTILE;
    constexpr int R_DMA = RTILE / 1024 / C::NWAVE, T_DMA = TTILE / 1024 / C::NWAVE, P_DMA = PTILE / 1024 / C::NWAVE;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const uint32_t lds0 = lds_base_addr(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;
    const int batch = blockIdx.x / p.nsenses;
    const int l = blockIdx.x - batch * p.nsenses;
    const int S = p.s;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const uint16_t *pg = reinterpret_cast<const uint16_t *>(p.dpt) + batch * p.dpt_bs;   // row (s*k + l), 128 queries

    const int q0 = p.t0 + wave * 32;
    const int my_q = q0 + l31;
    const bool wave_has_rows = q0 < S;
    const float c2 = p.scale * kLog2e;
    const int k_end = min(S, p.t0 + C::TS);
    const int nkb = (k_end + C::BK - 1) / C::BK;

    if (p.dk * 2 != C::KROW || p.dk * 2 != C::TROW) {   // pad slots of the images must read as 0
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < 2 * STAGE; off += C::NT * 16) lds_write_16B(smem, off, z);
        __syncthreads();
    }

    u32x4 qf[KD];
    float lse2 = 0.f;
    {
        const int q = min(my_q, S - 1);
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 a = {0u, 0u, 0u, 0u};
            if (col < p.dk) a = ld_global_16B(qg + (int64_t)q * p.qk_rs + col);
            qf[s] = a;
        }
        lse2 = p.lse[((int64_t)batch * p.nsenses + l) * p.lse_stride + q] * kLog2e;
#pragma unroll
        for (int s = 0; s < KD; ++s) settle(qf[s]);
        settle(lse2);
    }

    // DMA descriptors: K row image, K transposed-read image, dP^T rows
    int rr[R_DMA], rc[R_DMA], tr[T_DMA], tc[T_DMA], pr[P_DMA], pc[P_DMA];
#pragma unroll
    for (int j = 0; j < R_DMA; ++j) {
        rr[j] = (wave * R_DMA + j) * (1024 / C::KROW) + lane / C::KSLOTS;
        rc[j] = ((lane % C::KSLOTS) ^ k_swz<C::KROW>(rr[j])) * 8;
    }
#pragma unroll
    for (int j = 0; j < T_DMA; ++j) {
        const int c = (wave * T_DMA + j) * 64 + lane;
        tr[j] = c / (NVK * 4);
        const int stored = c - tr[j] * (NVK * 4);
        int c64 = stored >> 2;
        if (NVK == 2) c64 ^= (tr[j] >> 1) & 1;
        if (NVK == 4) c64 ^= tr[j] & 3;
        tc[j] = ((c64 << 2) | (stored & 3)) * 8;
    }
#pragma unroll
    for (int j = 0; j < P_DMA; ++j) {
        const int c = (wave * P_DMA + j) * 64 + lane;   // 16 chunks of 16 B per 256-B row
        pr[j] = c >> 4;
        const int stored = c & 15;
        const int c64 = (stored >> 2) ^ (pr[j] & 3);
        pc[j] = ((c64 << 2) | (stored & 3)) * 8;
    }
    auto issue = [&](int kb) {
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + (kb & 1) * STAGE);
#pragma unroll
        for (int j = 0; j < R_DMA; ++j) {
            const int64_t row = min(kb * C::BK + rr[j], S - 1);
            if (rc[j] < p.dk) dma16_d(kg + row * p.qk_rs + rc[j], st + (wave * R_DMA + j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < T_DMA; ++j) {
            const int64_t row = min(kb * C::BK + tr[j], S - 1);
            if (tc[j] < p.dk) dma16_d(kg + row * p.qk_rs + tc[j], st + RTILE + (wave * T_DMA + j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < P_DMA; ++j) {
            const int64_t key = min(kb * C::BK + pr[j], k_end - 1);
            dma16_d(pg + (key * p.nsenses + l) * (int64_t)C::TS + pc[j], st + RTILE + TTILE + (wave * P_DMA + j) * 1024);
        }
    };

    // accumulators: A1 = sum_s (P dP) k_s and A2 = sum_s P k_s as (d_k x my queries), D = sum_s P dP
    f32x16 a1[NVK], a2[NVK];
#pragma unroll
    for (int n = 0; n < NVK; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { a1[n][r] = 0.f; a2[n][r] = 0.f; }
    float dsum = 0.f;

    int r_read_off[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) r_read_off[s] = l31 * C::KROW + (((2 * s + hh) ^ k_swz<C::KROW>(l31)) * 16);
    const int t_row_lane = 4 * hh + ((lane & 15) >> 2);
    const int t_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    int t_read_off[NVK];
#pragma unroll
    for (int n = 0; n < NVK; ++n) t_read_off[n] = v_lds_off<NVK>(t_row_lane, n * 4 + t_ch_lane) + (lane & 1) * 8;
    // dP^T tile: 256-B rows = 4 chunks of 64 B (one per wave's 32 queries), chunk index swizzled with row & 3
    const int p_read_off = v_lds_off<4>(t_row_lane, wave * 4 + t_ch_lane) + (lane & 1) * 8;

    // Row reference r[t] ~ D[t], estimated from the first 32 keys as sum P dP / sum P (key 0 is visible to every query,
    // so the denominator is positive): the softmax backward is invariant under dP -> dP - r[t], and a component of
    // dout . C that is common to all keys of a row -- the bias of the sense network's last layer is one -- would
    // otherwise survive into the 16-bit products P dP and cancel only in A1 - D A2
    // (tests: ..._with_a_common_offset_in_dout_c).  With g = P (dP - r):
    //     D = sum g + r sum P,      dq = scale (sum g k - (sum g + r (sum P - 1)) sum P k).
    float rref = 0.f, psum = 0.f;
    // one 32-key sub-block; REF: only sum P dP and sum P over it (into dsum / psum), no products
    auto sub_block = [&](const char *k_r, const char *k_t, const char *dp_t, int kb, int kk, auto REF) {
        constexpr bool kRef = decltype(REF)::value;
        const int kbase = kb * C::BK + kk * 32;
        f32x16 st_;
#pragma unroll
        for (int r = 0; r < 16; ++r) st_[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const u32x4 a = lds_read_16B(k_r, r_read_off[s] + kk * 32 * C::KROW);
            st_ = E::mfma(a, qf[s], st_);
        }
        u32x4 pf[2], gf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // dP^T of keys {0..3, 8..11} + 4 hh + 16 ks for my query: same order as the S^T registers
            const int rows = (kk * 32 + ks * 16) * 256;
            const u32x2 lo = lds_read_tr16_8B(dp_t, p_read_off + rows);
            const u32x2 hi = lds_read_tr16_8B(dp_t, p_read_off + rows + 8 * 256);
            const uint32_t dw[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = ks * 8 + 2 * i;
                const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float p0 = fast_exp2(fmaf(st_[r], c2, -lse2));
                float p1 = fast_exp2(fmaf(st_[r + 1], c2, -lse2));
                if (key > my_q || key >= S) p0 = 0.f;
                if (key + 1 > my_q || key + 1 >= S) p1 = 0.f;
                const uint32_t d = dw[i];   // by-value copy (bp_common.h)
                const float g0 = p0 * (E::lo_f32(d) - rref), g1 = p1 * (E::hi_f32(d) - rref);
                dsum += g0 + g1;
                psum += p0 + p1;
                pf[ks][i] = E::pack2(p0, p1);
                gf[ks][i] = E::pack2(g0, g1);
            }
        }
        if (kRef) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rows = (kk * 32 + ks * 16) * C::TROW;
#pragma unroll
            for (int n = 0; n < NVK; ++n) {
                const u32x2 lo = lds_read_tr16_8B(k_t, t_read_off[n] + rows);
                const u32x2 hi = lds_read_tr16_8B(k_t, t_read_off[n] + rows + 8 * C::TROW);
                const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
                a1[n] = E::mfma(a, gf[ks], a1[n]);
                a2[n] = E::mfma(a, pf[ks], a2[n]);
            }
        }
    };
    if (nkb > 0) issue(0);
    for (int kb = 0; kb < nkb; ++kb) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kb + 1 < nkb) issue(kb + 1);
        if (!wave_has_rows || kb * C::BK > q0 + 31) continue;
        const char *st = smem + (kb & 1) * STAGE;
        const char *k_r = st, *k_t = st + RTILE, *dp_t = st + RTILE + TTILE;
        if (kb == 0) {
            sub_block(k_r, k_t, dp_t, 0, 0, std::true_type{});
            const float num = xhalf_sum(dsum), den = xhalf_sum(psum);
            rref = den > 0.f ? num / den : 0.f;
            // the slab holds 16-bit values: a reference of the same precision keeps dP - r exact for dP near r
            rref = E::lo_f32(E::pack2(rref, 0.f));
            dsum = 0.f;
            psum = 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kbase = kb * C::BK + kk * 32;
            if (kbase >= S || kbase > q0 + 31) continue;
            sub_block(k_r, k_t, dp_t, kb, kk, std::false_type{});
        }
    }

    if (!wave_has_rows) return;
    const float g_tot = xhalf_sum(dsum), p_tot = xhalf_sum(psum);
    const float d_tot = fmaf(rref, p_tot - 1.f, g_tot);    // D - r: what multiplies sum P k
    if (my_q < S) {
        if (hh == 0) p.dsum[((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q] = fmaf(rref, p_tot, g_tot);   // D itself
        uint16_t *dqg = reinterpret_cast<uint16_t *>(p.dq) + batch * p.dq_bs + (int64_t)my_q * p.dq_rs + (int64_t)l * p.dq_ss;
#pragma unroll
        for (int n = 0; n < NVK; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = n * 32 + 8 * g + 4 * hh;
                if (d0 < p.dk) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = p.scale * (a1[n][4 * g + i] - d_tot * a2[n][4 * g + i]);
                    u32x2 w = {E::pack2(v[0], v[1]), E::pack2(v[2], v[3])};
                    *reinterpret_cast<u32x2 *>(dqg + d0) = w;
                }
            }
    }
}

// ---- dk ----------------------------------------------------------------------------------------------------
template <class ET, int KD>
__global__ __launch_bounds__(256) void sense_dk_kernel(const SenseGradParams p) {
    using C = SenseGradCfg<KD>;
    using E = Elem<ET>;
    constexpr int NVK = C::NVK;
    // the slab's 128 queries of this (batch, sense): Q row image | Q transposed-read image | lse, D
    constexpr int RTILE = C::TS * C::KROW, TTILE = C::TS * C::TROW, STATS = 2 * C::TS * 4;
    constexpr int R_DMA = RTILE / 1024 / C::NWAVE, T_DMA = TTILE / 1024 / C::NWAVE;
    __shared__ __attribute__((aligned(16))) char smem[RTILE + TTILE + STATS];
    const uint32_t lds0 = lds_base_addr(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;
    const int per_bl = (min(p.s, p.t0 + C::TS) + 127) / 128;           // 128-key tiles this slab can see
    const int bl = blockIdx.x / per_bl, ktile = blockIdx.x - bl * per_bl;
    const int batch = bl / p.nsenses;
    const int l = bl - batch * p.nsenses;
    const int S = p.s;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const uint16_t *pg = reinterpret_cast<const uint16_t *>(p.dpt) + batch * p.dpt_bs;
    const float *lse_g = p.lse + ((int64_t)batch * p.nsenses + l) * p.lse_stride;
    const float *dsum_g = p.dsum + ((int64_t)batch * p.nsenses + l) * p.lse_stride;

    const int key0 = ktile * 128 + wave * 32;
    const int my_key = key0 + l31;
    const int k_end = min(S, p.t0 + C::TS);
    const bool wave_has_keys = key0 < k_end;
    const float c2 = p.scale * kLog2e;
    const int nq = min(C::TS, S - p.t0);   // queries that exist in the slab

    {   // pad slots of the images must read as 0 (also rows past the sequence)
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < RTILE + TTILE + STATS; off += C::NT * 16) lds_write_16B(smem, off, z);
        __syncthreads();
    }
    // ---- the slab's Q rows (two images) and per-query statistics: one DMA round ---------------------------------
#pragma unroll
    for (int j = 0; j < R_DMA; ++j) {
        const int row = (wave * R_DMA + j) * (1024 / C::KROW) + lane / C::KSLOTS;
        const int col = ((lane % C::KSLOTS) ^ k_swz<C::KROW>(row)) * 8;
        const int64_t q = min(p.t0 + row, S - 1);
        if (col < p.dk) dma16_d(qg + q * p.qk_rs + col, lds0 + (wave * R_DMA + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < T_DMA; ++j) {
        const int c = (wave * T_DMA + j) * 64 + lane;
        const int row = c / (NVK * 4), stored = c - row * (NVK * 4);
        int c64 = stored >> 2;
        if (NVK == 2) c64 ^= (row >> 1) & 1;
        if (NVK == 4) c64 ^= row & 3;
        const int col = ((c64 << 2) | (stored & 3)) * 8;
        const int64_t q = min(p.t0 + row, S - 1);
        if (col < p.dk) dma16_d(qg + q * p.qk_rs + col, lds0 + RTILE + (wave * T_DMA + j) * 1024);
    }
    if (wave < 2) {   // wave 0: lse of the 128 queries (two 64-float pieces), wave 1: D
        const float *src = wave == 0 ? lse_g : dsum_g;
        dma4(src + min(p.t0 + lane, S - 1), lds0 + RTILE + TTILE + wave * 512);
        dma4(src + min(p.t0 + 64 + lane, S - 1), lds0 + RTILE + TTILE + wave * 512 + 256);
    }

    u32x4 kf[KD];
    {
        const int key = min(my_key, S - 1);
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 a = {0u, 0u, 0u, 0u};
            if (col < p.dk) a = ld_global_16B(kg + (int64_t)key * p.qk_rs + col);
            kf[s] = a;
        }
    }
    // my row of dP^T: 128 queries = 256 bytes, in the register order of S (runs of 4 queries)
    u32x2 dprow[4][4];   // [32-query block][g]
    {
        const uint16_t *row = pg + ((int64_t)min(my_key, k_end - 1) * p.nsenses + l) * (int64_t)C::TS;
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
#pragma unroll
            for (int g = 0; g < 4; ++g) dprow[qb][g] = *reinterpret_cast<const u32x2 *>(row + qb * 32 + 8 * g + 4 * hh);
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (!wave_has_keys) return;

    int r_read_off[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) r_read_off[s] = l31 * C::KROW + (((2 * s + hh) ^ k_swz<C::KROW>(l31)) * 16);
    const int t_row_lane = 4 * hh + ((lane & 15) >> 2);
    const int t_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    int t_read_off[NVK];
#pragma unroll
    for (int n = 0; n < NVK; ++n) t_read_off[n] = v_lds_off<NVK>(t_row_lane, n * 4 + t_ch_lane) + (lane & 1) * 8;
    const char *q_r = smem, *q_t = smem + RTILE, *stats = smem + RTILE + TTILE;

    f32x16 dk[NVK];
#pragma unroll
    for (int n = 0; n < NVK; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[n][r] = 0.f;

#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        const int qbase = p.t0 + qb * 32;
        if (qb * 32 >= nq || qbase + 31 < key0) continue;   // no such queries, or all of them before my first key
        f32x16 s_;
#pragma unroll
        for (int r = 0; r < 16; ++r) s_[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const u32x4 a = lds_read_16B(q_r, r_read_off[s] + qb * 32 * C::KROW);
            s_ = E::mfma(a, kf[s], s_);
        }
        u32x4 dsf[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x4 l4 = lds_read_16B(stats, (qb * 32 + 8 * g + 4 * hh) * 4);
            const u32x4 d4 = lds_read_16B(stats, 512 + (qb * 32 + 8 * g + 4 * hh) * 4);
            const u32x2 dp2 = dprow[qb][g];
            const uint32_t dpw[2] = {dp2[0], dp2[1]};
            float de[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const int q = qbase + 8 * g + 4 * hh + i;
                const uint32_t lw = l4[i], dw = d4[i], pw = dpw[i >> 1];   // by-value copies (bp_common.h)
                const float pv = fast_exp2(fmaf(s_[r], c2, -as_f32(lw) * kLog2e));
                const float dp = (i & 1) ? E::hi_f32(pw) : E::lo_f32(pw);
                const bool dead = q >= S || my_key >= S || my_key > q;
                de[i] = dead ? 0.f : pv * (dp - as_f32(dw));
            }
            dsf[g >> 1][(g & 1) * 2 + 0] = E::pack2(de[0], de[1]);
            dsf[g >> 1][(g & 1) * 2 + 1] = E::pack2(de[2], de[3]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rows = (qb * 32 + ks * 16) * C::TROW;
#pragma unroll
            for (int n = 0; n < NVK; ++n) {
                const u32x2 lo = lds_read_tr16_8B(q_t, t_read_off[n] + rows);
                const u32x2 hi = lds_read_tr16_8B(q_t, t_read_off[n] + rows + 8 * C::TROW);
                dk[n] = E::mfma(u32x4{lo[0], lo[1], hi[0], hi[1]}, dsf[ks], dk[n]);
            }
        }
    }

    if (my_key >= S) return;
    float *acc = p.dk_acc + batch * p.dka_bs + (int64_t)my_key * p.dka_rs + (int64_t)l * p.dka_ss;
#pragma unroll
    for (int n = 0; n < NVK; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = n * 32 + 8 * g + 4 * hh;
            if (d0 < p.dk) {
                float *dst = acc + d0;
                const f32x4 old = *reinterpret_cast<const f32x4 *>(dst);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = old[i] + p.scale * dk[n][4 * g + i];
                *reinterpret_cast<f32x4 *>(dst) = v;
            }
        }
}

// ---- launchers -------------------------------------------------------------------------------------------------
static int persistent_grid() {
    thread_local int cached_dev = -1, cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached_dev = dev;
    }
    return cus;
}

template <class ET, int KD>
static hipError_t launch_dc_kd(MixBwdParams p, hipStream_t stream) {
    const hipError_t armed = arm_mix_queues(p.queues, stream);   // sense_mix_dma.hip
    if (armed != hipSuccess) return armed;
    const int njobs = p.b * p.n_chunks * p.n_ktiles;
    const int cus = persistent_grid();
    dim3 g(njobs < cus ? njobs : cus), t(512);
    if (p.dout_cols % 256 == 0) hipLaunchKernelGGL((sense_mix_dc_kernel<ET, KD, true>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((sense_mix_dc_kernel<ET, KD, false>), g, t, 0, stream, p);
    return hipGetLastError();
}

template <class ET>
static hipError_t launch_dc_et(const MixBwdParams &p, hipStream_t stream) {
    switch ((p.dk + 15) / 16) {
        case 1: return launch_dc_kd<ET, 1>(p, stream);
        case 2: return launch_dc_kd<ET, 2>(p, stream);
        case 3: return launch_dc_kd<ET, 3>(p, stream);
        case 4: return launch_dc_kd<ET, 4>(p, stream);
        case 5: return launch_dc_kd<ET, 5>(p, stream);
        case 6: return launch_dc_kd<ET, 6>(p, stream);
        case 7: return launch_dc_kd<ET, 7>(p, stream);
        default: return launch_dc_kd<ET, 8>(p, stream);
    }
}

hipError_t launch_sense_mix_dc(const MixBwdParams &p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_dc_et<BF16>(p, stream) : launch_dc_et<F16>(p, stream);
}

template <class ET, int KD>
static hipError_t launch_grad_kd(const SenseGradParams &p, hipStream_t stream) {
    hipLaunchKernelGGL((sense_dq_kernel<ET, KD>), dim3(p.b * p.nsenses), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int per_bl = ((p.s < p.t0 + 128 ? p.s : p.t0 + 128) + 127) / 128;
    hipLaunchKernelGGL((sense_dk_kernel<ET, KD>), dim3(p.b * p.nsenses * per_bl), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <class ET>
static hipError_t launch_grad_et(const SenseGradParams &p, hipStream_t stream) {
    switch ((p.dk + 15) / 16) {
        case 1: return launch_grad_kd<ET, 1>(p, stream);
        case 2: return launch_grad_kd<ET, 2>(p, stream);
        case 3: return launch_grad_kd<ET, 3>(p, stream);
        case 4: return launch_grad_kd<ET, 4>(p, stream);
        case 5: return launch_grad_kd<ET, 5>(p, stream);
        case 6: return launch_grad_kd<ET, 6>(p, stream);
        case 7: return launch_grad_kd<ET, 7>(p, stream);
        default: return launch_grad_kd<ET, 8>(p, stream);
    }
}

hipError_t launch_sense_dq_dk(const SenseGradParams &p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_grad_et<BF16>(p, stream) : launch_grad_et<F16>(p, stream);
}

}  // namespace bp
