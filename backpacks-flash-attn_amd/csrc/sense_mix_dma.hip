// Fused Backpack sense combination, LDS-DMA ring version (the fast path for 16-byte-aligned shapes).
//
//     out[b,t,:] = sum_l sum_{s<=t} exp(scale * q_l[t].k_l[s] - lse[b,l,t]) * C[b,s,l,:]
//
// Same contraction and tile algebra as sense_mix.hip (see there and bp_common.h); what changes is how
// the K_l / C_l tiles reach LDS.  The register-staged kernel prefetches ONE 32-key tile ahead and is
// latency-bound (rocprof r01_a: 61 % of wave cycles in s_waitcnt, L2 hit rate 15 %: four query
// tiles sweep the same C at different speeds, so most tiles come from HBM / Infinity Cache).  Here
//   * tiles are 64 keys (40 KB: 32 KB of C + 8 KB of K) in a 3-slot LDS ring (120 KB, one workgroup
//     per CU), filled by `global_load_lds_dwordx4` (no VGPR round trip, no ds_write pass);
//   * two tiles are always in flight (>= 2 us of HBM latency covered by >= 76 MFMAs per SIMD);
//   * waits are COUNTED: each wave issues exactly DMA_PER_STAGE DMA instructions per tile, so
//     `s_waitcnt vmcnt(DMA_PER_STAGE)` = "my share of the oldest tile has landed", then ONE raw
//     `s_barrier` per tile makes every wave's share visible and retires the slot read last step;
//   * the DMA writes LDS linearly (wave base + lane*16), so the XOR swizzles that make ds_read_b128
//     (K) and ds_read_b64_tr_b16 (C) conflict-free are applied to the per-lane SOURCE address.
// Default loop order: sense outer, 64-key tile inner; blocks dispatched heaviest query tiles first.
// Alternative kept behind -DBP_MIX_SUPER=1 (+ BP_MIX_ORDER=lockstep): key SUPER-tile (256 keys) outer,
// sense middle, 64-key tile inner, for KD <= 4.  MEASURED (r01_d, B=64 S=1024 k=16 d=768): it does what
// it was built for -- HBM traffic 5.7 GB -> 2.1 GB per launch, L2 hit rate 18 % -> 74 % -- and is still
// SLOWER: 1.82 ms (lockstep groups) / 1.49 ms (heaviest-first) against 1.39 ms for this default.  The
// LDS-DMA fill itself reaches 130 GB/s per CU out of L2 but only 25 GB/s per CU (6.4 TB/s chip) out of
// HBM (scripts/probes/dma_rate.hip), the default order needs ~0.9 ms of pure HBM time per launch, and
// the compute stream alone takes 1.0 ms ("no DMA" ablation) -- yet whole-group dispatch loses more to
// its tail and to all CUs of an XCD pulling the same lines at once than the saved traffic returns.
// How the alternative works:
// key SUPER-tile (256 keys) outer, sense middle, 64-key tile inner.
// The query tiles of one (batch, column chunk) group then walk C in the same order at the same pace
// (a step costs the same for every tile; tile t merely stops after super-tile t), so when they are
// co-resident on one XCD the group pulls each C tile from HBM once and the others hit L2 -- with the
// old sense-outer order the four tiles swept C at different speeds and re-streamed it 2.5x
// (rocprof r01_c: 5.98 GB per launch against 1.91 GB algorithmic).  Memory locality improves too: one
// super-tile is a contiguous 6 MB slab of the (B,S,k,d) buffer.  The price is 4x more sense switches;
// the per-sense operands of a wave (its 32 query fragments and their log-sum-exp) therefore arrive
// through a per-wave LDS "mailbox" filled by the same DMA queue one step ahead (the lane that DMAs
// a fragment is the lane that reads it back), so a switch costs KD+1 ds_reads and no vmcnt drain.
// Rows past the sequence are fetched from a clamped (valid) row: their probabilities are exactly 0
// by the causal mask and 0 * finite = 0; LDS is zeroed once so never-written pad slots are 0.
#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"

namespace bp {

template <int KD, bool WEIGHTED = false>
struct MixDmaCfg {
    static constexpr int BM = 256, BK = 64, NB = 8, BNC = 256, NT = 512, NWAVE = 8, NSTAGE = 3;
    static constexpr int KROW = KD <= 4 ? 128 : 256;   // bytes per K row (power of two, XOR-swizzled)
    static constexpr int KSLOTS = KROW / 16;
    static constexpr int CROW = 512;
    static constexpr int KTILE = BK * KROW;
    static constexpr int CTILE = BK * CROW;
    static constexpr int WTILE = WEIGHTED ? NWAVE * 256 : 0;   // per wave: the tile's 64 key weights (fp32)
    static constexpr int STAGE = KTILE + CTILE + WTILE;
    static constexpr int K_DMA = KTILE / 1024 / NWAVE;   // DMA instructions per wave per tile (1 or 2)
    static constexpr int C_DMA = CTILE / 1024 / NWAVE;   // 4
    static constexpr int DMA_PER_STAGE = K_DMA + C_DMA + (WEIGHTED ? 1 : 0);
    static constexpr int K_ROWS_PER_DMA = 1024 / KROW;   // 8 or 4
#ifndef BP_MIX_SUPER
#define BP_MIX_SUPER 0   // measured slower on MI355X (see the header comment); kept as an A/B build switch
#endif
    static constexpr bool SUPER = BP_MIX_SUPER && KD <= 4;   // super-tile loop order + Q mailbox (LDS budget)
    static constexpr int SUP = BM / BK;                  // key tiles per super-tile
    static constexpr int QBOX_WAVE = KD * 1024 + 256;    // KD fragments (64 lanes x 16 B) + 64 x lse
    static constexpr int QBOX_OFF = NSTAGE * STAGE;
    static constexpr int SMEM = QBOX_OFF + (SUPER ? NWAVE * QBOX_WAVE : 0);
};

// Position of a pipeline step: super-tile, sense, key tile inside the super-tile.
struct MixCursor {
    int st, l, kk;
};

template <class ET, int KD, bool FULL, bool WEIGHTED>
__global__ __launch_bounds__(512) void sense_mix_dma_kernel(const MixParams p) {
    using C = MixDmaCfg<KD, WEIGHTED>;
    using E = Elem<ET>;
    __shared__ __attribute__((aligned(16))) char smem[C::SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    int grp, slot;
    if (p.order >= 2) {
        // Per XCD: whole (batch, chunk) groups, heaviest tile first inside a group, the column chunks of
        // one batch next to each other (same K tiles, adjacent pieces of the same C rows).  The query
        // tiles of a group then start together and walk C in lockstep -> one HBM read per group, the rest
        // L2 hits.  In-order dispatch of whole groups leaves a long tail though (the last group's
        // heaviest tile starts last), so with order 3 only the first 2/3 of an XCD's groups go out this
        // way and the rest heaviest-tiles-first: same makespan as pure heaviest-first in a list-scheduling
        // model (8.0 vs the 7.5 ideal for 24 groups on 32 CUs; pure groups: 10.0), ~40 % less C traffic.
        const int xcd = blockIdx.x & 7;
        const int s8 = blockIdx.x >> 3;
        const int gpx = ((p.b + 7) / 8) * p.n_chunks;           // groups per XCD
        const int nq = p.order == 2 ? gpx : (2 * gpx) / 3;      // dispatched as whole groups
        int j;
        if (s8 < nq * p.n_qtiles) {
            j = s8 / p.n_qtiles;
            slot = s8 - j * p.n_qtiles;
        } else {
            const int r = s8 - nq * p.n_qtiles, rest = gpx - nq;
            slot = r / rest;
            j = nq + r - slot * rest;
        }
        const int bb = (j / p.n_chunks) * 8 + xcd;
        if (bb >= p.b) return;
        grp = bb * p.n_chunks + (j % p.n_chunks);
    } else if (p.order == 0) {
        if (!xcd_map(blockIdx.x, p.b * p.n_chunks, p.n_qtiles, grp, slot)) return;
    } else {
        // heaviest query tiles of every group first (list scheduling with the longest jobs first),
        // a group still always lands on the same XCD
        const int ngroups = p.b * p.n_chunks;
        const int per_xcd = (ngroups + 7) / 8;
        const int s8 = blockIdx.x >> 3;
        slot = s8 / per_xcd;
        grp = (s8 - slot * per_xcd) * 8 + (blockIdx.x & 7);
        if (grp >= ngroups) return;
    }
    const int qt = p.n_qtiles - 1 - slot;
    const int batch = grp / p.n_chunks;
    const int chunk = grp - batch * p.n_chunks;
    const int col_base = chunk * C::BNC;
    const int S = p.s;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.qk_bs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.qk_bs;
    const uint16_t *cg = reinterpret_cast<const uint16_t *>(p.c) + batch * p.c_bs;

    const int k_end = min(S, qt * C::BM + C::BM);
    const int nkb = (k_end + C::BK - 1) / C::BK;
    const int nsteps = p.nsenses * nkb;
    // steps are ordered (super-tile, sense, tile); without SUPER there is one super-tile of nkb tiles
    const int sup = C::SUPER ? C::SUP : nkb;
    const int n_super = (nkb + sup - 1) / sup;
    const int nkb_last = nkb - sup * (n_super - 1);
    auto advance = [&](MixCursor &c) {
        if (++c.kk == (c.st + 1 < n_super ? sup : nkb_last)) {
            c.kk = 0;
            if (++c.l == p.nsenses) { c.l = 0; ++c.st; }
        }
    };

    const int q0 = qt * C::BM + wave * 32;
    const int my_q = q0 + l31;
    const int my_q_clamped = min(my_q, S - 1);
    const bool wave_has_rows = q0 < S;
    const int my_diag_sub = q0 >> 5;   // index of the 32-key sub-block that holds my diagonal
    const float c2 = p.scale_log2e;
    const int nb_live = FULL ? C::NB : min(C::NB, (p.dout - col_base + 31) / 32);

    // ---- zero the ring once: pad slots that no DMA ever writes must read as 0 -----------------------
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < C::SMEM; off += C::NT * 16) lds_write_16B(smem, off, z);
    }
    __syncthreads();

    // ---- per-lane DMA source descriptors (tile-invariant parts) ------------------------------------
    // K piece j of this wave: rows (wave*K_DMA + j)*RPD + lane/KSLOTS, stored slot lane%KSLOTS
    int k_row[C::K_DMA], k_col[C::K_DMA];
    bool k_on[C::K_DMA];
#pragma unroll
    for (int j = 0; j < C::K_DMA; ++j) {
        const int row = (wave * C::K_DMA + j) * C::K_ROWS_PER_DMA + lane / C::KSLOTS;
        const int logical = (lane % C::KSLOTS) ^ k_swz<C::KROW>(row);
        k_row[j] = row;
        k_col[j] = logical * 8;
        k_on[j] = logical * 8 < p.dk;
    }
    // C piece j of this wave: rows (wave*C_DMA + j)*2 + lane/32, stored chunk lane%32
    int c_row[C::C_DMA], c_col[C::C_DMA];
    bool c_on[C::C_DMA];
#pragma unroll
    for (int j = 0; j < C::C_DMA; ++j) {
        const int row = (wave * C::C_DMA + j) * 2 + (lane >> 5);
        const int stored = lane & 31;
        const int logical = (((stored >> 2) ^ (row & 3)) << 2) | (stored & 3);
        c_row[j] = row;
        c_col[j] = col_base + logical * 8;
        c_on[j] = c_col[j] < p.dout;
    }

    const uint32_t lds0 = lds_base_addr(smem);
    // DMA piece `j` (0 .. DMA_PER_STAGE-1: the K pieces, then the C pieces) of pipeline step `step`
    auto issue_piece = [&](int step, const MixCursor &c, int j) {
        const int l = c.l;
        const int kb = c.st * sup + c.kk;
        const uint32_t stage_off = lds0 + (step % C::NSTAGE) * C::STAGE;
        if (j < C::K_DMA) {
            const int key = min(kb * C::BK + k_row[j], S - 1);
            const uint16_t *src = kg + (int64_t)l * p.qk_ss + (int64_t)key * p.qk_rs + k_col[j];
            if (k_on[j]) dma16_d(src, stage_off + (wave * C::K_DMA + j) * 1024);
        } else if (WEIGHTED && j == C::K_DMA + C::C_DMA) {
            // key weights of this (sense, tile): lane i fetches w[key0 + i] into the wave's own 256-B slot
            const float *src = p.kw + batch * p.kw_bs + (int64_t)l * p.kw_ss + min(kb * C::BK + lane, S - 1);
            dma4(src, stage_off + C::KTILE + C::CTILE + wave * 256);
        } else {
            const int jc = j - C::K_DMA;
            const int key = min(kb * C::BK + c_row[jc], S - 1);
            const uint16_t *src = cg + (int64_t)l * p.c_ss + (int64_t)key * p.c_rs + c_col[jc];
            if (c_on[jc]) dma16_d(src, stage_off + C::KTILE + (wave * C::C_DMA + jc) * 1024);
        }
    };
    auto issue = [&](int step, const MixCursor &c) {
#pragma unroll
        for (int j = 0; j < C::DMA_PER_STAGE; ++j) issue_piece(step, c, j);
    };
    // Inside the main loop the pieces of tile step+2 are NOT issued in one burst after the barrier
    // (eight waves x five 1-KiB requests at once back up the CU's vector-memory issue path, and every
    // wave sits in that queue: ablation r01_d, "no DMA" = -33 % time) but in SLOTS spread over the step,
    // each one behind a group of MFMAs that keeps the matrix pipe busy while the request issues.
    constexpr int N_SLOTS = 5;
    auto issue_slot = [&](int step, const MixCursor &c, int slot) {
#if !defined(BP_ABL_MIX_NOSYNC) && !defined(BP_ABL_MIX_NODMA)
        if (step < nsteps) {
#pragma unroll
            for (int j = 0; j < C::DMA_PER_STAGE; ++j)
                if (j * N_SLOTS / C::DMA_PER_STAGE == slot) issue_piece(step, c, j);
        }
#endif
    };

    f32x16 acc[C::NB];
#pragma unroll
    for (int n = 0; n < C::NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    // lane-constant LDS read offsets
    int k_read_off[KD];   // K fragment (A operand of S^T): row l31 (+32*kk), logical slot 2*s + hh
#pragma unroll
    for (int s = 0; s < KD; ++s)
        k_read_off[s] = l31 * C::KROW + (((2 * s + hh) ^ k_swz<C::KROW>(l31)) * 16);
    // (k_swz only looks at row bits 0..3, so +32 rows keeps the same swizzle)
    const int c_row_lane = 4 * hh + ((lane & 15) >> 2);
    const int c_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    int c_read_off[C::NB];   // C^T fragment: (row c_row_lane, 16-col group of block n), +8 rows keeps swizzle
#pragma unroll
    for (int n = 0; n < C::NB; ++n) c_read_off[n] = v_lds_off<C::NB>(c_row_lane, n * 4 + c_ch_lane) + (lane & 1) * 8;

    u32x4 qf[KD];
    float lse2 = 0.f;

    // per-sense operands of this wave: my query's fragments (B operand of S^T = K Q^T) and its LSE
    const int qbox = C::QBOX_OFF + wave * C::QBOX_WAVE;
    auto issue_q = [&](int l) {   // SUPER: into the mailbox, through the DMA queue
        const uint16_t *row = qg + (int64_t)my_q_clamped * p.qk_rs + (int64_t)l * p.qk_ss;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            if (col < p.dk) dma16_d(row + col, lds0 + qbox + s * 1024);
        }
        dma4(p.lse + ((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q_clamped, lds0 + qbox + KD * 1024);
    };
    auto take_q = [&](int l) {
        if constexpr (C::SUPER) {
#pragma unroll
            for (int s = 0; s < KD; ++s) qf[s] = lds_read_16B(smem, qbox + s * 1024 + lane * 16);
            lse2 = *reinterpret_cast<const float *>(smem + qbox + KD * 1024 + lane * 4) * kLog2e;
        } else {
            const uint16_t *row = qg + (int64_t)my_q_clamped * p.qk_rs + (int64_t)l * p.qk_ss;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const int col = 16 * s + 8 * hh;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (col < p.dk) v = ld_global_16B(row + col);
                qf[s] = v;
            }
            lse2 = p.lse[((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q_clamped] * kLog2e;
        }
    };

    // ---- prologue: mailbox of step 0, then two tiles in flight ---------------------------------------
    MixCursor cur = {0, 0, 0}, cur1 = cur, cur2;
    if (C::SUPER && wave_has_rows) issue_q(0);
    issue(0, cur);
    advance(cur1);
    cur2 = cur1;
    if (nsteps > 1) issue(1, cur1);
    advance(cur2);

    // One pipeline step.  SLOT: the ring slot as a compile-time constant (the loop below is unrolled by the ring
    // depth so that the LDS addresses of the operand reads fold into instruction offsets).
    auto ring_step = [&](int step, auto SLOT) {
        constexpr int kSlot = decltype(SLOT)::value;
        const int l = cur.l;
        const int kb = cur.st * sup + cur.kk;
        // my share of tile `step` (and my mailbox, which is older than tile step+1 in the queue) has
        // landed; the tile after it may still be in flight ...
#if !defined(BP_ABL_MIX_NOSYNC) && !defined(BP_ABL_MIX_NODMA)   // ablation builds: timing only, wrong results
        if (step + 1 < nsteps) wait_vmcnt<C::DMA_PER_STAGE>(); else wait_vmcnt<0>();
#endif
#if !defined(BP_ABL_MIX_NOSYNC) && !defined(BP_ABL_MIX_NOBARRIER)
        // ... and so has everybody else's; all waves are also done reading tile step-1
        __builtin_amdgcn_s_barrier();
#endif
        if (cur.kk == 0 && wave_has_rows) take_q(l);
        if (C::SUPER && step + 1 < nsteps && cur1.kk == 0 && wave_has_rows) {
            // next step starts a new (super-tile, sense): refill the mailbox.  Queued BEFORE tile step+2, so
            // the counted wait at the top of the next step covers it.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of the old contents are done
            issue_q(cur1.l);
        }
        // tile step+2 refills the slot that was read during step-1: pieces go out in slots 0..4 below
#ifdef BP_MIX_BURST
        for (int sl = 0; sl < N_SLOTS; ++sl) issue_slot(step + 2, cur2, sl);
#else
        issue_slot(step + 2, cur2, 0);
#endif

        {
            const int stage_off = kSlot * C::STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int sub = kb * 2 + kk;
                const bool live = wave_has_rows && sub <= my_diag_sub;
                const int coff = stage_off + C::KTILE + kk * 32 * C::CROW;
                u32x4 pf[2];
                auto pv = [&](int ks) {
                    const int rows = coff + ks * 16 * C::CROW;
#pragma unroll
                    for (int n = 0; n < C::NB; ++n) {
                        if (FULL || n < nb_live) {
#ifdef BP_ABL_MIX_NOCREAD
                            u32x4 a = qf[n % KD];
                            asm volatile("" : "+v"(a));
#else
                            const u32x2 lo = lds_read_tr16_8B(smem, c_read_off[n] + rows);
                            const u32x2 hi = lds_read_tr16_8B(smem, c_read_off[n] + rows + 8 * C::CROW);
                            const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
#endif
                            acc[n] = E::mfma(a, pf[ks], acc[n]);
                        }
                    }
                };
                if (live) {
                    // one 32-key sub-block: S^T (KD MFMAs) -> P^T -> O^T += C^T P^T (2*NB MFMAs)
                    const int koff = stage_off + kk * 32 * C::KROW;
                    f32x16 st;
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#ifndef BP_ABL_MIX_NOS
#pragma unroll
                    for (int s = 0; s < KD; ++s) {
                        const u32x4 a = lds_read_16B(smem, k_read_off[s] + koff);
                        st = E::mfma(a, qf[s], st);
                    }
#ifndef BP_ABL_MIX_NOEXP
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = fast_exp2(fmaf(st[r], c2, -lse2));
#endif
#else
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = lse2;
#endif
                    if (WEIGHTED) {
                        // intervention hook: alpha[b, l, :, key] *= w[b, l, key]  (register r holds key
                        // (r & 3) + 8 (r >> 2) + 4 hh of the sub-block: four runs of four consecutive keys)
                        const int wbase = stage_off + C::KTILE + C::CTILE + wave * 256 + (kk * 32 + 4 * hh) * 4;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const u32x4 w4 = lds_read_16B(smem, wbase + g * 32);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const uint32_t wi = w4[i];   // by-value copy (bp_common.h, as_f32)
                                st[4 * g + i] *= as_f32(wi);
                            }
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            pf[ks][i] = E::pack2(st[ks * 8 + 2 * i], st[ks * 8 + 2 * i + 1]);
                    if (sub == my_diag_sub) {
                        // Diagonal sub-block (its first key is q0): clear the 16-bit P entries whose key
                        // lies above my query.  Done on the packed words with AND masks, in a small
                        // wave-uniform branch, so the common path carries no mask arithmetic and the MFMA
                        // code exists once.  (AND also kills an inf from an invisible, larger score.)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int r0 = ks * 8 + 2 * i;
                                const int rel0 = (r0 & 3) + 8 * (r0 >> 2) + 4 * hh;   // rel of r0 + 1 is rel0 + 1
                                const uint32_t keep = (rel0 <= l31 ? 0x0000ffffu : 0u) | (rel0 + 1 <= l31 ? 0xffff0000u : 0u);
                                pf[ks][i] &= keep;
                            }
                    }
                    pv(0);
                }
#ifndef BP_MIX_BURST
                issue_slot(step + 2, cur2, 2 * kk + 1);
#endif
                if (live) pv(1);
#ifndef BP_MIX_BURST
                issue_slot(step + 2, cur2, 2 * kk + 2);
#endif
            }
        }
        advance(cur); advance(cur1); advance(cur2);
    };
    static_assert(C::NSTAGE == 3, "the unrolled loop assumes a 3-slot ring");
    for (int step = 0; step < nsteps; step += 3) {
        ring_step(step, std::integral_constant<int, 0>{});
        if (step + 1 < nsteps) ring_step(step + 1, std::integral_constant<int, 1>{});
        if (step + 2 < nsteps) ring_step(step + 2, std::integral_constant<int, 2>{});
    }

    if (!wave_has_rows || my_q >= S) return;
    uint16_t *og = reinterpret_cast<uint16_t *>(p.o) + batch * p.o_bs + (int64_t)my_q * p.o_rs;
#pragma unroll
    for (int n = 0; n < C::NB; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = col_base + n * 32 + 8 * g + 4 * hh;
            if (col < p.dout) {
                u32x2 w = {E::pack2(acc[n][4 * g + 0], acc[n][4 * g + 1]),
                           E::pack2(acc[n][4 * g + 2], acc[n][4 * g + 3])};
                *reinterpret_cast<u32x2 *>(og + col) = w;
            }
        }
}

template <class ET, int KD>
static hipError_t launch_kd(const MixParams &p, hipStream_t stream) {
    const int grid = p.order >= 2 ? ((p.b + 7) / 8) * 8 * p.n_chunks * p.n_qtiles
                                  : xcd_grid(p.b * p.n_chunks, p.n_qtiles);
    dim3 g(grid), t(512);
    if (p.kw != nullptr) {
        if (p.dout % 256 == 0) hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, true, true>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, false, true>), g, t, 0, stream, p);
    } else if (p.dout % 256 == 0) hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, true, false>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, false, false>), g, t, 0, stream, p);
    return hipGetLastError();
}

template <class ET>
static hipError_t launch_et(const MixParams &p, hipStream_t stream) {
    switch ((p.dk + 15) / 16) {
        case 1: return launch_kd<ET, 1>(p, stream);
        case 2: return launch_kd<ET, 2>(p, stream);
        case 3: return launch_kd<ET, 3>(p, stream);
        case 4: return launch_kd<ET, 4>(p, stream);
        case 5: return launch_kd<ET, 5>(p, stream);
        case 6: return launch_kd<ET, 6>(p, stream);
        case 7: return launch_kd<ET, 7>(p, stream);
        default: return launch_kd<ET, 8>(p, stream);
    }
}

// Requires: d_k % 8 == 0, d_out % 8 == 0, all bases 16-byte aligned, all strides multiples of 8.
hipError_t launch_sense_mix_dma(const MixParams &p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_et<BF16>(p, stream) : launch_et<F16>(p, stream);
}

}  // namespace bp
