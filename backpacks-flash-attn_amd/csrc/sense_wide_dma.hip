// Wide senses, fast path: the reference's two few-sense configurations exactly -- backpack-mini-flash-vecs-4.yaml (k = 4,
// d_k = 160) and ...-vecs-1.yaml (k = 1, d_k = 640), training/configs/experiment/owt/ -- on 16-byte friendly operands with
// S a multiple of 32.  Everything else wider than 128 keeps the general kernels of sense_wide.hip.
//
//   sense_mix_wide_dma_kernel   out = sum_l softmax_causal(q_l k_l^T / sqrt(d_k)) C_l, alpha never stored  (backpack.py:313)
//   sense_lse_wide_dma_kernel   log-sum-exp of every (sense, query) row                                      (backpack.py:116-122)
//
// Same tile algebra as the rest of this directory (bp_common.h: S^T = K Q^T on v_mfma_f32_32x32x16, one query per lane,
// P^T straight into the second GEMM).  What differs from sense_wide.hip, whose schedule is the simple staged one
// (registers -> LDS, one __syncthreads per key block, 128 output columns per workgroup: S^T recomputed five times for
// d = 640, every MFMA behind its own LDS round trip):
//   * the K rows and the content rows of a 32-key block reach LDS by DMA (`global_load_lds_dwordx4`, bp_dma.h) into a
//     ring: block i + 1 is in flight while block i is multiplied; one s_barrier per block.  RING is a template parameter (2 to
//     4 slots with a counted vmcnt); three and four slots measured nothing over two (d_k = 160, B = 1024: 6.07 / 6.19 / 6.19
//     ms, profiles/r06_i_*), so two are shipped and every wait drains the ring;
//   * a workgroup covers NB * 32 = 320 output columns, so S^T and the exponentials are recomputed twice per query block
//     instead of five times; d_k = 160 runs eight waves per workgroup (256 queries share a block's rows; 4 waves x 320
//     columns: 7.96 ms, 8 waves x 160 columns: 9.3 ms, 8 x 320: 6.15 ms, profiles/r06_h_*), d_k = 640 four (its 160 fragment
//     and 160 accumulator registers leave one wave per SIMD);
//   * the LDS operands of both MFMA runs are requested two MFMAs ahead (mfma_stream, bp_common.h);
//   * the K image has an ODD row pitch (2 KD + 1 sixteen-byte slots): b128 reads of 16 consecutive rows touch every bank
//     once; the content image is XOR-swizzled on the DMA's source side for ds_read_b64_tr_b16.
// Algorithmic work per launch: 2 pairs (d_k + d) k B flop; bytes as SURVEY section 8(d): (4 + 2k + 2) S d B.
#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"

namespace bp {

template <int KD, int NW, int NB, int RING = 2, int SUB = 1>
struct WideDmaCfg {
    static constexpr int BK = 32 * SUB;                // keys per ring step: SUB 32-key blocks (one S^T / P V run each)
    static constexpr int NT = NW * 64;
    static constexpr int BM = NW * 32;                 // queries per workgroup (a wave owns 32)
    static constexpr int KSLOTS = 2 * KD + 1;          // 16-byte slots per K row, one of them padding (odd pitch)
    static constexpr int KROW = KSLOTS * 16;
    static constexpr int K_PIECES = (BK * KSLOTS + 63) / 64;     // 1-KiB DMA pieces; the K region is whole pieces
    static constexpr int KREGION = K_PIECES * 1024;
    static constexpr int CSLOTS = NB * 4;              // 16-byte slots per content row
    static constexpr int CROW = CSLOTS * 16;
    static constexpr int C_PIECES = BK * CSLOTS / 64;  // 2 NB (0 for the LSE kernel: K only)
    static constexpr int PIECES = K_PIECES + C_PIECES;
    static constexpr int NPW = (PIECES + NW - 1) / NW; // DMA instructions per wave and step: the SAME number for every wave, so
    static constexpr int STAGE = NPW * NW * 1024;      // that `s_waitcnt vmcnt((RING - 2) NPW)` means "my share of the oldest
                                                       // block in flight has landed" (pieces past PIECES re-fetch a K chunk
                                                       // into the stage's unused tail)
    static constexpr int WAIT = (RING - 2) * NPW;      // my pieces of the younger blocks that may still be in flight
    static_assert(RING >= 2 && RING * STAGE <= 160 * 1024, "LDS budget");
    static_assert(WAIT <= 63, "vmcnt is a 6-bit field");
};

// byte offset of (row, logical 16-byte chunk ch) in a content image of NB 64-byte chunks per row; ds_read_b64_tr_b16 serves
// 32 lanes = 4 consecutive rows x 64 B at once, the four row segments must lie in four different quarters of the banks:
//   NB = 5: rows are 320 B = 5 quarters apart -- distinct by construction;
//   NB = 10: 640 B = 10 quarters -- rows r and r + 2 would collide: chunk index XOR ((row >> 1) & 1).
template <int NB> BP_DEV int wide_c_off(int row, int ch) {
    static_assert(NB == 5 || NB == 10, "swizzle derived for 5 and 10 column blocks");
    int c64 = ch >> 2;
    if (NB == 10) c64 ^= (row >> 1) & 1;
    return row * (NB * 64) + ((c64 << 2) | (ch & 3)) * 16;
}

// The ring: per-lane source offsets of my DMA pieces (constant over the sweep), and the issue of one step's pieces.
template <int KD, int NW, int NB, int RING = 2, int SUB = 1>
struct WideRing {
    using C = WideDmaCfg<KD, NW, NB, RING, SUB>;
    uint32_t voff[C::NPW];
    // GATHER: the content rows are rows of a table picked by an index per key (bp_sense_mix_gather).  A content piece's
    // descriptor is then (row of the block) << 16 | byte offset of its column inside the chunk; the table row's byte offset
    // (index * row bytes, 32-bit: tables below 4 GiB) is added per step from the workgroup's index array in LDS.
    BP_DEV void setup(int wave, int lane, int64_t k_rs, int64_t c_rs, int col_base, int dout, bool gather = false) {
#pragma unroll
        for (int j = 0; j < C::NPW; ++j) {
            const int pi = wave * C::NPW + j;
            const int g = pi * 64 + lane;                     // K region: linear slot -> (row, chunk); pad slots and the
            const int krow = min(g / C::KSLOTS, C::BK - 1);   // region's tail re-fetch a valid chunk (never read)
            const int kch = min(g - (g / C::KSLOTS) * C::KSLOTS, 2 * KD - 1);
            uint32_t off = (uint32_t)(krow * k_rs + kch * 8) * 2u;
            if constexpr (NB > 0) if (pi >= C::K_PIECES && pi < C::PIECES) {
                const int c = (pi - C::K_PIECES) * 64 + lane;
                const int row = c / C::CSLOTS, stored = c - row * C::CSLOTS;
                int c64 = stored >> 2;
                if (NB == 10) c64 ^= (row >> 1) & 1;
                int col = col_base + ((c64 << 2) | (stored & 3)) * 8;
                if (col >= dout) col = col_base;              // columns past d_out: any finite data (never stored)
                off = gather ? ((uint32_t)row << 16) | (uint32_t)((col - col_base) * 2)
                             : (uint32_t)(row * c_rs + col) * 2u;
            }
            voff[j] = off;
        }
    }
    BP_DEV void issue(int wave, uint32_t stage, const uint16_t *kt, const uint16_t *ct) const {
#pragma unroll
        for (int j = 0; j < C::NPW; ++j) {
            const int pi = wave * C::NPW + j;
            dma16_s((NB > 0 && pi >= C::K_PIECES && pi < C::PIECES) ? ct : kt, voff[j],
                    __builtin_amdgcn_readfirstlane(stage + pi * 1024));
        }
    }
    // ids: the workgroup's row indices (uint32, already clamped to the table) of key 0, 1, ... in LDS; key0: first key of the block
    BP_DEV void issue_gather(int wave, uint32_t stage, const uint16_t *kt, const uint16_t *ct, const char *ids, int key0,
                             uint32_t row_bytes) const {
#pragma unroll
        for (int j = 0; j < C::NPW; ++j) {
            const int pi = wave * C::NPW + j;
            const bool content = NB > 0 && pi >= C::K_PIECES && pi < C::PIECES;   // wave-uniform
            uint32_t off = voff[j];
            if (content) {
                const uint32_t id = *reinterpret_cast<const uint32_t *>(ids + (key0 + (int)(off >> 16)) * 4);
                off = id * row_bytes + (off & 0xffffu);
            }
            dma16_s(content ? ct : kt, off, __builtin_amdgcn_readfirstlane(stage + pi * 1024));
        }
    }
};

// S^T (32 keys x 32 queries) of the block in `kbuf`: two accumulation chains, operands two MFMAs ahead
// (TWO_CHAINS: with one wave per SIMD a single chain of dependent MFMAs leaves a bubble behind each; with two waves the
// partner fills it and the second chain's 16 registers are better spent elsewhere)
template <class ET, int KD, int KROW, bool TWO_CHAINS>
BP_DEV f32x16 wide_dma_scores(const char *kbuf, const u32x4 (&qf)[KD], int l31, int hh) {
    using E = Elem<ET>;
    f32x16 st0, st1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st0[r] = 0.f; st1[r] = 0.f; }
    const int k_lane_off = l31 * KROW + hh * 16;
    mfma_stream<KD>([&](int i) { return lds_read_16B(kbuf, k_lane_off + i * 32); },
                    [&](int i, const u32x4 &a) {
                        if (TWO_CHAINS && (i & 1)) st1 = E::mfma(a, qf[i], st1);
                        else st0 = E::mfma(a, qf[i], st0);
                    });
    if (TWO_CHAINS) {
        settle_acc(st1);   // the wait for the matrix pipe belongs HERE, not behind whatever branch follows (bp_common.h)
        pin_acc(st0);
#pragma unroll
        for (int r = 0; r < 16; ++r) st0[r] += st1[r];
    } else {
        settle_acc(st0);
    }
    return st0;
}

template <int KD> BP_DEV void wide_dma_load_q(u32x4 (&qf)[KD], const uint16_t *qrow, int hh) {
#pragma unroll
    for (int s = 0; s < KD; ++s) qf[s] = ld_global_16B(qrow + 16 * s + 8 * hh);
#pragma unroll
    for (int s = 0; s < KD; ++s) settle(qf[s]);
}

// ---- fused mix ---------------------------------------------------------------------------------------------------------
template <class ET, int KD, int NW, int NB, int RING, bool GATHER, int SUB>
__global__ __launch_bounds__(NW * 64) void sense_mix_wide_dma_kernel(const MixParams p) {
    using C = WideDmaCfg<KD, NW, NB, RING, SUB>;
    using E = Elem<ET>;
    constexpr int kIdsOff = RING * C::STAGE;           // GATHER: the job's row indices behind the ring, 4 bytes per key
    static_assert(!GATHER || kIdsOff + kMixGatherMaxKeys * 4 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) char smem[kIdsOff + (GATHER ? kMixGatherMaxKeys * 4 : 0)];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int S = p.s;
    const uint32_t lds0 = lds_base_addr(smem);
    const int n_qtiles = (S + C::BM - 1) / C::BM;
    const int n_chunks = (p.dout + NB * 32 - 1) / (NB * 32);
    int grp, slot;
    if (!xcd_map(blockIdx.x, p.b * n_chunks, n_qtiles, grp, slot)) return;
    const int qt = n_qtiles - 1 - slot;               // heaviest query tiles first
    const int batch = grp / n_chunks, chunk = grp - batch * n_chunks;
    const int col_base = chunk * NB * 32;
    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.qk_bs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.qk_bs;
    const uint16_t *cg = reinterpret_cast<const uint16_t *>(p.c) + batch * p.c_bs;   // (GATHER: the table, c_bs = 0)
    const int k_end = min(S, qt * C::BM + C::BM);
    const int nkb = k_end / C::BK;                    // S % BK == 0 (launcher)
    const int nsteps = p.nsenses * nkb;
    if (GATHER) {   // my keys' table rows, clamped as unsigned values (a negative or too large index reads the LAST row)
        const int32_t *idx = p.row_index + batch * p.idx_bs;
        for (int i = tid; i < k_end; i += C::NT)
            *reinterpret_cast<uint32_t *>(smem + kIdsOff + i * 4) = min((uint32_t)idx[i], p.last_table_row);
        __syncthreads();
    }
    const int q0 = qt * C::BM + wave * 32, my_q = q0 + l31;
    const bool wave_has_rows = q0 < S;                // then every row of the wave exists
    const int my_last_kb = q0 / 32;                   // in 32-key blocks
    const float c2 = p.scale_log2e;
    const int nb_live = min(NB, (p.dout - col_base + 31) / 32);

    WideRing<KD, NW, NB, RING, SUB> ring;
    ring.setup(wave, lane, p.qk_rs, p.c_rs, col_base, p.dout, GATHER);
    int l_i = 0, kb_i = 0, slot_i = 0;                // (sense, key block) and ring slot of the next issue
    auto issue = [&]() {
        const uint16_t *kt = kg + (int64_t)l_i * p.qk_ss + (int64_t)kb_i * C::BK * p.qk_rs;
        if (GATHER) {
            ring.issue_gather(wave, lds0 + slot_i * C::STAGE, kt, cg + (int64_t)l_i * p.c_ss + col_base, smem + kIdsOff,
                              kb_i * C::BK, (uint32_t)p.c_rs * 2u);
        } else {
            const uint16_t *ct = cg + (int64_t)l_i * p.c_ss + (int64_t)kb_i * C::BK * p.c_rs;
            ring.issue(wave, lds0 + slot_i * C::STAGE, kt, ct);
        }
        if (++kb_i == nkb) { kb_i = 0; ++l_i; }
        if (++slot_i == RING) slot_i = 0;
    };

    f32x16 acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // content fragments (A operand of O^T = C^T P^T): lane -> (row, 16-byte chunk) of the 4 x 64 B a tr read serves
    const int c_row_lane = 4 * hh + ((lane & 15) >> 2);
    const int c_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    const int c_sub = (lane & 1) * 8;
    // block n's chunk index is n ^ b (NB = 10; b = (row >> 1) & 1, the same for row + 8 and row + 16): two lane offsets,
    // the rest are instruction offsets
    const int c_off_even = wide_c_off<NB>(c_row_lane, c_ch_lane) + c_sub;
    const int c_off_odd = wide_c_off<NB>(c_row_lane, 4 + c_ch_lane) + c_sub - 64;
    auto c_read_off = [&](int n) { return ((NB == 10 && (n & 1)) ? c_off_odd : c_off_even) + n * 64; };

    u32x4 qf[KD];
    float lse2 = 0.f;
    const uint16_t *qrow = qg + (int64_t)min(my_q, S - 1) * p.qk_rs;

    for (int t = 0; t < RING - 1 && t < nsteps; ++t) issue();
    int slot_r = 0;                        // ring slot of this step's block
    for (int step = 0; step < nsteps; ++step) {
        const int l = step / nkb, kb = step - l * nkb;
        // my pieces of this step's block have landed (RING - 2 younger blocks may stay in flight; in the sweep's last steps
        // there are fewer of them: wait for all) ...
        if (RING > 2 && step + RING - 2 < nsteps) wait_vmcnt<C::WAIT>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();      // ... so have everybody's; nobody reads the slot of the previous step any more
        if (kb == 0 && wave_has_rows) {    // a new sense: my row's fragments and log-sum-exp (the compiler's wait for these
                                           // loads also waits for the DMA in flight: once per sense)
            wide_dma_load_q<KD>(qf, qrow + (int64_t)l * p.qk_ss, hh);
            lse2 = p.lse[((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q] * kLog2e;
            settle(lse2);
        }
        if (step + RING - 1 < nsteps) issue();
        const char *stage = smem + slot_r * C::STAGE;
        if (++slot_r == RING) slot_r = 0;
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub)
        if (wave_has_rows && kb * SUB + sub <= my_last_kb) {
            const char *kbuf = stage + sub * 32 * C::KROW;
            const char *cbuf = stage + C::KREGION + sub * 32 * C::CROW;
            f32x16 st = wide_dma_scores<ET, KD, C::KROW, (NW <= 4)>(kbuf, qf, l31, hh);
            u32x4 pf[2];
            if (kb * SUB + sub == my_last_kb) {   // the diagonal block: exact zeros above the diagonal
                const int lim = l31 - 4 * hh;   // my_q - kb * 32 - 4 hh
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = ks * 8 + 2 * i;
                        float e0 = fast_exp2(fmaf(st[r], c2, -lse2)), e1 = fast_exp2(fmaf(st[r + 1], c2, -lse2));
                        if ((r & 3) + 8 * (r >> 2) > lim) e0 = 0.f;
                        if (((r + 1) & 3) + 8 * ((r + 1) >> 2) > lim) e1 = 0.f;
                        pf[ks][i] = E::pack2(e0, e1);
                    }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = ks * 8 + 2 * i;
                        pf[ks][i] = E::pack2(fast_exp2(fmaf(st[r], c2, -lse2)), fast_exp2(fmaf(st[r + 1], c2, -lse2)));
                    }
            }
            mfma_stream<2 * NB>(
                [&](int i) {
                    const int ks = i / NB, n = i - ks * NB;
                    const int off = c_read_off(n) + ks * 16 * C::CROW;
                    const u32x2 lo = lds_read_tr16_8B(cbuf, off);
                    const u32x2 hi = lds_read_tr16_8B(cbuf, off + 8 * C::CROW);
                    return u32x4{lo[0], lo[1], hi[0], hi[1]};
                },
                [&](int i, const u32x4 &a) {
                    const int ks = i / NB, n = i - ks * NB;
                    if (n < nb_live) acc[n] = E::mfma(a, pf[ks], acc[n]);
                });
        }
    }

    if (!wave_has_rows) return;
    settle_acc(acc[NB - 1]);   // (the loop's exit branch separates the last MFMAs from the stores)
#pragma unroll
    for (int n = 0; n < NB - 1; ++n) pin_acc(acc[n]);
    uint16_t *og = reinterpret_cast<uint16_t *>(p.o) + batch * p.o_bs + (int64_t)my_q * p.o_rs;
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = col_base + n * 32 + 8 * g + 4 * hh;   // d_out % 8 == 0: a group of 4 is inside or outside
            if (col < p.dout) {
                const u32x2 w = {E::pack2(acc[n][4 * g + 0], acc[n][4 * g + 1]), E::pack2(acc[n][4 * g + 2], acc[n][4 * g + 3])};
                *reinterpret_cast<u32x2 *>(og + col) = w;
            }
        }
}

// ---- LSE ---------------------------------------------------------------------------------------------------------------
template <class ET, int KD, int NW, int RING, int SUB>
__global__ __launch_bounds__(NW * 64) void sense_lse_wide_dma_kernel(const MixParams p, float *lse_out) {
    using C = WideDmaCfg<KD, NW, 0, RING, SUB>;
    __shared__ __attribute__((aligned(16))) char smem[RING * C::STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int S = p.s;
    const uint32_t lds0 = lds_base_addr(smem);
    const int n_qtiles = (S + C::BM - 1) / C::BM;
    int grp, slot;
    if (!xcd_map(blockIdx.x, p.b * p.nsenses, n_qtiles, grp, slot)) return;
    const int qt = n_qtiles - 1 - slot;
    const int batch = grp / p.nsenses, l = grp - batch * p.nsenses;
    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const int k_end = min(S, qt * C::BM + C::BM);
    const int nkb = k_end / C::BK;
    const int q0 = qt * C::BM + wave * 32, my_q = q0 + l31;
    const bool wave_has_rows = q0 < S;
    const int my_last_kb = q0 / 32;                   // in 32-key blocks
    const float c2 = p.scale_log2e;

    WideRing<KD, NW, 0, RING, SUB> ring;
    ring.setup(wave, lane, p.qk_rs, 0, 0, 0);
    u32x4 qf[KD];
    if (wave_has_rows) wide_dma_load_q<KD>(qf, qg + (int64_t)min(my_q, S - 1) * p.qk_rs, hh);
    float m_run = -INFINITY, l_run = 0.f;
    int kb_i = 0, slot_i = 0, slot_r = 0;
    auto issue = [&]() {
        const uint16_t *kt = kg + (int64_t)kb_i * C::BK * p.qk_rs;
        ring.issue(wave, lds0 + slot_i * C::STAGE, kt, kt);
        ++kb_i;
        if (++slot_i == RING) slot_i = 0;
    };
    for (int t = 0; t < RING - 1 && t < nkb; ++t) issue();
    for (int kb = 0; kb < nkb; ++kb) {
#ifdef BP_WIDE_WAIT_ALL   // probe build: deeper ring, but every wait drains it
        wait_vmcnt<0>();
#else
        if (RING > 2 && kb + RING - 2 < nkb) wait_vmcnt<C::WAIT>();
        else wait_vmcnt<0>();
#endif
        __builtin_amdgcn_s_barrier();
        if (kb + RING - 1 < nkb) issue();
        const char *stage = smem + slot_r * C::STAGE;
        if (++slot_r == RING) slot_r = 0;
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub)
        if (wave_has_rows && kb * SUB + sub <= my_last_kb) {
            f32x16 st = wide_dma_scores<ET, KD, C::KROW, (NW <= 4)>(stage + sub * 32 * C::KROW, qf, l31, hh);
            if (kb * SUB + sub == my_last_kb) {
                const int lim = l31 - 4 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) > lim) st[r] = -INFINITY;
            }
            float mxa = fmaxf(st[0], st[1]), mxb = fmaxf(st[2], st[3]);
#pragma unroll
            for (int r = 4; r < 16; r += 2) { mxa = fmaxf(mxa, st[r]); mxb = fmaxf(mxb, st[r + 1]); }
            const float mx = xhalf_max(fmaxf(mxa, mxb));       // finite: key kb * 32 is visible to every row of the block
            const float m_new = fmaxf(m_run, mx);
            const float mc = m_new * c2;
            const float alpha = fast_exp2(m_run * c2 - mc);    // 0 for a fresh row (m_run = -inf)
            m_run = m_new;
            float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                rs0 += fast_exp2(fmaf(st[r], c2, -mc));
                rs1 += fast_exp2(fmaf(st[r + 1], c2, -mc));
            }
            l_run = l_run * alpha + (rs0 + rs1);
        }
    }
    if (!wave_has_rows) return;
    const float l_tot = xhalf_sum(l_run);
    if (hh == 0)
        lse_out[((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q] =
            l_tot > 0.f ? (m_run * c2 + fast_log2(l_tot)) * kLn2 : -INFINITY;
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
// d_k = 160: eight waves x 160 columns (256 queries share a block's K and content rows; two waves per SIMD);
// d_k = 640: four waves x 320 columns (160 fragment + 160 accumulator registers: one wave per SIMD owns the file).
#ifndef BP_WIDE160_NW
#define BP_WIDE160_NW 8
#endif
#ifndef BP_WIDE160_NB
#define BP_WIDE160_NB 10
#endif
#ifndef BP_WIDE160_RING
#define BP_WIDE160_RING 2
#endif
#ifndef BP_WIDE160_LSE_RING
#define BP_WIDE160_LSE_RING 2
#endif
#ifndef BP_WIDE640_LSE_RING
#define BP_WIDE640_LSE_RING 2
#endif

// 32-key blocks per ring step at d_k = 160 when the length allows (s % (32 SUB) == 0).  Two blocks per step -- half the
// barriers and waits, 61 KB stages -- measured SLOWER for the mix (6.13 -> 6.43 ms at B = 1024, table form 6.59 -> 6.68; LSE
// 1.17 -> 1.15; profiles/r06_w_*): the barrier count is not what the step waits for.  One block per step is shipped.
#ifndef BP_WIDE160_SUB
#define BP_WIDE160_SUB 1
#endif

bool sense_wide_dma_takes(int s, int dk, int dout, bool vec_qk, bool vec_c, bool weighted) {
#ifdef BP_WIDE_NO_DMA   // variant build for A/B runs: everything wide on the staged kernels of sense_wide.hip
    return false;
#endif
    return vec_qk && vec_c && !weighted && (dk == 160 || dk == 640) && s % 32 == 0 && s > 0 && dout % 8 == 0;
}

template <class ET, int KD, int NW, int NB, int RING, int SUB>
static hipError_t launch_mix_wide_dma_cfg(const MixParams &p, hipStream_t stream) {
    using C = WideDmaCfg<KD, NW, NB, RING, SUB>;
    const int n_qtiles = (p.s + C::BM - 1) / C::BM;
    const int n_chunks = (p.dout + NB * 32 - 1) / (NB * 32);
    const dim3 grid(xcd_grid(p.b * n_chunks, n_qtiles)), block(C::NT);
    if (p.row_index != nullptr)
        hipLaunchKernelGGL((sense_mix_wide_dma_kernel<ET, KD, NW, NB, RING, true, SUB>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((sense_mix_wide_dma_kernel<ET, KD, NW, NB, RING, false, SUB>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_sense_mix_wide_dma(const MixParams &p, int dtype, hipStream_t stream) {
    if (p.dk == 160 && BP_WIDE160_SUB > 1 && p.s % (32 * BP_WIDE160_SUB) == 0)
        return dtype == 1
                   ? launch_mix_wide_dma_cfg<BF16, 10, BP_WIDE160_NW, BP_WIDE160_NB, BP_WIDE160_RING, BP_WIDE160_SUB>(p, stream)
                   : launch_mix_wide_dma_cfg<F16, 10, BP_WIDE160_NW, BP_WIDE160_NB, BP_WIDE160_RING, BP_WIDE160_SUB>(p, stream);
    if (p.dk == 160)
        return dtype == 1 ? launch_mix_wide_dma_cfg<BF16, 10, BP_WIDE160_NW, BP_WIDE160_NB, BP_WIDE160_RING, 1>(p, stream)
                          : launch_mix_wide_dma_cfg<F16, 10, BP_WIDE160_NW, BP_WIDE160_NB, BP_WIDE160_RING, 1>(p, stream);
    return dtype == 1 ? launch_mix_wide_dma_cfg<BF16, 40, 4, 10, 2, 1>(p, stream)
                      : launch_mix_wide_dma_cfg<F16, 40, 4, 10, 2, 1>(p, stream);
}

template <class ET, int KD, int NW, int RING, int SUB>
static hipError_t launch_lse_wide_dma_cfg(const MixParams &p, float *lse, hipStream_t stream) {
    using C = WideDmaCfg<KD, NW, 0, RING, SUB>;
    const dim3 grid(xcd_grid(p.b * p.nsenses, (p.s + C::BM - 1) / C::BM)), block(C::NT);
    hipLaunchKernelGGL((sense_lse_wide_dma_kernel<ET, KD, NW, RING, SUB>), grid, block, 0, stream, p, lse);
    return hipGetLastError();
}

hipError_t launch_sense_lse_wide_dma(const void *q, const void *k, float *lse, int64_t lse_stride, int64_t qk_bs,
                                     int64_t qk_rs, int64_t qk_ss, int b, int s, int nsenses, int dk, float scale_log2e,
                                     int dtype, hipStream_t stream) {
    MixParams p{};
    p.q = q; p.k = k; p.qk_bs = qk_bs; p.qk_rs = qk_rs; p.qk_ss = qk_ss;
    p.lse_stride = lse_stride; p.b = b; p.s = s; p.nsenses = nsenses; p.dk = dk; p.scale_log2e = scale_log2e;
    if (dk == 160 && BP_WIDE160_SUB > 1 && s % (32 * BP_WIDE160_SUB) == 0)
        return dtype == 1 ? launch_lse_wide_dma_cfg<BF16, 10, 8, BP_WIDE160_LSE_RING, BP_WIDE160_SUB>(p, lse, stream)
                          : launch_lse_wide_dma_cfg<F16, 10, 8, BP_WIDE160_LSE_RING, BP_WIDE160_SUB>(p, lse, stream);
    if (dk == 160)
        return dtype == 1 ? launch_lse_wide_dma_cfg<BF16, 10, 8, BP_WIDE160_LSE_RING, 1>(p, lse, stream)
                          : launch_lse_wide_dma_cfg<F16, 10, 8, BP_WIDE160_LSE_RING, 1>(p, lse, stream);
    return dtype == 1 ? launch_lse_wide_dma_cfg<BF16, 40, 4, BP_WIDE640_LSE_RING, 1>(p, lse, stream)
                      : launch_lse_wide_dma_cfg<F16, 40, 4, BP_WIDE640_LSE_RING, 1>(p, lse, stream);
}

}  // namespace bp
