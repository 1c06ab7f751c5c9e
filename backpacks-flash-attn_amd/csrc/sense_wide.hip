// Sense kernels for WIDE senses, 128 < d_k <= 640 (gfx950).
//
// The reference's few-sense ablations -- training/configs/experiment/owt/backpack-mini-flash-vecs-4.yaml (k = 4, d_k = 160)
// and ...-vecs-1.yaml (k = 1, d_k = 640) -- lie beyond the width the LDS-DMA kernels (and the reference's own attention
// kernels, csrc/flash_attn/fmha_api.cpp:245) take: a query's fragments no longer fit the register file next to the
// accumulators.  These three kernels cover them natively; same tile algebra as the rest of this directory (bp_common.h:
// S^T = K Q^T on v_mfma_f32_32x32x16, one query per lane, P^T straight into the second GEMM), with two changes:
//   * a query's fragments for one sense stay in REGISTERS for the whole key sweep in both width classes -- 48 registers up
//     to d_k = 192, 160 up to d_k = 640, where the workgroup's one wave per SIMD owns the whole 512-entry file and hipcc
//     parks what exceeds the 256 architectural registers in accumulation registers (no scratch) -- while the K rows of a
//     32-key block sit in LDS (up to 1296 bytes each);
//   * d_k is a run-time loop bound inside its class: one instantiation per dtype, alignment class and width class.
// With few senses this path is small next to the trunk (k = 1: 3 % of the Mini k = 64 mix flops), so the schedule is the
// simple one of flash_fwd.hip / sense_mix.hip (tiles staged through registers into a double-buffered LDS image, one
// __syncthreads per key block), not a ring.
//   sense_lse_wide_kernel    log-sum-exp of every (sense, query) row          (reference backpack.py:116-122, first half)
//   sense_alpha_wide_kernel  alpha (B,k,S,S) materialised, zeros above the diagonal     (backpack.py:116-122)
//   sense_mix_wide_kernel    out = sum_l alpha_l C_l, alpha never stored               (backpack.py:313)
#include "bp_common.h"
#include "bp_kernels.h"

namespace bp {

// KDT: compile-time bound of the 16-column steps of the sense width.  Two classes are instantiated:
//   KDT = 12  (d_k <= 192, e.g. vecs-4's 160): 48 fragment registers, 12.8 KB K tiles, two (mix: registers) to six (LSE,
//             alpha: LDS) workgroups per CU;
//   KDT = 40  (d_k <= 640, vecs-1): 160 fragment registers (the first version re-fetched them per key block from the L2:
//             33 ms per mix launch at Mini k = 1, B = 1024, profiles/r06_d_bench_mini_k1_native.json), 41.5 KB K tiles,
//             one workgroup per CU.
template <int KDT>
struct WideCfgT {
    static constexpr int BM = 128;                 // queries per workgroup (4 waves x 32)
    static constexpr int BK = 32;                  // keys per block
    static constexpr int NT = 256;
    static constexpr int KD_MAX = KDT;
    static constexpr int KROW_MAX = KD_MAX * 32 + 16;
    static constexpr int KTILE_MAX = BK * KROW_MAX;
    static constexpr int K_ITERS_MAX = (BK * KD_MAX * 2 + NT - 1) / NT;   // 16-byte chunks per thread and tile
    static constexpr int NB = 4;                   // 32-column blocks of the output per workgroup (mix)
    static constexpr int CROW = NB * 64;
    static constexpr int CTILE = BK * CROW;        // 8 192 bytes
    static constexpr int CCH = NB * 4;
    static constexpr int C_ITERS = (BK * CCH + NT - 1) / NT;   // 2
};
constexpr int kWideSmallKd = 12, kWideLargeKd = 40;

// K rows [kb*32, kb*32+32) of one sense -> registers -> LDS image (row pitch krow = 32*KD + 16 bytes: conflict-free b128)
template <bool VEC, int KDT>
struct WideKLoader {
    using WideCfg = WideCfgT<KDT>;
    u32x4 reg[WideCfg::K_ITERS_MAX];
    BP_DEV void fetch(const uint16_t *kg, int64_t k_rs, int kb, int S, int dk, int kd, int tid) {
        const int kch = kd * 2;
#pragma unroll
        for (int i = 0; i < WideCfg::K_ITERS_MAX; ++i) {
            const int c = tid + i * WideCfg::NT;
            const int row = c / kch, ch = c - row * kch;
            const int key = kb * WideCfg::BK + row;
            u32x4 v = {0u, 0u, 0u, 0u};   // keys past the sequence and columns past d_k are ZERO
            if (c < WideCfg::BK * kch && key < S && ch * 8 < dk) {
                const uint16_t *r = kg + (int64_t)key * k_rs;
                v = VEC ? ld_global_16B(r + ch * 8) : ld_global_8x2B(r, ch * 8, dk);
            }
            reg[i] = v;
        }
    }
    BP_DEV void stash(char *kbuf, int kd, int tid) const {
        const int kch = kd * 2, krow = kd * 32 + 16;
#pragma unroll
        for (int i = 0; i < WideCfg::K_ITERS_MAX; ++i) {
            const int c = tid + i * WideCfg::NT;
            const int row = c / kch, ch = c - row * kch;
            if (c < WideCfg::BK * kch) lds_write_16B(kbuf, row * krow + ch * 16, reg[i]);
        }
    }
};

// The query fragments of my row for one sense (B operand of S^T = K Q^T): column 16 s + 8 hh .. +7 of step s.
template <bool VEC, int KDT>
BP_DEV void wide_load_q(u32x4 (&qf)[KDT], const uint16_t *qrow, bool q_valid, int dk, int hh) {
#pragma unroll
    for (int s = 0; s < KDT; ++s) {
        const int col = 16 * s + 8 * hh;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q_valid && col < dk) v = VEC ? ld_global_16B(qrow + col) : ld_global_8x2B(qrow, col, dk);
        qf[s] = v;
    }
}

// S^T (32 keys x 32 queries) of one key block for my wave's 32 queries: st[r] = q[my_q] . k[kb*32 + (r&3) + 8*(r>>2) + 4*hh].
// Two accumulation chains (even / odd steps) so that consecutive MFMAs do not wait for each other; the fragments of the
// steps >= kd hold zeros and are skipped.
template <class ET, int KDT>
BP_DEV f32x16 wide_scores(const char *kbuf, const u32x4 (&qf)[KDT], int kd, int l31, int hh) {
    using E = Elem<ET>;
    f32x16 st0, st1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st0[r] = 0.f; st1[r] = 0.f; }
    const int krow = kd * 32 + 16;
    const int k_lane_off = l31 * krow + hh * 16;
#pragma unroll
    for (int s = 0; s < KDT; ++s)
        if (s < kd) {
            const u32x4 a = lds_read_16B(kbuf, k_lane_off + s * 32);
            if (s & 1) st1 = E::mfma(a, qf[s], st1);
            else st0 = E::mfma(a, qf[s], st0);
        }
    settle_acc(st1);   // the run-time bound `s < kd` puts a branch behind every MFMA: wait for the matrix pipe here (bp_common.h)
    pin_acc(st0);
#pragma unroll
    for (int r = 0; r < 16; ++r) st0[r] += st1[r];
    return st0;
}

struct WideParams {          // the three kernels' common part (all strides in 16-bit elements)
    const uint16_t *q, *k;   // q_l[t] = q + b*qk_bs + t*qk_rs + l*qk_ss
    int64_t qk_bs, qk_rs, qk_ss;
    float *lse;              // (b, nsenses, lse_stride) fp32, natural log
    int64_t lse_stride;
    int b, s, nsenses, dk;
    float scale_log2e;
};

// ---- LSE ------------------------------------------------------------------------------------------------------------
template <class ET, bool VEC, int KDT>
__global__ __launch_bounds__(256) void sense_lse_wide_kernel(const WideParams p) {
    using WideCfg = WideCfgT<KDT>;
    __shared__ __attribute__((aligned(16))) char smem[2 * WideCfg::KTILE_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int n_qtiles = (p.s + WideCfg::BM - 1) / WideCfg::BM;
    int grp, slot;
    if (!xcd_map(blockIdx.x, p.b * p.nsenses, n_qtiles, grp, slot)) return;
    const int qt = n_qtiles - 1 - slot;
    const int batch = grp / p.nsenses, l = grp - batch * p.nsenses;
    const int S = p.s, dk = p.dk, kd = (dk + 15) / 16;
    const int ktile = WideCfg::BK * (kd * 32 + 16);
    const uint16_t *qg = p.q + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const uint16_t *kg = p.k + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const int k_end = min(S, qt * WideCfg::BM + WideCfg::BM);
    const int nkb = (k_end + WideCfg::BK - 1) / WideCfg::BK;
    const int q0 = qt * WideCfg::BM + wave * 32, my_q = q0 + l31;
    const bool wave_has_rows = q0 < S;
    const int my_last_kb = q0 / WideCfg::BK;
    const float c2 = p.scale_log2e;
    const uint16_t *qrow = qg + (int64_t)min(my_q, S - 1) * p.qk_rs;

    WideKLoader<VEC, KDT> ld;
    u32x4 qf[KDT];
    wide_load_q<VEC, KDT>(qf, qrow, my_q < S, dk, hh);
    float m_run = -INFINITY, l_run = 0.f;
    ld.fetch(kg, p.qk_rs, 0, S, dk, kd, tid);
    ld.stash(smem, kd, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nkb) ld.fetch(kg, p.qk_rs, kb + 1, S, dk, kd, tid);
        if (wave_has_rows && kb <= my_last_kb) {
            f32x16 st = wide_scores<ET, KDT>(smem + cur * ktile, qf, kd, l31, hh);
            const int lim = min(S - 1, my_q) - kb * WideCfg::BK - 4 * hh;   // last visible key of my row, block-relative
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) > lim) st[r] = -INFINITY;
            float mx = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = xhalf_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float mc = (m_new == -INFINITY) ? 0.f : m_new * c2;
            const float alpha = fast_exp2(m_run * c2 - mc);
            m_run = m_new;
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) rs += fast_exp2(fmaf(st[r], c2, -mc));
            l_run = l_run * alpha + rs;
        }
        if (kb + 1 < nkb) ld.stash(smem + (cur ^ 1) * ktile, kd, tid);
        __syncthreads();
    }
    if (!wave_has_rows || my_q >= S) return;
    const float l_tot = xhalf_sum(l_run);
    if (hh == 0)
        p.lse[((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q] =
            l_tot > 0.f ? (m_run * c2 + fast_log2(l_tot)) * kLn2 : -INFINITY;
}

// ---- alpha materialised ------------------------------------------------------------------------------------------------
struct WideAlphaParams {
    WideParams w;
    uint16_t *alpha;          // (b, nsenses, s, s) 16-bit, contiguous
    int vec_store;            // 8-byte stores of 4 keys are aligned (s % 4 == 0, base aligned)
};

template <class ET, bool VEC, int KDT>
__global__ __launch_bounds__(256) void sense_alpha_wide_kernel(const WideAlphaParams pa) {
    using E = Elem<ET>;
    using WideCfg = WideCfgT<KDT>;
    const WideParams &p = pa.w;
    __shared__ __attribute__((aligned(16))) char smem[2 * WideCfg::KTILE_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int n_qtiles = (p.s + WideCfg::BM - 1) / WideCfg::BM;
    int grp, slot;
    if (!xcd_map(blockIdx.x, p.b * p.nsenses, n_qtiles, grp, slot)) return;
    const int qt = n_qtiles - 1 - slot;
    const int batch = grp / p.nsenses, l = grp - batch * p.nsenses;
    const int S = p.s, dk = p.dk, kd = (dk + 15) / 16;
    const int ktile = WideCfg::BK * (kd * 32 + 16);
    const uint16_t *qg = p.q + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const uint16_t *kg = p.k + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const int k_end = min(S, qt * WideCfg::BM + WideCfg::BM);
    const int nkb = (k_end + WideCfg::BK - 1) / WideCfg::BK;
    const int q0 = qt * WideCfg::BM + wave * 32, my_q = q0 + l31;
    const bool wave_has_rows = q0 < S;
    const int my_last_kb = q0 / WideCfg::BK;
    const float c2 = p.scale_log2e;
    const uint16_t *qrow = qg + (int64_t)min(my_q, S - 1) * p.qk_rs;
    const float lse2 = (wave_has_rows && my_q < S)
                           ? p.lse[((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q] * kLog2e : 0.f;
    uint16_t *arow = pa.alpha + (((int64_t)batch * p.nsenses + l) * S + min(my_q, S - 1)) * S;

    auto store4 = [&](int key0, float x0, float x1, float x2, float x3) {   // 4 consecutive keys of my row
        if (pa.vec_store) {
            if (key0 < S) {   // (s % 4 == 0: a group is inside or outside as a whole)
                const u32x2 w = {E::pack2(x0, x1), E::pack2(x2, x3)};
                *reinterpret_cast<u32x2 *>(arow + key0) = w;
            }
        } else {
            if (key0 + 0 < S) arow[key0 + 0] = E::from_float(x0);
            if (key0 + 1 < S) arow[key0 + 1] = E::from_float(x1);
            if (key0 + 2 < S) arow[key0 + 2] = E::from_float(x2);
            if (key0 + 3 < S) arow[key0 + 3] = E::from_float(x3);
        }
    };

    WideKLoader<VEC, KDT> ld;
    u32x4 qf[KDT];
    wide_load_q<VEC, KDT>(qf, qrow, my_q < S, dk, hh);
    ld.fetch(kg, p.qk_rs, 0, S, dk, kd, tid);
    ld.stash(smem, kd, tid);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nkb) ld.fetch(kg, p.qk_rs, kb + 1, S, dk, kd, tid);
        if (wave_has_rows && kb <= my_last_kb) {
            f32x16 st = wide_scores<ET, KDT>(smem + cur * ktile, qf, kd, l31, hh);
            const int lim = my_q - kb * WideCfg::BK - 4 * hh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = fast_exp2(fmaf(st[r], c2, -lse2));
                if ((r & 3) + 8 * (r >> 2) > lim) e = 0.f;      // above the diagonal: exact zeros
                st[r] = e;
            }
            if (my_q < S) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    store4(kb * WideCfg::BK + 8 * g + 4 * hh, st[4 * g], st[4 * g + 1], st[4 * g + 2], st[4 * g + 3]);
            }
        }
        if (kb + 1 < nkb) ld.stash(smem + (cur ^ 1) * ktile, kd, tid);
        __syncthreads();
    }
    // the key blocks entirely above my wave's rows: zeros, written explicitly (the caller's buffer is uninitialised)
    if (wave_has_rows && my_q < S) {
        const int nkb_all = (S + WideCfg::BK - 1) / WideCfg::BK;
        for (int kb = my_last_kb + 1; kb < nkb_all; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) store4(kb * WideCfg::BK + 8 * g + 4 * hh, 0.f, 0.f, 0.f, 0.f);
    }
}

// ---- fused mix -----------------------------------------------------------------------------------------------------------
template <class ET, bool VEC_QK, bool VEC_C, int KDT>
__global__ __launch_bounds__(256) void sense_mix_wide_kernel(const MixParams p) {
    using E = Elem<ET>;
    using W = WideCfgT<KDT>;
    __shared__ __attribute__((aligned(16))) char smem[2 * (W::KTILE_MAX + W::CTILE)];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int S = p.s, dk = p.dk, kd = (dk + 15) / 16;
    const int ktile = W::BK * (kd * 32 + 16), stage = ktile + W::CTILE;
    const int n_qtiles = (S + W::BM - 1) / W::BM;
    const int n_chunks = (p.dout + W::NB * 32 - 1) / (W::NB * 32);
    int grp, slot;
    if (!xcd_map(blockIdx.x, p.b * n_chunks, n_qtiles, grp, slot)) return;
    const int qt = n_qtiles - 1 - slot;
    const int batch = grp / n_chunks, chunk = grp - batch * n_chunks;
    const int col_base = chunk * W::NB * 32;
    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.qk_bs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.qk_bs;
    const uint16_t *cg = reinterpret_cast<const uint16_t *>(p.c) + batch * p.c_bs;
    const int k_end = min(S, qt * W::BM + W::BM);
    const int nkb = (k_end + W::BK - 1) / W::BK;
    const int nsteps = p.nsenses * nkb;
    const int q0 = qt * W::BM + wave * 32, my_q = q0 + l31;
    const bool wave_has_rows = q0 < S;
    const int my_last_kb = q0 / W::BK;
    const float c2 = p.scale_log2e;
    const int nb_live = min(W::NB, (p.dout - col_base + 31) / 32);

    WideKLoader<VEC_QK, KDT> ld;
    u32x4 qf[KDT];
    u32x4 creg[W::C_ITERS];
    auto fetch = [&](int step) {
        const int l = step / nkb, kb = step - l * nkb;
        ld.fetch(kg + (int64_t)l * p.qk_ss, p.qk_rs, kb, S, dk, kd, tid);
#pragma unroll
        for (int i = 0; i < W::C_ITERS; ++i) {
            const int c = tid + i * W::NT;
            const int row = c / W::CCH, ch = c - row * W::CCH;
            const int key = kb * W::BK + row;
            const int col = col_base + ch * 8;
            u32x4 v = {0u, 0u, 0u, 0u};   // keys past the sequence / columns past d_out are ZERO
            if (key < S && col < p.dout) {
                const uint16_t *r = cg + (int64_t)key * p.c_rs + (int64_t)l * p.c_ss;
                v = VEC_C ? ld_global_16B(r + col) : ld_global_8x2B(r, col, p.dout);
            }
            creg[i] = v;
        }
    };
    auto stash = [&](int buf) {
        char *kb_ = smem + buf * stage;
        ld.stash(kb_, kd, tid);
        char *cb_ = kb_ + ktile;
#pragma unroll
        for (int i = 0; i < W::C_ITERS; ++i) {
            const int c = tid + i * W::NT;
            const int row = c / W::CCH, ch = c - row * W::CCH;
            lds_write_16B(cb_, v_lds_off<W::NB>(row, ch), creg[i]);
        }
    };

    f32x16 acc[W::NB];
#pragma unroll
    for (int n = 0; n < W::NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int c_row_lane = 4 * hh + ((lane & 15) >> 2);
    const int c_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    const int c_sub = (lane & 1) * 8;
    float lse2 = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        const int l = step / nkb, kb = step - l * nkb;
        if (step + 1 < nsteps) fetch(step + 1);
        const uint16_t *qrow = qg + (int64_t)min(my_q, S - 1) * p.qk_rs + (int64_t)l * p.qk_ss;
        if (kb == 0 && wave_has_rows) {   // new sense: my row's log-sum-exp (and its fragments, when they live in registers)
            lse2 = (my_q < S) ? p.lse[((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q] * kLog2e : 0.f;
            wide_load_q<VEC_QK, KDT>(qf, qrow, my_q < S, dk, hh);
        }
        if (wave_has_rows && kb <= my_last_kb) {
            const char *kbuf = smem + cur * stage;
            const char *cbuf = kbuf + ktile;
            f32x16 st = wide_scores<ET, KDT>(kbuf, qf, kd, l31, hh);
            const int lim = my_q - kb * W::BK - 4 * hh;
            const float *kw = p.kw != nullptr ? p.kw + batch * p.kw_bs + (int64_t)l * p.kw_ss : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rel = (r & 3) + 8 * (r >> 2);
                float e = fast_exp2(fmaf(st[r], c2, -lse2));
                if (rel > lim || my_q >= S) e = 0.f;
                if (kw != nullptr) e *= kw[min(kb * W::BK + rel + 4 * hh, S - 1)];   // intervention hook
                st[r] = e;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 pf;
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[i] = E::pack2(st[ks * 8 + 2 * i], st[ks * 8 + 2 * i + 1]);
                const int row0 = ks * 16 + c_row_lane;
#pragma unroll
                for (int n = 0; n < W::NB; ++n)
                    if (n < nb_live) {
                        const int ch = n * 4 + c_ch_lane;
                        const u32x2 lo = lds_read_tr16_8B(cbuf, v_lds_off<W::NB>(row0, ch) + c_sub);
                        const u32x2 hi = lds_read_tr16_8B(cbuf, v_lds_off<W::NB>(row0 + 8, ch) + c_sub);
                        const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
                        acc[n] = E::mfma(a, pf, acc[n]);
                    }
            }
        }
        if (step + 1 < nsteps) stash(cur ^ 1);
        __syncthreads();
    }

    if (!wave_has_rows || my_q >= S) return;
    uint16_t *og = reinterpret_cast<uint16_t *>(p.o) + batch * p.o_bs + (int64_t)my_q * p.o_rs;
#pragma unroll
    for (int n = 0; n < W::NB; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = col_base + n * 32 + 8 * g + 4 * hh;
            const float x0 = acc[n][4 * g + 0], x1 = acc[n][4 * g + 1];
            const float x2 = acc[n][4 * g + 2], x3 = acc[n][4 * g + 3];
            if (VEC_C) {
                if (col < p.dout) {
                    u32x2 w = {E::pack2(x0, x1), E::pack2(x2, x3)};
                    *reinterpret_cast<u32x2 *>(og + col) = w;
                }
            } else {
                if (col + 0 < p.dout) og[col + 0] = E::from_float(x0);
                if (col + 1 < p.dout) og[col + 1] = E::from_float(x1);
                if (col + 2 < p.dout) og[col + 2] = E::from_float(x2);
                if (col + 3 < p.dout) og[col + 3] = E::from_float(x3);
            }
        }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
static WideParams wide_params(const void *q, const void *k, float *lse, int64_t lse_stride, int64_t qk_bs, int64_t qk_rs,
                              int64_t qk_ss, int b, int s, int nsenses, int dk, float scale_log2e) {
    WideParams w;
    w.q = static_cast<const uint16_t *>(q); w.k = static_cast<const uint16_t *>(k);
    w.qk_bs = qk_bs; w.qk_rs = qk_rs; w.qk_ss = qk_ss;
    w.lse = lse; w.lse_stride = lse_stride;
    w.b = b; w.s = s; w.nsenses = nsenses; w.dk = dk; w.scale_log2e = scale_log2e;
    return w;
}

// one instantiation per (dtype, alignment class, width class)
#define BP_WIDE_DISPATCH(KERNEL, DTYPE, VEC, DK, ...)                                                          \
    do {                                                                                                       \
        const bool small_ = (DK) <= 16 * kWideSmallKd;                                                         \
        if ((DTYPE) == 1) {                                                                                    \
            if (VEC) { if (small_) hipLaunchKernelGGL((KERNEL<BF16, true, kWideSmallKd>), __VA_ARGS__);        \
                       else hipLaunchKernelGGL((KERNEL<BF16, true, kWideLargeKd>), __VA_ARGS__); }             \
            else { if (small_) hipLaunchKernelGGL((KERNEL<BF16, false, kWideSmallKd>), __VA_ARGS__);           \
                   else hipLaunchKernelGGL((KERNEL<BF16, false, kWideLargeKd>), __VA_ARGS__); }                \
        } else {                                                                                               \
            if (VEC) { if (small_) hipLaunchKernelGGL((KERNEL<F16, true, kWideSmallKd>), __VA_ARGS__);         \
                       else hipLaunchKernelGGL((KERNEL<F16, true, kWideLargeKd>), __VA_ARGS__); }              \
            else { if (small_) hipLaunchKernelGGL((KERNEL<F16, false, kWideSmallKd>), __VA_ARGS__);            \
                   else hipLaunchKernelGGL((KERNEL<F16, false, kWideLargeKd>), __VA_ARGS__); }                 \
        }                                                                                                      \
    } while (0)

hipError_t launch_sense_lse_wide(const void *q, const void *k, float *lse, int64_t lse_stride, int64_t qk_bs,
                                 int64_t qk_rs, int64_t qk_ss, int b, int s, int nsenses, int dk, float scale_log2e,
                                 int dtype, bool vec, hipStream_t stream) {
    const WideParams w = wide_params(q, k, lse, lse_stride, qk_bs, qk_rs, qk_ss, b, s, nsenses, dk, scale_log2e);
    const dim3 grid(xcd_grid(b * nsenses, (s + 127) / 128)), block(256);
    BP_WIDE_DISPATCH(sense_lse_wide_kernel, dtype, vec, dk, grid, block, 0, stream, w);
    return hipGetLastError();
}

hipError_t launch_sense_alpha_wide(const void *q, const void *k, float *lse, int64_t lse_stride, void *alpha,
                                   int64_t qk_bs, int64_t qk_rs, int64_t qk_ss, int b, int s, int nsenses, int dk,
                                   float scale_log2e, int dtype, bool vec, hipStream_t stream) {
    WideAlphaParams pa;
    pa.w = wide_params(q, k, lse, lse_stride, qk_bs, qk_rs, qk_ss, b, s, nsenses, dk, scale_log2e);
    pa.alpha = static_cast<uint16_t *>(alpha);
    pa.vec_store = (s % 4 == 0) && ((reinterpret_cast<uintptr_t>(alpha) & 7) == 0);
    const dim3 grid(xcd_grid(b * nsenses, (s + 127) / 128)), block(256);
    BP_WIDE_DISPATCH(sense_alpha_wide_kernel, dtype, vec, dk, grid, block, 0, stream, pa);
    return hipGetLastError();
}

template <class ET, int KDT>
static hipError_t launch_mix_wide_et(const MixParams &p, bool vq, bool vc, hipStream_t stream) {
    using W = WideCfgT<KDT>;
    const int n_qtiles = (p.s + W::BM - 1) / W::BM;
    const int n_chunks = (p.dout + W::NB * 32 - 1) / (W::NB * 32);
    const dim3 grid(xcd_grid(p.b * n_chunks, n_qtiles)), block(W::NT);
    if (vq && vc) hipLaunchKernelGGL((sense_mix_wide_kernel<ET, true, true, KDT>), grid, block, 0, stream, p);
    else if (vq) hipLaunchKernelGGL((sense_mix_wide_kernel<ET, true, false, KDT>), grid, block, 0, stream, p);
    else if (vc) hipLaunchKernelGGL((sense_mix_wide_kernel<ET, false, true, KDT>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((sense_mix_wide_kernel<ET, false, false, KDT>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_sense_mix_wide(const MixParams &p, int dtype, bool vec_qk, bool vec_c, hipStream_t stream) {
    const bool small_ = p.dk <= 16 * kWideSmallKd;
    if (dtype == 1)
        return small_ ? launch_mix_wide_et<BF16, kWideSmallKd>(p, vec_qk, vec_c, stream)
                      : launch_mix_wide_et<BF16, kWideLargeKd>(p, vec_qk, vec_c, stream);
    return small_ ? launch_mix_wide_et<F16, kWideSmallKd>(p, vec_qk, vec_c, stream)
                  : launch_mix_wide_et<F16, kWideLargeKd>(p, vec_qk, vec_c, stream);
}

}  // namespace bp
