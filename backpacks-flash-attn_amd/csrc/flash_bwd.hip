// Attention backward for gfx950 (SURVEY.md section 8(f) "next" row 1): dQ, dK, dV of
//     O = softmax(scale * Q K^T [+causal]) V
// from Q, K, V, dO, O and the forward's row log-sum-exp L (D_i = sum_d dO_i[d] * O_i[d] is formed here).
// Takes the place of the reference's fmha_bwd_dq_dk_dv_loop_kernel
// (csrc/flash_attn/src/fmha_bwd_launch_template.h:31-114, src/fmha_dgrad_kernel_1xN_loop.h:95-711;
// Python side flash_attn/flash_attn_interface.py:31-47,70-84).  P is recomputed from L, as upstream.
//
//     P_ij  = exp(scale * q_i.k_j - L_i)            dV_j = sum_i P_ij dO_i
//     dP_ij = dO_i . v_j                            dS_ij = P_ij (dP_ij - D_i)
//     dQ_i  = scale * sum_j dS_ij k_j               dK_j = scale * sum_i dS_ij q_i
// With attention dropout (training; reference fmha_dgrad_kernel_1xN_loop.h:405-711 regenerates its Philox mask the
// same way): Z_ij = keep_ij / (1 - p) from bp_philox.h, the forward's mask bit for bit, and
//     dV_j = sum_i Z_ij P_ij dO_i                   dP_ij = Z_ij (dO_i . v_j)        (D_i already carries Z through O)
//
// Two kernels, both deterministic (no atomics -- the reference's sequence-parallel variant adds dQ
// with atomics and is only allclose-reproducible, tests/test_flash_attn.py:768-772):
//   * dkdv: a wave owns 32 KEYS; K and V live in registers as MFMA B operands; it sweeps the query
//     tiles that can see those keys.  S = Q K^T and dP = dO V^T come out with lane = key and the
//     queries along the registers, which is exactly the B-operand layout of the two products that
//     contract over queries (dV^T = dO^T P, dK^T = Q^T dS); their A operands are transposing LDS reads.
//   * dq (runs FIRST): a wave owns 32 QUERIES (Q, dO, L, D in registers); it sweeps key tiles as the forward
//     does: S^T = K Q^T, dP^T = V dO^T, dS^T feeds dQ^T = K^T dS^T.  Its prologue forms D for its rows from
//     the dO and O fragments it holds anyway and publishes it for the dkdv kernel (which follows on the same
//     stream), so no separate reduction pass over dO and O is needed.
// A tile that is read both row-wise (ds_read_b128) and transposed (ds_read_b64_tr_b16) is kept as two
// LDS images, each with the swizzle its read pattern needs; tiles arrive by the LDS-DMA ring (bp_dma.h).
#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"
#include "bp_philox.h"

namespace bp {

namespace {

template <int KD, int NV>
struct BwdCfg {
    static constexpr int NT = 256, NWAVE = 4, NSTAGE = 2, BT = 64;   // BT: rows of a streamed tile
    static constexpr int RROW = KD <= 4 ? 128 : 256;                 // row-image pitch (b128 reads)
    static constexpr int RSLOTS = RROW / 16;
    static constexpr int TROW = NV * 64;                             // transposed-read image pitch
    static constexpr int TCH = NV * 4;
    static constexpr int RTILE = BT * RROW;
    static constexpr int TTILE = BT * TROW;
    static constexpr int R_DMA = RTILE / 1024 / NWAVE;               // DMA instructions per wave
    static constexpr int T_DMA = TTILE / 1024 / NWAVE;
    static constexpr int R_ROWS_PER_DMA = 1024 / RROW;
};

// per-lane DMA descriptor (tile row, first column) of one 1-KiB piece of a row image /
// transposed-read image
template <class C>
BP_DEV void r_piece(int wave, int lane, int j, int &row, int &col) {
    row = (wave * C::R_DMA + j) * C::R_ROWS_PER_DMA + lane / C::RSLOTS;
    col = ((lane % C::RSLOTS) ^ k_swz<C::RROW>(row)) * 8;
}
template <class C, int NV>
BP_DEV void t_piece(int wave, int lane, int j, int &row, int &col) {
    const int c = (wave * C::T_DMA + j) * 64 + lane;
    row = c / C::TCH;
    const int stored = c - row * C::TCH;
    int c64 = stored >> 2;
    if (NV == 2) c64 ^= (row >> 1) & 1;
    if (NV == 4) c64 ^= row & 3;
    col = ((c64 << 2) | (stored & 3)) * 8;
}

}  // namespace

// =====================================================================================================
// dK, dV
// =====================================================================================================
template <int KD, int NV>
struct DkdvCfg {
    using C = BwdCfg<KD, NV>;
    // stage = Q row image | Q transposed-read image | dO row image | dO transposed-read image | stats
    static constexpr int STATS = 4 * 512;   // per wave: 64 x lse2, 64 x D (fp32)
    static constexpr int STAGE = 2 * C::RTILE + 2 * C::TTILE + STATS;
    static constexpr int SMEM = C::NSTAGE * STAGE;
};

// one 128-key tile `kt` of (sample, head) `bh`
template <class ET, int KD, int NV, bool DROP>
BP_DEV void flash_bwd_dkdv_tile(const FlashBwdParams p, char *smem, const uint32_t lds0, const int bh, const int kt) {
    using C = BwdCfg<KD, NV>;
    using E = Elem<ET>;
    constexpr int STATS = DkdvCfg<KD, NV>::STATS;
    constexpr int STAGE = DkdvCfg<KD, NV>::STAGE;
    constexpr int DMA_PER_STAGE = 2 * C::R_DMA + 2 * C::T_DMA + 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    const int batch = bh / p.h;
    const int head = bh - batch * p.h;

    int seq_q, seq_k;
    int64_t q_row0, k_row0;
    if (p.cu_q != nullptr) {
        const int a = p.cu_q[batch], b = p.cu_q[batch + 1];
        const int c = p.cu_k[batch], d = p.cu_k[batch + 1];
        seq_q = b - a; seq_k = d - c; q_row0 = a; k_row0 = c;
    } else {
        seq_q = p.max_sq; seq_k = p.max_sk;
        q_row0 = (int64_t)batch * p.max_sq; k_row0 = (int64_t)batch * p.max_sk;
    }
    if (kt * 128 >= seq_k) return;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + q_row0 * p.q_rs + (int64_t)head * p.q_hs;
    const uint16_t *dog = reinterpret_cast<const uint16_t *>(p.dout) + q_row0 * p.do_rs + (int64_t)head * p.do_hs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + k_row0 * p.k_rs + (int64_t)head * p.k_hs;
    const uint16_t *vg = reinterpret_cast<const uint16_t *>(p.v) + k_row0 * p.v_rs + (int64_t)head * p.v_hs;
    const float *lse_g = p.lse + ((int64_t)batch * p.h + head) * p.lse_stride;
    const float *dsum_g = p.dsum + ((int64_t)batch * p.h + head) * p.lse_stride;

    const int key0 = kt * 128 + wave * 32;        // first key of this wave
    const int my_key = key0 + l31;
    const bool wave_has_keys = key0 < seq_k;
    const float c2 = p.scale * kLog2e;

    // query tiles that can see this workgroup's keys: causal -> queries >= first key
    const int qt_begin = p.causal ? (kt * 128) / C::BT : 0;
    const int nqt = (seq_q + C::BT - 1) / C::BT;

    if (p.d * 2 != C::RROW) {   // pad slots of the row images must read as 0
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < C::NSTAGE * STAGE; off += C::NT * 16) lds_write_16B(smem, off, z);
        __syncthreads();
    }

    DropoutStream rng = {0u, 0u};
    if (DROP) rng = dropout_stream(p.rng_state, (uint32_t)bh);

    // ---- K and V fragments of my 32 keys: B operands (lane = key, 8 consecutive d) -----------------
    u32x4 kf[KD], vf[KD];
    {
        const int key = min(my_key, seq_k - 1);
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
            if (col < p.d) {
                a = ld_global_16B(kg + (int64_t)key * p.k_rs + col);
                b = ld_global_16B(vg + (int64_t)key * p.v_rs + col);
            }
            kf[s] = a;
            vf[s] = b;
        }
#pragma unroll
        for (int s = 0; s < KD; ++s) { settle(kf[s]); settle(vf[s]); }   // see bp_common.h: no vmcnt(0) in the loop
    }

    // ---- DMA descriptors -----------------------------------------------------------------------------
    int rr[C::R_DMA], rc[C::R_DMA], tr[C::T_DMA], tc[C::T_DMA];
#pragma unroll
    for (int j = 0; j < C::R_DMA; ++j) r_piece<C>(wave, lane, j, rr[j], rc[j]);
#pragma unroll
    for (int j = 0; j < C::T_DMA; ++j) t_piece<C, NV>(wave, lane, j, tr[j], tc[j]);
    auto issue = [&](int qt) {
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + ((qt - qt_begin) % C::NSTAGE) * STAGE);
        const int row_base = qt * C::BT;
#pragma unroll
        for (int j = 0; j < C::R_DMA; ++j) {
            const int64_t row = min(row_base + rr[j], seq_q - 1);
            if (rc[j] < p.d) {
                dma16_d(qg + row * p.q_rs + rc[j], st + (wave * C::R_DMA + j) * 1024);
                dma16_d(dog + row * p.do_rs + rc[j], st + C::RTILE + C::TTILE + (wave * C::R_DMA + j) * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < C::T_DMA; ++j) {
            const int64_t row = min(row_base + tr[j], seq_q - 1);
            if (tc[j] < p.d) {
                dma16_d(qg + row * p.q_rs + tc[j], st + C::RTILE + (wave * C::T_DMA + j) * 1024);
                dma16_d(dog + row * p.do_rs + tc[j], st + 2 * C::RTILE + C::TTILE + (wave * C::T_DMA + j) * 1024);
            }
        }
        // row statistics of the 64 queries (private copy per wave): lanes 0..15 L, 16..31 D
        if (lane < 32) {
            const int i4 = (lane & 15) * 4;
            const float *src = (lane < 16 ? lse_g : dsum_g) + min(row_base + i4, (int)p.lse_stride - 4);
            dma16_d(reinterpret_cast<const uint16_t *>(src), st + 2 * C::RTILE + 2 * C::TTILE + wave * 512);
        }
    };

    f32x16 dk[NV], dv[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[n][r] = 0.f; dv[n][r] = 0.f; }

    int r_read_off[KD];   // A operand rows (lane = query l31): slot 2s+hh swizzled
#pragma unroll
    for (int s = 0; s < KD; ++s) r_read_off[s] = l31 * C::RROW + (((2 * s + hh) ^ k_swz<C::RROW>(l31)) * 16);
    int t_read_off[NV];   // transposed-read operand: d block n, 4 consecutive rows
    {
        const int row_lane = 4 * hh + ((lane & 15) >> 2);
        const int ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
#pragma unroll
        for (int n = 0; n < NV; ++n) t_read_off[n] = v_lds_off<NV>(row_lane, n * 4 + ch_lane) + (lane & 1) * 8;
    }

    if (qt_begin < nqt) issue(qt_begin);
    // one ring step; SLOT = ring slot as a compile-time constant (loop unrolled by the ring depth: static LDS offsets)
    auto ring_step = [&](int qt, auto SLOT) {
        constexpr int kSlot = decltype(SLOT)::value;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (qt + 1 < nqt) issue(qt + 1);
        if (!wave_has_keys) return;
        const char *st = smem + kSlot * STAGE;
        const char *q_r = st, *q_t = st + C::RTILE;
        const char *do_r = st + C::RTILE + C::TTILE, *do_t = st + 2 * C::RTILE + C::TTILE;
        const char *stats = st + 2 * C::RTILE + 2 * C::TTILE + wave * 512;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qbase = qt * C::BT + qb * 32;          // first query of this 32-row sub-block
            if (qbase >= seq_q) continue;
            if (p.causal && qbase + 31 < key0) continue;     // every query is before my first key
            // ---- S = Q K^T and dP = dO V^T : rows = queries (registers), column = my key ---------------
            f32x16 s_, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s_[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const u32x4 a = lds_read_16B(q_r, r_read_off[s] + qb * 32 * C::RROW);
                s_ = E::mfma(a, kf[s], s_);
                const u32x4 b = lds_read_16B(do_r, r_read_off[s] + qb * 32 * C::RROW);
                dp = E::mfma(b, vf[s], dp);
            }
            // ---- P = exp2(S*c - L*log2e), dS = P (dP - D) ------------------------------------------------
            u32x4 pf[2], dsf[2];
            uint32_t keep = 0xffffu;   // bit 4g+i: query qbase + 8g + 4hh + i keeps my key
            if (DROP) keep = dropout_keep_collane(rng, p.drop_thr, (uint32_t)qbase, (uint32_t)my_key, hh);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 l4 = lds_read_16B(stats, (qb * 32 + 8 * g + 4 * hh) * 4);
                const u32x4 d4 = lds_read_16B(stats, 256 + (qb * 32 + 8 * g + 4 * hh) * 4);
                float pe[4], de[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * g + i;
                    const int q = qbase + 8 * g + 4 * hh + i;
                    float pv = fast_exp2(fmaf(s_[r], c2, -as_f32(l4[i]) * kLog2e));
                    const bool dead = q >= seq_q || my_key >= seq_k || (p.causal && my_key > q);
                    // selects, not multiplies: L / D of rows past the sequence are uninitialised (maybe NaN)
                    float z = 1.f;
                    if (DROP) z = ((keep >> r) & 1u) ? p.drop_scale : 0.f;
                    pe[i] = dead ? 0.f : (DROP ? pv * z : pv);
                    de[i] = dead ? 0.f : pv * ((DROP ? dp[r] * z : dp[r]) - as_f32(d4[i]));
                }
                // regs 8*ks .. 8*ks+7 are the B operand of K-step ks (queries {0..3, 8..11} + 4hh + 16ks)
                pf[g >> 1][(g & 1) * 2 + 0] = E::pack2(pe[0], pe[1]);
                pf[g >> 1][(g & 1) * 2 + 1] = E::pack2(pe[2], pe[3]);
                dsf[g >> 1][(g & 1) * 2 + 0] = E::pack2(de[0], de[1]);
                dsf[g >> 1][(g & 1) * 2 + 1] = E::pack2(de[2], de[3]);
            }
            // ---- dV^T += dO^T P ; dK^T += Q^T dS   (contraction over the 32 queries) -------------------
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int rows = (qb * 32 + ks * 16) * C::TROW;
#pragma unroll
                for (int n = 0; n < NV; ++n) {
                    const u32x2 lo = lds_read_tr16_8B(do_t, t_read_off[n] + rows);
                    const u32x2 hi = lds_read_tr16_8B(do_t, t_read_off[n] + rows + 8 * C::TROW);
                    dv[n] = E::mfma(u32x4{lo[0], lo[1], hi[0], hi[1]}, pf[ks], dv[n]);
                    const u32x2 lo2 = lds_read_tr16_8B(q_t, t_read_off[n] + rows);
                    const u32x2 hi2 = lds_read_tr16_8B(q_t, t_read_off[n] + rows + 8 * C::TROW);
                    dk[n] = E::mfma(u32x4{lo2[0], lo2[1], hi2[0], hi2[1]}, dsf[ks], dk[n]);
                }
            }
        }
    };
    static_assert(C::NSTAGE == 2, "unrolled by the 2-slot ring");
    for (int qt = qt_begin; qt < nqt; qt += 2) {
        ring_step(qt, std::integral_constant<int, 0>{});
        if (qt + 1 < nqt) ring_step(qt + 1, std::integral_constant<int, 1>{});
    }

    if (!wave_has_keys || my_key >= seq_k) return;
    uint16_t *dkg = reinterpret_cast<uint16_t *>(p.dk) + (k_row0 + my_key) * p.dk_rs + (int64_t)head * p.dk_hs;
    uint16_t *dvg = reinterpret_cast<uint16_t *>(p.dv) + (k_row0 + my_key) * p.dv_rs + (int64_t)head * p.dv_hs;
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = n * 32 + 8 * g + 4 * hh;
            if (d0 < p.d) {
                u32x2 a = {E::pack2(dk[n][4 * g] * p.scale, dk[n][4 * g + 1] * p.scale),
                           E::pack2(dk[n][4 * g + 2] * p.scale, dk[n][4 * g + 3] * p.scale)};
                u32x2 b = {E::pack2(dv[n][4 * g], dv[n][4 * g + 1]), E::pack2(dv[n][4 * g + 2], dv[n][4 * g + 3])};
                *reinterpret_cast<u32x2 *>(dkg + d0) = a;
                *reinterpret_cast<u32x2 *>(dvg + d0) = b;
            }
        }
}

// =====================================================================================================
// dQ
// =====================================================================================================
// one 128-query tile `qt` of (sample, head) `bh`
template <class ET, int KD, int NV, bool DROP>
BP_DEV void flash_bwd_dq_tile(const FlashBwdParams p, char *smem, const uint32_t lds0, const int bh, const int qt) {
    using C = BwdCfg<KD, NV>;
    using E = Elem<ET>;
    // stage = K row image | K transposed-read image | V row image
    constexpr int STAGE = 2 * C::RTILE + C::TTILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    const int batch = bh / p.h;
    const int head = bh - batch * p.h;

    int seq_q, seq_k;
    int64_t q_row0, k_row0;
    if (p.cu_q != nullptr) {
        const int a = p.cu_q[batch], b = p.cu_q[batch + 1];
        const int c = p.cu_k[batch], d = p.cu_k[batch + 1];
        seq_q = b - a; seq_k = d - c; q_row0 = a; k_row0 = c;
    } else {
        seq_q = p.max_sq; seq_k = p.max_sk;
        q_row0 = (int64_t)batch * p.max_sq; k_row0 = (int64_t)batch * p.max_sk;
    }
    if (qt * 128 >= seq_q) return;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + q_row0 * p.q_rs + (int64_t)head * p.q_hs;
    const uint16_t *dog = reinterpret_cast<const uint16_t *>(p.dout) + q_row0 * p.do_rs + (int64_t)head * p.do_hs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + k_row0 * p.k_rs + (int64_t)head * p.k_hs;
    const uint16_t *vg = reinterpret_cast<const uint16_t *>(p.v) + k_row0 * p.v_rs + (int64_t)head * p.v_hs;

    int k_end = seq_k;
    if (p.causal) k_end = min(seq_k, qt * 128 + 128);
    const int nkb = (k_end + C::BT - 1) / C::BT;

    const int q0 = qt * 128 + wave * 32;
    const int my_q = q0 + l31;
    const bool wave_has_rows = q0 < seq_q;
    const float c2 = p.scale * kLog2e;

    if (p.d * 2 != C::RROW) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < C::NSTAGE * STAGE; off += C::NT * 16) lds_write_16B(smem, off, z);
        __syncthreads();
    }

    DropoutStream rng = {0u, 0u};
    if (DROP) rng = dropout_stream(p.rng_state, (uint32_t)bh);

    u32x4 qf[KD], dof[KD];
    float lse2 = 0.f, dsum = 0.f;
    {
        const int q = min(my_q, seq_q - 1);
        const uint16_t *og = reinterpret_cast<const uint16_t *>(p.out) + (q_row0 + q) * p.o_rs + (int64_t)head * p.o_hs;
        float part = 0.f;   // my 8*KD columns of dO . O
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u}, o = {0u, 0u, 0u, 0u};
            if (col < p.d) {
                a = ld_global_16B(qg + (int64_t)q * p.q_rs + col);
                b = ld_global_16B(dog + (int64_t)q * p.do_rs + col);
                o = ld_global_16B(og + col);
            }
            qf[s] = a;
            dof[s] = b;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t bw = b[i], ow = o[i];   // by-value copies (bp_common.h, as_f32)
                part = fmaf(E::lo_f32(bw), E::lo_f32(ow), part);
                part = fmaf(E::hi_f32(bw), E::hi_f32(ow), part);
            }
        }
        const int64_t so = ((int64_t)batch * p.h + head) * p.lse_stride + q;
        lse2 = p.lse[so] * kLog2e;
        dsum = xhalf_sum(part);   // the two half-waves hold the two 8-column halves of every 16
        if (hh == 0 && wave_has_rows && my_q < seq_q) p.dsum[so] = dsum;
#pragma unroll
        for (int s = 0; s < KD; ++s) { settle(qf[s]); settle(dof[s]); }
        settle(lse2); settle(dsum);
    }

    int rr[C::R_DMA], rc[C::R_DMA], tr[C::T_DMA], tc[C::T_DMA];
#pragma unroll
    for (int j = 0; j < C::R_DMA; ++j) r_piece<C>(wave, lane, j, rr[j], rc[j]);
#pragma unroll
    for (int j = 0; j < C::T_DMA; ++j) t_piece<C, NV>(wave, lane, j, tr[j], tc[j]);
    auto issue = [&](int kb) {
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + (kb % C::NSTAGE) * STAGE);
#pragma unroll
        for (int j = 0; j < C::R_DMA; ++j) {
            const int64_t row = min(kb * C::BT + rr[j], seq_k - 1);
            if (rc[j] < p.d) {
                dma16_d(kg + row * p.k_rs + rc[j], st + (wave * C::R_DMA + j) * 1024);
                dma16_d(vg + row * p.v_rs + rc[j], st + C::RTILE + C::TTILE + (wave * C::R_DMA + j) * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < C::T_DMA; ++j) {
            const int64_t row = min(kb * C::BT + tr[j], seq_k - 1);
            if (tc[j] < p.d) dma16_d(kg + row * p.k_rs + tc[j], st + C::RTILE + (wave * C::T_DMA + j) * 1024);
        }
    };

    f32x16 dq[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[n][r] = 0.f;

    int r_read_off[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) r_read_off[s] = l31 * C::RROW + (((2 * s + hh) ^ k_swz<C::RROW>(l31)) * 16);
    int t_read_off[NV];
    {
        const int row_lane = 4 * hh + ((lane & 15) >> 2);
        const int ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
#pragma unroll
        for (int n = 0; n < NV; ++n) t_read_off[n] = v_lds_off<NV>(row_lane, n * 4 + ch_lane) + (lane & 1) * 8;
    }

    if (nkb > 0) issue(0);
    auto ring_step = [&](int kb, auto SLOT) {
        constexpr int kSlot = decltype(SLOT)::value;
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kb + 1 < nkb) issue(kb + 1);
        if (!wave_has_rows) return;
        if (p.causal && kb * C::BT > q0 + 31) return;
        const char *st = smem + kSlot * STAGE;
        const char *k_r = st, *k_t = st + C::RTILE, *v_r = st + C::RTILE + C::TTILE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kbase = kb * C::BT + kk * 32;
            if (kbase >= seq_k) continue;
            if (p.causal && kbase > q0 + 31) continue;
            // S^T = K Q^T and dP^T = V dO^T : rows = keys (registers), column = my query
            f32x16 st_, dpt;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st_[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const u32x4 a = lds_read_16B(k_r, r_read_off[s] + kk * 32 * C::RROW);
                st_ = E::mfma(a, qf[s], st_);
                const u32x4 b = lds_read_16B(v_r, r_read_off[s] + kk * 32 * C::RROW);
                dpt = E::mfma(b, dof[s], dpt);
            }
            u32x4 dsf[2];
            uint32_t keep = 0xffffu;   // bit 4g+i: my query keeps key kbase + 8g + 4hh + i
            if (DROP) keep = dropout_keep_rowlane(rng, p.drop_thr, (uint32_t)my_q, (uint32_t)kbase, hh);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float de[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * g + i;
                    const int key = kbase + 8 * g + 4 * hh + i;
                    const float pv = fast_exp2(fmaf(st_[r], c2, -lse2));
                    const bool dead = key >= seq_k || (p.causal && key > my_q);
                    float dpe = dpt[r];
                    if (DROP) dpe = ((keep >> r) & 1u) ? dpe * p.drop_scale : 0.f;
                    de[i] = dead ? 0.f : pv * (dpe - dsum);
                }
                dsf[g >> 1][(g & 1) * 2 + 0] = E::pack2(de[0], de[1]);
                dsf[g >> 1][(g & 1) * 2 + 1] = E::pack2(de[2], de[3]);
            }
            // dQ^T += K^T dS^T  (contraction over the 32 keys)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int rows = (kk * 32 + ks * 16) * C::TROW;
#pragma unroll
                for (int n = 0; n < NV; ++n) {
                    const u32x2 lo = lds_read_tr16_8B(k_t, t_read_off[n] + rows);
                    const u32x2 hi = lds_read_tr16_8B(k_t, t_read_off[n] + rows + 8 * C::TROW);
                    dq[n] = E::mfma(u32x4{lo[0], lo[1], hi[0], hi[1]}, dsf[ks], dq[n]);
                }
            }
        }
    };
    static_assert(C::NSTAGE == 2, "unrolled by the 2-slot ring");
    for (int kb = 0; kb < nkb; kb += 2) {
        ring_step(kb, std::integral_constant<int, 0>{});
        if (kb + 1 < nkb) ring_step(kb + 1, std::integral_constant<int, 1>{});
    }

    if (!wave_has_rows || my_q >= seq_q) return;
    uint16_t *dqg = reinterpret_cast<uint16_t *>(p.dq) + (q_row0 + my_q) * p.dq_rs + (int64_t)head * p.dq_hs;
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = n * 32 + 8 * g + 4 * hh;
            if (d0 < p.d) {
                u32x2 a = {E::pack2(dq[n][4 * g] * p.scale, dq[n][4 * g + 1] * p.scale),
                           E::pack2(dq[n][4 * g + 2] * p.scale, dq[n][4 * g + 3] * p.scale)};
                *reinterpret_cast<u32x2 *>(dqg + d0) = a;
            }
        }
}

// Kernels: a causal workgroup takes the heaviest remaining tile and the lightest of its (sample, head) -- tiles t
// and n-1-t -- so that every workgroup carries the same work (in-order round-robin dispatch, see flash_fwd_dma.hip).
template <class ET, int KD, int NV, bool DROP>
__global__ __launch_bounds__(256) void flash_bwd_dkdv_kernel(const FlashBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem[DkdvCfg<KD, NV>::SMEM];
    const uint32_t lds0 = lds_base_addr(smem);
    const int n = (p.max_sk + 127) / 128;
    const bool pair = p.causal && n > 1;
    int bh, slot;
    if (!xcd_map(blockIdx.x, p.b * p.h, pair ? (n + 1) / 2 : n, bh, slot)) return;   // key tile 0 = most work
    const int other = n - 1 - slot;
    const int npass = (pair && other != slot) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass) __syncthreads();
        flash_bwd_dkdv_tile<ET, KD, NV, DROP>(p, smem, lds0, bh, pass ? other : slot);
    }
}

template <class ET, int KD, int NV, bool DROP>
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(const FlashBwdParams p) {
    using C = BwdCfg<KD, NV>;
    __shared__ __attribute__((aligned(16))) char smem[C::NSTAGE * (2 * C::RTILE + C::TTILE)];
    const uint32_t lds0 = lds_base_addr(smem);
    const int n = (p.max_sq + 127) / 128;
    const bool pair = p.causal && n > 1;
    int bh, slot;
    if (!xcd_map(blockIdx.x, p.b * p.h, pair ? (n + 1) / 2 : n, bh, slot)) return;
    const int heavy = n - 1 - slot;
    const int npass = (pair && heavy != slot) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass) __syncthreads();
        flash_bwd_dq_tile<ET, KD, NV, DROP>(p, smem, lds0, bh, pass ? slot : heavy);
    }
}

template <class ET, int KD, int NV, bool DROP>
static hipError_t launch_drop(const FlashBwdParams &p, hipStream_t stream) {
    // dq first: it also produces the D vector the dkdv kernel consumes
    const int nq = (p.max_sq + 127) / 128, nk = (p.max_sk + 127) / 128;
    const int gq = xcd_grid(p.b * p.h, (p.causal && nq > 1) ? (nq + 1) / 2 : nq);
    hipLaunchKernelGGL((flash_bwd_dq_kernel<ET, KD, NV, DROP>), dim3(gq), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int gk = xcd_grid(p.b * p.h, (p.causal && nk > 1) ? (nk + 1) / 2 : nk);
    hipLaunchKernelGGL((flash_bwd_dkdv_kernel<ET, KD, NV, DROP>), dim3(gk), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <class ET, int KD, int NV>
static hipError_t launch_one(const FlashBwdParams &p, hipStream_t stream) {
    return p.drop_thr != 0u ? launch_drop<ET, KD, NV, true>(p, stream) : launch_drop<ET, KD, NV, false>(p, stream);
}

template <class ET>
static hipError_t launch_et(const FlashBwdParams &p, hipStream_t stream) {
    switch ((p.d + 15) / 16) {
        case 1: return launch_one<ET, 1, 1>(p, stream);
        case 2: return launch_one<ET, 2, 1>(p, stream);
        case 3: return launch_one<ET, 3, 2>(p, stream);
        case 4: return launch_one<ET, 4, 2>(p, stream);
        case 5: return launch_one<ET, 5, 4>(p, stream);   // d_h = 80 (Mini)
        case 6: return launch_one<ET, 6, 4>(p, stream);
        case 7: return launch_one<ET, 7, 4>(p, stream);
        default: return launch_one<ET, 8, 4>(p, stream);
    }
}

// head_dim % 8 == 0 and <= 128 (the trunk's 64 / 80 and the senses' 48/40/24), 16-byte friendly strides.
hipError_t launch_flash_bwd(const FlashBwdParams &p, int dtype, hipStream_t stream) {
    if (p.d > 128) return hipErrorNotSupported;
    return dtype == 1 ? launch_et<BF16>(p, stream) : launch_et<F16>(p, stream);
}

}  // namespace bp
