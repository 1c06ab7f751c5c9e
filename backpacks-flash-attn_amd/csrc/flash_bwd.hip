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
//     dV_j = sum_i Z_ij P_ij dO_i                   dS_ij = P_ij (Z_ij dP_ij - D_i)   (D_i already carries Z through O)
//
// Two kernels, both deterministic (no atomics -- the reference's sequence-parallel variant adds dQ
// with atomics and is only allclose-reproducible, tests/test_flash_attn.py:768-772):
//   * dq (runs FIRST): a wave owns 32 QUERIES (Q, dO in registers); it sweeps key tiles as the forward does:
//     S^T = K Q^T, dP^T = V dO^T, dS^T feeds dQ^T = K^T dS^T.  Its prologue forms D for its rows from the dO and O
//     fragments it holds anyway and publishes -D and -L / scale for the dkdv kernel (same stream).
//   * dkdv: a wave owns 32 KEYS; K and V live in registers as MFMA B operands; it sweeps the query tiles that can
//     see those keys.  S = Q K^T and dP = dO V^T come out with lane = key and the queries along the registers,
//     which is exactly the B-operand layout of the two products that contract over queries (dV^T = dO^T P,
//     dK^T = Q^T dS); their A operands are transposing LDS reads.
// What the round-2 version of this file spent its time on (r03_a PMC: 15-16 VALU + 6 SALU per MFMA, three times
// the tile body's floor) and what replaced it:
//   * ONE LDS image per streamed tile serves both read patterns -- row-wise ds_read_b128 (A operand of the
//     products that contract over the head dimension) and ds_read_b64_tr_b16 (A operand of the products that
//     contract over the tile's rows): 16-byte slot s of row r lives at slot s ^ swz(r), with swz chosen so that
//     the 16 rows of a b128 pass hit 16 different 4-bank groups AND the 4 rows of a transposing pass lie in the
//     4 different bank quarters (bwd_swz below).  Half the DMA instructions, half the LDS of the two-image form.
//   * DMA in the "saddr" form: scalar tile pointer advanced on the scalar unit, per-lane byte offsets fixed at
//     kernel start (a second, clamped set for the one partial tile a sweep can meet) -- zero VALU per tile.
//   * -L / scale and -D are not subtracted per element: they are the INITIAL VALUE of the S and dP MFMA
//     accumulators (dkdv: the per-query values are read from the tile's statistics straight into the accumulator
//     registers; dq: -D is a constant register vector used as the C operand of the first MFMA, -L a per-lane scalar
//     inside the exponent's fma).  A score element then costs  mul (or fma), exp, mul  and the packs.
//   * two tile bodies per kernel, in sequential loops with one body each: the clean body (every (query, key) pair
//     of the wave's sub-block visible, no sequence end) carries no compare or select at all; the edge body (the
//     diagonal sub-block, the sequence's last tile) masks with selects against per-lane limits.
#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"
#include "bp_philox.h"

namespace bp {

// Development builds only (-DBP_BWD_PROFILE, scripts/probes/flash_bwd_timeline): wave 0 of every workgroup stamps
// s_memtime at the phases of each pass; never in the shipped library.
#ifdef BP_BWD_PROFILE
__device__ unsigned long long g_bwd_prof[2][8192][2][8];   // [kernel: 0 dq, 1 dkdv][workgroup][pass][stamp]
#define BWD_STAMP(kern, pass, k)                                                                            \
    do {                                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_bwd_prof[kern][blockIdx.x][pass][k] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define BWD_STAMP(kern, pass, k) do { } while (0)
#endif

namespace {

// Development builds only (-DBP_BWD_WHATIF=<bits>, scripts/probes/flash_bwd_whatif): timing builds that DELETE one kind
// of work (results are garbage on purpose).  1 no v_exp_f32, 2 no S / dP products, 4 no dV / dK / dQ products, 8 no
// s_barrier (racy), 16 no epilogue stores, 32 no softmax VALU at all.  0 in the shipped library: every test below folds away.
#ifndef BP_BWD_WHATIF
#define BP_BWD_WHATIF 0
#elif BP_BWD_WHATIF != 0
#warning "BP_BWD_WHATIF: timing build of flash_bwd.hip -- gradients are garbage (bp_build_flags() reports it, bp_hip refuses it as the default library)"
#endif
constexpr int kWhatIf = BP_BWD_WHATIF;

// ring depth of the streamed tiles: tile t + NSTAGE - 1 is requested at the start of step t (three slots measured
// against two on one box, r03_j: 0.631 vs 0.616 ms at B = 64 -- the tile latency is not what the waves wait for)
#ifndef BP_BWD_NSTAGE
#define BP_BWD_NSTAGE 2
#endif
// (the two passes of a paired causal workgroup chained into one tile stream, so that the ring never drains between
// them: correct and 7 % slower, scripts/probes/flash_bwd_chain)

template <int KD>
struct BwdCfg {
    static constexpr int NT = 256, NWAVE = 4, BT = 64;               // BT: rows of a streamed tile
    static constexpr int NSTAGE = KD <= 4 ? BP_BWD_NSTAGE : 2;       // (wide heads: 32 KB stages, two of them)
    static constexpr int NV = (KD + 1) / 2;                          // 32-wide blocks of the head dimension
    static constexpr int ROW = KD <= 4 ? 128 : 256;                  // bytes per tile row in LDS (power of two)
    static constexpr int SLOTS = ROW / 16;
    static constexpr int TILE = BT * ROW;
    static constexpr int DMA = TILE / 1024 / NWAVE;                  // 1-KiB pieces per wave per tile: 2 or 4
    static constexpr int ROWS_PER_DMA = 1024 / ROW;
};

// Slot swizzle of the single LDS image (see the header).  Row bits 0..3 only: +16 / +32 rows keep a lane's swizzle.
//   ROW = 128 (two rows per 256-byte bank line): rows of equal parity r = 2j + p need 8 different slots -> a
//     bijection of j; a transposing pass reads rows 4m .. 4m+3, of which 4m and 4m+2 share a bank half -> bit 2 of
//     the swizzle (the 64-byte chunk select) must differ between j = 2m and 2m + 1: swz = (j & 1) << 2 | j >> 1.
//   ROW = 256 (one row per bank line): 16 rows need 16 different slots, 4 consecutive rows 4 different quarters:
//     swz = (r & 3) << 2 | (r >> 2) & 3.
template <int ROW> BP_DEV int bwd_swz(int row) {
    return ROW == 128 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : (((row & 3) << 2) | ((row >> 2) & 3));
}

// per-lane source descriptor of 1-KiB piece j of a tile: tile row and first element column of the lane's 16 bytes.
// Piece j of wave w is piece j * NWAVE + w of the tile: a wave's pieces lie PIECE_ROWS (16 or 32) rows apart, the swizzle
// only looks at row bits 0..3, so all of them share ONE per-lane byte offset and the piece index moves into the scalar
// base pointer (piece_base) -- one offset register per streamed tensor instead of DMA of them.
template <class C> BP_DEV void piece(int wave, int lane, int &row, int &col) {
    row = wave * C::ROWS_PER_DMA + lane / C::SLOTS;
    col = ((lane % C::SLOTS) ^ bwd_swz<C::ROW>(row)) * 8;
}
template <class C> BP_DEV const uint16_t *piece_base(const uint16_t *tile, int j, int64_t row_stride) {
    return tile + (int64_t)(j * C::NWAVE * C::ROWS_PER_DMA) * row_stride;
}
template <class C> BP_DEV int piece_lds(int wave, int j) { return (j * C::NWAVE + wave) * 1024; }

// lane offsets of the two read patterns
template <class C> BP_DEV int row_read_off(int l31, int hh, int s) {
    return l31 * C::ROW + (((2 * s + hh) ^ bwd_swz<C::ROW>(l31)) * 16);
}
// transposing read of head-dim block n, rows (4 hh + quad row) + 8 h: see lds_read_tr16_8B
template <class C> BP_DEV int tr_read_off(int lane, int n, int h) {
    const int row = 4 * (lane >> 5) + ((lane & 15) >> 2) + 8 * h;
    const int slot = 4 * n + ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    return row * C::ROW + ((slot ^ bwd_swz<C::ROW>(row)) * 16) + (lane & 1) * 8;
}

// wait until at most `ahead` tiles of PER DMA instructions each are still in flight (ahead: wave-uniform, 0..2)
template <int PER> BP_DEV void ring_wait(int ahead) {
    if (ahead <= 0) wait_vmcnt<0>();
    else if (ahead == 1) wait_vmcnt<PER>();
    else wait_vmcnt<2 * PER>();
}

// Epilogue store of one accumulator block (rows d = 32 n + 8 g + 4 hh + i, column = lane l31) as 16-byte pieces: the
// two half-waves own adjacent 8-byte pieces of every 16; v_permlane32_swap hands lanes 0..31 both pieces of the even
// groups and lanes 32..63 both pieces of the odd groups.  `row` points at this lane's (key / query) row, element d = 0.
// EXCHANGE = false: plain 8-byte pieces.  The kernels for head dims > 64 take that form: their dropout variants need
// more than 256 registers (accumulators in AGPRs), and there the exchange form produced NaN dK / dV and memory faults
// (r03_q: S = 200, d = 80, causal, p = 0.17; every other variant and the 8-byte form were right) -- a code-generation
// problem around v_permlane32_swap that was not chased further.
template <class E, bool EXCHANGE>
BP_DEV void store_block16(uint16_t *row, const f32x16 &acc, float scale, int n, int hh, int d) {
    if constexpr (!EXCHANGE) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = n * 32 + 8 * g + 4 * hh;
            if (d0 < d) {
                u32x2 a = {E::pack2(acc[4 * g] * scale, acc[4 * g + 1] * scale),
                           E::pack2(acc[4 * g + 2] * scale, acc[4 * g + 3] * scale)};
                *reinterpret_cast<u32x2 *>(row + d0) = a;
            }
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
        uint32_t a0 = E::pack2(acc[4 * g] * scale, acc[4 * g + 1] * scale);
        uint32_t a1 = E::pack2(acc[4 * g + 2] * scale, acc[4 * g + 3] * scale);
        uint32_t b0 = E::pack2(acc[4 * g + 4] * scale, acc[4 * g + 5] * scale);
        uint32_t b1 = E::pack2(acc[4 * g + 6] * scale, acc[4 * g + 7] * scale);
        auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const uint32_t w0 = r0[0], w1 = r1[0], w2 = r0[1], w3 = r1[1];   // by-value copies (bp_common.h, as_f32)
        const int d0 = n * 32 + 8 * (g + hh);
        if (d0 < d) *reinterpret_cast<u32x4 *>(row + d0) = u32x4{w0, w1, w2, w3};
    }
}

struct SeqInfo {
    int seq_q, seq_k;
    int64_t q_row0, k_row0;
};
BP_DEV SeqInfo seq_info(const FlashBwdParams &p, int batch) {
    SeqInfo s;
    if (p.cu_q != nullptr) {
        const int a = p.cu_q[batch], b = p.cu_q[batch + 1];
        const int c = p.cu_k[batch], d = p.cu_k[batch + 1];
        s.seq_q = b - a; s.seq_k = d - c; s.q_row0 = a; s.k_row0 = c;
    } else {
        s.seq_q = p.max_sq; s.seq_k = p.max_sk;
        s.q_row0 = (int64_t)batch * p.max_sq; s.k_row0 = (int64_t)batch * p.max_sk;
    }
    return s;
}

}  // namespace

// statistics workspace: per (batch, head) two rows of lse_stride floats: -D, then -L / scale
BP_DEV float *stats_row(const FlashBwdParams &p, int batch, int head) {
    return p.dsum + ((int64_t)batch * p.h + head) * 2 * p.lse_stride;
}

// =====================================================================================================
// dK, dV
// =====================================================================================================
template <int KD>
struct DkdvCfg {
    using C = BwdCfg<KD>;
    // stage = Q image | dO image | 64 x -D | 64 x -L/scale
    static constexpr int Q_OFF = 0, DO_OFF = C::TILE, D_OFF = 2 * C::TILE, L_OFF = 2 * C::TILE + 256;
    static constexpr int STAGE = 2 * C::TILE + 512;
    static constexpr int SMEM = C::NSTAGE * STAGE;
};

// one 128-key tile `kt` of (sample, head) `bh`
template <class ET, int KD, bool FULLD, bool DROP>
BP_DEV void flash_bwd_dkdv_tile(const FlashBwdParams p, char *smem, const uint32_t lds0, const int wave, const int bh,
                                  const int kt, const int pass) {
    BWD_STAMP(1, pass, 0);
    using C = BwdCfg<KD>;
    using G = DkdvCfg<KD>;
    using E = Elem<ET>;
    constexpr int NV = C::NV;

    // (the lane index is recomputed per pass, not `threadIdx.x & 63`, and the wave index arrives as a scalar: everything
    // derived from threadIdx.x would otherwise be hoisted in front of the two passes of a paired workgroup and live --
    // spilled -- across the first pass's tile loops)
    const int lane = lane_id_now();
    const int tid = wave * 64 + lane;
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    const int batch = bh / p.h;
    const int head = bh - batch * p.h;
    const SeqInfo si = seq_info(p, batch);
    const int seq_q = si.seq_q, seq_k = si.seq_k;
    if (kt * 128 >= seq_k) return;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + si.q_row0 * p.q_rs + (int64_t)head * p.q_hs;
    const uint16_t *dog = reinterpret_cast<const uint16_t *>(p.dout) + si.q_row0 * p.do_rs + (int64_t)head * p.do_hs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + si.k_row0 * p.k_rs + (int64_t)head * p.k_hs;
    const uint16_t *vg = reinterpret_cast<const uint16_t *>(p.v) + si.k_row0 * p.v_rs + (int64_t)head * p.v_hs;
    const float *stats_g = stats_row(p, batch, head);

    const int key0 = kt * 128 + wave * 32;        // first key of this wave
    const int my_key = key0 + l31;
    const bool wave_has_keys = key0 < seq_k;
    const float c2 = p.scale * kLog2e;

    // query tiles that can see this workgroup's keys: causal -> queries >= first key
    const int qt_begin = p.causal ? (kt * 128) / C::BT : 0;
    const int nqt = (seq_q + C::BT - 1) / C::BT;
    if (qt_begin >= nqt) {
        // no query sees these keys (causal, seq_k > seq_q): their gradients are zero
        if (wave_has_keys && my_key < seq_k) {
            uint16_t *dkg = reinterpret_cast<uint16_t *>(p.dk) + (si.k_row0 + my_key) * p.dk_rs + (int64_t)head * p.dk_hs;
            uint16_t *dvg = reinterpret_cast<uint16_t *>(p.dv) + (si.k_row0 + my_key) * p.dv_rs + (int64_t)head * p.dv_hs;
            for (int d0 = 4 * hh; d0 < p.d; d0 += 8) {
                *reinterpret_cast<u32x2 *>(dkg + d0) = u32x2{0u, 0u};
                *reinterpret_cast<u32x2 *>(dvg + d0) = u32x2{0u, 0u};
            }
        }
        return;
    }

    if (!FULLD) {   // pad slots of the images must read as 0 (they meet zero K / V columns in the MFMAs)
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < G::SMEM; off += C::NT * 16) lds_write_16B(smem, off, z);
        __syncthreads();
    }

    DropoutStream rng = {0u, 0u};
    if (DROP) rng = dropout_stream(p.rng_state, (uint32_t)bh);

    // ---- K and V fragments of my 32 keys: B operands (lane = key, 8 consecutive d).  Requested here, awaited behind
    //      the first tile's DMA issue: the two latencies overlap (r03_n timeline: 3.9 k + 8 k ticks in sequence before)
    u32x4 kf[KD], vf[KD];
    {
        const int key = min(my_key, seq_k - 1);
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
            if (FULLD || col < p.d) {
                a = ld_global_16B(kg + (int64_t)key * p.k_rs + col);
                b = ld_global_16B(vg + (int64_t)key * p.v_rs + col);
            }
            kf[s] = a;
            vf[s] = b;
        }
    }

    // ---- DMA: constant per-lane byte offsets, scalar tile pointers --------------------------------------
    const int qt_partial = (seq_q % C::BT) != 0 ? seq_q / C::BT : -1;
    const int last_row = seq_q - 1 - (seq_q / C::BT) * C::BT;
    // (the clamped offsets of the one partial tile a sweep can meet are recomputed from the lane's row in that cold
    // step instead of living in registers all along: four registers the three-waves-per-SIMD bound does not have)
    uint32_t q_voff, do_voff;
    bool piece_live;
    {
        int row, col;
        piece<C>(wave, lane, row, col);
        piece_live = FULLD || col < p.d;
        q_voff = (uint32_t)(row * p.q_rs + col) * 2u;
        do_voff = (uint32_t)(row * p.do_rs + col) * 2u;
    }
    // (a scalar int, not the bool `wave < 2`: hipcc kept a per-lane copy of the bool alive across the pass -- one more spill)
    int stats_wave = 1 - (wave >> 1);
    asm volatile("" : "+s"(stats_wave));
    const int64_t q_tile_stride = (int64_t)C::BT * p.q_rs, do_tile_stride = (int64_t)C::BT * p.do_rs;
    const uint16_t *qt_ptr = qg + (int64_t)qt_begin * q_tile_stride;     // tile of the NEXT issue
    const uint16_t *dot_ptr = dog + (int64_t)qt_begin * do_tile_stride;
    auto issue = [&](int qt) {
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + ((qt - qt_begin) % C::NSTAGE) * G::STAGE);
        if (piece_live) {
            if (__builtin_expect(qt == qt_partial, 0)) {
#pragma unroll
                for (int j = 0; j < C::DMA; ++j) {
                    // rows past the sequence's last one re-read it; the offset is taken from the TILE base here (from
                    // the piece base it could come out negative, and the instruction adds it unsigned)
                    int row, col;
                    piece<C>(wave, lane_id_now(), row, col);
                    row = min(row + j * C::NWAVE * C::ROWS_PER_DMA, last_row);
                    dma16_s(qt_ptr, (uint32_t)(row * p.q_rs + col) * 2u,
                            __builtin_amdgcn_readfirstlane(st + G::Q_OFF + piece_lds<C>(wave, j)));
                    dma16_s(dot_ptr, (uint32_t)(row * p.do_rs + col) * 2u,
                            __builtin_amdgcn_readfirstlane(st + G::DO_OFF + piece_lds<C>(wave, j)));
                }
            } else {
#pragma unroll
                for (int j = 0; j < C::DMA; ++j) {
                    dma16_s(piece_base<C>(qt_ptr, j, p.q_rs), q_voff,
                            __builtin_amdgcn_readfirstlane(st + G::Q_OFF + piece_lds<C>(wave, j)));
                    dma16_s(piece_base<C>(dot_ptr, j, p.do_rs), do_voff,
                            __builtin_amdgcn_readfirstlane(st + G::DO_OFF + piece_lds<C>(wave, j)));
                }
            }
        }
        // the tile's 64 x -D (wave 0) and 64 x -L/scale (wave 1); rows past the sequence: anything, they are masked
        if (stats_wave) {
            const float *src = stats_g + wave * p.lse_stride + min(qt * C::BT + lane_id_now(), (int)p.lse_stride - 1);
            dma4(src, st + G::D_OFF + wave * 256);
        }
        qt_ptr += q_tile_stride;
        dot_ptr += do_tile_stride;
    };

    f32x16 dk[NV], dv[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[n][r] = 0.f; dv[n][r] = 0.f; }

    int r_off[KD];   // A operand rows (lane = query l31)
#pragma unroll
    for (int s = 0; s < KD; ++s) r_off[s] = row_read_off<C>(l31, hh, s);
    int t_off[NV][2];   // transposing reads: head-dim block n, rows +0 / +8
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        t_off[n][0] = tr_read_off<C>(lane, n, 0);
        t_off[n][1] = tr_read_off<C>(lane, n, 1);
    }

    // One 32-query sub-block `qb` of the tile in `st`.  EDGE: mask what is not a (query, key) pair of the problem.
    auto sub_block = [&](const char *st, int qt, int qb, auto EDGE) {
        constexpr bool kEdge = decltype(EDGE)::value;
        const int qbase = qt * C::BT + qb * 32;          // first query of this sub-block
        // ---- S = Q K^T - L/scale and dP = dO V^T - D : rows = queries (registers), column = my key; the row
        //      constants are the accumulators' initial values
        f32x16 s_, dp;
        float dneg[DROP ? 16 : 1];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const u32x4 l4 = lds_read_16B(st, G::L_OFF + (qb * 32 + 8 * g + 4 * hh) * 4);
            const u32x4 d4 = lds_read_16B(st, G::D_OFF + (qb * 32 + 8 * g + 4 * hh) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t lw = l4[i], dw = d4[i];   // by-value copies (bp_common.h, as_f32)
                s_[4 * g + i] = as_f32(lw);
                if (DROP) { dp[4 * g + i] = 0.f; dneg[4 * g + i] = as_f32(dw); }
                else dp[4 * g + i] = as_f32(dw);
            }
        }
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            if (kWhatIf & 2) break;
            const u32x4 a = lds_read_16B(st, G::Q_OFF + r_off[s] + qb * 32 * C::ROW);
            s_ = E::mfma(a, kf[s], s_);
            const u32x4 b = lds_read_16B(st, G::DO_OFF + r_off[s] + qb * 32 * C::ROW);
            dp = E::mfma(b, vf[s], dp);
        }
        // ---- P = exp2((S - L/scale) c), dS = P (dP - D) ----------------------------------------------------
        uint32_t keep = 0xffffu;   // bit 4g+i: query qbase + 8g + 4hh + i keeps my key
        if (DROP) keep = dropout_keep_collane(rng, p.drop_thr, (uint32_t)qbase, (uint32_t)my_key, hh);
        // register r holds query qbase + (r & 3) + 8 (r >> 2) + 4 hh: dead iff that is before my key (causal), past
        // the sequence, or my key does not exist -> two per-lane limits against compile-time constants
        int lim_lo = 0, lim_hi = 64;
        if (kEdge) {
            lim_lo = p.causal ? my_key - qbase - 4 * hh : 0;
            lim_hi = seq_q - qbase - 4 * hh;
            if (my_key >= seq_k) lim_hi = 0;
        }
        u32x4 pf[2], dsf[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float pe[4], de[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const float pv = (kWhatIf & 1) ? s_[r] * c2 : fast_exp2(s_[r] * c2);
                if (DROP) {
                    const float z = ((keep >> r) & 1u) ? p.drop_scale : 0.f;
                    pe[i] = pv * z;
                    de[i] = pv * fmaf(dp[r], z, dneg[r]);
                } else {
                    pe[i] = pv;
                    de[i] = pv * dp[r];
                }
                if (kEdge) {
                    // selects, not multiplies: the statistics of rows past the sequence are uninitialised (maybe NaN)
                    const int c = i + 8 * g;
                    const bool dead = c < lim_lo || c >= lim_hi;
                    pe[i] = dead ? 0.f : pe[i];
                    de[i] = dead ? 0.f : de[i];
                }
            }
            // regs 8*ks .. 8*ks+7 are the B operand of K-step ks (queries {0..3, 8..11} + 4hh + 16ks)
            pf[g >> 1][(g & 1) * 2 + 0] = E::pack2(pe[0], pe[1]);
            pf[g >> 1][(g & 1) * 2 + 1] = E::pack2(pe[2], pe[3]);
            dsf[g >> 1][(g & 1) * 2 + 0] = E::pack2(de[0], de[1]);
            dsf[g >> 1][(g & 1) * 2 + 1] = E::pack2(de[2], de[3]);
        }
        if (kWhatIf & 32) {   // no softmax VALU: the raw accumulator bits stand in for the packed operands
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) { pf[ks][i] = as_u32(s_[ks * 8 + i]); dsf[ks][i] = as_u32(dp[ks * 8 + i]); }
        }
        if (kWhatIf & 4) asm volatile("" ::"v"(pf[0]), "v"(pf[1]), "v"(dsf[0]), "v"(dsf[1]));
        // ---- dV^T += dO^T P ; dK^T += Q^T dS   (contraction over the 32 queries) -------------------
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (kWhatIf & 4) break;
            const int rows = (qb * 32 + ks * 16) * C::ROW;
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const u32x2 lo = lds_read_tr16_8B(st, G::DO_OFF + t_off[n][0] + rows);
                const u32x2 hi = lds_read_tr16_8B(st, G::DO_OFF + t_off[n][1] + rows);
                dv[n] = E::mfma(u32x4{lo[0], lo[1], hi[0], hi[1]}, pf[ks], dv[n]);
                const u32x2 lo2 = lds_read_tr16_8B(st, G::Q_OFF + t_off[n][0] + rows);
                const u32x2 hi2 = lds_read_tr16_8B(st, G::Q_OFF + t_off[n][1] + rows);
                dk[n] = E::mfma(u32x4{lo2[0], lo2[1], hi2[0], hi2[1]}, dsf[ks], dk[n]);
            }
        }
    };

    // ring step: my share of tile qt has landed (the younger tiles may still be in flight: every tile costs a wave
    // the same number of DMA instructions, so the wait is a count), everybody's after the barrier; refill the slot
    // that was read during the previous step
    auto step_begin = [&](int qt) -> const char * {
        const int ahead = min(nqt - 1 - qt, C::NSTAGE - 2);   // tiles requested after tile qt
        if (stats_wave) ring_wait<2 * C::DMA + 1>(ahead);
        else ring_wait<2 * C::DMA>(ahead);
        if (!(kWhatIf & 8)) __builtin_amdgcn_s_barrier();
        if (qt + C::NSTAGE - 1 < nqt) issue(qt + C::NSTAGE - 1);
        return smem + ((qt - qt_begin) % C::NSTAGE) * G::STAGE;
    };
    // a tile near the diagonal or the sequence end: per sub-block skip / masked body
    auto edge_tile = [&](int qt) {
        const char *st = step_begin(qt);
        if (!wave_has_keys) return;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qbase = qt * C::BT + qb * 32;
            if (qbase >= seq_q) continue;
            if (p.causal && qbase + 31 < key0) continue;     // every query is before my first key
            sub_block(st, qt, qb, std::true_type{});
        }
    };
    // (the two sub-blocks woven inside the wave -- S/dP of B under A's softmax, dV/dK of A under B's -- needs 212
    // VGPRs, i.e. two waves per SIMD, and measured 3-5 % slower than this plain body at three: scripts/probes/flash_bwd_woven)
    auto clean_tile = [&](int qt) {
        const char *st = step_begin(qt);
        sub_block(st, qt, 0, std::false_type{});
        sub_block(st, qt, 1, std::false_type{});
    };

    // Three sequential loops, one body each (an if/else join of the 2 x NV accumulator sets inside ONE loop makes the
    // register allocator copy or spill them, DESIGN.md "compiler findings"):
    //   edge tiles up to the one that holds my diagonal | clean tiles | the sequence's last, partial tile(s).
    // A wave without keys, or whose keys run past the sequence, takes the edge body throughout.
    int clean_begin = nqt, clean_end = nqt;
    if (wave_has_keys && key0 + 32 <= seq_k) {
        // first tile whose every query is at or after my last key (key0 is a multiple of 32) ... last full tile
        clean_begin = p.causal ? max(qt_begin, (key0 + 31 + C::BT) / C::BT) : qt_begin;
        clean_end = seq_q / C::BT;
    }
    clean_begin = min(clean_begin, nqt);
    clean_end = min(max(clean_end, clean_begin), nqt);

    BWD_STAMP(1, pass, 2);   // descriptors done
    for (int t = 0; t < C::NSTAGE - 1; ++t)
        if (qt_begin + t < nqt) issue(qt_begin + t);
#pragma unroll
    for (int s = 0; s < KD; ++s) { settle(kf[s]); settle(vf[s]); }   // see bp_common.h: no vmcnt(0) in the loop
    BWD_STAMP(1, pass, 1);   // K / V fragments arrived
    int qt = qt_begin;
    for (; qt < clean_begin; ++qt) { edge_tile(qt); if (qt == qt_begin) BWD_STAMP(1, pass, 3); }
    BWD_STAMP(1, pass, 4);   // leading edge tiles done
    for (; qt < clean_end; ++qt) clean_tile(qt);
    BWD_STAMP(1, pass, 5);   // clean tiles done
    for (; qt < nqt; ++qt) edge_tile(qt);
    BWD_STAMP(1, pass, 6);   // all tiles done

    if (!wave_has_keys) return;
    int key_row = min(my_key, seq_k - 1);
    // (opaque here: hipcc otherwise forms the two 64-bit output row pointers in front of the tile loops and carries --
    // spills -- them across the clean loop)
    asm volatile("" : "+v"(key_row));
    uint16_t *dkg = reinterpret_cast<uint16_t *>(p.dk) + (si.k_row0 + key_row) * p.dk_rs + (int64_t)head * p.dk_hs;
    uint16_t *dvg = reinterpret_cast<uint16_t *>(p.dv) + (si.k_row0 + key_row) * p.dv_rs + (int64_t)head * p.dv_hs;
    const int d_lim = my_key < seq_k ? p.d : 0;   // lanes past the sequence exchange, but store nothing
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        if (kWhatIf & 16) { asm volatile("" ::"v"(dk[n]), "v"(dv[n])); continue; }
        store_block16<E, (KD <= 4)>(dkg, dk[n], p.scale, n, hh, d_lim);
        store_block16<E, (KD <= 4)>(dvg, dv[n], 1.f, n, hh, d_lim);
    }
    BWD_STAMP(1, pass, 7);   // stores issued
}

// =====================================================================================================
// dQ
// =====================================================================================================
// one 128-query tile `qt` of (sample, head) `bh`
template <class ET, int KD, bool FULLD, bool DROP>
BP_DEV void flash_bwd_dq_tile(const FlashBwdParams p, char *smem, const uint32_t lds0, const int wave, const int bh,
                                const int qt, const int pass) {
    BWD_STAMP(0, pass, 0);
    using C = BwdCfg<KD>;
    using E = Elem<ET>;
    constexpr int NV = C::NV;
    // stage = K image | V image
    constexpr int K_OFF = 0, V_OFF = C::TILE, STAGE = 2 * C::TILE;

    // (the lane index is recomputed per pass, not `threadIdx.x & 63`, and the wave index arrives as a scalar: everything
    // derived from threadIdx.x would otherwise be hoisted in front of the two passes of a paired workgroup and live --
    // spilled -- across the first pass's tile loops)
    const int lane = lane_id_now();
    const int tid = wave * 64 + lane;
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    const int batch = bh / p.h;
    const int head = bh - batch * p.h;
    const SeqInfo si = seq_info(p, batch);
    const int seq_q = si.seq_q, seq_k = si.seq_k;
    if (qt * 128 >= seq_q) return;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + si.q_row0 * p.q_rs + (int64_t)head * p.q_hs;
    const uint16_t *dog = reinterpret_cast<const uint16_t *>(p.dout) + si.q_row0 * p.do_rs + (int64_t)head * p.do_hs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + si.k_row0 * p.k_rs + (int64_t)head * p.k_hs;
    const uint16_t *vg = reinterpret_cast<const uint16_t *>(p.v) + si.k_row0 * p.v_rs + (int64_t)head * p.v_hs;

    int k_end = seq_k;
    if (p.causal) k_end = min(seq_k, qt * 128 + 128);
    const int nkb = (k_end + C::BT - 1) / C::BT;

    const int q0 = qt * 128 + wave * 32;
    const int my_q = q0 + l31;
    const bool wave_has_rows = q0 < seq_q;
    const float c2 = p.scale * kLog2e;

    if (!FULLD) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < C::NSTAGE * STAGE; off += C::NT * 16) lds_write_16B(smem, off, z);
        __syncthreads();
    }

    DropoutStream rng = {0u, 0u};
    if (DROP) rng = dropout_stream(p.rng_state, (uint32_t)bh);

    // my row's Q, dO, O fragments and L: requested here, consumed behind the first tile's DMA issue
    u32x4 qf[KD], dof[KD], of[KD];
    float lse_row;
    {
        const int q = min(my_q, seq_q - 1);
        const uint16_t *og = reinterpret_cast<const uint16_t *>(p.out) + (si.q_row0 + q) * p.o_rs + (int64_t)head * p.o_hs;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u}, o = {0u, 0u, 0u, 0u};
            if (FULLD || col < p.d) {
                a = ld_global_16B(qg + (int64_t)q * p.q_rs + col);
                b = ld_global_16B(dog + (int64_t)q * p.do_rs + col);
                o = ld_global_16B(og + col);
            }
            qf[s] = a;
            dof[s] = b;
            of[s] = o;
        }
        lse_row = p.lse[((int64_t)batch * p.h + head) * p.lse_stride + q];
    }
    // row constants, filled in behind the first DMA issue (below): -L / scale, -D of my row; -D as the MFMA C operand
    // of the first K-step of every dP chain (never written: the chains start from it); -L is a per-lane scalar here
    // and rides the exponent's fma for free
    float lneg = 0.f, dneg = 0.f, lneg2 = 0.f;
    f32x16 c_d;

    // ---- DMA -----------------------------------------------------------------------------------------------
    const int kb_partial = (seq_k % C::BT) != 0 ? seq_k / C::BT : -1;
    const int last_row = seq_k - 1 - (seq_k / C::BT) * C::BT;
    uint32_t k_voff, v_voff;   // (one offset per tensor, see piece(); partial tile: clamped in its own cold step)
    bool piece_live;
    {
        int row, col;
        piece<C>(wave, lane, row, col);
        piece_live = FULLD || col < p.d;
        k_voff = (uint32_t)(row * p.k_rs + col) * 2u;
        v_voff = (uint32_t)(row * p.v_rs + col) * 2u;
    }
    const int64_t k_tile_stride = (int64_t)C::BT * p.k_rs, v_tile_stride = (int64_t)C::BT * p.v_rs;
    const uint16_t *kt_ptr = kg, *vt_ptr = vg;
    auto issue = [&](int kb) {
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + (kb % C::NSTAGE) * STAGE);
        if (piece_live) {
            if (__builtin_expect(kb == kb_partial, 0)) {
#pragma unroll
                for (int j = 0; j < C::DMA; ++j) {
                    int row, col;   // (offset from the TILE base, see the dK/dV kernel)
                    piece<C>(wave, lane_id_now(), row, col);
                    row = min(row + j * C::NWAVE * C::ROWS_PER_DMA, last_row);
                    dma16_s(kt_ptr, (uint32_t)(row * p.k_rs + col) * 2u,
                            __builtin_amdgcn_readfirstlane(st + K_OFF + piece_lds<C>(wave, j)));
                    dma16_s(vt_ptr, (uint32_t)(row * p.v_rs + col) * 2u,
                            __builtin_amdgcn_readfirstlane(st + V_OFF + piece_lds<C>(wave, j)));
                }
            } else {
#pragma unroll
                for (int j = 0; j < C::DMA; ++j) {
                    dma16_s(piece_base<C>(kt_ptr, j, p.k_rs), k_voff,
                            __builtin_amdgcn_readfirstlane(st + K_OFF + piece_lds<C>(wave, j)));
                    dma16_s(piece_base<C>(vt_ptr, j, p.v_rs), v_voff,
                            __builtin_amdgcn_readfirstlane(st + V_OFF + piece_lds<C>(wave, j)));
                }
            }
        }
        kt_ptr += k_tile_stride;
        vt_ptr += v_tile_stride;
    };

    f32x16 dq[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[n][r] = 0.f;

    int r_off[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) r_off[s] = row_read_off<C>(l31, hh, s);
    int t_off[NV][2];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        t_off[n][0] = tr_read_off<C>(lane, n, 0);
        t_off[n][1] = tr_read_off<C>(lane, n, 1);
    }

    // one 32-key sub-block kk of the tile in `st`
    auto sub_block = [&](const char *st, int kb, int kk, auto EDGE) {
        constexpr bool kEdge = decltype(EDGE)::value;
        const int kbase = kb * C::BT + kk * 32;
        // S^T = K Q^T and dP^T = V dO^T - D : rows = keys (registers), column = my query
        f32x16 st_, dpt = c_d;
#pragma unroll
        for (int r = 0; r < 16; ++r) st_[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            if (kWhatIf & 2) break;
            const u32x4 a = lds_read_16B(st, K_OFF + r_off[s] + kk * 32 * C::ROW);
            st_ = E::mfma(a, qf[s], st_);
            const u32x4 b = lds_read_16B(st, V_OFF + r_off[s] + kk * 32 * C::ROW);
            dpt = E::mfma(b, dof[s], dpt);
        }
        uint32_t keep = 0xffffu;   // bit 4g+i: my query keeps key kbase + 8g + 4hh + i
        if (DROP) keep = dropout_keep_rowlane(rng, p.drop_thr, (uint32_t)my_q, (uint32_t)kbase, hh);
        // register r holds key kbase + (r & 3) + 8 (r >> 2) + 4 hh: dead iff beyond the last key my row may see
        int lim = 64;
        if (kEdge) {
            int last = seq_k - 1;
            if (p.causal) last = min(last, my_q);
            lim = last - kbase - 4 * hh;
        }
        u32x4 dsf[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float de[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const float pv = (kWhatIf & 1) ? fmaf(st_[r], c2, lneg2) : fast_exp2(fmaf(st_[r], c2, lneg2));
                if (DROP) {
                    const float z = ((keep >> r) & 1u) ? p.drop_scale : 0.f;
                    de[i] = pv * fmaf(dpt[r], z, dneg);
                } else {
                    de[i] = pv * dpt[r];
                }
                if (kEdge) de[i] = (i + 8 * g > lim) ? 0.f : de[i];
            }
            dsf[g >> 1][(g & 1) * 2 + 0] = E::pack2(de[0], de[1]);
            dsf[g >> 1][(g & 1) * 2 + 1] = E::pack2(de[2], de[3]);
        }
        if (kWhatIf & 32) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) dsf[ks][i] = as_u32(st_[ks * 8 + i]) ^ as_u32(dpt[ks * 8 + i]);
        }
        if (kWhatIf & 4) asm volatile("" ::"v"(dsf[0]), "v"(dsf[1]));
        // dQ^T += K^T dS^T  (contraction over the 32 keys)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (kWhatIf & 4) break;
            const int rows = (kk * 32 + ks * 16) * C::ROW;
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const u32x2 lo = lds_read_tr16_8B(st, K_OFF + t_off[n][0] + rows);
                const u32x2 hi = lds_read_tr16_8B(st, K_OFF + t_off[n][1] + rows);
                dq[n] = E::mfma(u32x4{lo[0], lo[1], hi[0], hi[1]}, dsf[ks], dq[n]);
            }
        }
    };

    auto step_begin = [&](int kb) -> const char * {
        ring_wait<2 * C::DMA>(min(nkb - 1 - kb, C::NSTAGE - 2));
        if (!(kWhatIf & 8)) __builtin_amdgcn_s_barrier();
        if (kb + C::NSTAGE - 1 < nkb) issue(kb + C::NSTAGE - 1);
        return smem + (kb % C::NSTAGE) * STAGE;
    };
    // Which key tiles this wave computes, and which of them the clean body may take: every key exists and every
    // (query, key) pair of my 32 rows is visible (as in flash_fwd_dma.hip).
    const int my_nkb = !wave_has_rows ? 0 : p.causal ? min(nkb, (q0 + 31) / C::BT + 1) : nkb;
    const int my_clean_end = !wave_has_rows ? 0 : p.causal ? min(seq_k / C::BT, (q0 + 1) / C::BT) : seq_k / C::BT;

    BWD_STAMP(0, pass, 2);
    for (int t = 0; t < C::NSTAGE - 1; ++t)
        if (t < nkb) issue(t);

    {
        float part = 0.f;   // my 8*KD columns of dO . O
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const u32x4 b = dof[s], o = of[s];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t bw = b[i], ow = o[i];   // by-value copies (bp_common.h, as_f32)
                part = fmaf(E::lo_f32(bw), E::lo_f32(ow), part);
                part = fmaf(E::hi_f32(bw), E::hi_f32(ow), part);
            }
        }
        // (a row without keys has L = -inf and meets no key tile: any finite stand-in will do)
        lneg = lse_row == -INFINITY ? 0.f : -lse_row / p.scale;
        dneg = -xhalf_sum(part);   // the two half-waves hold the two 8-column halves of every 16
        if (hh == 0 && wave_has_rows && my_q < seq_q) {
            float *st = stats_row(p, batch, head);
            st[min(my_q, seq_q - 1)] = dneg;
            st[p.lse_stride + min(my_q, seq_q - 1)] = lneg;
        }
#pragma unroll
        for (int s = 0; s < KD; ++s) { settle(qf[s]); settle(dof[s]); }
        settle(lneg); settle(dneg);
    }
    BWD_STAMP(0, pass, 1);   // Q / dO / O / L arrived, D formed
#pragma unroll
    for (int r = 0; r < 16; ++r) c_d[r] = DROP ? 0.f : dneg;
    lneg2 = lneg * c2;
    int kb = 0;
    for (; kb < min(my_clean_end, nkb); ++kb) {
        const char *st = step_begin(kb);
        if (kb == 0) BWD_STAMP(0, pass, 3);   // first tile landed
        sub_block(st, kb, 0, std::false_type{});
        sub_block(st, kb, 1, std::false_type{});
    }
    BWD_STAMP(0, pass, 5);   // clean tiles done
    for (; kb < nkb; ++kb) {
        const char *st = step_begin(kb);
        if (kb >= my_nkb) continue;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kbase = kb * C::BT + kk * 32;
            if (kbase >= seq_k) continue;
            if (p.causal && kbase > q0 + 31) continue;
            sub_block(st, kb, kk, std::true_type{});
        }
    }
    BWD_STAMP(0, pass, 6);   // all tiles done

    if (!wave_has_rows) return;
    int q_row = min(my_q, seq_q - 1);
    asm volatile("" : "+v"(q_row));   // (as in the dK/dV epilogue: keep the output pointer out of the tile loops)
    uint16_t *dqg = reinterpret_cast<uint16_t *>(p.dq) + (si.q_row0 + q_row) * p.dq_rs + (int64_t)head * p.dq_hs;
    const int d_lim = my_q < seq_q ? p.d : 0;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        if (kWhatIf & 16) { asm volatile("" ::"v"(dq[n])); continue; }
        store_block16<E, (KD <= 4)>(dqg, dq[n], p.scale, n, hh, d_lim);
    }
    BWD_STAMP(0, pass, 7);
}


// Kernels: a causal workgroup takes the heaviest remaining tile and the lightest of its (sample, head) -- tiles t
// and n-1-t -- so that every workgroup carries the same work (in-order round-robin dispatch, see flash_fwd_dma.hip).
// (the dropout variants spill 500+ registers at three waves per SIMD: they keep two)
#ifndef BP_BWD_DKDV_MINWAVES
#define BP_BWD_DKDV_MINWAVES(KD, DROP) ((KD) <= 4 && !(DROP) ? 3 : (KD) <= 4 ? 2 : 1)
#endif

template <class ET, int KD, bool FULLD, bool DROP>
__global__ __launch_bounds__(256, BP_BWD_DKDV_MINWAVES(KD, DROP)) void flash_bwd_dkdv_kernel(const FlashBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem[DkdvCfg<KD>::SMEM];
    const uint32_t lds0 = lds_base_addr(smem);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = (p.max_sk + 127) / 128;
    const bool pair = p.causal && n > 1;
    int bh, slot;
    if (!xcd_map(blockIdx.x, p.b * p.h, pair ? (n + 1) / 2 : n, bh, slot)) return;   // key tile 0 = most work
    const int other = n - 1 - slot;
    const int npass = (pair && other != slot) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass) __syncthreads();
        flash_bwd_dkdv_tile<ET, KD, FULLD, DROP>(p, smem, lds0, wave, bh, pass ? other : slot, pass);
    }
}

// waves per SIMD the register allocator must leave room for in the dQ kernel (512 VGPRs per SIMD lane)
#ifndef BP_BWD_DQ_MINWAVES
#define BP_BWD_DQ_MINWAVES(KD) ((KD) <= 4 ? 3 : 2)
#endif

template <class ET, int KD, bool FULLD, bool DROP>
__global__ __launch_bounds__(256, BP_BWD_DQ_MINWAVES(KD)) void flash_bwd_dq_kernel(const FlashBwdParams p) {
    using C = BwdCfg<KD>;
    __shared__ __attribute__((aligned(16))) char smem[C::NSTAGE * 2 * C::TILE];
    const uint32_t lds0 = lds_base_addr(smem);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = (p.max_sq + 127) / 128;
    const bool pair = p.causal && n > 1;
    int bh, slot;
    if (!xcd_map(blockIdx.x, p.b * p.h, pair ? (n + 1) / 2 : n, bh, slot)) return;
    const int heavy = n - 1 - slot;
    const int npass = (pair && heavy != slot) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass) __syncthreads();
        flash_bwd_dq_tile<ET, KD, FULLD, DROP>(p, smem, lds0, wave, bh, pass ? slot : heavy, pass);
    }
}

template <class ET, int KD, bool FULLD, bool DROP>
static hipError_t launch_drop(const FlashBwdParams &p, hipStream_t stream) {
    // dq first: it also produces the row statistics the dkdv kernel consumes
    const int nq = (p.max_sq + 127) / 128, nk = (p.max_sk + 127) / 128;
    const int gq = xcd_grid(p.b * p.h, (p.causal && nq > 1) ? (nq + 1) / 2 : nq);
    hipLaunchKernelGGL((flash_bwd_dq_kernel<ET, KD, FULLD, DROP>), dim3(gq), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int gk = xcd_grid(p.b * p.h, (p.causal && nk > 1) ? (nk + 1) / 2 : nk);
    hipLaunchKernelGGL((flash_bwd_dkdv_kernel<ET, KD, FULLD, DROP>), dim3(gk), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <class ET, int KD>
static hipError_t launch_one(const FlashBwdParams &p, hipStream_t stream) {
    const bool drop = p.drop_thr != 0u;
    if constexpr (KD == 4 || KD == 8) {
        if (p.d == KD * 16)
            return drop ? launch_drop<ET, KD, true, true>(p, stream) : launch_drop<ET, KD, true, false>(p, stream);
    }
    return drop ? launch_drop<ET, KD, false, true>(p, stream) : launch_drop<ET, KD, false, false>(p, stream);
}

template <class ET>
static hipError_t launch_et(const FlashBwdParams &p, hipStream_t stream) {
    switch ((p.d + 15) / 16) {
        case 1: return launch_one<ET, 1>(p, stream);
        case 2: return launch_one<ET, 2>(p, stream);
        case 3: return launch_one<ET, 3>(p, stream);
        case 4: return launch_one<ET, 4>(p, stream);
        case 5: return launch_one<ET, 5>(p, stream);   // d_h = 80 (Mini)
        case 6: return launch_one<ET, 6>(p, stream);
        case 7: return launch_one<ET, 7>(p, stream);
        default: return launch_one<ET, 8>(p, stream);
    }
}

// head_dim % 8 == 0 and <= 128 (the trunk's 64 / 80 and the senses' 48/40/24), 16-byte friendly strides.
hipError_t launch_flash_bwd(const FlashBwdParams &p, int dtype, hipStream_t stream) {
    if (p.d > 128) return hipErrorNotSupported;
    return dtype == 1 ? launch_et<BF16>(p, stream) : launch_et<F16>(p, stream);
}

#ifdef BP_BWD_PROFILE
extern "C" int bp_dev_bwd_prof(unsigned long long *host, int clear) {
    if (clear) {
        void *ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_bwd_prof)) != hipSuccess) return -1;
        return hipMemset(ptr, 0, sizeof(g_bwd_prof)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_prof), sizeof(g_bwd_prof)) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace bp
