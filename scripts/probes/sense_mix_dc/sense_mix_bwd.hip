// Gradient of the fused sense contraction with respect to the CONTENT:
//
//     dC[b,s,l,:] = w[b,l,s] * sum_{t>=s} P_l[t,s] * dout[b,t,:],     P_l[t,s] = exp(scale q_l[t].k_l[s] - lse[b,l,t])
//
// It is the forward kernel (sense_mix_dma.hip) with queries and keys swapped: a wave owns 32 KEYS (their k_l
// fragments live in registers as the MFMA B operand), the workgroup owns 256 keys x 256 output columns of one sense,
// and it streams 64-QUERY tiles -- q_l rows, the matching dout rows and their log-sum-exp -- through the same
// 3-slot LDS-DMA ring.  S = Q K^T comes out with lane = key and the queries along the registers, which is the B
// layout of dC^T += dout^T P (A operand = transposing LDS reads of the dout tile): P never leaves registers and
// alpha is not materialised.  Causal work shrinks with the key tile (tile 0 sees every query, the last one 256),
// so a workgroup takes key tiles kt and n-1-kt (equal work per workgroup, see flash_fwd_dma.hip on the dispatcher).
#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"

namespace bp {

namespace {
template <int KD>
struct DcCfg {
    static constexpr int BKEY = 256, BQ = 64, NB = 8, BNC = 256, NT = 512, NWAVE = 8, NSTAGE = 3;
    static constexpr int QROW = 128;                 // bytes per q row image (d_k <= 64), XOR-swizzled slots
    static constexpr int GROW = 512;                 // bytes per dout row image (256 columns)
    static constexpr int QTILE = BQ * QROW;          // 8 KiB
    static constexpr int GTILE = BQ * GROW;          // 32 KiB
    static constexpr int LTILE = NWAVE * 256;        // per wave: the tile's 64 log-sum-exp values
    static constexpr int STAGE = QTILE + GTILE + LTILE;
    static constexpr int G_DMA = GTILE / 1024 / NWAVE;           // 4
    static constexpr int DMA_PER_STAGE = 1 + G_DMA + 1;          // q piece, dout pieces, lse piece
    static constexpr int SMEM = NSTAGE * STAGE;
};
}  // namespace

template <class ET, int KD, bool FULL>
BP_DEV void sense_mix_dc_tile(const MixBwdParams p, char *smem, const uint32_t lds0, const int grp, const int kt) {
    using C = DcCfg<KD>;
    using E = Elem<ET>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int S = p.s;
    if (kt * C::BKEY >= S) return;

    const int chunk = grp % p.n_chunks;
    const int bl = grp / p.n_chunks;                 // batch * nsenses + sense
    const int batch = bl / p.nsenses, l = bl - batch * p.nsenses;
    const int col_base = chunk * C::BNC;
    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.qk_bs + (int64_t)l * p.qk_ss;
    const uint16_t *gg = reinterpret_cast<const uint16_t *>(p.g) + batch * p.g_bs;
    const float *lse_g = p.lse + (int64_t)bl * p.lse_stride;

    const int key0 = kt * C::BKEY + wave * 32;
    const int my_key = key0 + l31;
    const bool wave_has_keys = key0 < S;
    const float c2 = p.scale_log2e;
    const int nb_live = FULL ? C::NB : min(C::NB, (p.dout - col_base + 31) / 32);

    // never-written pad slots of the row images must read as finite zeros
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int off = tid * 16; off < C::SMEM; off += C::NT * 16) lds_write_16B(smem, off, z);
    }
    __syncthreads();

    // my keys' fragments: B operand of S = Q K^T (lane = key, 8 consecutive d_k)
    u32x4 kf[KD];
    {
        const uint16_t *row = kg + (int64_t)min(my_key, S - 1) * p.qk_rs;
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            const int col = 16 * s + 8 * hh;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (col < p.dk) v = ld_global_16B(row + col);
            kf[s] = v;
        }
#pragma unroll
        for (int s = 0; s < KD; ++s) settle(kf[s]);
    }

    // DMA descriptors (tile-invariant parts)
    const int q_row = wave * 8 + (lane >> 3);                                   // one 1-KiB q piece per wave
    const int q_col = ((lane & 7) ^ k_swz<C::QROW>(q_row)) * 8;
    const bool q_on = q_col < p.dk;
    int g_row[C::G_DMA], g_col[C::G_DMA];
    bool g_on[C::G_DMA];
#pragma unroll
    for (int j = 0; j < C::G_DMA; ++j) {
        const int row = (wave * C::G_DMA + j) * 2 + (lane >> 5);
        const int stored = lane & 31;
        const int logical = (((stored >> 2) ^ (row & 3)) << 2) | (stored & 3);
        g_row[j] = row;
        g_col[j] = col_base + logical * 8;
        g_on[j] = g_col[j] < p.dout;
    }
    const int qt_begin = (kt * C::BKEY) / C::BQ;          // causal: only queries >= the tile's first key matter
    const int nqt = (S + C::BQ - 1) / C::BQ;
    const int nsteps = nqt - qt_begin;
    auto issue = [&](int step) {
        const uint32_t st = __builtin_amdgcn_readfirstlane(lds0 + (step % C::NSTAGE) * C::STAGE);
        const int qbase = (qt_begin + step) * C::BQ;
        if (q_on) dma16_d(qg + (int64_t)min(qbase + q_row, S - 1) * p.qk_rs + q_col, st + wave * 1024);
#pragma unroll
        for (int j = 0; j < C::G_DMA; ++j)
            if (g_on[j])
                dma16_d(gg + (int64_t)min(qbase + g_row[j], S - 1) * p.g_rs + g_col[j],
                        st + C::QTILE + (wave * C::G_DMA + j) * 1024);
        dma4(lse_g + min(qbase + lane, S - 1), st + C::QTILE + C::GTILE + wave * 256);
    };

    f32x16 acc[C::NB];
#pragma unroll
    for (int n = 0; n < C::NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    int q_read_off[KD];   // A operand of S: row l31 (+32 per sub-block), logical slot 2 s + hh
#pragma unroll
    for (int s = 0; s < KD; ++s) q_read_off[s] = l31 * C::QROW + (((2 * s + hh) ^ k_swz<C::QROW>(l31)) * 16);
    const int g_row_lane = 4 * hh + ((lane & 15) >> 2);
    const int g_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    int g_read_off[C::NB];   // dout^T fragment of column block n (see sense_mix_dma.hip, c_read_off)
#pragma unroll
    for (int n = 0; n < C::NB; ++n) g_read_off[n] = v_lds_off<C::NB>(g_row_lane, n * 4 + g_ch_lane) + (lane & 1) * 8;

    if (nsteps > 0) issue(0);
    if (nsteps > 1) issue(1);
    auto ring_step = [&](int step) {
        const int kSlot = step % C::NSTAGE;
        if (step + 1 < nsteps) wait_vmcnt<C::DMA_PER_STAGE>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (step + 2 < nsteps) issue(step + 2);
        if (!wave_has_keys) return;
        const int stage = kSlot * C::STAGE;
        const int lbase = stage + C::QTILE + C::GTILE + wave * 256;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int qbase = (qt_begin + step) * C::BQ + qb * 32;
            if (qbase >= S || qbase + 31 < key0) continue;      // past the sequence / every query before my first key
            // ---- S = Q K^T: rows = 32 queries (registers), column = my key ------------------------------
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const u32x4 a = lds_read_16B(smem, stage + q_read_off[s] + qb * 32 * C::QROW);
                st = E::mfma(a, kf[s], st);
            }
            // ---- P = exp2(S c - lse2[q]) ---------------------------------------------------------------------
            u32x4 pf[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 l4 = lds_read_16B(smem, lbase + (qb * 32 + 8 * g + 4 * hh) * 4);
                float pe[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t lw = l4[i];   // by-value copy (bp_common.h, as_f32)
                    pe[i] = fast_exp2(fmaf(st[4 * g + i], c2, -as_f32(lw) * kLog2e));
                }
                pf[g >> 1][(g & 1) * 2 + 0] = E::pack2(pe[0], pe[1]);
                pf[g >> 1][(g & 1) * 2 + 1] = E::pack2(pe[2], pe[3]);
            }
            // keys and query sub-blocks are both 32-aligned: the only partly causal sub-block is the one that starts
            // at my first key; there query `rel` is visible to key l31 iff rel >= l31.  AND on the packed words (also
            // kills NaN / inf that an uninitialised log-sum-exp past the sequence end may have produced).
            if (qbase == key0 || qbase + 32 > S) {
                const bool diag = qbase == key0;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r0 = ks * 8 + 2 * i;
                        const int rel0 = (r0 & 3) + 8 * (r0 >> 2) + 4 * hh;   // rel of r0 + 1 is rel0 + 1
                        const bool lo = (!diag || rel0 >= l31) && qbase + rel0 < S;
                        const bool hi = (!diag || rel0 + 1 >= l31) && qbase + rel0 + 1 < S;
                        pf[ks][i] &= (lo ? 0x0000ffffu : 0u) | (hi ? 0xffff0000u : 0u);
                    }
            }
            // ---- dC^T += dout^T P  (contraction over the 32 queries) --------------------------------------
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int rows = stage + C::QTILE + (qb * 32 + ks * 16) * C::GROW;
#pragma unroll
                for (int n = 0; n < C::NB; ++n) {
                    if (FULL || n < nb_live) {
                        const u32x2 lo = lds_read_tr16_8B(smem, g_read_off[n] + rows);
                        const u32x2 hi = lds_read_tr16_8B(smem, g_read_off[n] + rows + 8 * C::GROW);
                        acc[n] = E::mfma(u32x4{lo[0], lo[1], hi[0], hi[1]}, pf[ks], acc[n]);
                    }
                }
            }
        }
    };
    for (int step = 0; step < nsteps; ++step) ring_step(step);

    // ---- epilogue.  A lane holds 4-column pieces of ITS key's row, and rows of dcontent lie k*d_out*2 bytes apart:
    // storing from that layout scatters 8-byte granules over 1.6 GB (measured: 5.4 ms for the whole kernel).  So
    // each wave transposes its 32 keys x 256 columns through 16 KiB of the (now idle) ring -- 16-byte chunks XOR-
    // swizzled with the key so neither side has bank conflicts -- and writes 512-byte rows with full-line stores.
    __syncthreads();                                   // every wave is done reading the ring
    if (!wave_has_keys) return;
    float w = 1.f;
    if (p.kw != nullptr) w = p.kw[batch * p.kw_bs + (int64_t)l * p.kw_ss + min(my_key, S - 1)];
    char *scratch = smem + wave * 16384;
#pragma unroll
    for (int n = 0; n < C::NB; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chunk = (n * 4 + g) ^ l31;       // 16-byte chunk (8 columns) of my row, swizzled
            const u32x2 v = {E::pack2(acc[n][4 * g + 0] * w, acc[n][4 * g + 1] * w),
                             E::pack2(acc[n][4 * g + 2] * w, acc[n][4 * g + 3] * w)};
            *reinterpret_cast<u32x2 *>(scratch + l31 * 512 + chunk * 16 + hh * 8) = v;
        }
    // (same wave wrote and reads: no barrier, the compiler's lgkmcnt wait orders the LDS accesses)
    uint16_t *og = reinterpret_cast<uint16_t *>(p.dc) + batch * p.dc_bs + (int64_t)l * p.dc_ss;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = 2 * j + hh;                      // key row inside the wave's 32
        const int key = key0 + r;
        const int col = col_base + l31 * 8;            // this lane's 8 columns of that row
        const u32x4 v = lds_read_16B(scratch, r * 512 + ((l31 ^ r) * 16));
        if (key < S && col < p.dout) *reinterpret_cast<u32x4 *>(og + (int64_t)key * p.dc_rs + col) = v;
    }
}

template <class ET, int KD, bool FULL>
__global__ __launch_bounds__(512) void sense_mix_dc_kernel(const MixBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem[DcCfg<KD>::SMEM];
    const uint32_t lds0 = lds_base_addr(smem);
    int grp, kt;
    if (!xcd_map(blockIdx.x, p.b * p.nsenses * p.n_chunks, p.n_ktiles, grp, kt)) return;   // key tile 0 = most work first
    sense_mix_dc_tile<ET, KD, FULL>(p, smem, lds0, grp, kt);
}

template <class ET, int KD>
static hipError_t launch_dc_kd(const MixBwdParams &p, hipStream_t stream) {
    const int grid = xcd_grid(p.b * p.nsenses * p.n_chunks, p.n_ktiles);
    dim3 g(grid), t(512);
    if (p.dout % 256 == 0) hipLaunchKernelGGL((sense_mix_dc_kernel<ET, KD, true>), g, t, 0, stream, p);
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
        default: return hipErrorNotSupported;
    }
}

// d_k % 8 == 0 and <= 64, d_out % 8 == 0, 16-byte aligned bases, strides multiples of 8
hipError_t launch_sense_mix_dcontent(const MixBwdParams &p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_dc_et<BF16>(p, stream) : launch_dc_et<F16>(p, stream);
}

}  // namespace bp
