// Fused Backpack sense combination for gfx950 -- alpha is never written to memory:
//
//     out[b,t,:] = sum_l sum_{s<=t} exp(scale * q_l[t].k_l[s] - lse[b,l,t]) * C[b,s,l,:]
//
// The reference does this with eager ops (training/src/models/backpack.py:116-122 softmax, :313
// `torch.sum(contextualization @ content, dim=1)`), materialising alpha (B,k,S,S) and a per-sense
// (B,k,S,d) temporary.  Here it is ONE contraction over the index (key s, sense l): because the
// per-(sense, query) log-sum-exp is known up front (a cheap LSE-only pass of the flash kernel over
// the k "heads" of width d_k), every probability is final when it is produced, so all k senses
// accumulate into the same O tile and nothing is ever rescaled.
//
// Schedule (one workgroup = 8 waves = 256 queries x 256 output columns of one sample):
//   for each sense l:   Q_l fragments + lse_l of my 32 queries -> registers
//     for each 32-key block up to the diagonal:
//       S^T = K_l Q_l^T        (KD MFMAs, d_k zero-padded to 16*KD)
//       P^T = exp2(S^T*c - lse*log2e), causal zeroing on the diagonal block, -> 16-bit in registers
//       O^T += C_l^T P^T       (16 MFMAs: 8 column blocks x 2 key halves), C^T via ds_read_b64_tr_b16
// K_l / C_l tiles are shared by the 8 waves through double-buffered LDS and fetched one step
// ahead into registers.  All query tiles of one (sample, column chunk) run on one XCD.
#include "bp_common.h"
#include "bp_kernels.h"

namespace bp {

template <int KD>
struct MixCfg {
    static constexpr int BM = 256;             // queries per workgroup
    static constexpr int BK = 32;              // keys per step
    static constexpr int NB = 8;               // 32-column blocks per workgroup
    static constexpr int BNC = NB * 32;        // output columns per workgroup
    static constexpr int NT = 512;
    static constexpr int KROW = KD * 32 + 16;  // bytes, padded (conflict-free ds_read_b128)
    static constexpr int CROW = NB * 64;       // bytes, XOR-swizzled 64-B chunks
    static constexpr int KTILE = BK * KROW;
    static constexpr int CTILE = BK * CROW;
    static constexpr int STAGE = KTILE + CTILE;
    static constexpr int KCH = KD * 2;
    static constexpr int CCH = NB * 4;
    static constexpr int K_ITERS = (BK * KCH + NT - 1) / NT;   // 1
    static constexpr int C_ITERS = (BK * CCH + NT - 1) / NT;   // 2
};

template <class ET, int KD, bool VEC_QK, bool VEC_C>
__global__ __launch_bounds__(512) void sense_mix_kernel(const MixParams p) {
    using C = MixCfg<KD>;
    using E = Elem<ET>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    int grp, slot;
    if (!xcd_map(blockIdx.x, p.b * p.n_chunks, p.n_qtiles, grp, slot)) return;
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

    const int q0 = qt * C::BM + wave * 32;
    const int my_q = q0 + l31;
    const bool wave_has_rows = q0 < S;
    const int my_last_kb = q0 / C::BK;   // diagonal block of this wave (q0 is a multiple of 32)
    const float c2 = p.scale_log2e;
    // column blocks that exist in this chunk (wave-uniform)
    const int nb_live = min(C::NB, (p.dout - col_base + 31) / 32);

    u32x4 kreg[C::K_ITERS];
    u32x4 creg[C::C_ITERS];
    auto fetch = [&](int step) {
        const int l = step / nkb;
        const int kb = step - l * nkb;
#pragma unroll
        for (int i = 0; i < C::K_ITERS; ++i) {
            const int c = tid + i * C::NT;
            const int row = c / C::KCH, ch = c - row * C::KCH;
            const int key = kb * C::BK + row;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (c < C::BK * C::KCH && key < S && ch * 8 < p.dk) {
                const uint16_t *r = kg + (int64_t)key * p.qk_rs + (int64_t)l * p.qk_ss;
                v = VEC_QK ? ld_global_16B(r + ch * 8) : ld_global_8x2B(r, ch * 8, p.dk);
            }
            kreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < C::C_ITERS; ++i) {
            const int c = tid + i * C::NT;
            const int row = c / C::CCH, ch = c - row * C::CCH;
            const int key = kb * C::BK + row;
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
        char *kb_ = smem + buf * C::STAGE;
        char *cb_ = kb_ + C::KTILE;
#pragma unroll
        for (int i = 0; i < C::K_ITERS; ++i) {
            const int c = tid + i * C::NT;
            const int row = c / C::KCH, ch = c - row * C::KCH;
            if (c < C::BK * C::KCH) lds_write_16B(kb_, row * C::KROW + ch * 16, kreg[i]);
        }
#pragma unroll
        for (int i = 0; i < C::C_ITERS; ++i) {
            const int c = tid + i * C::NT;
            const int row = c / C::CCH, ch = c - row * C::CCH;
            lds_write_16B(cb_, v_lds_off<C::NB>(row, ch), creg[i]);
        }
    };

    f32x16 acc[C::NB];
#pragma unroll
    for (int n = 0; n < C::NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    const int k_lane_off = l31 * C::KROW + hh * 16;
    const int c_row_lane = 4 * hh + ((lane & 15) >> 2);
    const int c_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    const int c_sub = (lane & 1) * 8;

    u32x4 qf[KD];
    float lse2 = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        const int l = step / nkb;
        const int kb = step - l * nkb;
        if (step + 1 < nsteps) fetch(step + 1);
        if (kb == 0 && wave_has_rows) {
            // new sense: my query's fragments (B operand of S^T = K Q^T) and its log-sum-exp
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const int col = 16 * s + 8 * hh;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (my_q < S && col < p.dk) {
                    const uint16_t *row = qg + (int64_t)my_q * p.qk_rs + (int64_t)l * p.qk_ss;
                    v = VEC_QK ? ld_global_16B(row + col) : ld_global_8x2B(row, col, p.dk);
                }
                qf[s] = v;
            }
            const float lse = (my_q < S) ? p.lse[((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q] : 0.f;
            lse2 = lse * kLog2e;
        }
        if (wave_has_rows && kb <= my_last_kb) {
            const char *kbuf = smem + cur * C::STAGE;
            const char *cbuf = kbuf + C::KTILE;
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const u32x4 a = lds_read_16B(kbuf, k_lane_off + s * 32);
                st = E::mfma(a, qf[s], st);
            }
            settle_acc(st);   // wait for the matrix pipe in this block (bp_common.h)
            const bool diag = (kb == my_last_kb);
            const float *kw = p.kw != nullptr ? p.kw + batch * p.kw_bs + (int64_t)l * p.kw_ss : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = fast_exp2(fmaf(st[r], c2, -lse2));
                const int key = kb * C::BK + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (diag && key > my_q) e = 0.f;
                if (kw != nullptr) e *= kw[min(key, S - 1)];   // intervention hook: alpha[:, key] scaled
                st[r] = e;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 pf;
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[i] = E::pack2(st[ks * 8 + 2 * i], st[ks * 8 + 2 * i + 1]);
                const int row0 = ks * 16 + c_row_lane;
#pragma unroll
                for (int n = 0; n < C::NB; ++n) {
                    if (n < nb_live) {
                        const int ch = n * 4 + c_ch_lane;
                        const u32x2 lo = lds_read_tr16_8B(cbuf, v_lds_off<C::NB>(row0, ch) + c_sub);
                        const u32x2 hi = lds_read_tr16_8B(cbuf, v_lds_off<C::NB>(row0 + 8, ch) + c_sub);
                        const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
                        acc[n] = E::mfma(a, pf, acc[n]);
                    }
                }
            }
        }
        if (step + 1 < nsteps) stash(cur ^ 1);
        __syncthreads();
    }

    if (!wave_has_rows || my_q >= S) return;
    uint16_t *og = reinterpret_cast<uint16_t *>(p.o) + batch * p.o_bs + (int64_t)my_q * p.o_rs;
#pragma unroll
    for (int n = 0; n < C::NB; ++n)
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

template <class ET, int KD>
static hipError_t launch_kd(const MixParams &p, bool vq, bool vc, hipStream_t stream) {
    const int grid = xcd_grid(p.b * p.n_chunks, p.n_qtiles);
    dim3 g(grid), t(512);
    if (vq && vc) hipLaunchKernelGGL((sense_mix_kernel<ET, KD, true, true>), g, t, 0, stream, p);
    else if (vq) hipLaunchKernelGGL((sense_mix_kernel<ET, KD, true, false>), g, t, 0, stream, p);
    else if (vc) hipLaunchKernelGGL((sense_mix_kernel<ET, KD, false, true>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((sense_mix_kernel<ET, KD, false, false>), g, t, 0, stream, p);
    return hipGetLastError();
}

template <class ET>
static hipError_t launch_et(const MixParams &p, bool vq, bool vc, hipStream_t stream) {
    const int kd = (p.dk + 15) / 16;
    switch (kd) {
        case 1: return launch_kd<ET, 1>(p, vq, vc, stream);
        case 2: return launch_kd<ET, 2>(p, vq, vc, stream);
        case 3: return launch_kd<ET, 3>(p, vq, vc, stream);
        case 4: return launch_kd<ET, 4>(p, vq, vc, stream);
        case 5: return launch_kd<ET, 5>(p, vq, vc, stream);
        case 6: return launch_kd<ET, 6>(p, vq, vc, stream);
        case 7: return launch_kd<ET, 7>(p, vq, vc, stream);
        default: return launch_kd<ET, 8>(p, vq, vc, stream);
    }
}

hipError_t launch_sense_mix(const MixParams &p, int dtype, bool vec_qk, bool vec_c, hipStream_t stream) {
    return dtype == 1 ? launch_et<BF16>(p, vec_qk, vec_c, stream) : launch_et<F16>(p, vec_qk, vec_c, stream);
}

}  // namespace bp
