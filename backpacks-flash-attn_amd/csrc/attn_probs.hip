// Materialise normalised attention probabilities for gfx950:
//     P[b,h,i,j] = exp(scale * q_i.k_j - lse[b,h,i])   (exactly 0 where masked)
// This is the memory-bound sweep of the path: 4*S*d bytes in, 2*H*S^2 bytes out per sample.
// Used (a) as the second pass of bp_sense_alpha -- the eager softmax of ContextSelfAttn.forward
// (training/src/models/backpack.py:116-122) whose (B,k,S,S) result callers such as
// training/src/models/intervened_models.py:78-101 edit in place -- and (b) for
// `return_attn_probs=True` of the flash interface (flash_attn/flash_attn_interface.py:242-267).
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); 64-key K tiles go through
// double-buffered LDS exactly as in flash_fwd.hip; each wave computes S^T = K Q^T for its 32 rows and
// exponentiates against the row's final LSE.  In S^T layout a lane owns 4-key slivers of ONE row, so
// storing from registers would touch 32 rows x 16 B per instruction; instead every wave transposes its
// 32 x 64 tile through a private 4 KB LDS scratch (XOR-swizzled 8-byte units) and writes whole
// 128-byte row segments: 8 rows x 128 B per store instruction.  Key blocks entirely above the
// diagonal are not computed, only zero-filled with the same full-line stores.
#include "bp_common.h"
#include "bp_kernels.h"
#include "bp_philox.h"

namespace bp {

template <int KD>
struct ProbsCfg {
    static constexpr int BM = 128, BN = 64, NT = 256;
    static constexpr int KROW = KD * 32 + 16;
    static constexpr int KTILE = BN * KROW;
    static constexpr int KCH = KD * 2;
    static constexpr int K_ITERS = (BN * KCH + NT - 1) / NT;
};

template <class ET, int KD, bool VEC>
__global__ __launch_bounds__(256) void attn_probs_kernel(const ProbsParams p) {
    using C = ProbsCfg<KD>;
    using E = Elem<ET>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::KTILE + 4 * 4096];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    const int n_qtiles = (p.sq + C::BM - 1) / C::BM;
    int bh, qt;
    if (!xcd_map(blockIdx.x, p.b * p.h, n_qtiles, bh, qt)) return;
    const int batch = bh / p.h;
    const int head = bh - batch * p.h;

    const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.q_bs + (int64_t)head * p.q_hs;
    const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.k_bs + (int64_t)head * p.k_hs;
    uint16_t *pg = reinterpret_cast<uint16_t *>(p.p) + batch * p.p_bs + (int64_t)head * p.p_hs;

    const int q0 = qt * C::BM + wave * 32;
    const int my_q = q0 + l31;
    const bool wave_has_rows = q0 < p.sq;
    const float c2 = p.scale_log2e;
    const int nkb_all = (p.sk + C::BN - 1) / C::BN;                 // blocks to WRITE
    int k_end = p.sk;
    if (p.causal) k_end = min(p.sk, qt * C::BM + C::BM);
    const int nkb = (k_end + C::BN - 1) / C::BN;                    // blocks to COMPUTE (workgroup)

    u32x4 qf[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) {
        const int col = 16 * s + 8 * hh;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (my_q < p.sq && col < p.d) {
            const uint16_t *row = qg + (int64_t)my_q * p.q_rs;
            v = VEC ? ld_global_16B(row + col) : ld_global_8x2B(row, col, p.d);
        }
        qf[s] = v;
    }
    // dropout report (bp_attn_probs_dropout): dropped entries get their sign bit set
    const bool drop = p.drop_thr != 0u;
    DropoutStream rng = {0u, 0u};
    if (drop) rng = dropout_stream(p.rng_state, (uint32_t)bh);
    auto keep_bits = [&](int kb, int kk) {
        return drop ? dropout_keep_rowlane(rng, p.drop_thr, (uint32_t)my_q, (uint32_t)(kb * C::BN + kk * 32), hh)
                    : 0xffffu;
    };
    float lse2 = 0.f;
    if (my_q < p.sq) lse2 = p.lse[((int64_t)batch * p.h + head) * p.lse_stride + my_q] * kLog2e;

    u32x4 kreg[C::K_ITERS];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int i = 0; i < C::K_ITERS; ++i) {
            const int c = tid + i * C::NT;
            const int row = c / C::KCH, ch = c - row * C::KCH;
            const int key = kb * C::BN + row;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (c < C::BN * C::KCH && key < p.sk && ch * 8 < p.d) {
                const uint16_t *r = kg + (int64_t)key * p.k_rs;
                v = VEC ? ld_global_16B(r + ch * 8) : ld_global_8x2B(r, ch * 8, p.d);
            }
            kreg[i] = v;
        }
    };
    auto stash = [&](int buf) {
        char *kb_ = smem + buf * C::KTILE;
#pragma unroll
        for (int i = 0; i < C::K_ITERS; ++i) {
            const int c = tid + i * C::NT;
            const int row = c / C::KCH, ch = c - row * C::KCH;
            if (c < C::BN * C::KCH) lds_write_16B(kb_, row * C::KROW + ch * 16, kreg[i]);
        }
    };

    const int k_lane_off = l31 * C::KROW + hh * 16;
    // a lane writes 4 consecutive keys (8 bytes) of its own query row per (kk, g)
    uint16_t *prow = pg + (int64_t)my_q * p.p_rs;
    const bool row_ok = my_q < p.sq;
    const bool vec_store = p.p_vec != 0;   // every row start 8-byte aligned (checked on the host)

    auto store4 = [&](int key0, float x0, float x1, float x2, float x3) {
        if (!row_ok) return;
        if (vec_store && key0 + 3 < p.sk) {
            u32x2 w = {E::pack2(x0, x1), E::pack2(x2, x3)};
            *reinterpret_cast<u32x2 *>(prow + key0) = w;
        } else {
            if (key0 + 0 < p.sk) prow[key0 + 0] = E::from_float(x0);
            if (key0 + 1 < p.sk) prow[key0 + 1] = E::from_float(x1);
            if (key0 + 2 < p.sk) prow[key0 + 2] = E::from_float(x2);
            if (key0 + 3 < p.sk) prow[key0 + 3] = E::from_float(x3);
        }
    };

    // ---- full-line path: tile -> wave-private LDS scratch -> 16-byte row-contiguous stores -------------
    // scratch image: [32 rows][16 units of 8 B], unit u of row r stored at u ^ (r & 15)
    char *scratch = smem + 2 * C::KTILE + wave * 4096;
    const bool fast_rows = p.p_vec16 != 0;
    const int rd_row = lane >> 3;          // + 8 * j : row this lane stores in pass j
    const int rd_chunk = lane & 7;         // 16-byte chunk of the 128-byte row segment
    auto store_tile_rows = [&](int kb, bool zeros) {
        // whole 64-key tile inside the row: rows q0 .. q0+31, byte columns kb*128 .. +127
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = rd_row + 8 * j;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (!zeros) {
                const int u0 = (2 * rd_chunk) ^ (r & 15), u1 = (2 * rd_chunk + 1) ^ (r & 15);
                const u32x2 a = *reinterpret_cast<const u32x2 *>(scratch + r * 128 + u0 * 8);
                const u32x2 b = *reinterpret_cast<const u32x2 *>(scratch + r * 128 + u1 * 8);
                v = u32x4{a[0], a[1], b[0], b[1]};
            }
            if (q0 + r < p.sq) {
                u32x4 *dst = reinterpret_cast<u32x4 *>(pg + (int64_t)(q0 + r) * p.p_rs + kb * C::BN + rd_chunk * 8);
                // write-once stream of 2*k*S^2 bytes per sample: non-temporal stores keep it from allocating in L2
                // (0.527 -> 0.440 ms at B=64 on one box, r02_p; the pure-write ceiling, torch fill, is 6.9 TB/s)
                __builtin_nontemporal_store(v, dst);
            }
        }
    };

    if (nkb > 0) {
        fetch(0);
        stash(0);
        __syncthreads();
    }
    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nkb) fetch(kb + 1);
        const bool dead_block = p.causal && kb * C::BN > q0 + 31;   // all keys above my 32 rows
        const bool full_tile = fast_rows && (kb * C::BN + C::BN <= p.sk);
        if (wave_has_rows && full_tile) {
            if (dead_block) {
                store_tile_rows(kb, true);
            } else {
                const char *kbuf = smem + cur * C::KTILE;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    f32x16 st;
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
                    for (int s = 0; s < KD; ++s) {
                        const u32x4 a = lds_read_16B(kbuf, k_lane_off + kk * 32 * C::KROW + s * 32);
                        st = E::mfma(a, qf[s], st);
                    }
                    const uint32_t keep = keep_bits(kb, kk);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float e[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int key = kb * C::BN + kk * 32 + 8 * g + 4 * hh + i;
                            e[i] = fast_exp2(fmaf(st[4 * g + i], c2, -lse2));
                            if (p.causal && key > my_q) e[i] = 0.f;
                            if (!((keep >> (4 * g + i)) & 1u)) e[i] = -e[i];
                        }
                        // my 4 keys = 8-byte unit (kk*8 + 2*g + hh) of row l31
                        const int unit = (kk * 8 + 2 * g + hh) ^ (l31 & 15);
                        u32x2 w = {E::pack2(e[0], e[1]), E::pack2(e[2], e[3])};
                        *reinterpret_cast<u32x2 *>(scratch + l31 * 128 + unit * 8) = w;
                    }
                }
                store_tile_rows(kb, false);   // same wave wrote the scratch: LDS ops complete in order
            }
        } else if (wave_has_rows) {
            if (dead_block) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int g = 0; g < 4; ++g) store4(kb * C::BN + kk * 32 + 8 * g + 4 * hh, 0.f, 0.f, 0.f, 0.f);
            } else {
                const char *kbuf = smem + cur * C::KTILE;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    f32x16 st;
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
                    for (int s = 0; s < KD; ++s) {
                        const u32x4 a = lds_read_16B(kbuf, k_lane_off + kk * 32 * C::KROW + s * 32);
                        st = E::mfma(a, qf[s], st);
                    }
                    const uint32_t keep = keep_bits(kb, kk);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb * C::BN + kk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        float e = fast_exp2(fmaf(st[r], c2, -lse2));
                        if (key >= p.sk || (p.causal && key > my_q)) e = 0.f;
                        if (!((keep >> r) & 1u)) e = -e;
                        st[r] = e;
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        store4(kb * C::BN + kk * 32 + 8 * g + 4 * hh, st[4 * g], st[4 * g + 1], st[4 * g + 2], st[4 * g + 3]);
                }
            }
        }
        if (kb + 1 < nkb) stash(cur ^ 1);
        __syncthreads();
    }
    // key blocks past the workgroup's causal range: zeros only
    if (wave_has_rows) {
        int kb = nkb;
        if (fast_rows)
            for (; kb < nkb_all && kb * C::BN + C::BN <= p.sk; ++kb) store_tile_rows(kb, true);
        for (; kb < nkb_all; ++kb)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int g = 0; g < 4; ++g) store4(kb * C::BN + kk * 32 + 8 * g + 4 * hh, 0.f, 0.f, 0.f, 0.f);
    }
}

template <class ET, int KD>
static hipError_t launch_kd(const ProbsParams &p, bool vec, hipStream_t stream) {
    const int n_qtiles = (p.sq + 127) / 128;
    dim3 g(xcd_grid(p.b * p.h, n_qtiles)), t(256);
    if (vec) hipLaunchKernelGGL((attn_probs_kernel<ET, KD, true>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((attn_probs_kernel<ET, KD, false>), g, t, 0, stream, p);
    return hipGetLastError();
}

template <class ET>
static hipError_t launch_et(const ProbsParams &p, bool vec, hipStream_t stream) {
    switch ((p.d + 15) / 16) {
        case 1: return launch_kd<ET, 1>(p, vec, stream);
        case 2: return launch_kd<ET, 2>(p, vec, stream);
        case 3: return launch_kd<ET, 3>(p, vec, stream);
        case 4: return launch_kd<ET, 4>(p, vec, stream);
        case 5: return launch_kd<ET, 5>(p, vec, stream);
        case 6: return launch_kd<ET, 6>(p, vec, stream);
        case 7: return launch_kd<ET, 7>(p, vec, stream);
        default: return launch_kd<ET, 8>(p, vec, stream);
    }
}

hipError_t launch_attn_probs(const ProbsParams &p, int dtype, bool vec, hipStream_t stream) {
    return dtype == 1 ? launch_et<BF16>(p, vec, stream) : launch_et<F16>(p, vec, stream);
}

}  // namespace bp
