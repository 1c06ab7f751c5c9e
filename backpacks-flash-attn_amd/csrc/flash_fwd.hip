// Fused attention forward for gfx950: O = softmax(scale * Q K^T [+causal]) V, LSE.
//
// Takes the place of the reference's fmha_fwd_loop_kernel / device_1xN_loop
// (csrc/flash_attn/src/fmha_fwd_launch_template.h:41-91, src/fmha_fprop_kernel_1xN.h:199-696)
// but is a different schedule built for CDNA4:
//   * one workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows, so
//     running max / sum never leave the wave (the reference splits KEYS across warps and reduces
//     through shared memory);
//   * Q-tile outer, K/V inner with O and the softmax state in registers for the whole sweep
//     (the reference loops K-blocks outermost and round-trips O through an fp32 HBM scratch);
//   * S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16; P^T never leaves registers;
//   * K/V tiles of 64 keys are double-buffered in LDS, fetched one tile ahead into registers
//     (global loads in flight under the MFMAs), K rows padded by 16 B and V rows XOR-swizzled so
//     ds_read_b128 / ds_read_b64_tr_b16 are bank-conflict free;
//   * workgroups of one (batch, head) are placed on one XCD (shared L2 for K/V), heaviest first.
#include "bp_common.h"
#include "bp_kernels.h"

namespace bp {

template <int KD, int NV, bool HAS_V>
struct FlashCfg {
    static constexpr int BM = 128;            // queries per workgroup
    static constexpr int BN = 64;             // keys per tile
    static constexpr int NT = 256;            // threads
    static constexpr int KROW = KD * 32 + 16; // bytes per K row in LDS (16 B pad: conflict-free b128)
    static constexpr int VROW = NV * 64;      // bytes per V row in LDS (XOR-swizzled 64-B chunks)
    static constexpr int KTILE = BN * KROW;
    static constexpr int VTILE = HAS_V ? BN * VROW : 0;
    static constexpr int STAGE = KTILE + VTILE;
    static constexpr int KCH = KD * 2;        // 16-B chunks per K row
    static constexpr int VCH = NV * 4;        // 16-B chunks per V row
    static constexpr int K_ITERS = (BN * KCH + NT - 1) / NT;
    static constexpr int V_ITERS = HAS_V ? (BN * VCH + NT - 1) / NT : 0;
};

template <class ET, int KD, int NV, bool HAS_V, bool VEC>
__global__ __launch_bounds__(256) void flash_fwd_kernel(const FlashParams p) {
    using C = FlashCfg<KD, NV, HAS_V>;
    using E = Elem<ET>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;

    int bh, slot;
    if (!xcd_map(blockIdx.x, p.b * p.h, p.n_qtiles, bh, slot)) return;
    const int qt = p.n_qtiles - 1 - slot;  // heaviest (last) query tile first
    const int batch = bh / p.h;
    const int head = bh - batch * p.h;

    // sequence extents: rows cu[b]..cu[b+1] of the packed (total, h, d) arrays
    // (reference: BlockInfoPadded, csrc/flash_attn/src/fmha_kernel.h:44-75), or a fixed-length batch
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

    const int q0 = qt * C::BM + wave * 32;   // first query row of this wave
    const int my_q = q0 + l31;
    const bool wave_has_rows = q0 < seq_q;
    const float c2 = p.scale_log2e;

    // ---- Q fragments (B operand of S^T = K Q^T): query l31, d = 16*s + 8*hh .. +7 ----------------
    u32x4 qf[KD];
#pragma unroll
    for (int s = 0; s < KD; ++s) {
        const int col = 16 * s + 8 * hh;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (my_q < seq_q && col < p.d) {
            const uint16_t *row = qg + (int64_t)my_q * p.q_rs;
            v = VEC ? ld_global_16B(row + col) : ld_global_8x2B(row, col, p.d);
        }
        qf[s] = v;
    }

    // ---- tile loader: global -> registers (issued early) -> LDS (written late) -------------------
    u32x4 kreg[C::K_ITERS];
    u32x4 vreg[HAS_V ? C::V_ITERS : 1];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int i = 0; i < C::K_ITERS; ++i) {
            const int c = tid + i * C::NT;
            const int row = c / C::KCH, ch = c - row * C::KCH;
            const int key = kb * C::BN + row;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (c < C::BN * C::KCH && key < seq_k && ch * 8 < p.d) {
                const uint16_t *r = kg + (int64_t)key * p.k_rs;
                v = VEC ? ld_global_16B(r + ch * 8) : ld_global_8x2B(r, ch * 8, p.d);
            }
            kreg[i] = v;
        }
        if (HAS_V) {
#pragma unroll
            for (int i = 0; i < C::V_ITERS; ++i) {
                const int c = tid + i * C::NT;
                const int row = c / C::VCH, ch = c - row * C::VCH;
                const int key = kb * C::BN + row;
                u32x4 v = {0u, 0u, 0u, 0u};   // rows past the sequence MUST be zero: 0 * NaN = NaN in PV
                if (c < C::BN * C::VCH && key < seq_k && ch * 8 < p.d) {
                    const uint16_t *r = vg + (int64_t)key * p.v_rs;
                    v = VEC ? ld_global_16B(r + ch * 8) : ld_global_8x2B(r, ch * 8, p.d);
                }
                vreg[i] = v;
            }
        }
    };
    auto stash = [&](int buf) {
        char *kb_ = smem + buf * C::STAGE;
#pragma unroll
        for (int i = 0; i < C::K_ITERS; ++i) {
            const int c = tid + i * C::NT;
            const int row = c / C::KCH, ch = c - row * C::KCH;
            if (c < C::BN * C::KCH) lds_write_16B(kb_, row * C::KROW + ch * 16, kreg[i]);
        }
        if (HAS_V) {
            char *vb_ = kb_ + C::KTILE;
#pragma unroll
            for (int i = 0; i < C::V_ITERS; ++i) {
                const int c = tid + i * C::NT;
                const int row = c / C::VCH, ch = c - row * C::VCH;
                if (c < C::BN * C::VCH) lds_write_16B(vb_, v_lds_off<NV>(row, ch), vreg[i]);
            }
        }
    };

    f32x16 acc[HAS_V ? NV : 1];
#pragma unroll
    for (int n = 0; n < (HAS_V ? NV : 1); ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float m_run = -INFINITY;   // running max of the raw scores of my query (same in both halves)
    float l_run = 0.f;         // running sum of exp over the keys THIS lane holds (halves add up)

    // lane-constant LDS offsets
    const int k_lane_off = l31 * C::KROW + hh * 16;                   // + kk*32*KROW + s*32
    const int v_row_lane = 4 * hh + ((lane & 15) >> 2);               // + kk*32 + ks*16 (+8)
    const int v_ch_lane16 = ((lane >> 4) & 1) * 2;                    // 16-col group -> 16-B chunk pair

    auto block = [&](int kb, const char *kbuf, const char *vbuf, auto MASKED) {
        constexpr bool kMasked = decltype(MASKED)::value;
        f32x16 st[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kk][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const u32x4 a = lds_read_16B(kbuf, k_lane_off + kk * 32 * C::KROW + s * 32);
                st[kk] = E::mfma(a, qf[s], st[kk]);
            }
        }
        if (kMasked) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * C::BN + kk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const bool dead = key >= seq_k || (p.causal && key > my_q);
                    if (dead) st[kk][r] = -INFINITY;
                }
        }
        float mx = st[0][0];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kk][r]);
        mx = fmaxf(mx, xhalf(mx));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float mc = m_use * c2;
        const float alpha = fast_exp2(m_run * c2 - mc);
        m_run = m_new;
        float rs = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = fast_exp2(fmaf(st[kk][r], c2, -mc));
                st[kk][r] = e;
                rs += e;
            }
        l_run = l_run * alpha + rs;
        if (HAS_V) {
#pragma unroll
            for (int n = 0; n < NV; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] *= alpha;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 pf;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        pf[i] = E::pack2(st[kk][ks * 8 + 2 * i], st[kk][ks * 8 + 2 * i + 1]);
                    const int row0 = kk * 32 + ks * 16 + v_row_lane;
#pragma unroll
                    for (int n = 0; n < NV; ++n) {
                        const int ch = n * 4 + v_ch_lane16;   // 16-B chunk holding my 16-col group
                        // my 4 columns sit in chunk ch (cols 0-7 of the group) or ch+1 (cols 8-15)
                        const int chx = ch + ((lane & 3) >> 1);
                        const int sub = (lane & 1) * 8;
                        const u32x2 lo = lds_read_tr16_8B(vbuf, v_lds_off<NV>(row0, chx) + sub);
                        const u32x2 hi = lds_read_tr16_8B(vbuf, v_lds_off<NV>(row0 + 8, chx) + sub);
                        const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
                        acc[n] = E::mfma(a, pf, acc[n]);
                    }
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
        const bool active = wave_has_rows && !(p.causal && kb * C::BN > q0 + 31);
        if (active) {
            const char *kbuf = smem + cur * C::STAGE;
            const char *vbuf = kbuf + C::KTILE;
            const bool need_mask = (kb * C::BN + C::BN > seq_k) || (p.causal && kb * C::BN + C::BN - 1 > q0);
            if (need_mask) block(kb, kbuf, vbuf, std::true_type{});
            else block(kb, kbuf, vbuf, std::false_type{});
        }
        if (kb + 1 < nkb) stash(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    if (!wave_has_rows) return;
    const float l_tot = l_run + xhalf(l_run);
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
                    const float x0 = acc[n][4 * g + 0] * inv, x1 = acc[n][4 * g + 1] * inv;
                    const float x2 = acc[n][4 * g + 2] * inv, x3 = acc[n][4 * g + 3] * inv;
                    if (VEC) {
                        if (d0 < p.d) {
                            u32x2 w = {E::pack2(x0, x1), E::pack2(x2, x3)};
                            *reinterpret_cast<u32x2 *>(og + d0) = w;
                        }
                    } else {
                        if (d0 + 0 < p.d) og[d0 + 0] = E::from_float(x0);
                        if (d0 + 1 < p.d) og[d0 + 1] = E::from_float(x1);
                        if (d0 + 2 < p.d) og[d0 + 2] = E::from_float(x2);
                        if (d0 + 3 < p.d) og[d0 + 3] = E::from_float(x3);
                    }
                }
        }
    }
}

template <class ET, int KD, int NV, bool HAS_V>
static hipError_t launch_one(const FlashParams &p, bool vec, hipStream_t stream) {
    const int grid = xcd_grid(p.b * p.h, p.n_qtiles);
    if (vec)
        hipLaunchKernelGGL((flash_fwd_kernel<ET, KD, NV, HAS_V, true>), dim3(grid), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((flash_fwd_kernel<ET, KD, NV, HAS_V, false>), dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <class ET, bool HAS_V>
static hipError_t launch_dim(const FlashParams &p, bool vec, hipStream_t stream) {
    const int kd = (p.d + 15) / 16;
    switch (kd) {
        case 1: return launch_one<ET, 1, 1, HAS_V>(p, vec, stream);
        case 2: return launch_one<ET, 2, 1, HAS_V>(p, vec, stream);
        case 3: return launch_one<ET, 3, 2, HAS_V>(p, vec, stream);
        case 4: return launch_one<ET, 4, 2, HAS_V>(p, vec, stream);
        case 5: return launch_one<ET, 5, 3, HAS_V>(p, vec, stream);
        case 6: return launch_one<ET, 6, 3, HAS_V>(p, vec, stream);
        case 7: return launch_one<ET, 7, 4, HAS_V>(p, vec, stream);
        default: return launch_one<ET, 8, 4, HAS_V>(p, vec, stream);
    }
}

hipError_t launch_flash_fwd(const FlashParams &p, int dtype, bool vec, hipStream_t stream) {
    const bool has_v = p.v != nullptr;
    if (dtype == 1) return has_v ? launch_dim<BF16, true>(p, vec, stream) : launch_dim<BF16, false>(p, vec, stream);
    return has_v ? launch_dim<F16, true>(p, vec, stream) : launch_dim<F16, false>(p, vec, stream);
}

}  // namespace bp
