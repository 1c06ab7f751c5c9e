// Counter-based random numbers for in-kernel dropout (training path).
//
// The reference draws its dropout masks from Philox4x32-10 keyed by the torch generator's (seed, offset),
// one subsequence per 16x16 block of its sm80 tile (csrc/flash_attn/src/fmha_fprop_kernel_1xN.h:494-506,
// 673-681, src/fmha/philox.cuh; LayerNorm: csrc/layer_norm/ln_fwd_kernels.cuh:76,112-134), and compares 16-bit
// lanes of the output with a 16-bit threshold.  What callers can observe is (1) keep probability 1 - p, (2) the
// SAME mask in forward and backward from the saved generator state, (3) independence between calls / heads /
// positions.  The bit stream itself is tied to that tile shape and is not part of the contract.
//
// Here the mask is a pure function of (rng_state, batch*head, query index, key index) -- no tile shape in it --
// so the forward (lane = query), the dQ kernel (lane = query), the dK/dV kernel (lane = key) and the
// probability dump all regenerate identical bits, and a host restatement (tests/philox_ref.py) reproduces them:
//
//   Philox2x32-10 (Salmon et al., SC'11; multiplier 0xD256D193, Weyl key increment 0x9E3779B9).
//   One call yields 64 bits = four 16-bit uniforms = the run of 4 consecutive keys 4*s4 .. 4*s4+3 of one query
//   -- exactly the unit a lane of the S^T = K Q^T accumulator owns (bp_common.h).
//     stream key   : (a, b)   = philox(counter = (offset_lo, offset_hi), key = seed_lo)
//                    (kb, sb) = philox(counter = (a ^ batch_head, b),    key = seed_hi)
//     element bits : (r0, r1) = philox(counter = (query, s4 + sb),       key = kb)
//                    u16[i]   = halfword i of r0 | r1 << 32,   keep iff u16[i] < keep_threshold
//   keep_threshold = round((1 - p) * 65536); kept values are scaled by 1 / (1 - p).
#pragma once
#include "bp_common.h"

namespace bp {

constexpr uint32_t kPhiloxM = 0xD256D193u;
constexpr uint32_t kPhiloxW = 0x9E3779B9u;

BP_DEV void philox2x32(uint32_t c0, uint32_t c1, uint32_t key, uint32_t &o0, uint32_t &o1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t prod = (uint64_t)kPhiloxM * c0;
        c0 = (uint32_t)(prod >> 32) ^ key ^ c1;
        c1 = (uint32_t)prod;
        key += kPhiloxW;
    }
    o0 = c0;
    o1 = c1;
}

// What a kernel needs to know about dropout.  thr == 0 means "no dropout" (p = 0).
struct DropoutParams {
    const uint64_t *rng_state;   // device pointer: {seed, offset}
    uint32_t thr;                // keep iff u16 < thr; 0: dropout off
    float rp_keep;               // 1 / (1 - p)
};

// Per-(batch, head) stream: wave-uniform, computed once per workgroup.
struct DropoutStream {
    uint32_t key, salt;
};

BP_DEV DropoutStream dropout_stream(const uint64_t *rng_state, uint32_t batch_head) {
    const uint64_t seed = rng_state[0], offset = rng_state[1];
    uint32_t a, b, kb, sb;
    philox2x32((uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, a, b);
    philox2x32(a ^ batch_head, b, (uint32_t)(seed >> 32), kb, sb);
    DropoutStream s;
    s.key = __builtin_amdgcn_readfirstlane(kb);
    s.salt = __builtin_amdgcn_readfirstlane(sb);
    return s;
}

// The four 16-bit uniforms of (row, run of 4 consecutive columns 4*c4 .. 4*c4+3), packed as halfwords of
// (lo, hi).
BP_DEV void dropout_bits4(const DropoutStream &st, uint32_t row, uint32_t c4, uint32_t &lo, uint32_t &hi) {
    philox2x32(row, c4 + st.salt, st.key, lo, hi);
}

BP_DEV uint32_t dropout_u16(uint32_t lo, uint32_t hi, int i) {
    const uint32_t w = (i & 2) ? hi : lo;
    return (i & 1) ? (w >> 16) : (w & 0xffffu);
}

// ---- lane = row layout (forward, dQ, probability dump): a lane holds, for its row, the runs of 4 columns
//      col0 + 8*g + 4*hh + i (g = 0..3, i = 0..3) of a 32-column sub-block.  Returns a 16-bit mask: bit
//      4*g + i set = KEEP.
BP_DEV uint32_t dropout_keep_rowlane(const DropoutStream &st, uint32_t thr, uint32_t row, uint32_t col0, int hh) {
    uint32_t keep = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t lo, hi;
        dropout_bits4(st, row, (col0 >> 2) + 2 * g + hh, lo, hi);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (dropout_u16(lo, hi, i) < thr) keep |= 1u << (4 * g + i);
    }
    return keep;
}

// ---- lane = column layout (dK/dV): lane l31 holds column col = colbase + l31 and, along the registers, the
//      rows row0 + 8*g + 4*hh + i.  The four lanes of a quad own 4 consecutive columns (one run) and one
//      register group g is 4 consecutive rows: lane j of the quad evaluates the call of row i = j, the quad
//      then exchanges words with DPP broadcasts (no LDS).  `col` must be 4-aligned at lane l31 & ~3.
template <int I> BP_DEV uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, I | (I << 2) | (I << 4) | (I << 6), 0xf, 0xf, true);
}

BP_DEV uint32_t dropout_keep_collane(const DropoutStream &st, uint32_t thr, uint32_t row0, uint32_t col, int hh) {
    const int j = col & 3;
    uint32_t keep = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t lo, hi;
        dropout_bits4(st, row0 + 8 * g + 4 * hh + j, col >> 2, lo, hi);
        // row i of this group was evaluated by quad lane i; I need halfword j of it
        const uint32_t sh = (j & 1) * 16;
        const bool use_hi = (j & 2) != 0;
        // both words are fetched by every lane (DPP reads need the source lane active: no per-lane branch here)
        const uint32_t a0 = quad_bcast<0>(lo), b0 = quad_bcast<0>(hi), a1 = quad_bcast<1>(lo), b1 = quad_bcast<1>(hi);
        const uint32_t a2 = quad_bcast<2>(lo), b2 = quad_bcast<2>(hi), a3 = quad_bcast<3>(lo), b3 = quad_bcast<3>(hi);
        if ((((use_hi ? b0 : a0) >> sh) & 0xffffu) < thr) keep |= 1u << (4 * g + 0);
        if ((((use_hi ? b1 : a1) >> sh) & 0xffffu) < thr) keep |= 1u << (4 * g + 1);
        if ((((use_hi ? b2 : a2) >> sh) & 0xffffu) < thr) keep |= 1u << (4 * g + 2);
        if ((((use_hi ? b3 : a3) >> sh) & 0xffffu) < thr) keep |= 1u << (4 * g + 3);
    }
    return keep;
}

}  // namespace bp
