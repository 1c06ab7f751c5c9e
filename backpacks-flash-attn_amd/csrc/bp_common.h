// Shared device helpers for the gfx950 (CDNA4) Backpack kernels.
//
// Conventions used by every kernel in this directory
//   * wave = 64 lanes; `hh = lane >> 5` is the half-wave, `l31 = lane & 31`.
//   * matrix tiles use v_mfma_f32_32x32x16_{bf16,f16}:
//       A (32 x 16): lane supplies A[l31][8*hh + t], t = 0..7  (8 consecutive K elements)
//       B (16 x 32): lane supplies B[8*hh + t][l31]
//       D (32 x 32): reg r of a lane is D[(r&3) + 8*(r>>2) + 4*hh][l31]
//   * scores are computed TRANSPOSED, S^T = K Q^T, so one lane owns one query column and 16 of
//     the 32 keys of a block: row max / row sum are in-lane plus ONE exchange with lane^32.
//   * P^T feeds the second GEMM (O^T = V^T P^T) straight from those registers: regs 8*ks..8*ks+7
//     are keys {0..3, 8..11} + 4*hh + 16*ks, and the V^T operand is fetched with
//     ds_read_b64_tr_b16 in the SAME key order, so no cross-lane shuffle of P is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define BP_DEV __device__ __forceinline__

// Bit casts take their operand BY VALUE on purpose: hipcc / clang 22 (ROCm 7.2) evaluates
// `__builtin_bit_cast(float, vec[i])` -- an ext-vector ELEMENT lvalue as operand -- as a cast of
// element 0 whatever i is (seen in the IR as a splat; it silently broke a permlane exchange and a
// residual load before the parity tests caught it).  Passing through a by-value parameter is safe.
BP_DEV float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
BP_DEV uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }

struct BF16 {};
struct F16 {};

template <class ET> struct Elem;

template <> struct Elem<BF16> {
    static BP_DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static BP_DEV uint32_t pack2(float lo, float hi) {
        f32x2 x = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
    }
    static BP_DEV uint16_t from_float(float x) {
        return __builtin_bit_cast(uint16_t, (__bf16)x);
    }
    static BP_DEV float lo_f32(uint32_t w) { return as_f32(w << 16); }
    static BP_DEV float hi_f32(uint32_t w) { return as_f32(w & 0xffff0000u); }
    // c + lo + hi of a packed pair in ONE VALU instruction (v_dot2c_f32_bf16 against {1, 1})
    static BP_DEV float add_pair(uint32_t w, float c) {
        const bf16x2 one = {(__bf16)1.0f, (__bf16)1.0f};
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w), one, c, false);
    }
};

template <> struct Elem<F16> {
    static BP_DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static BP_DEV uint32_t pack2(float lo, float hi) {
        f32x2 x = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, f16x2));
    }
    static BP_DEV uint16_t from_float(float x) {
        return __builtin_bit_cast(uint16_t, (_Float16)x);
    }
    static BP_DEV float lo_f32(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
    static BP_DEV float hi_f32(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
    static BP_DEV float add_pair(uint32_t w, float c) {   // v_dot2c_f32_f16 against {1, 1}
        const f16x2 one = {(_Float16)1.0f, (_Float16)1.0f};
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, w), one, c, false);
    }
};

// 8 consecutive 16-bit elements as one 16-byte global load (pointer must be 16-B aligned).
BP_DEV u32x4 ld_global_16B(const uint16_t *p) { return *reinterpret_cast<const u32x4 *>(p); }

// Element-wise loader for rows that are not 16-byte friendly (odd head dims such as d_k = 10).
// Reads elements [col0, col0+8) of a row of `ncols` valid elements; the rest are zero.
BP_DEV u32x4 ld_global_8x2B(const uint16_t *row, int col0, int ncols) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = col0 + 2 * i;
        uint32_t lo = (c < ncols) ? row[c] : 0u;
        uint32_t hi = (c + 1 < ncols) ? row[c + 1] : 0u;
        w[i] = lo | (hi << 16);
    }
    return u32x4{w[0], w[1], w[2], w[3]};
}

// Make a value loaded with a plain global load "arrive" here: the empty asm reads and rewrites the
// register, so the compiler waits for the load at this point and treats the result as ready afterwards.
// Without it, operands loaded in a kernel's prologue and first used inside the main loop get their
// `s_waitcnt vmcnt(0)` INSIDE the loop, which also drains the LDS-DMA queue the compiler cannot see
// (the next tile, issued just before) and serialises the ring.
BP_DEV void settle(u32x4 &v) { asm volatile("" : "+v"(v)); }
BP_DEV void settle(float &v) { asm volatile("" : "+v"(v)); }

// Drain the matrix pipe behind the last MFMA of a run whose results VECTOR instructions read next.  gfx950 does not interlock
// "matrix pipe writes a VGPR -> VALU reads it": the compiler pads the gap with s_nop (11 issue slots for the 8-pass
// 32x32x16), but hipcc / clang 22 (ROCm 7.2) only does so reliably INSIDE a basic block: with a branch between the MFMA and
// its first reader -- a loop back-edge, the early exit of a run-time-bounded K loop -- the padding was found missing or short
// (round 6: `v_max_f32 v59, v3, v3` one slot behind the v_mfma writing v[0:15] in sense_lse_wide_dma_kernel<.., 10, 8, 4>;
// the stale register only moved a softmax reference maximum, i.e. the result by one ulp from launch to launch, found by a
// repeatability probe).  An empty asm pin on the accumulator does not help (the hazard recognizer ignores inline asm
// operands), so the wait is spelled out: 12 slots, in the MFMA's own block, tied to the accumulator so that neither the MFMA
// nor its readers move across it.  `pin_acc` orders further accumulators of the same run behind the drain (volatile asm
// statements keep their order).  scripts/mfma_hazard_scan.py checks every kernel of the library for early reads
// (tests/test_code_objects.py).
BP_DEV void settle_acc(f32x16 &v) { asm volatile("s_nop 7\n\ts_nop 3" : "+v"(v)); }
BP_DEV void pin_acc(f32x16 &v) { asm volatile("" : "+v"(v)); }

// Lane index recomputed on the spot (two VALU instructions) instead of being kept in a register from kernel entry: a
// value every cold address computation needs is live across all the loops, and under register pressure hipcc spills
// exactly such values -- the reload is a scratch load whose `s_waitcnt vmcnt(0)` also drains the LDS-DMA ring
// (flash_bwd_dkdv: one such reload per edge step in front of the statistics DMA, found in the round-4 listings).
// volatile: not hoisted, not merged with other calls.
BP_DEV int lane_id_now() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

// LDS accessors on a byte offset into one shared array.
BP_DEV u32x4 lds_read_16B(const char *smem, int off) {
    return *reinterpret_cast<const u32x4 *>(smem + off);
}
BP_DEV void lds_write_16B(char *smem, int off, u32x4 v) {
    *reinterpret_cast<u32x4 *>(smem + off) = v;
}
// ds_read_b64_tr_b16: within each 16-lane group the 16 x (4 x 16-bit) words are transposed, so a
// lane that points at row (j>>2), columns 4*(j&3).. of a [4][16] block (j = lane & 15) receives
// column j of that block: 4 consecutive ROWS of one column.  Address must be 8-byte aligned.
BP_DEV u32x2 lds_read_tr16_8B(const char *smem, int off) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr_t;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(smem + off));
    return __builtin_bit_cast(u32x2, v);
}

// A run of N MFMAs whose LDS operands are requested two MFMAs ahead.  hipcc's own order is "read, s_waitcnt
// lgkmcnt(0), MFMA" per MFMA -- a full LDS round trip in front of each, 75-85 clocks per MFMA where the pipe needs 32
// (sense-mix timeline, r03_aa).  LDS reads are memory operations and keep their program order relative to volatile
// asm, so the empty pins hold fetch(i + 2) in front of MFMA i; the compiler's own counted lgkmcnt waits follow.
//   fetch(i) -> u32x4 operand of MFMA i;  use(i, a): the MFMA, followed by a pin on the accumulator it wrote.
template <int N, class Fetch, class Use> BP_DEV void mfma_stream(Fetch &&fetch, Use &&use) {
    u32x4 a0 = fetch(0), a1 = fetch(N > 1 ? 1 : 0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        u32x4 a2 = a0;
        if (i + 2 < N) a2 = fetch(i + 2);
        asm volatile("" : "+v"(a0));
        use(i, a0);
        a0 = a1;
        a1 = a2;
    }
}

// Byte offset of 16-B chunk `ch` of row `row` in a V / content tile whose rows hold NV 64-byte
// chunks.  ds_read_b64_tr_b16 serves 32 lanes at once = 4 consecutive rows x 64 B; the XOR on the
// 64-B chunk index puts those 4 row segments in the 4 different quarters of the 64 banks.
//   NV=1: rows are 64 B, quarters differ by construction;  NV=3: 192-B rows rotate by 3 quarters.
template <int NV> BP_DEV int v_lds_off(int row, int ch) {
    int c64 = ch >> 2;
    if (NV == 2) c64 ^= (row >> 1) & 1;
    if (NV == 4 || NV == 8) c64 ^= row & 3;
    return row * (NV * 64) + ((c64 << 2) | (ch & 3)) * 16;
}

BP_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
BP_DEV float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// exchange with the other half-wave (lane ^ 32) through the LDS crossbar (ds_bpermute)
BP_DEV float xhalf(float x) { return __shfl_xor(x, 32); }

// max / sum of x over the two half-waves (lanes l and l^32), result in both lanes, with ONE
// v_permlane32_swap instead of an LDS round trip: the swap exchanges lanes 32..63 of its first
// operand with lanes 0..31 of its second, so {r0, r1} hold {own, other} in one order or the other.
BP_DEV void xhalf_pair(float x, float &a, float &b) {
    const uint32_t u = as_u32(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const uint32_t r0 = r[0], r1 = r[1];   // by-value copies, see as_f32
    a = as_f32(r0);
    b = as_f32(r1);
}
BP_DEV float xhalf_max(float x) {
    float a, b;
    xhalf_pair(x, a, b);
    return fmaxf(a, b);
}
BP_DEV float xhalf_sum(float x) {
    float a, b;
    xhalf_pair(x, a, b);
    return a + b;
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// XCD-aware work mapping: the dispatcher places block L on XCD L % 8; give every XCD whole `group`s of `per_group` consecutive
// work items so blocks that share K/V (or C) tiles also share an L2.  Returns false for the padding blocks.
// Which groups an XCD takes: CONSECUTIVE ones -- XCD x takes groups [x n, (x + 1) n), n = ceil(ngroups / 8), in order (round
// 6; rounds 1-5 dealt them round-robin, group g on XCD g % 8).  Groups are (sample, head) or (sample, sense) or (sample, column
// chunk) pairs, and neighbours share memory: heads whose rows are narrower than or not aligned to a 128-byte line (d_h = 80:
// 160-byte rows; the senses of the LSE pre-pass: 96- or 32-byte rows) share cache LINES, the column chunks of a sample share
// all their K rows.  On one XCD, close in time, that sharing is L2 hits; dealt round-robin every L2 fetched it again: flash
// forward at d_h = 80 10.8 -> 5.4 GB fetched per launch (-3.9 % time), LSE pre-pass at k = 64 21.8 -> 4.3 GB (-2.5 %: it is
// bound by its exponentials), Small LSE -4.3 %, d_h = 64 (128-byte rows: nothing shared) -0.5 % (profiles/r06_z_*).
// -DBP_XCD_SEQ=0 builds the round-robin deal for A/B runs.
#ifndef BP_XCD_SEQ
#define BP_XCD_SEQ 1
#endif
BP_DEV bool xcd_map(int block, int ngroups, int per_group, int &group, int &item) {
    const int xcd = block & 7;
    const int slot = block >> 3;
    item = slot % per_group;
#if BP_XCD_SEQ
    const int per_xcd = (ngroups + 7) >> 3;
    group = xcd * per_xcd + slot / per_group;
    return slot / per_group < per_xcd && group < ngroups;
#else
    group = (slot / per_group) * 8 + xcd;
    return group < ngroups;
#endif
}
// Persistent sense-mix launches (forward and dC): jobs of queue q (one queue per XCD), heaviest query tile first.  A group is a (sample, column chunk) pair; ALL chunks
// of a sample go to the queue sample mod 8, so the workgroups that stream the sample's K rows share one L2
// (-2 % at B = 64 / 128 against group mod 8, r02_u).
BP_DEV int mix_queue_groups(int b, int n_chunks, int q) { return b > q ? ((b - q + 7) / 8) * n_chunks : 0; }
BP_DEV int mix_queue_group(int n_chunks, int q, int local) {
    const int bl = local / n_chunks;
    return (bl * 8 + q) * n_chunks + (local - bl * n_chunks);
}

inline int xcd_grid(int ngroups, int per_group) { return ((ngroups + 7) / 8) * 8 * per_group; }

}  // namespace bp
