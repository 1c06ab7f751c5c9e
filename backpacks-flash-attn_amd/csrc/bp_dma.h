// LDS-DMA ring helpers shared by the fast-path kernels (flash_fwd_dma.hip, sense_mix_dma.hip).
//
// Tiles reach LDS with `global_load_lds_dwordx4` (64 lanes x 16 B per instruction, LDS destination
// = wave-uniform base + lane*16, per-lane global source), completion is tracked by hand:
//   * every wave issues the same number of DMA instructions per tile, so
//     `s_waitcnt vmcnt(that number)` means "my share of the OLDEST tile in flight has landed";
//   * one raw `s_barrier` per tile then makes everybody's share visible and, at the same time,
//     retires the ring slot that was read during the previous step (it is refilled right after).
// Because the DMA writes LDS linearly, the XOR swizzles that make the MFMA operand reads
// bank-conflict free are applied to the per-lane SOURCE address.
#pragma once
#include "bp_common.h"

namespace bp {

typedef __attribute__((address_space(3))) void lmem_v;

// One LDS-DMA instruction.  Inline asm ON PURPOSE: hipcc treats the builtin form
// (__builtin_amdgcn_global_load_lds) as a pending LDS write and drains the whole DMA queue with
// `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16, which serialises the ring.  M0 carries
// the LDS base; it is saved and restored inside the statement because the compiler owns M0.
BP_DEV void dma16(const uint16_t *g, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_addr)
        : "memory");
}

// dma16 for call sites under a per-lane condition: inside divergent control flow the compiler may keep
// the (uniform) LDS address in a VGPR, which the "s" constraint of the asm rejects.
BP_DEV void dma16_d(const uint16_t *g, uint32_t lds_addr) {
    dma16(g, __builtin_amdgcn_readfirstlane(lds_addr));
}

// 4 bytes per lane (LDS destination = base + lane*4): per-lane scalars such as a row's log-sum-exp.
BP_DEV void dma4(const void *g, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
        : "memory");
}

// Same, "saddr" form: wave-uniform 64-bit base in SGPRs + per-lane 32-bit BYTE offset.  The per-tile
// address update then happens on the scalar unit (base += tile stride) and costs no VALU.
// M0 is named as clobbered instead of being saved and restored around the instruction (two scalar moves per piece
// less).  hipcc notes that a reserved register on a clobber list "may not be preserved": that is the intent -- M0 is never
// allocated to a value, the compiler writes it immediately in front of each instruction of its own that reads it
// (none in these kernels: DS instructions do not use M0 on gfx9+), so nothing can be live in it across the statement.
BP_DEV void dma16_s(const uint16_t *uniform_base, uint32_t lane_byte_off, uint32_t lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(lane_byte_off), "s"(uniform_base), "s"(lds_addr)
        : "memory", "m0");
}

// dma16_s for data that is read once (the content stream of the sense mix): non-temporal hint.
BP_DEV void dma16_s_nt(const uint16_t *uniform_base, uint32_t lane_byte_off, uint32_t lds_addr) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1 nt"
        :
        : "v"(lane_byte_off), "s"(uniform_base), "s"(lds_addr)
        : "memory", "m0");
}

// (Rounds 3-4 had `ld_global_16B_async` here: plain global loads in inline asm whose completion the caller owned, for
// operands requested a ring step ahead.  Removed in round 5: the compiler may copy an asm OUTPUT register at any time -- a
// phi copy at a control-flow join, a live-range split -- and such a copy read the destination before the data had landed
// (sense_mix_dma.hip's next-sense operands: NaN outputs).  Operands that must travel across ring steps go through LDS
// with the DMA forms above; loads whose result is used right away are plain C++ loads.)

BP_DEV uint32_t lds_base_addr(char *smem) { return (uint32_t)(uintptr_t)(lmem_v *)smem; }

template <int N> BP_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// K rows live in LDS with a power-of-two pitch (128 or 256 B) and XOR-swizzled 16-B slots:
// ds_read_b128 by 32 lanes at 32 consecutive rows and one logical slot touches every bank once.
// Only row bits 0..3 enter, so adding 32 rows keeps a lane's swizzle.
template <int KROW> BP_DEV int k_swz(int row) { return KROW == 128 ? ((row >> 1) & 7) : (row & 15); }

}  // namespace bp
