// Fused Backpack sense combination, LDS-DMA ring version (the fast path for 16-byte-aligned shapes).
//
//     out[b,t,:] = sum_l sum_{s<=t} exp(scale * q_l[t].k_l[s] - lse[b,l,t]) * C[b,s,l,:]
//
// Same contraction and tile algebra as sense_mix.hip (see there and bp_common.h).  One workgroup = 8 waves
// = 256 queries x 256 output columns of one sample; sense outer, 64-key tile inner; a pipeline step = one
// (sense, key tile): S^T (KD MFMAs per 32 keys) -> P^T = exp2(S^T c - lse) -> O^T += C^T P^T (16 MFMAs per
// 32 keys), probabilities final on first touch thanks to the LSE pre-pass.
//   * tiles are 64 keys (40 KB: 32 KB of C + 8 KB of K) in a 3-slot LDS ring (120 KB, one workgroup per CU),
//     filled by `global_load_lds_dwordx4` (no VGPR round trip, no ds_write pass), two tiles always in flight,
//     COUNTED waits + ONE raw s_barrier per tile (bp_dma.h); the XOR swizzles that make ds_read_b128 (K) and
//     ds_read_b64_tr_b16 (C) conflict-free are applied to the per-lane SOURCE address.  Every lane of every
//     DMA piece moves VALID data (pad slots receive a duplicate of column 0): K pad columns meet zero Q columns,
//     C pad columns feed output columns that are never stored, so nothing is predicated and nothing is zeroed.
//   * the two waves of a SIMD (w and w + 4) leave a tile barrier together, so in a one-phase step both sit in their
//     softmax (VALU, matrix pipe idle) and then both in their MFMAs (VALU idle): the per-phase clock account of round 3
//     (scripts/probes/mix_timeline) found the sum of the phases' lower bounds equal to the measured step, 4085 clocks
//     against 2432 of matrix-pipe time.  A clean step is therefore two halves under two barriers -- X: S^T, softmax of
//     key half 0, 8 MFMAs with the exponentials of half 1 between them; Y: the other 24 MFMAs -- and waves 4-7 run the
//     clean loop one barrier late, so X of one wave always meets Y of its partner.  Every MFMA operand that comes from LDS
//     is requested two MFMAs ahead (mfma_stream, bp_common.h).  Steps that touch the diagonal (some waves dead or
//     masking) keep the simple per-half form.
//   * PERSISTENT launch: one workgroup per CU pulls (group, query tile) jobs from 8 per-XCD queues, heaviest
//     query tiles first, stealing from the other queues when its own is empty.  The hardware dispatcher places
//     workgroups in order and round-robin: with one workgroup per CU and jobs of 4,3,2,1 units that gave rounds of
//     4+3+2 = 9 units per CU against 7.5 ideal (dispatch_order probe); a static snake assignment lost
//     to run-time variance (r01).  A group's tiles still prefer one XCD, i.e. one L2 holds its C tiles.
//     Queue state: a 64-byte record of device memory, the caller's (`queue_ws`) or one of a small ring owned by the
//     library, zeroed by a one-wave kernel in front of every launch, see arm_mix_queues.
// Rows past the sequence are fetched from a clamped (valid) row: their probabilities are exactly 0 by the
// causal mask and 0 * finite = 0.
#include <atomic>

#include "bp_common.h"
#include "bp_dma.h"
#include "bp_kernels.h"

// the dense content stream: read once per job under the shipped ticket order -> non-temporal; the paired-ticket probe
// (BP_MIX_ORDER == 2) WANTS the group's other workgroups to hit these lines in the L2
#if defined(BP_MIX_ORDER) && BP_MIX_ORDER == 2
#define BP_MIX_CONTENT_DMA dma16_s
#else
#define BP_MIX_CONTENT_DMA dma16_s_nt
#endif

namespace bp {

// Development builds only (-DBP_MIX_PROFILE, scripts/probes/mix_timeline): every wave adds up the s_memtime ticks its
// clean steps spend in each phase; never in the shipped library.
#ifdef BP_MIX_PROFILE
__device__ unsigned long long g_mix_prof[256][8][12];   // [workgroup][wave][phase 0..5, 6 = clean steps, 7 = job ticks]
#define MIX_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#define MIX_ADD(k, expr) prof[k] += (expr)
#else
#define MIX_TICK(var) do { } while (0)
#define MIX_ADD(k, expr) do { } while (0)
#endif

#ifndef BP_MIX_X_PRIO
#define BP_MIX_X_PRIO 3
#endif

template <int KD, bool WEIGHTED = false>
struct MixDmaCfg {
    static constexpr int BM = 256, BK = 64, NB = 8, BNC = 256, NT = 512, NWAVE = 8, NSTAGE = 3;
    static constexpr int KROW = KD <= 4 ? 128 : 256;   // bytes per K row (power of two, XOR-swizzled)
    static constexpr int KSLOTS = KROW / 16;
    static constexpr int CROW = 512;
    static constexpr int KTILE = BK * KROW;
    static constexpr int CTILE = BK * CROW;
    static constexpr int WTILE = WEIGHTED ? NWAVE * 256 : 0;   // per wave: the tile's 64 key weights (fp32)
    static constexpr int STAGE = KTILE + CTILE + WTILE;
    static constexpr int K_DMA = KTILE / 1024 / NWAVE;   // DMA instructions per wave per tile (1 or 2)
    static constexpr int C_DMA = CTILE / 1024 / NWAVE;   // 4
    static constexpr int DMA_PER_STAGE = K_DMA + C_DMA + (WEIGHTED ? 1 : 0);
    static constexpr int K_ROWS_PER_DMA = 1024 / KROW;   // 8 or 4
    static constexpr int JOB_OFF = NSTAGE * STAGE;       // 16 bytes: job broadcast
    static constexpr int SMEM = JOB_OFF + 16;
    // Next sense's query operands, staged through LDS by DMA (see request_q): per wave KD fragments of 64 lanes x 16 B and
    // 64 x 4 B of log-sum-exp.  Only where the ring leaves room (128-byte K rows) and the flat two-phase loop runs.
    static constexpr bool ASYNC_Q = !WEIGHTED && KD <= 3;   // (d_k = 64 spills with the staging reads; wider ones have no LDS left)
    static constexpr int QSTAGE_OFF = SMEM;
    static constexpr int QSTAGE_WAVE = KD * 1024 + 256;
    static constexpr int QSTAGE_BYTES = ASYNC_Q ? NWAVE * QSTAGE_WAVE : 0;
    // GATHER (bp_sense_mix_gather): the content rows are rows of a TABLE (one per token id), picked by an index per key.
    // The job's row indices live behind the staging area as u16: 8 KB = 4096 keys (= kMixGatherMaxKeys, bp_kernels.h), tables
    // of up to 65 536 rows (= kMixGatherMaxRows: any GPT-2 vocabulary); the byte offset row * row bytes is formed per piece.
    // (A u16 / u32 switch per launch costs the d_k = 48 instantiation five spilled registers: larger tables are gathered by
    // the caller, bp_hip.sense_mix_gather_supported.)
    static constexpr int GATHER_OFF = SMEM + QSTAGE_BYTES;
    static constexpr int GATHER_BYTES = 8192;
    static constexpr int LDS_BYTES = SMEM + QSTAGE_BYTES;
    static_assert(LDS_BYTES + GATHER_BYTES <= 160 * 1024, "LDS budget");
};

template <class ET, int KD, bool FULL, bool WEIGHTED, bool GATHER = false>
__global__ __launch_bounds__(512) void sense_mix_dma_kernel(const MixParams p) {
    using C = MixDmaCfg<KD, WEIGHTED>;
    using E = Elem<ET>;
    __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES + (GATHER ? C::GATHER_BYTES : 0)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hh = lane >> 5;
    const int S = p.s;
    const float c2 = p.scale_log2e;
    const uint32_t lds0 = lds_base_addr(smem);

    // lane-constant LDS read offsets
    int k_read_off[KD];   // K fragment (A operand of S^T): row l31 (+32*kk), logical slot 2*s + hh
#pragma unroll
    for (int s = 0; s < KD; ++s)
        k_read_off[s] = l31 * C::KROW + (((2 * s + hh) ^ k_swz<C::KROW>(l31)) * 16);
    // (k_swz only looks at row bits 0..3, so +32 rows keeps the same swizzle)
    const int c_row_lane = 4 * hh + ((lane & 15) >> 2);
    const int c_ch_lane = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    // C^T fragment: (row c_row_lane, 16-col group of block n), +8 rows keeps swizzle.  The swizzle XORs the 64-B chunk
    // index n with row & 3, i.e. only its low two bits: block n + 4 sits exactly 256 bytes after block n, so four
    // lane offsets + an immediate serve the eight blocks.
    static_assert(C::NB == 8, "c_read_off assumes 8 column blocks");
    int c_read_off[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) c_read_off[n] = v_lds_off<C::NB>(c_row_lane, n * 4 + c_ch_lane) + (lane & 1) * 8;

    // ---- job queues -------------------------------------------------------------------------------------
    MixQueues *queues = p.queues;
    uint32_t exhausted = 0;   // bit q: queue q has no jobs left (wave-uniform, only thread 0 uses it)
    const int my_xcd = blockIdx.x & 7;
#if defined(BP_MIX_ORDER) && BP_MIX_ORDER == 2
    // Paired tickets (probe): a ticket is the PAIR of query tiles (n-1-t, t) of one (sample, column chunk) group -- the long one
    // first, the short one right behind it on the same workgroup -- and a group's pairs are consecutive tickets.  Every
    // ticket of a group then costs the same (n + 1 tile sweeps), the group's workgroups start together at key 0 and stream
    // the same content rows in step: one fetch from HBM, the rest L2 hits, where heaviest-first re-streams every tile.
    int pending = -1;         // the short tile of my current ticket (thread 0 only)
    const int n_pairs = (p.n_qtiles + 1) / 2;
#endif
    auto next_job = [&]() -> int {   // thread 0 only; returns grp * 256 + qt, or -1
#if defined(BP_MIX_ORDER) && BP_MIX_ORDER == 2
        if (pending >= 0) { const int j = pending; pending = -1; return j; }
        for (int t = 0; t < 8; ++t) {
            const int q = (my_xcd + t) & 7;
            if (exhausted & (1u << q)) continue;
            const int groups = mix_queue_groups(p.b, p.n_chunks, q);
            const int njobs = groups * n_pairs;
            const int idx = njobs > 0 ? (int)atomicAdd(&queues->ticket[q], 1u) : njobs;
            if (idx < njobs) {
                const int gl = idx / n_pairs, pr = idx - gl * n_pairs;
                const int grp = mix_queue_group(p.n_chunks, q, gl);
                const int lng = p.n_qtiles - 1 - pr;
                if (pr != lng) pending = grp * 256 + pr;
                return grp * 256 + lng;
            }
            exhausted |= 1u << q;
        }
        return -1;
#endif
        for (int t = 0; t < 8; ++t) {
            const int q = (my_xcd + t) & 7;
            if (exhausted & (1u << q)) continue;
            const int groups = mix_queue_groups(p.b, p.n_chunks, q);
            const int njobs = groups * p.n_qtiles;
            const int idx = njobs > 0 ? (int)atomicAdd(&queues->ticket[q], 1u) : njobs;
            if (idx < njobs) {
                // all groups' heaviest tiles first (a group's tiles together, so that its C slab is re-read while it might
                // still be cached, gained nothing: 1.32 / 1.34 ms against 1.29 / 1.28 ms at B = 64, same fetch traffic, r02_e)
                // (sample-major order -- one sample's twelve jobs together -- fetches 4 % less and runs 2-3 % slower, r02_w)
                // (table form, r04_ab: walking the queue column chunk by column chunk, so that the rows in flight chip-wide are
                // one chunk's third of the table, is 1-2 % slower at B = 64 ... 2048 -- the memory-side cache does not pay it back)
                // (round 6, S = 4096, review item "lockstep": -DBP_MIX_ORDER=1 hands out a group's query tiles as CONSECUTIVE
                // tickets, heaviest first, so that the CUs of one XCD stream one (sample, column chunk) slab at a time --
                // measured in profiles/r06_c_*; a development switch, the shipped order is the one below)
#if defined(BP_MIX_ORDER) && BP_MIX_ORDER == 1
                const int gl = idx / p.n_qtiles;
                return mix_queue_group(p.n_chunks, q, gl) * 256 + (p.n_qtiles - 1 - (idx - gl * p.n_qtiles));
#endif
                const int slot = idx / groups;
                const int grp = mix_queue_group(p.n_chunks, q, idx - slot * groups);
                return grp * 256 + (p.n_qtiles - 1 - slot);
            }
            exhausted |= 1u << q;
        }
        return -1;
    };

#ifdef BP_MIX_PROFILE
    unsigned long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (;;) {
        __syncthreads();   // every wave is done with the previous job's ring (and has read its job word)
        if (tid == 0) *reinterpret_cast<int *>(smem + C::JOB_OFF) = next_job();
        __syncthreads();
        const int job = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(smem + C::JOB_OFF));
        if (job < 0) break;
        const int grp = job >> 8, qt = job & 255;
        MIX_TICK(job_t0);
        const int batch = grp / p.n_chunks;
        const int chunk = grp - batch * p.n_chunks;
        const int col_base = chunk * C::BNC;

        const uint16_t *qg = reinterpret_cast<const uint16_t *>(p.q) + batch * p.qk_bs;
        const uint16_t *kg = reinterpret_cast<const uint16_t *>(p.k) + batch * p.qk_bs;
        const uint16_t *cg = reinterpret_cast<const uint16_t *>(p.c) + (GATHER ? 0 : batch * p.c_bs);

        const int k_end = min(S, qt * C::BM + C::BM);
        const int nkb = (k_end + C::BK - 1) / C::BK;
        const int nsteps = p.nsenses * nkb;
        // key tiles below this index are full and entirely visible to every row of the workgroup
        const int nkb_clean = WEIGHTED ? 0 : min((qt * C::BM) / C::BK, S / C::BK);

        const int q0 = qt * C::BM + wave * 32;
        const int my_q = q0 + l31;
        const int my_q_clamped = min(my_q, S - 1);
        const bool wave_has_rows = q0 < S;
        const int my_diag_sub = q0 >> 5;   // index of the 32-key sub-block that holds my diagonal
        const int nb_live = FULL ? C::NB : min(C::NB, (p.dout - col_base + 31) / 32);

        // Per-lane byte offsets of my DMA pieces inside a tile (the scalar tile base is added by the DMA instruction).
        // K piece j of this wave: rows (wave*K_DMA + j)*RPD + lane/KSLOTS, stored slot lane%KSLOTS; C piece j: rows
        // (wave*C_DMA + j)*2 + lane/32, stored chunk lane%32.  They are rebuilt PER JOB from an opaque copy of the lane
        // index: as loop invariants hipcc hoists the row / column tables to kernel entry and keeps them alive across
        // the whole job loop -- ten registers that the d_k = 48 instantiation then spilled to scratch (round-3 review).
        // The job's only possible partial tile (the last one, when the sequence ends inside it) clamps its rows to the
        // final valid key inside issue(), in a cold branch, instead of carrying a second offset set.
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int kb_partial = (k_end == S && (S % C::BK) != 0) ? nkb - 1 : -1;
        const int last_row = S - 1 - (nkb - 1) * C::BK;
        auto k_piece_row = [&](int j) { return (wave * C::K_DMA + j) * C::K_ROWS_PER_DMA + lane_o / C::KSLOTS; };
        auto c_piece_row = [&](int j) { return (wave * C::C_DMA + j) * 2 + (lane_o >> 5); };
        uint32_t k_voff[C::K_DMA], c_voff[C::C_DMA];
#pragma unroll
        for (int j = 0; j < C::K_DMA; ++j) {
            const int row = k_piece_row(j);
            const int logical = (lane_o % C::KSLOTS) ^ k_swz<C::KROW>(row);
            const int col = logical * 8 < p.dk ? logical * 8 : 0;   // pad slot: a duplicate of column 0 (finite)
            k_voff[j] = (uint32_t)(row * p.qk_rs + col) * 2u;
        }
#pragma unroll
        for (int j = 0; j < C::C_DMA; ++j) {
            const int row = c_piece_row(j);
            const int stored = lane_o & 31;
            const int logical = (((stored >> 2) ^ (row & 3)) << 2) | (stored & 3);
            const int col = (FULL || col_base + logical * 8 < p.dout) ? col_base + logical * 8 : col_base;
            c_voff[j] = (uint32_t)((GATHER ? 0 : row * p.c_rs) + col) * 2u;   // (GATHER: the row part comes from the table)
        }
        if (GATHER) {
            // table row of every key of the job, clamped past the sequence end (those keys are masked) and to the table
            // (an index outside it reads its last row, never memory outside it); every wave is done with the previous
            // job's table (the __syncthreads in front of the job word)
            const int32_t *idx = p.row_index + (int64_t)batch * p.idx_bs;
            for (int i = tid; i < nkb * C::BK; i += C::NT) {
                const uint32_t row = min((uint32_t)idx[min(i, S - 1)], p.last_table_row);
                *reinterpret_cast<uint16_t *>(smem + C::GATHER_OFF + i * 2) = (uint16_t)row;
            }
            __syncthreads();
        }

        // DMA pieces of the tile two steps ahead, (l2, kb2), into ring slot `slot`; `pieces` selects a subset (bit j).
        // The tile's base pointers kt2 / ct2 are carried and advanced on the scalar unit once per step (advance2 below):
        // recomputing them from (l, kb) in each of a step's calls cost ~120 SALU instructions per step.
        int l2 = 0, kb2 = 0;                     // (sense, tile) of step + 2
        const uint16_t *ks2 = kg, *cs2 = cg;     // key / content base of sense l2
        const uint16_t *kt2 = kg, *ct2 = cg;     // ... of tile kb2 in it
        const int64_t k_tile_step = (int64_t)C::BK * p.qk_rs, c_tile_step = (int64_t)C::BK * p.c_rs;
        // GATHER: piece j of tile kb2 = two table rows, their indices read from the job's LDS table; the base is the
        // SENSE's (table + l2 * c_ss), and table rows are re-read by other jobs, so the loads stay cacheable
        const uint32_t gather_row_bytes = (uint32_t)p.c_rs * 2u;
        auto gather_piece = [&](int j, uint32_t lds_dst) {
            const int key = kb2 * C::BK + c_piece_row(j);
            const uint32_t row = *reinterpret_cast<const uint16_t *>(smem + C::GATHER_OFF + key * 2);
            dma16_s(cs2, row * gather_row_bytes + c_voff[j], lds_dst);
        };
        auto issue = [&](int, int, int slot, uint32_t pieces) {
            const uint32_t stage_off = lds0 + slot * C::STAGE;
            if (__builtin_expect(kb2 == kb_partial, 0)) {
#pragma unroll
                for (int j = 0; j < C::K_DMA; ++j)
                    if ((pieces >> j) & 1u) {
                        const uint32_t back = (uint32_t)(max(k_piece_row(j) - last_row, 0) * p.qk_rs) * 2u;
                        dma16_s(kt2, k_voff[j] - back, __builtin_amdgcn_readfirstlane(stage_off + (wave * C::K_DMA + j) * 1024));
                    }
#pragma unroll
                for (int j = 0; j < C::C_DMA; ++j)
                    if ((pieces >> (C::K_DMA + j)) & 1u) {
                        if (GATHER) {
                            gather_piece(j, __builtin_amdgcn_readfirstlane(stage_off + C::KTILE + (wave * C::C_DMA + j) * 1024));
                            continue;
                        }
                        const uint32_t back = (uint32_t)(max(c_piece_row(j) - last_row, 0) * p.c_rs) * 2u;
                        BP_MIX_CONTENT_DMA(ct2, c_voff[j] - back,
                                           __builtin_amdgcn_readfirstlane(stage_off + C::KTILE + (wave * C::C_DMA + j) * 1024));
                    }
            } else {
#pragma unroll
                for (int j = 0; j < C::K_DMA; ++j)
                    if ((pieces >> j) & 1u) dma16_s(kt2, k_voff[j], stage_off + (wave * C::K_DMA + j) * 1024);
#pragma unroll
                for (int j = 0; j < C::C_DMA; ++j)
                    if ((pieces >> (C::K_DMA + j)) & 1u) {
                        if (GATHER) gather_piece(j, stage_off + C::KTILE + (wave * C::C_DMA + j) * 1024);
                        // the content stream is read once per job: non-temporal (-1.6 % at B=64, r02_p)
                        else BP_MIX_CONTENT_DMA(ct2, c_voff[j], stage_off + C::KTILE + (wave * C::C_DMA + j) * 1024);
                    }
            }
            if (WEIGHTED && ((pieces >> (C::K_DMA + C::C_DMA)) & 1u)) {
                // key weights of this (sense, tile): lane i fetches w[key0 + i] into the wave's own 256-B slot
                const float *src = p.kw + batch * p.kw_bs + (int64_t)l2 * p.kw_ss + min(kb2 * C::BK + lane, S - 1);
                dma4(src, stage_off + C::KTILE + C::CTILE + wave * 256);
            }
        };
        constexpr uint32_t kAllPieces = (1u << C::DMA_PER_STAGE) - 1u;

        f32x16 acc[C::NB];
#pragma unroll
        for (int n = 0; n < C::NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

        u32x4 qf[KD];
#pragma unroll
        for (int s = 0; s < KD; ++s) qf[s] = u32x4{0u, 0u, 0u, 0u};
        float lse2 = 0.f;
        // per-sense operands of this wave: my query's fragments (B operand of S^T = K Q^T) and its LSE
        auto take_q = [&](int l) {
            const uint16_t *row = qg + (int64_t)my_q_clamped * p.qk_rs + (int64_t)l * p.qk_ss;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const int col = 16 * s + 8 * hh;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (col < p.dk) v = ld_global_16B(row + col);
                qf[s] = v;
            }
            lse2 = p.lse[((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q_clamped] * kLog2e;
        };

        // The same for the NEXT sense, requested in the middle of a sense's last ring step (between its X and Y parts) and
        // adopted behind it -- THROUGH LDS: every lane's fragments and its log-sum-exp are DMA'd into the wave's own staging
        // area and read back after the ring's counted wait.  The DMAs are older than the pieces of tile + 2, which that step
        // issues in its Y part, so `vmcnt(DMA_PER_STAGE)` covers them and no compiler-placed vmcnt(0) stalls the wave
        // once per sense (a quarter of the steps of a light query tile).
        // Rounds 3-4 used asynchronous loads into REGISTERS here.  That is only sound while the compiler never copies the
        // destination registers before the data has landed, which it is free to do: with the request on two paths (live
        // / dead wave) hipcc loaded into temporaries and placed the phi copies `v_mov home, temporary` at the join, in
        // front of the wait -- garbage query fragments, NaN outputs (round-5 listing).  LDS has no such hazard, and the
        // staged operands cost no registers while they travel.
        const int qstage = C::QSTAGE_OFF + wave * C::QSTAGE_WAVE;
        auto request_q = [&](int l) {
            const uint16_t *row = qg + (int64_t)my_q_clamped * p.qk_rs + (int64_t)l * p.qk_ss;
#pragma unroll
            for (int s = 0; s < KD; ++s)
                dma16(row + min(16 * s + 8 * hh, p.dk - 8), __builtin_amdgcn_readfirstlane(lds0 + qstage + s * 1024));
            dma4(p.lse + ((int64_t)batch * p.nsenses + l) * p.lse_stride + my_q_clamped, lds0 + qstage + KD * 1024);
        };
        auto adopt_q = [&]() {
            wait_vmcnt<C::DMA_PER_STAGE>();   // everything older than the last step's DMA pieces has landed
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                qf[s] = lds_read_16B(smem, qstage + s * 1024 + lane * 16);
                if (16 * s + 8 * hh >= p.dk) qf[s] = u32x4{0u, 0u, 0u, 0u};
            }
            lse2 = *reinterpret_cast<const float *>(smem + qstage + KD * 1024 + lane * 4) * kLog2e;
        };

        // S^T of the 32-key half kk of the tile in ring slot byte offset `stage`
        auto scores = [&](int stage, int kk) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                const u32x4 a = lds_read_16B(smem, k_read_off[s] + stage + kk * 32 * C::KROW);
                st = E::mfma(a, qf[s], st);
            }
            return st;
        };
        auto exponentiate = [&](f32x16 &st) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = fast_exp2(fmaf(st[r], c2, -lse2));
        };
        auto pack = [&](const f32x16 &st, u32x4 (&pf)[2]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[ks][i] = E::pack2(st[ks * 8 + 2 * i], st[ks * 8 + 2 * i + 1]);
        };
        // C^T operand of column block n at LDS byte offset `rows` (16 keys x 256 columns)
        auto c_operand = [&](int rows, int n) {
            const u32x2 lo = lds_read_tr16_8B(smem, c_read_off[n & 3] + (n >> 2) * 256 + rows);
            const u32x2 hi = lds_read_tr16_8B(smem, c_read_off[n & 3] + (n >> 2) * 256 + rows + 8 * C::CROW);
            return u32x4{lo[0], lo[1], hi[0], hi[1]};
        };
        // O^T += C^T P^T over N consecutive 16-key steps (pk[0..N-1]) from key row `row0` of the tile at `stage`: 8 N MFMAs
        // as ONE operand stream with the C^T operand of MFMA i + 2 requested before MFMA i issues (mfma_stream,
        // bp_common.h) -- hipcc on its own puts "2 ds_read, s_waitcnt lgkmcnt(0)" in front of every MFMA: 75-85 clocks
        // per MFMA where the pipe needs 32 (r03_aa/ab timelines).  `mid(i)` runs behind MFMA i (DMA issue points).
        auto pv_stream = [&](int stage, int row0, const auto &pk, auto &&mid) {
            constexpr int N = sizeof(pk) / sizeof(pk[0]) * C::NB;
            const int base = stage + C::KTILE + row0 * C::CROW;
            mfma_stream<N>([&](int i) { return c_operand(base + (i >> 3) * 16 * C::CROW, i & 7); },
                           [&](int i, const u32x4 &a) {
                               if (FULL || (i & 7) < nb_live) acc[i & 7] = E::mfma(a, pk[i >> 3], acc[i & 7]);
                               asm volatile("" : "+v"(acc[i & 7]));
                               mid(i);
                           });
        };

        // Causal mask of one 32-key half on its packed P^T words.  `rel` = the half's sub-block index minus the index of
        // the sub-block that holds this wave's diagonal (wave-uniform): < 0 entirely visible, == 0 the diagonal sub-block
        // (its first key is q0: clear the 16-bit entries whose key lies above my query), > 0 entirely invisible.  AND
        // masks, so an inf from an invisible, larger score dies too.
        auto mask_half = [&](u32x4 (&pf)[2], int rel) {
            if (rel > 0) {
                pf[0] = pf[1] = u32x4{0u, 0u, 0u, 0u};
            } else if (rel == 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r0 = ks * 8 + 2 * i;
                        const int rel0 = (r0 & 3) + 8 * (r0 >> 2) + 4 * hh;   // rel of r0 + 1 is rel0 + 1
                        const uint32_t keep = (rel0 <= l31 ? 0x0000ffffu : 0u) | (rel0 + 1 <= l31 ? 0xffff0000u : 0u);
                        pf[ks][i] &= keep;
                    }
            }
        };

        // The step is cut into a vector-heavy half X (S^T of both key halves, softmax of half 0, the first 8 MFMAs of
        // half 0 with the exponentials of half 1 between them, pack) and a matrix-only half Y (the other 24 MFMAs),
        // and the two waves of a SIMD (w and w + 4) run them in ANTI-PHASE: waves 4-7 enter the job's loop one barrier
        // late, so X of one wave always meets Y of the other (see the loop below).  X hands Y three packed operands.
        // Since round 5 EVERY step has this form, the ones that touch the diagonal too (`edge`: the masks above on the
        // packed words; a wave whose rows lie entirely above the tile -- `live` false -- only issues its DMA share and
        // keeps the barrier count).  Before, diagonal steps ran a one-phase per-half form in which both waves of a SIMD sat
        // in their softmax and then both in their MFMAs: 4340 clocks per step against 3260 for a clean one, with 40 % of a
        // job's steps on the diagonal at S = 1024 (profiles/r05_*), and the anti-phase had to be re-established per sense.
        u32x4 pfc[3];   // P^T of half 0 keys 16..31, of half 1 keys 0..15 and 16..31
        auto step_x = [&](int stage, int kb, int slot2, bool dma, bool live, bool edge) {
            if (!live) {
                if (dma) issue(0, 0, slot2, kAllPieces);
                return;
            }
            // X is one dependent chain (S^T -> softmax -> first MFMAs), Y a stream of independent MFMAs: without a
            // priority the older waves 0-3 win every arbitration, and X of waves 4-7 starves behind their Y (2230
            // clocks against 1360 the other way round, r03_ab timeline)
            __builtin_amdgcn_s_setprio(BP_MIX_X_PRIO);
            f32x16 st0, st1;
            {
                // S^T of both key halves as one operand stream, alternating accumulators (the per-half form waits for
                // an LDS round trip in front of each of its KD dependent MFMAs: ~600 clocks for the six of d_k = 48)
#pragma unroll
                for (int r = 0; r < 16; ++r) st0[r] = st1[r] = 0.f;
                mfma_stream<2 * KD>(
                    [&](int i) { return lds_read_16B(smem, k_read_off[i >> 1] + stage + (i & 1) * 32 * C::KROW); },
                    [&](int i, const u32x4 &a) {
                        if (i & 1) { st1 = E::mfma(a, qf[i >> 1], st1); asm volatile("" : "+v"(st1)); }
                        else { st0 = E::mfma(a, qf[i >> 1], st0); asm volatile("" : "+v"(st0)); }
                    });
            }
            u32x4 pf0[2];
            exponentiate(st0);
            pack(st0, pf0);
            if (edge) mask_half(pf0, 2 * kb - my_diag_sub);
            if (dma) issue(0, 0, slot2, 0x03u);
            // 8 MFMAs of half 0, keys 0..15, each followed by 2 fma + 2 exp of half 1; the C operand of MFMA n+1 is
            // requested before MFMA n issues, so the LDS latency hides behind a full MFMA
            {
                const int rows = stage + C::KTILE;
                u32x4 a = c_operand(rows, 0);
#pragma unroll
                for (int n = 0; n < C::NB; ++n) {
                    u32x4 a_next = a;
                    if (n + 1 < C::NB) a_next = c_operand(rows, n + 1);
                    asm volatile("" : "+v"(a));
                    if (FULL || n < nb_live) acc[n] = E::mfma(a, pf0[0], acc[n]);   // (partial last column chunk: d = 640, 384, ...)
                    asm volatile("" : "+v"(acc[n]));
                    float x0 = st1[2 * n], x1 = st1[2 * n + 1];
                    asm volatile("" : "+v"(x0), "+v"(x1));
                    x0 = fast_exp2(fmaf(x0, c2, -lse2));
                    x1 = fast_exp2(fmaf(x1, c2, -lse2));
                    asm volatile("" : "+v"(x0), "+v"(x1));
                    st1[2 * n] = x0;
                    st1[2 * n + 1] = x1;
                    a = a_next;
                }
            }
            if (dma) issue(0, 0, slot2, kAllPieces & ~0x03u);
            pfc[0] = pf0[1];
            u32x4 pf1[2];
            pack(st1, pf1);
            if (edge) mask_half(pf1, 2 * kb + 1 - my_diag_sub);
            pfc[1] = pf1[0];
            pfc[2] = pf1[1];
            asm volatile("" : "+v"(pfc[0]), "+v"(pfc[1]), "+v"(pfc[2]));   // the packs belong to X, not behind the barrier
            __builtin_amdgcn_s_setprio(0);
        };
        auto step_y = [&](int stage, int slot2, bool dma, bool live) {
            if (!live) {
                if (dma) issue(0, 0, slot2, kAllPieces);
                return;
            }
            if (dma) issue(0, 0, slot2, 0x03u);
            pv_stream(stage, 16, pfc, [&](int i) {
                if (i == 11 && dma) issue(0, 0, slot2, kAllPieces & ~0x03u);
            });
        };

        // ---- key-weighted launches (the intervention hook): every step in the simple one-phase per-half form
        auto weighted_step = [&](int stage, int l, int kb, int slot2) {
            issue(0, 0, slot2, 0x01u);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int sub = kb * 2 + kk;
                const bool live = wave_has_rows && sub <= my_diag_sub;
                u32x4 pf[2];
                if (live) {
                    f32x16 st = scores(stage, kk);
                    exponentiate(st);
                    if (WEIGHTED) {
                        // alpha[b, l, :, key] *= w[b, l, key]  (register r holds key (r & 3) + 8 (r >> 2) + 4 hh of the
                        // sub-block: four runs of four consecutive keys)
                        const int wbase = stage + C::KTILE + C::CTILE + wave * 256 + (kk * 32 + 4 * hh) * 4;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const u32x4 w4 = lds_read_16B(smem, wbase + g * 32);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const uint32_t wi = w4[i];   // by-value copy (bp_common.h, as_f32)
                                st[4 * g + i] *= as_f32(wi);
                            }
                        }
                    }
                    pack(st, pf);
                    mask_half(pf, sub - my_diag_sub);
                    pv_stream(stage, kk * 32, pf, [](int) {});
                }
                issue(0, 0, slot2, kk == 0 ? 0x0eu : (kAllPieces & ~0x0fu));
            }
        };

        // ---- pipeline: two tiles in flight ---------------------------------------------------------------
        // step -> (sense, key tile); a step past the end re-fetches the last tile (harmless, keeps every wave's
        // DMA count per step constant so the counted wait below never changes).  The clean and the edge steps
        // of a sense run in two consecutive loops, each with ONE body: the accumulators then never meet at an
        // if/else join (hipcc answers such a join of 128 registers with copies and spills).
        auto advance2 = [&]() {
            if (kb2 + 1 < nkb) {
                ++kb2;
                kt2 += k_tile_step;
                ct2 += c_tile_step;
            } else if (l2 + 1 < p.nsenses) {
                ++l2;
                kb2 = 0;
                ks2 += p.qk_ss;
                cs2 += p.c_ss;
                kt2 = ks2;
                ct2 = cs2;
            }
        };
        issue(0, 0, 0, kAllPieces);
        advance2();
        issue(l2, kb2, 1, kAllPieces);
        advance2();

        int slot = 0;                            // ring slot of the current step
        auto step_begin = [&]() {
            wait_vmcnt<C::DMA_PER_STAGE>();   // my share of the current tile has landed (the next may be in flight)
            __builtin_amdgcn_s_barrier();     // ... and everybody's; all waves are done reading slot (slot + 2) % 3
        };
        auto step_end = [&]() {
            slot = slot == 2 ? 0 : slot + 1;
            advance2();
        };
        if (wave_has_rows) take_q(0);
        if (WEIGHTED) {
            for (int l = 0; l < p.nsenses; ++l) {
                if (l > 0 && wave_has_rows) take_q(l);
                for (int kb = 0; kb < nkb; ++kb) {
                    step_begin();
                    weighted_step(slot * C::STAGE, l, kb, slot >= 1 ? slot - 1 : 2);
                    step_end();
                }
            }
        } else {
            // Per sense: the clean tiles (entirely below the workgroup's queries) in the two-phase form, then the tiles
            // that touch the diagonal in ONE phase per step.
            //   clean: waves 0-3 run  [bar X bar Y]  per tile and one closing barrier; waves 4-7 one opening barrier and
            //   then the same [bar X bar Y]: between any two barriers one wave of a SIMD is in X (VALU + 14 MFMAs) and
            //   the other in Y (24 MFMAs) of the same or the previous tile.  Ring: tile t + 2 goes to the slot of tile
            //   t - 1, last read by Y of waves 4-7 in front of the barrier that ends X(t) of waves 0-3 -- so every wave
            //   issues it right behind that barrier (start of Y for waves 0-3, of X for waves 4-7); every wave waits for
            //   its share of the next tile before each barrier (a no-op on every other one).
            //   diagonal: X and Y of a step back to back under one barrier (`step_x` / `step_y` with the causal masks on
            //   the packed words; a wave whose rows lie above the tile only issues its DMA share).  From the third of a
            //   full query tile's four diagonal steps on every SIMD holds at most ONE live wave, and what bounds the step
            //   is that wave's own dependent chain: S^T of both halves as one stream, the first MFMAs between the second
            //   half's exponentials and one 24-MFMA stream instead of round 4's per-half form (scores, softmax, 16 MFMAs,
            //   twice: ~3190 clocks per diagonal step whoever was live, profiles/r05_c_timeline_*).  Carrying the
            //   anti-phase THROUGH the diagonal steps (two barriers there too) was built first and measured slower:
            //   a lone wave then serialises X, barrier, Y (3216 clocks per diagonal step; r05_c).
            const bool late = wave >= C::NWAVE / 2;
            for (int l = 0; l < p.nsenses; ++l) {
                const int next_sense = (C::ASYNC_Q && wave_has_rows && l + 1 < p.nsenses) ? l + 1 : -1;
                if (!C::ASYNC_Q && l > 0 && wave_has_rows) take_q(l);   // (no LDS left for the staging: plain loads)
                if (nkb_clean > 0) {
                    if (late) step_begin();
                    for (int kb = 0; kb < nkb_clean; ++kb) {
                        const int slot2 = slot >= 1 ? slot - 1 : 2;
                        MIX_TICK(t0);
                        step_begin();
                        MIX_TICK(t1);
                        step_x(slot * C::STAGE, kb, slot2, late, true, false);   // (waves without rows run on clamped operands: nothing is stored)
#ifdef BP_MIX_PROFILE
                        asm volatile("" : "+v"(acc[7]), "+v"(pfc[0]), "+v"(pfc[1]), "+v"(pfc[2]));
#endif
                        MIX_TICK(t2);
                        step_begin();
                        MIX_TICK(t3);
                        step_y(slot * C::STAGE, slot2, !late, true);
#ifdef BP_MIX_PROFILE
                        asm volatile("" : "+v"(acc[7]));
#endif
                        MIX_TICK(t4);
                        MIX_ADD(0, t1 - t0); MIX_ADD(1, t2 - t1); MIX_ADD(2, t3 - t2); MIX_ADD(3, t4 - t3); MIX_ADD(6, 1);
                        step_end();
                    }
                    if (!late) __builtin_amdgcn_s_barrier();
                }
                for (int kb = nkb_clean; kb < nkb; ++kb) {   // (never empty: the diagonal tile is one of these)
                    const bool live = wave_has_rows && 2 * kb <= my_diag_sub;
                    const int slot2 = slot >= 1 ? slot - 1 : 2;
                    MIX_TICK(e0);
                    step_begin();
                    MIX_TICK(e1);
                    step_x(slot * C::STAGE, kb, slot2, false, live, true);
                    if (kb == nkb - 1 && next_sense >= 0) request_q(next_sense);   // in front of the step's DMA pieces (in Y)
                    step_y(slot * C::STAGE, slot2, true, live);
#ifdef BP_MIX_PROFILE
                    asm volatile("" : "+v"(acc[7]));
#endif
                    MIX_TICK(e2);
                    // profile slots: clean steps 0 wait X, 1 X, 2 wait Y, 3 Y; diagonal steps 4 whole step, 5 X + Y of LIVE waves;
                    // 6 = #clean steps + (#diagonal steps << 32) + (#live diagonal steps << 48); 7 job ticks
                    MIX_ADD(4, e2 - e0);
                    MIX_ADD(5, live ? e2 - e1 : 0);
                    MIX_ADD(6, (1ull << 32) + (live ? (1ull << 48) : 0ull));
#ifdef BP_MIX_PROFILE
                    {   // 8 / 9: X + Y and count of the live steps in which the SIMD's other wave (wave ^ 4) is dead; 10 / 11: ... is live
                        const int pq0 = qt * C::BM + (wave ^ 4) * 32;
                        const bool partner_live = pq0 < S && 2 * kb <= (pq0 >> 5);
                        if (live) { prof[partner_live ? 10 : 8] += e2 - e1; prof[partner_live ? 11 : 9] += 1; }
                    }
#endif
                    step_end();
                }
                if (next_sense >= 0) adopt_q();
            }
        }
        wait_vmcnt<0>();   // the two re-fetched tiles: nothing may still be landing when the next job refills the ring

        if (wave_has_rows && my_q < S) {
            uint16_t *og = reinterpret_cast<uint16_t *>(p.o) + batch * p.o_bs + (int64_t)my_q * p.o_rs;
#pragma unroll
            for (int n = 0; n < C::NB; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = col_base + n * 32 + 8 * g + 4 * hh;
                    if (col < p.dout) {
                        u32x2 w = {E::pack2(acc[n][4 * g + 0], acc[n][4 * g + 1]),
                                   E::pack2(acc[n][4 * g + 2], acc[n][4 * g + 3])};
                        *reinterpret_cast<u32x2 *>(og + col) = w;
                    }
                }
        }
        MIX_TICK(job_t1);
        MIX_ADD(7, job_t1 - job_t0);
    }
#ifdef BP_MIX_PROFILE
    if (lane == 0 && blockIdx.x < 256)
        for (int k = 0; k < 12; ++k) g_mix_prof[blockIdx.x][wave][k] = prof[k];
#endif
}

// Queue state of a persistent launch: one 64-byte record of device memory, zeroed by a one-wave kernel enqueued right in
// front of the kernel (stream-ordered, so it is captured into a HIP graph with the launch and a launch that died
// cannot leave a stale record behind).  The record belongs to ONE launch until that launch has completed:
//   * callers that capture graphs or run launches concurrently on several streams pass their own record
//     (`queue_ws` of the C ABI; the Python binding always does -- a graph then owns the record it replays);
//   * queue_ws == NULL takes the next record of a small ring owned by the library: enough for eager launches, which
//     can only meet on one record if 64 of them are in flight at once.
constexpr int kMixQueueRing = 64;
__device__ MixQueues g_mix_queues[kMixQueueRing];

static MixQueues *next_queue_record() {
    static std::atomic<unsigned> counter{0};
    thread_local int cached_dev = -1;
    thread_local MixQueues *base = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev != cached_dev) {
        void *ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_mix_queues)) != hipSuccess) return nullptr;
        base = static_cast<MixQueues *>(ptr);
        cached_dev = dev;
    }
    return base + (counter.fetch_add(1u, std::memory_order_relaxed) % kMixQueueRing);
}

__global__ void arm_mix_queues_kernel(MixQueues *queues) {
    if (threadIdx.x < 16) reinterpret_cast<unsigned int *>(queues)[threadIdx.x] = 0u;
}

// shared with the dC launch (sense_mix_bwd.hip).  A one-wave KERNEL, not hipMemsetAsync: captured into a HIP graph the
// latter becomes a memset node, and replaying graphs of memset + persistent-kernel nodes faulted sporadically on
// ROCm 7.2 (r03_f: graph replays alone, no eager launch in flight); a kernel node in front of a kernel node does not.
hipError_t arm_mix_queues(MixQueues *&queues, hipStream_t stream) {
    if (queues == nullptr) queues = next_queue_record();
    if (queues == nullptr) return hipErrorInvalidDevice;
    hipLaunchKernelGGL(arm_mix_queues_kernel, dim3(1), dim3(64), 0, stream, queues);
    return hipGetLastError();
}

static int mix_persistent_grid() {
    thread_local int cached_dev = -1, cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached_dev = dev;
    }
    return cus;
}

template <class ET, int KD>
static hipError_t launch_kd(MixParams p, hipStream_t stream) {
    const hipError_t armed = arm_mix_queues(p.queues, stream);
    if (armed != hipSuccess) return armed;
    const int njobs = p.b * p.n_chunks * p.n_qtiles;
    const int cus = mix_persistent_grid();
    dim3 g(njobs < cus ? njobs : cus), t(512);   // 120 KB of LDS: one workgroup per CU
    if (p.row_index != nullptr) {   // (bp_api.hip: never together with key weights)
        if (p.dout % 256 == 0) hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, true, false, true>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, false, false, true>), g, t, 0, stream, p);
    } else if (p.kw != nullptr) {
        if (p.dout % 256 == 0) hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, true, true>), g, t, 0, stream, p);
        else hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, false, true>), g, t, 0, stream, p);
    } else if (p.dout % 256 == 0) hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, true, false>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((sense_mix_dma_kernel<ET, KD, false, false>), g, t, 0, stream, p);
    return hipGetLastError();
}

template <class ET>
static hipError_t launch_et(const MixParams &p, hipStream_t stream) {
    switch ((p.dk + 15) / 16) {
        case 1: return launch_kd<ET, 1>(p, stream);
        case 2: return launch_kd<ET, 2>(p, stream);
        case 3: return launch_kd<ET, 3>(p, stream);
        case 4: return launch_kd<ET, 4>(p, stream);
        case 5: return launch_kd<ET, 5>(p, stream);
        case 6: return launch_kd<ET, 6>(p, stream);
        case 7: return launch_kd<ET, 7>(p, stream);
        default: return launch_kd<ET, 8>(p, stream);
    }
}

#ifdef BP_MIX_PROFILE
extern "C" int bp_dev_mix_prof(unsigned long long *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mix_prof), sizeof(g_mix_prof)) == hipSuccess ? 0 : -1;
}
#endif

// Requires: d_k % 8 == 0, d_out % 8 == 0, all bases 16-byte aligned, all strides multiples of 8, n_qtiles <= 256.
hipError_t launch_sense_mix_dma(const MixParams &p, int dtype, hipStream_t stream) {
    return dtype == 1 ? launch_et<BF16>(p, stream) : launch_et<F16>(p, stream);
}

}  // namespace bp
