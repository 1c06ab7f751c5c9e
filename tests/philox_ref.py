"""Host restatement of the dropout bit stream of libbackpack_hip.so (csrc/bp_philox.h) -- test infrastructure.

Philox2x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the same family
the reference draws from through curand's Philox4x32-10, csrc/flash_attn/src/fmha/philox.cuh) in plain numpy
integer arithmetic.  The kernels' masks are pure functions of (seed, offset, stream index, row, column), so a GPU
test can demand BIT-EXACT agreement with this file instead of statistics only.
"""
import numpy as np

M = np.uint64(0xD256D193)
W = 0x9E3779B9
MASK32 = np.uint64(0xFFFFFFFF)


def philox2x32(c0, c1, key):
    """c0, c1: uint32 arrays (broadcastable), key: python int or uint32 array -> (o0, o1) uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint64) & MASK32
    c0, c1, key = np.broadcast_arrays(c0, c1, key)
    c0, c1, key = c0.copy(), c1.copy(), key.copy()
    for _ in range(10):
        prod = c0.astype(np.uint64) * M
        hi = (prod >> np.uint64(32)).astype(np.uint32)
        lo = (prod & MASK32).astype(np.uint32)
        c0, c1 = hi ^ key.astype(np.uint32) ^ c1, lo
        key = (key + np.uint64(W)) & MASK32
    return c0, c1


def stream(seed, offset, index):
    """Per-(batch, head) [or per-call, index 0] stream key and counter salt (bp_philox.h: dropout_stream)."""
    seed, offset = int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1)
    a, b = philox2x32(offset & 0xFFFFFFFF, offset >> 32, seed & 0xFFFFFFFF)
    kb, sb = philox2x32(a ^ np.asarray(index, dtype=np.uint32), b, seed >> 32)
    return kb, sb


def threshold(p):
    t = int(np.rint((np.float32(1.0) - np.float32(p)) * np.float32(65536.0)))
    return min(max(t, 1), 65535)


def uniforms16(seed, offset, index, rows, cols):
    """(rows, cols) uint16-valued uniforms of stream `index`: element (r, c) is halfword c & 3 of the call with
    counter (r, c >> 2 + salt)."""
    kb, sb = stream(seed, offset, index)
    c4 = (np.arange((cols + 3) // 4, dtype=np.uint32) + sb).astype(np.uint32)
    lo, hi = philox2x32(np.arange(rows, dtype=np.uint32)[:, None], c4[None, :], kb)
    out = np.stack([lo & np.uint32(0xFFFF), lo >> np.uint32(16), hi & np.uint32(0xFFFF), hi >> np.uint32(16)], axis=-1)
    return out.reshape(rows, -1)[:, :cols]


def attention_keep_mask(seed, offset, batch, nheads, seqlen_q, seqlen_k, p):
    """bool (batch, nheads, seqlen_q, seqlen_k): True = kept, as bp_flash_fwd_dropout draws it."""
    thr = threshold(p)
    out = np.empty((batch, nheads, seqlen_q, seqlen_k), dtype=bool)
    for b in range(batch):
        for h in range(nheads):
            out[b, h] = uniforms16(seed, offset, b * nheads + h, seqlen_q, seqlen_k) < thr
    return out


def rows_keep_mask(seed, offset, rows, cols, p):
    """bool (rows, cols): True = kept, as bp_dropout_add_layer_norm draws it (one stream, index 0; row index low 32
    bits in the counter, rows < 2^32)."""
    return uniforms16(seed, offset, 0, rows, cols) < threshold(p)
