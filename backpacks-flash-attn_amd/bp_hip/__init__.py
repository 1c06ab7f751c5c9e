"""ctypes binding of libbackpack_hip.so (C ABI: include/bp_hip.h).

This is the only place where Python touches the native library.  torch is used for what the
reference's pybind layer used ATen for: device memory, the current stream and output allocation
(csrc/flash_attn/fmha_api.cpp:211,267-276).  There is NO fallback: if the shared library is
missing or a call fails, a RuntimeError is raised (the reference raises ImportError /
RuntimeError in the same situations, flash_attn/flash_attn_interface.py:5).
"""
import contextlib
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('BP_HIP_LIB') or os.path.join(_HERE, 'libbackpack_hip.so')  # env: A/B builds only
ABI_VERSION = 9

_lib = None

# Backward routes that are NOT the fused HIP kernels -- autograd through an eager recomputation of the attention for head
# dims the HIP backward lacks (flash_attn_interface._FlashAttnFuncBase._bwd), the alpha-rebuilding backward of the sense
# mix for key-weighted / non-multiple-of-8 shapes (_sense_mix_backward_rebuild), the eager LayerNorm backward for rows wider
# than 2048 columns (flash_attn/ops/layer_norm.py) -- are opt-in: without
# `with bp_hip.allow_eager_fallback():` they raise instead of running silently (round-3 review: "no fallback" must mean it).
_eager_fallback = False


@contextlib.contextmanager
def allow_eager_fallback(enabled=True):
    """Within the block, backward passes whose shape the fused HIP kernels do not take may use the slower routes named
    above (they need (B, k, S, S)-sized buffers / eager attention).  Wrapping EITHER the forward or the backward() call is
    enough: the autograd Functions record the flag in their context at forward time (`fallback_recorded`) and also read it
    when backward runs.  The flag itself is process-wide on purpose: autograd runs the backward of CUDA tensors on its own
    device thread, where a thread-local set by the caller of backward() would not be seen."""
    global _eager_fallback
    previous, _eager_fallback = _eager_fallback, bool(enabled)
    try:
        yield
    finally:
        _eager_fallback = previous


def eager_fallback_allowed(ctx=None):
    """The flag now, or what `ctx` (an autograd context) recorded when its forward ran."""
    return bool(_eager_fallback or getattr(ctx, 'fallback_recorded', False))


_i32, _i64, _f32, _ptr = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/bp_hip.h declares
SIGNATURES = {
    'bp_strerror': (ctypes.c_char_p, [_i32]),
    'bp_abi_version': (_i32, []),
    'bp_build_flags': (_i32, []),
    'bp_flash_fwd': (_i32, [_ptr] * 7 + [_i32] * 5 + [_i64] * 9 + [_f32, _i32, _i32, _ptr]),
    'bp_flash_fwd_dropout': (_i32, [_ptr] * 7 + [_i32] * 5 + [_i64] * 9 + [_f32, _i32, _i32, _f32, _ptr, _ptr]),
    'bp_flash_bwd_ws_floats': (_i64, [_i32, _i32, _i64]),
    'bp_flash_bwd': (_i32, [_ptr] * 7 + [_i64] + [_ptr] * 5 + [_i32] * 5 + [_i64] * 17 + [_f32, _i32, _i32, _ptr]),
    'bp_flash_bwd_dropout': (_i32, [_ptr] * 7 + [_i64] + [_ptr] * 5 + [_i32] * 5 + [_i64] * 17
                             + [_f32, _i32, _i32, _f32, _ptr, _ptr]),
    'bp_attn_probs': (_i32, [_ptr] * 4 + [_i32] * 5 + [_i64] * 10 + [_f32, _i32, _i32, _ptr]),
    'bp_attn_probs_dropout': (_i32, [_ptr] * 4 + [_i32] * 5 + [_i64] * 10 + [_f32, _i32, _i32, _f32, _ptr, _ptr]),
    'bp_sense_lse': (_i32, [_ptr] * 2 + [_i32] * 4 + [_i64] * 4 + [_f32, _i32, _ptr]),
    'bp_sense_alpha': (_i32, [_ptr] * 3 + [_i32] * 5 + [_i64] * 4 + [_f32, _i32, _ptr]),
    'bp_sense_mix': (_i32, [_ptr] * 4 + [_i32] * 6 + [_i64] * 9 + [_f32, _i32, _ptr, _ptr]),
    'bp_sense_mix_weighted': (_i32, [_ptr] * 5 + [_i32] * 6 + [_i64] * 11 + [_f32, _i32, _ptr, _ptr]),
    'bp_sense_mix_gather': (_i32, [_ptr] * 5 + [_i32] * 6 + [_i64] * 10 + [_f32, _i32, _ptr, _ptr]),
    'bp_sense_mix_dc': (_i32, [_ptr] * 4 + [_i32] * 5 + [_i64] * 9 + [_f32, _i32, _ptr, _ptr]),
    'bp_sense_dq_dk': (_i32, [_ptr] * 6 + [_i32] * 5 + [_i64] * 11 + [_f32, _i32, _ptr]),
    'bp_add_layer_norm': (_i32, [_ptr] * 6 + [_i64, _i32, _f32] + [_i32] * 4 + [_ptr]),
    'bp_dropout_add_layer_norm': (_i32, [_ptr] * 7 + [_i64, _i32, _f32] + [_i32] * 5 + [_f32, _ptr, _ptr]),
    'bp_dropout_add_layer_norm_bwd': (_i32, [_ptr] * 9 + [_i64, _i32, _f32] + [_i32] * 4 + [_f32, _ptr, _ptr]),
    'bp_ln_bwd_ws_floats': (_i64, [_i32, _i32]),
    'bp_dropout_add_layer_norm_scaled': (_i32, [_ptr] * 9 + [_i64, _i32, _f32] + [_i32] * 5 + [_f32, _ptr, _ptr]),
    'bp_dropout_add_layer_norm_scaled_bwd': (_i32, [_ptr] * 13 + [_i64, _i64, _i32, _f32] + [_i32] * 4 + [_f32, _ptr, _ptr]),
    'bp_softmax_bwd_causal': (_i32, [_ptr, _ptr, _i64, _i32, _f32, _i32, _ptr]),
    'bp_add_layer_norm_bwd': (_i32, [_ptr] * 9 + [_i64, _i32, _f32, _i32, _i32, _i32, _ptr]),
    'bp_xentropy_fwd': (_i32, [_ptr] * 4 + [_i64, _i32, _i64, _f32, _i32, _i32, _ptr]),
    'bp_xentropy_bwd': (_i32, [_ptr] * 5 + [_i64, _i32, _i64, _i64, _f32, _i32, _i32, _ptr]),
    'bp_bias_grad_ws_floats': (_i64, [_i64, _i32]),
    'bp_bias_gelu_fwd': (_i32, [_ptr] * 4 + [_i64, _i32, _i32, _ptr]),
    'bp_bias_gelu_bwd': (_i32, [_ptr] * 5 + [_i64, _i32, _i32, _i32, _ptr]),
    'bp_column_sum': (_i32, [_ptr] * 3 + [_i64, _i32, _i32, _i32, _ptr]),
}


def lib():
    """Load (once) and return the native library; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python backpacks-flash-attn_amd/build_hip.py` '
                '(or __graft_entry__.build()).  There is no non-HIP fallback for this path.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.bp_abi_version() != ABI_VERSION:
            raise RuntimeError('libbackpack_hip.so ABI version mismatch; rebuild it')
        flags = handle.bp_build_flags()
        if flags & 3:
            # timing builds (-DBP_FWD_WHATIF / -DBP_BWD_WHATIF) delete work on purpose: never as the default library
            if not os.environ.get('BP_HIP_LIB'):
                raise RuntimeError(f'{LIB_PATH} is a what-if TIMING build (bp_build_flags() = {flags}): its results are '
                                   'garbage; rebuild without -DBP_FWD_WHATIF / -DBP_BWD_WHATIF')
            import warnings
            warnings.warn(f'bp_hip: {LIB_PATH} is a what-if timing build (flags {flags}): results are garbage on purpose')
        _lib = handle
    return _lib


def is_available():
    return os.path.exists(LIB_PATH)


def _check(code, what):
    if code != 0:
        raise RuntimeError(f'{what} failed: {lib().bp_strerror(code).decode()} (code {code})')


def _dtype_code(t):
    if t.dtype == torch.float16:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise RuntimeError(f'bp_hip: expected fp16 or bf16 tensors, got {t.dtype}')


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('bp_hip: tensors must live on the GPU (no CPU fallback exists)')


def round_up(x, m):
    return (x + m - 1) // m * m


def new_rng_state(device, generator=None):
    """Two fresh int64 words {seed, offset} ON THE DEVICE, drawn from torch's CUDA generator (so
    `torch.manual_seed` makes dropout reproducible) or from the caller's `generator` -- the role of the
    at::Generator philox state the reference hands its kernels (csrc/flash_attn/fmha_api.cpp:314-320, `gen_`
    argument).  Staying on the device keeps the call free of host synchronisation and legal inside HIP-graph
    capture; save the tensor to regenerate the mask in backward.  A CPU generator is accepted too (its two words
    are copied to the device: one small host-to-device transfer)."""
    if generator is not None and torch.device(generator.device).type != torch.device(device).type:
        return torch.randint(-2 ** 63, 2 ** 63 - 1, (2,), dtype=torch.int64, generator=generator).to(device)
    return torch.randint(-2 ** 63, 2 ** 63 - 1, (2,), dtype=torch.int64, device=device, generator=generator)


def _dropout_args(dropout_p, rng_state, device):
    dropout_p = float(dropout_p)
    if not 0.0 <= dropout_p < 1.0:
        raise RuntimeError('bp_hip: dropout_p must be in [0, 1)')
    if dropout_p == 0.0:
        return 0.0, None, None
    if rng_state is None:
        rng_state = new_rng_state(device)
    if rng_state.dtype != torch.int64 or rng_state.numel() != 2 or not rng_state.is_cuda \
            or not rng_state.is_contiguous():
        raise RuntimeError('bp_hip: rng_state must be a contiguous int64 tensor of 2 elements on the GPU')
    return dropout_p, rng_state, rng_state.data_ptr()


def flash_fwd(q, k, v, out, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale,
              causal, dropout_p=0.0, rng_state=None):
    """q (total_q,H,D), k/v (total_k,H,D), out like q (written in place); returns
    softmax_lse (B,H,roundup(max_seqlen_q,16)) fp32.  cu_seqlens_* int32 (B+1) or both None for a
    fixed-length batch whose size is total_q // max_seqlen_q.
    dropout_p > 0: attention dropout inside the kernel from `rng_state` (new_rng_state; created when None --
    pass your own to be able to hand the same state to flash_bwd / attn_probs)."""
    dropout_p, rng_state, rng_ptr = _dropout_args(dropout_p, rng_state, q.device)
    _require_cuda(q, k, v, out, cu_seqlens_q, cu_seqlens_k)
    if q.dim() != 3 or k.dim() != 3:
        raise RuntimeError('bp_hip.flash_fwd: q, k, v must be (total, nheads, headdim)')
    if not (q.dtype == k.dtype and (v is None or (v.dtype == q.dtype and out.dtype == q.dtype))):
        raise RuntimeError('bp_hip.flash_fwd: q, k, v, out must share one dtype')
    for t in (q, k, v, out):
        if t is not None and t.stride(-1) != 1:
            raise RuntimeError('bp_hip.flash_fwd: last dimension must be contiguous')
    total_q, nheads, d = q.shape
    if k.shape[1] != nheads or k.shape[2] != d or (v is not None and v.shape != k.shape):
        raise RuntimeError('bp_hip.flash_fwd: q/k/v shape mismatch')
    if v is not None and out.shape != q.shape:
        raise RuntimeError('bp_hip.flash_fwd: out must have the shape of q')
    if cu_seqlens_q is not None:
        for cu in (cu_seqlens_q, cu_seqlens_k):
            if cu.dtype != torch.int32 or not cu.is_contiguous():
                raise RuntimeError('bp_hip.flash_fwd: cu_seqlens must be contiguous int32')
        batch = cu_seqlens_q.numel() - 1
        if cu_seqlens_k.numel() - 1 != batch:
            raise RuntimeError('bp_hip.flash_fwd: cu_seqlens_q / cu_seqlens_k length mismatch')
    else:
        batch = total_q // max_seqlen_q
        if batch * max_seqlen_q != total_q or batch * max_seqlen_k != k.shape[0]:
            raise RuntimeError('bp_hip.flash_fwd: fixed-length batch does not divide total rows')
    if batch <= 0:
        raise RuntimeError('bp_hip.flash_fwd: empty batch')
    # (Callers that KNOW the batch is fixed-length pass cu_seqlens = None -- the modules of this package do; the row
    # count alone does not say so: an over-allocated (B * max_seqlen)-row buffer with shorter sequences in cu_seqlens is
    # legal, as in the reference's mha_fwd, which only reads the offsets.)
    lse_len = round_up(max_seqlen_q, 16)
    lse = torch.empty((batch, nheads, lse_len), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        code = lib().bp_flash_fwd_dropout(
            q.data_ptr(), k.data_ptr(), v.data_ptr() if v is not None else None,
            out.data_ptr() if out is not None else None, lse.data_ptr(),
            cu_seqlens_q.data_ptr() if cu_seqlens_q is not None else None,
            cu_seqlens_k.data_ptr() if cu_seqlens_k is not None else None,
            batch, nheads, d, int(max_seqlen_q), int(max_seqlen_k),
            q.stride(0), q.stride(1), k.stride(0), k.stride(1),
            v.stride(0) if v is not None else 0, v.stride(1) if v is not None else 0,
            out.stride(0) if out is not None else 0, out.stride(1) if out is not None else 0,
            lse_len, float(softmax_scale), int(bool(causal)), _dtype_code(q), dropout_p, rng_ptr, _stream())
    _check(code, 'bp_flash_fwd_dropout')
    return lse


def flash_bwd_supported(q):
    """Shapes the HIP backward covers (others recompute eagerly in the autograd Functions)."""
    return (q.is_cuda and q.dtype in (torch.float16, torch.bfloat16) and q.shape[-1] % 8 == 0
            and q.shape[-1] <= 128)


def flash_bwd(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
              max_seqlen_k, softmax_scale, causal, dropout_p=0.0, rng_state=None):
    """dq, dk, dv (written in place) of the fused attention; arguments as the reference's
    _flash_attn_backward (flash_attn/flash_attn_interface.py:31-47).  `out` and `softmax_lse` are
    the forward's results; with dropout, (dropout_p, rng_state) must be the forward's."""
    if float(dropout_p) > 0.0 and rng_state is None:
        raise RuntimeError('bp_hip.flash_bwd: dropout_p > 0 needs the rng_state the forward used')
    dropout_p, rng_state, rng_ptr = _dropout_args(dropout_p, rng_state, q.device)
    _require_cuda(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k)
    # the same checks as flash_fwd, plus the gradient buffers: a mismatched dk/dv or a wrong max_seqlen would
    # otherwise be an out-of-bounds write inside the kernel (the reference checks these in mha_bwd,
    # csrc/flash_attn/fmha_api.cpp:375-420)
    if q.dim() != 3 or k.dim() != 3:
        raise RuntimeError('bp_hip.flash_bwd: q, k, v must be (total, nheads, headdim)')
    total_q, nheads, d = q.shape
    for name, t, like in (('k', k, None), ('v', v, k), ('out', out, q), ('dout', dout, q), ('dq', dq, q),
                          ('dk', dk, k), ('dv', dv, k)):
        if t.dtype != q.dtype:
            raise RuntimeError(f'bp_hip.flash_bwd: {name} must have the dtype of q')
        if like is not None and t.shape != like.shape:
            raise RuntimeError(f'bp_hip.flash_bwd: {name} has shape {tuple(t.shape)}, expected {tuple(like.shape)}')
    if k.shape[1] != nheads or k.shape[2] != d:
        raise RuntimeError('bp_hip.flash_bwd: q/k shape mismatch')
    for name, t in (('q', q), ('k', k), ('v', v), ('dq', dq), ('dk', dk), ('dv', dv)):
        if t.stride(-1) != 1:
            raise RuntimeError(f'bp_hip.flash_bwd: last dimension of {name} must be contiguous')
    if dout.stride(-1) != 1:
        dout = dout.contiguous()
    if cu_seqlens_q is not None:
        if cu_seqlens_k is None:
            raise RuntimeError('bp_hip.flash_bwd: cu_seqlens_q and cu_seqlens_k must be given together')
        for cu in (cu_seqlens_q, cu_seqlens_k):
            if cu.dtype != torch.int32 or not cu.is_contiguous():
                raise RuntimeError('bp_hip.flash_bwd: cu_seqlens must be contiguous int32')
        batch = cu_seqlens_q.numel() - 1
        if cu_seqlens_k.numel() - 1 != batch:
            raise RuntimeError('bp_hip.flash_bwd: cu_seqlens_q / cu_seqlens_k length mismatch')
    else:
        if cu_seqlens_k is not None:
            raise RuntimeError('bp_hip.flash_bwd: cu_seqlens_q and cu_seqlens_k must be given together')
        batch = total_q // max_seqlen_q
        if batch * max_seqlen_q != total_q or batch * max_seqlen_k != k.shape[0]:
            raise RuntimeError('bp_hip.flash_bwd: fixed-length batch does not divide total rows')
    if batch <= 0:
        raise RuntimeError('bp_hip.flash_bwd: empty batch')
    if softmax_lse.dtype != torch.float32 or not softmax_lse.is_contiguous() or \
            softmax_lse.shape != (batch, nheads, round_up(max_seqlen_q, 16)):
        raise RuntimeError('bp_hip.flash_bwd: softmax_lse must be the contiguous fp32 (batch, nheads, '
                           'roundup(max_seqlen_q, 16)) tensor the forward returned')
    lse_len = softmax_lse.shape[-1]
    if out.stride(-1) != 1:
        out = out.contiguous()
    # workspace for the row statistics -D_i = -sum_d dO_i[d] * O_i[d] and -L_i / scale (filled by the dQ kernel, read
    # by the dK/dV kernel)
    dsum = torch.empty(int(lib().bp_flash_bwd_ws_floats(batch, nheads, lse_len)), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        code = lib().bp_flash_bwd_dropout(
            dout.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
            softmax_lse.data_ptr(), dsum.data_ptr(), dsum.numel(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
            cu_seqlens_q.data_ptr() if cu_seqlens_q is not None else None,
            cu_seqlens_k.data_ptr() if cu_seqlens_k is not None else None,
            batch, nheads, d, int(max_seqlen_q), int(max_seqlen_k),
            dout.stride(0), dout.stride(1), q.stride(0), q.stride(1), k.stride(0), k.stride(1),
            v.stride(0), v.stride(1), out.stride(0), out.stride(1), dq.stride(0), dq.stride(1),
            dk.stride(0), dk.stride(1), dv.stride(0), dv.stride(1), lse_len, float(softmax_scale),
            int(bool(causal)), _dtype_code(q), dropout_p, rng_ptr, _stream())
    _check(code, 'bp_flash_bwd_dropout')
    return dq, dk, dv


def attn_probs(q, k, lse, softmax_scale, causal, dropout_p=0.0, rng_state=None):
    """q (B,Sq,H,D), k (B,Sk,H,D) (any batch/row/head strides), lse (B,H,>=Sq) fp32 ->
    probabilities (B,H,Sq,Sk) in q's dtype.  With (dropout_p, rng_state) of a flash_fwd call: entries that
    call dropped carry a set SIGN BIT (decode with torch.signbit; kept / dropped magnitudes are the undropped P)."""
    _require_cuda(q, k, lse)
    if float(dropout_p) > 0.0 and rng_state is None:
        raise RuntimeError('bp_hip.attn_probs: dropout_p > 0 needs the rng_state of the forward call')
    dropout_p, rng_state, rng_ptr = _dropout_args(dropout_p, rng_state, q.device)
    b, sq, h, d = q.shape
    sk = k.shape[1]
    if q.stride(-1) != 1 or k.stride(-1) != 1 or not lse.is_contiguous():
        raise RuntimeError('bp_hip.attn_probs: bad strides')
    probs = torch.empty((b, h, sq, sk), dtype=q.dtype, device=q.device)
    with torch.cuda.device(q.device):
        code = lib().bp_attn_probs_dropout(
            q.data_ptr(), k.data_ptr(), lse.data_ptr(), probs.data_ptr(), b, h, d, sq, sk,
            q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
            lse.shape[-1], probs.stride(0), probs.stride(1), probs.stride(2),
            float(softmax_scale), int(bool(causal)), _dtype_code(q), dropout_p, rng_ptr, _stream())
    _check(code, 'bp_attn_probs_dropout')
    return probs


def _queue_ws(device):
    """The 64-byte ticket record of ONE persistent sense-mix launch (include/bp_hip.h: queue_ws).  A fresh tensor per
    call from torch's stream-ordered caching allocator: while a HIP graph is being captured it comes from the graph's
    private pool, so the graph owns the record it replays and no eager launch can ever share it.  The caller keeps the
    tensor in a local until the launch call has returned: the block then goes back to the allocator of the stream the
    launch was enqueued on (allocation and launch both use torch's CURRENT stream), whose reuse is stream-ordered."""
    return torch.empty(16, dtype=torch.int32, device=device)


def _check_qk(qk):
    _require_cuda(qk)
    if qk.dim() != 5 or qk.shape[2] != 2 or qk.stride(-1) != 1:
        raise RuntimeError('bp_hip: qk must be (B, S, 2, k, d_k) with a contiguous last dim')
    return qk.shape[0], qk.shape[1], qk.shape[3], qk.shape[4]


def _vector_friendly_qk(qk):
    """The LDS-DMA kernels move 16-byte chunks, i.e. need d_k % 8 == 0.  The Mini k=64 ablation has
    d_k = 10 (training/configs/experiment/owt/backpack-mini-flash-vecs-64.yaml): zero-pad the head
    dimension to the next multiple of 8 (zeros add nothing to q.k) instead of falling to the
    element-wise loader.  Returns (qk', true d_k) -- callers keep scaling by the TRUE d_k."""
    dk = qk.shape[-1]
    if dk % 8 == 0:
        return qk, dk
    pad = (-dk) % 8
    return torch.nn.functional.pad(qk, (0, pad)), dk


def sense_lse(qk, softmax_scale=None):
    """qk (B,S,2,k,d_k) -> lse (B,k,roundup(S,16)) fp32: log-sum-exp of every causal row."""
    b, s, k, dk = _check_qk(qk)
    scale = softmax_scale or dk ** -0.5
    qk, _ = _vector_friendly_qk(qk)
    dk = qk.shape[-1]
    lse = torch.empty((b, k, round_up(s, 16)), dtype=torch.float32, device=qk.device)
    with torch.cuda.device(qk.device):
        code = lib().bp_sense_lse(qk.data_ptr(), lse.data_ptr(), b, s, k, dk,
                                  qk.stride(0), qk.stride(1), qk.stride(2), qk.stride(3),
                                  float(scale), _dtype_code(qk), _stream())
    _check(code, 'bp_sense_lse')
    return lse


def _lse_ws(qk, lse, b, s, k):
    if lse is None:
        return torch.empty((b, k, round_up(s, 16)), dtype=torch.float32, device=qk.device), 0
    if lse.shape != (b, k, round_up(s, 16)) or lse.dtype != torch.float32 or not lse.is_contiguous():
        raise RuntimeError('bp_hip: lse must be the tensor returned by sense_lse for this qk')
    return lse, 1


def sense_alpha(qk, softmax_scale=None, lse=None):
    """qk (B,S,2,k,d_k) -> alpha (B,k,S,S), causal softmax over keys, exact zeros above the
    diagonal (replaces backpack.py:116-122).  `lse`: optional result of sense_lse(qk)."""
    b, s, k, dk = _check_qk(qk)
    scale = softmax_scale or dk ** -0.5
    qk, _ = _vector_friendly_qk(qk)
    dk = qk.shape[-1]
    alpha = torch.empty((b, k, s, s), dtype=qk.dtype, device=qk.device)
    ws, ready = _lse_ws(qk, lse, b, s, k)
    with torch.cuda.device(qk.device):
        code = lib().bp_sense_alpha(qk.data_ptr(), alpha.data_ptr(), ws.data_ptr(), ready, b, s, k, dk,
                                    qk.stride(0), qk.stride(1), qk.stride(2), qk.stride(3),
                                    float(scale), _dtype_code(qk), _stream())
    _check(code, 'bp_sense_alpha')
    return alpha


def sense_mix(qk, content, softmax_scale=None, out=None, lse=None, key_weight=None):
    """Fused sum_l softmax_causal(q_l k_l^T * scale) @ C_l without materialising alpha.

    qk (B,S,2,k,d_k); content in its storage layout (B,S,k,d_out) (the reference's
    `content` (B,k,S,d_out) is `.transpose(1,2)` of it -- pass that view transposed back, it is
    free); returns (B,S,d_out).  Replaces backpack.py:305+313.
    key_weight (B,k,S), optional: alpha[b,l,:,s] (= content row s of sense l) is scaled by
    key_weight[b,l,s] inside the kernel -- the intervention hook of intervened_models.py:97-101 and
    test_genderbias.py:71-78 (C ABI bp_sense_mix_weighted)."""
    b, s, k, dk = _check_qk(qk)
    _require_cuda(content)
    if content.dim() != 4 or content.shape[:3] != (b, s, k) or content.stride(-1) != 1:
        raise RuntimeError('bp_hip.sense_mix: content must be (B, S, k, d_out), last dim contiguous')
    if content.dtype != qk.dtype:
        raise RuntimeError('bp_hip.sense_mix: qk and content dtypes differ')
    dout = content.shape[3]
    scale = softmax_scale or dk ** -0.5
    qk, _ = _vector_friendly_qk(qk)
    dk = qk.shape[-1]
    if out is None:
        out = torch.empty((b, s, dout), dtype=qk.dtype, device=qk.device)
    ws, ready = _lse_ws(qk, lse, b, s, k)
    kw = None
    if key_weight is not None:
        _require_cuda(key_weight)
        if key_weight.shape != (b, k, s):
            raise RuntimeError('bp_hip.sense_mix: key_weight must be (B, k, S)')
        kw = key_weight.to(torch.float32)
        if kw.stride(-1) != 1:
            kw = kw.contiguous()
    queue_ws = _queue_ws(qk.device)   # alive until the launch call has returned (stream-ordered reuse afterwards)
    with torch.cuda.device(qk.device):
        code = lib().bp_sense_mix_weighted(
            qk.data_ptr(), content.data_ptr(), kw.data_ptr() if kw is not None else None, out.data_ptr(),
            ws.data_ptr(), ready, b, s, k, dk, dout,
            qk.stride(0), qk.stride(1), qk.stride(2), qk.stride(3),
            content.stride(0), content.stride(1), content.stride(2),
            kw.stride(0) if kw is not None else 0, kw.stride(1) if kw is not None else 0,
            out.stride(0), out.stride(1), float(scale), _dtype_code(qk), queue_ws.data_ptr(), _stream())
    del queue_ws
    _check(code, 'bp_sense_mix_weighted')
    return out


SENSE_MAX_DK = 640     # widest sense bp_sense_lse / _alpha / _mix take (include/bp_hip.h; > 128: csrc/sense_wide.hip)


WIDE_RING_DK = (160, 640)   # sense widths beyond 128 the LDS-DMA ring kernels take (csrc/sense_wide_dma.hip), with S % 32 == 0


def _wide_ring_takes(dk, seqlen):
    return dk in WIDE_RING_DK and seqlen % 32 == 0


def sense_mix_gather_supported(qk, table, seqlen):
    """Shapes bp_sense_mix_gather takes (include/bp_hip.h): the 16-byte vector path, seqlen <= 4096, 32-bit byte offsets into
    the table; senses up to 128 wide with at most 65 536 table rows (any GPT-2 vocabulary), or the reference's two few-sense
    widths (160 / 640, seqlen a multiple of 32: the ring kernels of csrc/sense_wide_dma.hip, any row count)."""
    dk = round_up(qk.shape[-1], 8)
    wide = dk > 128
    return ((dk <= 128 or _wide_ring_takes(dk, seqlen)) and qk.is_cuda and table.is_cuda and table.dim() == 3
            and table.stride(-1) == 1 and table.shape[2] % 8 == 0
            and table.stride(0) % 8 == 0 and table.stride(1) % 8 == 0 and table.data_ptr() % 16 == 0
            and seqlen <= 4096 and (wide or table.shape[0] <= 65536)
            and (not wide or _vector_friendly_strides(qk))
            and table.shape[0] * table.stride(0) * table.element_size() < 2 ** 32)


def _vector_friendly_strides(qk):
    return qk.data_ptr() % 16 == 0 and all(st % 8 == 0 for st in qk.stride()[:4])


def sense_mix_gather_limits(qk, table, seqlen):
    """Which limit of bp_sense_mix_gather a call exceeds, as text (callers log it when they fall back to a torch gather)."""
    why = []
    dk = round_up(qk.shape[-1], 8)
    if dk > 128 and not _wide_ring_takes(dk, seqlen):
        why.append(f'sense width {qk.shape[-1]} > 128 and not {WIDE_RING_DK} at a sequence length that is a multiple of 32 '
                   '(those senses take the dense kernel)')
    if seqlen > 4096:
        why.append(f'sequence length {seqlen} > 4096 (a job keeps its keys\' row indices in LDS)')
    if table.shape[0] > 65536 and dk <= 128:
        why.append(f'{table.shape[0]} table rows > 65536 (u16 row indices)')
    if table.shape[0] * table.stride(0) * table.element_size() >= 2 ** 32:
        why.append('table of 4 GiB or more (32-bit byte offsets)')
    if not why:
        why.append('table layout (last dim contiguous, 16-byte aligned rows, d % 8 == 0 required)')
    return '; '.join(why)


def sense_mix_gather(qk, table, row_index, softmax_scale=None, out=None, lse=None):
    """sense_mix with the content rows read from a table: content[b, s, l, :] = table[row_index[b, s], l, :].

    qk (B,S,2,k,d_k); table (rows, k, d_out), e.g. the content network's output for the distinct tokens of the batch;
    row_index (B,S) int32 (torch.unique's inverse); returns (B,S,d_out).  The (B,S,k,d_out) content tensor of
    backpack.py:276 is never materialised (C ABI bp_sense_mix_gather)."""
    b, s, k, dk = _check_qk(qk)
    _require_cuda(table, row_index)
    if table.dim() != 3 or table.shape[1] != k or table.stride(-1) != 1:
        raise RuntimeError('bp_hip.sense_mix_gather: table must be (rows, k, d_out), last dim contiguous')
    if table.dtype != qk.dtype:
        raise RuntimeError('bp_hip.sense_mix_gather: qk and table dtypes differ')
    if row_index.shape != (b, s) or row_index.dtype != torch.int32 or row_index.stride(-1) != 1:
        raise RuntimeError('bp_hip.sense_mix_gather: row_index must be (B, S) int32, unit stride along S')
    dout = table.shape[2]
    scale = softmax_scale or dk ** -0.5
    qk, _ = _vector_friendly_qk(qk)
    dk = qk.shape[-1]
    if out is None:
        out = torch.empty((b, s, dout), dtype=qk.dtype, device=qk.device)
    ws, ready = _lse_ws(qk, lse, b, s, k)
    queue_ws = _queue_ws(qk.device)   # alive until the launch call has returned
    with torch.cuda.device(qk.device):
        code = lib().bp_sense_mix_gather(
            qk.data_ptr(), table.data_ptr(), row_index.data_ptr(), out.data_ptr(), ws.data_ptr(), ready,
            b, s, k, dk, dout, table.shape[0],
            qk.stride(0), qk.stride(1), qk.stride(2), qk.stride(3),
            table.stride(0), table.stride(1), row_index.stride(0),
            out.stride(0), out.stride(1), float(scale), _dtype_code(qk), queue_ws.data_ptr(), _stream())
    del queue_ws
    _check(code, 'bp_sense_mix_gather')
    return out


def _ln_16bit_code(*tensors):
    """dtype code of the 16-bit type in play (x0 / residual / weights may each be 16-bit or fp32)."""
    for t in tensors:
        if t is not None and t.dtype in (torch.float16, torch.bfloat16):
            return _dtype_code(t), t.dtype
    return 1, torch.bfloat16


def _ln_scales(x0, weight, rowscale, colscale):
    """Checked, contiguous (rowscale, colscale) of the fused LayerNorm: rowscale one value per row in x0's dtype,
    colscale one per column in the weights' dtype (reference: csrc/layer_norm/ln_api.cpp:122-135)."""
    cols = x0.shape[-1]
    if rowscale is not None:
        _require_cuda(rowscale)
        if rowscale.numel() != x0.numel() // cols or rowscale.dtype != x0.dtype:
            raise RuntimeError('bp_hip.add_layer_norm: rowscale must hold one value per row, in x0\'s dtype')
        rowscale = rowscale.contiguous()
    if colscale is not None:
        _require_cuda(colscale)
        if colscale.shape != (cols,) or colscale.dtype != weight.dtype:
            raise RuntimeError('bp_hip.add_layer_norm: colscale (layerscale) must be (cols,) in the weights\' dtype')
        colscale = colscale.contiguous()
    return rowscale, colscale


def add_layer_norm(x0, x1, weight, bias, eps, residual_dtype=None, return_residual=True, dropout_p=0.0,
                   rng_state=None, return_dropout_mask=False, rowscale=None, colscale=None):
    """Fused z = LayerNorm(dropout(x0) / (1 - p) + x1) (fp32 math) and the updated residual stream
    x = dropout(x0) / (1 - p) + x1.

    x0 (..., cols) fp16 / bf16 / fp32 contiguous; x1 same shape (fp32 or x0's dtype) or None; weight/bias
    (cols,) fp32 or 16-bit.  Returns (z in x0's dtype, x in residual_dtype) -- or z alone -- and, when asked,
    the uint8 keep mask (all ones without dropout).  residual_dtype defaults to x1's dtype (reference rule,
    csrc/layer_norm/ln_api.cpp:99-102).  Replaces dropout_add_layer_norm (flash_attn/ops/layer_norm.py:207-217).
    dropout_p > 0 draws from `rng_state` (new_rng_state; pass your own to hand the same state to
    add_layer_norm_bwd).  rowscale (one value per row, x0's dtype) / colscale (cols, the weights' dtype): x0 is multiplied
    by rowscale[row] before and by colscale[col] after the dropout -- DropPath and LayerScale (`rowscale`, `layerscale` of
    the reference's dropout_add_layer_norm)."""
    _require_cuda(x0, x1, weight, bias)
    if x0.dtype not in (torch.float16, torch.bfloat16, torch.float32):
        raise RuntimeError(f'bp_hip.add_layer_norm: x0 must be fp16, bf16 or fp32, got {x0.dtype}')
    cols = x0.shape[-1]
    if weight.shape != (cols,) or bias.shape != (cols,) or weight.dtype != bias.dtype:
        raise RuntimeError('bp_hip.add_layer_norm: weight/bias must be (cols,) of one dtype')
    code_dt, dt16 = _ln_16bit_code(x0, x1, weight)
    x0_f32 = x0.dtype == torch.float32
    if weight.dtype not in (torch.float32, dt16):
        raise RuntimeError('bp_hip.add_layer_norm: weight dtype must be fp32 or the 16-bit dtype in use')
    x0c = x0.contiguous()
    x1c = None
    if x1 is not None:
        if x1.shape != x0.shape or x1.dtype not in (torch.float32, x0.dtype):
            raise RuntimeError('bp_hip.add_layer_norm: residual must match x0 (fp32 or input dtype)')
        x1c = x1.contiguous()
    if residual_dtype is None:
        residual_dtype = x1.dtype if x1 is not None else x0.dtype
    if residual_dtype not in (torch.float32, x0.dtype):
        raise RuntimeError('bp_hip.add_layer_norm: residual dtype must be fp32 or the input dtype')
    dropout_p, rng_state, rng_ptr = _dropout_args(dropout_p, rng_state, x0.device)
    z = torch.empty_like(x0c)
    xo = torch.empty(x0c.shape, dtype=residual_dtype, device=x0.device) if return_residual else None
    dmask = None
    if return_dropout_mask:
        dmask = (torch.empty(x0c.shape, dtype=torch.uint8, device=x0.device) if dropout_p > 0.0
                 else torch.ones(x0c.shape, dtype=torch.uint8, device=x0.device))
    rows = x0c.numel() // cols
    wc, bc = weight.contiguous(), bias.contiguous()
    rowscale, colscale = _ln_scales(x0c, weight, rowscale, colscale)
    with torch.cuda.device(x0.device):
        code = lib().bp_dropout_add_layer_norm_scaled(
            x0c.data_ptr(), x1c.data_ptr() if x1c is not None else None, wc.data_ptr(), bc.data_ptr(),
            rowscale.data_ptr() if rowscale is not None else None,
            colscale.data_ptr() if colscale is not None else None,
            z.data_ptr(), xo.data_ptr() if xo is not None else None,
            dmask.data_ptr() if (dmask is not None and dropout_p > 0.0) else None, rows, cols, float(eps), code_dt,
            int(x0_f32), int(x1c is not None and x1c.dtype == torch.float32), int(residual_dtype == torch.float32),
            int(weight.dtype == torch.float32), dropout_p, rng_ptr, _stream())
    _check(code, 'bp_dropout_add_layer_norm_scaled')
    outs = (z, xo) if return_residual else (z,)
    if return_dropout_mask:
        outs = outs + (dmask,)
    return outs if len(outs) > 1 else outs[0]


def softmax_bwd_causal_supported(alpha):
    return alpha.is_cuda and alpha.dtype in (torch.float16, torch.bfloat16) and alpha.shape[-1] % 8 == 0 \
        and alpha.shape[-1] <= 4096


def softmax_bwd_causal_(alpha, dalpha, softmax_scale):
    """In place: dalpha (..., S, S) <- scale * alpha * (dalpha - rowsum(alpha * dalpha)) below/on the diagonal,
    0 above it.  alpha as returned by sense_alpha.  Returns dalpha (now the gradient of the raw scores q.k)."""
    _require_cuda(alpha, dalpha)
    if alpha.shape != dalpha.shape or alpha.dtype != dalpha.dtype or not alpha.is_contiguous() \
            or not dalpha.is_contiguous() or alpha.shape[-1] != alpha.shape[-2]:
        raise RuntimeError('bp_hip.softmax_bwd_causal_: alpha and dalpha must be equal-shaped contiguous (..., S, S)')
    s = alpha.shape[-1]
    with torch.cuda.device(alpha.device):
        code = lib().bp_softmax_bwd_causal(alpha.data_ptr(), dalpha.data_ptr(), alpha.numel() // (s * s), s,
                                           float(softmax_scale), _dtype_code(alpha), _stream())
    _check(code, 'bp_softmax_bwd_causal')
    return dalpha


SLAB = 128   # queries per slab of the dq / dk backward (fixed by the kernels)


def sense_mix_dc(qk, dout, lse, softmax_scale, like):
    """dcontent (B,S,k,d_out) of sense_mix: bp_sense_mix_dc (alpha recomputed from the saved lse, never stored).
    `like`: the forward's content tensor (shape / dtype / device of the result)."""
    b, s, k, dk = _check_qk(qk)
    dcontent = torch.empty((b, s, k, like.shape[-1]), dtype=like.dtype, device=like.device)
    queue_ws = _queue_ws(qk.device)
    with torch.cuda.device(qk.device):
        code = lib().bp_sense_mix_dc(
            qk.data_ptr(), dout.data_ptr(), lse.data_ptr(), dcontent.data_ptr(), b, s, k, dk, like.shape[-1],
            qk.stride(0), qk.stride(1), qk.stride(2), qk.stride(3), dout.stride(0), dout.stride(1),
            dcontent.stride(0), dcontent.stride(1), dcontent.stride(2), float(softmax_scale), _dtype_code(qk),
            queue_ws.data_ptr(), _stream())
    del queue_ws
    _check(code, 'bp_sense_mix_dc')
    return dcontent


def sense_dqk(qk, content, dout, lse, softmax_scale):
    """dqk (like qk) of sense_mix.  The 768-deep product dP_l[t,s] = dout[t].C[s,l] is a plain GEMM and goes to the
    BLAS library, slab by slab (128 queries), TRANSPOSED into a (B, S*k, 128) buffer that is reused: the content's
    own storage (B, S*k, d) is its left operand as it stands, so nothing is copied or permuted.  bp_sense_dq_dk then
    turns each slab into dq rows and dk contributions (softmax backward + the two thin products) on the fly."""
    b, s, k, dk = _check_qk(qk)
    d = content.shape[-1]
    c_flat = content.reshape(b, s * k, d)                      # a view for the (B,S,k,d) storage layout
    dqk = torch.empty_like(qk)
    dk_acc = torch.zeros((b, s, k, dk), dtype=torch.float32, device=qk.device)
    dsum = torch.empty((b, k, round_up(s, 16)), dtype=torch.float32, device=qk.device)
    buf = torch.empty(b * s * k * SLAB, dtype=qk.dtype, device=qk.device)
    code_dt, stream = _dtype_code(qk), _stream()
    for t0 in range(0, s, SLAB):
        n = min(s, t0 + SLAB) * k
        rows = dout[:, t0:t0 + SLAB]
        if rows.shape[1] < SLAB:                               # last, partial slab: pad the queries with zeros
            rows = torch.nn.functional.pad(rows, (0, 0, 0, SLAB - rows.shape[1]))
        dpt = buf[:b * n * SLAB].view(b, n, SLAB)
        torch.bmm(c_flat[:, :n], rows.transpose(1, 2), out=dpt)
        with torch.cuda.device(qk.device):
            code = lib().bp_sense_dq_dk(
                qk.data_ptr(), dpt.data_ptr(), lse.data_ptr(), dsum.data_ptr(), dqk.data_ptr(), dk_acc.data_ptr(),
                b, s, k, dk, t0, qk.stride(0), qk.stride(1), qk.stride(2), qk.stride(3), dpt.stride(0),
                dqk.stride(0), dqk.stride(1), dqk.stride(3), dk_acc.stride(0), dk_acc.stride(1), dk_acc.stride(2),
                float(softmax_scale), code_dt, stream)
        _check(code, 'bp_sense_dq_dk')
    dqk[:, :, 1] = dk_acc
    return dqk


def _fused_mix_backward_ok(qk, content, key_weight):
    return (key_weight is None and qk.shape[-1] % 8 == 0 and qk.shape[-1] <= 128 and content.shape[-1] % 8 == 0
            and qk.is_contiguous()
            and content.stride(-1) == 1 and content.stride(2) == content.shape[-1]
            and content.stride(1) == content.shape[2] * content.shape[-1]
            and content.stride(0) == content.shape[1] * content.stride(1) and qk.shape[1] <= 65536)


class SenseMixFn(torch.autograd.Function):
    """Differentiable fused sense contraction.  Forward: LSE pre-pass + bp_sense_mix_weighted (alpha never stored).
    Backward (second half), alpha recomputed from the saved LSE inside the kernels:
        dC_l = alpha_l^T dout                                   bp_sense_mix_dc   (the forward kernel, roles swapped)
        dP_l = dout C_l^T   (slabs of 128 queries, BLAS)  ->  dS = alpha (dP - rowsum(alpha dP))
        dq_l = scale dS k_l,  dk_l = scale dS^T q_l             bp_sense_dq_dk
    Peak extra memory: one (B, S*k, 128) 16-bit slab; the reference's autograd keeps three (B,k,S,S) fp32 tensors
    alive from the forward on.  With an intervention `key_weight` (inference-time experiments) or shapes the fused
    kernels do not take, the older alpha-rebuilding formulation below runs instead."""

    @staticmethod
    def forward(ctx, qk, content, softmax_scale, key_weight):
        scale = softmax_scale or qk.shape[-1] ** -0.5
        lse = sense_lse(qk, scale)
        out = sense_mix(qk, content, scale, lse=lse, key_weight=key_weight)
        ctx.save_for_backward(qk, content, lse, key_weight)
        ctx.scale = scale
        ctx.fallback_recorded = eager_fallback_allowed()
        return out

    @staticmethod
    def backward(ctx, dout):
        qk, content, lse, key_weight = ctx.saved_tensors
        dout = dout.contiguous()
        if not _fused_mix_backward_ok(qk, content, key_weight):
            # wide senses (128 < d_k <= 640: csrc/sense_wide.hip) have no fused backward and need none: with few senses
            # alpha is small (k S^2 per sample), so the alpha-rebuilding route IS their backward -- no opt-in asked
            wide = key_weight is None and qk.shape[-1] > 128
            if not wide and not eager_fallback_allowed(ctx):
                raise RuntimeError(
                    'bp_hip.SenseMixFn.backward: the fused backward kernels take d_k % 8 == 0, d_out % 8 == 0, a contiguous '
                    '(B,S,k,d) content and no key_weight; this call would take the alpha-rebuilding route (two (B,k,S,S) '
                    'buffers + BLAS GEMMs).  Opt in with `with bp_hip.allow_eager_fallback():` around the forward or around backward().')
            return _sense_mix_backward_rebuild(ctx, qk, content, lse, key_weight, dout)
        dcontent = sense_mix_dc(qk, dout, lse, ctx.scale, content) if ctx.needs_input_grad[1] else None
        dqk = sense_dqk(qk, content, dout, lse, ctx.scale) if ctx.needs_input_grad[0] else None
        return dqk, dcontent, None, None


def _sense_mix_backward_rebuild(ctx, qk, content, lse, key_weight, dout):
    """alpha rebuilt once by bp_sense_alpha from the saved LSE (two (B,k,S,S) 16-bit buffers), batched GEMMs,
    bp_softmax_bwd_causal in place, two thin GEMMs: the round-1 formulation, kept for key_weight / odd shapes."""
    b, s, _, k, dk = qk.shape
    alpha = sense_alpha(qk, ctx.scale, lse=lse)                               # (B,k,S,S)
    weighted = alpha if key_weight is None else alpha * key_weight.unsqueeze(2).to(alpha.dtype)
    g = dout.unsqueeze(1)                                                      # (B,1,S,d)
    dcontent = None
    if ctx.needs_input_grad[1]:
        dcontent = torch.matmul(weighted.transpose(2, 3), g).transpose(1, 2)   # (B,S,k,d) view of (B,k,S,d)
    dqk = None
    if ctx.needs_input_grad[0]:
        dalpha = torch.matmul(g, content.permute(0, 2, 3, 1))                  # (B,k,S_t,S_s)
        if key_weight is not None:
            dalpha = dalpha * key_weight.unsqueeze(2).to(dalpha.dtype)
        dalpha = dalpha.contiguous()
        if softmax_bwd_causal_supported(alpha):
            ds = softmax_bwd_causal_(alpha, dalpha, ctx.scale)
        else:
            a32, d32 = alpha.float(), dalpha.float()
            ds = (ctx.scale * a32 * (d32 - (a32 * d32).sum(-1, keepdim=True))).to(alpha.dtype)
        q, kk = qk[:, :, 0].transpose(1, 2), qk[:, :, 1].transpose(1, 2)       # (B,k,S,dk)
        dq = torch.matmul(ds, kk)
        dkk = torch.matmul(ds.transpose(2, 3), q)
        dqk = torch.stack([dq.transpose(1, 2), dkk.transpose(1, 2)], dim=2)   # (B,S,2,k,dk)
    return dqk, dcontent, None, None


def sense_mix_autograd(qk, content, softmax_scale=None, key_weight=None):
    """sense_mix that records a backward when gradients are enabled (training), the plain kernel otherwise."""
    if torch.is_grad_enabled() and (qk.requires_grad or content.requires_grad):
        return SenseMixFn.apply(qk, content, softmax_scale, key_weight)
    return sense_mix(qk, content, softmax_scale, key_weight=key_weight)


class SenseAlphaFn(torch.autograd.Function):
    """Differentiable bp_sense_alpha: backward = bp_softmax_bwd_causal + the two thin GEMMs."""

    @staticmethod
    def forward(ctx, qk, softmax_scale):
        scale = softmax_scale or qk.shape[-1] ** -0.5
        alpha = sense_alpha(qk, scale)
        ctx.save_for_backward(qk, alpha)
        ctx.scale = scale
        return alpha

    @staticmethod
    def backward(ctx, dalpha):
        qk, alpha = ctx.saved_tensors
        dalpha = dalpha.contiguous().clone()
        if softmax_bwd_causal_supported(alpha):
            ds = softmax_bwd_causal_(alpha, dalpha, ctx.scale)
        else:
            a32, d32 = alpha.float(), dalpha.float()
            ds = (ctx.scale * a32 * (d32 - (a32 * d32).sum(-1, keepdim=True))).to(alpha.dtype)
        q, kk = qk[:, :, 0].transpose(1, 2), qk[:, :, 1].transpose(1, 2)
        dq = torch.matmul(ds, kk)
        dkk = torch.matmul(ds.transpose(2, 3), q)
        return torch.stack([dq.transpose(1, 2), dkk.transpose(1, 2)], dim=2), None


def sense_alpha_autograd(qk, softmax_scale=None):
    if torch.is_grad_enabled() and qk.requires_grad:
        return SenseAlphaFn.apply(qk, softmax_scale)
    return sense_alpha(qk, softmax_scale)


LN_BWD_WS_ROWS = 1024


def add_layer_norm_bwd_supported(x0_dtype, cols):
    return x0_dtype in (torch.float16, torch.bfloat16, torch.float32) and cols % 4 == 0 and cols <= 2048


def add_layer_norm_bwd(dz, dx_in, x, weight, eps, want_dx1, dropout_p=0.0, rng_state=None, rowscale=None, colscale=None,
                       x0=None):
    """Backward of add_layer_norm: (dx0 in dz's dtype, dx1 in x's dtype or None, dweight, dbias[, dcolscale]).
    dz (..., cols) in the forward's x0 / z dtype; dx_in gradient of the residual output (x's dtype) or None; x
    the summed stream the forward normalised (fp32 or dz's dtype).  With dropout, (dropout_p, rng_state) are the
    forward's: dx0 passes the regenerated mask and the 1 / (1 - p) scale.  rowscale / colscale: the forward's; with a
    colscale the forward's x0 is needed too and dcolscale is appended.  Replaces dropout_add_ln_bwd
    (flash_attn/ops/layer_norm.py:27-52)."""
    _require_cuda(dz, dx_in, x, weight, x0)
    cols = dz.shape[-1]
    dzc, xc = dz.contiguous(), x.contiguous()
    dxc = dx_in.contiguous() if dx_in is not None else None
    if xc.dtype not in (torch.float32, dzc.dtype) or (dxc is not None and dxc.dtype != xc.dtype):
        raise RuntimeError('bp_hip.add_layer_norm_bwd: x / dx_in must be fp32 or dz\'s dtype, and agree')
    if float(dropout_p) > 0.0 and rng_state is None:
        raise RuntimeError('bp_hip.add_layer_norm_bwd: dropout_p > 0 needs the rng_state the forward used')
    dropout_p, rng_state, rng_ptr = _dropout_args(dropout_p, rng_state, dz.device)
    rowscale, colscale = _ln_scales(dzc, weight, rowscale, colscale)
    x0c = None
    if colscale is not None:
        if x0 is None or x0.shape != dz.shape or x0.dtype != dz.dtype:
            raise RuntimeError('bp_hip.add_layer_norm_bwd: the gradient of colscale needs the forward\'s x0')
        x0c = x0.contiguous()
    code_dt, _ = _ln_16bit_code(dzc, xc, weight)
    rows = dzc.numel() // cols
    dx0 = torch.empty_like(dzc)
    dx1 = torch.empty_like(xc) if want_dx1 else None
    dw, db = torch.empty_like(weight), torch.empty_like(weight)
    dcs = torch.empty_like(weight) if colscale is not None else None
    ws = torch.empty(int(lib().bp_ln_bwd_ws_floats(cols, int(colscale is not None))), dtype=torch.float32, device=dz.device)
    with torch.cuda.device(dz.device):
        code = lib().bp_dropout_add_layer_norm_scaled_bwd(
            dzc.data_ptr(), dxc.data_ptr() if dxc is not None else None, xc.data_ptr(),
            x0c.data_ptr() if x0c is not None else None, weight.data_ptr(),
            rowscale.data_ptr() if rowscale is not None else None, colscale.data_ptr() if colscale is not None else None,
            dx0.data_ptr(), dx1.data_ptr() if dx1 is not None else None, dw.data_ptr(), db.data_ptr(),
            dcs.data_ptr() if dcs is not None else None, ws.data_ptr(), ws.numel(), rows, cols, float(eps), code_dt,
            int(dzc.dtype == torch.float32), int(xc.dtype == torch.float32), int(weight.dtype == torch.float32),
            dropout_p, rng_ptr, _stream())
    _check(code, 'bp_dropout_add_layer_norm_scaled_bwd')
    return (dx0, dx1, dw, db) if colscale is None else (dx0, dx1, dw, db, dcs)


def _xent_dtype(t):
    code = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}.get(t.dtype)
    if code is None:
        raise RuntimeError(f'bp_hip: cross entropy takes fp16 / bf16 / fp32 logits, got {t.dtype}')
    return code


def xentropy_fwd(logits, labels, smoothing=0.0, total_classes=-1):
    """(losses, lse), both fp32 (rows,): the reference's xentropy_cuda_lib.forward
    (flash_attn/losses/cross_entropy.py:37,54).  logits (rows, cols), last dim contiguous; labels int64."""
    _require_cuda(logits, labels)
    if logits.dim() != 2 or logits.stride(1) != 1 or labels.shape != (logits.shape[0],):
        raise RuntimeError('bp_hip.xentropy_fwd: logits (rows, cols) with unit last stride, labels (rows,)')
    labels = labels.to(torch.int64).contiguous()
    rows, cols = logits.shape
    losses = torch.empty(rows, dtype=torch.float32, device=logits.device)
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        code = lib().bp_xentropy_fwd(logits.data_ptr(), labels.data_ptr(), losses.data_ptr(), lse.data_ptr(),
                                     rows, cols, logits.stride(0), float(smoothing), int(total_classes),
                                     _xent_dtype(logits), _stream())
    _check(code, 'bp_xentropy_fwd')
    return losses, lse


def xentropy_bwd(grad_losses, logits, lse, labels, smoothing=0.0, inplace=False, total_classes=-1):
    """d loss / d logits in the logits' dtype; `inplace` overwrites `logits` (upstream's inplace_backward,
    cross_entropy.py:103-105)."""
    _require_cuda(grad_losses, logits, lse, labels)
    rows, cols = logits.shape
    grad = logits if inplace else torch.empty_like(logits)
    g = grad_losses.to(torch.float32).contiguous()
    labels = labels.to(torch.int64).contiguous()
    with torch.cuda.device(logits.device):
        code = lib().bp_xentropy_bwd(g.data_ptr(), logits.data_ptr(), lse.data_ptr(), labels.data_ptr(),
                                     grad.data_ptr(), rows, cols, logits.stride(0), grad.stride(0),
                                     float(smoothing), int(total_classes), _xent_dtype(logits), _stream())
    _check(code, 'bp_xentropy_bwd')
    return grad


def bias_gelu_supported(x):
    """Shapes / dtypes bp_bias_gelu_* and bp_column_sum take (callers use the torch expressions otherwise)."""
    return x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.shape[-1] % 8 == 0 and x.numel() > 0


def _rows_cols(x):
    if x.stride(-1) != 1 or not x.is_contiguous():
        raise RuntimeError('bp_hip: bias/GELU kernels take contiguous (rows, cols) tensors')
    return x.numel() // x.shape[-1], x.shape[-1]


def bias_gelu_fwd(x, bias=None, save_pre=False, out=None):
    """y = gelu_tanh(x + bias) for a contiguous 16-bit (..., cols) tensor; returns (y, pre) where pre = x + bias
    (16-bit) when save_pre -- without a bias, pre IS x.  `out` may be x itself (in place).  C ABI bp_bias_gelu_fwd:
    the elementwise half of the reference's `linear_gelu_forward` (flash_attn/ops/fused_dense.py:220)."""
    _require_cuda(x, bias, out)
    rows, cols = _rows_cols(x)
    if bias is not None and (bias.shape != (cols,) or bias.dtype != x.dtype or not bias.is_contiguous()):
        raise RuntimeError('bp_hip.bias_gelu_fwd: bias must be a contiguous (cols,) tensor of x\'s dtype')
    y = torch.empty_like(x) if out is None else out
    pre = torch.empty_like(x) if (save_pre and bias is not None) else None
    with torch.cuda.device(x.device):
        code = lib().bp_bias_gelu_fwd(x.data_ptr(), bias.data_ptr() if bias is not None else None,
                                      pre.data_ptr() if pre is not None else None, y.data_ptr(), rows, cols,
                                      _dtype_code(x), _stream())
    _check(code, 'bp_bias_gelu_fwd')
    return y, (pre if pre is not None else (x if save_pre else None))


def _bias_grad_ws(rows, cols, device):
    return torch.empty(int(lib().bp_bias_grad_ws_floats(rows, cols)), dtype=torch.float32, device=device)


def bias_gelu_bwd(grad, pre, bias_grad_dtype=None, inplace=False):
    """(dpre, dbias): dpre = grad * gelu_tanh'(pre) in grad's dtype (written over `grad` when inplace) and, when
    bias_grad_dtype is given (fp32 or grad's dtype), dbias = column sums of dpre -- one pass over the two tensors, the
    reference's `bias_gelu_linear_dgrad_bgrad` epilogue (flash_attn/ops/fused_dense.py:290) minus the GEMM."""
    _require_cuda(grad, pre)
    rows, cols = _rows_cols(grad)
    if pre.shape != grad.shape or pre.dtype != grad.dtype or not pre.is_contiguous():
        raise RuntimeError('bp_hip.bias_gelu_bwd: pre must match grad')
    dpre = grad if inplace else torch.empty_like(grad)
    dbias = ws = None
    if bias_grad_dtype is not None:
        if bias_grad_dtype not in (torch.float32, grad.dtype):
            raise RuntimeError('bp_hip.bias_gelu_bwd: bias gradient dtype must be fp32 or grad\'s dtype')
        dbias = torch.empty(cols, dtype=bias_grad_dtype, device=grad.device)
        ws = _bias_grad_ws(rows, cols, grad.device)
    with torch.cuda.device(grad.device):
        code = lib().bp_bias_gelu_bwd(grad.data_ptr(), pre.data_ptr(), dpre.data_ptr(),
                                      dbias.data_ptr() if dbias is not None else None,
                                      ws.data_ptr() if ws is not None else None, rows, cols, _dtype_code(grad),
                                      int(bias_grad_dtype == torch.float32), _stream())
    _check(code, 'bp_bias_gelu_bwd')
    return dpre, dbias


def column_sum(grad, out_dtype=None):
    """dbias (cols,) = sum over the rows of a contiguous 16-bit (..., cols) tensor, deterministic (C ABI
    bp_column_sum): the bias-gradient half of the reference's `linear_bias_wgrad` (fused_dense.py:66,281)."""
    _require_cuda(grad)
    rows, cols = _rows_cols(grad)
    out_dtype = out_dtype or grad.dtype
    if out_dtype not in (torch.float32, grad.dtype):
        raise RuntimeError('bp_hip.column_sum: output dtype must be fp32 or grad\'s dtype')
    dbias = torch.empty(cols, dtype=out_dtype, device=grad.device)
    ws = _bias_grad_ws(rows, cols, grad.device)
    with torch.cuda.device(grad.device):
        code = lib().bp_column_sum(grad.data_ptr(), dbias.data_ptr(), ws.data_ptr(), rows, cols, _dtype_code(grad),
                                   int(out_dtype == torch.float32), _stream())
    _check(code, 'bp_column_sum')
    return dbias


class GraphedForward:
    """One forward of `module` captured in a HIP graph (torch.cuda.CUDAGraph -> hipGraph) and replayed:
    for launch-bound shapes (Backpack-Micro, B=4, S=128: ~100 launches of ~20 us each) replay is 4x
    faster than eager launching; at the headline shape the kernels are long and it changes nothing.
    Every C-ABI launch of this package goes to torch's current stream, so it is captured like a torch op.

        fwd = bp_hip.GraphedForward(model, example_ids)     # static shapes
        logits = fwd(ids)                                   # copies ids into the static input, replays
    The returned tensor is the graph's static output buffer (overwritten by the next call).
    """

    def __init__(self, module, example_input, select=lambda out: getattr(out, 'logits', out)):
        self.static_in = example_input.clone()
        self.select = select
        # caches a module keeps OUTSIDE the graph (the Backpack model's whole-vocabulary sense table): refreshed in place
        # in front of every replay, so in-place weight updates reach the replays as they reach the captured kernels
        self.refresh = getattr(module, 'refresh_inference_caches', None)
        side = torch.cuda.Stream(device=example_input.device)
        side.wait_stream(torch.cuda.current_stream(example_input.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):              # warm-up outside capture (lazy inits, workspace allocations)
                module(self.static_in)
        torch.cuda.current_stream(example_input.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = select(module(self.static_in))
        # The captured kernels hold the ADDRESS of such a cache.  Pin its storage (BackpackModel.train() would otherwise
        # free the table, and the refresh after the next eval() would build it somewhere else) and remember it: a replay
        # whose refresh did not land in this storage is refused instead of reading a freed block.
        self._pinned = []
        for m in module.modules():
            if hasattr(m, 'pin_sense_table') and getattr(m, '_sense_table', None) is not None:
                m.pin_sense_table()
                self._pinned.append((m, m._sense_table[1]))
        self.module = module

    def __call__(self, x):
        if self.module.training and self._pinned:
            raise RuntimeError('bp_hip.GraphedForward: the graph was captured in eval mode with the cached sense table; '
                               'call module.eval() before replaying it')
        self.static_in.copy_(x)
        if self.refresh is not None:
            with torch.no_grad():
                self.refresh()
        for m, table in self._pinned:
            now = m._sense_table[1] if m._sense_table is not None else None
            if now is None or now.data_ptr() != table.data_ptr() or m._sense_table[0] is None:
                raise RuntimeError('bp_hip.GraphedForward: the sense table the graph reads was replaced or could not be '
                                   'refreshed in place; capture a new GraphedForward')
        self.graph.replay()
        return self.static_out
