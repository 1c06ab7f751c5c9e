"""-m gpu: the overflow-retry branch of the fast attention forward (csrc/flash_fwd_dma.hip, `tile`).

The steady-state tile body keeps a row's reference maximum fixed and validates the tile afterwards: every lane's
partial row sum must stay <= 2^30 (bf16) / 2^14 (fp16); a tile that fails is redone by the exact online-softmax body
(true maximum, rescale of O and l) -- the step the reference takes for EVERY tile
(csrc/flash_attn/src/fmha_fprop_kernel_1xN.h:429-444, src/fmha/softmax.h:238-251).  Random N(0,1) inputs never
reach the limit, so these cases build score spikes in CLEAN key tiles (not tile 0, not the diagonal tile) and

  * replay the kernel's reference-point bookkeeping on the host (`replay_retries`) and ASSERT that the limit is
    crossed, by how much and in which tiles -- the test cannot silently stop covering the branch;
  * include a spike so large that skipping the retry would overflow the 16-bit P (fp16: > e^11.1, bf16: > e^88.7)
    and turn the row into inf/NaN, next to one just above the limit (only the replay proves that one);
  * compare O and the LSE with the fp32 oracle under the reference's 2x-eager rule, for the trunk kernel
    (HAS_V, head dims 64 / 128 / 40), the LSE-only variant behind every sense-mix call (bp_sense_lse) and the
    dropout variant (oracle given the documented mask).
"""
import math

import pytest
import torch

import philox_ref as P
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LIMIT = {torch.bfloat16: 2.0 ** 30, torch.float16: 2.0 ** 14}          # flash_fwd_dma.hip: ProbLimit
# nats above the row's reference point: (just above the limit, far above what 16-bit P could hold at all)
SPIKES = {torch.bfloat16: (22.5, 100.0), torch.float16: (10.2, 30.0)}
BM, BN = 128, 64                                                        # FlashDmaCfg::BM / BN
# The no-dropout trunk kernel sums the 16-bit ROUNDED probabilities (they are what multiplies V), so its LSE carries
# their rounding: at most half an ulp relative on the row sum = 2^-8 (bf16) / 2^-11 (fp16) absolute on the log, reached
# by rows with one or two keys; the inputs here (N(0, 0.5^2): flat rows) sit closer to that bound than N(0,1) ones.
LSE_TOL = {torch.bfloat16: 2.0 ** -8 + 1e-4, torch.float16: 1e-3}


def _bp():
    import bp_hip
    return bp_hip


def replay_retries(q, k, scale, causal, limit):
    """Host replay of the kernel's per-wave bookkeeping.  q (S,H,D), k (S,H,D) fp32 (holding 16-bit values).
    Returns [(head, first query of the wave, key tile, largest lane row sum / limit)] for every fast-body tile whose
    validation fails.  Mirrors flash_fwd_tile: wave = 32 queries; tile kb is exact (textbook step) when kb == 0 or
    kb >= my_clean_end; a lane sums the keys with bit 2 equal to its half-wave over both 32-key halves; a failed tile is
    repeated by the whole wave, but only the rows whose own sums failed take its maximum into their reference point."""
    s, h, _ = q.shape
    scores = torch.einsum('thd,shd->hts', q.double(), k.double()) * scale           # natural-log units
    half = ((torch.arange(BN) >> 2) & 1).bool()
    events = []
    for head in range(h):
        for q0 in range(0, s, 32):
            rows = torch.arange(q0, min(q0 + 32, s))
            qt = q0 // BM
            k_end = min(s, qt * BM + BM) if causal else s
            nkb = (k_end + BN - 1) // BN
            my_nkb = min(nkb, (q0 + 31) // BN + 1) if causal else nkb
            clean_end = min(s // BN, (q0 + 1) // BN) if causal else s // BN
            m = torch.full((len(rows),), float('-inf'), dtype=torch.float64)
            for kb in range(my_nkb):
                keys = torch.arange(kb * BN, min(kb * BN + BN, s))
                st = scores[head][rows][:, keys]
                if causal:
                    st = st.masked_fill(keys[None, :] > rows[:, None], float('-inf'))
                exact = kb == 0 or kb >= clean_end
                moves = torch.ones(len(rows), dtype=torch.bool)
                if not exact:
                    p = torch.exp(st - m[:, None])
                    lane = torch.stack([p[:, ~half[:len(keys)]].sum(1), p[:, half[:len(keys)]].sum(1)])
                    worst = lane.max().item()
                    if not worst <= limit:
                        events.append((head, q0, kb, worst / limit))
                        exact = True
                        moves = ~(lane <= limit).all(0)      # in a retry only the rows whose own sums failed move
                if exact:
                    m = torch.where(moves, torch.maximum(m, st.max(1).values), m)
    return events


def spiked_qk(s, h, d, dtype, seed, spikes):
    """q, k (1,S,H,D) in `dtype`, N(0, 0.5^2); then, one after the other, for each (head, query, key, nats):
    k[key] = q[query] * f with f chosen so that this pair's scaled score lies `nats` above the largest score the row
    has met in the key tiles before the spike's tile -- an upper bound of the kernel's reference point there, so the
    fast body's p for this pair is >= e^nats (and <= e^(nats + ~0.5): the reference point is the first tile's
    maximum unless an earlier retry moved it up)."""
    gen = torch.Generator().manual_seed(seed)
    q = (torch.randn(1, s, h, d, generator=gen) * 0.5).to(dtype)
    k = (torch.randn(1, s, h, d, generator=gen) * 0.5).to(dtype)
    scale = d ** -0.5
    for head, query, key, nats in spikes:
        qv = q[0, query, head].float()
        seen = (k[0, :key // BN * BN, head].float() @ qv).max().item() * scale
        k[0, key, head] = (qv * (seen + nats) / (qv.dot(qv).item() * scale)).to(dtype)
    return q, k


def _check(got, ref32, eager, name, atol=1e-5):
    err = (got.float().cpu() - ref32.float()).abs().max().item()
    base = (eager.float() - ref32.float()).abs().max().item()
    print(f'{name}: err {err:.3e} eager {base:.3e}')
    assert torch.isfinite(got.float()).all(), name
    assert err <= 2 * base + atol, (name, err, base)


def _assert_branch_is_driven(events, dtype, expect):
    """`expect`: {(head, wave q0, key tile)} that must fail the validation; the largest excess must be beyond what
    the 16-bit type could even represent, the smallest still a clear factor above the limit."""
    assert {e[:3] for e in events} >= expect, (events, expect)
    assert all(e[3] >= 1.5 for e in events if e[:3] in expect), events      # clear of rounding doubt
    overflow = {torch.float16: 65504.0, torch.bfloat16: 3.0e38}[dtype] / LIMIT[dtype]
    assert max(e[3] for e in events) > overflow, events


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('d', [64, 128, 40])
def test_flash_fwd_retry_branch(d, dtype):
    bp = _bp()
    s, h = 512, 2
    lo, hi = SPIKES[dtype]
    # (head, query, key, nats): query 450 sits in wave q0 = 448 (clean tiles 1..6), query 300 in wave 288 (clean 1..3)
    spikes = [(0, 450, 200, lo), (0, 450, 330, hi), (1, 300, 150, hi), (1, 460, 140, lo)]
    q, k = spiked_qk(s, h, d, dtype, 11 + d, spikes)
    v = torch.randn(1, s, h, d, generator=torch.Generator().manual_seed(3)).to(dtype)
    scale = d ** -0.5
    events = replay_retries(q[0].float(), k[0].float(), scale, True, LIMIT[dtype])
    _assert_branch_is_driven(events, dtype, {(0, 448, 3), (0, 448, 5), (1, 288, 2), (1, 448, 2)})
    assert all(0 < kb < (q0 + 1) // BN for _, q0, kb, _ in events)          # clean tiles only, never tile 0

    ref32, _, lse_ref = R.attention_fp32(q, k, v, causal=True, softmax_scale=scale)
    eager = R.attention_fp32(q, k, v, causal=True, softmax_scale=scale, upcast=False, reorder_ops=True)[0]
    out = torch.empty(s, h, d, dtype=dtype, device=DEV)
    lse = bp.flash_fwd(q[0].to(DEV), k[0].to(DEV), v[0].to(DEV), out, None, None, s, s, scale, True)
    _check(out, R.attention_fp32(q.float(), k.float(), v.float(), causal=True, softmax_scale=scale)[0][0], eager[0],
           f'retry d={d} {dtype}')
    assert (lse[0, :, :s].cpu() - lse_ref[0]).abs().max().item() < LSE_TOL[dtype]
    assert abs(lse[0, 0, 450].item() - lse_ref[0, 0, 450].item()) < 1e-3        # one-hot row: l = 1 exactly
    # the spiked rows are one-hot on the spike key: the output IS that value row (to 16-bit rounding)
    assert (out[450, 0].float().cpu() - v[0, 330, 0].float()).abs().max().item() < 2e-2
    # non-causal sweep of the same inputs: every tile but the first is a fast-body tile
    ev_nc = replay_retries(q[0].float(), k[0].float(), scale, False, LIMIT[dtype])
    assert {(0, 448, 3), (0, 448, 5)} <= {e[:3] for e in ev_nc}
    ref32, _, lse_ref = R.attention_fp32(q.float(), k.float(), v.float(), causal=False, softmax_scale=scale)
    eager = R.attention_fp32(q, k, v, causal=False, softmax_scale=scale, upcast=False, reorder_ops=True)[0]
    lse = bp.flash_fwd(q[0].to(DEV), k[0].to(DEV), v[0].to(DEV), out, None, None, s, s, scale, False)
    _check(out, ref32[0], eager[0], f'retry non-causal d={d} {dtype}')
    assert (lse[0, :, :s].cpu() - lse_ref[0]).abs().max().item() < LSE_TOL[dtype]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('dk', [48, 16])
def test_sense_lse_and_mix_retry_branch(dk, dtype):
    """LSE-only variant (HAS_V = false): bp_sense_lse, and the fused mix / alpha that consume its result."""
    bp = _bp()
    s, k, d = 512, 4, 256
    lo, hi = SPIKES[dtype]
    spikes = [(0, 450, 200, lo), (0, 450, 330, hi), (2, 300, 150, hi), (3, 200, 70, lo)]
    q, kk = spiked_qk(s, k, dk, dtype, 5 + dk, spikes)
    scale = dk ** -0.5
    events = replay_retries(q[0].float(), kk[0].float(), scale, True, LIMIT[dtype])
    _assert_branch_is_driven(events, dtype, {(0, 448, 3), (0, 448, 5), (2, 288, 2), (3, 192, 1)})
    qk = torch.stack([q, kk], dim=2)                                        # (1, S, 2, k, dk)
    lse = bp.sense_lse(qk.to(DEV))
    _, alpha32, lse_ref = R.attention_fp32(q.float(), kk.float(), None, causal=True, softmax_scale=scale)
    assert torch.isfinite(lse[0, :, :s]).all()
    assert (lse[0, :, :s].cpu() - lse_ref[0]).abs().max().item() < 2e-3
    c = torch.randn(1, s, k, d, generator=torch.Generator().manual_seed(9)).to(dtype)
    want = torch.einsum('blts,bsld->btd', alpha32, c.float())
    alpha16 = R.attention_fp32(q, kk, None, causal=True, softmax_scale=scale, upcast=False, reorder_ops=True)[1]
    eager = torch.stack([alpha16[:, l] @ c[:, :, l] for l in range(k)]).sum(0)
    _check(bp.sense_mix(qk.to(DEV), c.to(DEV)), want, eager, f'mix after retried lse dk={dk} {dtype}', atol=2e-3)
    _check(bp.sense_alpha(qk.to(DEV)), alpha32, alpha16, f'alpha after retried lse dk={dk} {dtype}', atol=4e-3)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_flash_fwd_dropout_retry_branch(dtype):
    """DROP = true instantiation: same spikes, oracle given the documented keep mask (tests/philox_ref.py)."""
    bp = _bp()
    s, h, d, p_drop, seed, offset = 512, 2, 64, 0.17, 77, 12345
    lo, hi = SPIKES[dtype]
    spikes = [(0, 450, 200, lo), (0, 450, 330, hi), (1, 300, 150, hi)]
    q, k = spiked_qk(s, h, d, dtype, 21, spikes)
    v = torch.randn(1, s, h, d, generator=torch.Generator().manual_seed(4)).to(dtype)
    scale = d ** -0.5
    events = replay_retries(q[0].float(), k[0].float(), scale, True, LIMIT[dtype])
    _assert_branch_is_driven(events, dtype, {(0, 448, 3), (0, 448, 5), (1, 288, 2)})
    keep = torch.from_numpy(P.attention_keep_mask(seed, offset, 1, h, s, s, p_drop))
    kw = dict(causal=True, softmax_scale=scale, dropout_p=p_drop, dropout_mask=keep)
    ref32, _, lse_ref = R.attention_fp32(q.float(), k.float(), v.float(), **kw)
    eager = R.attention_fp32(q, k, v, upcast=False, reorder_ops=True, **kw)[0]
    out = torch.empty(s, h, d, dtype=dtype, device=DEV)
    rng = torch.tensor([seed, offset], dtype=torch.int64, device=DEV)
    lse = bp.flash_fwd(q[0].to(DEV), k[0].to(DEV), v[0].to(DEV), out, None, None, s, s, scale, True, p_drop, rng)
    _check(out, ref32[0], eager[0], f'retry dropout {dtype}')
    assert (lse[0, :, :s].cpu() - lse_ref[0]).abs().max().item() < 2e-3


def test_replay_finds_nothing_on_plain_inputs():
    """The replay itself: N(0,1) inputs (what every other test uses) never reach the limit -- which is why these
    cases exist."""
    torch.manual_seed(0)
    q, k = torch.randn(512, 2, 64).bfloat16().float(), torch.randn(512, 2, 64).bfloat16().float()
    assert replay_retries(q, k, 0.125, True, LIMIT[torch.float16]) == []
