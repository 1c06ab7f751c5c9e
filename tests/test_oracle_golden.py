"""CPU: the oracle (oracle/ref_cpu.py) against the golden vectors captured from the real
reference (tests/golden/make_golden.py).  fp32 eager ops are deterministic here, so the bar is
tight (1e-5); the stored -inf-mask variant documents the <=5e-6 gap to the -10000 additive mask."""
import numpy as np
import torch

from conftest import from_bits16, load_golden
from oracle import ref_cpu as R

TAGS = ['dk24', 'dk48', 'dk40', 'dk10']


def test_g1_g2_sense_alpha_and_mix():
    g = load_golden('g12_sense.npz')
    for tag in TAGS:
        w, b, h = from_bits16(g[f'{tag}_w']), from_bits16(g[f'{tag}_b']), from_bits16(g[f'{tag}_h'])
        k = int(g[f'{tag}_k'])
        alpha = R.context_self_attn(h, w, b, k)
        assert (alpha - torch.from_numpy(g[f'{tag}_alpha'])).abs().max().item() < 1e-6
        s = alpha.shape[-1]
        assert torch.count_nonzero(torch.triu(alpha, 1)) == 0          # bit-exact causal zeros
        assert torch.allclose(alpha.sum(-1), torch.ones(alpha.shape[:-1]), atol=1e-5)
        content = from_bits16(g[f'{tag}_content']).transpose(1, 2)      # (B,k,S,dout) view
        mixed = R.sense_mix(alpha, content)
        assert (mixed - torch.from_numpy(g[f'{tag}_mixed'])).abs().max().item() < 1e-5
        # einsum restatement identity of the contraction
        ein = torch.einsum('blts,blsd->btd', alpha, content)
        assert (ein - mixed).abs().max().item() < 1e-5
        fused = R.sense_mix_from_qk_fp32(torch.nn.functional.linear(h, w, b).reshape(
            h.shape[0], h.shape[1], 2, k, -1), content)
        assert (fused - mixed).abs().max().item() < 1e-5


def test_g9_wide_sense_alpha_and_mix():
    """G9 (make_golden_r6.py): the reference's ContextSelfAttn and sense combination at the widths of its few-sense
    ablation configs (d_k = 160 / 640), the shapes csrc/sense_wide_dma.hip runs."""
    g = load_golden('g9_wide_sense.npz')
    for tag in ('dk160', 'dk640'):
        w, b, h = from_bits16(g[f'{tag}_w']), from_bits16(g[f'{tag}_b']), from_bits16(g[f'{tag}_h'])
        k = int(g[f'{tag}_k'])
        assert w.shape[1] // k == int(tag[2:])
        alpha = R.context_self_attn(h, w, b, k)
        assert (alpha - torch.from_numpy(g[f'{tag}_alpha'])).abs().max().item() < 1e-6
        assert torch.count_nonzero(torch.triu(alpha, 1)) == 0
        content = from_bits16(g[f'{tag}_content']).transpose(1, 2)
        mixed = R.sense_mix(alpha, content)
        assert (mixed - torch.from_numpy(g[f'{tag}_mixed'])).abs().max().item() < 1e-5
        fused = R.sense_mix_from_qk_fp32(torch.nn.functional.linear(h, w, b).reshape(
            h.shape[0], h.shape[1], 2, k, -1), content)
        assert (fused - mixed).abs().max().item() < 1e-5


def test_g3_trunk_attention():
    g = load_golden('g3_trunk_attn.npz')
    for tag in ('h64', 'h80'):
        qkv = from_bits16(g[f'{tag}_qkv'])
        dh = qkv.shape[-1]
        for layer in (0, 5, 11):
            scale = dh ** -0.5 / (layer + 1)
            want = torch.from_numpy(g[f'{tag}_L{layer}_out'])
            got = R.self_attention_eager(qkv, True, scale)
            assert (got - want).abs().max().item() < 1e-6
            o, attn, lse = R.attention_fp32(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True,
                                            softmax_scale=scale)
            assert (o - want).abs().max().item() < 5e-6              # -inf mask vs -10000 mask
            assert (lse - torch.from_numpy(g[f'{tag}_L{layer}_lse'])).abs().max().item() < 1e-5
            assert torch.count_nonzero(torch.triu(attn, 1)) == 0
        kpm = torch.from_numpy(g[f'{tag}_kpm'])
        got = R.self_attention_eager(qkv, False, None, kpm)
        assert (got - torch.from_numpy(g[f'{tag}_kpm_out'])).abs().max().item() < 1e-6


def test_g4_whole_model_nano():
    g = load_golden('g4_nano_model.npz')
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    cfg = dict(n_embd=64, n_head=2, n_layer=2, num_content_vectors=4,
               layer_norm_epsilon=float(g['layer_norm_epsilon']), scale_attn_by_inverse_layer_idx=True)
    st = R.backpack_forward(sd, cfg, torch.from_numpy(g['ids']), return_stages=True)
    for name in ('trunk', 'alpha', 'content', 'hidden', 'logits'):
        err = (st[name] - torch.from_numpy(g[name])).abs().max().item()
        assert err < 2e-5, (name, err)


def test_g5_varlen():
    g = load_golden('g5_varlen.npz')
    qkv = from_bits16(g['qkv_unpad'])
    cu = torch.from_numpy(g['cu_seqlens'])
    for causal in (False, True):
        out, lses = R.varlen_attention_fp32(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, cu, causal=causal)
        assert (out - torch.from_numpy(g[f'out_causal{int(causal)}'])).abs().max().item() < 1e-6
        lse = torch.cat(lses, dim=1)
        assert (lse - torch.from_numpy(g[f'lse_causal{int(causal)}'])).abs().max().item() < 1e-5


def test_empty_rows_and_edge_cases():
    # a sequence with no keys: zeros out, -inf LSE (fmha_fprop_kernel_1xN.h:592-596)
    q = torch.randn(5, 2, 16)
    k = torch.randn(0, 2, 16)
    out, lses = R.varlen_attention_fp32(q, k, k, torch.tensor([0, 5]), torch.tensor([0, 0]))
    assert torch.count_nonzero(out) == 0 and torch.isinf(lses[0]).all()
    # S = 1: alpha is exactly 1
    qk = torch.randn(1, 1, 2, 4, 8)
    assert torch.equal(R.sense_alpha_from_qk(qk), torch.ones(1, 4, 1, 1))


def test_init_state_dict_matches_reference_parameter_count():
    cfg = R.make_config('small')
    sd = R.init_state_dict(cfg)
    n = sum(v.numel() for k, v in sd.items() if k != 'lm_head.weight')
    assert n == 170_482_944 or abs(n - 170.48e6) < 0.02e6, n   # README.md:89 "170M", SURVEY 8c


def test_intervention_oracle_against_reference_golden():
    """oracle/ref_cpu.py restatement of intervened_models.py vs the logits of the reference classes (G6)."""
    g = load_golden('g6_interventions.npz')
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd/')}
    ids = torch.from_numpy(g['ids'])
    cw = torch.from_numpy(g['content_weights'])
    cfg = dict(n_embd=64, n_head=2, n_layer=2, num_content_vectors=16,
               layer_norm_epsilon=float(g['layer_norm_epsilon']), scale_attn_by_inverse_layer_idx=True)
    scale = float(g['annealing_scale'])
    for tag, anneal in (('anneal', True), ('plain', False)):
        got = R.weighted_backpack_logits(sd, cfg, ids, cw, annealing_scale=scale, anneal=anneal)
        assert (got - torch.from_numpy(g['weighted_' + tag])).abs().max().item() < 5e-5
        got = R.negative_weighted_backpack_logits(sd, cfg, ids, cw, annealing_scale=scale, anneal=anneal)
        assert (got - torch.from_numpy(g['negative_' + tag])).abs().max().item() < 5e-5
    senses = {int(w): torch.from_numpy(g['sense/%d' % w]) for w in g['sense_words']}
    got = R.replaced_word_logits(sd, cfg, ids, senses)
    assert (got - torch.from_numpy(g['replaced'])).abs().max().item() < 5e-5


def test_cross_entropy_oracle_against_the_reference_tests_own_oracle():
    """oracle softmax_cross_entropy (+ grad) == torch.nn.CrossEntropyLoss(label_smoothing) on fp32 logits, which
    is what the reference's test pins its kernel to (tests/losses/test_cross_entropy.py:31-41)."""
    torch.manual_seed(0)
    x = torch.randn(64, 1000, requires_grad=True)
    y = torch.randint(0, 1000, (64,))
    y[::7] = -100
    for s in (0.0, 0.9):
        losses, lse = R.softmax_cross_entropy(x.detach(), y, s)
        want = torch.nn.functional.cross_entropy(x, y, label_smoothing=s, reduction='none')
        assert (losses - want.detach()).abs().max().item() < 2e-5
        assert (lse - torch.logsumexp(x.detach(), 1)).abs().max().item() < 1e-6
        g = torch.randn(64)
        (gx,) = torch.autograd.grad(want, x, g)
        got = R.softmax_cross_entropy_grad(g, x.detach(), y, s)
        assert (got - gx).abs().max().item() < 1e-6
