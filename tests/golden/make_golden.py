"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference -- it gets the committed .npz files plus oracle/ref_cpu.py.

    python tests/golden/make_golden.py

What it does
  1. imports the reference Python with three arithmetic-neutral shims:
     transformers-5 generation aliases, a torchvision.ops.StochasticDepth identity stub,
     `backpack.FusedDense = nn.Linear` (FusedDense's CPU forward is F.linear).
  2. runs the reference modules on seeded inputs and writes inputs + outputs to .npz
  3. asserts that oracle/ref_cpu.py reproduces every one of them (max abs error printed and
     written to tests/golden/PINNING.txt).
The files hold data only: tensors in, tensors out, and (for the nano model) its weights.
"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


def import_reference():
    import transformers
    import transformers.generation as tg
    for name in ('GreedySearchDecoderOnlyOutput', 'SampleDecoderOnlyOutput'):
        if not hasattr(tg, name):
            setattr(tg, name, tg.GenerateDecoderOnlyOutput)

    class StochasticDepth(torch.nn.Module):
        def __init__(self, p, mode):
            super().__init__()
            self.p, self.mode = p, mode

        def forward(self, x):
            return x

    tv = types.ModuleType('torchvision')
    tv.__spec__ = importlib.machinery.ModuleSpec('torchvision', None)
    tvo = types.ModuleType('torchvision.ops')
    tvo.__spec__ = importlib.machinery.ModuleSpec('torchvision.ops', None)
    tvo.StochasticDepth = StochasticDepth
    tv.ops = tvo
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.ops', tvo)
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, 'training'))
    import src.models.backpack as bp
    bp.FusedDense = torch.nn.Linear
    import flash_attn.modules.mha as mha
    return bp, mha


def np32(t):
    return t.detach().to(torch.float32).cpu().numpy()


def bf16_exact(t):
    """Round to bf16-representable values (kept in fp32) so the 16-bit kernels consume the
    very same inputs as the fp32 reference run."""
    return t.detach().to(torch.bfloat16).to(torch.float32)


def bits16(t):
    """Store a bf16-exact fp32 tensor as its 16 bf16 bits (uint16) -- halves the fixture."""
    assert torch.equal(t.to(torch.bfloat16).to(torch.float32), t.float())
    return t.detach().to(torch.bfloat16).contiguous().view(torch.uint16).cpu().numpy()


def main():
    bp, mha = import_reference()
    from oracle import ref_cpu as R
    log = []

    def check(name, got, want, tol):
        err = (got.float() - want.float()).abs().max().item()
        log.append(f'{name}: oracle vs reference max|diff| = {err:.3e} (tol {tol:.1e})')
        print(log[-1])
        assert err <= tol, name

    torch.set_num_threads(8)

    # ---- G1/G2: ContextSelfAttn + sense mix, several (d, k) so d_k = 24, 48, 40, 10 -------------
    g12 = {}
    for tag, d, k, s in (('dk24', 384, 16, 64), ('dk48', 768, 16, 32), ('dk40', 160, 4, 32),
                         ('dk10', 160, 16, 48)):
        torch.manual_seed({'dk24': 1, 'dk48': 2, 'dk40': 3, 'dk10': 4}[tag])
        mod = bp.ContextSelfAttn(k, d)
        # make the scores non-trivial: default Linear init gives tiny logits
        with torch.no_grad():
            mod.Wqkv.weight.copy_(bf16_exact(mod.Wqkv.weight * 6.0))
            mod.Wqkv.bias.copy_(bf16_exact(torch.randn(2 * d) * 0.5))
        h = bf16_exact(torch.randn(2, s, d))
        dout = 96 if tag != 'dk24' else d
        content = bf16_exact(torch.randn(2, s, k * dout)).reshape(2, s, k, dout).transpose(1, 2)
        with torch.no_grad():
            alpha = mod(h)
            mixed = torch.sum(alpha @ content, dim=1)
        o_alpha = R.context_self_attn(h, mod.Wqkv.weight, mod.Wqkv.bias, k)
        check(f'G1 alpha {tag}', o_alpha, alpha, 1e-6)
        check(f'G2 mix {tag}', R.sense_mix(o_alpha, content), mixed, 1e-5)
        assert torch.count_nonzero(torch.triu(alpha, 1)) == 0  # strictly-upper triangle exactly 0
        g12.update({f'{tag}_w': bits16(mod.Wqkv.weight), f'{tag}_b': bits16(mod.Wqkv.bias),
                    f'{tag}_h': bits16(h), f'{tag}_content': bits16(content.transpose(1, 2).contiguous()),
                    f'{tag}_alpha': np32(alpha), f'{tag}_mixed': np32(mixed),
                    f'{tag}_k': np.int64(k)})
    np.savez_compressed(os.path.join(HERE, 'g12_sense.npz'), **g12)

    # ---- G3: trunk eager attention at the per-layer scales, d_h = 64 and 80 ---------------------
    g3 = {}
    for tag, heads, dh, s in (('h64', 4, 64, 128), ('h80', 4, 80, 96)):
        torch.manual_seed(7 + dh)
        qkv = bf16_exact(torch.randn(2, s, 3, heads, dh) * 1.5)
        g3[f'{tag}_qkv'] = bits16(qkv)
        for layer in (0, 5, 11):
            scale = dh ** -0.5 / (layer + 1)
            mod = mha.SelfAttention(causal=True, softmax_scale=scale)
            with torch.no_grad():
                out = mod(qkv)
            check(f'G3 eager {tag} L{layer}', R.self_attention_eager(qkv, True, scale), out, 1e-6)
            o32, _, lse = R.attention_fp32(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True,
                                           softmax_scale=scale)
            check(f'G3 -inf vs -10000 {tag} L{layer}', o32, out, 5e-6)
            g3[f'{tag}_L{layer}_out'] = np32(out)
            g3[f'{tag}_L{layer}_lse'] = np32(lse)
        # non-causal + key padding through the eager twin
        kpm = torch.arange(s)[None, :] < torch.tensor([[s], [s - 37]])
        mod = mha.SelfAttention(causal=False)
        with torch.no_grad():
            out = mod(qkv, key_padding_mask=kpm)
        check(f'G3 eager {tag} kpm', R.self_attention_eager(qkv, False, None, kpm), out, 1e-6)
        g3[f'{tag}_kpm'] = kpm.numpy()
        g3[f'{tag}_kpm_out'] = np32(out)
    np.savez_compressed(os.path.join(HERE, 'g3_trunk_attn.npz'), **g3)

    # ---- G4: whole model, "nano" config with the weights in the fixture ------------------------
    def build(cfg_kwargs):
        cfg = bp.BackpackConfig(**cfg_kwargs)
        model = bp.BackpackLMHeadModel(cfg).eval()
        return cfg, model

    nano = dict(n_embd=64, n_head=2, n_layer=2, num_content_vectors=4, vocab_size=96, n_positions=32,
                scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                use_flash_attn=False, fused_bias_fc=False, fused_dense_gelu_dense=False,
                fused_dropout_add_ln=False, pad_vocab_size_multiple=8)
    torch.manual_seed(0)
    cfg, model = build(nano)
    with torch.no_grad():  # default init gives near-uniform alpha; sharpen so the test has teeth
        model.transformer.contextualization_attn.Wqkv.weight.mul_(8.0)
        for layer in model.transformer.gpt2_model.layers:
            layer.mixer.Wqkv.weight.mul_(6.0)
    ids = torch.randint(0, 96, (2, 32), generator=torch.Generator().manual_seed(0))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        t = model.transformer
        h = t.gpt2_model(ids)
        alpha = t.contextualization_attn(h)
        content = t.content_model(ids)
        hidden = t(ids)
        logits = model(ids).logits
    ocfg = dict(n_embd=64, n_head=2, n_layer=2, num_content_vectors=4, layer_norm_epsilon=cfg.layer_norm_epsilon,
                scale_attn_by_inverse_layer_idx=True)
    st = R.backpack_forward(sd, ocfg, ids, return_stages=True)
    check('G4 nano trunk', st['trunk'], h, 2e-5)
    check('G4 nano alpha', st['alpha'], alpha, 1e-6)
    check('G4 nano content', st['content'], content, 2e-5)
    check('G4 nano hidden', st['hidden'], hidden, 2e-5)
    check('G4 nano logits', st['logits'], logits, 2e-5)
    out = {('sd/' + k): np32(v) for k, v in sd.items()}
    out.update(ids=ids.numpy(), trunk=np32(h), alpha=np32(alpha), content=np32(content),
               hidden=np32(hidden), logits=np32(logits),
               layer_norm_epsilon=np.float64(cfg.layer_norm_epsilon))
    np.savez_compressed(os.path.join(HERE, 'g4_nano_model.npz'), **out)

    # Micro at BASELINE config 1 shape: checked here, only a digest is stored (weights are 166 MB)
    torch.manual_seed(0)
    micro = dict(nano, n_embd=384, n_head=6, n_layer=6, num_content_vectors=16, vocab_size=50257,
                 n_positions=128)
    cfg, model = build(micro)
    ids = torch.randint(0, 50257, (4, 128), generator=torch.Generator().manual_seed(0))
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    with torch.no_grad():
        hidden = model.transformer(ids)
        logits = model(ids).logits
    ocfg = dict(n_embd=384, n_head=6, n_layer=6, num_content_vectors=16,
                layer_norm_epsilon=cfg.layer_norm_epsilon, scale_attn_by_inverse_layer_idx=True)
    st = R.backpack_forward(sd, ocfg, ids, return_stages=True)
    check('G4 micro hidden (B=4,S=128)', st['hidden'], hidden, 2e-5)
    check('G4 micro logits (B=4,S=128)', st['logits'], logits, 5e-5)
    nparam = sum(p.numel() for p in model.parameters())
    log.append(f'micro parameter count = {nparam}')

    # ---- G5: varlen through the reference's padding helpers -------------------------------------
    from flash_attn.bert_padding import unpad_input, pad_input
    torch.manual_seed(11)
    lens = torch.tensor([97, 128, 33, 1])
    smax, heads, dh = 128, 4, 64
    x = bf16_exact(torch.randn(4, smax, 3 * heads * dh))
    mask = torch.arange(smax)[None, :] < lens[:, None]
    x_unpad, indices, cu, max_s = unpad_input(x, mask)
    qkv_unpad = x_unpad.reshape(-1, 3, heads, dh)
    qkv = x.reshape(4, smax, 3, heads, dh)
    g5 = dict(qkv_unpad=bits16(qkv_unpad), cu_seqlens=cu.numpy(), lens=lens.numpy(),
              indices=indices.numpy(), max_s=np.int64(max_s))
    for causal in (False, True):
        o, _, lse = R.attention_fp32(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=causal,
                                     query_padding_mask=mask, key_padding_mask=mask)
        # the eager twin with its -10000 key-padding mask must agree on the valid rows
        with torch.no_grad():
            e = mha.SelfAttention(causal=causal)(qkv, key_padding_mask=mask)
        e = e.masked_fill(~mask[:, :, None, None], 0.0)
        check(f'G5 varlen causal={causal} (-inf oracle vs eager reference)', o, e, 5e-6)
        ov, lv = R.varlen_attention_fp32(qkv_unpad[:, 0], qkv_unpad[:, 1], qkv_unpad[:, 2], cu, cu,
                                         causal=causal)
        repadded = pad_input(ov.reshape(ov.shape[0], -1), indices, 4, smax).reshape(4, smax, heads, dh)
        check(f'G5 varlen causal={causal} (unpadded oracle vs padded)', repadded, o, 1e-6)
        g5[f'out_causal{int(causal)}'] = np32(ov)
        g5[f'lse_causal{int(causal)}'] = np.concatenate([np32(l) for l in lv], axis=1)
    np.savez_compressed(os.path.join(HERE, 'g5_varlen.npz'), **g5)

    # ---- G6: the control-experiment models of intervened_models.py on a 16-sense nano model ---------
    sys.path.insert(0, os.path.join(REF, 'training', 'src'))
    import src.models.intervened_models as im
    nano16 = dict(nano, num_content_vectors=16)
    torch.manual_seed(6)
    cfg6, model6 = build(nano16)
    with torch.no_grad():
        model6.transformer.contextualization_attn.Wqkv.weight.mul_(8.0)
        model6.transformer.content_model.final_mlp.fc2.weight.mul_(6.0)   # content logits large enough to anneal
    ids6 = torch.randint(0, 96, (2, 32), generator=torch.Generator().manual_seed(6))
    sd6 = {k: v.detach().clone() for k, v in model6.state_dict().items()}
    gen6 = torch.Generator().manual_seed(66)
    content_weights = torch.rand(96, 16, generator=gen6) * 2.0
    sense_dict = {int(ids6[0, 3]): torch.randn(16, 64, generator=gen6), int(ids6[1, 20]): torch.randn(16, 64, generator=gen6)}

    def make(cls, **attrs):
        # the constructors move a tensor to 'cuda'; build the object without running them (forward untouched)
        obj = cls.__new__(cls)
        torch.nn.Module.__init__(obj)
        for k, v in attrs.items():
            setattr(obj, k, v)
        return obj

    common = dict(backpack_network=model6, content_weights=content_weights, target_weight=torch.zeros(96),
                  annealing_scale=30.0, upweight_nearby=True)   # sims are 0.04..0.4 here: spans the sigmoid
    ocfg6 = dict(n_embd=64, n_head=2, n_layer=2, num_content_vectors=16, layer_norm_epsilon=cfg6.layer_norm_epsilon,
                 scale_attn_by_inverse_layer_idx=True)
    g6 = {('sd/' + k): np32(v) for k, v in sd6.items()}
    g6.update(ids=ids6.numpy(), content_weights=np32(content_weights),
              sense_words=np.array(sorted(sense_dict)), layer_norm_epsilon=np.float64(cfg6.layer_norm_epsilon),
              annealing_scale=np.float64(30.0))
    for w in sorted(sense_dict):
        g6['sense/%d' % w] = np32(sense_dict[w])
    with torch.no_grad():
        for tag, anneal in (('anneal', True), ('plain', False)):
            want = make(im.WeightedBackpackLMHeadModel, anneal=anneal, **common)(ids6).logits
            got = R.weighted_backpack_logits(sd6, ocfg6, ids6, content_weights, annealing_scale=30.0, anneal=anneal)
            check('G6 weighted ' + tag, got, want, 5e-5)
            g6['weighted_' + tag] = np32(want)
            want = make(im.NegativeWeightedBackpackLMHeadModel, anneal=anneal, **common)(ids6).logits
            got = R.negative_weighted_backpack_logits(sd6, ocfg6, ids6, content_weights, annealing_scale=30.0,
                                                      anneal=anneal)
            check('G6 negative ' + tag, got, want, 5e-5)
            g6['negative_' + tag] = np32(want)
        content6 = model6.transformer.content_model(ids6)
        scores = im.mask_annealing(model6, ids6, torch.zeros(96), content6, 30.0, True)
        check('G6 annealing scores', R.mask_annealing_scores(sd6['lm_head.weight'], ids6, content6, 30.0), scores, 1e-6)
        g6['annealing_scores'] = np32(scores)
        want = make(im.ReplacedWordLMHeadModel, backpack_network=model6, sense_dict=sense_dict)(ids6).logits
        got = R.replaced_word_logits(sd6, ocfg6, ids6, sense_dict)
        check('G6 replaced words', got, want, 5e-5)
        g6['replaced'] = np32(want)
    np.savez_compressed(os.path.join(HERE, 'g6_interventions.npz'), **g6)

    with open(os.path.join(HERE, 'PINNING.txt'), 'w') as f:
        f.write('oracle/ref_cpu.py checked against the imported reference (torch %s)\n' % torch.__version__)
        f.write('\n'.join(log) + '\n')
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
