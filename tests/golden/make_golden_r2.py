"""Round-2 golden vectors (same rules as make_golden.py: runs only in the build container, imports the REAL
reference from /root/reference, writes data-only .npz files, asserts this repo's mirror reproduces them).

    python tests/golden/make_golden_r2.py

  G7  g7_hf_remap.npz   a tiny random Hugging Face GPT-2 state dict pushed through the reference's
                        `remap_state_dict_gpt2` (training/src/models/backpack.py:354-409) and back through
                        `remap_state_dict_flash` (training/demo_convert.py:22-85), plus the logits of
                        transformers' own GPT2LMHeadModel on those weights (pins the LayerNorm half-block shift
                        end to end: the flash-layout model must reproduce them).
  G8  g8_generation.npz the reference's `greedy_decode` (training/src/utils/generation.py:50-75) on the nano
                        model of G4: the generated token ids and the RETURNED LENGTH (max_length - 1).
"""
import ast
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from make_golden import import_reference, np32  # noqa: E402


def reference_function(path, name, namespace):
    """Execute ONE function definition of a reference file that cannot be imported as a module here
    (demo_convert.py imports the Lightning task at module level).  Nothing of it is stored."""
    tree = ast.parse(open(path).read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    code = compile(ast.Module(body=[node], type_ignores=[]), path, 'exec')
    exec(code, namespace)
    return namespace[name]


def tiny_hf_state_dict(n_layer, d, vocab, n_pos, seed):
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale=0.2):
        return torch.randn(*shape, generator=g) * scale

    sd = OrderedDict()
    sd['wte.weight'] = rnd(vocab, d)
    sd['wpe.weight'] = rnd(n_pos, d)
    for i in range(n_layer):
        p = 'h.%d.' % i
        sd[p + 'ln_1.weight'] = 1 + rnd(d)
        sd[p + 'ln_1.bias'] = rnd(d)
        sd[p + 'attn.bias'] = torch.tril(torch.ones(n_pos, n_pos)).view(1, 1, n_pos, n_pos)
        sd[p + 'attn.c_attn.weight'] = rnd(d, 3 * d)
        sd[p + 'attn.c_attn.bias'] = rnd(3 * d)
        sd[p + 'attn.c_proj.weight'] = rnd(d, d)
        sd[p + 'attn.c_proj.bias'] = rnd(d)
        sd[p + 'ln_2.weight'] = 1 + rnd(d)
        sd[p + 'ln_2.bias'] = rnd(d)
        sd[p + 'mlp.c_fc.weight'] = rnd(d, 4 * d)
        sd[p + 'mlp.c_fc.bias'] = rnd(4 * d)
        sd[p + 'mlp.c_proj.weight'] = rnd(4 * d, d)
        sd[p + 'mlp.c_proj.bias'] = rnd(d)
    sd['ln_f.weight'] = 1 + rnd(d)
    sd['ln_f.bias'] = rnd(d)
    return sd


def main():
    bp, mha = import_reference()
    import re
    import torch.nn.functional as F
    import transformers
    # this repo's mirror, loaded by path: the package name `src` is the reference's here
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'bp_hf_convert', os.path.join(ROOT, 'backpacks-flash-attn_amd', 'flash_attn', 'utils', 'hf_convert.py'))
    hf_convert = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hf_convert)

    # ---- G7 ------------------------------------------------------------------------------------
    n_layer, d, vocab, n_pos = 3, 16, 45, 12
    cfg = transformers.GPT2Config(n_embd=d, n_head=2, n_layer=n_layer, vocab_size=vocab, n_positions=n_pos,
                                  resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    cfg.pad_vocab_size_multiple = 8
    hf_sd = tiny_hf_state_dict(n_layer, d, vocab, n_pos, seed=7)
    padded_cfg = transformers.GPT2Config(**{**cfg.to_dict(), 'vocab_size': 48})
    ref_flash = bp.remap_state_dict_gpt2(OrderedDict(hf_sd), padded_cfg)
    remap_flash = reference_function(os.path.join(REF, 'training', 'demo_convert.py'), 'remap_state_dict_flash',
                                     {'re': re, 'OrderedDict': OrderedDict, 'F': F, 'torch': torch})
    ref_back = remap_flash(OrderedDict(ref_flash), padded_cfg)

    mine_flash = hf_convert.remap_state_dict_gpt2(hf_sd, padded_cfg)
    assert set(mine_flash) == set(ref_flash), set(mine_flash) ^ set(ref_flash)
    for k in ref_flash:
        assert torch.equal(mine_flash[k], ref_flash[k]), k
    mine_back = hf_convert.remap_state_dict_flash(mine_flash, padded_cfg)
    assert set(mine_back) == set(ref_back), set(mine_back) ^ set(ref_back)
    for k in ref_back:
        assert torch.equal(mine_back[k], ref_back[k]), k

    # transformers' own GPT-2 on the HF weights: the semantic anchor of the LayerNorm shift
    hf_model = transformers.GPT2LMHeadModel(cfg).eval()
    full = {('transformer.' + k): v for k, v in hf_sd.items() if not k.endswith('.attn.bias')}
    full['lm_head.weight'] = hf_sd['wte.weight']
    missing = hf_model.load_state_dict(full, strict=False)
    assert not [k for k in missing.missing_keys if not k.endswith(('.attn.bias', '.attn.masked_bias'))], missing
    assert not missing.unexpected_keys, missing
    ids = torch.randint(0, vocab, (2, n_pos), generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        hf_logits = hf_model(ids).logits
    # the reference's flash GPTLMHeadModel (eager flags) on the remapped weights must agree with HF
    import flash_attn.models.gpt as ref_gpt
    ref_cfg = transformers.GPT2Config(**{**cfg.to_dict()})
    ref_cfg.pad_vocab_size_multiple = 8
    ref_model = ref_gpt.GPTLMHeadModel(ref_cfg).eval()
    ref_model.load_state_dict(ref_flash)
    with torch.no_grad():
        ref_logits = ref_model(ids).logits
    err = (ref_logits[..., :vocab] - hf_logits).abs().max().item()
    print('G7: reference flash model vs transformers GPT2LMHeadModel on remapped weights: %.2e' % err)
    assert err < 1e-4

    out = {('hf/' + k): np32(v) for k, v in hf_sd.items()}
    out.update({('flash/' + k): np32(v) for k, v in ref_flash.items()})
    out.update({('back/' + k): np32(v) for k, v in ref_back.items()})
    out.update(ids=ids.numpy(), hf_logits=np32(hf_logits), n_layer=n_layer, n_embd=d, n_head=2, vocab_size=vocab,
               padded_vocab_size=48, n_positions=n_pos)
    np.savez_compressed(os.path.join(HERE, 'g7_hf_remap.npz'), **out)

    # ---- G8 ------------------------------------------------------------------------------------
    import src.utils.generation as ref_gen     # resolves to the REFERENCE (sys.path order of import_reference)
    assert ref_gen.__file__.startswith(REF), ref_gen.__file__
    g4 = np.load(os.path.join(HERE, 'g4_nano_model.npz'))
    sd = {k[3:]: torch.from_numpy(g4[k]) for k in g4.files if k.startswith('sd/')}
    nano = dict(n_embd=64, n_head=2, n_layer=2, num_content_vectors=4, vocab_size=96, n_positions=32,
                scale_attn_by_inverse_layer_idx=True, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                use_flash_attn=False, fused_bias_fc=False, fused_dense_gelu_dense=False,
                fused_dropout_add_ln=False, pad_vocab_size_multiple=8)
    model = bp.BackpackLMHeadModel(bp.BackpackConfig(**nano)).eval()
    model.load_state_dict(sd)
    prompt = torch.from_numpy(g4['ids'])[:1, :5].clone()
    g8 = dict(prompt=prompt.numpy())
    for max_length in (6, 12, 20):
        res = ref_gen.greedy_decode(prompt, model, max_length)
        g8['greedy_%d' % max_length] = res.sequences.numpy()
        g8['scores_%d' % max_length] = np32(res.scores[0])
        assert len(res.scores) == 1
        print('G8: greedy_decode(max_length=%d) returned shape %s' % (max_length, tuple(res.sequences.shape)))
    torch.manual_seed(5)
    g8['sample_12_shape'] = np.array(ref_gen.sample(prompt, model, 12).sequences.shape)
    np.savez_compressed(os.path.join(HERE, 'g8_generation.npz'), **g8)
    print('round-2 golden vectors written to', HERE)


if __name__ == '__main__':
    main()
