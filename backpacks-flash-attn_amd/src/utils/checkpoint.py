"""Checkpoint helpers -- mirror of the parts of the reference's training/src/utils/checkpoint.py the
Backpack release uses (README.md:108-126): load a Lightning `.ckpt` and strip the `model.` prefix
(:8-29, :68-76) so `BackpackLMHeadModel.load_state_dict` accepts it key for key."""
from pathlib import Path

import torch


def load_checkpoint(path, device='cpu'):
    path = Path(path).expanduser()
    if path.is_dir():
        raise NotImplementedError('DeepSpeed checkpoint directories are out of scope')
    return torch.load(path, map_location=device, weights_only=False)


def remove_model_prefix(state_dict):
    """Lightning stores the network under `state_dict['state_dict']` with keys `model.<name>`."""
    inner = state_dict['state_dict'] if 'state_dict' in state_dict else state_dict
    return {(k[len('model.'):] if k.startswith('model.') else k): v for k, v in inner.items()}


def load_backpack_checkpoint(model, path, device='cpu', strict=True):
    """`model.load_state_dict(remove_model_prefix(load_checkpoint(path)))` with the tied LM head
    re-tied afterwards."""
    sd = remove_model_prefix(load_checkpoint(path, device))
    result = model.load_state_dict(sd, strict=strict)
    if hasattr(model, 'tie_weights'):
        model.tie_weights()
    return result
