"""`state_dict_from_pretrained` -- mirror of the reference's flash_attn/utils/pretrained.py:7-8: the weights
file of a Hugging Face model as a plain state dict.  Besides a hub name (resolved through the local
transformers cache; this build never needs the network) a path to a weights file or to a directory holding
one is accepted, `.safetensors` included."""
import os

import torch


def _load_file(path):
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location='cpu', weights_only=True)


def state_dict_from_pretrained(model_name):
    if os.path.isfile(model_name):
        return _load_file(model_name)
    from transformers.utils import SAFE_WEIGHTS_NAME, WEIGHTS_NAME
    if os.path.isdir(model_name):
        for name in (SAFE_WEIGHTS_NAME, WEIGHTS_NAME):
            if os.path.isfile(os.path.join(model_name, name)):
                return _load_file(os.path.join(model_name, name))
        raise FileNotFoundError('no %s / %s under %s' % (SAFE_WEIGHTS_NAME, WEIGHTS_NAME, model_name))
    from transformers.utils.hub import cached_file
    for name in (WEIGHTS_NAME, SAFE_WEIGHTS_NAME):      # upstream looks for pytorch_model.bin only
        try:
            return _load_file(cached_file(model_name, name))
        except (OSError, ValueError):
            continue
    raise FileNotFoundError('no cached weights for %r (the hub is not reachable from this build)' % model_name)
