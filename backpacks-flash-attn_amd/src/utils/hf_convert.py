"""Kept for the reference's import path (training/src/...): the converter itself lives beside the model it serves,
`flash_attn/utils/hf_convert.py` (the reference keeps `remap_state_dict_gpt2` next to the model,
flash_attn/models/gpt.py / training/src/models/backpack.py:354-409), so `flash_attn` does not depend on `src`."""
from flash_attn.utils.hf_convert import (  # noqa: F401
    gpt2_trunk_state_dict, load_non_optimized_model, remap_state_dict_flash, remap_state_dict_gpt2)
