"""MI355X-native drop-in for the `flash_attn` Python package of john-hewitt/backpacks-flash-attn:
same module paths and call signatures for the Backpack forward path, HIP kernels underneath
(bp_hip -> libbackpack_hip.so)."""
__version__ = '0.2.6.post1+gfx950'
