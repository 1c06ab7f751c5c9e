#!/bin/bash
# round 6, run r: a hunt on the wide ring kernels only (drawn lengths 32 ... 768, 1-4 senses, any output width, dense and table
# form, backward through the alpha-rebuilding route) and the regular drawn sweep with the new draw
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
BP_FUZZ_RING=1 BP_FUZZ_SEEDS=250 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "sense_kernels_on_drawn_shapes" 2>&1 | tail -n 6 | grep -v "^$" | tee gpurun_out/r06_r_ring_hunt.txt
BP_FUZZ_SEEDS=120 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "sense_kernels_on_drawn_shapes" 2>&1 | tail -n 6 | grep -v "^$" | tee gpurun_out/r06_r_sense_hunt.txt
