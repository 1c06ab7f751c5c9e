#!/bin/bash
# round 6, run v: long hunt on the final tree -- every drawn-shape sweep with 1500 seeds (the sense sweep draws the ring
# kernels' shapes too), then the ring-only hunt with 600 more, then the property tests
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
BP_FUZZ_SEEDS=1500 BP_FUZZ_MODELS=150 timeout 3300 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v "^$\|amdgpu.ids" | tail -n 4 | tee gpurun_out/r06_v_long_hunt.txt
BP_FUZZ_RING=1 BP_FUZZ_SEEDS=600 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "sense_kernels_on_drawn_shapes" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -n 3 | tee gpurun_out/r06_v_ring_hunt.txt
timeout 1200 python -m pytest tests/test_gpu_properties.py -m gpu -x -q 2>&1 | grep -v "^$\|amdgpu.ids" | tail -n 3 | tee gpurun_out/r06_v_properties.txt
