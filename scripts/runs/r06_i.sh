#!/bin/bash
# round 6, run i: repeatability of the wide-sense ring kernels on every variant; ring depth of the 8-wave x 320-column shape
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
L=$PWD/backpacks-flash-attn_amd/bp_hip
for v in "" _w160a _w160b _w160b2; do echo "== lib$v"; BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 600 python scripts/debug/wide_determinism.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_i_determinism.txt
TAG=r06_i_k4 bash scripts/gpu_run.sh ab w160b,w160b2,w160b3 --which mix --batch 1024 --reps 2 --extra "--senses 4 --d 640"
