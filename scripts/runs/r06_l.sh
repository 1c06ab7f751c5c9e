#!/bin/bash
# round 6, run l: after the matrix-pipe drains (settle_acc): the four-slot LSE ring that varied by one ulp must repeat bit
# for bit now; kernel tests; timings of the wide kernels at the bench batch
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
L=$PWD/backpacks-flash-attn_amd/bp_hip
for v in "" _lse4; do echo "== lib$v"; BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/debug/wide_lse_probe.py 2>&1 | grep -v amdgpu.ids | head -3; done | tee gpurun_out/r06_l_probe.txt
timeout 600 python scripts/debug/wide_determinism.py --reps 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_l_determinism.txt
TAG=r06_l_kernels bash scripts/gpu_run.sh tests -x -q tests/test_gpu_kernels.py
TAG=r06_l_cfg bash scripts/gpu_run.sh tests -x -q -k "few_sense or wide"
TAG=r06_l_k4 bash scripts/gpu_run.sh ab default --which lse,mix --batch 1024 --reps 2 --extra "--senses 4 --d 640"
TAG=r06_l_k1 bash scripts/gpu_run.sh ab default --which lse,mix --batch 1024 --reps 2 --extra "--senses 1 --d 640"
