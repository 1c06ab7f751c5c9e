#!/bin/bash
# round 6, run k: (1) the LSE ring's 1-ulp run-to-run variation: four slots with every wait draining the ring (lse4all) and
# three slots with the counted wait (lse3) against the shipped two slots; (2) counters of the d_k = 160 mix (8 waves x 320 columns)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
L=$PWD/backpacks-flash-attn_amd/bp_hip
for v in "" _lse4all _lse3; do echo "== lib$v"; BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/debug/wide_lse_probe.py 2>&1 | grep -v amdgpu.ids | head -3; done | tee gpurun_out/r06_k_probe.txt
bash scripts/gpu_run.sh pmc r06_k_mix160 --which mix --batch 256 --senses 4 --d 640 --iters 5
bash scripts/gpu_run.sh pmc r06_k_mix640 --which mix --batch 256 --senses 1 --d 640 --iters 5
