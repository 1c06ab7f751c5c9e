#!/bin/bash
# round 4, run i: fewer, longer flash-forward workgroups (2 / 4 tile pairs per workgroup): is there a per-workgroup cost?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_i
export TMPDIR=/tmp
L=$PWD/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip_ppw4.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py -m gpu -x -q -k "flash or retry or lse" > gpurun_out/r04_i/pytest_ppw4.log 2>&1; echo "ppw4: $(tail -1 gpurun_out/r04_i/pytest_ppw4.log)"
timeout 1200 python scripts/ab_kernels.py --libs default,ppw2,ppw4 --which flash,lse --batch 64,256,1024 --reps 3 --out gpurun_out/r04_i/ab_flash_pairs_per_wg.jsonl > gpurun_out/r04_i/ab.log 2>&1
tail -19 gpurun_out/r04_i/ab.log
timeout 600 python scripts/ab_kernels.py --libs default,ppw2,ppw4 --which flash --batch 256 --extra=--noncausal --reps 2 > gpurun_out/r04_i/ab_nc.log 2>&1
tail -4 gpurun_out/r04_i/ab_nc.log
