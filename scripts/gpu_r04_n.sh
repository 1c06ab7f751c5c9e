#!/bin/bash
# round 4, run n: lean edge tiles of the flash forward (BP_FWD_LEAN: 1 first tile, 2 diagonal tile, 4 dead half skipped)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_n
export TMPDIR=/tmp
LIBDIR=$PWD/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$LIBDIR/libbackpack_hip_lean.so timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py tests/test_gpu_stress.py tests/test_gpu_dropout.py -m gpu -x -q -k "flash or retry or lse or attn" > gpurun_out/r04_n/parity_lean.log 2>&1
tail -3 gpurun_out/r04_n/parity_lean.log
timeout 1200 python scripts/ab_kernels.py --libs default,lean3,lean --which flash,lse --batch 64,256 --reps 3 --out gpurun_out/r04_n/ab_lean.jsonl > gpurun_out/r04_n/ab.log 2>&1
tail -16 gpurun_out/r04_n/ab.log
timeout 600 python scripts/ab_kernels.py --libs default,lean3,lean --which flash --batch 16 --seq 4096 --reps 2 > gpurun_out/r04_n/ab_4k.log 2>&1
tail -5 gpurun_out/r04_n/ab_4k.log
