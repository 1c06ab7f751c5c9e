#!/bin/bash
# round 4, run q: two 32-query blocks per wave in the flash forward (BP_FWD_QB=2: 256-query workgroup tiles)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_q
export TMPDIR=/tmp
BP_FWD_QB=2 timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py tests/test_gpu_stress.py -m gpu -q -k "flash or retry or lse or attn" > gpurun_out/r04_q/parity_qb2.log 2>&1
tail -3 gpurun_out/r04_q/parity_qb2.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "flash" > gpurun_out/r04_q/parity_qb1.log 2>&1
tail -2 gpurun_out/r04_q/parity_qb1.log
timeout 1200 python scripts/ab_kernels.py --libs default,default+BP_FWD_QB=2 --which flash --batch 16,64,256 --reps 3 --out gpurun_out/r04_q/ab.jsonl > gpurun_out/r04_q/ab.log 2>&1
tail -8 gpurun_out/r04_q/ab.log
timeout 600 python scripts/ab_kernels.py --libs default,default+BP_FWD_QB=2 --which flash --batch 16 --seq 4096 --reps 2 > gpurun_out/r04_q/ab_4k.log 2>&1
tail -3 gpurun_out/r04_q/ab_4k.log
timeout 600 python scripts/ab_kernels.py --libs default,default+BP_FWD_QB=2 --which flash --batch 16 --seq 4096 --extra=--noncausal --reps 2 > gpurun_out/r04_q/ab_nc4k.log 2>&1
tail -3 gpurun_out/r04_q/ab_nc4k.log
