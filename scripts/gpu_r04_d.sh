#!/bin/bash
# round 4, run d: 8-wave (256-query) flash-forward workgroups vs the shipped 4-wave ones; non-temporal LayerNorm variants by batch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_d
export TMPDIR=/tmp
L=$PWD/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip_nw8.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py -m gpu -x -q -k "flash or retry or lse or alpha" > gpurun_out/r04_d/pytest_nw8.log 2>&1
echo "nw8: $(tail -1 gpurun_out/r04_d/pytest_nw8.log)"
timeout 900 python scripts/ab_kernels.py --libs default,nw8 --which flash,lse --batch 64,256 --reps 3 --out gpurun_out/r04_d/ab_flash_nw8.jsonl > gpurun_out/r04_d/ab_flash_nw8.log 2>&1
tail -9 gpurun_out/r04_d/ab_flash_nw8.log
timeout 900 python scripts/ab_kernels.py --libs default,nw8 --which flash --batch 16 --seq 4096 --reps 2 > gpurun_out/r04_d/ab_flash_nw8_s4096.log 2>&1
tail -3 gpurun_out/r04_d/ab_flash_nw8_s4096.log
timeout 900 python scripts/ab_kernels.py --libs default,lnnt1,lnnt2,lnnt3 --which ln --batch 64,512,1536 --reps 3 --out gpurun_out/r04_d/ab_ln_nt.jsonl > gpurun_out/r04_d/ab_ln_nt.log 2>&1
tail -13 gpurun_out/r04_d/ab_ln_nt.log
