#!/bin/bash
# round 4, run c: woven flash-forward tile (3 and 4 waves per SIMD) against the shipped kernel: parity, then same-box A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_c
export TMPDIR=/tmp
L=$PWD/backpacks-flash-attn_amd/bp_hip
for v in weave3 weave4; do
  BP_HIP_LIB=$L/libbackpack_hip_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py -m gpu -x -q -k "flash or retry" > gpurun_out/r04_c/pytest_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r04_c/pytest_$v.log)"
done
timeout 900 python scripts/ab_kernels.py --libs default,weave3,weave4 --which flash --batch 64,256 --reps 3 --out gpurun_out/r04_c/ab_flash_s1024.jsonl > gpurun_out/r04_c/ab_flash_s1024.log 2>&1
tail -7 gpurun_out/r04_c/ab_flash_s1024.log
timeout 900 python scripts/ab_kernels.py --libs default,weave3,weave4 --which flash --batch 16 --seq 4096 --extra=--noncausal --reps 3 --out gpurun_out/r04_c/ab_flash_s4096nc.jsonl > gpurun_out/r04_c/ab_flash_s4096nc.log 2>&1
tail -4 gpurun_out/r04_c/ab_flash_s4096nc.log
