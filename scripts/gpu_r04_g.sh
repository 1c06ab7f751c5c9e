#!/bin/bash
# round 4, run g: co-resident flash-forward workgroups de-synchronised (pair index rotated by residency round / light tile first)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_g
export TMPDIR=/tmp
L=$PWD/backpacks-flash-attn_amd/bp_hip
for v in stg1 stg3; do
  BP_HIP_LIB=$L/libbackpack_hip_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py -m gpu -x -q -k "flash or retry or lse" > gpurun_out/r04_g/pytest_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/r04_g/pytest_$v.log)"
done
timeout 1200 python scripts/ab_kernels.py --libs default,stg1,stg2,stg3 --which flash,lse --batch 64,256,1024 --reps 3 --out gpurun_out/r04_g/ab_flash_stagger.jsonl > gpurun_out/r04_g/ab_flash_stagger.log 2>&1
tail -26 gpurun_out/r04_g/ab_flash_stagger.log
