#!/bin/bash
# round 4, run s: flash forward with and without the cu_seqlens reads (fixed-length entry)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_s
export TMPDIR=/tmp
timeout 900 python scripts/ab_kernels.py --libs default,default+BP_BENCH_FIXED_LEN=1 --which flash --batch 64,256 --reps 4 --out gpurun_out/r04_s/ab.jsonl > gpurun_out/r04_s/ab.log 2>&1
tail -6 gpurun_out/r04_s/ab.log
