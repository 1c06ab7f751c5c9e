#!/bin/bash
# round 4, run ac: is the sense mix bound by its content traffic?  the gathering form against tables of 50 257 rows (1.2 GB),
# 2048 rows (50 MB: memory-side cache) and 64 rows (1.5 MB: L2)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_ac
export TMPDIR=/tmp
timeout 1500 python scripts/ab_kernels.py --libs default,default+BP_BENCH_TABLE_ROWS=2048,default+BP_BENCH_TABLE_ROWS=64 --which mixgather --batch 64,512,2048 --reps 2 --iters 5 --out gpurun_out/r04_ac/ab.jsonl > gpurun_out/r04_ac/ab.log 2>&1
tail -10 gpurun_out/r04_ac/ab.log
