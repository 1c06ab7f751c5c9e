#!/bin/bash
# round 4, run ab: job order of the gathering sense mix: column-chunk-major (default build) against sample-major (smajor)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_ab
export TMPDIR=/tmp
timeout 1500 python scripts/ab_kernels.py --libs default,smajor --which mixgather --batch 64,512,2048 --reps 3 --iters 5 --out gpurun_out/r04_ab/ab.jsonl > gpurun_out/r04_ab/ab.log 2>&1
tail -8 gpurun_out/r04_ab/ab.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gather" 2>&1 | tail -2
