#!/bin/bash
# round 4, run p: what-if timing probes of the flash forward (BP_FWD_WHATIF; results are garbage on purpose)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_p
export TMPDIR=/tmp
L=default,wi1,wi32,wi16,wi64,wi113,wi2,wi4,wi6,wi8,wi9,wi119,wi127
timeout 1500 python scripts/ab_kernels.py --libs $L --which flash --batch 16 --seq 4096 --extra=--noncausal --reps 2 --out gpurun_out/r04_p/whatif_nc4k.jsonl > gpurun_out/r04_p/nc4k.log 2>&1
tail -14 gpurun_out/r04_p/nc4k.log
timeout 1500 python scripts/ab_kernels.py --libs $L --which flash --batch 256 --reps 2 --out gpurun_out/r04_p/whatif_c1k.jsonl > gpurun_out/r04_p/c1k.log 2>&1
tail -14 gpurun_out/r04_p/c1k.log
