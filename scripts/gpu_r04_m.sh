#!/bin/bash
# round 4, run m: raised wave priority while a wave issues its MFMAs (S^T and / or P V), against the shipped kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_m
export TMPDIR=/tmp
timeout 1200 python scripts/ab_kernels.py --libs default,prio7,prio15,prio6,prio5 --which flash,lse --batch 64,256 --reps 3 --out gpurun_out/r04_m/ab_prio.jsonl > gpurun_out/r04_m/ab.log 2>&1
tail -21 gpurun_out/r04_m/ab.log
timeout 900 python scripts/ab_kernels.py --libs default,prio7,prio15,prio6,prio5 --which flash --batch 16 --seq 4096 --extra=--noncausal --reps 2 > gpurun_out/r04_m/ab_nc.log 2>&1
tail -6 gpurun_out/r04_m/ab_nc.log
