#!/bin/bash
# round 6, run zd: kernel traces of the other workloads' bench commands on the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
bash scripts/gpu_run.sh stats r06_zd_mini_k64_b1024 --workload mini-k64-1024 --steps 5 --warmup 2 --batch 1024 > /dev/null
bash scripts/gpu_run.sh stats r06_zd_small4096_fp16_b256 --workload small-4096-fp16 --steps 5 --warmup 2 --batch 256 > /dev/null
bash scripts/gpu_run.sh stats r06_zd_small1024_b2048 --steps 5 --warmup 2 --batch 2048 > /dev/null
head -12 gpurun_out/r06_zd_mini_k64_b1024_kernel_stats.txt | cut -c1-150
