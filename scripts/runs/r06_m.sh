#!/bin/bash
# round 6, run m: bench lines of the two few-sense workloads on the ring kernels (sense_wide_dma.hip), with the kernel trace
# of the same command
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for w in mini-k4-1024 mini-k1-1024; do
  bash scripts/gpu_run.sh bench r06_m_$w --workload $w --steps 10 --warmup 3 --no-cpu-baseline
done
bash scripts/gpu_run.sh stats r06_m_mini_k4_b1024 --workload mini-k4-1024 --steps 5 --warmup 2 --batch 1024
bash scripts/gpu_run.sh stats r06_m_mini_k1_b1024 --workload mini-k1-1024 --steps 5 --warmup 2 --batch 1024
