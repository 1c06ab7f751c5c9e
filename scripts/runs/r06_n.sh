#!/bin/bash
# round 6, run n: the table form of the wide ring kernels (bp_sense_mix_gather at d_k = 160 / 640, ABI 9): parity, whole-model
# tests, bench lines of the two few-sense workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
TAG=r06_n_wide bash scripts/gpu_run.sh tests -x -q -k "wide or few_sense or gather or table"
for w in mini-k4-1024 mini-k1-1024; do
  bash scripts/gpu_run.sh bench r06_n_$w --workload $w --steps 10 --warmup 3 --no-cpu-baseline
done
