#!/bin/bash
# round 6, run zb: bench lines of every workload on the final tree (consecutive groups per XCD in xcd_map)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
bash scripts/gpu_run.sh bench r06_zb_default --steps 20 --warmup 5
for w in small-4096-fp16 mini-k64-1024 mini-k4-1024 mini-k1-1024; do
  bash scripts/gpu_run.sh bench r06_zb_$w --workload $w --steps 10 --warmup 3 --no-cpu-baseline
done
