#!/bin/bash
# round 4, run r: per-phase clock account + workgroup lifetimes of the flash forward, full kernel and what-if floors
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_r
export TMPDIR=/tmp
L=$PWD/backpacks-flash-attn_amd/bp_hip
for v in fwdprof fwdprof6 fwdprof119; do
  for args in "--batch 128" "--batch 16 --seq 4096 --noncausal" "--batch 16 --seq 4096"; do
    echo "== $v $args" >> gpurun_out/r04_r/phases.jsonl
    BP_HIP_LIB=$L/libbackpack_hip_$v.so timeout 300 python scripts/probes/flash_fwd_phases/phases.py $args >> gpurun_out/r04_r/phases.jsonl 2>> gpurun_out/r04_r/err.log
  done
done
cat gpurun_out/r04_r/phases.jsonl
tail -5 gpurun_out/r04_r/err.log
