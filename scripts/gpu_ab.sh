#!/bin/bash
# usage: bash scripts/gpu_ab.sh <variant> [bench_kernels args]   -- A/B of libbackpack_hip.so vs a variant build
cd "$GRAFT_REPO_ROOT"
v=$1; shift
mkdir -p gpurun_out
L=gpurun_out/ab_$v.log; : > $L
for lib in default $v; do
  echo "== $lib" >> $L
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$v.so; fi
  timeout 300 python scripts/bench_kernels.py "$@" 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
