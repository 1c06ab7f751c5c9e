#!/bin/bash
# usage: bash scripts/gpu_abl.sh <bench_kernels args>  -- times every bp_hip/libbackpack_hip_abl_*.so
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/abl.log; : > $L
for lib in default $(ls backpacks-flash-attn_amd/bp_hip/ | grep "libbackpack_hip_abl_" ); do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/$lib; fi
  printf "%-40s " $lib >> $L
  timeout 300 python scripts/bench_kernels.py "$@" 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; [print(json.loads(l)['kernel'], json.loads(l)['ms'], end='   ') for l in sys.stdin if l.startswith('{')]; print()" >> $L
done
cat $L
