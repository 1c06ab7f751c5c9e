cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in oldstore oldpro default; do
  if [ $v = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$v.so; fi
  echo "== $v"; timeout 600 python scripts/debug/r03_bwd_case.py > gpurun_out/r03_q_$v.log 2>&1; grep -v amdgpu gpurun_out/r03_q_$v.log | grep -v "^  File\|Extension" | cut -c1-220 | head -24
done
