# round-2 checkpoint n: flash forward per-workgroup timeline (development build)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_prof.so
( timeout 300 python scripts/probes/flash_timeline/timeline.py --batch 64
  timeout 300 python scripts/probes/flash_timeline/timeline.py --batch 256
  timeout 300 python scripts/probes/flash_timeline/timeline.py --batch 64 --noncausal
  timeout 300 python scripts/probes/flash_timeline/timeline.py --batch 16 --seq 4096 ) > gpurun_out/r02_n_timeline.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_n_timeline.log
