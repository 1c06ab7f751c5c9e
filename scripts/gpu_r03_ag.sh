cd "$GRAFT_REPO_ROOT"
bash scripts/gpu_pmc.sh mix_b64_new --which mix --batch 64 --iters 5
cat gpurun_out/pmc_mix_b64_new/summary.txt | head -60
export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_mixold.so
bash scripts/gpu_pmc.sh mix_b64_old --which mix --batch 64 --iters 5
cat gpurun_out/pmc_mix_b64_old/summary.txt | head -60
