# round-2 checkpoint w: sense mix job order inside an XCD queue: heaviest tiles of all samples first (shipped) vs sample-major
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip_sm.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout 600 -k "mix" 2>&1 | grep -E "passed|failed" > gpurun_out/t_r02_w.log
( for rep in 1 2 3; do for v in "" _sm; do for b in 16 64 128 512; do
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which mix --batch $b --iters 20 | sed "s/sense_mix/sense_mix$v/"
done; done; done ) > gpurun_out/r02_w_mix_order.log 2>&1
export BP_HIP_LIB=$L/libbackpack_hip_sm.so
bash scripts/gpu_pmc.sh r02_w_sm_b64 --which mix --batch 64 --iters 5
cp gpurun_out/pmc_r02_w_sm_b64/summary.txt gpurun_out/r02_w_pmc_sample_major_b64.txt; rm -rf gpurun_out/pmc_r02_w_sm_b64
cat gpurun_out/t_r02_w.log; grep -v amdgpu.ids gpurun_out/r02_w_mix_order.log; grep -E "FETCH_SIZE|TCC_HIT|TCC_MISS" gpurun_out/r02_w_pmc_sample_major_b64.txt
