# round-2 checkpoint u: sense mix: committed kernel (_head) vs working tree (XCD queue by sample + asynchronous per-sense Q prefetch)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward.py tests/test_gpu_configs.py -q -m gpu --timeout 600 -k "mix" 2>&1 | grep -E "passed|failed" > gpurun_out/t_r02_u.log
( for rep in 1 2 3; do for v in _head ""; do for b in 4 16 64 128; do
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which mix --batch $b --iters 30 | sed "s/sense_mix/sense_mix$v/"
done; done; done ) > gpurun_out/r02_u_mix_xcd.log 2>&1
cat gpurun_out/t_r02_u.log; grep -v amdgpu.ids gpurun_out/r02_u_mix_xcd.log
