# round-2 checkpoint k: flash forward A/B on one box: committed kernel (_head) vs working tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropout.py tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_backward.py -q -m gpu --timeout 900  2>&1 | tail -5 > gpurun_out/t_r02_k.log
for rep in 1 2 3; do
for v in _head ""; do
  for cfg in "1024 64" "1024 256" "512 128" "4096 16"; do
    set -- $cfg
    BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash --seq $1 --batch $2 --iters 30 | sed "s/flash_fwd/flash_fwd$v/"
  done
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash --seq 4096 --batch 16 --iters 30 --noncausal | sed "s/flash_fwd/flash_fwd$v noncausal/"
done
done > gpurun_out/r02_k_flash_ab.log 2>&1
cat gpurun_out/t_r02_k.log; grep -v amdgpu.ids gpurun_out/r02_k_flash_ab.log
