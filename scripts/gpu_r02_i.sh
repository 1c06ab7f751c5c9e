# round-2 checkpoint i: flash forward per-pass fixed cost (Q of both query tiles requested up front, first DMA before the Q wait, cheaper first tile)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2000 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropout.py tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_backward.py -q -m gpu --timeout 900 2>&1 | tail -40 > gpurun_out/t_r02_i.log
for rep in 1 2; do timeout 300 python scripts/bench_kernels.py --which flash,lse --batch 64 --iters 30; done > gpurun_out/r02_i_flash.log 2>&1
timeout 300 python scripts/bench_kernels.py --which flash --batch 256 --iters 20 >> gpurun_out/r02_i_flash.log 2>&1
timeout 300 python scripts/bench_kernels.py --which flash --batch 16 --seq 4096 --iters 20 --noncausal >> gpurun_out/r02_i_flash.log 2>&1
timeout 300 python scripts/bench_kernels.py --which flash --batch 16 --seq 4096 --iters 20 >> gpurun_out/r02_i_flash.log 2>&1
timeout 300 python scripts/bench_kernels.py --which flash --batch 8 --seq 4096 --iters 20 --dtype fp16 >> gpurun_out/r02_i_flash.log 2>&1
cat gpurun_out/t_r02_i.log; grep -v amdgpu.ids gpurun_out/r02_i_flash.log
