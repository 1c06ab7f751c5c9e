# round-2 checkpoint b: new flash forward (row-sum overflow test), in-kernel dropout everywhere, LN with fp32 x0
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dropout.py -q -m gpu --timeout 600 -x 2>&1 | tail -25 > gpurun_out/t_r02_b_dropout.log
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 --deselect tests/test_gpu_dropout.py 2>&1 | tail -25 > gpurun_out/t_r02_b.log
for b in 16 64 256; do timeout 300 python scripts/bench_kernels.py --which flash,lse --batch $b --iters 30; done > gpurun_out/r02_b_flash.log 2>&1
timeout 300 python scripts/bench_kernels.py --which flash --batch 16 --seq 4096 --iters 20 >> gpurun_out/r02_b_flash.log 2>&1
timeout 300 python scripts/bench_kernels.py --which flash --batch 16 --seq 4096 --iters 20 --noncausal >> gpurun_out/r02_b_flash.log 2>&1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_r02_b.log 2>&1
cat gpurun_out/t_r02_b_dropout.log; cat gpurun_out/t_r02_b.log; grep -v amdgpu.ids gpurun_out/r02_b_flash.log; grep "^{" gpurun_out/bench_r02_b.log | cut -c1-250
