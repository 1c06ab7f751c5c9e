# round-2 checkpoint d: persistent + software-pipelined sense mix; dropout test failures with tracebacks
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dropout.py -q -m gpu --timeout 600 -x -k "documented_function" 2>&1 | tail -60 > gpurun_out/t_r02_d_mask.log
timeout 600 python -m pytest tests/test_gpu_dropout.py -q -m gpu --timeout 600 -k "autocast" 2>&1 | tail -30 > gpurun_out/t_r02_d_amp.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_backward.py tests/test_gpu_configs.py -q -m gpu --timeout 900 -k "mix or model or config or interven or sense or backpack" 2>&1 | tail -30 > gpurun_out/t_r02_d_mix.log
for b in 4 16 64 128; do timeout 300 python scripts/bench_kernels.py --which mix --batch $b --iters 30; done > gpurun_out/r02_d_mix.log 2>&1
cat gpurun_out/t_r02_d_mask.log | tail -45; tail -5 gpurun_out/t_r02_d_amp.log; tail -12 gpurun_out/t_r02_d_mix.log; grep -v amdgpu.ids gpurun_out/r02_d_mix.log
