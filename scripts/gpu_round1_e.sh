mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu --timeout 600 2>&1 | tail -5 > gpurun_out/t6.log
echo "== default" > gpurun_out/kb5.log; python scripts/bench_kernels.py --which flash,lse,mix >> gpurun_out/kb5.log 2>&1
python scripts/bench_kernels.py --which flash,lse --batch 256 --iters 5 >> gpurun_out/kb5.log 2>&1
