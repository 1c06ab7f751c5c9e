mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "sense or mix" --timeout 600 2>&1 | tail -5 > gpurun_out/t5.log
echo "== heavy-first" > gpurun_out/kb4.log; python scripts/bench_kernels.py --which mix >> gpurun_out/kb4.log 2>&1
echo "== grouped" >> gpurun_out/kb4.log; BP_MIX_ORDER=grouped python scripts/bench_kernels.py --which mix >> gpurun_out/kb4.log 2>&1
echo "== b8 heavy-first" >> gpurun_out/kb4.log; python scripts/bench_kernels.py --which mix --batch 8 >> gpurun_out/kb4.log 2>&1
echo "== b16" >> gpurun_out/kb4.log; python scripts/bench_kernels.py --which mix --batch 16 >> gpurun_out/kb4.log 2>&1
echo "== b256" >> gpurun_out/kb4.log; python scripts/bench_kernels.py --which mix --batch 256 --iters 5 >> gpurun_out/kb4.log 2>&1
