mkdir -p gpurun_out; rm -f gpurun_out/kb10.log
timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -4 > gpurun_out/t11.log
python scripts/bench_kernels.py --which alpha >> gpurun_out/kb10.log 2>&1
python scripts/bench_kernels.py --which alpha --seq 4096 --batch 2 >> gpurun_out/kb10.log 2>&1
