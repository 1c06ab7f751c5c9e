mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -8 > gpurun_out/t9.log
timeout 600 python bench.py > gpurun_out/bench_r01_c.log 2>&1
