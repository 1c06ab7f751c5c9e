#!/bin/bash
# backward kernels: parity tests + microbench + the not-yet-run forward tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_backward.py -x -q 2>&1 | tail -25 > gpurun_out/bwd_tests.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "intervention or seq4096" 2>&1 | tail -8 > gpurun_out/new_fwd_tests.log
timeout 300 python scripts/bench_kernels.py --which bwd,flash > gpurun_out/bwd_bench.log 2>&1
timeout 300 python scripts/bench_kernels.py --which bwd --seq 2048 --batch 16 >> gpurun_out/bwd_bench.log 2>&1
cat gpurun_out/bwd_tests.log gpurun_out/new_fwd_tests.log gpurun_out/bwd_bench.log
