#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/settle_bench.log; : > $L
timeout 300 python scripts/bench_kernels.py --which flash,lse,bwd >> $L 2>&1
BP_FLASH_IMPL=dma2 timeout 300 python scripts/bench_kernels.py --which flash >> $L 2>&1
timeout 300 python scripts/bench_kernels.py --which flash --seq 4096 --batch 8 --noncausal >> $L 2>&1
timeout 300 python scripts/bench_kernels.py --which flash,bwd --seq 2048 --batch 16 >> $L 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward.py -x -q -k "flash" 2>&1 | tail -4 >> $L
cat $L
