#!/bin/bash
# mix kernel with super-tile loop order: parity, A/B of block orders, HBM traffic
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "mix or alpha" 2>&1 | tail -8 > gpurun_out/mix2_tests.log
for o in default heavy grouped; do
  echo "order=$o" >> gpurun_out/mix2_bench.log
  BP_MIX_ORDER=$o timeout 300 python scripts/bench_kernels.py --which mix >> gpurun_out/mix2_bench.log 2>&1
done
echo "mini-ish k=64 d=384 (b16)" >> gpurun_out/mix2_bench.log
timeout 300 python scripts/bench_kernels.py --which mix --senses 64 --d 384 --batch 16 >> gpurun_out/mix2_bench.log 2>&1
echo "s=2048 b=16" >> gpurun_out/mix2_bench.log
timeout 300 python scripts/bench_kernels.py --which mix --seq 2048 --batch 16 >> gpurun_out/mix2_bench.log 2>&1
bash scripts/gpu_pmc.sh mix2 --which mix --iters 5
cat gpurun_out/mix2_tests.log gpurun_out/mix2_bench.log gpurun_out/pmc_mix2/summary.txt
