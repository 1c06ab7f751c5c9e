# round-2 checkpoint a: full GPU suite on the round-2 host changes, mix-kernel batch sweep (HBM vs Infinity Cache), bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 -x 2>&1 | tail -25 > gpurun_out/t_r02_a.log
for b in 2 4 8 16 32 64 128; do timeout 300 python scripts/bench_kernels.py --which mix,lse --batch $b --iters 30; done > gpurun_out/r02_a_mix_sweep.log 2>&1
timeout 300 python scripts/bench_kernels.py --which flash,bwd,alpha --batch 64 --iters 30 > gpurun_out/r02_a_flash.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r02_a.log 2>&1
cat gpurun_out/t_r02_a.log; cat gpurun_out/r02_a_mix_sweep.log gpurun_out/r02_a_flash.log; grep "^{" gpurun_out/bench_r02_a.log | cut -c1-600
