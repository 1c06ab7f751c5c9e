# round-2 checkpoint g: full suite on the fused backward, bench with --batch auto, train step (reference recipe) + trace
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -15 > gpurun_out/t_r02_g.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02_g_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r02_g.log 2>&1
for b in 16 32; do timeout 600 python scripts/bench_train_step.py --batch $b; done > gpurun_out/r02_g_train.log 2>&1
timeout 600 python scripts/bench_train_step.py --batch 32 --pure-bf16 --dropout 0 >> gpurun_out/r02_g_train.log 2>&1
timeout 600 python scripts/bench_kernels.py --which mixbwd,alpha --batch 64 --iters 10 > gpurun_out/r02_g_k.log 2>&1
timeout 600 python scripts/bench_kernels.py --which mixbwd --batch 16 --iters 10 >> gpurun_out/r02_g_k.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02_g_train -- python $GRAFT_REPO_ROOT/scripts/bench_train_step.py --batch 32 --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02_g_train.log 2>&1)
db=$(ls gpurun_out/prof_r02_g_train/*/*_results.db 2>/dev/null | head -1)
[ -n "$db" ] && python scripts/rocprof_summary.py $db gpurun_out/r02_g_train_step_kernel_stats_small1024_b32.txt > /dev/null
rm -rf gpurun_out/prof_r02_g_train
cat gpurun_out/t_r02_g.log gpurun_out/r02_g_smoke.log | tail -25; grep -v amdgpu.ids gpurun_out/bench_r02_g.log gpurun_out/r02_g_train.log gpurun_out/r02_g_k.log | cut -c1-1500; head -45 gpurun_out/r02_g_train_step_kernel_stats_small1024_b32.txt | cut -c1-170
