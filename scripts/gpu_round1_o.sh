# round-1 checkpoint e: full GPU suite, smoke, bench, rocprof of the bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -4 > gpurun_out/t_r01_e.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r01_e.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r01_e.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_e -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_e.log 2>&1
cd "$GRAFT_REPO_ROOT"
db=$(find gpurun_out/prof_r01_e -name "*.db" | head -1)
python scripts/rocprof_summary.py $db > gpurun_out/r01_e_kernel_stats.txt 2>&1
cat gpurun_out/t_r01_e.log; tail -2 gpurun_out/smoke_r01_e.log; grep "^{" gpurun_out/bench_r01_e.log | cut -c1-700; head -12 gpurun_out/r01_e_kernel_stats.txt | cut -c1-170
