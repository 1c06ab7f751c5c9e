# round-1 checkpoint d: full GPU test suite, bench line, rocprof kernel trace of the bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -4 > gpurun_out/t_r01_d.log
timeout 900 python bench.py > gpurun_out/bench_r01_d.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_d -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_d.log 2>&1
cd "$GRAFT_REPO_ROOT"
db=$(find gpurun_out/prof_r01_d -name "*.db" | head -1)
python scripts/rocprof_summary.py $db > gpurun_out/r01_d_kernel_stats.txt 2>&1
cat gpurun_out/t_r01_d.log; tail -2 gpurun_out/bench_r01_d.log; head -30 gpurun_out/r01_d_kernel_stats.txt | cut -c1-200
