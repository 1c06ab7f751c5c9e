# round-1 checkpoint f: full GPU suite, smoke, bench (+other workloads), rocprof of the bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -4 > gpurun_out/t_r01_g.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r01_g.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r01_g.log 2>&1
for w in small-4096-fp16 mini-k64-1024 micro-128; do timeout 900 python bench.py --workload $w --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/bench_r01_g_$w.json; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_g -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_g.log 2>&1
cd "$GRAFT_REPO_ROOT"
python scripts/rocprof_summary.py $(find gpurun_out/prof_r01_g -name "*.db" | head -1) > gpurun_out/r01_g_kernel_stats.txt 2>&1
cat gpurun_out/t_r01_g.log; tail -1 gpurun_out/smoke_r01_g.log; grep "^{" gpurun_out/bench_r01_g.log | cut -c1-1500; head -10 gpurun_out/r01_g_kernel_stats.txt | cut -c1-170
for w in small-4096-fp16 mini-k64-1024 micro-128; do python -c "import json; d=json.load(open('gpurun_out/bench_r01_g_$w.json')); print('$w', d['value'], d['ms_per_step'])"; done
