# round-3 final validation of the committed tree: full GPU suite, smoke, default bench, kernel stats at the picked batch,
# training step with kernel stats
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --timeout 900 > $O/t_r03_am_full.log 2>&1
grep -E "passed|failed|error" $O/t_r03_am_full.log | tail -3 > $O/r03_am_pytest_gpu_tail.txt
grep -E "^FAILED|^ERROR" $O/t_r03_am_full.log | head -20 >> $O/r03_am_pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03_am_smoke.log 2>&1
timeout 1500 python bench.py > $O/r03_am_bench_small1024_auto.log 2>&1
grep -h "^{" $O/r03_am_bench_small1024_auto.log > $O/r03_am_bench_small1024_auto.json
B=$(python -c "import json; print(json.load(open('$O/r03_am_bench_small1024_auto.json'))['config']['batch_per_gpu'])")
prof() { tag=$1; shift; (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -- python "$@" > $O/prof_$tag.log 2>&1)
  db=$(ls $O/prof_$tag/*/*_results.db 2>/dev/null | head -1); [ -n "$db" ] && python scripts/rocprof_summary.py $db $O/r03_am_kernel_stats_$tag.txt > /dev/null; rm -rf $O/prof_$tag; }
prof small1024_b$B $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 --batch $B
timeout 600 python scripts/bench_train_step.py --batch 32 --steps 5 --warmup 2 2>/dev/null | grep "^{" > $O/r03_am_train_step.jsonl
prof train_small1024_b32 $GRAFT_REPO_ROOT/scripts/bench_train_step.py --batch 32 --steps 4 --warmup 2
cat $O/r03_am_pytest_gpu_tail.txt; tail -2 $O/r03_am_smoke.log; cut -c1-400 $O/r03_am_bench_small1024_auto.json; cat $O/r03_am_train_step.jsonl | cut -c1-300
