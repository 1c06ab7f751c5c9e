# round-2 checkpoint o: validation of the committed state: full GPU suite, smoke, default bench (sweep to the max that fits)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/t_r02_o_full.log 2>&1
grep -E "passed|failed|error" $O/t_r02_o_full.log | tail -3 > $O/t_r02_o.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r02_o_smoke.log 2>&1
timeout 1200 python bench.py > $O/r02_o_bench_small1024_auto.log 2>&1
cat $O/t_r02_o.log; tail -2 $O/r02_o_smoke.log; grep -h "^{" $O/r02_o_bench_small1024_auto.log | cut -c1-1500
