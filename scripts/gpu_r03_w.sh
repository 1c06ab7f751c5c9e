# round-3 checkpoint w: final validation of the committed tree -- full GPU suite, smoke, default bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --timeout 900 > $O/t_r03_w_full.log 2>&1
grep -E "passed|failed|error" $O/t_r03_w_full.log | tail -3 > $O/t_r03_w.log
grep -E "^FAILED|^ERROR" $O/t_r03_w_full.log | head -20 >> $O/t_r03_w.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03_w_smoke.log 2>&1
timeout 1500 python bench.py > $O/r03_w_bench_small1024_auto.log 2>&1
cat $O/t_r03_w.log; tail -2 $O/r03_w_smoke.log; grep -h "^{" $O/r03_w_bench_small1024_auto.log | cut -c1-300
