# round-3 checkpoint m: full GPU suite on the committed code + smoke + train step bench + kernel trace of the train step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --timeout 900 > $O/t_r03_m_full.log 2>&1
grep -E "passed|failed|error" $O/t_r03_m_full.log | tail -3 > $O/t_r03_m.log
grep -E "^FAILED|^ERROR" $O/t_r03_m_full.log | head -20 >> $O/t_r03_m.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r03_m_smoke.log 2>&1
for b in 32; do timeout 600 python scripts/bench_train_step.py --batch $b; done > $O/r03_m_train.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_train --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_train_step.py --batch 32 --steps 4 --warmup 2 > $O/r03_m_train_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof_train/*/*kernel_stats.csv | head -1); head -45 $f > $O/r03_m_train_step_kernel_stats_small1024_b32.txt; rm -rf $O/prof_train
cat $O/t_r03_m.log; tail -1 $O/r03_m_smoke.log; grep -h "^{" $O/r03_m_train.log | cut -c1-400; head -30 $O/r03_m_train_step_kernel_stats_small1024_b32.txt | cut -c1-200
