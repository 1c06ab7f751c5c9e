# round-2 checkpoint v: final validation of the committed code + PMC at the batches the sweeps pick
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $O/t_r02_v_full.log 2>&1
grep -E "passed|failed|error" $O/t_r02_v_full.log | tail -3 > $O/t_r02_v.log
grep -E "^FAILED|^ERROR" $O/t_r02_v_full.log | head -20 >> $O/t_r02_v.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r02_v_smoke.log 2>&1
timeout 1200 python bench.py > $O/r02_v_bench_small1024_auto.log 2>&1
timeout 600 python bench.py --batch 64 --no-cpu-baseline > $O/r02_v_bench_small1024_b64.log 2>&1
for b in 16 32; do timeout 600 python scripts/bench_train_step.py --batch $b; done > $O/r02_v_train.log 2>&1
bash scripts/gpu_pmc.sh r02_v_small_b64 --which flash,lse,mix --batch 64 --iters 5
bash scripts/gpu_pmc.sh r02_v_small_b1024 --which flash,lse,mix --batch 1024 --iters 3
bash scripts/gpu_pmc.sh r02_v_small_b1536 --which flash,lse,mix --batch 1536 --iters 3
bash scripts/gpu_pmc.sh r02_v_4096_b128 --which flash,lse,mix --batch 128 --seq 4096 --dtype fp16 --iters 3
bash scripts/gpu_pmc.sh r02_v_mini_b256 --which flash,lse,mix --batch 256 --heads 8 --headdim 80 --senses 64 --d 640 --iters 3
for t in small_b64 small_b1024 small_b1536 4096_b128 mini_b256; do cp $O/pmc_r02_v_$t/summary.txt $O/r02_v_pmc_$t.txt; rm -rf $O/pmc_r02_v_$t; done
cat $O/t_r02_v.log; tail -1 $O/r02_v_smoke.log; grep -h "^{" $O/r02_v_bench_*.log $O/r02_v_train.log | cut -c1-330
