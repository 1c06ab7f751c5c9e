# round-3 checkpoint n: evidence at the committed code -- default bench (batch sweep to 87 % of HBM), B=64 bench, kernel stats at
# the picked batch, PMC (traffic) at the batches the sweep can pick, other BASELINE configs
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python bench.py > $O/r03_n_bench_small1024_auto.log 2>&1
timeout 600 python bench.py --batch 64 --no-cpu-baseline > $O/r03_n_bench_small1024_b64.log 2>&1
timeout 900 python bench.py --workload small-4096-fp16 --no-cpu-baseline > $O/r03_n_bench_4096.log 2>&1
timeout 900 python bench.py --workload mini-k64-1024 --no-cpu-baseline > $O/r03_n_bench_mini.log 2>&1
B=$(grep -h "^{" $O/r03_n_bench_small1024_auto.log | python -c "import sys,json; print(json.loads(sys.stdin.readline())['config']['batch_per_gpu'])")
prof() { tag=$1; shift; (cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/prof_$tag.log 2>&1)
  db=$(ls $O/prof_$tag/*/*_results.db 2>/dev/null | head -1); [ -n "$db" ] && python scripts/rocprof_summary.py $db $O/r03_n_kernel_stats_$tag.txt > /dev/null; rm -rf $O/prof_$tag; }
prof small1024_b$B --batch $B
for b in 512 2048 2304; do bash scripts/gpu_pmc.sh r03_n_small_b$b --which flash,lse,mix --batch $b --iters 3 > /dev/null 2>&1; cp $O/pmc_r03_n_small_b$b/summary.txt $O/r03_n_pmc_small_b$b.txt; rm -rf $O/pmc_r03_n_small_b$b; done
echo "picked batch $B"; grep -h "^{" $O/r03_n_bench_*.log | cut -c1-600
