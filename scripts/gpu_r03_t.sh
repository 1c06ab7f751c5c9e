# round-3 checkpoint t: bench lines at the committed code (default = headline, B=64, the other BASELINE configs)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python bench.py > $O/r03_t_bench_small1024_auto.log 2>&1
timeout 600 python bench.py --batch 64 --no-cpu-baseline > $O/r03_t_bench_small1024_b64.log 2>&1
timeout 900 python bench.py --workload small-4096-fp16 --no-cpu-baseline > $O/r03_t_bench_4096.log 2>&1
timeout 900 python bench.py --workload mini-k64-1024 --no-cpu-baseline > $O/r03_t_bench_mini.log 2>&1
grep -h "^{" $O/r03_t_bench_*.log | cut -c1-250
