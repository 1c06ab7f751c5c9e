# round-2 checkpoint t: final evidence at the committed code: default bench, B=64 bench, kernel stats and PMC (instruction ratios) for the flash forward
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r02_t_smoke.log 2>&1
timeout 1200 python bench.py > $O/r02_t_bench_small1024_auto.log 2>&1
timeout 600 python bench.py --batch 64 --no-cpu-baseline > $O/r02_t_bench_small1024_b64.log 2>&1
timeout 600 python bench.py --workload small-4096-fp16 --no-cpu-baseline > $O/r02_t_bench_4096.log 2>&1
timeout 600 python bench.py --workload mini-k64-1024 --no-cpu-baseline > $O/r02_t_bench_mini.log 2>&1
prof() { tag=$1; shift; (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/prof_$tag.log 2>&1)
  db=$(ls $O/prof_$tag/*/*_results.db 2>/dev/null | head -1); [ -n "$db" ] && python scripts/rocprof_summary.py $db $O/r02_t_kernel_stats_$tag.txt > /dev/null; rm -rf $O/prof_$tag; }
prof small1024_b64 --batch 64
prof small1024_b512 --batch 512
bash scripts/gpu_pmc.sh r02_t_small_b64 --which flash,lse,mix,alpha --batch 64 --iters 5
bash scripts/gpu_pmc.sh r02_t_small_b512 --which flash,lse,mix --batch 512 --iters 3
bash scripts/gpu_pmc.sh r02_t_4096_noncausal_b16 --which flash --batch 16 --seq 4096 --noncausal --iters 5
for t in small_b64 small_b512 4096_noncausal_b16; do cp $O/pmc_r02_t_$t/summary.txt $O/r02_t_pmc_$t.txt; rm -rf $O/pmc_r02_t_$t; done
tail -1 $O/r02_t_smoke.log; grep -h "^{" $O/r02_t_bench_*.log | cut -c1-400
