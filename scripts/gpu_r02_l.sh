# round-2 checkpoint l: evidence run at the committed code: bench lines of all workloads, rocprofv3 kernel stats, PMC passes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python bench.py > $O/r02_l_bench_small1024_auto.log 2>&1
timeout 600 python bench.py --batch 64 --no-cpu-baseline > $O/r02_l_bench_small1024_b64.log 2>&1
timeout 600 python bench.py --workload mini-k64-1024 --no-cpu-baseline > $O/r02_l_bench_mini.log 2>&1
timeout 600 python bench.py --workload small-4096-fp16 --no-cpu-baseline > $O/r02_l_bench_4096.log 2>&1
timeout 600 python bench.py --workload micro-128 --batch 4 --no-cpu-baseline > $O/r02_l_bench_micro.log 2>&1
timeout 600 python bench.py --workload micro-128 --batch 4 --graph --no-cpu-baseline > $O/r02_l_bench_micro_graph.log 2>&1
prof() { tag=$1; shift; (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/prof_$tag.log 2>&1)
  db=$(ls $O/prof_$tag/*/*_results.db 2>/dev/null | head -1); [ -n "$db" ] && python scripts/rocprof_summary.py $db $O/r02_l_kernel_stats_$tag.txt > /dev/null; rm -rf $O/prof_$tag; }
prof small1024_b64 --batch 64
prof small1024_b512 --batch 512
prof mini_k64_1024_b32 --workload mini-k64-1024 --batch 32
prof small4096_fp16_b8 --workload small-4096-fp16 --batch 8
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_alpha -- python $GRAFT_REPO_ROOT/scripts/bench_kernels.py --which alpha --batch 64 --iters 10 > $O/prof_alpha.log 2>&1)
db=$(ls $O/prof_alpha/*/*_results.db 2>/dev/null | head -1); [ -n "$db" ] && python scripts/rocprof_summary.py $db $O/r02_l_kernel_stats_attn_probs_b64.txt > /dev/null; rm -rf $O/prof_alpha
bash scripts/gpu_pmc.sh r02_l_small_b64 --which flash,lse,mix,alpha --batch 64 --iters 5
bash scripts/gpu_pmc.sh r02_l_small_b512 --which flash,lse,mix --batch 512 --iters 3
bash scripts/gpu_pmc.sh r02_l_small_b128 --which flash,lse,mix --batch 128 --iters 5
bash scripts/gpu_pmc.sh r02_l_small_b256 --which flash,lse,mix --batch 256 --iters 5
bash scripts/gpu_pmc.sh r02_l_mini_b32 --which flash,lse,mix --batch 32 --heads 8 --headdim 80 --senses 64 --d 640 --iters 5
bash scripts/gpu_pmc.sh r02_l_4096_b8 --which flash,lse,mix --batch 8 --seq 4096 --dtype fp16 --iters 5
for t in small_b64 small_b512 small_b128 small_b256 mini_b32 4096_b8; do cp $O/pmc_r02_l_$t/summary.txt $O/r02_l_pmc_$t.txt; rm -rf $O/pmc_r02_l_$t/sq1 $O/pmc_r02_l_$t/sq2 $O/pmc_r02_l_$t/tcc1 $O/pmc_r02_l_$t/tcc2 $O/pmc_r02_l_$t/grbm; done
grep -h "^{" $O/r02_l_bench_*.log | cut -c1-700
