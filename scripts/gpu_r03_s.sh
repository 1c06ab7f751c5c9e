cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for v in gelu8 gelu2; do bash scripts/gpu_ab.sh $v --which gelu --batch 32 --iters 20 > /dev/null; grep "==\|bias_gelu_bwd" $O/ab_$v.log; bash scripts/gpu_ab.sh $v --which gelu --batch 32 --d 3072 --iters 20 > /dev/null; grep "==\|bias_gelu_bwd" $O/ab_$v.log; done
timeout 1500 python bench.py > $O/r03_s_bench_small1024_auto.log 2>&1; grep -h "^{" $O/r03_s_bench_small1024_auto.log | cut -c1-300
