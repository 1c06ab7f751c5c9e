cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py -q -m gpu -x > $O/r03_k_bwd.log 2>&1; tail -3 $O/r03_k_bwd.log | cut -c1-300
V=${1:-nopipe}
bash scripts/gpu_ab.sh $V --which bwd --batch 64 --iters 20 > /dev/null; cp $O/ab_$V.log $O/r03_k_ab_b64.log
bash scripts/gpu_ab.sh $V --which bwd --batch 32 --iters 20 > /dev/null; cat $O/r03_k_ab_b64.log $O/ab_$V.log
