cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_dense.py -q -m gpu -x > $O/r03_x_fd.log 2>&1; tail -3 $O/r03_x_fd.log | cut -c1-200
python scripts/bench_kernels.py --which gelu --batch 32 --iters 20 2>/dev/null | grep "^{" > $O/r03_x_gelu.jsonl
python scripts/bench_kernels.py --which gelu --batch 32 --d 3072 --iters 20 2>/dev/null | grep "^{" >> $O/r03_x_gelu.jsonl
cat $O/r03_x_gelu.jsonl
