# round-3 checkpoint b: new tests (retry, fused dense, queue_ws stress, config-3 readiness, generator) + bias/GELU rates
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests/test_gpu_retry.py tests/test_gpu_fused_dense.py tests/test_gpu_stress.py tests/test_gpu_configs.py tests/test_gpu_dropout.py tests/test_gpu_backward.py tests/test_gpu_model.py -q -m gpu --timeout 900 > $O/t_r03_b_full.log 2>&1
tail -5 $O/t_r03_b_full.log > $O/t_r03_b.log
grep -E "^FAILED|^ERROR" $O/t_r03_b_full.log | head -40 >> $O/t_r03_b.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout 900 -k "4096" > $O/t_r03_b_4096.log 2>&1
tail -15 $O/t_r03_b_4096.log >> $O/t_r03_b.log
python scripts/bench_kernels.py --which gelu --batch 32 --iters 10 > $O/r03_b_gelu.jsonl 2>&1
python scripts/bench_kernels.py --which gelu --batch 32 --d 3072 --iters 10 >> $O/r03_b_gelu.jsonl 2>&1
for b in 32; do timeout 600 python scripts/bench_train_step.py --batch $b; done > $O/r03_b_train.log 2>&1
cat $O/t_r03_b.log; cat $O/r03_b_gelu.jsonl; grep -h "^{" $O/r03_b_train.log
