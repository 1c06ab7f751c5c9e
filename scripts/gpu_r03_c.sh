cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_dense.py -q -m gpu --timeout 900 -x > $O/t_r03_c_fd.log 2>&1
tail -15 $O/t_r03_c_fd.log
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu --timeout 900 -x -s -k "config3 or torchrun" > $O/t_r03_c_cfg.log 2>&1
tail -40 $O/t_r03_c_cfg.log
timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_gpu_dropout.py -q -m gpu --timeout 900 -x -k "graphs or queue_ws or generator or rescale" > $O/t_r03_c_misc.log 2>&1
tail -15 $O/t_r03_c_misc.log
python scripts/bench_kernels.py --which gelu --batch 32 --iters 10 > $O/r03_c_gelu.jsonl 2>&1
python scripts/bench_kernels.py --which gelu --batch 32 --d 3072 --iters 10 >> $O/r03_c_gelu.jsonl 2>&1
cat $O/r03_c_gelu.jsonl
