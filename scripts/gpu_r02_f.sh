# round-2 checkpoint f: fused sense-mix backward
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py -q -m gpu --timeout 900 -x -k "sense_mix_backward or training_step" 2>&1 | tail -60 > gpurun_out/t_r02_f.log
timeout 600 python scripts/bench_train_step.py > gpurun_out/r02_f_train.log 2>&1
cat gpurun_out/t_r02_f.log | tail -50; tail -5 gpurun_out/r02_f_train.log
