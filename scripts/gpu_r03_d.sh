cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
HIP_LAUNCH_BLOCKING=1 timeout 300 python scripts/debug/r03_fault_a.py > $O/r03_d_fault_a.log 2>&1
tail -20 $O/r03_d_fault_a.log | cut -c1-200
HIP_LAUNCH_BLOCKING=1 timeout 600 python -m pytest tests/test_gpu_stress.py -q -m gpu -x -s -k "queue_ws" > $O/r03_d_q1.log 2>&1
head -5 $O/r03_d_q1.log | cut -c1-200; tail -3 $O/r03_d_q1.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_stress.py -q -m gpu -x -s -k "graphs" > $O/r03_d_q2.log 2>&1
head -5 $O/r03_d_q2.log | cut -c1-200; tail -3 $O/r03_d_q2.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_dropout.py tests/test_gpu_configs.py -q -m gpu -x -k "not config3 and not torchrun" > $O/r03_d_bwd.log 2>&1
tail -12 $O/r03_d_bwd.log | cut -c1-300
python scripts/bench_kernels.py --which bwd --batch 64 --iters 10 > $O/r03_d_bwd.jsonl 2>&1
python scripts/bench_kernels.py --which bwd --batch 32 --iters 10 >> $O/r03_d_bwd.jsonl 2>&1
cat $O/r03_d_bwd.jsonl
