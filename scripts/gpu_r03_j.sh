cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_dropout.py -q -m gpu -x > $O/r03_j_bwd.log 2>&1; tail -3 $O/r03_j_bwd.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu --timeout 900 -x -s -k "config3 or torchrun" > $O/r03_j_cfg.log 2>&1
grep -v "^  File\|Extension" $O/r03_j_cfg.log | tail -25 | cut -c1-200
bash scripts/gpu_ab.sh ns2 --which bwd --batch 64 --iters 20 > /dev/null; cp $O/ab_ns2.log $O/r03_j_ab_b64.log
bash scripts/gpu_ab.sh ns2 --which bwd --batch 32 --iters 20 > /dev/null; cat $O/r03_j_ab_b64.log $O/ab_ns2.log
