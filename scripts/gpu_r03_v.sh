cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python scripts/debug/r03_bwd_case.py > $O/r03_v_case.log 2>&1; grep -v amdgpu $O/r03_v_case.log | grep -v "^  File\|Extension" | cut -c1-200 | head -12
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_dropout.py tests/test_gpu_configs.py -q -m gpu -x > $O/r03_v_bwd.log 2>&1; tail -3 $O/r03_v_bwd.log | cut -c1-300
bash scripts/gpu_ab.sh nochain --which bwd --batch 64 --iters 20 > /dev/null; cp $O/ab_nochain.log $O/r03_v_ab_b64.log
bash scripts/gpu_ab.sh nochain --which bwd --batch 32 --iters 20 > /dev/null; cat $O/r03_v_ab_b64.log $O/ab_nochain.log
