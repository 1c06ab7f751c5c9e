cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_configs.py tests/test_gpu_dropout.py -q -m gpu -x > gpurun_out/r03_aj_tests.log 2>&1; grep "passed\|failed" gpurun_out/r03_aj_tests.log | tail -2
for lib in default mixold; do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$lib.so; fi
  python scripts/bench_kernels.py --which mixbwd --batch 64 --iters 10 2>/dev/null | grep "dqk" | cut -c1-160
done
