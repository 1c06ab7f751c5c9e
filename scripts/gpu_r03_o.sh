cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_dropout.py tests/test_gpu_configs.py -q -m gpu -x > $O/r03_o_bwd.log 2>&1; tail -3 $O/r03_o_bwd.log | cut -c1-300
python scripts/bench_kernels.py --which bwd --batch 64 --iters 20 2>/dev/null | grep "^{" > $O/r03_o_bwd.jsonl
python scripts/bench_kernels.py --which bwd --batch 32 --iters 20 2>/dev/null | grep "^{" >> $O/r03_o_bwd.jsonl
cat $O/r03_o_bwd.jsonl
BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_bwdprof.so timeout 300 python scripts/probes/flash_bwd_timeline/timeline.py --batch 64 > $O/r03_o_bwd_timeline_b64.txt 2>&1
cat $O/r03_o_bwd_timeline_b64.txt
