cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_fwdsplit.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py tests/test_gpu_dropout.py -q -m gpu -x -k "flash" > $O/r03_ap_tests.log 2>&1; echo "fwdsplit: $(grep 'passed\|failed' $O/r03_ap_tests.log | tail -1)"
L=$O/r03_ap_flash_split_issue_ab.jsonl; : > $L
for rep in 1 2 3; do
for lib in default fwdsplit; do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$lib.so; fi
  for B in 64 256; do
  python scripts/bench_kernels.py --which flash --batch $B --iters 20 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", \"shape\": \"causal S=1024 B=$B\", /" >> $L
  done
  python scripts/bench_kernels.py --which flash --batch 16 --seq 4096 --noncausal --iters 20 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", \"shape\": \"noncausal S=4096 B=16\", /" >> $L
done; done
unset BP_HIP_LIB
python - <<'PY'
import json
for l in open('gpurun_out/r03_ap_flash_split_issue_ab.jsonl'):
    r=json.loads(l); print(r['lib'], r['shape'], round(r['ms'],4), round(r['tflops'],1))
PY
