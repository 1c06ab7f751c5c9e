cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_stress.py tests/test_gpu_configs.py -q -m gpu -x -k "mix or sense or backpack or persistent or config3 or mini" > $O/r03_ah_tests.log 2>&1; grep "passed\|failed" $O/r03_ah_tests.log | tail -2
L=$O/r03_ah_dc_antiphase_ab.jsonl; : > $L
for rep in 1 2 3; do
for B in 8 32 64; do
for lib in default mixold; do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$lib.so; fi
  python scripts/bench_kernels.py --which mixbwd --batch $B --iters 10 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", /" >> $L
done; done; done
unset BP_HIP_LIB
python - <<'PY'
import json
for l in open('gpurun_out/r03_ah_dc_antiphase_ab.jsonl'):
    r=json.loads(l); print(r['lib'], r['kernel'], r.get('batch'), round(r['ms'],4), round(r.get('tflops',0),1))
PY
