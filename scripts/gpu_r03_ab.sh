cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py tests/test_gpu_stress.py -q -m gpu -x -k "mix or sense or persistent" > $O/r03_ab_tests.log 2>&1; tail -4 $O/r03_ab_tests.log | cut -c1-300
L=$O/r03_ab_mix_antiphase_ab.jsonl; : > $L
for rep in 1 2 3; do
for B in 4 16 64 128; do
for lib in default mixold; do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$lib.so; fi
  python scripts/bench_kernels.py --which mix --batch $B --iters 20 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", /" >> $L
done; done; done
unset BP_HIP_LIB
python - <<'PY'
import json
for l in open('gpurun_out/r03_ab_mix_antiphase_ab.jsonl'):
    r=json.loads(l); print(r['lib'], r['batch'], round(r['ms'],4), round(r['tflops'],1))
PY
export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_mixprof.so
for B in 16 64; do python scripts/probes/mix_timeline/timeline.py --batch $B 2>&1 | grep "^{" | tee -a $O/r03_ab_mix_timeline.jsonl; done
