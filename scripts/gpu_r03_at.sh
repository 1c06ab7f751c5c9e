cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_mixsplit.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "mix or sense" 2>&1 | grep "passed\|failed" | tail -1
L=gpurun_out/r03_at_mix_simple_split_ab.jsonl; : > $L
for rep in 1 2 3; do
for lib in default mixsplit; do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$lib.so; fi
  for B in 4 64 128; do
  python scripts/bench_kernels.py --which mix --batch $B --iters 20 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", /" >> $L
  done
done; done
python - <<'PY'
import json, collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r03_at_mix_simple_split_ab.jsonl'):
    r=json.loads(l); d[(r['lib'], r['batch'])].append(r['ms'])
for k in sorted(d): print(k, [round(x,4) for x in d[k]])
PY
