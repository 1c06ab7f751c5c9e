cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for w in mini-k64-1024 small-4096-fp16; do
python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep "^{" > $O/r03_ak_bench_$w.json
python - <<PY
import json
r=json.load(open('gpurun_out/r03_ak_bench_$w.json'))
print('$w', r['value'], r['ms_per_step'], r['config']['batch_per_gpu'])
for k in r['kernels']: print('  ', k['kernel'], k['avg_ms'], k.get('mfma_frac'), k.get('hbm_frac'))
PY
done
