# other BASELINE configs through bench.py + the new model test
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "mini_k64" 2>&1 | tail -4
for w in small-4096-fp16 mini-k64-1024 micro-128; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/bench_r01_d_$w.json
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_r01_d_$w.json'))
print('$w', d['value'], d['unit'], d['ms_per_step'], 'ms', d['config'].get('batch_per_gpu'))
for k in d['kernels']: print('   ', k['kernel'], k['launches_per_step'], round(k['avg_ms'],4), 'ms', round(k['tflops'],1), 'TF', round(k['gbps']), 'GB/s')
PY
done
