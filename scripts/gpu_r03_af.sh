cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x > $O/r03_af_pytest.log 2>&1; tail -3 $O/r03_af_pytest.log | cut -c1-300
python bench.py --batch 512 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | grep "^{" > $O/r03_af_bench_b512.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r03_af_bench_b512.json'))
print(r['value'], r['ms_per_step'])
for k in r['kernels']: print(k)
PY
