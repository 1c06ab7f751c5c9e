#!/bin/bash
# round 6, run a: baseline bench with clock / power fields and full-step content orders; GEMM tuning experiment
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_a_bench_default.json 2> gpurun_out/r06_a_bench_default.err; echo "bench rc=$?" ) 
tail -c 600 gpurun_out/r06_a_bench_default.err
( timeout 1500 python scripts/gemm_tune.py --batch 2048 --out gpurun_out/r06_a_tunable_small1024_b2048.csv > gpurun_out/r06_a_gemm_tune.jsonl 2> gpurun_out/r06_a_gemm_tune.err; echo "tune rc=$?" )
tail -c 800 gpurun_out/r06_a_gemm_tune.err
grep -v '"solution"' gpurun_out/r06_a_gemm_tune.jsonl | cut -c1-400
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_a_bench_default.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','content_per_position','content_cached_table','step_clock_power')})
print(d['roofline'])
PY
