#!/bin/bash
# round 5, run q: validation of the committed tree after the per-row retry, the property / drawn-shape tests and the few-sense path: full GPU suite, smoke(), default bench, the other workloads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_q
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1; echo "pytest exit $?" >> $O/pytest_full.log
grep -E "passed|failed|exit" $O/pytest_full.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 python bench.py --workload small-4096-fp16 --no-cpu-baseline > $O/bench_4096.json 2> $O/bench_4096.err
timeout 900 python bench.py --workload mini-k64-1024 --no-cpu-baseline > $O/bench_mini.json 2> $O/bench_mini.err
timeout 300 python bench.py --workload micro-128 --no-cpu-baseline --batch 4 --steps 50 --warmup 5 --graph > $O/bench_micro_graph.json 2> $O/bench_micro_graph.err
python - <<'PY'
import json
for f in ('bench_default','bench_4096','bench_mini','bench_micro_graph'):
    try:
        d=json.loads(open('gpurun_out/r05_q/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['config']['batch_per_gpu'], d['config'].get('hbm_frac_peak'), {k:(v or {}).get('value') for k,v in d.items() if k.startswith('content_')}, d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), [(k['kernel'][:14],k['avg_ms'],k['mfma_frac'],k['hbm_frac']) for k in d.get('kernels',[])])
    except Exception as e: print(f,'ERR',e)
PY
