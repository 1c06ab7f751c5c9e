#!/bin/bash
# round 4 final validation on the committed tree: full GPU suite, smoke(), default bench, kernel stats under rocprofv3, the
# other workloads, training step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=gpurun_out/r04_zz
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1; echo "pytest exit $?" >> $O/pytest_full.log
grep -E "passed|failed|exit" $O/pytest_full.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/bench.py --no-cpu-baseline --batch 1664 --steps 5 --warmup 2 > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err)
db=$(ls $O/prof/*/*_results.db 2>/dev/null | head -1); python scripts/rocprof_summary.py $db $O/kernel_stats_small1024_b1664_dedup.txt | head -9; rm -rf $O/prof
timeout 900 python bench.py --workload small-4096-fp16 --no-cpu-baseline > $O/bench_4096.json 2> $O/bench_4096.err
timeout 900 python bench.py --workload mini-k64-1024 --no-cpu-baseline > $O/bench_mini.json 2> $O/bench_mini.err
timeout 600 python scripts/bench_train_step.py --batch 32 > $O/train_step.jsonl 2> $O/train_step.err
python - <<'PY'
import json
for f in ('bench_default','bench_under_rocprof','bench_4096','bench_mini'):
    try:
        d=json.loads(open('gpurun_out/r04_zz/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['config']['batch_per_gpu'], d['config'].get('hbm_frac_peak'), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), [(k['kernel'][:14],k['avg_ms'],k['mfma_frac'],k['hbm_frac']) for k in d['kernels']])
    except Exception as e: print(f,'ERR',e)
print(open('gpurun_out/r04_zz/train_step.jsonl').read()[:400])
PY
