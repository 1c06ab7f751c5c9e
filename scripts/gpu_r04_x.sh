#!/bin/bash
# round 4, run x: bp_sense_mix_gather (the mix kernel reads the per-token table itself): parity + bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04_x
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_stress.py tests/test_abi.py -m gpu -q -x -k "mix or dedup or model or config or abi or nano" > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2
timeout 900 python bench.py --batch 1664 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_b1664.json 2> $O/bench_b1664.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench_auto.json 2> $O/bench_auto.err
python - <<'PY'
import json
for f in ('bench_b1664','bench_auto'):
    try:
        d=json.loads(open('gpurun_out/r04_x/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['config']['batch_per_gpu'], d['config'].get('hbm_frac_peak'), d.get('content_per_position'), d.get('roofline',{}).get('frac'), [(k['kernel'][:14],k['avg_ms'],k['mfma_frac']) for k in d['kernels']])
    except Exception as e: print(f,'ERR',e)
PY
tail -3 $O/bench_auto.err
