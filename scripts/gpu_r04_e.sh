#!/bin/bash
# round 4, run e: full GPU suite (new tests), default bench with the persistent logits block, LayerNorm nt variant in the model, train step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_e
export TMPDIR=/tmp
L=$PWD/backpacks-flash-attn_amd/bp_hip
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_e/pytest_full.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_e/pytest_full.log
tail -4 gpurun_out/r04_e/pytest_full.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04_e/bench_default.json 2> gpurun_out/r04_e/bench_default.err; tail -3 gpurun_out/r04_e/bench_default.err
python - <<'PY'
import json
for f in ('gpurun_out/r04_e/bench_default.json',):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['config']['batch_per_gpu'], d['config'].get('hbm_frac_peak'))
        for k in d['kernels']: print('  ', k['kernel'], k['avg_ms'], k['mfma_frac'], k['hbm_frac'])
        for r in d.get('batch_sweep', []): print('   sweep', r)
    except Exception as e: print(f, 'ERR', e)
PY
for lib in default lnnt4; do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$L/libbackpack_hip_$lib.so; fi
  for b in 64 1536; do
    timeout 600 python bench.py --no-cpu-baseline --batch $b --steps 10 > gpurun_out/r04_e/bench_${lib}_b$b.json 2>> gpurun_out/r04_e/bench_ab.err
    python -c "
import json;d=json.loads(open('gpurun_out/r04_e/bench_${lib}_b$b.json').read().strip().splitlines()[-1]);print('$lib b$b', d['value'], d['ms_per_step'], [ (k['kernel'][:12],k['avg_ms']) for k in d['kernels']])"
  done
done
unset BP_HIP_LIB
timeout 600 python scripts/bench_train_step.py --batch 32 > gpurun_out/r04_e/train_step.jsonl 2> gpurun_out/r04_e/train_step.err; cat gpurun_out/r04_e/train_step.jsonl
