#!/bin/bash
# round 6, run d: wide-sense kernels (second version: fragments in registers for d_k <= 192, 2-3 workgroups per CU) -- tests, bench
# lines against the eager op sequence of rounds 1-5; the MFMA stream with LDS operand traffic and its clock / power; then the
# whole GPU suite on the tree as it stands (16-byte epilogue stores on, ABI 8)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== wide tests"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -k "wide or few_sense" 2>&1 | tail -5 | tee gpurun_out/r06_d_pytest_wide.txt
echo "== wide-sense bench lines: native, then the eager op sequence"
for w in mini-k4-1024 mini-k1-1024; do
  for mode in native eager; do
    extra=""; [ $mode = eager ] && extra="--eager-senses"
    timeout 900 python bench.py --workload $w --steps 10 --warmup 3 --no-clock-probe --no-cpu-baseline --batch 1024 $extra > gpurun_out/r06_d_bench_${w}_$mode.json 2> gpurun_out/r06_d_bench_${w}_$mode.err; echo "$w $mode rc=$?"
    python - $w $mode <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r06_d_bench_%s_%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ('value', 'ms_per_step')}, 'per position', (d.get('content_per_position') or {}).get('value'))
    for r in d.get('kernels', []): print('   ', r['kernel'], r['avg_ms'], r['mfma_frac'], r['launches_per_step'])
except Exception as e:
    print('no line', e); print(open('gpurun_out/r06_d_bench_%s_%s.err' % (sys.argv[1], sys.argv[2])).read()[-1500:])
PY
  done
done
echo "== mfma stream clock"
timeout 300 python scripts/mfma_stream_clock.py --seconds 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_d_mfma_stream_clock.jsonl | cut -c1-420
echo "== whole GPU suite"
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 12 | tee gpurun_out/r06_d_pytest_gpu_tail.txt
