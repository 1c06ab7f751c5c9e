#!/bin/bash
# round 6, run x: validation of the final tree as the driver will run it -- the -m gpu suite, smoke(), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | grep "passed\|failed\|error" | tail -n 3 | tee gpurun_out/r06_x_pytest_gpu_tail.txt
bash scripts/gpu_run.sh smoke
T0=$(date +%s); python bench.py > gpurun_out/r06_x_bench_default.json 2> gpurun_out/r06_x_bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_x_bench_default.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'dtype')})
print({k: v for k, v in d['roofline'].items() if k in ('kernel', 'bound', 'achieved', 'peak', 'frac', 'traffic', 'avg_launch_ms', 'sclk_mhz_mean', 'power_w_mean', 'frac_at_sustained_clock')})
print(d['content_per_position'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
