#!/bin/bash
# round 4, run f: evidence on the shipped kernels -- rocprofv3 kernel stats of the default bench command, PMC passes at the
# batch the bench picks (1920) and at 64, the other BASELINE workloads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out/r04_f
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_f/prof -- python $R/bench.py --no-cpu-baseline --batch 1920 --steps 5 --warmup 2 > $R/gpurun_out/r04_f/bench_under_rocprof.json 2> $R/gpurun_out/r04_f/bench_under_rocprof.err
cd $R
db=$(ls gpurun_out/r04_f/prof/*/*_results.db 2>/dev/null | head -1); python scripts/rocprof_summary.py $db gpurun_out/r04_f/kernel_stats_small1024_b1920.txt | head -24
bash scripts/gpu_pmc.sh r04_f_b1920 --which flash,lse,mix --batch 1920 --iters 3; cp gpurun_out/pmc_r04_f_b1920/summary.txt gpurun_out/r04_f/pmc_small_b1920.txt
bash scripts/gpu_pmc.sh r04_f_b64 --which flash,lse,mix,bwd,mixbwd --batch 64 --iters 5; cp gpurun_out/pmc_r04_f_b64/summary.txt gpurun_out/r04_f/pmc_small_b64.txt
rm -rf gpurun_out/pmc_r04_f_b1920/*/ gpurun_out/pmc_r04_f_b64/*/ gpurun_out/r04_f/prof
timeout 900 python bench.py > gpurun_out/r04_f/bench_default.json 2> gpurun_out/r04_f/bench_default.err
timeout 900 python bench.py --workload small-4096-fp16 --no-cpu-baseline > gpurun_out/r04_f/bench_4096.json 2> gpurun_out/r04_f/bench_4096.err
timeout 900 python bench.py --workload mini-k64-1024 --no-cpu-baseline > gpurun_out/r04_f/bench_mini.json 2> gpurun_out/r04_f/bench_mini.err
python - <<'PY'
import json
for f in ('bench_default','bench_4096','bench_mini'):
    try:
        d=json.loads(open('gpurun_out/r04_f/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['config']['batch_per_gpu'], d['config'].get('hbm_frac_peak'), [(k['kernel'][:14],k['avg_ms'],k['mfma_frac'],k['hbm_frac']) for k in d['kernels']])
    except Exception as e: print(f,'ERR',e)
PY
