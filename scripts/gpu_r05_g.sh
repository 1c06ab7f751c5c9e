#!/bin/bash
# round 5, run g: validation of the tree as it stands -- full GPU suite, smoke(), the default bench, rocprofv3 kernel stats of
# the same command at the batch it picks (2048), PMC passes for all three single-GPU workloads at their bench batches,
# the other workloads, training step (batch sweep top + kernel stats), dQ kernel at four waves per SIMD (A/B)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=gpurun_out/r05_g
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1; echo "pytest exit $?" >> $O/pytest_full.log
grep -E "passed|failed|exit" $O/pytest_full.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/bench.py --no-cpu-baseline --batch 2048 --steps 5 --warmup 2 > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err)
db=$(ls $O/prof/*/*_results.db 2>/dev/null | head -1); python scripts/rocprof_summary.py $db $O/kernel_stats_small1024_b2048.txt | head -9; rm -rf $O/prof
bash scripts/gpu_pmc.sh r05_g_small --which flash,lse,mixgather --batch 2048 --iters 3; cp gpurun_out/pmc_r05_g_small/summary.txt $O/pmc_small_b2048.txt
bash scripts/gpu_pmc.sh r05_g_4096 --which flash,lse,mixgather --batch 256 --seq 4096 --dtype fp16 --iters 3; cp gpurun_out/pmc_r05_g_4096/summary.txt $O/pmc_small4096_fp16_b256.txt
bash scripts/gpu_pmc.sh r05_g_mini --which flash,lse,mixgather --batch 1024 --heads 8 --headdim 80 --senses 64 --d 640 --iters 3; cp gpurun_out/pmc_r05_g_mini/summary.txt $O/pmc_mini_k64_b1024.txt
rm -rf gpurun_out/pmc_r05_g_*/*/
timeout 900 python bench.py --workload small-4096-fp16 --no-cpu-baseline > $O/bench_4096.json 2> $O/bench_4096.err
timeout 900 python bench.py --workload mini-k64-1024 --no-cpu-baseline > $O/bench_mini.json 2> $O/bench_mini.err
timeout 600 python scripts/bench_train_step.py --batch 32 > $O/train_step.jsonl 2> $O/train_step.err
timeout 600 python scripts/bench_train_step.py --batch 256 >> $O/train_step.jsonl 2>> $O/train_step.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/proft -- python $R/scripts/bench_train_step.py --batch 32 --steps 4 --warmup 2 > $R/$O/train_under_rocprof.json 2> $R/$O/train_under_rocprof.err)
db=$(ls $O/proft/*/*_results.db 2>/dev/null | head -1); python scripts/rocprof_summary.py $db $O/kernel_stats_train_small1024_b32.txt | head -16; rm -rf $O/proft
python scripts/ab_kernels.py --libs default,dq4 --which bwd --batch 64,256 --reps 3 --out $O/ab_flash_bwd_dq_four_waves.jsonl | tail -5
python - <<'PY'
import json
for f in ('bench_default','bench_under_rocprof','bench_4096','bench_mini'):
    try:
        d=json.loads(open('gpurun_out/r05_g/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['config']['batch_per_gpu'], d['config'].get('hbm_frac_peak'), {k:(v or {}).get('value') for k,v in d.items() if k.startswith('content_')}, d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), [(k['kernel'][:14],k['avg_ms'],k['mfma_frac'],k['hbm_frac']) for k in d['kernels']])
    except Exception as e: print(f,'ERR',e)
print(open('gpurun_out/r05_g/train_step.jsonl').read()[:900])
PY
