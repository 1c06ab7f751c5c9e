#!/bin/bash
# round 5, run a: the new parity tests (gather offsets beyond 2 GiB, config 4 at its real size, token tables against the
# oracle, cached whole-vocabulary table), the default bench with the three content orders, content-order threshold sweep,
# generation before/after, training-step batch sweep, mix / backward kernel baselines for this round's kernel work
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=gpurun_out/r05_a
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -m gpu -x -q -k "oracle or whole_model or sense_table or dedup or config4 or config1" > $O/pytest_models.log 2>&1; echo "exit $?" >> $O/pytest_models.log
tail -4 $O/pytest_models.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stress.py -m gpu -x -q -k "gather or varlen_reads" > $O/pytest_gather.log 2>&1; echo "exit $?" >> $O/pytest_gather.log
tail -4 $O/pytest_gather.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python scripts/bench_content_modes.py > $O/content_modes_small.jsonl 2> $O/content_modes_small.err
timeout 300 python bench.py --workload micro-128 --no-cpu-baseline --batch 4 --steps 50 --warmup 5 > $O/bench_micro.json 2> $O/bench_micro.err
timeout 300 python bench.py --workload micro-128 --no-cpu-baseline --batch 4 --steps 50 --warmup 5 --graph > $O/bench_micro_graph.json 2> $O/bench_micro_graph.err
timeout 300 python bench.py --workload micro-128 --no-cpu-baseline --batch 4 --steps 50 --warmup 5 --graph --content position > $O/bench_micro_graph_pp.json 2> $O/bench_micro_graph_pp.err
timeout 600 python scripts/bench_generate.py > $O/generate.jsonl 2> $O/generate.err
for b in 32 64 128 192; do timeout 600 python scripts/bench_train_step.py --batch $b >> $O/train_step_sweep.jsonl 2>> $O/train_step_sweep.err; done
timeout 600 python scripts/bench_kernels.py --which mix,bwd,mixbwd --batch 64 > $O/kernels_b64.jsonl 2> $O/kernels_b64.err
timeout 600 python scripts/bench_kernels.py --which mix --batch 512 > $O/kernels_mix_b512.jsonl 2>> $O/kernels_b64.err
python - <<'PY'
import json
for f in ('bench_default','bench_micro','bench_micro_graph','bench_micro_graph_pp'):
    try:
        d=json.loads(open('gpurun_out/r05_a/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['config']['batch_per_gpu'], d['config']['content_network'][:30], {k:(v or {}).get('value') for k,v in d.items() if k.startswith('content_')}, d.get('roofline',{}).get('frac'), [(k['kernel'][:14],k['avg_ms'],k['mfma_frac']) for k in d.get('kernels',[])])
    except Exception as e: print(f,'ERR',e)
for f in ('content_modes_small.jsonl','generate.jsonl','train_step_sweep.jsonl','kernels_b64.jsonl','kernels_mix_b512.jsonl'):
    print('==',f); print(open('gpurun_out/r05_a/'+f).read()[:3000])
PY
