#!/bin/bash
# round 6, run p (final tree of round 6): the evidence set on the tree as it stands -- smoke, default bench line, rocprofv3 kernel stats of the same
# command at the batch it picks, counter passes of the hot kernels at that batch (-> profiles/traffic.json), the other
# workloads' bench lines, the training step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
T=${TAG_PREFIX:-r06_p}
echo "== full GPU suite"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/${T}_pytest_gpu_tail.txt
bash scripts/gpu_run.sh smoke
bash scripts/gpu_run.sh bench ${T}_default --steps 20 --warmup 5
B=$(python -c "import json; print(json.loads(open('gpurun_out/${T}_default_bench.json').read().strip().splitlines()[-1])['config']['batch_per_gpu'])" 2>/dev/null || echo 2048)
echo "batch picked: $B"
bash scripts/gpu_run.sh stats ${T}_small1024_b$B --steps 5 --warmup 2 --batch $B
TAG=${T} bash scripts/gpu_run.sh pmc ${T}_small_b$B --which flash,lse,mixgather,mix --batch $B --iters 3 > /dev/null
cp gpurun_out/${T}_small_b${B}_pmc.txt gpurun_out/${T}_pmc_small_b$B.txt 2>/dev/null; head -n 30 gpurun_out/${T}_small_b${B}_pmc.txt
for w in small-4096-fp16 mini-k64-1024 mini-k4-1024 mini-k1-1024; do
  bash scripts/gpu_run.sh bench ${T}_$w --workload $w --steps 10 --warmup 3 --no-cpu-baseline
done
bash scripts/gpu_run.sh bench ${T}_micro --workload micro-128 --steps 50 --warmup 10 --batch 4 --no-cpu-baseline
bash scripts/gpu_run.sh bench ${T}_micro_graph --workload micro-128 --steps 200 --warmup 10 --batch 4 --graph --no-cpu-baseline
timeout 600 python scripts/kernel_power.py --batch $B 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_kernel_power.jsonl | cut -c1-260
timeout 900 python scripts/bench_train_step.py --batch 32 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee gpurun_out/${T}_train_step_b32.jsonl
timeout 900 python scripts/bench_train_step.py --batch 192 2>&1 | grep -v amdgpu.ids | tail -n 2 | tee gpurun_out/${T}_train_step_b192.jsonl
