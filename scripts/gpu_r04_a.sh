#!/bin/bash
# round 4, run a: full GPU suite on the ABI-4 / scratch-free kernels, same-box A/B against the round-3 kernels, default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_a
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_a/pytest_full.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_a/pytest_full.log
tail -5 gpurun_out/r04_a/pytest_full.log
timeout 600 python scripts/ab_kernels.py --libs default,r3k --which mix,mixbwd,bwd --batch 64,256 --reps 3 --out gpurun_out/r04_a/ab_spills.jsonl > gpurun_out/r04_a/ab_spills.log 2>&1
tail -16 gpurun_out/r04_a/ab_spills.log
timeout 900 python bench.py > gpurun_out/r04_a/bench_default.json 2> gpurun_out/r04_a/bench_default.err
tail -c 3000 gpurun_out/r04_a/bench_default.json
