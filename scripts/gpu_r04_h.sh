#!/bin/bash
# round 4, run h: LayerNorm rowscale / colscale (ABI 5) -- the reference's sweep incl. the scaled rows, then the whole suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_h
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dropout.py -m gpu -x -q -k "layer_norm" > gpurun_out/r04_h/pytest_ln.log 2>&1; tail -3 gpurun_out/r04_h/pytest_ln.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_h/pytest_full.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_h/pytest_full.log
grep -E "passed|failed|exit" gpurun_out/r04_h/pytest_full.log | tail -3
timeout 300 python scripts/bench_kernels.py --which ln --batch 64 2>&1 | tail -2
