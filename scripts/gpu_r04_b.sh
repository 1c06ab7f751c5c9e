#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_b
export TMPDIR=/tmp
python scripts/debug/r04_bwd_fault.py 2>&1 | grep -c " ok " 
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04_b/pytest_full.log 2>&1; echo "pytest exit $?" >> gpurun_out/r04_b/pytest_full.log
tail -5 gpurun_out/r04_b/pytest_full.log
