#!/bin/bash
# round 6, run za: consecutive groups per XCD as the default of xcd_map (variant xrr = the round-robin deal of rounds 1-5):
# the kernels that use the mapping besides the flash forward -- wide ring mix / LSE, attention backward, alpha dump -- then the
# full GPU suite on the new default
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
TAG=r06_za_k4 bash scripts/gpu_run.sh ab default,xrr --which lse,mix,mixgather --batch 1024 --reps 3 --extra "--senses 4 --d 640"
TAG=r06_za_k1 bash scripts/gpu_run.sh ab default,xrr --which lse,mix,mixgather --batch 1024 --reps 3 --extra "--senses 1 --d 640"
TAG=r06_za_bwd bash scripts/gpu_run.sh ab default,xrr --which bwd,alpha --batch 64 --reps 3
TAG=r06_za_bwd80 bash scripts/gpu_run.sh ab default,xrr --which flash,bwd --batch 64 --reps 3 --extra "--heads 8 --headdim 80"
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | grep "passed\|failed\|error" | tail -n 3 | tee gpurun_out/r06_za_pytest_gpu_tail.txt
