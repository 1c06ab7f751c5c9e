#!/bin/bash
# round 6, run o: table form of the wide ring kernels against "torch gathers, then bp_sense_mix" on one box; the tests again
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
TAG=r06_o_wide bash scripts/gpu_run.sh tests -x -q -k "wide or few_sense or gather or table"
TAG=r06_o_k4 bash scripts/gpu_run.sh ab default --which mixgather,mixgatherref --batch 1024 --reps 2 --extra "--senses 4 --d 640"
TAG=r06_o_k1 bash scripts/gpu_run.sh ab default --which mixgather,mixgatherref --batch 1024 --reps 2 --extra "--senses 1 --d 640"
