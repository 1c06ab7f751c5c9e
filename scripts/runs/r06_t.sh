#!/bin/bash
# round 6, run t: paired tickets in the narrow sense mix (variant mixp: -DBP_MIX_ORDER=2) -- a ticket = the query-tile pair
# (n-1-t, t) of one (sample, chunk) group, a group's pairs consecutive: equal-cost tickets that stream the same rows in step
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
L=$PWD/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip_mixp.so TAG=r06_t_mixp bash scripts/gpu_run.sh tests -x -q -k "sense_mix or mix_golden or gather"
TAG=r06_t_4096 bash scripts/gpu_run.sh ab default,mixp --which mix,mixgather --batch 64,256 --seq 4096 --reps 2 --extra "--dtype fp16"
TAG=r06_t_1024 bash scripts/gpu_run.sh ab default,mixp --which mix,mixgather --batch 256,2048 --seq 1024 --reps 2
TAG=r06_t_k64 bash scripts/gpu_run.sh ab default,mixp --which mixgather --batch 1024 --seq 1024 --reps 2 --extra "--senses 64 --d 640"
BP_HIP_LIB=$L/libbackpack_hip_mixp.so bash scripts/gpu_run.sh pmc r06_t_mixp4096 --which mix,mixgather --batch 64 --seq 4096 --dtype fp16 --iters 3 > /dev/null
grep -A3 "FETCH_SIZE\|TCC_HIT\|TCC_MISS" gpurun_out/r06_t_mixp4096_pmc.txt | grep "FETCH_SIZE\|TCC_HIT\|TCC_MISS\|bp::" | head -20
