#!/bin/bash
# round 6, run w: two 32-key blocks per ring step at d_k = 160 (variant sub2: -DBP_WIDE160_SUB=2): half the barriers and waits
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
L=$PWD/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip_sub2.so TAG=r06_w_sub2 bash scripts/gpu_run.sh tests -x -q -k "wide or few_sense or mini-k4"
TAG=r06_w_k4 bash scripts/gpu_run.sh ab default,sub2 --which lse,mix,mixgather --batch 1024 --reps 3 --extra "--senses 4 --d 640"
