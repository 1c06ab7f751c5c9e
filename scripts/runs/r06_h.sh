#!/bin/bash
# round 6, run h: ring depth and tile shape of the d_k = 160 wide-sense kernels (variants: w160a = 4 waves x 320 columns,
# w160b = 8 waves x 320 columns, w160r2 = two-slot ring), d_k = 640 LSE on a three-slot ring
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
TAG=r06_h_wide bash scripts/gpu_run.sh tests -x -q -k "wide or few_sense"
TAG=r06_h_k4 bash scripts/gpu_run.sh ab default,w160a,w160b,w160r2 --which lse,mix --batch 1024 --reps 2 --extra "--senses 4 --d 640"
TAG=r06_h_k1 bash scripts/gpu_run.sh ab default --which lse,mix --batch 1024 --reps 2 --extra "--senses 1 --d 640"
for v in w160a w160b; do BP_HIP_LIB=$PWD/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$v.so TAG=r06_h_$v bash scripts/gpu_run.sh tests -x -q -k "wide_senses_lse"; done
