#!/bin/bash
# round 6, run g: wide senses on the LDS-DMA ring (csrc/sense_wide_dma.hip) -- parity tests, then the same-box A/B against
# the staged kernels of sense_wide.hip (variant build `widestaged`: -DBP_WIDE_NO_DMA)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
TAG=r06_g_wide bash scripts/gpu_run.sh tests -x -q -k "wide or few_sense"
TAG=r06_g_k4 bash scripts/gpu_run.sh ab default,widestaged --which lse,mix --batch 256,1024 --reps 2 --extra "--senses 4 --d 640"
TAG=r06_g_k1 bash scripts/gpu_run.sh ab default,widestaged --which lse,mix --batch 256,1024 --reps 2 --extra "--senses 1 --d 640"
