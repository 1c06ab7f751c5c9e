#!/bin/bash
# round 6, run zc (after the XCD placement change): counter passes of the hot kernels at the bench batches of the other workloads on the final tree
# (-> profiles/traffic.json: roofline.traffic of their bench lines)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
bash scripts/gpu_run.sh pmc r06_zc_mini_k4_b1024 --which flash,lse,mixgather --batch 1024 --heads 8 --headdim 80 --senses 4 --d 640 --iters 3 > /dev/null
bash scripts/gpu_run.sh pmc r06_zc_mini_k1_b1024 --which flash,lse,mixgather --batch 1024 --heads 8 --headdim 80 --senses 1 --d 640 --iters 3 > /dev/null
bash scripts/gpu_run.sh pmc r06_zc_mini_k64_b1024 --which flash,lse,mixgather --batch 1024 --heads 8 --headdim 80 --senses 64 --d 640 --iters 3 > /dev/null
bash scripts/gpu_run.sh pmc r06_zc_small4096_fp16_b256 --which flash,lse,mixgather --batch 256 --seq 4096 --dtype fp16 --iters 3 > /dev/null
ls gpurun_out | grep r06_zc
