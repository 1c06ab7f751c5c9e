#!/bin/bash
# round 6, run z: consecutive (sample, head) groups on one XCD in the flash forward / LSE pre-pass (variant xseq:
# -DBP_FWD_XCD_SEQ=1) -- heads whose rows share cache lines (d_h = 80, the senses' 96- / 32-byte rows) then share an L2
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
L=$PWD/backpacks-flash-attn_amd/bp_hip
TAG=r06_z_mini bash scripts/gpu_run.sh ab default,xseq --which flash --batch 1024 --reps 3 --extra "--heads 8 --headdim 80"
TAG=r06_z_small bash scripts/gpu_run.sh ab default,xseq --which flash,lse --batch 2048 --reps 3
TAG=r06_z_k64 bash scripts/gpu_run.sh ab default,xseq --which lse --batch 1024 --reps 3 --extra "--senses 64 --d 640"
TAG=r06_z_4096 bash scripts/gpu_run.sh ab default,xseq --which flash,lse --batch 256 --seq 4096 --reps 2 --extra "--dtype fp16"
BP_HIP_LIB=$L/libbackpack_hip_xseq.so bash scripts/gpu_run.sh pmc r06_z_xseq_mini --which flash,lse --batch 1024 --heads 8 --headdim 80 --senses 64 --d 640 --iters 3 > /dev/null
grep "bp::\|FETCH_SIZE\|TCC_HIT\|TCC_MISS" gpurun_out/r06_z_xseq_mini_pmc.txt | grep -v arm_mix -A3 | head -12
