#!/bin/bash
# round 4, run k: odd K pitch for every non-power-of-two width (d_k = 16 / 24 / 48 senses, small head dims): parity + A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_k
export TMPDIR=/tmp
L=$PWD/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip_oddall.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py tests/test_gpu_configs.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/r04_k/pytest.log 2>&1; echo "oddall tests: $(tail -1 gpurun_out/r04_k/pytest.log)"
timeout 600 python scripts/ab_kernels.py --libs default,oddall --which lse --batch 64,256 --reps 3 > gpurun_out/r04_k/ab_lse_small.log 2>&1; echo "Small senses (k=16, d_k=48)"; tail -4 gpurun_out/r04_k/ab_lse_small.log
timeout 600 python scripts/ab_kernels.py --libs default,oddall --which lse --batch 32,128 --reps 3 --extra="--senses 64 --d 1024" > gpurun_out/r04_k/ab_lse_k64.log 2>&1; echo "k=64, d_k=16"; tail -4 gpurun_out/r04_k/ab_lse_k64.log
timeout 600 python scripts/ab_kernels.py --libs default,oddall --which flash --batch 64 --reps 3 --extra="--heads 16 --headdim 48" > gpurun_out/r04_k/ab_flash_d48.log 2>&1; echo "flash d=48"; tail -2 gpurun_out/r04_k/ab_flash_d48.log
timeout 600 python scripts/ab_kernels.py --libs default,oddall --which flash --batch 64 --reps 3 --extra="--heads 24 --headdim 32" > gpurun_out/r04_k/ab_flash_d32.log 2>&1; echo "flash d=32"; tail -2 gpurun_out/r04_k/ab_flash_d32.log
