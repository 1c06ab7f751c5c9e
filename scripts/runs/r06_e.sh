#!/bin/bash
# round 6, run e: prologue order of a flash pass (first tile's DMA before the Q wait) A/B + bits; the MFMA stream with LDS operand
# traffic and its clock / power; the whole GPU suite on the tree as it stands
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
echo "== qlate bits"; timeout 900 python scripts/flash_variant_check.py --libs default,qlate 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r06_e_qlate_bits.txt
echo "== qlate A/B"
timeout 900 python scripts/ab_kernels.py --libs default+BP_BENCH_FIXED_LEN=1,qlate+BP_BENCH_FIXED_LEN=1 --which flash,lse --batch 64,256,2048 --reps 4 --out gpurun_out/r06_e_ab_flash_issue_first.jsonl 2>&1 | grep -v amdgpu.ids | tail -14
timeout 600 python scripts/ab_kernels.py --libs default+BP_BENCH_FIXED_LEN=1,qlate+BP_BENCH_FIXED_LEN=1 --which flash --batch 16,64 --seq 4096 --reps 3 --extra "--dtype fp16" 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r06_e_ab_flash_issue_first_4096.txt
echo "== mfma stream clock"
timeout 300 python scripts/mfma_stream_clock.py --seconds 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_e_mfma_stream_clock.jsonl | cut -c1-420
echo "== whole GPU suite"
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 12 | tee gpurun_out/r06_e_pytest_gpu_tail.txt
