#!/bin/bash
# round 5, run b: sense mix with EVERY step in the two-phase anti-phase form (diagonal steps included): parity, then
# same-box A/B against the round-4 kernel (bp_hip/libbackpack_hip_r4.so)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_b
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stress.py tests/test_gpu_model.py tests/test_gpu_backward.py -m gpu -x -q -k "mix or gather or sense or model or micro or interven or weighted" > $O/pytest_mix.log 2>&1; echo "exit $?" >> $O/pytest_mix.log
tail -5 $O/pytest_mix.log
python scripts/ab_kernels.py --libs r4,default --which mix,mixgather --batch 64,512 --reps 3 --out $O/ab_mix_small1024.jsonl | tail -12
python scripts/ab_kernels.py --libs r4,default --which mix --batch 128 --reps 2 --extra "--senses 64 --d 640" --out $O/ab_mix_mini_k64.jsonl | tail -4
python scripts/ab_kernels.py --libs r4,default --which mix --batch 64 --seq 4096 --reps 2 --extra "--dtype fp16" --out $O/ab_mix_small4096_fp16.jsonl | tail -4
