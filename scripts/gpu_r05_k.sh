#!/bin/bash
# round 5, run k: v_pk_fma_f32 (two scores per fused multiply-add) in the forward softmax paths: parity + same-box A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_k
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py -m gpu -x -q -k "flash_fwd or lse or alpha or mix or gather or retry or golden" > $O/pytest_fwd.log 2>&1; echo "exit $?" >> $O/pytest_fwd.log
tail -3 $O/pytest_fwd.log
python scripts/ab_kernels.py --libs r5i,default --which flash,lse,mixgather --batch 64,512 --reps 3 --out $O/ab_pkfma_small1024.jsonl | tail -13
python scripts/ab_kernels.py --libs r5i,default --which flash,lse,mixgather --batch 64 --seq 4096 --reps 2 --extra "--dtype fp16" --out $O/ab_pkfma_small4096_fp16.jsonl | tail -7
python scripts/ab_kernels.py --libs r5i,default --which flash,lse,mixgather --batch 128 --reps 2 --extra "--heads 8 --headdim 80 --senses 64 --d 640" --out $O/ab_pkfma_mini_k64.jsonl | tail -7
