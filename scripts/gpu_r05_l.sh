#!/bin/bash
# round 5, run l: sense mix, waves 4-7 issue their DMA share in Y too (off X's critical path; one phase less of latency budget)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_l
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "mix or gather" > $O/pytest_mix.log 2>&1; echo "exit $?" >> $O/pytest_mix.log
tail -3 $O/pytest_mix.log
python scripts/ab_kernels.py --libs r5i,default --which mix,mixgather --batch 64,512 --reps 3 --out $O/ab_mix_small1024.jsonl | tail -9
python scripts/ab_kernels.py --libs r5i,default --which mixgather --batch 128 --reps 2 --extra "--senses 64 --d 640" --out $O/ab_mix_mini_k64.jsonl | tail -2
python scripts/ab_kernels.py --libs r5i,default --which mixgather --batch 64 --seq 4096 --reps 2 --extra "--dtype fp16" --out $O/ab_mix_small4096_fp16.jsonl | tail -2
