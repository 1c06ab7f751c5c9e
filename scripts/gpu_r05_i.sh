#!/bin/bash
# round 5, run i: partial column chunks skip their dead blocks in X too: parity + A/B (Mini d = 640, Micro d = 384)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_i
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "mix or gather" > $O/pytest_mix.log 2>&1; echo "exit $?" >> $O/pytest_mix.log
tail -3 $O/pytest_mix.log
python scripts/ab_kernels.py --libs r5e,default --which mix,mixgather --batch 128 --reps 3 --extra "--senses 64 --d 640" --out $O/ab_mix_mini_k64.jsonl | tail -4
python scripts/ab_kernels.py --libs r5e,default --which mix --batch 256 --reps 3 --extra "--senses 16 --d 384" --out $O/ab_mix_micro_d384.jsonl | tail -2
python scripts/ab_kernels.py --libs r5e,default --which mixgather --batch 512 --reps 2 --out $O/ab_mix_small.jsonl | tail -2
