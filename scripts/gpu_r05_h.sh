#!/bin/bash
# round 5, run h: sense mix with the clean tiles of ALL senses in one two-phase sweep, then the diagonal tiles (one anti-phase
# fill / drain per job instead of per sense): parity, same-box A/B against the previous build (r5e) 
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_h
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stress.py tests/test_gpu_model.py tests/test_gpu_backward.py -m gpu -x -q -k "mix or gather or sense or model or micro or weighted or interv" > $O/pytest_mix.log 2>&1; echo "exit $?" >> $O/pytest_mix.log
tail -3 $O/pytest_mix.log
python scripts/ab_kernels.py --libs r5e,default --which mix,mixgather --batch 64,512 --reps 3 --out $O/ab_mix_small1024.jsonl | tail -9
python scripts/ab_kernels.py --libs r5e,default --which mixgather --batch 128 --reps 3 --extra "--senses 64 --d 640" --out $O/ab_mix_mini_k64.jsonl | tail -2
python scripts/ab_kernels.py --libs r5e,default --which mixgather --batch 64 --seq 4096 --reps 2 --extra "--dtype fp16" --out $O/ab_mix_small4096_fp16.jsonl | tail -2
python scripts/ab_kernels.py --libs r5e,default --which mixgather --batch 256 --seq 256 --reps 2 --out $O/ab_mix_small256.jsonl | tail -2
