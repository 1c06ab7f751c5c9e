#!/bin/bash
# round 5, run d: sense mix with every step in the two-phase form + next-sense operands staged through LDS + scalar gather
# indices: parity, same-box A/B against the round-4 kernel, phase timelines of both
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=gpurun_out/r05_d
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stress.py tests/test_gpu_model.py tests/test_gpu_backward.py -m gpu -x -q -k "mix or gather or sense or model or micro or weighted or interv" > $O/pytest_mix.log 2>&1; echo "exit $?" >> $O/pytest_mix.log
tail -4 $O/pytest_mix.log
df -h /tmp | tail -1
python scripts/ab_kernels.py --libs r4,default --which mix,mixgather --batch 64,512 --reps 3 --out $O/ab_mix_small1024.jsonl | tail -10
python scripts/ab_kernels.py --libs r4,default --which mix --batch 128 --reps 2 --extra "--senses 64 --d 640" --out $O/ab_mix_mini_k64.jsonl | tail -3
python scripts/ab_kernels.py --libs r4,default --which mix --batch 64 --seq 4096 --reps 2 --extra "--dtype fp16" --out $O/ab_mix_small4096_fp16.jsonl | tail -3
L=$R/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip_mixprof.so python scripts/probes/mix_timeline/timeline2.py --batch 64 > $O/timeline_new_b64.json 2>$O/timeline_new.err

cat $O/timeline_new_b64.json
