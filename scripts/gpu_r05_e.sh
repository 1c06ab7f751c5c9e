#!/bin/bash
# round 5, run e: mix kernel = round-4 loop structure + streamed one-phase diagonal steps + LDS-staged next-sense operands:
# quick parity, A/B against round 4, phase timeline with lone / paired split
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=gpurun_out/r05_e
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "mix or gather" > $O/pytest_mix.log 2>&1; echo "exit $?" >> $O/pytest_mix.log
tail -3 $O/pytest_mix.log
python scripts/ab_kernels.py --libs r4,default --which mix,mixgather --batch 64,512 --reps 3 --out $O/ab_mix_small1024.jsonl | tail -9
python scripts/ab_kernels.py --libs r4,default --which mix --batch 128 --reps 2 --extra "--senses 64 --d 640" --out $O/ab_mix_mini_k64.jsonl | tail -2
L=$R/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip_mixprof.so python scripts/probes/mix_timeline/timeline2.py --batch 64 > $O/timeline_new_b64.json 2>$O/timeline_new.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_e/timeline_new_b64.json'))
for k,v in d.items():
    if k!='per wave': print(k, v)
for w,v in d['per wave'].items(): print(w, v)
PY
