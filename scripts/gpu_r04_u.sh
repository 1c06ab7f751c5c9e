#!/bin/bash
# round 4, run u: kernel breakdown of the training step (config 3, batch 32) under rocprofv3
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=gpurun_out/r04_u
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/scripts/bench_train_step.py --batch 32 --steps 5 --warmup 2 > $R/$O/train_under_rocprof.jsonl 2> $R/$O/train_under_rocprof.err)
db=$(ls $O/prof/*/*_results.db 2>/dev/null | head -1); python scripts/rocprof_summary.py $db $O/kernel_stats_train_small1024_b32.txt | head -45 | cut -c1-230; rm -rf $O/prof
cat $O/train_under_rocprof.jsonl | cut -c1-300
