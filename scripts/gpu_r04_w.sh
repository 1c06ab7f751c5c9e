#!/bin/bash
# round 4, run w: kernel breakdown of the forward step with the deduplicated content network
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=gpurun_out/r04_w
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -- python $R/bench.py --no-cpu-baseline --batch 1664 --steps 5 --warmup 2 > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err)
db=$(ls $O/prof/*/*_results.db 2>/dev/null | head -1); python scripts/rocprof_summary.py $db $O/kernel_stats_small1024_b1664_dedup.txt | head -24 | cut -c1-250; rm -rf $O/prof
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_w/bench_under_rocprof.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('content_per_position'))
PY
