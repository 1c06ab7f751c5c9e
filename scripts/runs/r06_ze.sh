#!/bin/bash
# round 6, run ze: the round-5 tree (9033c84, checked out into _r05_tree/ for this run only) and the final tree on ONE box,
# interleaved: the default bench command, the Mini k = 64 and the few-sense workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
R5=$PWD/_r05_tree
run() {  # tree tag args...
  tree=$1; tag=$2; shift 2
  ( cd $tree && timeout 900 python bench.py --no-cpu-baseline "$@" 2> /dev/null | tail -n 1 ) > gpurun_out/r06_ze_${tag}.json
  python - "$tag" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r06_ze_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
r = d.get('roofline', {})
print(sys.argv[1], d['value'], d['ms_per_step'], (d.get('content_per_position') or {}).get('value'), r.get('kernel'), r.get('frac'), r.get('avg_launch_ms'))
PY
}
for rep in 1 2; do
  run $R5 r05_default_$rep --steps 10 --warmup 3
  run $PWD r06_default_$rep --steps 10 --warmup 3
done
run $R5 r05_mini_k64 --workload mini-k64-1024 --steps 10 --warmup 3
run $PWD r06_mini_k64 --workload mini-k64-1024 --steps 10 --warmup 3
run $R5 r05_small4096 --workload small-4096-fp16 --steps 10 --warmup 3
run $PWD r06_small4096 --workload small-4096-fp16 --steps 10 --warmup 3
