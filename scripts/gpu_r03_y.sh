cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out/r03_y_mix_layout.jsonl
: > $O
for rep in 1 2 3; do
for B in 16 64 128; do
for L in bskd bksd; do
python scripts/bench_kernels.py --which mix --batch $B --iters 20 --content-layout $L 2>/dev/null | grep "^{" >> $O
done; done; done
cat $O | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['layout'], r.get('batch'), round(r['ms'],4), round(r['tflops'],1))
"
