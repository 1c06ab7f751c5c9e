cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
L=$O/r03_ai_bwd_dq_stream_ab.jsonl; : > $L
for rep in 1 2 3; do
for lib in bwddq bwd0; do
  export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$lib.so
  for B in 32 64; do
  python scripts/bench_kernels.py --which bwd --batch $B --iters 20 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", \"shape\": \"causal S=1024 B=$B\", /" >> $L
  done
  python scripts/bench_kernels.py --which bwd --batch 8 --seq 8192 --iters 10 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", \"shape\": \"causal S=8192 B=8\", /" >> $L
done; done
python - <<'PY'
import json
for l in open('gpurun_out/r03_ai_bwd_dq_stream_ab.jsonl'):
    r=json.loads(l); print(r['lib'], r['shape'], round(r['ms'],4), round(r['tflops'],1))
PY
