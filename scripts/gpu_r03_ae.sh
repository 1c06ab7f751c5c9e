cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_dropout.py -q -m gpu -x > $O/r03_ae_tests.log 2>&1; tail -3 $O/r03_ae_tests.log | cut -c1-300
L=$O/r03_ae_bwd_stream_ab.jsonl; : > $L
for rep in 1 2 3; do
for lib in default bwdold bwds2; do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$lib.so; fi
  for B in 32 64; do
  python scripts/bench_kernels.py --which bwd --batch $B --iters 20 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", \"shape\": \"causal S=1024 B=$B\", /" >> $L
  done
  python scripts/bench_kernels.py --which bwd --batch 8 --seq 8192 --iters 10 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", \"shape\": \"causal S=8192 B=8\", /" >> $L
done; done
for lib in default bwdold; do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$lib.so; fi
  python scripts/bench_train_step.py --batch 32 --steps 5 --warmup 2 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$lib\", /" >> $O/r03_ae_train_ab.jsonl
done
unset BP_HIP_LIB
python - <<'PY'
import json
for l in open('gpurun_out/r03_ae_bwd_stream_ab.jsonl'):
    r=json.loads(l); print(r['lib'], r['shape'], round(r['ms'],4), round(r['tflops'],1))
for l in open('gpurun_out/r03_ae_train_ab.jsonl'):
    r=json.loads(l); print(r['lib'], r['ms_per_step'], r['value'])
PY
