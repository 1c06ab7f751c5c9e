# round-2 checkpoint p: non-temporal hints A/B (same box): xentropy backward store, sense-mix dC store, sense-mix content DMA
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout 600 -k "alpha or probs" 2>&1 | tail -3 > gpurun_out/t_r02_p.log
( for rep in 1 2 3; do for v in "" _nt; do
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which mix,mixbwd,xent --batch 64 --iters 20 | sed "s/\"kernel\": \"/\"kernel\": \"v$v:/"
done; done ) > gpurun_out/r02_p_nt.log 2>&1
cat gpurun_out/t_r02_p.log; grep -v amdgpu.ids gpurun_out/r02_p_nt.log
