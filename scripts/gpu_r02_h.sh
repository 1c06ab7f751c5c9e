# round-2 checkpoint h: software-pipelined flash forward (2 waves per SIMD); variants: 3-slot ring (default), 4-slot ring
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropout.py tests/test_gpu_configs.py -q -m gpu --timeout 900 -x 2>&1 | tail -15 > gpurun_out/t_r02_h.log
for v in "" _s4; do
  for rep in 1 2; do
    BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash,lse --batch 64 --iters 30 | sed "s/\"kernel\": \"/\"kernel\": \"v$v:/"
  done
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash --batch 256 --iters 20 | sed "s/\"kernel\": \"/\"kernel\": \"v$v:/"
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash --batch 16 --seq 4096 --iters 20 --noncausal | sed "s/\"kernel\": \"/\"kernel\": \"v$v:noncausal:/"
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash --batch 16 --seq 4096 --iters 20 | sed "s/\"kernel\": \"/\"kernel\": \"v$v:/"
done > gpurun_out/r02_h_flash.log 2>&1
cat gpurun_out/t_r02_h.log; grep -v amdgpu.ids gpurun_out/r02_h_flash.log
