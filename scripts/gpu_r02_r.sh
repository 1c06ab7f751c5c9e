# round-2 checkpoint r: flash forward "ping-pong" kernel (8 waves, two query tiles alternating matrix / softmax phases) vs the shipped kernel
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip
BP_HIP_LIB=$L/libbackpack_hip_pp4.so timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout 900 -k "flash" -x 2>&1 | tail -15 > gpurun_out/t_r02_r.log
for rep in 1 2 3; do
for v in "" _pp4 _pp2; do
  for cfg in "1024 64" "1024 256" "4096 16"; do
    set -- $cfg
    BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash --seq $1 --batch $2 --iters 30 | sed "s/flash_fwd/flash_fwd$v/"
  done
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash --seq 4096 --batch 16 --iters 30 --noncausal | sed "s/flash_fwd/flash_fwd$v noncausal/"
done
done > gpurun_out/r02_r_flash_pp.log 2>&1
cat gpurun_out/t_r02_r.log; grep -v amdgpu.ids gpurun_out/r02_r_flash_pp.log
