# round-2 checkpoint y: flash forward time versus resident workgroups per CU (unused dynamic LDS caps the occupancy: 4 (shipped), 3, 2, 1)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip
( for rep in 1 2; do for v in "" _occ3 _occ2 _occ1; do
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash --seq 1024 --batch 256 --iters 20 | sed "s/flash_fwd/flash_fwd$v/"
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/bench_kernels.py --which flash --seq 4096 --batch 16 --iters 20 --noncausal | sed "s/flash_fwd/flash_fwd$v noncausal/"
done; done ) > gpurun_out/r02_y_flash_occupancy.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_y_flash_occupancy.log
