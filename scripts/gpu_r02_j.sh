# round-2 checkpoint j: flash forward time versus key tiles per query-tile pass (fixed cost of a pass vs cost of a tile)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "128 512" "256 256" "512 128" "1024 64" "2048 32" "4096 16" "8192 8"; do
  set -- $cfg
  timeout 300 python scripts/bench_kernels.py --which flash --seq $1 --batch $2 --iters 30 --noncausal | sed 's/flash_fwd/flash_fwd noncausal/'
  timeout 300 python scripts/bench_kernels.py --which flash --seq $1 --batch $2 --iters 30
done
done > gpurun_out/r02_j_flash_seq_sweep.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_j_flash_seq_sweep.log
