# round-2 checkpoint e: mix job order A/B (group-major default vs heaviest-first), flash forward after the speculation fix
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dropout.py -q -m gpu --timeout 600 -k "documented_function" 2>&1 | tail -5 > gpurun_out/t_r02_e_mask.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout 600 -k "flash or mix" 2>&1 | tail -5 > gpurun_out/t_r02_e_k.log
for rep in 1 2; do
for b in 16 64 128; do timeout 300 python scripts/bench_kernels.py --which mix --batch $b --iters 30; done
for b in 16 64 128; do BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_heavy.so timeout 300 python scripts/bench_kernels.py --which mix --batch $b --iters 30 | sed 's/sense_mix/sense_mix_heavy_first/'; done
done > gpurun_out/r02_e_mix.log 2>&1
for b in 64 256; do timeout 300 python scripts/bench_kernels.py --which flash,lse --batch $b --iters 30; done > gpurun_out/r02_e_flash.log 2>&1
timeout 300 python scripts/bench_kernels.py --which flash --batch 16 --seq 4096 --iters 20 --noncausal >> gpurun_out/r02_e_flash.log 2>&1
bash scripts/gpu_pmc.sh r02_e_mix --which mix --batch 64 --iters 10
cat gpurun_out/t_r02_e_mask.log gpurun_out/t_r02_e_k.log; grep -v amdgpu.ids gpurun_out/r02_e_mix.log gpurun_out/r02_e_flash.log; cat gpurun_out/pmc_r02_e_mix/summary.txt
