mkdir -p gpurun_out; rm -f gpurun_out/kb11.log
BP_FLASH_IMPL=dma2 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "flash" --timeout 600 2>&1 | tail -4 > gpurun_out/t12.log
echo "== dma (default)" >> gpurun_out/kb11.log; python scripts/bench_kernels.py --which flash >> gpurun_out/kb11.log 2>&1
echo "== dma2 nwave4" >> gpurun_out/kb11.log; BP_FLASH_IMPL=dma2 python scripts/bench_kernels.py --which flash >> gpurun_out/kb11.log 2>&1
BP_FLASH_IMPL=dma2 python scripts/bench_kernels.py --which flash --noncausal --seq 4096 --batch 8 >> gpurun_out/kb11.log 2>&1
echo "== dma2 nwave2" >> gpurun_out/kb11.log; BP_HIP_LIB=$PWD/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_nw2.so BP_FLASH_IMPL=dma2 python scripts/bench_kernels.py --which flash >> gpurun_out/kb11.log 2>&1
BP_HIP_LIB=$PWD/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_nw2.so BP_FLASH_IMPL=dma2 python scripts/bench_kernels.py --which flash --noncausal --seq 4096 --batch 8 >> gpurun_out/kb11.log 2>&1
