mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "sense or mix or alpha" --timeout 600 2>&1 | tail -15 > gpurun_out/t3.log
echo "== default (dma mix)" > gpurun_out/kb2.log; python scripts/bench_kernels.py --which flash,mix,lse >> gpurun_out/kb2.log 2>&1
echo "== staged mix" >> gpurun_out/kb2.log; BP_MIX_IMPL=staged python scripts/bench_kernels.py --which mix >> gpurun_out/kb2.log 2>&1
echo "== vgprform lib" >> gpurun_out/kb2.log; BP_HIP_LIB=$PWD/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_vgprform.so python scripts/bench_kernels.py --which flash,mix,lse >> gpurun_out/kb2.log 2>&1
BP_HIP_LIB=$PWD/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_vgprform.so timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout 600 2>&1 | tail -5 >> gpurun_out/t3.log
