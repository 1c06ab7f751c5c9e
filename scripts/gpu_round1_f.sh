mkdir -p gpurun_out; rm -f gpurun_out/kb6.log
for v in "" _s2 _s4 _s2w4 _prio; do
echo "== variant '$v'" >> gpurun_out/kb6.log
BP_HIP_LIB=$PWD/backpacks-flash-attn_amd/bp_hip/libbackpack_hip$v.so python scripts/bench_kernels.py --which flash,lse >> gpurun_out/kb6.log 2>&1
BP_HIP_LIB=$PWD/backpacks-flash-attn_amd/bp_hip/libbackpack_hip$v.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "flash_fwd_fixed_len and 64-True-dtype0 or varlen" --timeout 300 2>&1 | tail -1 >> gpurun_out/kb6.log
done
