mkdir -p gpurun_out
for v in noperm nosaddr; do
echo "== $v" >> gpurun_out/dbg.log
BP_HIP_LIB=$PWD/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "flash_fwd_fixed_len and 128-64-True-dtype0 or varlen or determinism" --timeout 300 2>&1 | tail -4 >> gpurun_out/dbg.log
done
