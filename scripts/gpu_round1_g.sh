mkdir -p gpurun_out; rm -f gpurun_out/kb7.log
for v in "" _nolds _noldsexp _ldsonly; do
echo "== variant '$v'" >> gpurun_out/kb7.log
BP_HIP_LIB=$PWD/backpacks-flash-attn_amd/bp_hip/libbackpack_hip$v.so python scripts/bench_kernels.py --which flash >> gpurun_out/kb7.log 2>&1
done
