cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
for m in fp32 view contig einsum; do HIP_LAUNCH_BLOCKING=1 timeout 120 python scripts/debug/r03_blas_fault.py $m > $O/r03_h.log 2>&1; echo "== $m: $(grep -c 'Memory access' $O/r03_h.log) faults; $(grep 'ok' $O/r03_h.log | tr '\n' ';')"; done
timeout 1500 python -m pytest tests/test_gpu_stress.py tests/test_gpu_model.py tests/test_gpu_kernels.py -q -m gpu -x > $O/r03_h_t.log 2>&1; tail -4 $O/r03_h_t.log | cut -c1-300
