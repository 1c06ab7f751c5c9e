# round-2 checkpoint x: race / determinism stress of the ring kernels
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for rep in 1 2; do timeout 900 python -m pytest tests/test_gpu_stress.py -q -m gpu --timeout 600 2>&1 | tail -4; done > gpurun_out/t_r02_x.log 2>&1
cat gpurun_out/t_r02_x.log
