cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_world2.py -q -m gpu -x -s > gpurun_out/r03_z_world2.log 2>&1
tail -30 gpurun_out/r03_z_world2.log | cut -c1-400
