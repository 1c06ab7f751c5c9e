mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "add_layer_norm" --timeout 300 -x 2>&1 | tail -40 > gpurun_out/dbg.log
