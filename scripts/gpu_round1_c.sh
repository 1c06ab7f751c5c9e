mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu --timeout 600 2>&1 | tail -15 > gpurun_out/t4.log
echo "== default (dma)" > gpurun_out/kb3.log; python scripts/bench_kernels.py --which flash,mix,lse,alpha >> gpurun_out/kb3.log 2>&1
echo "== staged flash" >> gpurun_out/kb3.log; BP_FLASH_IMPL=staged python scripts/bench_kernels.py --which flash,lse >> gpurun_out/kb3.log 2>&1
echo "== seq 4096 fp16 b8" >> gpurun_out/kb3.log; python scripts/bench_kernels.py --which flash,mix,lse --seq 4096 --batch 8 --dtype fp16 >> gpurun_out/kb3.log 2>&1
echo "== mini k64" >> gpurun_out/kb3.log; python scripts/bench_kernels.py --which flash,mix,lse --heads 8 --headdim 80 --senses 64 --d 640 --batch 32 >> gpurun_out/kb3.log 2>&1
