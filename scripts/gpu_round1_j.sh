mkdir -p gpurun_out; rm -f gpurun_out/kb9.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "sense or mix or alpha" --timeout 300 2>&1 | tail -2 > gpurun_out/t10.log
echo "== mini k64 (padded d_k)" >> gpurun_out/kb9.log; python scripts/bench_kernels.py --which flash,mix,lse --heads 8 --headdim 80 --senses 64 --d 640 --batch 32 >> gpurun_out/kb9.log 2>&1
echo "== tunableop" >> gpurun_out/kb9.log
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunable.csv timeout 1200 python bench.py --no-cpu-baseline --warmup 3 --steps 10 > gpurun_out/bench_tunable.log 2>&1
tail -1 gpurun_out/bench_tunable.log | cut -c1-200 >> gpurun_out/kb9.log
cp /tmp/tunable*.csv gpurun_out/ 2>/dev/null
