# round-2 checkpoint c: dropout tests, full suite, PMC counters of the new flash forward
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_dropout.py -q -m gpu --timeout 600 2>&1 | tail -40 > gpurun_out/t_r02_c_dropout.log
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 --ignore tests/test_gpu_dropout.py 2>&1 | tail -25 > gpurun_out/t_r02_c.log
bash scripts/gpu_pmc.sh r02_c_flash --which flash --batch 64 --iters 10
bash scripts/gpu_pmc.sh r02_c_flash_nc --which flash --batch 16 --seq 4096 --noncausal --iters 10
cat gpurun_out/t_r02_c_dropout.log; cat gpurun_out/t_r02_c.log; cat gpurun_out/pmc_r02_c_flash/summary.txt gpurun_out/pmc_r02_c_flash_nc/summary.txt
