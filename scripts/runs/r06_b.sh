#!/bin/bash
# round 6, run b: the persistent flash forward (-DBP_FWD_PERSIST=1 build) -- bits against the shipped kernel, the flash tests
# on it, same-box A/B timing, the no-math what-if pair; plus the new model tests on the default library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_persist.so
echo "== bits"; timeout 900 python scripts/flash_variant_check.py --libs default,persist 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/r06_b_bits.txt
echo "== flash tests on the persistent build"
BP_HIP_LIB=$P timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py tests/test_gpu_properties.py tests/test_gpu_dropout.py -m gpu -q -x -k "flash or retry or causal or attention or trunk or lse or whole or model" 2>&1 | tail -8 | tee gpurun_out/r06_b_pytest_persist.txt
echo "== A/B"
timeout 900 python scripts/ab_kernels.py --libs default+BP_BENCH_FIXED_LEN=1,persist+BP_BENCH_FIXED_LEN=1 --which flash,lse --batch 64,256,2048 --reps 3 --out gpurun_out/r06_b_ab_flash_persist.jsonl 2>&1 | grep -v amdgpu.ids | tail -16
timeout 600 python scripts/ab_kernels.py --libs default+BP_BENCH_FIXED_LEN=1,persist+BP_BENCH_FIXED_LEN=1,default,persist --which flash --batch 16,64 --seq 4096 --reps 2 --extra "--dtype fp16" 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r06_b_ab_flash_persist_4096.txt
echo "== what-if (no MFMA, no softmax)"
timeout 600 python scripts/ab_kernels.py --libs wi3+BP_BENCH_FIXED_LEN=1,pwi3+BP_BENCH_FIXED_LEN=1 --which flash --batch 256 --reps 3 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r06_b_whatif.txt
echo "== new tests, default library"
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_kernels.py -m gpu -q -k "stochastic or chunks or sense_table_follows or bench or wide or few_sense" 2>&1 | tail -25 | tee gpurun_out/r06_b_pytest_new.txt
