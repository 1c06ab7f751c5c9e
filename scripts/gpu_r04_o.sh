#!/bin/bash
# round 4, run o: pre-scaled Q + exponent-origin accumulators (BP_FWD_PRESCALE), with and without the lean edge tiles
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04_o
export TMPDIR=/tmp
LIBDIR=$PWD/backpacks-flash-attn_amd/bp_hip
for v in default lean3 pre pre3; do
  if [ $v = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$LIBDIR/libbackpack_hip_$v.so; fi
  timeout 300 python scripts/debug/r04_fwd_accuracy.py >> gpurun_out/r04_o/accuracy.jsonl 2> gpurun_out/r04_o/acc_$v.err
done
unset BP_HIP_LIB
cat gpurun_out/r04_o/accuracy.jsonl
for v in lean3 pre3; do
BP_HIP_LIB=$LIBDIR/libbackpack_hip_$v.so timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py tests/test_gpu_stress.py tests/test_gpu_dropout.py -m gpu -q -k "flash or retry or lse or attn" > gpurun_out/r04_o/parity_$v.log 2>&1
tail -3 gpurun_out/r04_o/parity_$v.log
done
timeout 1200 python scripts/ab_kernels.py --libs default,lean3,pre,pre3 --which flash,lse --batch 64,256 --reps 3 --out gpurun_out/r04_o/ab.jsonl > gpurun_out/r04_o/ab.log 2>&1
tail -18 gpurun_out/r04_o/ab.log
timeout 600 python scripts/ab_kernels.py --libs default,lean3,pre,pre3 --which flash --batch 16 --seq 4096 --reps 2 > gpurun_out/r04_o/ab_4k.log 2>&1
tail -5 gpurun_out/r04_o/ab_4k.log
