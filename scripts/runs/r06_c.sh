#!/bin/bash
# round 6, run c: 16-byte epilogue stores of the flash forward alone (A/B + bits); sense-mix ticket order at S = 4096 (group-major
# against heaviest-first: time and FETCH_SIZE); the MFMA-only stream with its clock; bench lines of the wide-sense workloads;
# the three tests that failed in run b
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests fixed since run b"
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_abi.py -m gpu -q -k "stochastic or wide_senses_backward or abi or exports" 2>&1 | tail -6 | tee gpurun_out/r06_c_pytest_fixed.txt
echo "== t21 bits"; timeout 900 python scripts/flash_variant_check.py --libs default,t21 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r06_c_t21_bits.txt
echo "== t21 A/B"
timeout 900 python scripts/ab_kernels.py --libs default+BP_BENCH_FIXED_LEN=1,t21+BP_BENCH_FIXED_LEN=1 --which flash --batch 64,256,2048 --reps 4 --out gpurun_out/r06_c_ab_flash_wide_store.jsonl 2>&1 | grep -v amdgpu.ids | tail -8
timeout 600 python scripts/ab_kernels.py --libs default+BP_BENCH_FIXED_LEN=1,t21+BP_BENCH_FIXED_LEN=1 --which flash --batch 16,64 --seq 4096 --reps 3 --extra "--dtype fp16" 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r06_c_ab_flash_wide_store_4096.txt
echo "== mix order at S=4096"
timeout 900 python scripts/ab_kernels.py --libs default,mixgm --which mix,mixgather --batch 64 --seq 4096 --reps 3 --iters 5 --extra "--dtype fp16" --out gpurun_out/r06_c_ab_mix_order_4096.jsonl 2>&1 | grep -v amdgpu.ids | tail -6
timeout 600 python scripts/ab_kernels.py --libs default,mixgm --which mix --batch 512 --reps 2 --iters 5 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r06_c_ab_mix_order_1024.txt
for lib in default mixgm; do
  if [ $lib = default ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_$lib.so; fi
  out=$PWD/gpurun_out/pmc_r06_c_mix4096_$lib; mkdir -p $out; root=$PWD
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $out/tcc1 --output-format csv -- python $root/scripts/bench_kernels.py --which mix --batch 64 --seq 4096 --dtype fp16 --iters 3 > $out/tcc1.log 2>&1 )
  ( cd /tmp && timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $out/tcc2 --output-format csv -- python $root/scripts/bench_kernels.py --which mix --batch 64 --seq 4096 --dtype fp16 --iters 3 > $out/tcc2.log 2>&1 )
  python scripts/pmc_summary.py $out 2>/dev/null | grep -A4 "sense_mix_dma" | head -12 | tee gpurun_out/r06_c_pmc_mix4096_$lib.txt
done
unset BP_HIP_LIB
echo "== mfma stream clock"
timeout 300 python scripts/mfma_stream_clock.py --seconds 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c_mfma_stream_clock.jsonl
echo "== wide-sense bench lines"
for w in mini-k4-1024 mini-k1-1024; do
  timeout 900 python bench.py --workload $w --steps 10 --warmup 3 --no-clock-probe > gpurun_out/r06_c_bench_$w.json 2> gpurun_out/r06_c_bench_$w.err; echo "$w rc=$?"
  python - $w <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r06_c_bench_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'content_per_position', 'content_cached_table')}, d['config']['batch_per_gpu'])
    for r in d['kernels']: print(r['kernel'], r['avg_ms'], r['mfma_frac'], r['launches_per_step'])
except Exception as e:
    print('no line', e); print(open('gpurun_out/r06_c_bench_%s.err' % sys.argv[1]).read()[-1500:])
PY
done
