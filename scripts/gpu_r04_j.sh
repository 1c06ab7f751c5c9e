#!/bin/bash
# round 4, run j: odd K pitch (176 / 208 / 240 bytes) for head dims 65..112: parity, A/B against the 256-byte pitch, LDS conflicts
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out/r04_j
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_retry.py tests/test_gpu_dropout.py tests/test_gpu_stress.py -m gpu -x -q -k "flash" > gpurun_out/r04_j/pytest_flash.log 2>&1; echo "flash tests: $(tail -1 gpurun_out/r04_j/pytest_flash.log)"
for hd in 80 96 112; do
timeout 900 python scripts/ab_kernels.py --libs default,evenpitch --which flash --batch 64,256 --reps 3 --extra="--heads 8 --headdim $hd" --out gpurun_out/r04_j/ab_flash_d$hd.jsonl > gpurun_out/r04_j/ab_d$hd.log 2>&1
echo "d=$hd"; tail -4 gpurun_out/r04_j/ab_d$hd.log
done
cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $R/gpurun_out/r04_j/pmc --output-format csv -- python $R/scripts/bench_kernels.py --which flash --batch 64 --heads 8 --headdim 80 --iters 3 > $R/gpurun_out/r04_j/pmc.log 2>&1
cd $R; python scripts/pmc_summary.py gpurun_out/r04_j/pmc > gpurun_out/r04_j/pmc_lds_d80.txt; cat gpurun_out/r04_j/pmc_lds_d80.txt; rm -rf gpurun_out/r04_j/pmc
timeout 900 python bench.py --workload mini-k64-1024 --no-cpu-baseline > gpurun_out/r04_j/bench_mini.json 2> gpurun_out/r04_j/bench_mini.err
python -c "
import json;d=json.loads(open('gpurun_out/r04_j/bench_mini.json').read().strip().splitlines()[-1]);print('mini', d['value'], d['ms_per_step'], d['config']['batch_per_gpu'], [(k['kernel'][:14],k['avg_ms'],k['mfma_frac']) for k in d['kernels']])"
