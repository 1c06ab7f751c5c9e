mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu --timeout 600 2>&1 | tail -3 > gpurun_out/t8.log
timeout 900 python bench.py > gpurun_out/bench_r01_b.log 2>&1
python scripts/bench_kernels.py --which flash,lse,mix,alpha >> gpurun_out/kb8.log 2>&1
python scripts/bench_kernels.py --which flash,lse,mix --seq 4096 --batch 8 --dtype fp16 >> gpurun_out/kb8.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_b.log 2>&1
