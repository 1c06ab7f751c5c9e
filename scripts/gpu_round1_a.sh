mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu --timeout 600 2>&1 | tail -40 > gpurun_out/t2.log
for B in 16 64 128; do timeout 600 python bench.py --steps 5 --warmup 2 --batch $B --no-cpu-baseline >> gpurun_out/bench_sweep.log 2>&1; done
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_r01 -name "*stats*" | head; du -sh gpurun_out/prof_r01
