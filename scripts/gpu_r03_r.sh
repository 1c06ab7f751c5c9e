# round-3 checkpoint r: the debug cases, then the full GPU suite on the backward with overlapped prologue / 16-byte stores
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python scripts/debug/r03_bwd_case.py > $O/r03_r_case.log 2>&1; grep -v amdgpu $O/r03_r_case.log | grep -v "^  File\|Extension" | cut -c1-200 | head -14
timeout 3000 python -m pytest tests -q -m gpu --timeout 900 > $O/t_r03_r_full.log 2>&1
grep -E "passed|failed|error" $O/t_r03_r_full.log | tail -3 > $O/t_r03_r.log
grep -E "^FAILED|^ERROR" $O/t_r03_r_full.log | head -20 >> $O/t_r03_r.log
cat $O/t_r03_r.log
python scripts/bench_kernels.py --which bwd --batch 64 --iters 20 2>/dev/null | grep "^{"
python scripts/bench_kernels.py --which bwd --batch 32 --iters 20 2>/dev/null | grep "^{"
