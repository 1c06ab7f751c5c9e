# round-2 checkpoint s: full GPU suite on the working tree (summary line kept)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/t_r02_s_full.log 2>&1
grep -E "passed|failed|error" gpurun_out/t_r02_s_full.log | tail -5
grep -E "^FAILED|^ERROR" gpurun_out/t_r02_s_full.log | head -20
