# round-3 checkpoint a: retry-branch tests, rewritten S=4096 tests, dropout tests (rescale change); PMC evidence for
# the flash backward kernels (none existed in round 2)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_retry.py tests/test_gpu_kernels.py tests/test_gpu_dropout.py -q -m gpu --timeout 900 -x -k "retry or 4096 or dropout or replay" > $O/t_r03_a_full.log 2>&1
tail -30 $O/t_r03_a_full.log > $O/t_r03_a.log
python scripts/bench_kernels.py --which flash,bwd --batch 64 --iters 10 > $O/r03_a_bwd_base.jsonl 2>&1
python scripts/bench_kernels.py --which bwd --batch 32 --iters 10 >> $O/r03_a_bwd_base.jsonl 2>&1
bash scripts/gpu_pmc.sh r03_a_bwd_b64 --which bwd --batch 64 --iters 3
cp $O/pmc_r03_a_bwd_b64/summary.txt $O/r03_a_pmc_flash_bwd_b64.txt
cat $O/t_r03_a.log; cat $O/r03_a_bwd_base.jsonl; cat $O/r03_a_pmc_flash_bwd_b64.txt
