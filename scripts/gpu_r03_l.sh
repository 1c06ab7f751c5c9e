cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_nopipe.so
: > $O/r03_l_bwd_seq_sweep.jsonl
for cfg in "64 1024" "32 2048" "16 4096" "8 8192"; do set -- $cfg
  python scripts/bench_kernels.py --which flash,bwd --batch $1 --seq $2 --iters 10 2>/dev/null | grep "^{" >> $O/r03_l_bwd_seq_sweep.jsonl
  python scripts/bench_kernels.py --which flash,bwd --batch $1 --seq $2 --iters 10 --noncausal 2>/dev/null | grep "^{" | sed 's/"kernel": "/"kernel": "noncausal /' >> $O/r03_l_bwd_seq_sweep.jsonl
done
cat $O/r03_l_bwd_seq_sweep.jsonl
