#!/bin/bash
# round 5, run f: flash backward what-if builds (scripts/probes/flash_bwd_whatif) at the trunk shape and at S = 4096
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05_f
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
LIBS=default,bwdw1,bwdw32,bwdw2,bwdw4,bwdw6,bwdw8,bwdw16,bwdw38,bwdw54
python scripts/ab_kernels.py --libs $LIBS --which bwd --batch 64 --reps 3 --out $O/flash_bwd_whatif_causal_s1024_b64.jsonl | tail -12
python scripts/ab_kernels.py --libs $LIBS --which bwd --batch 256 --reps 2 --out $O/flash_bwd_whatif_causal_s1024_b256.jsonl | tail -12
python scripts/ab_kernels.py --libs $LIBS --which bwd --batch 16 --seq 4096 --reps 2 --out $O/flash_bwd_whatif_causal_s4096_b16.jsonl | tail -12
python scripts/ab_kernels.py --libs $LIBS --which bwd --batch 16 --seq 4096 --reps 2 --extra "--noncausal" --out $O/flash_bwd_whatif_noncausal_s4096_b16.jsonl | tail -12
