#!/bin/bash
# round 4, run aa: PMC passes on the shipped kernels at the batch the bench now picks (2048), mix in its gathering form
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out/r04_aa
bash scripts/gpu_pmc.sh r04_aa_b2048 --which flash,lse,mixgather --batch 2048 --iters 3; cp gpurun_out/pmc_r04_aa_b2048/summary.txt gpurun_out/r04_aa/pmc_small_b2048.txt
bash scripts/gpu_pmc.sh r04_aa_b64 --which mixgather,mix --batch 64 --iters 5; cp gpurun_out/pmc_r04_aa_b64/summary.txt gpurun_out/r04_aa/pmc_mix_gather_vs_dense_b64.txt
rm -rf gpurun_out/pmc_r04_aa_b2048/*/ gpurun_out/pmc_r04_aa_b64/*/
cd $R
grep -E "^bp::|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS" gpurun_out/r04_aa/pmc_small_b2048.txt | cut -c1-160
grep -E "^bp::|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS" gpurun_out/r04_aa/pmc_mix_gather_vs_dense_b64.txt | cut -c1-160
tail -3 gpurun_out/pmc_r04_aa_b2048/tcc1.log
