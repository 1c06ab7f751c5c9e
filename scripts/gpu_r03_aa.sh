cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out/r03_aa_mix_timeline.jsonl
: > $O
export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_mixprof.so
for B in 4 16 64 128; do
python scripts/probes/mix_timeline/timeline.py --batch $B 2>&1 | grep "^{" >> $O
done
cat $O
