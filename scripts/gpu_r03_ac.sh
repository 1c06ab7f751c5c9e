cd "$GRAFT_REPO_ROOT"
export BP_HIP_LIB=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip/libbackpack_hip_mixprof.so
for B in 16 64 128; do python scripts/probes/mix_timeline/timeline.py --batch $B 2>&1 | grep "^{" | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print({k: v for k, v in r.items() if 'job ticks' in k or 'kernel' in k or 'workgroups' in k or k == 'batch'})"; done
