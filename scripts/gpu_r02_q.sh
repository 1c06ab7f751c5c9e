# round-2 checkpoint q: fused add+LayerNorm with non-temporal loads / stores, measured inside the forward (same box)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/backpacks-flash-attn_amd/bp_hip
( for rep in 1 2 3; do for v in "" _ln1 _ln3 _ln7; do
  BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python bench.py --batch 64 --no-cpu-baseline --steps 10 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = {r['kernel']: r['avg_ms'] for r in d['kernels']}
        print(json.dumps({'variant': 'v$v', 'ms_per_step': d['ms_per_step'], 'ln_ms': k['add_layer_norm_kernel'], 'flash_ms': k['flash_fwd_kernel'], 'mix_ms': k['sense_mix_kernel']}))"
done; done ) > gpurun_out/r02_q_ln_nt.log 2>&1
grep -v amdgpu.ids gpurun_out/r02_q_ln_nt.log
