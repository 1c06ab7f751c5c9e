cd "$GRAFT_REPO_ROOT"
for o in default heavy lockstep; do echo "order=$o"; BP_MIX_ORDER=$o timeout 300 python scripts/bench_kernels.py --which mix 2>&1 | grep -v amdgpu; done
echo "s=2048 b=16"; for o in default heavy; do BP_MIX_ORDER=$o timeout 300 python scripts/bench_kernels.py --which mix --seq 2048 --batch 16 2>&1 | grep -v amdgpu; done
echo "s=512 b=64"; for o in default heavy; do BP_MIX_ORDER=$o timeout 300 python scripts/bench_kernels.py --which mix --seq 512 --batch 64 2>&1 | grep -v amdgpu; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "mix or alpha" 2>&1 | tail -2
