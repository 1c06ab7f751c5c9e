#!/bin/bash
# round 6, run j: which wide LSE variants repeat bit for bit (probe: 8 launches, elements that vary)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
L=$PWD/backpacks-flash-attn_amd/bp_hip
for v in "" _w160r2 _widestaged; do echo "== lib$v"; BP_HIP_LIB=$L/libbackpack_hip$v.so timeout 300 python scripts/debug/wide_lse_probe.py 2>&1 | grep -v amdgpu.ids | head -4; done | tee gpurun_out/r06_j_probe.txt
