#!/bin/bash
# One parameterised runner for everything a GPU call of this repository does (run ON the GPU box, e.g.
#   gpurun --timeout 1800 -- 'bash scripts/gpu_run.sh tests -k flash; bash scripts/gpu_run.sh bench r06_x').
# Outputs go to gpurun_out/ (scratch, merged back by gpurun); copy what should be judged into profiles/.
#
#   gpu_run.sh tests [pytest args]            the -m gpu suite (default: all of it, -x -q), tail to gpurun_out/<tag>_pytest.txt
#   gpu_run.sh smoke                          __graft_entry__.smoke()
#   gpu_run.sh bench <tag> [bench.py args]    one bench line -> gpurun_out/<tag>_bench.json (+ .err)
#   gpu_run.sh stats <tag> [bench.py args]    rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/<tag>_stats/
#   gpu_run.sh pmc <tag> <bench_kernels args> counter passes of a kernel micro-benchmark, ONE --pmc group per pass and never
#                                             together with tracing domains (MI355X_MICROARCH: HBM / rocprofv3 section)
#   gpu_run.sh ab <libs> [ab_kernels args]    same-box interleaved A/B of library builds (build_hip.py --variant NAME),
#                                             e.g. ab default,persist --which flash --batch 64,2048
#   gpu_run.sh kernels [bench_kernels args]   the per-kernel micro-benchmarks
# TAG defaults to "run"; set BP_HIP_LIB=<path> to run any step on a variant build.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cmd=${1:-help}; shift || true
case "$cmd" in
  tests)
    args=("$@"); [ ${#args[@]} -eq 0 ] && args=(-x -q)
    timeout ${BP_TIMEOUT:-3000} python -m pytest tests -m gpu "${args[@]}" 2>&1 | tail -n 15 | tee gpurun_out/${TAG:-run}_pytest.txt ;;
  smoke)
    timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 5 ;;
  bench)
    tag=${1:-run}; shift || true
    timeout ${BP_TIMEOUT:-1200} python bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
    echo "bench rc=$?"; tail -c 300 gpurun_out/${tag}_bench.err
    python - "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/%s_bench.json' % sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ('value', 'ms_per_step', 'content_per_position', 'content_cached_table')})
    print({k: v for k, v in d.get('roofline', {}).items() if k not in ('excluded_by_rule', 'sustained_probe', 'traffic_source')})
except Exception as e:   # noqa: BLE001
    print('no bench line:', e)
PY
    ;;
  stats)
    tag=${1:-run}; shift || true
    ( cd /tmp && timeout ${BP_TIMEOUT:-1500} rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/${tag}_stats" --output-format csv -- \
        python "$OLDPWD/bench.py" --no-cpu-baseline --no-clock-probe "$@" > "$OLDPWD/gpurun_out/${tag}_stats.log" 2>&1 )
    python scripts/rocprof_summary.py gpurun_out/${tag}_stats 2>/dev/null | head -n 25 | tee gpurun_out/${tag}_kernel_stats.txt ;;
  pmc)
    tag=${1:-run}; shift || true
    out=$PWD/gpurun_out/pmc_$tag; mkdir -p "$out"; ARGS="$*"; root=$PWD
    run() { name=$1; shift; ( cd /tmp && timeout 600 rocprofv3 --pmc "$@" -d "$out/$name" --output-format csv -- python "$root/scripts/bench_kernels.py" $ARGS > "$out/$name.log" 2>&1 ); }
    run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
    run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
    run tcc1 FETCH_SIZE
    run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
    python scripts/pmc_summary.py "$out" 2>/dev/null | tee gpurun_out/${tag}_pmc.txt ;;
  ab)
    libs=${1:-default}; shift || true
    timeout ${BP_TIMEOUT:-1200} python scripts/ab_kernels.py --libs "$libs" "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG:-run}_ab.txt | tail -n 20 ;;
  kernels)
    timeout ${BP_TIMEOUT:-900} python scripts/bench_kernels.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG:-run}_kernels.jsonl ;;
  *)
    sed -n '2,20p' "$0" ;;
esac
