# usage: bash scripts/gpu_pmc.sh <tag> <bench_kernels args...>   (run on the GPU box)
# Collects PMC counters for the micro-benchmark in separate passes (never with tracing domains).
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" -d $out/$name --output-format csv -- python $GRAFT_REPO_ROOT/scripts/bench_kernels.py $ARGS > $out/$name.log 2>&1; }
ARGS="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python scripts/pmc_summary.py $out > $out/summary.txt 2>&1
