#!/bin/bash
# round 4, run l: counters that have not been looked at -- VALU/MFMA co-execution, TA / LDS FIFO stalls, instruction fetch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=$R/gpurun_out/r04_l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; name=$2; shift 2; timeout 300 rocprofv3 --pmc "$@" -d $O/$tag/$name --output-format csv -- python $R/scripts/bench_kernels.py $ARGS > $O/$tag.$name.log 2>&1; }
for cfg in "s1024 --which flash,lse,mix --batch 256 --iters 3" "s4096nc --which flash --batch 16 --seq 4096 --noncausal --iters 3"; do
  set -- $cfg; tag=$1; shift; ARGS="$*"
  run $tag p1 SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU
  run $tag p2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM
  run $tag p3 SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
  run $tag p4 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_SMEM
  run $tag p5 SQ_LEVEL_WAVES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
  run $tag p6 GRBM_GUI_ACTIVE
  python $R/scripts/pmc_summary.py $O/$tag > $O/pmc_$tag.txt 2>&1
  rm -rf $O/$tag
done
grep -v "^    SQ_P" $O/pmc_s1024.txt | head -150
