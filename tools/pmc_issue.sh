#!/bin/bash
# Issue / stall breakdown of one bench workload's kernels (round 5): which unit of the CU is saturated?
#   tools/pmc_issue.sh <tag> [workload] [bench flags]      (through gpurun; results under gpurun_out/issue_<tag>/)
# Counter passes only (no trace domains); the same passes on tools/ubench/pmc_calib, whose vector-ALU load is known by construction,
# give the scale the kernel's counters are read against.  SSDR_LIB_PATH selects an ablated build for the per-phase passes.
TAG=${1:-issue}; WL=${2:-full}; shift; shift
OUT=gpurun_out/issue_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PMCBENCH="python bench.py --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline --no-extra --no-parity-probe --workload $WL $*"
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_SALU"
P4="SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
P5="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVES SQ_CYCLES"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT -o p$i -- $PMCBENCH > $OUT/p$i.log 2>&1 || echo "pass $i failed (rc $?)" >> $OUT/failed.txt
done
if [ -z "$SSDR_ISSUE_NO_CALIB" ]; then
  for i in 1 2; do
    P=$P1; [ $i = 2 ] && P=$P2
    timeout 300 rocprofv3 --pmc $P --output-format csv -d $OUT -o calib$i -- tools/ubench/pmc_calib > $OUT/calib$i.log 2>&1 || echo "calib pass $i failed" >> $OUT/failed.txt
  done
fi
rm -f $OUT/*_results.db
python tools/pmc_summary.py $OUT/p*_counter_collection.csv 2>/dev/null | grep -v synth > $OUT/pmc_raw.txt
[ -z "$SSDR_ISSUE_NO_CALIB" ] && python tools/pmc_summary.py $OUT/calib*_counter_collection.csv > $OUT/pmc_calib_raw.txt 2>/dev/null
cat $OUT/failed.txt 2>/dev/null; tail -40 $OUT/pmc_raw.txt
