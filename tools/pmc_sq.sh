#!/bin/bash
# the two SQ counter passes only, for one bench workload: tools/pmc_sq.sh <tag> [workload] [bench flags]   (through gpurun)
TAG=${1:-sq}; WL=${2:-full}; shift; shift
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PMCBENCH="python bench.py --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline --no-extra --no-parity-probe --workload $WL $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $OUT -o sq1 -- $PMCBENCH > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq2 -- $PMCBENCH > $OUT/sq2.log 2>&1
python tools/pmc_summary.py $OUT/sq1_counter_collection.csv $OUT/sq2_counter_collection.csv | grep -v synth > $OUT/pmc_summary.txt
rm -f $OUT/*_results.db
cat $OUT/pmc_summary.txt
