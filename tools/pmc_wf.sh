#!/bin/bash
# PMC passes for the waterfall kernel (each --pmc group in its own run; never with trace domains other than kernel)
# usage: tools/pmc_wf.sh <outdir-under-gpurun_out> [workload]
OUT=gpurun_out/$1; WL=${2:-wf}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p $OUT
run() { rocprofv3 --pmc "$@" --output-format csv -d $OUT -o p$N -- python bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$N.log 2>&1; N=$((N+1)); }
N=1
run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES
run SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run FETCH_SIZE GRBM_GUI_ACTIVE
run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python tools/pmc_summary.py $OUT/p?_counter_collection.csv | grep -v synth > $OUT/summary.txt
cat $OUT/summary.txt
