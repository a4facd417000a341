#!/bin/bash
# PMC passes + kernel trace for the SURVEY.md 8f kernels: tools/pmc_post.sh <tag>   (through gpurun; results under gpurun_out/<tag>/)
TAG=${1:-post}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python tools/post_probe.py 10 > $OUT/trace.log 2>&1
python tools/rocpd_stats.py $OUT/trace_results.db -100 > $OUT/kernel_stats.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $OUT -o sq1 -- python tools/post_probe.py 3 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq2 -- python tools/post_probe.py 3 > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python tools/post_probe.py 3 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python tools/post_probe.py 3 > $OUT/write.log 2>&1
python tools/pmc_summary.py $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv $OUT/sq1_counter_collection.csv $OUT/sq2_counter_collection.csv | grep -v "ssdr_wf\|ssdr_fused\|ssdr_synth\|ssdr_audio" > $OUT/pmc_summary.txt
rm -f $OUT/*_results.db
cat $OUT/kernel_stats.txt | head -20; cat $OUT/pmc_summary.txt
