#!/bin/bash
# Round profile on the GPU box: kernel trace (stats) + PMC passes (HBM traffic, SQ) for one bench workload.
#   tools/profile_round.sh <tag> [workload] [extra bench flags]     (through gpurun; results under gpurun_out/prof_<tag>/)
TAG=${1:-r02}; WL=${2:-full}; shift; shift
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BENCH="python bench.py --no-cpu-baseline --no-extra --no-parity-probe --workload $WL $*"      # defaults: 100 timed steps after 1.5 s clock spin-up + 3 warm-up steps
PMCBENCH="python bench.py --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline --no-extra --no-parity-probe --workload $WL $*"    # counters are per launch; clocks do not matter
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $BENCH > $OUT/trace.log 2>&1
python tools/rocpd_stats.py $OUT/trace_results.db -100 > $OUT/kernel_stats.txt
# PMC: separate passes, kernel dispatches only (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $PMCBENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- $PMCBENCH > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $OUT -o sq1 -- $PMCBENCH > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq2 -- $PMCBENCH > $OUT/sq2.log 2>&1
# matrix pipe: instructions and busy cycles (zero for the shipped kernels: profiles/README.md round 3, FIR on the f32 MFMA)
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT -o mfma -- $PMCBENCH > $OUT/mfma.log 2>&1
python tools/pmc_summary.py $OUT/mfma_counter_collection.csv > $OUT/pmc_mfma.txt 2>/dev/null
python tools/pmc_summary.py $OUT/fetch_counter_collection.csv $OUT/write_counter_collection.csv $OUT/sq1_counter_collection.csv $OUT/sq2_counter_collection.csv > $OUT/pmc_summary.txt
grep '"metric"' $OUT/trace.log > $OUT/bench_under_trace.json
python tools/traffic_json.py $OUT $OUT/bench_under_trace.json > $OUT/traffic.json
rm -f $OUT/*_results.db $OUT/*.csv.bak
cat $OUT/kernel_stats.txt; cat $OUT/traffic.json
