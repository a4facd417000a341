#!/bin/bash
# One GPU-box visit: runs the commands named on the command line, one log each.
#   gpurun -- 'bash tools/gpu_round.sh <tag> "<cmd>" "<cmd>" ...'   (logs under gpurun_out/<tag>/)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for cmd in "$@"; do
  i=$((i+1))
  echo "=== [$i] $cmd" | tee -a $OUT/log.txt
  timeout 900 bash -c "$cmd" > $OUT/cmd$i.txt 2>&1
  echo "rc=$?" | tee -a $OUT/log.txt
  tail -25 $OUT/cmd$i.txt
done
