#!/bin/bash
# one-read kernels against the two stages side by side at small and medium batch sizes (ssdr_run_chain's defaults have channel-count floors):
#   tools/ab_small_batches.sh "<workloads>" "<channel counts>"
W="${1:-am_narrow full}"; N="${2:-64 256 1024 4096 16384 24576 32768 49152}"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for wl in $W; do for ch in $N; do for f in 0 3; do
  printf "%-10s %6d ch fused=%d  " $wl $ch $f
  python bench.py --workload $wl --channels $ch --fused $f --steps 200 --warmup 5 --spinup 0.5 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'ms', d['config']['chain'][:30])"
done; done; done
