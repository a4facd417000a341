#!/bin/bash
# general-mode fused kernel (--fused 3) against ssdr_run_chain's default (side by side) and one after the other, interleaved rounds:
#   tools/ab_fused_gen.sh "<workloads>" [rounds] [steps]
W="${1:-mixed am_narrow}"; R=${2:-3}; STEPS=${3:-100}
for round in $(seq 1 $R); do for wl in $W; do for f in 3 1; do
  printf "%s %-10s fused=%s " $round $wl $f
  python bench.py --workload $wl --fused $f --steps $STEPS --warmup 2 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'ms', d['config']['chain'][:40])"
done; done; done
