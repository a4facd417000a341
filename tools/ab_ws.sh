#!/bin/bash
# the wave-specialised chain kernel (--fused 3) of the named library variants against ssdr_run_chain's default (the stages side by side),
# interleaved rounds on one box:   tools/ab_ws.sh "<variants: main = supersdr_amd/libssdr.so>" "<workloads>" [rounds] [steps]
V="${1:-main}"; W="${2:-mixed am_narrow}"; R=${3:-2}; STEPS=${4:-100}
run() {  # label lib fused workload
  printf "%-10s %-12s " $4 $1
  SSDR_LIB_PATH=$2 python bench.py --workload $4 --fused $3 --steps $STEPS --warmup 2 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'ms', 'chain_frac', round(d.get('chain_frac', 0) or 0, 4))"
}
for round in $(seq 1 $R); do for wl in $W; do
  run side_by_side $PWD/supersdr_amd/libssdr.so 0 $wl        # (--fused 0: the two stages side by side on two streams)
  for v in $V; do
    lib=$PWD/supersdr_amd/libssdr_$v.so; [ $v = main ] && lib=$PWD/supersdr_amd/libssdr.so
    run ws:$v $lib 3 $wl
  done
done; done
