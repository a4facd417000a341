#!/bin/bash
# the wave-specialised chain kernel (--fused 3) of the named library variants against the two stages side by side (--fused 0),
# interleaved rounds on one box:   tools/ab_ws.sh "<variants: main = supersdr_amd/libssdr.so>" "<workloads, or workload:flag=value>" [rounds] [steps]
V="${1:-main}"; W="${2:-mixed am_narrow}"; R=${3:-2}; STEPS=${4:-100}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() {  # label lib fused spec
  local wl=${4%%:*} fl=""; [ "$4" != "$wl" ] && fl="--${4#*:}"; fl=${fl/=/ }
  printf "%-20s %-12s " $4 $1
  SSDR_LIB_PATH=$2 python bench.py --workload $wl $fl --fused $3 --steps $STEPS --warmup 2 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'ms', d['config']['chain'][:34], (d.get('roofline') or {}).get('power', {}).get('avg_watts'), 'W', (d.get('roofline') or {}).get('power', {}).get('joules_per_step'), 'J')"
}
for round in $(seq 1 $R); do for spec in $W; do
  run side_by_side $PWD/supersdr_amd/libssdr.so 0 $spec        # (--fused 0: the two stages side by side on two streams)
  for v in $V; do
    lib=$PWD/supersdr_amd/libssdr_$v.so; [ $v = main ] && lib=$PWD/supersdr_amd/libssdr.so
    run ws:$v $lib 3 $spec
  done
done; done
