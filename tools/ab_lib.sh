#!/bin/bash
# library variants against the shipped library, ssdr_run_chain's defaults, interleaved rounds on one box:
#   tools/ab_lib.sh "<variants>" "<workloads, or workload:flag=value>" [rounds] [steps]
V="$1"; W="${2:-full mixed am_narrow}"; R=${3:-2}; STEPS=${4:-100}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for round in $(seq 1 $R); do for spec in $W; do
  wl=${spec%%:*}; fl=""; [ "$spec" != "$wl" ] && fl="--${spec#*:}"; fl=${fl/=/ }
  for v in main $V; do
    lib=$PWD/supersdr_amd/libssdr_$v.so; [ $v = main ] && lib=$PWD/supersdr_amd/libssdr.so
    printf "%-22s %-10s " "$spec" $v
    SSDR_LIB_PATH=$lib python bench.py --workload $wl $fl --steps $STEPS --warmup 2 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,3), 'M', round(d['ms_per_step'],3), 'ms', (d.get('roofline') or {}).get('power', {}).get('avg_watts'), 'W', (d.get('roofline') or {}).get('power', {}).get('joules_per_step'), 'J')"
  done
done; done
