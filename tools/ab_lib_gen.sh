#!/bin/bash
# library variants of the general-mode fused kernel, --fused 3: tools/ab_lib_gen.sh "<variants>" "<workloads>" [steps]
V="$1"; W="${2:-mixed am_narrow}"; STEPS=${3:-60}
for wl in $W; do for v in $V; do
  printf "%-10s %-10s " $wl $v
  lib=$PWD/supersdr_amd/libssdr_$v.so; [ $v = main ] && lib=$PWD/supersdr_amd/libssdr.so
  SSDR_LIB_PATH=$lib python bench.py --workload $wl --fused 3 --steps $STEPS --warmup 2 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],3), 'ms')"
done; done
