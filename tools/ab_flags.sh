#!/bin/bash
# A/B of bench.py flag sets in interleaved rounds: tools/ab_flags.sh "<flags A>|<flags B>|..." [rounds]
IFS='|' read -ra SETS <<< "$1"; ROUNDS="${2:-2}"
for round in $(seq 1 $ROUNDS); do for fl in "${SETS[@]}"; do
  printf "%s %-44s " "$round" "$fl"
  python bench.py --warmup 2 --no-cpu-baseline --verbose-line $fl 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2),'M rt; ms/step', round(d['ms_per_step'],3), {k.replace('roofline','r'):(round(v['achieved']),round(v['avg_kernel_ms'],3)) for k,v in d.items() if k.startswith('roofline')})"
done; done
