#!/bin/bash
# A/B bench of library variants on the GPU box: tools/ab_bench.sh "<variants>" "<workloads>" [steps]
# (variants are supersdr_amd/libssdr_<v>.so; interleaved rounds in one call, guide rule 24)
V="$1"; W="$2"; STEPS="${3:-100}"
for round in 1 2; do for wl in $W; do for v in $V; do
  printf "%s %-6s %-14s " "$round" "$wl" "$v"
  SSDR_LIB_PATH=$PWD/supersdr_amd/libssdr_$v.so python bench.py --workload $wl --steps $STEPS --warmup 2 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2),'M rt;', {k.replace('roofline','r'):(round(v['achieved']),round(v['avg_kernel_ms'],3)) for k,v in d.items() if k.startswith('roofline')})"
done; done; done
