# fused superframe kernel against the two per-stage kernels, interleaved rounds in one call: tools/ab_fused.sh [workload] [rounds]
WL=${1:-full}
for round in $(seq 1 ${2:-3}); do for f in 1 0; do
  printf "%s %s fused=%s " $round $WL $f
  python bench.py --workload $WL --fused $f --steps 100 --warmup 2 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3))"
done; done
