for sf in 4 8 32; do for f in 1 0 1 0; do
  printf "sf=%s fused=%s " $sf $f
  python bench.py --workload full --superframes $sf --fused $f --steps 60 --warmup 2 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3))"
done; done
