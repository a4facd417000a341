for round in 1 2 3; do for v in v3 v6; do for f in 1 0; do
  printf "%s %s fused=%s " $round $v $f
  SSDR_LIB_PATH=$PWD/supersdr_amd/libssdr_$v.so python bench.py --workload full --fused $f --steps 100 --warmup 2 --no-cpu-baseline --no-extra --no-parity-probe 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3))"
done; done; done
