# extra.mixed of the default bench line for the named library variants (debugging aid)
for v in "$@"; do
  lib=$PWD/supersdr_amd/libssdr_$v.so; [ $v = main ] && lib=$PWD/supersdr_amd/libssdr.so
  printf "%-10s " $v
  SSDR_LIB_PATH=$lib python bench.py --no-cpu-baseline --host-feed-extra 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), {k:(round(v['value']/1e6,2), round(v['ms_per_step'],3)) for k,v in d['extra'].items() if isinstance(v,dict) and 'value' in v and k in ('mixed','mixed_serial','mixed_chain_ws')})"
done
printf "alone      "; python bench.py --workload mixed --steps 60 --spinup 1.0 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3))"
