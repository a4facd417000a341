for args in "--workload wf" "--workload mixed" "--workload million --steps 5" "--workload decim4" "--workload wf --hop 512" "--workload wf --exact 1 --steps 3" "--workload full --host-feed 1 --channels 8192" "--workload full --host-feed 2 --channels 8192" "--workload full --concurrent 1" "--workload full --fused 0" "--workload full --superframes 2"; do
  printf "%-55s " "$args"
  timeout 300 python bench.py $args --steps 10 --warmup 1 --spinup 0.3 --no-cpu-baseline --verbose-line --no-extra --no-parity-probe 2>/tmp/err.txt | python -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]) if t else None
print(round(d['value']/1e6,2), round(d['ms_per_step'],3), d['roofline']['kernel'][:40]) if d else print('NO OUTPUT')" || tail -3 /tmp/err.txt
done
