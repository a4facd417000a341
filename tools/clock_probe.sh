#!/bin/bash
# Samples sclk / power / temperature while a bench workload runs: tools/clock_probe.sh <workload> [steps]
# (shows the DVFS ramp from the idle state and whether the kernel runs at the board's power cap)
WL="${1:-wf}"; STEPS="${2:-3000}"
rocm-smi --showmaxpower 2>/dev/null | grep -E "Max" | sed 's/.*: //' | tr '\n' ' '; echo "W cap"
python bench.py --workload $WL --steps $STEPS --warmup 2 --spinup 0 --no-cpu-baseline > /tmp/probe_bench.json 2>/dev/null &
BP=$!
n=0
while kill -0 $BP 2>/dev/null; do
  L=$(rocm-smi --showclocks --showpower --showuse --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Socket|GPU use|junction|memory\)" | sed 's/.*: //' | tr '\n' ' ')
  U=$(echo "$L" | awk '{print $(NF)}')
  if [ "${U:-0}" -gt 0 ] 2>/dev/null; then echo "$(date +%s.%N | cut -c7-14) $L"; n=$((n+1)); fi
  [ $n -ge 14 ] && break
done
wait $BP
tail -1 /tmp/probe_bench.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v['achieved']),round(v['avg_kernel_ms'],3)) for k,v in d.items() if k.startswith('roofline')})"
