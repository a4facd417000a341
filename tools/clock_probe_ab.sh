#!/bin/bash
# sclk / socket power (rocm-smi) while a bench workload runs: tools/clock_probe_ab.sh "<label>|<env assignments>|<bench flags>" ...   (through gpurun)
# e.g.  tools/clock_probe_ab.sh "full||--workload full" "two kernels||--workload full --fused 0 --overlap 0" "variant|SSDR_LIB_PATH=$PWD/supersdr_amd/libssdr_x.so|--workload full"
probe() {
  env $2 python bench.py $3 --steps ${STEPS:-4000} --warmup 2 --spinup 0 --no-cpu-baseline --no-extra --no-parity-probe > /tmp/pb.json 2>/dev/null &
  BP=$!
  sleep 3
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done
  wait $BP
  python -c "
import sys,json
d=json.loads(open('/tmp/pb.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms')"
}
for spec in "$@"; do
  IFS='|' read -r label envs flags <<< "$spec"
  echo "== $label"; probe "$label" "$envs" "$flags"
done
