#!/bin/bash
# gpurun_out/prof_<round>_* (tools/profile_all.sh) -> profiles/<round>_*: kernel stats, PMC summaries, traffic JSON (with commit and source hash),
# the bench line of the traced run.     tools/collect_profiles.sh r05
R=${1:-r05}
for d in gpurun_out/prof_${R}_*; do
  n=$(basename $d | sed "s/prof_${R}_//")
  cp $d/kernel_stats.txt profiles/${R}_${n}_kernel_stats.txt
  grep -v "rocclr\|synth" $d/pmc_summary.txt > profiles/${R}_${n}_pmc_summary.txt
  python - "$d/traffic.json" "profiles/${R}_traffic_${n}.json" <<'PY'
import json, sys
t = json.load(open(sys.argv[1]))
t["kernels"] = {k: v for k, v in t["kernels"].items() if "synth" not in k}
json.dump(t, open(sys.argv[2], "w"), indent=1)
PY
  cp $d/bench_under_trace.json profiles/${R}_bench_${n}_under_rocprof.json
done
grep -h '"csrc_sha256"' profiles/${R}_traffic_*.json | sort | uniq -c
