#!/usr/bin/env python3
"""profiles/rNN_traffic_<workload>.json from the FETCH_SIZE / WRITE_SIZE PMC passes of tools/profile_round.sh.

HBM bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024.  The factor 2 is the gfx950
correction of MI355X_MICROARCH.md (section HBM): FETCH_SIZE counts 128-B fabric requests at 64 B.  It was
re-calibrated on this repo's own access patterns: the waterfall kernel's known input (4096 B per line,
nothing else of size) reads FETCH_SIZE = 0.5005 x its byte count; WRITE_SIZE matches 1:1.

    python tools/traffic_json.py gpurun_out/prof_r02 gpurun_out/prof_r02/bench_under_trace.json > profiles/r02_traffic_full.json

The bench line of the traced run names the workload shape; kernels are keyed the way bench.py names them
("ssdr_wf_kernel<false, false>", "ssdr_audio_kernel<2>").
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def mean_per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                m = re.search(r"(ssdr_\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
                if m:
                    acc[m.group(1)].append(float(r["Counter_Value"]))
    return {k: sum(v[2:]) / len(v[2:]) if len(v) > 2 else sum(v) / len(v) for k, v in acc.items()}


def main():
    d = sys.argv[1]
    line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    cfg = line["config"]
    fetch = mean_per_kernel(d + "/fetch_counter_collection.csv", "FETCH_SIZE")
    write = mean_per_kernel(d + "/write_counter_collection.csv", "WRITE_SIZE")
    wl = cfg["workload_key"] if "workload_key" in cfg else [k for k in ("full", "wf", "mixed", "million", "decim4") if {"full": "configs[2]", "wf": "configs[1]", "mixed": "configs[3]",
                                                                     "million": "configs[4]", "decim4": "ssdr_set_decimation(4)"}[k] in cfg["workload"]][0]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    head = os.path.join(bench.ROOT, ".ssdr_head")          # written by tools/stamp_head.sh before the snapshot goes to the GPU box
    out = {"workload": wl, "git_commit": open(head).read().strip() if os.path.exists(head) else None, "csrc_sha256": bench.csrc_sha256(), "channels_per_gpu": cfg["channels_per_gpu"], "superframes_per_step": cfg["superframes_per_step"],
           "wf_hop": cfg.get("wf_hop", 1024),
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = 2*FETCH_KB*1024 + WRITE_KB*1024",
           "kernels": {}}
    for k in fetch:
        if k in write:
            out["kernels"][k] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write[k],
                                 "hbm_bytes_per_launch": 2 * fetch[k] * 1024 + write[k] * 1024}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
