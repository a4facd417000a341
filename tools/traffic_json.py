#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the FETCH_SIZE / WRITE_SIZE PMC passes of tools/profile_round.sh.

HBM bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024.  The factor 2 is the gfx950
correction of MI355X_MICROARCH.md (section HBM): FETCH_SIZE counts 128-B fabric requests at 64 B.  It was
re-calibrated on this repo's own access patterns: the waterfall kernel's known input (4096 B per line,
nothing else of size) reads FETCH_SIZE = 0.5005 x its byte count; WRITE_SIZE matches 1:1.

    python tools/traffic_json.py gpurun_out/prof_r01 full 65536 4 > profiles/r01_traffic.json
"""
import csv
import json
import sys
from collections import defaultdict


def mean_per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v[2:]) / len(v[2:]) if len(v) > 2 else sum(v) / len(v) for k, v in acc.items()}


def main():
    d, workload, channels, sframes = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    fetch = mean_per_kernel(d + "/fetch_counter_collection.csv", "FETCH_SIZE")
    write = mean_per_kernel(d + "/write_counter_collection.csv", "WRITE_SIZE")
    out = {"workload": workload, "channels_per_gpu": channels, "superframes_per_step": sframes,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = 2*FETCH_KB*1024 + WRITE_KB*1024",
           "kernels": {}}
    for k in fetch:
        short = "ssdr_wf_kernel" if "ssdr_wf_kernel" in k else "ssdr_audio_kernel" if "ssdr_audio_kernel" in k else None
        if short and k in write:
            out["kernels"][short] = {"FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write[k],
                                     "hbm_bytes_per_launch": 2 * fetch[k] * 1024 + write[k] * 1024}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
