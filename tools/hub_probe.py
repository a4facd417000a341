#!/usr/bin/env python3
"""The pipelined feed behind IQHub's ingest API on the GPU box: feed_block with 0 / 4 / 8 / 16 copy threads, and in place.
   python tools/hub_probe.py [channels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, supersdr_amd as S
from supersdr_amd import _lib as L
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for ip, ct in ((False, 0), (False, 4), (False, 8), (False, 16), (True, 0)):
    h = bench.measure_hub(S, L, torch, 0, ch, 16, 3, in_place=ip, copy_threads=ct)
    print("%-48s %.3f M real-time channels, %.2f ms per superframe of %d channels, %.1f GB/s of IQ" % (h["ingest"], h["value"] / 1e6, h["ms_per_superframe"], ch, h["host_GBps"]))
