#!/usr/bin/env python3
"""Runs the SURVEY.md 8f kernels a few times on the bench's post shape (for rocprofv3 passes): tools/post_probe.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process, loaded first)
import bench, supersdr_amd as S
from supersdr_amd import _lib as L
st = bench.measure_post(S, L, 0, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 5, spinup=0.0)
for k, v in st.items():
    print(k, round(v["avg_ms"], 3), "ms", round(v["GBps"]), "GB/s", round(v["GBps"] / 8000, 3))
