#!/usr/bin/env python3
"""configs[3]'s audio stage with the SSB channel filter shortened from 33 to 25 taps (a wider passband: the reference's tap formula,
utils_supersdr.py:334-344, gives N = 25 for a 1985 Hz half-width) -- an UPPER BOUND on what a 2-parallel fast-FIR form of the
33-tap filter (3 x 17 / 2 = 25.5 multiply-adds per output instead of 33, plus its pre / post additions) could buy.
   python tools/fir_bound_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np
import supersdr_amd as S
from supersdr_amd import _lib as L

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_ch, n_frames = 65536, 20
modes = ("am", "usb", "lsb", "nbfm")
for rounds in range(2):
    for hc in (3000.0, 4000.0):
        with S.SsdrEngine(n_ch) as eng:
            ps = []
            for c in range(388):
                m = modes[c % 4]
                over = {}
                if m == "usb":
                    over = dict(low_cut=30.0, high_cut=hc)
                if m == "lsb":
                    over = dict(low_cut=-hc, high_cut=-30.0)
                ps.append(S.default_params(m, f_shift_hz=((c * 37) % 97 - 48) * 100.0, **over))
            for first in range(0, n_ch, 388):
                eng.set_params(first, ps[: min(388, n_ch - first)])
            eng.reset_state()
            eng.synth_iq(n_frames)
            k, _ = eng.get_consts(0, 4)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 1.0:
                eng.run_audio(fetch=False)
            eng.sync()
            eng.set_profiling(True)
            eng.kernel_stats(L.K_AUDIO, reset=True)
            for _ in range(steps):
                eng.run_audio(fetch=False)
            ms, n = eng.kernel_stats(L.K_AUDIO)
            b = n_ch * n_frames * 3072.0
            print("SSB passband 30..%d Hz: %d taps; audio stage %.4f ms, %.3f of the HBM roof (paths %s)"
                  % (hc, int(k["ntap"][1]), ms / n, b / (ms / n) / 1e6 / 8000.0, eng.audio_paths()))
