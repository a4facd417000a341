#!/usr/bin/env python3
"""Fresh-seed soak of the wave-specialised chain kernel (GPU box): tests/test_gpu_parity.py::test_wave_specialised_kernel_equals_the_two_kernels
with random channel counts, N and seeds -- every result of ssdr_set_fused(ctx, 3) against the two kernels, bit for bit.
    python tools/soak_chain_ws.py [--first 100] [--count 200] [--out gpurun_out/soak_chain_ws.txt]"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--count", type=int, default=200)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import supersdr_amd as S
    import supersdr_amd.engine as _E
    _E.DEFAULT_CHAIN_FLOORS = (0, 0)                    # small batches through the one-read kernels too (ssdr_set_chain_floors)
    import test_gpu_parity as TP
    bad, t0, lines, chans = 0, time.time(), [], 0
    for seed in range(a.first, a.first + a.count):
        rng = np.random.default_rng(seed)
        n_ch = int(rng.choice([1, 2, 3, 8, 31, 64, 129, 300, 513]))
        n_avg = int(rng.choice([1, 1, 2, 3, 7, 10]))
        try:
            TP.test_wave_specialised_kernel_equals_the_two_kernels(S, n_ch, n_avg, seed)
            chans += n_ch
        except Exception as e:                                        # noqa: BLE001
            bad += 1
            tb = traceback.format_exc().strip().splitlines()
            lines.append("  seed %d (%d ch, N = %d): %s | %s" % (seed, n_ch, n_avg, type(e).__name__, " / ".join(x.strip() for x in tb[-3:])[:300]))
            print(lines[-1], flush=True)
    lines.append("wave-specialised chain kernel vs the two kernels: seeds %d..%d (random channel counts 1..513, N in {1,2,3,7,10}, five calls each incl. one of "
                 "132 frames and a retune): %d differed; %d channels compared in waterfall sums, PCM, RSSI, flags, carried state, raw history (%.0f s)"
                 % (a.first, a.first + a.count - 1, bad, chans, time.time() - t0))
    text = "\n".join(lines) + "\n"
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(text)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
