#!/usr/bin/env python3
"""Differential fuzz of the C-ABI's call sequences (GPU box).

Two contexts take the SAME random sequence of calls -- parameter changes over channel ranges (modes that move channels between
the three audio paths and switch the fused kernel's eligibility on and off), N, hop, float64 bins, resets of channel ranges,
checkpoint / restore, pushes of 2..24 frames, ssdr_run_chain / ssdr_run_wf / ssdr_run_audio in any order.  One context runs with the
library's fast paths as they come (fused kernel incl. its opt-in forms, the two stages side by side on two streams); the other with
all of them off (two kernels, one after the other).  Whatever the sequence, every result -- waterfall sums, PCM, RSSI, flags,
checksums, carried state, history, checkpoint blobs -- must be identical, call by call.

    python tools/fuzz_api.py [--first 1] [--count 50] [--ops 60] [--out gpurun_out/fuzz.txt]
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def random_params(S, rng, k, am_bias):
    out = []
    for _ in range(k):
        r = rng.random()
        if r < am_bias:
            out.append(S.default_params("am"))                                   # full-band AM: the fused kernel's path
        elif r < am_bias + 0.1:
            out.append(S.default_params("am", f_shift_hz=float(rng.integers(-3000, 3000))))      # full band, tuned: the lane-shift path
        else:
            m = ["am", "usb", "lsb", "cw", "nbfm"][int(rng.integers(0, 5))]
            kw = dict(f_shift_hz=float(rng.integers(-4000, 4000)), agc_on=int(rng.random() < 0.8), agc_hang=int(rng.random() < 0.3),
                      agc_decay=float(rng.choice([100, 1000, 4000])), wf_cal_db=float(rng.integers(-10, 11)))
            if m in ("am", "nbfm"):
                kw.update(low_cut=-float(rng.choice([2500, 4000])), high_cut=float(rng.choice([2500, 4000])))
            out.append(S.default_params(m, **kw))
    return out


def one_sequence(S, seed, n_ops, log):
    rng = np.random.default_rng(seed)
    n_ch = int(rng.choice([1, 2, 5, 16, 33, 70, 257]))
    am_bias = float(rng.choice([1.0, 1.0, 0.6, 0.0]))              # half of the sequences start where the fused kernel applies
    A, B = S.SsdrEngine(n_ch), S.SsdrEngine(n_ch)
    counts = {"fused": 0, "runs": 0}
    try:
        A.set_fused(int(rng.integers(1, 4)))                       # 3: the wave-specialised chain kernel (round 6) wherever the batch allows it
        B.set_fused(0)
        B.set_overlap(0)
        ps = random_params(S, rng, n_ch, am_bias)
        for e in (A, B):
            e.set_params(0, ps)
        have, hop, blob = False, 1024, None

        def same(what, a, b):
            ok = (a == b) if isinstance(a, (bytes, tuple, int, bool)) else np.array_equal(a, b)
            if not ok:
                raise AssertionError("%s differs (seed %d, %d channels, op %d)" % (what, seed, n_ch, counts["runs"]))

        def compare_results(lines_a, lines_b, audio):
            same("line count", lines_a, lines_b)
            same("checksums", A.output_checksum(), B.output_checksum())
            if lines_a:
                same("waterfall", A.fetch_wf(lines_a), B.fetch_wf(lines_b))
            if audio:
                (pa, ra), (pb, rb) = A.fetch_audio(), B.fetch_audio()
                same("pcm", pa, pb), same("rssi", ra, rb), same("flags", A.audio_flags(), B.audio_flags())
            (sa, ha), (sb, hb) = A.get_state(), B.get_state()
            same("state", sa.tobytes(), sb.tobytes()), same("history", ha, hb)

        for _ in range(n_ops):
            op = int(rng.integers(0, 14))
            if op <= 3 or not have:                                               # a new batch
                nf = int(rng.choice([2, 4, 8, 8, 10, 16, 24])) if hop == 1024 else int(rng.choice([1, 3, 8, 8, 9, 16, 21]))
                amp = float(rng.choice([0.0, 50.0, 3000.0, 20000.0]))
                iq = np.clip(np.rint(rng.standard_normal((n_ch, nf * 512, 2)) * max(amp / 3, 1.0) + amp * np.cos(np.arange(nf * 512) * 0.3)[None, :, None]),
                             -32768, 32767).astype(np.int16)
                if rng.random() < 0.1:
                    iq[:, 100:110] = 32767                                        # ADC overflow
                for e in (A, B):
                    e.push_iq(iq)
                have, audio_ran = True, False
            elif op <= 6:
                (la, fa), (lb, fb) = A.run_chain(), B.run_chain()
                counts["fused"] += int(fa != 0)
                counts["chain_ws"] = counts.get("chain_ws", 0) + int(fa == 2)
                counts["runs"] += 1
                assert not fb
                compare_results(la, lb, True)
            elif op == 7:
                la, lb = A.run_wf(fetch=False), B.run_wf(fetch=False)
                counts["runs"] += 1
                same("line count", la, lb)
                if la:
                    same("waterfall", A.fetch_wf(la), B.fetch_wf(lb))
            elif op == 8:
                (pa, ra), (pb, rb) = A.run_audio(), B.run_audio()
                counts["runs"] += 1
                same("pcm", pa, pb), same("rssi", ra, rb), same("flags", A.audio_flags(), B.audio_flags())
            elif op == 9:
                f = int(rng.integers(0, n_ch)) if rng.random() < 0.6 else 0      # (often the whole range: back to where the fused kernel applies)
                k = int(rng.integers(1, n_ch - f + 1)) if f else n_ch
                ps = random_params(S, rng, k, float(rng.choice([1.0, 1.0, 0.5, 0.0])))
                for e in (A, B):
                    e.set_params(f, ps)
            elif op == 10:
                n = int(rng.choice([1, 1, 2, 3, 10]))
                for e in (A, B):
                    e.set_averaging(n)
            elif op == 11:
                r = rng.random()
                if r < 0.4:
                    hop = 512 if hop == 1024 else 1024
                    for e in (A, B):
                        e.set_hop(hop)
                    have = False                                                  # (frame counts that fit one hop need not fit the other)
                elif r < 0.7:
                    on = bool(rng.random() < 0.5)
                    for e in (A, B):
                        e.set_exact_bins(on)
                else:
                    A.set_fused(int(rng.integers(0, 3)))
                    A.set_overlap(int(rng.random() < 0.8))
            elif op == 12:
                f = int(rng.integers(0, n_ch))
                k = int(rng.integers(1, n_ch - f + 1))
                for e in (A, B):
                    e.reset_state(f, k)
            else:
                if blob is None or rng.random() < 0.5:
                    ba, bb = A.checkpoint(), B.checkpoint()
                    same("checkpoint blob", ba, bb)
                    blob = (ba, hop)
                else:
                    for e in (A, B):
                        e.restore(blob[0])
                    hop, have = blob[1], False
        return counts
    finally:
        A.close()
        B.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=1)
    ap.add_argument("--count", type=int, default=50)
    ap.add_argument("--ops", type=int, default=60)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import supersdr_amd as S
    import supersdr_amd.engine as _E
    _E.DEFAULT_CHAIN_FLOORS = (0, 0)                    # small batches through the one-read kernels too (ssdr_set_chain_floors)
    lines, bad, fused, runs, t0 = [], 0, 0, 0, time.time()
    chain_ws = 0
    for seed in range(a.first, a.first + a.count):
        try:
            c = one_sequence(S, seed, a.ops, lines)
            fused += c["fused"]
            chain_ws += c.get("chain_ws", 0)
            runs += c["runs"]
        except Exception as e:                                                    # noqa: BLE001 -- reported per seed, the run goes on
            bad += 1
            tb = traceback.format_exc().strip().splitlines()
            lines.append("  seed %d: %s: %s | %s" % (seed, type(e).__name__, e, " / ".join(x.strip() for x in tb[-4:-1])[:300]))
            print(lines[-1], flush=True)
    lines.append("differential API fuzz: seeds %d..%d, %d calls each: %d sequences differed; %d kernel runs compared, %d of them through a fused kernel (%d through the wave-specialised one) (%.0f s)"
                 % (a.first, a.first + a.count - 1, a.ops, bad, runs, fused, chain_ws, time.time() - t0))
    text = "\n".join(lines) + "\n"
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(text)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
