#!/usr/bin/env python3
"""Run the seeded random sweeps of tests/ with seeds the suite does not hold.

    python tools/soak_parity.py [--first 1000] [--count 100] [--out gpurun_out/soak.txt]      (GPU box: kernels vs twin vs oracle)
    python tools/soak_parity.py --cpu [--first 1000] [--count 100]                            (no GPU: fp32 twin vs float64 oracle)

Each sweep is the test function itself (same asserts: bit-exact against the fp32 twin, tolerance rule of
tests/tolerances.py against the float64 oracle); a failing seed is reported with its assertion, the run goes on.
Exit code 1 if any seed failed.
"""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--count", type=int, default=100)
    ap.add_argument("--out", default=None)
    ap.add_argument("--cpu", action="store_true", help="the CPU sweep only: fp32 twin against the float64 oracle (tests/tolerances.py)")
    a = ap.parse_args()

    import tolerances as T
    if a.cpu:
        import test_oracle_known_answers as TK
        sweeps = [("twin vs float64 oracle, 24 ch x 6 frames", lambda s: TK.test_twin_tracks_float64_oracle_over_random_parameters(s, 6))]
        return run_sweeps(a, sweeps, T)
    import supersdr_amd as S
    import supersdr_amd.engine as _E
    _E.DEFAULT_CHAIN_FLOORS = (0, 0)                    # small batches through the one-read kernels too (ssdr_set_chain_floors)
    import twinlib
    import test_gpu_parity as TP
    import test_gpu_post as TQ
    twin = twinlib.load()

    sweeps = [
        ("parameter surface, 96 ch x 6 frames", lambda s: TP.test_random_parameter_surface_bit_exact_vs_twin(S, twin, s, 6)),
        ("parameter surface, 96 ch x 48 frames", lambda s: TP.test_random_parameter_surface_bit_exact_vs_twin(S, twin, s, 48) if s % 8 == 0 else None),
        ("waterfall batching / averaging", lambda s: TP.test_wf_random_batching_and_averaging(S, twin, s)),
    ]
    shapes = [(64, 1, 1024), (33, 10, 1024), (7, 3, 512), (20, 1, 512)]
    sweeps += [
        ("exact bins == NumPy float64, random levels", lambda s: TP.test_exact_bins_equal_the_float64_oracle_bit_for_bit(S, *shapes[s % 4], seed=s)),
        ("decimating front end, D = 2 / 4", lambda s: TP.test_decimating_front_end_bit_exact_vs_twin_and_oracle(S, twin, 2 + 2 * (s & 1), seed=s)),
        ("IQ at 20.25 kHz", lambda s: TP.test_iq_chain_at_20250_hz_bit_exact_vs_twin_and_oracle(S, twin, seed=s) if s % 2 == 0 else None),
    ]
    avgs = [1, 2, 7, 10, 33, 100]
    sweeps.append(("spectrum_db2col, random lines vs the oracle", lambda s: TQ.test_db2col_random_lines_vs_oracle(S, s, avgs[s % len(avgs)])))
    return run_sweeps(a, sweeps, T)


def run_sweeps(a, sweeps, T):
    lines, bad = [], 0
    for title, run in sweeps:
        t0, failed = time.time(), []
        for s in range(a.first, a.first + a.count):
            try:
                run(s)
            except Exception as e:                              # noqa: BLE001 -- an assertion of the test, reported per seed
                failed.append(s)
                tb = traceback.format_exc().strip().splitlines()
                lines.append("  seed %d: %s | %s" % (s, type(e).__name__, " / ".join(x.strip() for x in tb[-3:])[:400]))
        bad += len(failed)
        lines.append("%-44s seeds %d..%d: %d failed %s (%.0f s)" % (title, a.first, a.first + a.count - 1, len(failed), failed[:10], time.time() - t0))
        print(lines[-1], flush=True)
    rep = T.REPORT
    if rep:
        well, chans = sum(r["well_conditioned"] for r in rep), sum(r["channels"] for r in rep)
        lines.append("pcm vs float64 oracle over %d sweeps, %d channels: %d well conditioned (%.1f %%), their largest untrimmed RMS %.2e of full scale; "
                     "largest share of its propagated bound (EPS = 2^%d) any sample used: %.2f (%d samples at the FM discriminator's branch cut, a full turn allowed, set aside)"
                     % (len(rep), chans, well, 100.0 * well / max(chans, 1), max(r["untrimmed_max_well"] for r in rep),
                        int(round(__import__("math").log2(T.EPS))), max(r["worst_sample_not_at_the_branch_cut"] for r in rep), sum(r["samples_at_the_branch_cut"] for r in rep)))
        if "rms_all_channels_off_the_branch_cut_max" in rep[0]:
            out = sum(r["outside_plain_tolerance"] for r in rep)
            lines.append("channels outside the plain 1e-5 RMS: %d of %d (%.1f %%); largest RMS of ANY channel over its samples off the branch cut: %.2e"
                         % (out, chans, 100.0 * out / max(chans, 1), max(r["rms_all_channels_off_the_branch_cut_max"] for r in rep)))
        try:                                                       # the gate of tests/tolerances.py on the soak's own total (no small-sample allowance)
            n, out, allowed = T.assert_share_outside_plain(rep, strict_above=0)
            lines.append("gate: at most %.0f %% of the channels outside the plain 1e-5 RMS -- %d of %d, %d allowed: ok" % (100 * T.PLAIN_MISS_SHARE, out, n, allowed))
        except AssertionError as e:
            bad += 1
            lines.append("GATE FAILED: %r" % (e.args,))
        T.write_report(os.path.join(os.path.dirname(a.out) if a.out else os.path.join(ROOT, "gpurun_out"), "tolerance_report_soak.json"), rep)
    text = "\n".join(lines) + "\n"
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(text)
    print(text)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
