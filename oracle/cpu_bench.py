"""cpu_bench.py -- work units of bench.py's `cpu_baseline` leg: the NumPy float64 oracle on a block of channels.

*** TEST INFRASTRUCTURE ONLY *** (like everything under oracle/): imported by bench.py's cpu_baseline() and by its
spawned worker processes, never by the product path.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ssdr_oracle as O  # noqa: E402

PASSBAND = {"am": (-6000, 6000), "usb": (30, 3000), "lsb": (-3000, -30), "nbfm": (-6000, 6000), "cw": (400, 800)}


def chan_params(c, modes):
    """the bench's parameter pattern: mode by c mod len(modes), carrier (c*37 mod 97 - 48) * 100 Hz, reference passbands"""
    m = modes[c % len(modes)]
    lc, hc = PASSBAND[m]
    return O.ChanParams(mode=m, f_shift_hz=((c * 37) % 97 - 48) * 100.0, low_cut=lc, high_cut=hc)


def noop(_):
    return os.getpid()


def run_block(job):
    """(first_channel, n_channels, superframes, n_avg, modes, do_wf, do_audio) -> channels processed"""
    first, n, sf, n_avg, modes, do_wf, do_audio = job
    iq = O.synth_iq(n, sf * 1024, seed=7, first_ch=first)
    if do_wf:
        for c in range(n):
            O.wf_sum_lines(iq[c].reshape(-1, 1024, 2), n_avg if sf % n_avg == 0 else 1, 0.0)
    if do_audio:
        O.audio_chain(iq, [chan_params(first + c, modes) for c in range(n)])
    return n
