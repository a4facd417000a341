"""Seeded random channel parameters over the whole a13 parameter surface (shared by the CPU and GPU sweeps)."""
import numpy as np

MODES = ["am", "lsb", "usb", "cw", "nbfm"]


def draw(rng):
    mode = MODES[int(rng.integers(0, 5))]
    if mode in ("am", "nbfm"):
        hc = float(rng.choice([6000, 5000, 4000, 2500, 1200, 300]))
        lc = -hc if rng.random() < 0.7 else -float(rng.choice([6000, 3000, 800]))
    elif mode == "usb":
        lc = float(rng.integers(0, 600)); hc = lc + float(rng.integers(150, 3500))
    elif mode == "lsb":
        hc = -float(rng.integers(0, 600)); lc = hc - float(rng.integers(150, 3500))
    else:                                                   # cw: narrow, down to the 127-tap limit
        c = float(rng.integers(300, 1000)); w = float(rng.choice([50, 100, 200, 400, 800]))
        lc, hc = c - w / 2, c + w / 2
    return dict(mode=mode, f_shift_hz=float(rng.integers(-5900, 5901)) + float(rng.choice([0.0, 0.25, 0.5])),
                low_cut=lc, high_cut=hc, agc_on=int(rng.random() < 0.8), hang=int(rng.random() < 0.3),
                thresh=float(rng.integers(-130, -20)), slope=float(rng.integers(0, 11)),
                decay=float(rng.choice([20, 100, 400, 1000, 4000, 8000])), man_gain=float(rng.integers(0, 100)),
                wf_cal_db=float(rng.integers(-20, 21)), smeter_cal_db=float(rng.choice([-13.0, 0.0, -20.5])))


def signal(rng, n_ch, n_samples):
    """int16 [n_ch, n_samples, 2]: carriers of random level (a few near full scale, a few silent) with AM/FM and noise."""
    t = np.arange(n_samples)
    out = np.empty((n_ch, n_samples, 2), np.int16)
    for c in range(n_ch):
        amp = float(rng.choice([0.0, 30.0, 800.0, 8000.0, 23000.0]))
        f = float(rng.integers(-5500, 5500))
        am = 1 + 0.6 * np.sin(2 * np.pi * float(rng.integers(100, 2500)) * t / 12000.0)
        ph = 2 * np.pi * f * t / 12000.0 + 2.0 * np.sin(2 * np.pi * float(rng.integers(100, 2500)) * t / 12000.0)
        z = amp * am * np.exp(1j * ph) + float(rng.choice([0.0, 3.0, 300.0])) * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
        if rng.random() < 0.15:
            z[n_samples // 3: n_samples // 3 + 40] = 32767 * (1 + 1j)          # a burst at the rails
        out[c, :, 0] = np.clip(np.rint(z.real), -32768, 32767)
        out[c, :, 1] = np.clip(np.rint(z.imag), -32768, 32767)
    return out
