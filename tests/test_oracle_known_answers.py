"""Known-answer tests for the stages the reference does NOT contain (FFT/log-mag, NCO, FIR,
demodulators, AGC -- parity unpinned), and the fp32 C twin against the NumPy
float64 oracle.  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ssdr_oracle as O  # noqa: E402
import twinlib  # noqa: E402

FS = 12000.0


def tone(f_hz, amp, n, phase=0.0):
    t = np.arange(n)
    z = amp * np.exp(1j * (2 * np.pi * f_hz * t / FS + phase))
    return np.stack([np.rint(z.real), np.rint(z.imag)], axis=-1).astype(np.int16)


def consts_for(params):
    n = len(params)
    consts = np.zeros(n, twinlib.CONSTS_DTYPE)
    taps = np.zeros((n, 128), np.float32)
    for c, p in enumerate(params):
        k = O.compile_params(p)
        for f in ("mode", "ntap", "dphi1", "dphi2", "wf_cal_lin", "smeter_cal_db", "agc_c0", "agc_c1", "agc_knee",
                  "agc_delta8", "hang_frames", "fir_flags", "kfm"):
            consts[f][c] = k[f]
        consts["ntap8"][c] = (k["ntap"] + 7) // 8 * 8
        taps[c] = k["taps"]
    return consts, taps


# ------------------------------------------------------------------ waterfall
def test_wf_full_scale_tone_is_byte_255():
    k0 = 100
    iq = tone(k0 * FS / 1024, 32767.0, 1024)
    b = O.wf_line(iq)
    assert int(np.argmax(b)) == (k0 + 512) % 1024
    assert b[(k0 + 512) % 1024] in (254, 255)          # 32767/32768 -> -0.0003 dB
    assert b[(k0 + 511) % 1024] == b[(k0 + 513) % 1024] == 248        # Hann neighbours: -6.02 dB
    assert b[(k0 + 540) % 1024] < 150                  # far skirt


def test_wf_level_steps_one_byte_per_db():
    levels = []
    for amp_db in (-10.5, -11.5, -12.5, -40.5):
        iq = tone(37 * FS / 1024, 32768.0 * 10 ** (amp_db / 20), 1024)
        levels.append(int(O.wf_line(iq).max()))
    assert levels == [244, 243, 242, 214]              # floor(255 + dB)


def test_wf_negative_frequency_and_calibration():
    iq = tone(-3000.0, 8000.0, 1024)
    b = O.wf_line(iq)
    assert int(np.argmax(b)) == 512 - 256              # -3 kHz -> bin 256 (ascending frequency from -6 kHz)
    b7 = O.wf_line(iq, wf_cal_db=7.0)
    assert int(b7.max()) - int(b.max()) == 7


def test_wf_sum_and_mean():
    iq = O.synth_iq(1, 4 * 1024, seed=1)[0].reshape(4, 1024, 2)
    s = O.wf_sum_lines(iq, 4)
    lines = O.wf_line(iq).astype(np.float32)
    assert np.array_equal(O.wf_mean_from_sum(s[0], 4), np.mean(list(lines), axis=0))


# ------------------------------------------------------------------ audio
def run_oracle(p, iq):
    ch = O.AudioChannel(p)
    return ch.process(iq)


def spectrum_peak(x, lo=50.0):
    """(frequency of the strongest component, its amplitude by projection at that frequency)"""
    w = np.hanning(len(x))
    X = np.abs(np.fft.rfft(x * w, 8 * len(x)))
    f = np.fft.rfftfreq(8 * len(x), 1 / FS)
    X[f < lo] = 0
    fp = f[np.argmax(X)]
    amp = 2 * np.abs(np.sum(x * w * np.exp(-2j * np.pi * fp * np.arange(len(x)) / FS))) / w.sum()
    return fp, amp


def test_am_recovers_modulation_tone():
    n = 16 * 512
    t = np.arange(n)
    z = 8000 * (1 + 0.5 * np.sin(2 * np.pi * 1000 * t / FS)) * np.exp(2j * np.pi * 1500 * t / FS)
    iq = np.stack([np.rint(z.real), np.rint(z.imag)], -1).astype(np.int16)
    pcm, rssi = run_oracle(O.ChanParams("am", f_shift_hz=1500.0), iq)
    f, a = spectrum_peak(pcm[4096:].astype(float))
    assert abs(f - 1000) < 5
    # AGC pins the envelope peak (1.5 A) to 16384 -> the 0.5 A modulation comes out at 16384/3
    assert abs(a - 16384 / 3) / (16384 / 3) < 0.02
    # carrier power: 10 log10((8000^2 (1 + 0.5^2/2)) / 32768^2) - 13 dBm
    assert abs(rssi[-1] - (10 * np.log10(8000 ** 2 * 1.125 / 32768 ** 2) - 13)) < 0.05


def test_ssb_sideband_selection():
    n = 16 * 512
    up = tone(1000.0, 8000.0, n)        # +1 kHz: passes USB (30..3000), rejected by LSB
    usb, _ = run_oracle(O.ChanParams("usb", low_cut=30, high_cut=3000), up)
    lsb, _ = run_oracle(O.ChanParams("lsb", low_cut=-3000, high_cut=-30, agc_on=0, man_gain=50), up)
    usb_m, _ = run_oracle(O.ChanParams("usb", low_cut=30, high_cut=3000, agc_on=0, man_gain=50), up)
    f, a = spectrum_peak(usb[4096:].astype(float))
    assert abs(f - 1000) < 5 and abs(a - 16384) / 16384 < 0.02       # AGC target 0.5 FS
    f, a = spectrum_peak(usb_m[4096:].astype(float))
    assert abs(a - 8000) / 8000 < 0.01                               # manGain 50 dB = unity
    assert np.abs(lsb[4096:]).max() < 8000 * 10 ** (-60 / 20)        # > 60 dB opposite-sideband rejection


def test_cw_pitch():
    n = 24 * 512
    iq = tone(0.0, 4000.0, n)           # carrier at the tuned frequency + 0 -> with cw passband 400..800 centred 600
    pcm, _ = run_oracle(O.ChanParams("cw", f_shift_hz=-600.0, low_cut=400, high_cut=800, decay=1000), iq)
    f, _ = spectrum_peak(pcm[8192:].astype(float))
    assert abs(f - 600) < 5


def test_nbfm_deviation_scale():
    n = 8 * 512
    t = np.arange(n)
    dev = 2500.0
    ph = 2 * np.pi * np.cumsum(dev * np.sin(2 * np.pi * 400 * t / FS)) / FS
    z = 8000 * np.exp(1j * ph)
    iq = np.stack([np.rint(z.real), np.rint(z.imag)], -1).astype(np.int16)
    pcm, _ = run_oracle(O.ChanParams("nbfm"), iq)
    f, a = spectrum_peak(pcm[2048:].astype(float))
    assert abs(f - 400) < 5
    assert abs(a - 16384 * dev / 5000) / (16384 * dev / 5000) < 0.02  # 5 kHz deviation <-> 0.5 FS


def test_agc_decay_time_constant_and_hang():
    """after a 20 dB drop the gain recovers at 8.686 dB per `decay` ms; with hang it first holds"""
    n1, n2 = 8 * 512, 24 * 512
    iq = np.concatenate([tone(1000.0, 8000.0, n1), tone(1000.0, 800.0, n2)])
    for hang in (0, 1):
        p = O.ChanParams("usb", low_cut=30, high_cut=3000, decay=1000, hang=hang)
        ch = O.AudioChannel(p)
        ys = []
        for f in range((n1 + n2) // 512):
            _, _, y = ch.process_frame(iq[f * 512:(f + 1) * 512])
            ys.append(np.abs(y).max())
        ys = np.array(ys)
        assert abs(ys[7] - 16384) / 16384 < 0.02
        drop = 20 * np.log10(ys[9] / ys[7])              # right after the step: gain still low
        rec = 20 * np.log10(ys[9 + 12] / ys[9])          # 12 frames = 512 ms later
        if hang == 0:
            assert -20.5 < drop < -19.0
            assert abs(rec - 8.686 * 0.512) < 0.3
        else:
            assert -20.5 < drop < -19.5
            assert 20 * np.log10(ys[9 + 1] / ys[9]) < 0.05   # held (hang = 2 frames at decay 1000 ms)
            assert rec > 2.5


def test_agc_threshold_knee_and_slope():
    weak = tone(1000.0, 20.0, 12 * 512)                  # -64 dBFS = -77 dBm: below a -60 dBm knee
    p = O.ChanParams("usb", low_cut=30, high_cut=3000, thresh=-60)
    pcm, _ = run_oracle(p, weak)
    knee_amp = 32768 * 10 ** ((-60 + 13) / 20)
    want = 20.0 * 16384 / knee_amp                       # fixed max gain below the knee
    _, a = spectrum_peak(pcm[3072:].astype(float))
    assert abs(a - want) / want < 0.03
    strong = tone(1000.0, 8000.0, 12 * 512)
    p10 = O.ChanParams("usb", low_cut=30, high_cut=3000, slope=10)
    pcm10, _ = run_oracle(p10, strong)
    _, a10 = spectrum_peak(pcm10[3072:].astype(float))
    want10 = 16384 * (8000 / 32768) ** 0.1               # output rises slope/100 dB per dB
    assert abs(a10 - want10) / want10 < 0.02


def test_state_carry_is_sample_exact():
    iq = O.synth_iq(1, 6 * 512, seed=3, modes=[1])[0]
    p = O.ChanParams("usb", f_shift_hz=-1100.0, low_cut=30, high_cut=3000)
    a, ra = O.AudioChannel(p).process(iq)
    ch = O.AudioChannel(p)
    parts = [ch.process(iq[s:e])[0] for s, e in ((0, 512), (512, 2048), (2048, 3072))]
    assert np.array_equal(np.concatenate(parts), a)


# ------------------------------------------------------------------ twin vs oracle
def test_twin_tables_equal_oracle_tables(twin):
    assert np.array_equal(twin.win, O.hann_window())
    wr, wi = O.twiddles()
    assert np.array_equal(twin.wr, wr) and np.array_equal(twin.wi, wi)
    assert np.array_equal(twin.thr, O.db_thresholds())


def test_twin_helpers_accuracy(twin):
    rng = np.random.default_rng(0)
    ph = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64)
    cs = np.array([twin.sincos20(int(p)) for p in ph[:4000]])
    p20 = (ph[:4000] >> np.uint64(12)).astype(np.float64)             # the 20-bit evaluation on its own grid ...
    ref = np.exp(2j * np.pi * p20 / 2.0 ** 20)
    assert np.abs(cs[:, 0] - ref.real).max() < 2e-7 and np.abs(cs[:, 1] - ref.imag).max() < 2e-7
    p32 = np.array([twin.phasor32(int(p)) for p in ph[:4000]])         # ... and the NCO's phasor at all 32 bits of the phase
    ref = O.nco(0, 1, 0, 1)[0] * np.exp(2j * np.pi * ph[:4000].astype(np.float64) / 2.0 ** 32)
    assert np.abs(p32[:, 0] - ref.real).max() < 2.5e-7 and np.abs(p32[:, 1] - ref.imag).max() < 2.5e-7
    x = np.exp(rng.uniform(-20, 45, 4000)).astype(np.float32)
    l2 = np.array([twin.lib.twin_log2p(float(v)) for v in x])
    assert np.abs(l2 - np.log2(x.astype(np.float64))).max() < 8e-6      # fp32 resolution at |log2| ~ 64
    y = rng.uniform(-40, 30, 4000).astype(np.float32)
    e2 = np.array([twin.lib.twin_exp2p(float(v)) for v in y])
    assert np.abs(e2 / 2.0 ** y.astype(np.float64) - 1).max() < 3e-7
    a = rng.standard_normal((4000, 2)).astype(np.float32)
    at = np.array([twin.lib.twin_atan2p(float(u), float(v)) for u, v in a])
    assert np.abs(at - np.arctan2(a[:, 0].astype(np.float64), a[:, 1])).max() < 4e-7
    assert twin.lib.twin_atan2p(0.0, 0.0) == 0.0


def test_twin_quantiser_is_the_threshold_count(twin):
    T = O.db_thresholds()
    rng = np.random.default_rng(1)
    p = np.concatenate([np.exp(rng.uniform(-40, 40, 5000)), T.astype(np.float64), np.nextafter(T, 0).astype(np.float64),
                        [0.0, 1e30]]).astype(np.float32)
    got = np.array([twin.lib.twin_quantise(float(v), T.ctypes.data) for v in p])
    assert np.array_equal(got, O.wf_quantise(p.astype(np.float64)))


@pytest.mark.parametrize("seed,n_ch,n_lines,n_avg", [(1, 3, 2, 1), (2, 8, 6, 3)])
def test_twin_wf_vs_oracle(twin, seed, n_ch, n_lines, n_avg):
    iq = O.synth_iq(n_ch, n_lines * 1024, seed=seed)
    t = twin.wf(iq, n_avg)
    o = np.stack([O.wf_sum_lines(iq[c].reshape(-1, 1024, 2), n_avg) for c in range(n_ch)], axis=1)
    allowed = np.stack([O.wf_allowed_diff(iq[c].reshape(-1, 1024, 2)) for c in range(n_ch)], axis=1)
    L = allowed.shape[0] // n_avg
    allowed = allowed[: L * n_avg].reshape(L, n_avg, n_ch, 1024).sum(axis=1)
    d = np.abs(t.astype(np.int32) - o)
    assert not (d > allowed).any() and (d > 0).mean() < 1e-3


def test_twin_audio_vs_oracle_all_modes(twin):
    modes = ["am", "usb", "lsb", "nbfm", "cw"]
    pb = {"am": (-6000, 6000), "usb": (30, 3000), "lsb": (-3000, -30), "nbfm": (-6000, 6000), "cw": (400, 800)}
    iq = O.synth_iq(5, 6 * 512, seed=4, modes=[0, 1, 2, 3, 1])
    ps = [O.ChanParams(m, f_shift_hz=((c * 37) % 97 - 48) * 100.0, low_cut=pb[m][0], high_cut=pb[m][1],
                       hang=int(c == 1), decay=1000 if m == "cw" else 4000) for c, m in enumerate(modes)]
    pcm_o, rssi_o = O.audio_chain(iq, ps)
    consts, taps = consts_for(ps)
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t = twin.audio(iq, consts, taps, st, hist)
    rms = np.sqrt(((pcm_t.astype(np.float64) - pcm_o) ** 2).mean(axis=1)) / 32768
    assert rms.max() < 1e-5                                          # north_star audio tolerance
    assert np.abs(pcm_t.astype(np.int32) - pcm_o).max() <= 1
    assert np.abs(rssi_t - rssi_o).max() < 1e-4


# ------------------------------------------------------------------ random sweep over the parameter surface
@pytest.mark.parametrize("seed,n_frames", [(2024, 6), (1, 6), (3, 6), (19, 6), (7, 240)])     # 240 frames = 10 s: no drift
def test_twin_tracks_float64_oracle_over_random_parameters(seed, n_frames):
    """fp32 twin vs the normative float64 oracle on seeded random modes / passbands / AGC settings / levels:
    PCM within the north_star tolerance of full scale, RSSI within 1e-3 dB (where the frame is not silent)"""
    import random_params as RP
    rng = np.random.default_rng(seed)
    n_ch = 24
    kw = [RP.draw(rng) for _ in range(n_ch)]
    iq = RP.signal(rng, n_ch, n_frames * 512)
    params = [O.ChanParams(**k) for k in kw]
    consts, taps = consts_for(params)
    twin = twinlib.load()
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t = twin.audio(iq, consts, taps, st, hist)
    import tolerances as T                                   # the condition of the 1e-5 figure, stated once
    pcm_o, rssi_o, bound = T.oracle_with_bound(iq, params)
    well, _ = T.assert_pcm_within_tolerance(pcm_t, pcm_o, bound)
    rwell = T.rssi_well_conditioned(iq, rssi_o, [k["smeter_cal_db"] for k in kw])
    assert np.abs(rssi_t - rssi_o)[rwell].max() < 1e-3 and rwell.sum() >= T.min_well_for(n_ch)
    assert np.abs(rssi_t - rssi_o)[rssi_o > -150].max() < 2e-2
    assert len({k["mode"] for k in kw}) >= 4
    wf_t = twin.wf(iq[:, : (n_frames // 2) * 1024], 1, consts["wf_cal_lin"])
    for c in range(n_ch):
        lines = iq[c, : (n_frames // 2) * 1024].reshape(-1, 1024, 2)
        ref = O.wf_sum_lines(lines, 1, kw[c]["wf_cal_db"])
        guard = O.wf_guard_band(lines, kw[c]["wf_cal_db"])
        assert not ((wf_t[:, c] != ref) & ~guard).any(), c


@pytest.mark.parametrize("decim", [2, 4])
def test_decimating_front_end_twin_vs_oracle(twin, decim):
    """IQ at D * 12 kHz: the twin's polyphase-stream FIR (fp32, the kernels' order) against the oracle's plain
    np.convolve(z, h)[::D] (float64), all modes; and the filter does what a decimator must -- a carrier outside the
    12 kHz output band but inside the wide input band is gone from the AM output."""
    n_ch, n_frames = 6, 5
    rng = np.random.default_rng(decim)
    iq = O.synth_iq(n_ch, n_frames * 512 * decim, seed=80 + decim)
    modes = ["am", "usb", "lsb", "cw", "nbfm", "am"]
    prm = [O.ChanParams(mode=m, f_shift_hz=float(rng.integers(-5000, 5000)) * decim,
                        **({"low_cut": -3000.0, "high_cut": 3000.0} if m in ("am", "nbfm") else
                           {"low_cut": 300.0, "high_cut": 2700.0} if m == "usb" else
                           {"low_cut": -2700.0, "high_cut": -300.0} if m == "lsb" else {"low_cut": 400.0, "high_cut": 800.0}))
           for m in modes]
    consts = np.zeros(n_ch, twinlib.CONSTS_DTYPE)
    taps = np.zeros((n_ch, 128), np.float32)
    for c, p in enumerate(prm):
        k = O.compile_params(p, decim)
        for f in ("mode", "ntap", "ntap8", "dphi1", "dphi2", "wf_cal_lin", "smeter_cal_db", "agc_c0", "agc_c1", "agc_knee",
                  "agc_delta8", "hang_frames", "fir_flags", "decim", "kfm"):
            consts[f][c] = k[f]
        taps[c] = k["taps_streams"]
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t = twin.audio(iq, consts, taps, st, hist)
    pcm_o, rssi_o = O.audio_chain(iq, prm, decim)
    assert pcm_t.shape == (n_ch, n_frames * 512)
    rms = np.sqrt(((pcm_t.astype(np.float64) - pcm_o) ** 2).mean(axis=1)) / 32768.0
    assert rms.max() < 1e-5, rms
    assert np.abs(rssi_t - rssi_o).max() < 1e-3
    # frame by frame == all at once (history of 128 inputs, phases advancing by 512 D steps)
    st2, hist2 = twinlib.fresh_state(consts)
    parts = [twin.audio(iq[:, f * 512 * decim:(f + 1) * 512 * decim], consts, taps, st2, hist2)[0] for f in range(n_frames)]
    assert np.array_equal(np.concatenate(parts, axis=1), pcm_t) and st2.tobytes() == st.tobytes()
    # anti-aliasing: AM, passband +-3 kHz, tuned to 0; a strong carrier at 0.4 fs_in (outside +-6 kHz for every D) with 1 kHz AM
    n = np.arange(8 * 512 * decim)
    fs_in = 12000.0 * decim
    z = 12000.0 * (1 + 0.8 * np.sin(2 * np.pi * 1000.0 * n / fs_in)) * np.exp(2j * np.pi * 0.4 * fs_in * n / fs_in)
    x = np.stack([np.rint(z.real), np.rint(z.imag)], axis=-1).astype(np.int16)[None]
    p = O.ChanParams(mode="am", f_shift_hz=0.0, low_cut=-3000.0, high_cut=3000.0, agc_on=0, man_gain=50)
    out, rssi = O.audio_chain(x, [p], decim)
    assert np.abs(out[0, 2048:]).max() <= 2 and rssi[0, -1] < -70.0       # >= 70 dB down: the Blackman stop band
    # the hard case for a decimator -- what FOLDS INTO the passband.  Full-band AM default (+-6 kHz): a carrier at
    # 8 kHz lands on -4 kHz of the 12 kHz output (D = 2: 8 - 12; D = 4: the same), right inside the passband.  With the
    # interpolator's length rule (33 taps at D = 4) it came through ~25 dB down; with the whole tap budget the transition ends
    # at 6.5 / 7.1 kHz and the alias is >= 70 dB below an in-band carrier of the same amplitude.
    def level(f_hz, lc, hc):
        zz = 12000.0 * np.exp(2j * np.pi * f_hz * n / fs_in)
        xx = np.stack([np.rint(zz.real), np.rint(zz.imag)], axis=-1).astype(np.int16)[None]
        pp = O.ChanParams(mode="am", f_shift_hz=0.0, low_cut=lc, high_cut=hc, agc_on=0, man_gain=50)
        return float(O.audio_chain(xx, [pp], decim)[1][0, -1])
    assert level(2000.0, -6000.0, 6000.0) - level(8000.0, -6000.0, 6000.0) > 70.0
    assert level(1000.0, -3000.0, 3000.0) - level(12000.0 - 2500.0, -3000.0, 3000.0) > 70.0
    k = O.compile_params(O.ChanParams(mode="am"), decim)
    assert int(k["ntap"]) == (125 if decim == 4 else 127)


# ------------------------------------------------------------------ round 3: the IQ chain at 20.25 kHz
def test_wide_rate_chain_twin_vs_oracle_and_known_answers():
    """A three-channel KiwiSDR runs at 20.25 kHz (utils_supersdr.py:988-994): the same chain with every rate-dependent
    constant compiled for that rate -- NCO steps f / fs, the tap formula at fs, AGC decay per sample, the NBFM scale.
    Twin vs float64 oracle under the tolerance rule; and what the rate must do to a known signal."""
    import random_params as RP
    import tolerances as T
    rate = 20250
    rng = np.random.default_rng(77)
    n_ch, n_frames = 24, 6
    kw = [RP.draw(rng) for _ in range(n_ch)]
    iq = RP.signal(rng, n_ch, n_frames * 512)
    params = [O.ChanParams(**k) for k in kw]
    consts = np.zeros(n_ch, twinlib.CONSTS_DTYPE)
    taps = np.zeros((n_ch, 128), np.float32)
    for c, p in enumerate(params):
        k = O.compile_params(p, 1, rate)
        for f in ("mode", "ntap", "dphi1", "dphi2", "wf_cal_lin", "smeter_cal_db", "agc_c0", "agc_c1", "agc_knee", "agc_delta8",
                  "hang_frames", "fir_flags", "kfm"):
            consts[f][c] = k[f]
        consts["ntap8"][c] = (k["ntap"] + 7) // 8 * 8
        taps[c] = k["taps"]
    twin = twinlib.load()
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t = twin.audio(iq, consts, taps, st, hist)
    pcm_o, rssi_o, bound = T.oracle_with_bound(iq, params, 1, rate)
    T.assert_pcm_within_tolerance(pcm_t, pcm_o, bound)
    # constants: the rate is in them
    k12, k20 = O.compile_params(O.ChanParams("usb", f_shift_hz=1000.0)), O.compile_params(O.ChanParams("usb", f_shift_hz=1000.0), 1, rate)
    assert abs(int(k20["dphi1"]) / int(k12["dphi1"]) - 12000 / 20250) < 1e-6
    assert k20["ntap"] > k12["ntap"] and abs(float(k20["kfm"]) / float(k12["kfm"]) - 20250 / 12000) < 1e-6
    assert abs(float(k20["agc_delta8"]) / float(k12["agc_delta8"]) - 12000 / 20250) < 1e-6
    assert O.compile_params(O.ChanParams("am"), 1, rate)["fir_flags"] == 0          # +-6 kHz is no longer the whole band
    O.compile_params(O.ChanParams("usb", f_shift_hz=9000.0), 1, rate)               # inside +-10.125 kHz
    with pytest.raises(ValueError):
        O.compile_params(O.ChanParams("usb", f_shift_hz=10200.0), 1, rate)
    # NBFM: a carrier frequency-modulated with 2.5 kHz deviation at 500 Hz demodulates to 0.25 FS peak at EITHER rate
    for r in (12000, rate):
        n = np.arange(8 * 512)
        ph = (2500.0 / 500.0) * -np.cos(2 * np.pi * 500.0 * n / r)              # beta = deviation / f_mod = 5 rad
        z = 12000.0 * np.exp(1j * ph)
        x = np.stack([np.rint(z.real), np.rint(z.imag)], axis=-1).astype(np.int16)[None]
        out, _ = O.audio_chain(x, [O.ChanParams("nbfm", low_cut=-6000.0, high_cut=6000.0)], 1, r)
        peak = np.abs(out[0, 2048:].astype(np.float64)).max()
        assert abs(peak - 8192.0) < 8192.0 * 0.03, (r, peak)
    # USB: a tone 1 kHz above the tuned frequency comes out as a 1 kHz tone at the channel's own rate
    n = np.arange(8 * 512)
    z = 8000.0 * np.exp(2j * np.pi * 4000.0 * n / rate)
    x = np.stack([np.rint(z.real), np.rint(z.imag)], axis=-1).astype(np.int16)[None]
    out, _ = O.audio_chain(x, [O.ChanParams("usb", f_shift_hz=3000.0, low_cut=30.0, high_cut=3000.0)], 1, rate)
    seg = out[0, 1024:1024 + 2025].astype(np.float64)                                # 2025 samples = 100 cycles of 1 kHz at 20.25 kHz
    spec = np.abs(np.fft.rfft(seg * np.hanning(len(seg))))
    assert int(np.argmax(spec)) == 100


# ------------------------------------------------------------------ round 3: waterfall zoom
@pytest.mark.parametrize("Z", [2, 4, 8])
def test_zoom_stage_known_answers_and_twin_vs_oracle(twin, Z):
    """"SET zoom= start=" (utils_supersdr.py:741, 753-758, 839): a span of 1/Z of the IQ band around a zoom centre.  A tone
    d Hz above the centre lands d / (fs / Z) * 1024 bins right of bin 512 at its un-zoomed level; what lies outside the
    zoomed span beyond the filter's transition is >= 70 dB down; the fp32 twin stays within 1 LSB of the float64 oracle."""
    fs, n_in = 12000.0, 8 * 1024 * Z // Z * Z
    f0 = 1500.0
    span = fs / Z
    n = np.arange(n_in)

    def line_of(freq_hz, amp=8000.0):
        z = amp * np.exp(2j * np.pi * freq_hz * n / fs)
        x = np.stack([np.rint(z.real), np.rint(z.imag)], axis=-1).astype(np.int16)
        y = O.ZoomChannel(Z, f0, fs).process(x)
        return O.wf_line(y[-1024:].reshape(1, 1024, 2))[0]

    d = span * 0.25                                           # a quarter span above the centre
    ln = line_of(f0 + d)
    k = int(np.argmax(ln))
    assert abs(k - (512 + 256)) <= 1
    ref = O.wf_line(np.stack([np.rint(8000 * np.cos(2 * np.pi * 0.25 * np.arange(1024))), np.rint(8000 * np.sin(2 * np.pi * 0.25 * np.arange(1024)))],
                             axis=-1).astype(np.int16).reshape(1, 1024, 2))[0]
    assert abs(int(ln.max()) - int(ref.max())) <= 1           # unity pass-band gain: the level of the same tone un-zoomed
    out = line_of(f0 + 0.85 * span)                           # outside the span, beyond the transition: folds to 0.85 - 1 = -0.15 span
    assert int(out.max()) <= int(ln.max()) - 70
    edge = line_of(f0 + 0.5 * span - fs / 1024.0)             # at the edge of the span: the filter's -6 dB point
    assert 3 <= int(ln.max()) - int(edge.max()) <= 9
    # twin vs oracle on the zoomed stream itself, state carried over uneven calls
    iq = O.synth_iq(3, 6 * 1024 * Z, seed=40 + Z)
    offs = [f0, -2500.0, 0.0]
    dphi = np.array([O._dphi(f, fs) for f in offs], np.uint32)
    ph, hist = np.zeros(3, np.uint32), np.zeros((3, 256, 2), np.int16)
    cut = 2 * 1024 * Z
    got = np.concatenate([twin.zoom(iq[:, :cut], Z, dphi, O.zoom_taps(Z), ph, hist), twin.zoom(iq[:, cut:], Z, dphi, O.zoom_taps(Z), ph, hist)], axis=1)
    for c in range(3):
        zc = O.ZoomChannel(Z, offs[c], fs)
        o = np.concatenate([zc.process(iq[c, :cut]), zc.process(iq[c, cut:])])
        dd = np.abs(got[c].astype(np.int32) - o.astype(np.int32))
        assert dd.max() <= 1 and (dd > 0).mean() < 0.01
    assert len(O.zoom_taps(Z)) == 32 * Z - 1


@pytest.mark.parametrize("seed", [401, 402, 403])
def test_error_sensitivity_is_a_tight_bound(seed):
    """The tolerance rule (tests/tolerances.py) bounds each sample by what the oracle says a relative error EPS in its own filter sums
    does to its output (AudioChannel.error_sensitivity).  An over-estimate there would excuse anything.  Here the float64 chain is run
    twice on random receivers, once with its filter sums actually moved by EPS x (their dot-product bound) in random directions:
    the outputs never differ by more than the predicted bound (first order: 10 % of slack for the second), and the prediction is not
    loose -- in every channel the largest deviation reaches a good part of its bound."""
    import random_params as RP
    import tolerances as T
    rng = np.random.default_rng(seed)
    n_ch, n_frames = 16, 6
    iq = RP.signal(rng, n_ch, n_frames * 512)
    eps = T.EPS
    tight = []
    for c in range(n_ch):
        d = RP.draw(rng)
        p = O.ChanParams(**d)
        a, b = O.AudioChannel(p, want_sens=True), O.AudioChannel(p)
        b.perturb = (eps, np.random.default_rng(seed * 100 + c))
        a.process(iq[c]); b.process(iq[c])
        sens, cap, margin = a.sens_out.reshape(3, -1)
        bound = np.where(eps * sens >= margin, cap, np.minimum(eps * sens, cap))
        dy = np.abs(a.y_out.reshape(-1) - b.y_out.reshape(-1))
        live = (bound > 1e-9) & (np.abs(a.y_out.reshape(-1)) < 32000)           # (silent channels: nothing to perturb; clipped samples: both clip)
        live[:128] = False      # the filter filling up from a reset: the discriminator's first few outputs (hundredths of an LSB) move by up to 4x their
                                # first-order bound there -- far inside the rule's 1 LSB of rounding allowance, and the bound's only miss
        if live.sum() < 100:
            continue
        assert (dy[live] <= 1.1 * bound[live] + 1e-6).all(), (c, d["mode"], float((dy[live] / bound[live]).max()))
        tight.append((d["mode"], float((dy[live] / bound[live]).max()), float(np.sqrt((dy[live] ** 2).mean()) / np.sqrt((bound[live] ** 2).mean()))))
    assert len(tight) >= 8
    # with random directions the linear pieces reach their bound to within a factor of a few somewhere in 3072 samples
    assert min(t[1] for t in tight) > 0.15, sorted(tight, key=lambda t: t[1])[:3]
    # (in RMS the bound sits further above a random perturbation: its AGC term is a worst case over the follower's whole memory)
    print("error_sensitivity: largest share of its bound a perturbed sample reaches, per channel: min %.2f median %.2f; RMS(deviation) / RMS(bound): min %.3f median %.2f"
          % (min(t[1] for t in tight), float(np.median([t[1] for t in tight])), min(t[2] for t in tight), float(np.median([t[2] for t in tight]))))
