"""GPU parity tests: the HIP path (through the C-ABI) against the oracle.

  * bit-exact vs the fp32 twin (oracle/ssdr_twin.c) on every int16 bin and PCM sample
  * vs the normative NumPy float64 oracle: waterfall bins identical outside the
    threshold guard band (bins whose |X| is within the fp32 FFT error bound of a 1-dB
    threshold, ssdr_oracle.wf_allowed_diff) and bounded inside it; PCM within 1e-5 RMS
    of full scale (north_star tolerance)
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ssdr_oracle as O  # noqa: E402
import twinlib  # noqa: E402

pytestmark = pytest.mark.gpu

PCM_RMS_TOL = 1e-5          # of full scale, north_star
MODES = ["am", "usb", "lsb", "nbfm"]


@pytest.fixture(scope="module")
def S():
    import supersdr_amd
    return supersdr_amd


def oracle_wf(iq, n_avg, cal_db=0.0):
    n_ch = iq.shape[0]
    out = np.stack([O.wf_sum_lines(iq[c].reshape(-1, 1024, 2), n_avg, cal_db) for c in range(n_ch)], axis=1)
    return out


def oracle_guard(iq, n_avg):
    n_ch = iq.shape[0]
    g = np.stack([O.wf_allowed_diff(iq[c].reshape(-1, 1024, 2)) for c in range(n_ch)], axis=1)  # [lines, ch, 1024]
    L = g.shape[0] // n_avg
    return g[: L * n_avg].reshape(L, n_avg, n_ch, 1024).sum(axis=1)                              # steps allowed per summed bin


def mixed_params(S, n_ch, **over):
    ps, ops = [], []
    for c in range(n_ch):
        m = MODES[c % 4]
        fc = ((c * 37) % 97 - 48) * 100.0
        p = S.default_params(m, f_shift_hz=fc, **over)
        ps.append(p)
        ops.append(O.ChanParams(mode=m, f_shift_hz=fc, low_cut=p.low_cut, high_cut=p.high_cut, agc_on=p.agc_on,
                                hang=p.agc_hang, thresh=p.agc_thresh, slope=p.agc_slope, decay=p.agc_decay,
                                man_gain=p.agc_man_gain, wf_cal_db=p.wf_cal_db, smeter_cal_db=p.smeter_cal_db))
    return ps, ops


def test_quantiser_exhaustive(S):
    """every positive finite float32 power: bit-pattern estimate + one compare == binary search"""
    with S.SsdrEngine(1) as eng:
        assert eng.selftest_quantiser() == 0


def test_sqrt_exhaustive(S):
    """the AM envelope's sqrt (hardware estimate + residual correction) == IEEE sqrtf on every normal float"""
    with S.SsdrEngine(1) as eng:
        assert eng.selftest_sqrt() == 0


def test_sqrt_against_an_independent_ieee_sqrt(S):
    """the exhaustive self-test compares with the device's own sqrtf; here 4 M arguments -- random floats over the whole
    domain, every integer power I*I + Q*Q of a dense set of (I, Q), perfect squares and their neighbours -- against NumPy's
    correctly rounded float32 sqrt on the host"""
    rng = np.random.default_rng(7)
    bits = rng.integers(1, 0x5E800000, 1 << 21, dtype=np.uint32)             # positive floats below 2^62, denormals included
    x = bits.view(np.float32)
    i, q = rng.integers(-32768, 32768, (2, 1 << 20))
    pw = (i.astype(np.int64) ** 2 + q.astype(np.int64) ** 2).astype(np.float32)
    k = rng.integers(1, 46340, 1 << 18).astype(np.int64)
    sq = np.concatenate([(k * k).astype(np.float32), (k * k + 1).astype(np.float32), (k * k - 1).clip(1).astype(np.float32),
                         np.array([0.0, 1.0, 2.0, 2.0 ** 31, 2.0 ** 32], np.float32)])
    with S.SsdrEngine(1) as eng:
        a, _ = eng.sqrt_values(x)
        assert np.array_equal(a.view(np.uint32), np.sqrt(x).view(np.uint32))
        for arg in (pw, sq):
            a, b = eng.sqrt_values(arg)
            ref = np.sqrt(arg).view(np.uint32)
            assert np.array_equal(a.view(np.uint32), ref) and np.array_equal(b.view(np.uint32), ref)


def test_tables_match_oracle(S, twin):
    assert np.array_equal(S.table(0), O.hann_window())
    wr, wi = O.twiddles()
    assert np.array_equal(S.table(1), wr) and np.array_equal(S.table(2), wi)
    assert np.array_equal(S.table(3), O.db_thresholds())
    assert np.array_equal(S.table(0), twin.win) and np.array_equal(S.table(3), twin.thr)


@pytest.mark.parametrize("n_ch,n_lines,n_avg", [(1, 1, 1), (2, 3, 1), (7, 4, 2), (64, 6, 3), (33, 10, 10)])
def test_wf_bit_exact_vs_twin_and_oracle(S, twin, n_ch, n_lines, n_avg):
    iq = O.synth_iq(n_ch, n_lines * 1024, seed=11 + n_ch)
    with S.SsdrEngine(n_ch) as eng:
        eng.set_averaging(n_avg)
        eng.push_iq(iq)
        wf = eng.run_wf()
    ref_t = twin.wf(iq, n_avg)
    assert wf.shape == ref_t.shape == (n_lines // n_avg, n_ch, 1024)
    assert np.array_equal(wf, ref_t)
    ref_o = oracle_wf(iq, n_avg)
    guard = oracle_guard(iq, n_avg)
    diff = np.abs(wf.astype(np.int32) - ref_o)
    assert not (diff > guard).any()            # identical outside the guard band, bounded inside it
    assert (diff > 0).mean() < 1e-3            # and actual flips are a sliver (expected ~3e-4 of bins)


def test_wf_known_answer_full_scale_tone(S):
    """full-scale complex tone at FFT bin 100 -> byte 255 at shifted bin 612; bins away from it far below"""
    n = np.arange(1024)
    z = 32767.0 * np.exp(2j * np.pi * 100 * n / 1024)
    iq = np.stack([np.rint(z.real), np.rint(z.imag)], axis=-1).astype(np.int16)[None]
    with S.SsdrEngine(1) as eng:
        eng.push_iq(iq)
        wf = eng.run_wf()[0, 0]
    assert wf[612] in (254, 255) and int(np.argmax(wf)) == 612
    assert wf[611] == wf[613] and 248 <= wf[611] <= 249          # Hann side bins: -6.02 dB
    assert wf[100] < 180


def test_wf_carry_across_calls(S, twin):
    """averaging groups that straddle pushes: 3 pushes of 2,3,5 lines with N=4 == one push of 10 lines"""
    n_ch = 5
    iq = O.synth_iq(n_ch, 10 * 1024, seed=5)
    with S.SsdrEngine(n_ch) as eng:
        eng.set_averaging(4)
        outs = []
        for a, b in ((0, 2), (2, 5), (5, 10)):
            eng.push_iq(iq[:, a * 1024: b * 1024])
            outs.append(eng.run_wf())
    got = np.concatenate(outs, axis=0)
    assert [o.shape[0] for o in outs] == [0, 1, 1]
    assert np.array_equal(got, twin.wf(iq, 4))


def test_wf_calibration_and_edges(S, twin):
    """wf_cal_db shifts bytes; zero input -> byte 0; int16 extremes do not overflow"""
    n_ch = 4
    iq = O.synth_iq(n_ch, 1024, seed=3)
    iq[1] = 0
    iq[2, :, 0] = 32767
    iq[2, :, 1] = -32768
    ps = [S.default_params("am", wf_cal_db=db) for db in (0.0, 0.0, 0.0, 7.0)]
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.push_iq(iq)
        wf = eng.run_wf()
        consts, _ = eng.get_consts()
    assert np.array_equal(wf, twin.wf(iq, 1, consts["wf_cal_lin"]))
    assert (wf[0, 1] == 0).all()
    assert wf[0, 2].max() == 255                               # DC bin of a constant full-scale input, clamped
    o3 = O.wf_sum_lines(iq[3].reshape(-1, 1024, 2), 1, 7.0)
    g3 = O.wf_guard_band(iq[3].reshape(-1, 1024, 2), 7.0)
    assert not ((wf[:, 3] != o3) & ~g3).any()


def test_wf_calibration_extremes(S, twin):
    """The quantiser folds 2^-48 into the calibration factor and clamps with the multiply: the whole allowed calibration
    range (+-200 dB) gives the twin's bytes -- next to all 0 at the bottom, next to all 255 at the top -- and anything beyond is refused."""
    dbs = (-200.0, -120.0, -60.0, 0.0, 60.0, 120.0, 200.0)
    iq = O.synth_iq(len(dbs), 2 * 1024, seed=77)
    with S.SsdrEngine(len(dbs)) as eng:
        eng.set_params(0, [S.default_params("am", wf_cal_db=db) for db in dbs])
        eng.push_iq(iq)
        wf = eng.run_wf()
        consts, _ = eng.get_consts()
        assert np.array_equal(wf, twin.wf(iq, 1, consts["wf_cal_lin"]))
        means = wf.reshape(2, len(dbs), -1).mean(axis=(0, 2))
        assert (np.diff(means) >= 0).all() and means[0] < means[3] < means[-1], means
        for db in (200.5, -201.0, float("nan")):
            with pytest.raises(S.SsdrError):
                eng.set_params(0, [S.default_params("am", wf_cal_db=db)])


@pytest.mark.parametrize("n_ch,n_frames", [(1, 1), (4, 2), (16, 5), (67, 3), (2, 64), (3, 130)])   # > 64 frames: RSSI is converted in batches of 64
def test_audio_bit_exact_vs_twin_and_oracle(S, twin, n_ch, n_frames):
    iq = O.synth_iq(n_ch, n_frames * 512, seed=21 + n_ch)
    ps, ops = mixed_params(S, n_ch)
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.push_iq(iq)
        pcm, rssi = eng.run_audio()
        consts, taps = eng.get_consts()
        st_g, hist_g = eng.get_state()
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t = twin.audio(iq, consts, taps, st, hist)
    assert np.array_equal(pcm, pcm_t)
    assert np.array_equal(rssi, rssi_t)
    assert st_g.tobytes() == st.tobytes() and np.array_equal(hist_g, hist)
    pcm_o, rssi_o = O.audio_chain(iq, ops)
    rms = np.sqrt(((pcm.astype(np.float64) - pcm_o) ** 2).mean(axis=1)) / 32768.0
    assert rms.max() < PCM_RMS_TOL
    assert np.abs(rssi - rssi_o).max() < 1e-3


def test_one_very_long_call(S, twin):
    """a single call of 1000 frames (85 s of IQ) on a few channels: the frame loops, the NCO's 64-frame table refresh, the RSSI
    conversion in batches and the waterfall's group runs (hop 512, N = 7: 1000 lines, 142 groups and a remainder that carries)
    all cross their internal boundaries many times inside ONE launch; bit-exact vs the twin, PCM within tolerance of the oracle"""
    n_ch, n_frames = 5, 1000
    iq = O.synth_iq(n_ch, n_frames * 512, seed=1001)
    ps, ops = mixed_params(S, n_ch)
    ps[4], ops[4] = S.default_params("am"), O.ChanParams("am")                # one channel on the full-band AM path too
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.set_hop(512)
        eng.set_averaging(7)
        eng.push_iq(iq)
        lines, fused = eng.run_chain()
        wf = eng.fetch_wf(lines).copy()
        pcm, rssi = eng.fetch_audio()
        flags = eng.audio_flags()
        consts, taps = eng.get_consts()
        st_g, hist_g = eng.get_state()
    assert not fused and lines == 1000 // 7
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t, flags_t = twin.audio(iq, consts, taps, st, hist, want_flags=True)
    assert np.array_equal(pcm, pcm_t) and np.array_equal(rssi, rssi_t) and np.array_equal(flags, flags_t)
    assert st_g.tobytes() == st.tobytes() and np.array_equal(hist_g, hist)
    for c in range(n_ch):                                                        # hop 512: silence in front of the first half-line
        stream = np.concatenate([np.zeros((512, 2), np.int16), iq[c]])
        b = O.wf_lines_hop(stream, 512, 0.0).astype(np.int16)[:1000]
        ref = b[: lines * 7].reshape(lines, 7, 1024).sum(axis=1)
        seg = stream[np.arange(lines * 7)[:, None] * 512 + np.arange(1024)[None, :]]
        gb = O.wf_allowed_diff(seg).reshape(lines, 7, 1024).sum(axis=1)
        assert not (np.abs(wf[:, c].astype(np.int32) - ref) > gb).any(), c
    import tolerances as T
    pcm_o, rssi_o, bound = T.oracle_with_bound(iq, ops)
    T.assert_pcm_within_tolerance(pcm, pcm_o, bound, what="one 1000-frame call")


def test_full_band_paths_mode_switch_and_adc_overflow_flags(S, twin):
    """The reference's full-band passband (+-6 kHz at 12 kHz: the filter is a 4-sample delay) takes the kernel's shift
    paths -- AM without NCO and FIR (integer power), NBFM with a lane shift instead of the FIR.  PCM, RSSI, carried
    state and the per-frame ADC-overflow flags are bit-exact vs the twin; the modes swap mid-stream (the discriminator
    memory left by the AM path is the twin's); samples at the rails are flagged in exactly their frame, also the last
    four of a frame and the first of the next."""
    n_ch, n_frames = 12, 8
    iq = O.synth_iq(n_ch, n_frames * 512, seed=404)
    modes = ["am", "nbfm", "usb"] * 4
    iq[0, 1 * 512 + 511, 0] = 32767
    iq[1, 2 * 512 + 508, 1] = -32768
    iq[2, 3 * 512 + 0, 0] = -32767
    iq[3, 4 * 512 + 100, 1] = 32766                      # one below the rail: no flag
    iq[4, 0, 0] = iq[4, 0, 1] = -32768                   # I*I + Q*Q = 2^31
    iq[5, 7 * 512 + 509, 1] = 32767
    iq[6, 5 * 512 + 3, 0] = 32767
    want_flags = (np.abs(iq.astype(np.int32)).reshape(n_ch, n_frames, -1).max(axis=2) >= 32767).astype(np.uint8)
    assert want_flags.sum() == 6

    def params(ms):
        return [S.default_params(m, f_shift_hz=((c * 37) % 97 - 48) * 100.0) for c, m in enumerate(ms)]

    swapped = [{"am": "nbfm", "nbfm": "am", "usb": "usb"}[m] for m in modes]
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, params(modes))
        eng.push_iq(iq[:, : 3 * 512])
        p1, r1 = eng.run_audio()
        f1 = eng.audio_flags()
        c1, t1 = eng.get_consts()
        eng.set_params(0, params(swapped))
        eng.push_iq(iq[:, 3 * 512:])
        p2, r2 = eng.run_audio()
        f2 = eng.audio_flags()
        c2, t2 = eng.get_consts()
        st_g, hist_g = eng.get_state()
    assert (c1["fir_flags"][[0, 1]] == 1).all() and c1["fir_flags"][2] == 0
    st, hist = twinlib.fresh_state(c1)
    q1, s1, g1 = twin.audio(iq[:, : 3 * 512], c1, t1, st, hist, want_flags=True)
    q2, s2, g2 = twin.audio(iq[:, 3 * 512:], c2, t2, st, hist, want_flags=True)
    assert np.array_equal(p1, q1) and np.array_equal(p2, q2)
    assert np.array_equal(r1, s1) and np.array_equal(r2, s2)
    assert st_g.tobytes() == st.tobytes() and np.array_equal(hist_g, hist)
    got = np.concatenate([f1, f2], axis=1)
    assert np.array_equal(got, np.concatenate([g1, g2], axis=1)) and np.array_equal(got, want_flags)
    ops = [O.ChanParams(mode=m, f_shift_hz=((c * 37) % 97 - 48) * 100.0,
                        **({"low_cut": 30.0, "high_cut": 3000.0} if m == "usb" else {})) for c, m in enumerate(modes)]
    pcm_o, _ = O.audio_chain(iq[:, : 3 * 512], ops)
    rms = np.sqrt(((p1.astype(np.float64) - pcm_o) ** 2).mean(axis=1)) / 32768.0
    assert rms.max() < PCM_RMS_TOL


def test_untuned_channels_skip_the_mixer_bit_exactly(S, twin):
    """f_shift = 0 (and, for SSB, a passband centred on 0): the NCO's phasors are exactly (1, 0), the kernel converts
    instead of mixing -- same bits as the twin, which mixes; a retune mid-stream leaves the path, a retune back with
    a phase that is no longer zero does not re-enter it."""
    n_ch, n_frames = 6, 6
    iq = O.synth_iq(n_ch, n_frames * 512, seed=66)
    iq[2, 700, 1] = 32767
    ps = [S.default_params("am", f_shift_hz=0.0, low_cut=-3000.0, high_cut=3000.0),
          S.default_params("nbfm", f_shift_hz=0.0, low_cut=-4000.0, high_cut=4000.0),
          S.default_params("nbfm", f_shift_hz=0.0),                                  # full band: the lane-shift path
          S.default_params("usb", f_shift_hz=0.0, low_cut=-1500.0, high_cut=1500.0),  # centred passband: both NCOs idle
          S.default_params("am", f_shift_hz=100.0, low_cut=-3000.0, high_cut=3000.0),
          S.default_params("cw", f_shift_hz=-600.0)]                                 # f_shift + f_bc = 0, the re-mixer runs
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.push_iq(iq[:, : 2 * 512])
        p1, r1 = eng.run_audio()
        f1 = eng.audio_flags()
        c1, t1 = eng.get_consts()
        eng.set_params(0, [S.default_params("am", f_shift_hz=250.0, low_cut=-3000.0, high_cut=3000.0)])
        eng.push_iq(iq[:, 2 * 512: 4 * 512])
        p2, r2 = eng.run_audio()
        c2, t2 = eng.get_consts()
        eng.set_params(0, [ps[0]])                                                   # back to 0 Hz, phase now non-zero
        eng.push_iq(iq[:, 4 * 512:])
        p3, r3 = eng.run_audio()
        st_g, hist_g = eng.get_state()
    assert [int(x) for x in c1["dphi1"]][:4] == [0, 0, 0, 0] and int(c1["dphi1"][5]) == 0 and int(c1["dphi1"][4]) != 0
    st, hist = twinlib.fresh_state(c1)
    q1, s1, g1 = twin.audio(iq[:, : 2 * 512], c1, t1, st, hist, want_flags=True)
    q2, s2 = twin.audio(iq[:, 2 * 512: 4 * 512], c2, t2, st, hist)
    q3, s3 = twin.audio(iq[:, 4 * 512:], c1, t1, st, hist)
    assert np.array_equal(p1, q1) and np.array_equal(r1, s1) and np.array_equal(f1, g1) and f1[2, 1] == 1
    assert np.array_equal(p2, q2) and np.array_equal(p3, q3) and np.array_equal(r3, s3)
    assert st_g.tobytes() == st.tobytes() and np.array_equal(hist_g, hist) and int(st["phi1"][0]) != 0


def test_audio_state_carry_across_calls(S, twin):
    """frame-by-frame pushes == one multi-frame push (FIR history, NCO phase, DC, AGC all carried)"""
    n_ch, n_frames = 8, 6
    iq = O.synth_iq(n_ch, n_frames * 512, seed=77)
    ps, _ = mixed_params(S, n_ch)
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.push_iq(iq)
        pcm_all, rssi_all = eng.run_audio()
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        parts, rparts = [], []
        for a, b in ((0, 1), (1, 3), (3, 6)):
            eng.push_iq(iq[:, a * 512: b * 512])
            p, r = eng.run_audio()
            parts.append(p)
            rparts.append(r)
    assert np.array_equal(np.concatenate(parts, axis=1), pcm_all)
    assert np.array_equal(np.concatenate(rparts, axis=1), rssi_all)


@pytest.mark.parametrize("over", [dict(agc_hang=1), dict(agc_on=0, agc_man_gain=56.0), dict(agc_slope=6.0),
                                  dict(agc_thresh=-30.0), dict(agc_decay=400.0)])
def test_audio_agc_variants(S, twin, over):
    n_ch, n_frames = 8, 12
    iq = O.synth_iq(n_ch, n_frames * 512, seed=5)
    iq[:, 3 * 512: 6 * 512] //= 16            # level step down and back up: exercises decay / hang / knee
    ps, ops = mixed_params(S, n_ch, **over)
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.push_iq(iq)
        pcm, rssi = eng.run_audio()
        consts, taps = eng.get_consts()
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t = twin.audio(iq, consts, taps, st, hist)
    assert np.array_equal(pcm, pcm_t) and np.array_equal(rssi, rssi_t)
    pcm_o, _ = O.audio_chain(iq, ops)
    rms = np.sqrt(((pcm.astype(np.float64) - pcm_o) ** 2).mean(axis=1)) / 32768.0
    assert rms.max() < PCM_RMS_TOL


def test_audio_cw_long_filter_and_passband_change(S, twin):
    """CW: 127-tap filter (max history); then a passband change mid-stream keeps state"""
    n_ch, n_frames = 3, 4
    iq = O.synth_iq(n_ch, n_frames * 512, seed=9, modes=[1, 1, 1])
    ps = [S.default_params("cw", f_shift_hz=((c * 37) % 97 - 48) * 100.0 + 400.0) for c in range(n_ch)]
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.push_iq(iq[:, : 2 * 512])
        p1, _ = eng.run_audio()
        c1, t1 = eng.get_consts()
        ps2 = [S.default_params("usb", f_shift_hz=p.f_shift_hz) for p in ps]
        eng.set_params(0, ps2)
        eng.push_iq(iq[:, 2 * 512:])
        p2, _ = eng.run_audio()
        c2, t2 = eng.get_consts()
    assert (c1["ntap"] == 127).all()
    st, hist = twinlib.fresh_state(c1)
    q1, _ = twin.audio(iq[:, : 2 * 512], c1, t1, st, hist)
    q2, _ = twin.audio(iq[:, 2 * 512:], c2, t2, st, hist)
    assert np.array_equal(p1, q1) and np.array_equal(p2, q2)


def test_synth_input_roundtrip_and_parity(S, twin):
    """device-generated bench input: read it back, run both stages, compare with the twin on the same bytes"""
    n_ch, n_frames = 32, 4
    with S.SsdrEngine(n_ch) as eng:
        ps, _ = mixed_params(S, n_ch)
        eng.set_params(0, ps)
        eng.set_averaging(2)
        eng.synth_iq(n_frames, seed=1234)
        iq = eng.read_input()
        wf = eng.run_wf()
        pcm, rssi = eng.run_audio()
        consts, taps = eng.get_consts()
    assert iq.std() > 1000 and np.abs(iq).max() < 14000
    assert np.array_equal(wf, twin.wf(iq, 2, consts["wf_cal_lin"]))
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t = twin.audio(iq, consts, taps, st, hist)
    assert np.array_equal(pcm, pcm_t) and np.array_equal(rssi, rssi_t)


@pytest.mark.parametrize("seed", [31, 32, 33, 34])
def test_wf_random_batching_and_averaging(S, twin, seed):
    """random N in 1..100 and random ragged pushes (1..9 lines each): the lines that come out, in order, are the twin's
    N-line sums of the whole stream; nothing is lost or duplicated at call boundaries"""
    rng = np.random.default_rng(seed)
    n_ch = int(rng.integers(1, 12))
    n_avg = int(rng.choice([1, 2, 3, 5, 10, 37, 100]))
    total = int(rng.integers(n_avg, n_avg * 3 + 20))
    iq = O.synth_iq(n_ch, total * 1024, seed=seed)
    cal = rng.integers(-10, 11, n_ch).astype(np.float64)
    outs, pos = [], 0
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, [S.default_params("am", wf_cal_db=float(cal[c])) for c in range(n_ch)])
        eng.set_averaging(n_avg)
        while pos < total:
            k = int(min(total - pos, rng.integers(1, 10)))
            eng.push_iq(iq[:, pos * 1024:(pos + k) * 1024])
            outs.append(eng.run_wf())
            pos += k
        consts, _ = eng.get_consts()
    got = np.concatenate(outs, axis=0)
    ref = twin.wf(iq, n_avg, consts["wf_cal_lin"])
    assert got.shape == ref.shape == (total // n_avg, n_ch, 1024) and np.array_equal(got, ref)


@pytest.mark.parametrize("seed,n_frames", [(11, 6), (12, 6), (13, 6), (14, 48)])
def test_random_parameter_surface_bit_exact_vs_twin(S, twin, seed, n_frames):
    """seeded random modes / passbands (down to 50 Hz CW: 127 taps) / AGC laws / calibrations / levels incl. silence and
    bursts at the rails, 96 channels x 6 frames: PCM, RSSI, carried state, FIR history and waterfall bit-exact vs the twin"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random_params as RP
    rng = np.random.default_rng(seed)
    n_ch = 96
    kw = [RP.draw(rng) for _ in range(n_ch)]
    iq = RP.signal(rng, n_ch, n_frames * 512)
    ps = [S.default_params(k["mode"], f_shift_hz=k["f_shift_hz"], low_cut=k["low_cut"], high_cut=k["high_cut"],
                           agc_on=k["agc_on"], agc_hang=k["hang"], agc_thresh=k["thresh"], agc_slope=k["slope"],
                           agc_decay=k["decay"], agc_man_gain=k["man_gain"], wf_cal_db=k["wf_cal_db"],
                           smeter_cal_db=k["smeter_cal_db"]) for k in kw]
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        wfs, ps_, rs, fl, pos = [], [], [], [], 0
        while pos < n_frames:                                    # several calls: state crosses call boundaries
            k = min(n_frames - pos, 2 * int(rng.integers(1, 6)))
            eng.push_iq(iq[:, pos * 512:(pos + k) * 512])
            wfs.append(eng.run_wf())
            p, r = eng.run_audio()
            ps_.append(p)
            rs.append(r)
            fl.append(eng.audio_flags())
            pos += k
        consts, taps = eng.get_consts()
        st_g, hist_g = eng.get_state()
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t, flags_t = twin.audio(iq, consts, taps, st, hist, want_flags=True)
    assert np.array_equal(np.concatenate(ps_, axis=1), pcm_t)
    assert np.array_equal(np.concatenate(rs, axis=1), rssi_t)
    assert np.array_equal(np.concatenate(fl, axis=1), flags_t)
    assert st_g.tobytes() == st.tobytes() and np.array_equal(hist_g, hist)
    assert np.array_equal(np.concatenate(wfs), twin.wf(iq, 1, consts["wf_cal_lin"]))
    # ... and against the normative float64 definition, under the rule of tests/tolerances.py: every sample of every channel
    # within the oracle's own propagated bound, EVERY well-conditioned channel within 1e-5 RMS of full scale untrimmed
    import tolerances as T
    pcm_o, rssi_o, bound = T.oracle_with_bound(iq, [O.ChanParams(**k) for k in kw])
    T.assert_pcm_within_tolerance(np.concatenate(ps_, axis=1), pcm_o, bound)
    # the host-compiled constants are the oracle's
    for c in (0, 17, 95):
        k = O.compile_params(O.ChanParams(**kw[c]))
        assert int(consts["dphi1"][c]) == int(k["dphi1"]) and int(consts["ntap"][c]) == int(k["ntap"])
        assert np.array_equal(taps[c], k["taps"]) and consts["agc_knee"][c] == np.float32(k["agc_knee"])


def test_pipelined_feed_equals_push_and_run(S):
    """ssdr_feed_*: batches in flight on three streams give the results of push_iq + run_wf + run_audio in order
    (state, FIR history and partial waterfall sums carried), also when slots are reused and N does not divide a batch"""
    n_ch, nf, n_batches = 6, 4, 7
    iq = O.synth_iq(n_ch, n_batches * nf * 512, seed=91)
    ps, _ = mixed_params(S, n_ch)
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.set_averaging(3)
        want = []
        for b in range(n_batches):
            eng.push_iq(iq[:, b * nf * 512:(b + 1) * nf * 512])
            wf = eng.run_wf()
            pcm, rssi = eng.run_audio()
            want.append((wf.copy(), pcm.copy(), rssi.copy()))
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.set_averaging(3)
        eng.feed_open(nf, depth=3)
        got, sub = [], 0
        for b in range(n_batches):
            if sub - len(got) == 3:                              # every slot in flight: the next slot must be refused
                with pytest.raises(S.SsdrError):
                    eng.feed_slot()
                got.append(tuple(x.copy() for x in eng.feed_collect()))
            eng.feed_slot()[:] = iq[:, b * nf * 512:(b + 1) * nf * 512]
            eng.feed_submit()
            sub += 1
        while len(got) < n_batches:
            got.append(tuple(x.copy() for x in eng.feed_collect()))
        with pytest.raises(S.SsdrError):
            eng.feed_collect()
        eng.feed_close()
        eng.push_iq(iq[:, :nf * 512])                            # the ordinary path still works afterwards
        eng.run_audio()
    for b in range(n_batches):
        for k in range(3):
            assert np.array_equal(got[b][k], want[b][k]), (b, k)


def test_pipelined_feed_through_the_fused_kernel_equals_the_two_kernels(S):
    """The feed runs its batches through ssdr_run_chain: with every channel on the full-band AM path, N = 1 and 8-frame batches
    that is the fused superframe kernel.  Its waterfall lines, PCM, RSSI and ADC flags, batch after batch with the state
    carried, equal what the two per-stage kernels give synchronously (odd channel count: a half-empty wave; one clipping
    sample; one channel with the AGC hang)."""
    n_ch, nf, n_batches = 5, 8, 4
    iq = O.synth_iq(n_ch, n_batches * nf * 512, seed=313)
    iq[3, 9 * 512 + 17, 0] = 32767
    ps = [S.default_params("am", f_shift_hz=100.0 * c, agc_hang=int(c == 2)) for c in range(n_ch)]
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.set_fused(False)
        want = []
        for b in range(n_batches):
            eng.push_iq(iq[:, b * nf * 512:(b + 1) * nf * 512])
            lines, fused = eng.run_chain()
            assert not fused
            wf = eng.fetch_wf(lines)
            pcm, rssi = eng.fetch_audio()
            want.append((wf, pcm, rssi, eng.audio_flags()))
    assert sum(int(w[3].sum()) for w in want) == 1
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.set_profiling(True)
        eng.feed_open(nf, depth=2)
        got = []
        for b in range(n_batches):
            if b >= 2:
                got.append(tuple(x.copy() for x in eng.feed_collect()) + (eng.feed_flags.copy(),))
            eng.feed_slot()[:] = iq[:, b * nf * 512:(b + 1) * nf * 512]
            eng.feed_submit()
        while len(got) < n_batches:
            got.append(tuple(x.copy() for x in eng.feed_collect()) + (eng.feed_flags.copy(),))
        eng.feed_close()
        from supersdr_amd import _lib as L
        assert eng.kernel_stats(L.K_FUSED)[1] == n_batches and eng.kernel_stats(L.K_WF)[1] == 0      # it WAS the fused kernel
    for b in range(n_batches):
        for k in range(4):
            assert np.array_equal(got[b][k], want[b][k]), (b, k)


def test_pipelined_feed_through_the_wave_specialised_kernel_equals_the_two_kernels(S):
    """The feed's batches with every channel on the general audio path (SSB, CW, a narrowed AM passband), 8 frames per slot, N = 3: ssdr_run_chain
    takes ssdr_chain_ws_kernel (its ticket counter only counts up from slot to slot, the slots' input buffers change under it).  Waterfall sums,
    PCM, RSSI and ADC flags, batch after batch with the state carried, equal what the two per-stage kernels give synchronously."""
    n_ch, nf, n_batches = 21, 8, 5
    iq = O.synth_iq(n_ch, n_batches * nf * 512, seed=717)
    iq[7, 11 * 512 + 3, 1] = -32768
    kinds = [("usb", {}), ("lsb", {}), ("cw", {}), ("am", {"low_cut": -3000.0, "high_cut": 3000.0})]
    ps = [S.default_params(kinds[c % 4][0], f_shift_hz=150.0 * c - 1500.0, agc_hang=int(c == 2), **kinds[c % 4][1]) for c in range(n_ch)]
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.set_averaging(3)
        eng.set_fused(False)
        want = []
        for b in range(n_batches):
            eng.push_iq(iq[:, b * nf * 512:(b + 1) * nf * 512])
            lines, fused = eng.run_chain()
            assert not fused
            wf = eng.fetch_wf(lines).copy() if lines else np.zeros((0, n_ch, 1024), np.int16)
            pcm, rssi = eng.fetch_audio()
            want.append((wf, pcm.copy(), rssi.copy(), eng.audio_flags().copy()))
    assert sum(int(w[3].sum()) for w in want) >= 1
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.set_averaging(3)
        eng.set_profiling(True)
        eng.feed_open(nf, depth=2)
        got = []
        for b in range(n_batches):
            if b >= 2:
                got.append(tuple(x.copy() for x in eng.feed_collect()) + (eng.feed_flags.copy(),))
            eng.feed_slot()[:] = iq[:, b * nf * 512:(b + 1) * nf * 512]
            eng.feed_submit()
        while len(got) < n_batches:
            got.append(tuple(x.copy() for x in eng.feed_collect()) + (eng.feed_flags.copy(),))
        eng.feed_close()
        from supersdr_amd import _lib as L
        assert eng.kernel_stats(L.K_FUSED)[1] == n_batches and eng.kernel_stats(L.K_WF)[1] == 0 and eng.kernel_stats(L.K_AUDIO)[1] == 0
    for b in range(n_batches):
        for k in range(4):
            assert np.array_equal(np.asarray(got[b][k]).reshape(np.asarray(want[b][k]).shape), want[b][k]), (b, k)


def test_pipelined_feed_wire_mode(S):
    """SSDR_FEED_WIRE: SND bodies (big-endian, 17-byte header) in, same results as the int16 path, header rssi out"""
    import struct
    n_ch, nf, n_batches = 3, 2, 4
    iq = O.synth_iq(n_ch, n_batches * nf * 512, seed=17)
    ps, _ = mixed_params(S, n_ch)
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        want = []
        for b in range(n_batches):
            eng.push_iq(iq[:, b * nf * 512:(b + 1) * nf * 512])
            want.append((eng.run_wf().copy(), *(x.copy() for x in eng.run_audio())))
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.feed_open(nf, depth=2, wire=True)
        got = []
        for b in range(n_batches):
            slot = eng.feed_slot()
            for c in range(n_ch):
                for f in range(nf):
                    x = iq[c, (b * nf + f) * 512:(b * nf + f + 1) * 512]
                    hdr = struct.pack("<BI", 0, b * nf + f) + struct.pack(">H", 500 + 10 * c + f) + struct.pack("<BBII", 1, 0, 2, 3)
                    slot[c, f] = np.frombuffer(hdr + x.astype(">i2").tobytes(), np.uint8)
            eng.feed_submit()
            got.append(tuple(x.copy() for x in eng.feed_collect()))
        eng.feed_close()
    for b in range(n_batches):
        for k in range(3):
            assert np.array_equal(got[b][k], want[b][k]), (b, k)
        for c in range(n_ch):
            for f in range(nf):
                assert abs(got[b][3][c, f] - (0.1 * (500 + 10 * c + f) - 127)) < 1e-4


def test_error_codes(S):
    from supersdr_amd import _lib as L
    import ctypes as C
    ctx = L._P()
    assert L.lib.ssdr_create(0, 0, 1024, 512, C.byref(ctx)) == L.EINVAL
    assert L.lib.ssdr_create(0, 4, 512, 512, C.byref(ctx)) == L.EINVAL
    assert L.lib.ssdr_create(99, 4, 1024, 512, C.byref(ctx)) == L.ENODEV
    with S.SsdrEngine(2) as eng:
        with pytest.raises(S.SsdrError):
            eng.run_wf()                       # nothing pushed: SSDR_ESTATE
        with pytest.raises(S.SsdrError):
            eng.set_averaging(0)
        eng.push_iq(np.zeros((2, 512, 2), np.int16))
        with pytest.raises(S.SsdrError):
            eng.run_wf()                       # odd frame count: no whole 1024-sample line


def test_workers_end_to_end_on_gpu(S, twin):
    """kiwi_waterfall / kiwi_sound stand-ins fed through IQBatcher -> IQHub -> libssdr: what the two
    seams deliver equals the oracle on the same IQ (waterfall bytes bit-exact, PCM bit-exact vs twin)."""
    from supersdr_amd.workers import IQHub, bind_headless
    kiwi_waterfall, kiwi_sound = bind_headless().kiwi_waterfall, bind_headless().kiwi_sound
    from supersdr_amd.iqstream import IQBatcher

    class Disp:
        DISPLAY_WIDTH, WF_HEIGHT = 1024, 8

    n_ch = 3
    hub = IQHub(n_ch)
    wfs = [kiwi_waterfall("gpu", 0, "", 10, 7100.0, None, Disp(), hub=hub, channel=c, timeout=1.0) for c in range(n_ch)]
    pb = {"AM": (-6000, 6000), "USB": (30, 3000), "LSB": (-3000, -30)}          # the reference's passbands (utils_supersdr.py:46-50)
    snds = [kiwi_sound(7100.0 + ((c * 37) % 97 - 48) * 0.1, m, pb[m][0], pb[m][1], "", wfs[c], 4) for c, m in enumerate(["AM", "USB", "LSB"])]
    iq = O.synth_iq(n_ch, 4 * 1024, seed=31, modes=[0, 1, 2])
    feeders = [IQBatcher().attach(hub, c) for c in range(n_ch)]
    for k in range(8):                                        # 512-sample IQ frames, as the server sends them
        for c in range(n_ch):
            z = iq[c, k * 512:(k + 1) * 512].astype(np.float32)
            feeders[c]._process_iq_samples(k, (z[:, 0] + 1j * z[:, 1]).astype(np.complex64), -50.0, {})
    consts, taps = hub.engine.get_consts()
    ref_wf = twin.wf(iq, 1, consts["wf_cal_lin"])
    st, hist = twinlib.fresh_state(consts)
    ref_pcm, ref_rssi = twin.audio(iq, consts, taps, st, hist)
    for c in range(n_ch):
        for line in range(4):
            wfs[c].step()
            assert np.array_equal(wfs[c].spectrum, ref_wf[line, c].astype(np.float32))
        got = np.concatenate([snds[c].process_audio_stream() for _ in range(8)])
        assert np.array_equal(got, ref_pcm[c])
        assert snds[c].rssi == pytest.approx(float(ref_rssi[c, -1]))
    assert [int(m) for m in consts["mode"]] == [0, 2, 1]
    hub.close()


def test_hub_pipelined_feed_equals_the_synchronous_hub_and_flags_reach_the_seam(S):
    """IQHub on ring buffers: pipeline=True (ssdr_feed_*: pinned slots, three streams) hands out the same lines and
    frames as the synchronous hub, `depth - 1` superframes later; on the synchronous hub a clipped sample sets
    kiwi_sound.adc_overflow_flag for exactly its frame (utils_supersdr.py:1066-1067)."""
    from supersdr_amd.workers import IQHub, bind_headless
    kiwi_waterfall, kiwi_sound = bind_headless().kiwi_waterfall, bind_headless().kiwi_sound
    n_ch, n_sf = 3, 7
    iq = O.synth_iq(n_ch, n_sf * 1024, seed=61)
    iq[1, 2 * 1024 + 600, 1] = -32768                              # superframe 2, second audio frame, channel 1
    ps, _ = mixed_params(S, n_ch)
    hubs = [IQHub(n_ch, gpu_post=False), IQHub(n_ch, gpu_post=False, pipeline=True, depth=3)]
    for h in hubs:
        for c in range(n_ch):
            h.set_params(c, ps[c])
        for k in range(n_sf):
            for c in range(n_ch):
                h.feed(c, iq[c, k * 1024: k * 1024 + 300])
                h.feed(c, iq[c, k * 1024 + 300: (k + 1) * 1024])
    assert hubs[1].wf_queue[0].qsize() == n_sf - 2                 # two superframes still in flight
    hubs[1].flush()
    for c in range(n_ch):
        for k in range(n_sf):
            la, lb = hubs[0].wf_queue[c].get_nowait(), hubs[1].wf_queue[c].get_nowait()
            assert np.array_equal(la[0], lb[0]) and la[1] == lb[1] == 1
            for f in range(2):
                fa, fb = hubs[0].snd_queue[c].get_nowait(), hubs[1].snd_queue[c].get_nowait()
                assert np.array_equal(fa, fb) and fa.rssi == fb.rssi
                assert fa.adc_overflow == (c == 1 and k == 2 and f == 1)
    for h in hubs:
        h.close()

    class Disp:
        DISPLAY_WIDTH, WF_HEIGHT = 1024, 8
    hub = IQHub(1)
    wf = kiwi_waterfall("gpu", 0, "", 10, 7100.0, None, Disp(), hub=hub, channel=0, timeout=1.0)
    snd = kiwi_sound(7100.0, "AM", -6000, 6000, "", wf, 4)
    hub.feed(0, iq[1, 2 * 1024: 3 * 1024])
    flags = []
    for f in range(2):
        snd.process_audio_stream()
        flags.append(snd.adc_overflow_flag)
    assert flags == [False, True]
    hub.close()


def test_hub_bulk_ingest_on_the_gpu_equals_the_engine_run_directly(S):
    """Round 4: the hub's bulk paths on the real engine -- feed_block / reserve + commit into (pinned) superframe slots,
    ssdr_feed_submit_from straight out of them, SND bodies into a wire hub, K superframes per GPU run -- give bit for bit
    what ssdr_push_iq + ssdr_run_wf + ssdr_run_audio give on the same stream; on a lazy hub only the attached channel
    gets queue entries, and the whole-batch arrays (`last`, subscribe()) carry everybody's results."""
    from supersdr_amd.workers import IQHub
    from supersdr_amd.iqstream import int16_to_wire
    n_ch, n_sf = 2048, 6
    with S.SsdrEngine(n_ch) as e0:
        ps, _ = mixed_params(S, 8)
        e0.set_params(0, [ps[c % 8] for c in range(n_ch)])
        e0.synth_iq(2 * n_sf)
        iq = e0.read_input()                                                    # [n_ch, n_sf * 1024, 2]
        e0.push_iq(iq)
        wf_ref = e0.run_wf()
        pcm_ref, rssi_ref = e0.run_audio()
        flags_ref = e0.audio_flags()
    bodies = np.zeros((n_ch, 2 * n_sf, 2065), np.uint8)
    bodies[:, :, 17:] = iq.reshape(n_ch, 2 * n_sf, 512, 2).astype(">i2").view(np.uint8).reshape(n_ch, 2 * n_sf, 2048)
    assert bytes(bodies[3, 5, 17:]) == int16_to_wire(iq[3, 5 * 512: 6 * 512])[17:]
    for kw in (dict(), dict(pipeline=True, depth=3), dict(pipeline=True, depth=2, batch_superframes=2), dict(wire=True),
               dict(wire=True, pipeline=True, depth=3), dict(batch_superframes=3)):
        hub = IQHub(n_ch, gpu_post=False, lazy=True, **kw)
        hub.engine.set_params(0, [ps[c % 8] for c in range(n_ch)])           # (the engine takes blocks; the hub's set_params one channel)
        got = []
        hub.subscribe(lambda r: got.append((r.wf.copy(), r.pcm.copy(), r.rssi.copy(), r.flags.copy())))
        q = hub.attach(1234, wf=True, snd=True)
        K = kw.get("batch_superframes", 1)
        if kw.get("wire"):
            for f in range(0, 2 * n_sf, 2):
                hub.feed_wire_block(0, bodies[:1000, f:f + 1])                  # ragged: the first thousand run a frame ahead
                hub.feed_wire_block(1000, bodies[1000:, f:f + 2])
                hub.feed_wire_block(0, bodies[:1000, f + 1:f + 2])
        else:
            for k in range(n_sf):
                blk = iq[:, k * 1024:(k + 1) * 1024]
                if k % 2:
                    v = hub.reserve(0, n_ch)
                    assert v is not None
                    v[:, :1024] = blk
                    hub.commit(0, n_ch, 1024)
                else:
                    hub.feed_block(0, blk[:, :700])
                    hub.feed_block(0, blk[:700, 700:])
                    hub.feed_block(700, blk[700:, 700:])
        hub.flush()
        assert hub.superframes == n_sf // K and len(got) == n_sf // K and not hub.stalled.any()
        wf = np.concatenate([g[0] for g in got])
        pcm = np.concatenate([g[1] for g in got], axis=1)
        assert np.array_equal(wf, wf_ref) and np.array_equal(pcm, pcm_ref), kw
        assert np.array_equal(np.concatenate([g[2] for g in got], axis=1), rssi_ref)
        assert np.array_equal(np.concatenate([g[3] for g in got], axis=1), flags_ref)
        assert q["wf"].qsize() == n_sf and q["snd"].qsize() == 2 * n_sf and hub.wf_queue.attached(0) is None
        assert np.array_equal(q["wf"].get_nowait()[0], wf_ref[0, 1234]) and np.array_equal(q["snd"].get_nowait(), pcm_ref[1234, :512])
        hub.close()


def test_full_size_batch_properties(S, twin):
    """BASELINE configs[2]/[3] size (65536 channels, mixed modes, 10x binning): device-generated input,
    (i) a strided subset of channels bit-exact vs the twin on the same bytes, (ii) a channel's result does not
    depend on the batch it sits in, (iii) run-to-run determinism via a checksum over all outputs."""
    n_ch, n_frames, n_avg = 65536, 20, 10
    sub = np.arange(0, n_ch, 1021)[:48]
    with S.SsdrEngine(n_ch) as eng:
        ps, _ = mixed_params(S, 388)
        for first in range(0, n_ch, 388):
            eng.set_params(first, ps[: min(388, n_ch - first)])
        eng.set_averaging(n_avg)
        eng.synth_iq(n_frames, seed=99)
        iq_sub = np.stack([eng.read_input(int(c), 1)[0] for c in sub])
        sub64 = np.arange(3, n_ch, 255)[:256]               # the float64 check's channels: a strided 256, every mode 64 times
        iq64 = np.stack([eng.read_input(int(c), 1)[0] for c in sub64])
        wf = eng.run_wf()
        pcm, rssi = eng.run_audio()
        csum1 = (int(wf.astype(np.int64).sum()), int(pcm.astype(np.int64).sum()), float(rssi.astype(np.float64).sum()))
        eng.reset_state()
        eng.set_averaging(1)
        eng.set_averaging(n_avg)
        wf2 = eng.run_wf()
        pcm2, rssi2 = eng.run_audio()
        consts, taps = eng.get_consts()
    assert wf.shape == (1, n_ch, 1024) and pcm.shape == (n_ch, n_frames * 512)
    csum2 = (int(wf2.astype(np.int64).sum()), int(pcm2.astype(np.int64).sum()), float(rssi2.astype(np.float64).sum()))
    assert csum1 == csum2 and np.array_equal(pcm, pcm2)
    k, t = consts[sub], taps[sub]
    assert np.array_equal(wf[:, sub], twin.wf(iq_sub, n_avg, k["wf_cal_lin"]))
    st, hist = twinlib.fresh_state(k)
    pcm_t, rssi_t = twin.audio(iq_sub, k, t, st, hist)
    assert np.array_equal(pcm[sub], pcm_t) and np.array_equal(rssi[sub], rssi_t)
    # the same channels alone in a small batch
    with S.SsdrEngine(len(sub)) as eng:
        eng.set_params(0, [ps[int(c) % 388] for c in sub])
        eng.set_averaging(n_avg)
        eng.push_iq(iq_sub)
        assert np.array_equal(eng.run_wf(), wf[:, sub])
        p3, r3 = eng.run_audio()
        assert np.array_equal(p3, pcm[sub]) and np.array_equal(r3, rssi[sub])
    assert len(np.unique(k["mode"])) == 4 and pcm.std() > 1000
    # ... and against the float64 definition at this very shape (r4 item 7: a strided 256 channels instead of
    # eight, 64 of each mode, the oracle in a process pool): the N = 10 sums identical outside the guard band, PCM within the
    # north_star tolerance on every channel, every sample counted
    import oracle_pool as OP
    import tolerances as T
    _, ops = mixed_params(S, 388)
    o64 = [ops[int(c) % 388] for c in sub64]
    wf_o, gb = OP.wf(iq64, n_avg)
    d = np.abs(wf[:, sub64].astype(np.int32) - wf_o)
    assert not (d > gb).any()
    pcm_o, rssi_o, bound = OP.audio(iq64, o64, eps=T.EPS)
    well, rms = T.assert_pcm_within_tolerance(pcm[sub64], pcm_o, bound, what="configs[3] shape, 256 channels")
    assert well.all() and rms.max() < PCM_RMS_TOL           # BASELINE's own configuration: the plain figure on every channel
    assert [sum(p_.mode == m for p_ in o64) for m in ("am", "usb", "lsb", "nbfm")] == [64] * 4


def test_million_channel_batch(S, twin):
    """BASELINE's largest configuration: 2^20 channels in one context (one superframe): sampled channels from the
    whole range bit-exact vs the twin, nothing left unwritten, last channel included"""
    n_ch, n_frames = 1 << 20, 2
    sub = np.concatenate([np.arange(0, n_ch, 32749)[:30], [n_ch - 1]])
    with S.SsdrEngine(n_ch) as eng:
        ps, _ = mixed_params(S, 388)
        big = ps * (4096 // 388 + 1)
        for first in range(0, n_ch, 3880):
            eng.set_params(first, big[: min(3880, n_ch - first)])
        eng.synth_iq(n_frames, seed=7)
        iq_sub = np.stack([eng.read_input(int(c), 1)[0] for c in sub])
        wf = eng.run_wf()
        pcm, rssi = eng.run_audio()
        consts, taps = eng.get_consts()
    assert wf.shape == (1, n_ch, 1024) and pcm.shape == (n_ch, 1024) and rssi.shape == (n_ch, 2)
    k, t = consts[sub], taps[sub]
    assert np.array_equal(wf[:, sub], twin.wf(iq_sub, 1, k["wf_cal_lin"]))
    st, hist = twinlib.fresh_state(k)
    pcm_t, rssi_t = twin.audio(iq_sub, k, t, st, hist)
    assert np.array_equal(pcm[sub], pcm_t) and np.array_equal(rssi[sub], rssi_t)
    assert (wf.max(axis=2) > 0).all()                       # every channel's line was written (a carrier is in every channel)
    assert (np.abs(pcm).max(axis=1) > 0).all() and np.isfinite(rssi).all()


def _checksum(*arrays):
    return tuple(int(np.asarray(a).astype(np.int64).sum()) if np.asarray(a).dtype.kind in "iu" else
                 float(np.asarray(a).astype(np.float64).sum()) for a in arrays)


def test_parity_at_the_timed_shape_waterfall_only(S, twin):
    """bench.py --workload wf (BASELINE configs[1]) exactly as timed: 4096 channels x 256 lines in ONE launch, N = 1.
    A strided subset of channels, all 256 lines, bit-exact vs the twin on the same bytes; run-to-run checksum."""
    n_ch, n_lines = 4096, 256
    sub = np.arange(0, n_ch, 131)[:24]
    with S.SsdrEngine(n_ch) as eng:
        ps = [S.default_params("am", f_shift_hz=((c * 37) % 97 - 48) * 100.0) for c in range(388)]
        for first in range(0, n_ch, 388):
            eng.set_params(first, ps[: min(388, n_ch - first)])
        eng.synth_iq(2 * n_lines, seed=0x5D5D)
        iq_sub = np.stack([eng.read_input(int(c), 1)[0] for c in sub])
        sub64 = np.arange(0, n_ch, 16)[:256]                # the float64 check's channels: every 16th, 256 of them
        iq64 = np.stack([eng.read_input(int(c), 1)[0] for c in sub64])
        n1 = eng.run_wf(fetch=False)
        wf = eng.run_wf()                                   # the same launch again: N = 1 carries nothing
        consts, _ = eng.get_consts()
    assert n1 == n_lines and wf.shape == (n_lines, n_ch, 1024)
    assert np.array_equal(wf[:, sub], twin.wf(iq_sub, 1, consts["wf_cal_lin"][sub]))
    assert (wf.max(axis=2) > 150).all()                     # a carrier in every line of every channel
    # ... and against the float64 definition at this very shape (r4 item 7: 256 channels x all 256 lines = 6.7e7
    # bins, the oracle in a process pool) -- identical outside the guard band, never more than the allowed steps inside it
    import oracle_pool as OP
    wf_o, gb = OP.wf(iq64, 1)
    d = np.abs(wf[:, sub64].astype(np.int32) - wf_o)
    assert not (d > gb).any() and (d > 0).mean() < 1e-3
    print("waterfall at the timed shape vs the float64 oracle: %d of %d bins differ (all by one step, all inside the guard band)" % ((d > 0).sum(), d.size))
    with S.SsdrEngine(n_ch) as eng:                         # the float64 kernel at the timed shape: no guard band at all
        for first in range(0, n_ch, 388):
            eng.set_params(first, ps[: min(388, n_ch - first)])
        eng.set_exact_bins(True)
        eng.synth_iq(2 * n_lines, seed=0x5D5D)
        wfx = eng.run_wf()
    assert np.array_equal(wfx[:, sub64], wf_o) and (wfx != wf).mean() < 1e-3      # float64 kernel: bit for bit on all 6.7e7 bins
    with S.SsdrEngine(n_ch) as eng:
        for first in range(0, n_ch, 388):
            eng.set_params(first, ps[: min(388, n_ch - first)])
        eng.synth_iq(2 * n_lines, seed=0x5D5D)
        assert _checksum(eng.run_wf()) == _checksum(wf)


def test_parity_at_the_timed_shape_full_chain(S, twin):
    """bench.py's default workload (BASELINE configs[2]) exactly as timed: 65536 channels x 16 superframes per launch,
    N = 1, AM on the reference's full-band passband.  A strided subset bit-exact vs the twin (waterfall, PCM, RSSI,
    flags, carried state after the launch), then a second step on the carried state, and a checksum of everything."""
    n_ch, sf = 65536, 16
    sub = np.arange(0, n_ch, 2039)[:32]
    ps = [S.default_params("am", f_shift_hz=((c * 37) % 97 - 48) * 100.0) for c in range(388)]
    with S.SsdrEngine(n_ch) as eng:
        for first in range(0, n_ch, 388):
            eng.set_params(first, ps[: min(388, n_ch - first)])
        eng.reset_state()
        eng.synth_iq(2 * sf, seed=0x5D5D)
        assert eng.audio_paths() == (0, 0, n_ch)            # the path the bench times
        iq_sub = np.stack([eng.read_input(int(c), 1)[0] for c in sub])
        sub64 = np.arange(5, n_ch, 256)[:256]               # the float64 check's channels: a strided 256
        iq64 = np.stack([eng.read_input(int(c), 1)[0] for c in sub64])
        wf = eng.run_wf()
        pcm, rssi = eng.run_audio()
        flags = eng.audio_flags()
        sums1 = eng.output_checksum()
        pcm2, rssi2 = eng.run_audio()                       # the bench's next step: same input, carried state
        sums2 = eng.output_checksum()
        consts, taps = eng.get_consts()
        st_g, hist_g = eng.get_state()
        # ... and the way the bench takes since round 3: ssdr_run_chain = the fused superframe kernel on this configuration.
        # Two steps from the same fresh state must give the two kernels' bytes: checksums of everything, state, history
        eng.reset_state()
        lines, fused = eng.run_chain()
        assert fused and lines == sf and eng.output_checksum() == sums1
        lines, fused = eng.run_chain()
        assert fused and eng.output_checksum() == sums2
        st_f, hist_f = eng.get_state()
        assert st_f.tobytes() == st_g.tobytes() and np.array_equal(hist_f, hist_g)
    k, t = consts[sub], taps[sub]
    assert wf.shape == (sf, n_ch, 1024) and np.array_equal(wf[:, sub], twin.wf(iq_sub, 1, k["wf_cal_lin"]))
    st, hist = twinlib.fresh_state(k)
    pcm_t, rssi_t, flags_t = twin.audio(iq_sub, k, t, st, hist, want_flags=True)
    assert np.array_equal(pcm[sub], pcm_t) and np.array_equal(rssi[sub], rssi_t) and np.array_equal(flags[sub], flags_t)
    pcm_t2, rssi_t2 = twin.audio(iq_sub, k, t, st, hist)
    assert np.array_equal(pcm2[sub], pcm_t2) and np.array_equal(rssi2[sub], rssi_t2)
    assert st_g[sub].tobytes() == st.tobytes() and np.array_equal(hist_g[sub], hist)
    assert not flags.any() and (np.abs(pcm).max(axis=1) > 1000).all() and np.isfinite(rssi).all()
    # vs the float64 definition (r4 item 7: 256 strided channels x 16 superframes, the oracle in a process pool): the waterfall lines
    # identical outside the guard band, the PCM within the plain north_star figure on every one of them
    import oracle_pool as OP
    ops = [O.ChanParams(mode="am", f_shift_hz=((int(c) % 388 * 37) % 97 - 48) * 100.0) for c in sub64]
    pcm_o, _, _ = OP.audio(iq64, ops)
    rms = np.sqrt(((pcm[sub64].astype(np.float64) - pcm_o) ** 2).mean(axis=1)) / 32768.0
    assert rms.max() < PCM_RMS_TOL
    wf_o, gb = OP.wf(iq64, 1)
    d = np.abs(wf[:, sub64].astype(np.int32) - wf_o)
    assert not (d > gb).any() and (d > 0).mean() < 1e-3
    print("full chain at the timed shape vs the float64 oracle, 256 channels: PCM RMS max %.2e, %d of %d bins one step apart" % (rms.max(), (d > 0).sum(), d.size))


@pytest.mark.parametrize("n_ch,first_id", [(1 << 20, 0), (131072, 7 * 131072)])
def test_parity_at_configs4_timed_shape_through_the_fused_kernel(S, twin, n_ch, first_id):
    """BASELINE configs[4] exactly as bench.py times it (review r5, weak #3): 2^20 channels x 16 superframes in ONE launch of the fused
    kernel on one GPU (137 GB per launch: the > 4 Gi-element outputs, 64-bit indexing), and the per-rank shape of the 8-GPU run
    (131 072 channels, the LAST rank's channel ids).  ssdr_run_chain must take the fused kernel; two steps on the same resident input
    with carried state, as the bench's timed loop; a strided 64 channels incl. the last -- every waterfall line, PCM sample, RSSI float,
    ADC flag and the carried state after each step -- bit-exact vs the twin, fetched row by row (nothing of the 64 GiB of results is
    copied back whole); the checksum of ALL outputs equals a second fresh context's."""
    sf = 16
    sub = np.unique(np.concatenate([np.arange(0, n_ch, n_ch // 63 + 1)[:63], [n_ch - 1]]))
    assert len(sub) == 64 and sub[-1] == n_ch - 1
    period = 97
    ps = [S.default_params("am", f_shift_hz=(((first_id + c) * 37) % 97 - 48) * 100.0) for c in range(period)]
    big = ps * 40                                               # (97-channel parameter pattern, laid in runs of 3880)
    sums, got = [], {}
    for attempt in range(2):
        with S.SsdrEngine(n_ch) as eng:
            for first in range(0, n_ch, len(big)):
                eng.set_params(first, big[: min(len(big), n_ch - first)])
            eng.reset_state()
            eng.synth_iq(2 * sf, seed=0x5D5D, first_channel_id=first_id)
            assert eng.audio_paths() == (0, 0, n_ch)
            if attempt == 0:
                got["iq"] = np.stack([eng.read_input(int(c), 1)[0] for c in sub])
                got["k"], got["t"] = (np.concatenate(x) for x in zip(*[eng.get_consts(int(c), 1) for c in sub]))
            per_step = []
            for step in range(2):
                lines, fused = eng.run_chain()
                assert fused == 1 and lines == sf               # the kernel the bench's `million` workload times
                per_step.append(eng.output_checksum())
                if attempt == 0:
                    wf, pcm, rssi = eng.fetch_rows(sub, sf)
                    flags = eng.audio_flags()
                    st = [eng.get_state(int(c), 1) for c in sub]
                    got[step] = (wf, pcm, rssi, flags[sub], np.concatenate([x[0] for x in st]), np.concatenate([x[1] for x in st]), bool(flags.any()))
            sums.append(per_step)
    assert sums[0] == sums[1]                                   # a second fresh ctx reproduces every output byte of both steps
    k, t, iq = got["k"], got["t"], got["iq"]
    assert (k["fir_flags"] & 1).all() and (k["mode"] == 0).all()
    st, hist = twinlib.fresh_state(k)
    wf_t = twin.wf(iq, 1, k["wf_cal_lin"])
    for step in range(2):
        wf, pcm, rssi, flags, st_g, hist_g, any_flag = got[step]
        pcm_t, rssi_t, flags_t = twin.audio(iq, k, t, st, hist, want_flags=True)
        assert np.array_equal(wf, wf_t), "waterfall lines, step %d" % step
        assert np.array_equal(pcm, pcm_t) and np.array_equal(rssi, rssi_t) and np.array_equal(flags, flags_t), "audio, step %d" % step
        assert st_g.tobytes() == st.tobytes() and np.array_equal(hist_g, hist), "carried state after step %d" % step
        assert not any_flag and (np.abs(pcm).max(axis=1) > 1000).all() and (wf.max(axis=2) > 150).all()


def test_checkpoint_restore_continues_bit_exactly(S):
    """ssdr_checkpoint_save / _load: a fresh ctx restored from the blob continues every stream bit for bit --
    NCO phases, FIR history, DC / AGC / discriminator memory, the waterfall's partial sums in the middle of an N = 3
    group, the play_buffer history -- and a ssdr_set_params issued right after the restore does not wipe the state."""
    from supersdr_amd._lib import PlayChan
    n_ch, nf = 9, 4
    iq = O.synth_iq(n_ch, 3 * nf * 512, seed=71)
    ps, _ = mixed_params(S, n_ch)
    play = [PlayChan(100.0 + 5 * c, (c % 3 - 1) * 0.5) for c in range(n_ch)]

    def batch(eng, k):
        eng.push_iq(iq[:, k * nf * 512:(k + 1) * nf * 512])
        wf = eng.run_wf().copy()
        pcm, rssi = eng.run_audio()
        return wf, pcm.copy(), rssi.copy(), eng.run_playbuffer(play).copy()

    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, ps)
        eng.set_averaging(3)                       # 2 lines per batch: groups straddle the batches
        batch(eng, 0)
        blob = eng.checkpoint()
        want = [batch(eng, 1), batch(eng, 2)]
    with S.SsdrEngine(n_ch) as eng:                # a brand-new ctx: default parameters, no play_buffer buffers yet
        eng.restore(blob)
        eng.set_params(3, [ps[3]])                 # "any set_params issued before the first run" must keep the restored state
        got = [batch(eng, 1), batch(eng, 2)]
    for k in range(2):
        for a, b in zip(want[k], got[k]):
            assert a.shape == b.shape and np.array_equal(a, b), k
    assert want[0][0].shape[0] == 1 and want[1][0].shape[0] == 1    # 2 lines per batch, N = 3: a group closes in each of them
    with S.SsdrEngine(n_ch + 1) as eng:
        with pytest.raises(ValueError):
            eng.restore(blob)                      # channel count mismatch: the blob's size says so
    with S.SsdrEngine(n_ch) as eng:                # short, foreign and damaged blobs are refused, nothing is read out of bounds
        with pytest.raises(ValueError):
            eng.restore(blob[:-8])
        bad = bytearray(blob)
        bad[0] ^= 0xFF                             # magic
        with pytest.raises(S.SsdrError):
            eng.restore(bytes(bad))
        bad = bytearray(blob)
        bad[4:8] = (3).to_bytes(4, "little")       # a round-3 blob (same size, `kfm` word still padding) is refused
        with pytest.raises(S.SsdrError):
            eng.restore(bytes(bad))
        per_ch = 64 + 512 + 64 + 512 + 2048 + 64 + 2048          # consts, taps, state, hist, partial sums, play history, hop-512 tail
        bad = bytearray(blob)
        off = 48 + n_ch * per_ch + 2 * 88 + 16     # header (48 B) ... saved ssdr_chan_params[2].f_shift_hz
        bad[off:off + 8] = np.float64(1e9).tobytes()
        with pytest.raises(S.SsdrError):           # the parameters are what gets compiled on load: a set the compiler refuses is refused
            eng.restore(bytes(bad))
        eng.restore(blob)                          # and the ctx is still usable afterwards
        assert eng.averaging == 3
    with S.SsdrEngine(n_ch) as eng:                # the blob's compiled constants and taps are not what the kernels get: wipe them,
        wiped = bytearray(blob)                    # the restored streams still continue bit for bit (recompiled from the parameters)
        wiped[48:48 + n_ch * (64 + 512)] = bytes(n_ch * (64 + 512))
        eng.restore(bytes(wiped))
        again = batch(eng, 1)
        for a, b in zip(want[0], again):
            assert np.array_equal(a, b)
    with S.SsdrEngine(2) as eng:                   # the zoomed waterfall stream is not in the blob: said, not silently dropped
        eng.set_wf_zoom(2)
        with pytest.raises(S.SsdrError):
            eng.checkpoint()
    # the input rate and the waterfall framing travel with the blob
    iq2 = O.synth_iq(3, 4 * 1024, seed=72)
    with S.SsdrEngine(3) as eng:
        eng.set_decimation(2)
        eng.set_hop(512)
        eng.set_params(0, [S.default_params("usb", f_shift_hz=9000.0)] * 3)
        eng.push_iq(iq2[:, :2048])
        eng.run_wf(), eng.run_audio()
        blob2 = eng.checkpoint()
        eng.push_iq(iq2[:, 2048:])
        want2 = (eng.run_wf().copy(), eng.run_audio()[0].copy())
    with S.SsdrEngine(3) as eng:
        eng.restore(blob2)
        assert (eng.decim, eng.hop) == (2, 512)    # the wrapper's buffer bookkeeping follows the blob
        eng.push_iq(iq2[:, 2048:])
        got2 = (eng.run_wf().copy(), eng.run_audio()[0].copy())
        assert int(eng.get_consts()[0]["decim"][0]) == 2
    assert np.array_equal(want2[0], got2[0]) and np.array_equal(want2[1], got2[1]) and want2[0].shape[0] == 4


@pytest.mark.parametrize("n_avg", [1, 3])
def test_wf_hop_512_runs_of_groups_per_wave(S, twin, n_avg):
    """Hop 512 at a size where the host hands every wave a RUN of consecutive groups of its channel pair (grp_run > 1:
    more than 8 work items per resident wave) -- the layout the streaming-load scheme of the kernel relies on.  Two calls
    (the second starts inside an averaging group and on the carried half-line), odd channel count; a strided subset of
    channels bit-exact vs the twin, and every channel's lines through a checksum against a second engine fed frame by
    frame (runs of one)."""
    n_ch, calls = 1031, (601, 423)               # 516 pairs x 601 / 201 groups: runs of 9 / 3 groups at the first call
    n_frames = sum(calls)
    sub = np.arange(0, n_ch, 97)
    with S.SsdrEngine(n_ch) as eng:
        eng.set_hop(512)
        eng.set_averaging(n_avg)
        eng.synth_iq(n_frames, seed=99)
        iq = eng.read_input()
        outs, pos = [], 0
        for k in calls:
            eng.push_iq(iq[:, pos * 512:(pos + k) * 512])
            outs.append(eng.run_wf())
            pos += k
        consts, _ = eng.get_consts()
    got = np.concatenate(outs, axis=0)
    stream = np.concatenate([np.zeros((len(sub), 512, 2), np.int16), iq[sub]], axis=1)
    ref = twin.wf_hop(stream, 512, n_avg, consts["wf_cal_lin"][sub])
    assert got.shape == (n_frames // n_avg, n_ch, 1024) and np.array_equal(got[:, sub], ref)
    with S.SsdrEngine(n_ch) as eng:                  # the same stream in small pushes: one group per work item
        eng.set_hop(512)
        eng.set_averaging(n_avg)
        small, pos = [], 0
        while pos < n_frames:
            k = min(61, n_frames - pos)
            eng.push_iq(iq[:, pos * 512:(pos + k) * 512])
            small.append(eng.run_wf())
            pos += k
    assert np.array_equal(got, np.concatenate(small, axis=0))


@pytest.mark.parametrize("n_ch,n_avg", [(1, 1), (5, 1), (6, 3), (33, 10)])
def test_wf_hop_512_bit_exact_vs_twin_and_oracle(S, twin, n_ch, n_avg):
    """ssdr_set_hop(512): lines overlap by half (23.4 lines/s, the reference's MAX_FPS = 23, utils_supersdr.py:597).  One
    line per 512-sample frame, the half-line before the first frame carried from the previous call (silence at the
    start); ragged pushes incl. odd frame counts; N-line sums with groups straddling calls; bit-exact vs the twin, vs
    the float64 oracle outside the guard band."""
    n_frames = 23
    iq = O.synth_iq(n_ch, n_frames * 512, seed=40 + n_ch)
    cal = np.linspace(-6, 6, n_ch)
    outs, pos = [], 0
    with S.SsdrEngine(n_ch) as eng:
        eng.set_params(0, [S.default_params("am", wf_cal_db=float(cal[c])) for c in range(n_ch)])
        eng.set_hop(512)
        eng.set_averaging(n_avg)
        for k in (1, 2, 7, 3, 10):
            eng.push_iq(iq[:, pos * 512:(pos + k) * 512])
            outs.append(eng.run_wf())
            pos += k
        consts, _ = eng.get_consts()
    got = np.concatenate(outs, axis=0)
    stream = np.concatenate([np.zeros((n_ch, 512, 2), np.int16), iq], axis=1)          # silence in front
    ref = twin.wf_hop(stream, 512, n_avg, consts["wf_cal_lin"])
    assert got.shape == ref.shape == (n_frames // n_avg, n_ch, 1024) and np.array_equal(got, ref)
    if n_avg == 1:
        for c in range(min(n_ch, 3)):
            o = O.wf_lines_hop(stream[c], 512, cal[c]).astype(np.int32)
            idx = np.arange(n_frames)[:, None] * 512 + np.arange(1024)[None, :]
            gb = O.wf_allowed_diff(stream[c][idx], cal[c])
            assert not (np.abs(got[:, c].astype(np.int32) - o) > gb).any()
    # every second line of the hop-512 stream (the aligned ones) is a hop-1024 line of the same samples
    if n_avg == 1:
        with S.SsdrEngine(n_ch) as eng:
            eng.set_params(0, [S.default_params("am", wf_cal_db=float(cal[c])) for c in range(n_ch)])
            eng.push_iq(iq[:, : 22 * 512])
            assert np.array_equal(eng.run_wf(), got[1:22:2])
        with S.SsdrEngine(2) as eng:
            with pytest.raises(S.SsdrError):
                eng.set_hop(256)


@pytest.mark.parametrize("decim", [2, 4])
def test_decimating_front_end_bit_exact_vs_twin_and_oracle(S, twin, decim, seed=None):
    """ssdr_set_decimation(D): IQ at D * 12 kHz, decimating polyphase FIR in the audio kernel, waterfall
    lines from the wide stream.  Random parameter sets over all modes (up to the 127 / 125-tap cap), several calls with
    the state crossing them: PCM, RSSI, flags, carried state bit-exact vs the twin; vs the float64 oracle (plain
    convolve-and-take-every-D-th) within the PCM tolerance."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random_params as RP
    rng = np.random.default_rng(200 + decim if seed is None else seed)       # (seed: tools/soak_parity.py)
    n_ch, n_frames = 40, 6
    kw = [RP.draw(rng) for _ in range(n_ch)]
    for k in kw:
        k["f_shift_hz"] *= decim
    iq = RP.signal(rng, n_ch, n_frames * 512 * decim)
    ps = [S.default_params(k["mode"], f_shift_hz=k["f_shift_hz"], low_cut=k["low_cut"], high_cut=k["high_cut"],
                           agc_on=k["agc_on"], agc_hang=k["hang"], agc_thresh=k["thresh"], agc_slope=k["slope"],
                           agc_decay=k["decay"], agc_man_gain=k["man_gain"], wf_cal_db=k["wf_cal_db"],
                           smeter_cal_db=k["smeter_cal_db"]) for k in kw]
    with S.SsdrEngine(n_ch) as eng:
        eng.set_decimation(decim)
        eng.set_params(0, ps)
        pcms, rssis, flags, wfs, pos = [], [], [], [], 0
        for nf in (1, 3, 2):
            eng.push_iq(iq[:, pos * 512 * decim:(pos + nf) * 512 * decim])
            if (nf * decim) % 2 == 0:
                wfs.append((pos, nf, eng.run_wf()))
            p, r = eng.run_audio()
            pcms.append(p)
            rssis.append(r)
            flags.append(eng.audio_flags())
            pos += nf
        consts, taps = eng.get_consts()
        st_g, hist_g = eng.get_state()
    assert (consts["decim"] == decim).all() and (consts["fir_flags"] == 0).all()
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t, flags_t = twin.audio(iq, consts, taps, st, hist, want_flags=True)
    pcm = np.concatenate(pcms, axis=1)
    assert pcm.shape == (n_ch, n_frames * 512) and np.array_equal(pcm, pcm_t)
    assert np.array_equal(np.concatenate(rssis, axis=1), rssi_t) and np.array_equal(np.concatenate(flags, axis=1), flags_t)
    assert st_g.tobytes() == st.tobytes() and np.array_equal(hist_g, hist)
    for pos0, nf, wf in wfs:                                           # the waterfall sees the wide stream, 1024 samples per line
        seg = iq[:, pos0 * 512 * decim:(pos0 + nf) * 512 * decim]
        assert np.array_equal(wf, twin.wf(seg, 1, consts["wf_cal_lin"]))
    # vs the float64 oracle: every well-conditioned channel (tests/tolerances.py), no best-of selection
    import tolerances as T
    pcm_o, rssi_o, bound = T.oracle_with_bound(iq, [O.ChanParams(**k) for k in kw], decim)
    T.assert_pcm_within_tolerance(pcm, pcm_o, bound)
    with S.SsdrEngine(2) as eng:
        with pytest.raises(S.SsdrError):
            eng.set_decimation(3)
        eng.set_decimation(2)
        with pytest.raises(S.SsdrError):
            eng.feed_open(2, 3)                                        # the pipelined feed is 12 kHz only


@pytest.mark.parametrize("n_ch", [1, 6, 7, 300])
def test_fused_superframe_kernel_equals_the_two_kernels(S, twin, n_ch):
    """ssdr_run_chain with ssdr_set_fused: one kernel reads each 4 KB line once for its FFT and its two audio frames (the
    metric's configuration: full-band AM, N = 1, hop 1024).  Waterfall, PCM, RSSI, flags, carried state and FIR history
    are bit-identical to the two-kernel path and to the twin, over several calls (state crossing them, > 64 frames in
    one of them), odd channel counts (a half-empty wave), clipping samples; a configuration the fused kernel does not
    cover goes through the two kernels."""
    rng = np.random.default_rng(n_ch)
    calls = [2, 8, 130] if n_ch <= 7 else [2, 8, 10]          # (batches under 8 frames go through the two kernels: state crosses both ways)
    n_frames = sum(calls)
    iq = O.synth_iq(n_ch, n_frames * 512, seed=500 + n_ch)
    iq[0, 3 * 512 + 511, 1] = -32768                                  # last sample of a frame
    if n_ch > 5:
        iq[5, 5 * 512, 0] = 32767                                     # first sample of a frame
    ps = [S.default_params("am", f_shift_hz=float(rng.integers(-5900, 5900)), agc_hang=int(c % 3 == 0),
                           agc_decay=float(rng.choice([400.0, 4000.0])), wf_cal_db=float(rng.integers(-6, 7))) for c in range(n_ch)]
    outs = {}
    for fused in (False, True):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_params(0, ps)
            eng.set_fused(fused)
            wfs, pcms, rssis, flags, pos = [], [], [], [], 0
            for nf in calls:
                eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
                lines, was_fused = eng.run_chain()
                assert was_fused == (fused and nf >= 8) and lines == nf // 2
                wfs.append(eng.fetch_wf(lines))
                p, r = eng.fetch_audio()
                pcms.append(p)
                rssis.append(r)
                flags.append(eng.audio_flags())
                pos += nf
            consts, taps = eng.get_consts()
            st, hist = eng.get_state()
        outs[fused] = (np.concatenate(wfs), np.concatenate(pcms, axis=1), np.concatenate(rssis, axis=1),
                       np.concatenate(flags, axis=1), st.tobytes(), hist)
    for a, b in zip(outs[False], outs[True]):
        assert (a == b) if isinstance(a, bytes) else np.array_equal(a, b)
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t, flags_t = twin.audio(iq, consts, taps, st, hist, want_flags=True)
    assert np.array_equal(outs[True][1], pcm_t) and np.array_equal(outs[True][2], rssi_t) and np.array_equal(outs[True][3], flags_t)
    assert outs[True][4] == st.tobytes() and np.array_equal(outs[True][5], hist)
    assert np.array_equal(outs[True][0], twin.wf(iq, 1, consts["wf_cal_lin"]))
    assert outs[True][3].sum() == (2 if n_ch > 5 else 1)
    # not the fused kernel's configuration: a USB channel among them, or N = 3
    with S.SsdrEngine(2) as eng:
        eng.set_fused(True)
        eng.set_params(0, [S.default_params("am"), S.default_params("usb")])
        eng.push_iq(O.synth_iq(2, 1024, seed=1))
        assert eng.run_chain() == (1, False)
        eng.set_params(1, [S.default_params("am")])
        eng.set_averaging(3)
        assert eng.run_chain() == (0, False)


@pytest.mark.parametrize("n_ch", [1, 7, 300])
def test_fused_float64_kernel_equals_the_two_kernels_and_the_numpy_path(S, twin, n_ch):
    """ssdr_set_exact_bins + ssdr_run_chain on the metric's configuration: ssdr_fused_exact_am_kernel reads each line once for its
    float64 FFT and its two audio frames.  Waterfall, PCM, RSSI, flags, carried state and history are bit-identical to
    ssdr_wf_exact_kernel followed by the AM audio kernel (ssdr_set_fused(0)), over several calls (state crossing them, > 64 frames in one,
    a call under 8 frames through the two kernels in between), odd channel counts, clipping samples; the int16 bins equal the NumPy
    float64 path bit for bit, the audio the fp32 twin.  Hop 512 and N > 1 keep the two kernels."""
    rng = np.random.default_rng(40 + n_ch)
    calls = [8, 2, 130] if n_ch <= 7 else [8, 2, 10]
    n_frames = sum(calls)
    iq = O.synth_iq(n_ch, n_frames * 512, seed=700 + n_ch)
    iq[0, 3 * 512 + 511, 1] = -32768
    cal = [float(rng.integers(-6, 7)) for _ in range(n_ch)]
    ps = [S.default_params("am", f_shift_hz=float(rng.integers(-5900, 5900)), agc_hang=int(c % 3 == 0),
                           agc_decay=float(rng.choice([400.0, 4000.0])), wf_cal_db=cal[c]) for c in range(n_ch)]
    outs = {}
    for fused in (False, True):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_params(0, ps)
            eng.set_exact_bins(True)
            eng.set_fused(fused)
            wfs, pcms, rssis, flags, pos = [], [], [], [], 0
            for nf in calls:
                eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
                lines, was_fused = eng.run_chain()
                assert was_fused == (fused and nf >= 8) and lines == nf // 2
                wfs.append(eng.fetch_wf(lines))
                p, r = eng.fetch_audio()
                pcms.append(p), rssis.append(r), flags.append(eng.audio_flags())
                pos += nf
            consts, taps = eng.get_consts()
            st, hist = eng.get_state()
            if fused:                                                # what the float64 fused kernel does not cover stays with the two kernels
                eng.set_hop(512)
                eng.push_iq(iq[:, :8 * 512])
                assert eng.run_chain()[1] == 0
                eng.set_hop(1024)
                eng.set_averaging(3)
                eng.push_iq(iq[:, :8 * 512])
                assert eng.run_chain()[1] == 0
        outs[fused] = (np.concatenate(wfs), np.concatenate(pcms, axis=1), np.concatenate(rssis, axis=1),
                       np.concatenate(flags, axis=1), st.tobytes(), hist)
    for a, b in zip(outs[False], outs[True]):
        assert (a == b) if isinstance(a, bytes) else np.array_equal(a, b)
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t, flags_t = twin.audio(iq, consts, taps, st, hist, want_flags=True)
    assert np.array_equal(outs[True][1], pcm_t) and np.array_equal(outs[True][2], rssi_t) and np.array_equal(outs[True][3], flags_t)
    assert outs[True][4] == st.tobytes() and np.array_equal(outs[True][5], hist)
    for c in range(0, n_ch, max(1, n_ch // 6)):                    # the NumPy float64 path, bit for bit
        assert np.array_equal(outs[True][0][:, c], O.wf_sum_lines(iq[c].reshape(-1, 1024, 2), 1, cal[c])), c


@pytest.mark.parametrize("n_ch", [1, 7, 300])
def test_fused_kernel_at_hop_512_equals_the_two_kernels(S, twin, n_ch):
    """Round 4: ssdr_run_chain's one-read kernel at the reference's waterfall line rate (hop 512: a line per audio frame, the
    previous half-line re-read).  Waterfall lines, PCM, RSSI, flags, carried state, FIR history and the carried half-line are
    bit-identical to the two kernels and to the twin over several calls -- odd frame counts, the half-line crossing calls
    and crossing between the two ways, a clipping sample."""
    rng = np.random.default_rng(40 + n_ch)
    calls = [3, 9, 131] if n_ch <= 7 else [5, 8, 11]
    n_frames = sum(calls)
    iq = O.synth_iq(n_ch, n_frames * 512, seed=900 + n_ch)
    iq[0, 4 * 512 + 17, 0] = 32767
    ps = [S.default_params("am", f_shift_hz=float(rng.integers(-5900, 5900)), agc_hang=int(c % 2),
                           agc_decay=float(rng.choice([400.0, 4000.0])), wf_cal_db=float(rng.integers(-6, 7))) for c in range(n_ch)]
    outs = {}
    for fused in (False, True):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_hop(512)
            eng.set_params(0, ps)
            eng.set_fused(2 if fused else 0)           # (at hop 512 the one-read kernel is opt-in: the two stages side by side are faster)
            wfs, pcms, rssis, flags, pos = [], [], [], [], 0
            for nf in calls:
                eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
                lines, was_fused = eng.run_chain()
                assert was_fused == (fused and nf >= 8) and lines == nf
                wfs.append(eng.fetch_wf(lines))
                p, r = eng.fetch_audio()
                pcms.append(p)
                rssis.append(r)
                flags.append(eng.audio_flags())
                pos += nf
            consts, taps = eng.get_consts()
            st, hist = eng.get_state()
        outs[fused] = (np.concatenate(wfs), np.concatenate(pcms, axis=1), np.concatenate(rssis, axis=1),
                       np.concatenate(flags, axis=1), st.tobytes(), hist)
    for a, b in zip(outs[False], outs[True]):
        assert (a == b) if isinstance(a, bytes) else np.array_equal(a, b)
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t, flags_t = twin.audio(iq, consts, taps, st, hist, want_flags=True)
    assert np.array_equal(outs[True][1], pcm_t) and np.array_equal(outs[True][2], rssi_t) and np.array_equal(outs[True][3], flags_t)
    stream = np.concatenate([np.zeros((n_ch, 512, 2), np.int16), iq], axis=1)
    assert np.array_equal(outs[True][0], twin.wf_hop(stream, 512, 1, consts["wf_cal_lin"])) and outs[True][3].sum() == 1


@pytest.mark.parametrize("hop", [1024, 512])
def test_fused_kernel_with_time_binning_equals_the_two_kernels(S, hop):
    """ssdr_fused_am_kernel<., AVG = true> (round 4, opt-in: ssdr_set_fused(ctx, 2)): N = 3 sums kept in registers across the lines
    a wave walks, groups straddling the calls both ways -- bit-identical to the two kernels (waterfall sums, PCM, state)."""
    n_ch, calls = 7, [8, 10, 12, 9 if hop == 512 else 14]
    iq = O.synth_iq(n_ch, sum(calls) * 512, seed=77)
    outs = {}
    for f in (0, 2):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_hop(hop)
            eng.set_averaging(3)
            eng.set_fused(f)
            got, pos = [], 0
            for nf in calls:
                eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
                lines, was = eng.run_chain()
                assert was == (f == 2)
                got += [eng.fetch_wf(lines).copy(), eng.fetch_audio()[0].copy()]
                pos += nf
            got.append(eng.get_state()[0].tobytes())
        outs[f] = got
    assert sum(len(g) for g in outs[2][:-1:2]) == (sum(calls) // 3 if hop == 512 else sum(calls) // 2 // 3)
    for a, b in zip(outs[0], outs[2]):
        assert (a == b) if isinstance(a, bytes) else np.array_equal(a, b)
    with S.SsdrEngine(2) as eng:                            # the default (1) keeps N > 1 on the two kernels side by side
        eng.set_averaging(3)
        eng.push_iq(O.synth_iq(2, 8 * 512, seed=1))
        assert eng.run_chain()[1] == 0


def _chain_ws_params(S, n_ch, seed):
    """random receivers in every pairing of audio paths inside a workgroup's group of eight: general (short and long filters, CW's
    127 taps, mod=iq with its second output row), shift, AM-shift"""
    import random_params as RP
    rng = np.random.default_rng(seed)
    kinds = ["am", "nbfm", "usb", "lsb", "am_narrow", "nbfm_narrow", "usb_wide", "am", "usb", "cw", "iq", "am_very_narrow"]
    ps = []
    for c in range(n_ch):
        d = RP.draw(rng)
        kind = kinds[int(rng.integers(0, len(kinds)))]
        mode, lc, hc = {"am": ("am", -6000.0, 6000.0), "nbfm": ("nbfm", -6000.0, 6000.0), "usb": ("usb", 30.0, 3000.0), "lsb": ("lsb", -3000.0, -30.0),
                        "am_narrow": ("am", -4000.0, 4000.0), "nbfm_narrow": ("nbfm", -5000.0, 5000.0), "usb_wide": ("usb", 100.0, 3900.0),
                        "cw": ("cw", 300.0, 700.0), "iq": ("iq", -5000.0, 5000.0), "am_very_narrow": ("am", -1500.0, 1500.0)}[kind]
        ps.append(S.default_params(mode, f_shift_hz=d["f_shift_hz"] if rng.random() < 0.85 else 0.0, low_cut=lc, high_cut=hc, agc_on=d["agc_on"],
                                   agc_hang=d["hang"], agc_thresh=d["thresh"], agc_slope=d["slope"], agc_decay=d["decay"],
                                   agc_man_gain=d["man_gain"], wf_cal_db=d["wf_cal_db"], smeter_cal_db=d["smeter_cal_db"]))
    return ps


@pytest.mark.parametrize("n_ch,n_avg,seed", [(1, 1, 5), (2, 1, 6), (37, 1, 7), (64, 3, 8), (301, 10, 9)])
def test_wave_specialised_kernel_equals_the_two_kernels(S, n_ch, n_avg, seed):
    """Round 6: ssdr_chain_ws_kernel (ssdr_set_fused(ctx, 3)) -- both stages on one read of the input for ANY mix of audio paths, any
    filter and any N: audio waves hand every raw frame to an FFT wave of their workgroup through the LDS.  Bit-identical to the two
    kernels in everything they leave behind: waterfall sums (groups straddling the calls both ways), PCM, mod=iq pairs, RSSI,
    ADC-overflow flags, carried state and raw history, over several calls incl. a parameter change between two of them and a call
    of more than 64 frames (the RSSI keepers wrap)."""
    import random_params as RP
    calls = [8, 4, 14, 132 if n_ch <= 37 else 10, 6]
    rng = np.random.default_rng(1000 + seed)
    iq = RP.signal(rng, n_ch, sum(calls) * 512)
    ps = _chain_ws_params(S, n_ch, seed)
    outs = {}
    for f in (0, 3):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_averaging(n_avg)
            eng.set_fused(f)
            eng.set_params(0, ps)
            paths = eng.audio_paths()
            got, pos = [], 0
            for i, nf in enumerate(calls):
                eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
                lines, was = eng.run_chain()
                assert (was in (1, 2)) if f == 3 else was == 0, (was, paths)      # (1: every channel happens to be full-band AM)
                got += [eng.fetch_wf(lines).copy() if lines else np.zeros(0), eng.fetch_audio()[0].copy(), eng.fetch_audio()[1].copy(), eng.audio_flags().copy()]
                st, hist = eng.get_state()
                try:
                    pairs = eng.audio_iq().copy()          # (refused when the ctx holds no mod=iq channel: then by both)
                except Exception:
                    pairs = np.zeros(0)
                got += [st.tobytes(), hist.tobytes(), pairs]
                if i == 1:                      # a retune and a mode change between two calls: the carried state is re-read per call
                    eng.set_params(0, [S.default_params("lsb", f_shift_hz=-1234.5)])
                pos += nf
        outs[f] = got
    names = ["wf", "pcm", "rssi", "flags", "state", "hist", "iq pairs"]
    for k, (a, b) in enumerate(zip(outs[0], outs[3])):
        same = (a == b) if isinstance(a, bytes) else np.array_equal(a, b)
        assert same, "call %d: %s differs" % (k // 7, names[k % 7])


def test_wave_specialised_kernel_declines_what_it_does_not_cover(S):
    """hop 512, float64 bins, a waterfall zoom, a decimating front end: ssdr_run_chain falls back to the two kernels; a CW channel
    (127 taps) or an IQ-mode channel does not make it"""
    iq = O.synth_iq(4, 8 * 512, seed=3)
    for setup in ("cw", "iq", "hop", "exact", "zoom"):
        with S.SsdrEngine(4) as eng:
            eng.set_fused(3)
            eng.set_params(0, [S.default_params("usb")] * 4)
            if setup == "cw":
                eng.set_params(1, [S.default_params("cw")])
            elif setup == "iq":
                eng.set_params(2, [S.default_params("iq")])
            elif setup == "hop":
                eng.set_hop(512)
            elif setup == "zoom":
                eng.set_wf_zoom(2)
            else:
                eng.set_exact_bins(1)
            eng.push_iq(iq)
            assert eng.run_chain()[1] == (2 if setup in ("cw", "iq") else 0), setup
        with S.SsdrEngine(4) as eng:
            eng.set_fused(3)
            eng.set_params(0, [S.default_params("usb")] * 4)
            eng.push_iq(iq)
            assert eng.run_chain()[1] == 2


def test_run_chain_default_takes_the_wave_specialised_kernel_where_every_channel_filters(S):
    """ssdr_run_chain's default (ssdr_set_fused level 1): a batch whose channels ALL run the general audio path (SSB, CW, a narrowed
    AM passband: `change_passband`, utils_supersdr.py:1078-1092) goes through ssdr_chain_ws_kernel (fused == 2), with the bytes of
    the two kernels; one full-band channel among them, fewer than 8 frames, or level 0 and it is the stages side by side"""
    n_ch, calls = 77, [8, 16, 10]
    iq = O.synth_iq(n_ch, sum(calls) * 512, seed=606)
    kinds = [("usb", {}), ("lsb", {}), ("cw", {}), ("am", {"low_cut": -4000.0, "high_cut": 4000.0}), ("nbfm", {"low_cut": -5000.0, "high_cut": 5000.0})]
    ps = [S.default_params(kinds[c % 5][0], f_shift_hz=((c * 37) % 97 - 48) * 100.0, **kinds[c % 5][1]) for c in range(n_ch)]
    outs = {}
    for level in (0, 1):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_fused(level)
            eng.set_params(0, ps)
            assert eng.audio_paths()[0] == n_ch
            got, pos = [], 0
            for nf in calls:
                eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
                lines, was = eng.run_chain()
                assert was == (2 if level else 0)
                st, hist = eng.get_state()
                got += [eng.fetch_wf(lines).copy(), eng.fetch_audio()[0].copy(), eng.fetch_audio()[1].copy(), eng.audio_flags().copy(), st.tobytes(), hist.tobytes()]
                pos += nf
        outs[level] = got
    for k, (a, b) in enumerate(zip(outs[0], outs[1])):
        assert (a == b) if isinstance(a, bytes) else np.array_equal(a, b), "call %d, item %d differs" % (k // 6, k % 6)
    with S.SsdrEngine(4) as eng:
        eng.set_params(0, [S.default_params("usb")] * 4)
        eng.push_iq(iq[:4, :6 * 512])
        assert eng.run_chain()[1] == 0                    # a short batch: the per-call set-up would not pay
        eng.set_params(3, [S.default_params("am")])       # a full-band AM receiver among them: the AM-shift path
        eng.push_iq(iq[:4, :8 * 512])
        assert eng.run_chain()[1] == 0
        eng.set_fused(3)
        eng.push_iq(iq[:4, :8 * 512])
        assert eng.run_chain()[1] == 2


@pytest.mark.parametrize("workload,n_avg,sf,n_ch", [("am_narrow", 1, 16, 65536), ("mixed", 10, 10, 65536), ("am_narrow", 1, 16, 1 << 20)])
def test_wave_specialised_kernel_at_the_timed_shapes(S, workload, n_avg, sf, n_ch):
    """bench.py's extra.full_am_narrow (ssdr_run_chain's default there) and extra.mixed_chain_ws (configs[3] with ssdr_set_fused 3) exactly
    as timed -- 65536 channels, synthetic input, the bench's parameter pattern: two steps through ssdr_chain_ws_kernel leave the two
    kernels' bytes (checksums of every waterfall sum, PCM sample and RSSI value; carried state; raw history).  And configs[4]'s shape with
    every receiver on a narrowed passband: 2^20 channels x 16 superframes, 64 GiB of input and as much of results -- offsets beyond 2^32
    elements in every row the kernel addresses, half a million tickets."""
    modes = ("am",) if workload == "am_narrow" else ("am", "usb", "lsb", "nbfm")
    over = {"low_cut": -4000.0, "high_cut": 4000.0} if workload == "am_narrow" else {}
    period = 97 * len(modes)
    ps = [S.default_params(modes[c % len(modes)], f_shift_hz=((c * 37) % 97 - 48) * 100.0, **over) for c in range(period)]
    got = {}
    for level in (0, 3):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_averaging(n_avg)
            eng.set_fused(level)
            for first in range(0, n_ch, period):
                eng.set_params(first, ps[: min(period, n_ch - first)])
            eng.reset_state()
            eng.synth_iq(2 * sf, seed=0x5D5D)
            steps = []
            for _ in range(2):
                lines, was = eng.run_chain()
                assert was == (2 if level else 0) and lines == sf // n_avg
                steps.append(eng.output_checksum())
            st, hist = eng.get_state()
            got[level] = (steps, st.tobytes(), hist.tobytes())
    assert got[0] == got[3]


def test_run_chain_floors_keep_small_batches_on_the_two_kernels(S):
    """ssdr_run_chain's default takes a one-read kernel only from a batch size on (ssdr_get_chain_floors, from the device: one channel pair per
    resident wave for the fused AM kernel, 16 pairs per trio for the wave-specialised one) -- below it the two stages side by side are 2-3 x
    faster (profiles/r06_ab_small_batches.txt).  The levels that ask for a kernel explicitly ignore its floor; the floors can be set."""
    with S.SsdrEngine(64, chain_floors="library") as eng:
        am_floor, ws_floor = eng.chain_floors()
        assert 1024 <= am_floor <= 65536 and am_floor <= ws_floor <= 262144
        eng.synth_iq(16)
        assert eng.run_chain()[1] == 0                    # 64 full-band AM receivers: far below the floor
        eng.set_fused(2)
        assert eng.run_chain()[1] == 1
        eng.set_fused(1)
        eng.set_params(0, [S.default_params("usb")] * 64)
        assert eng.run_chain()[1] == 0
        eng.set_fused(3)
        assert eng.run_chain()[1] == 2
        eng.set_fused(1)
        eng.set_chain_floors(0, 64)
        assert eng.chain_floors() == (0, 64) and eng.run_chain()[1] == 2
        eng.set_chain_floors(0, 65)
        assert eng.run_chain()[1] == 0
    with S.SsdrEngine(am_floor, chain_floors="library") as eng:      # at the floor: the fused AM kernel
        eng.synth_iq(8)
        assert eng.run_chain()[1] == 1
    with S.SsdrEngine(ws_floor, chain_floors="library") as eng:      # at the floor: the wave-specialised kernel
        ps = [S.default_params("usb", f_shift_hz=100.0 * (c % 9)) for c in range(256)]
        for first in range(0, ws_floor, 256):
            eng.set_params(first, ps[: min(256, ws_floor - first)])
        eng.synth_iq(8)
        assert eng.run_chain()[1] == 2


def test_run_chain_side_by_side_stages_are_bit_identical_and_joined_before_what_depends_on_them(S):
    """Round 4: a batch ssdr_run_chain does not fuse runs its audio stage on a second stream beside the waterfall kernel
    (ssdr_set_overlap, default on).  Same bytes as one after the other -- waterfall sums (N = 3 groups straddling calls, hop 512),
    PCM, RSSI, flags, carried state, history, checksums -- with everything that depends on the audio stage issued right behind
    the call: the next input, a parameter change, a state read-back, a reset of some channels, play_buffer, a checkpoint."""
    from supersdr_amd._lib import PlayChan
    n_ch, calls = 333, [6, 10, 3, 9, 12]
    iq = O.synth_iq(n_ch, sum(calls) * 512, seed=404)
    ps, _ = mixed_params(S, n_ch)
    play = [PlayChan(100.0, 0.0)] * n_ch
    outs = {}
    for overlap in (0, 1):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_overlap(overlap)
            eng.set_hop(512)
            eng.set_averaging(3)
            eng.set_params(0, ps)
            got, pos = [], 0
            for i, nf in enumerate(calls):
                eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
                lines, fused = eng.run_chain()
                assert not fused
                if i == 1:
                    eng.set_params(5, [S.default_params("cw", f_shift_hz=700.0)])        # right behind the launch
                if i == 2:
                    eng.reset_state(100, 7)
                if i == 3:
                    blob = eng.checkpoint()
                got.append((eng.output_checksum(), eng.get_state()[0].tobytes()))
                got.append(eng.fetch_wf(lines).copy())
                got.append(eng.fetch_audio()[0].copy())
                got.append(eng.audio_flags().copy())
                got.append(eng.run_playbuffer(play)[:, :64].copy())
                pos += nf
            got.append(blob)
        outs[overlap] = got
    for a, b in zip(outs[0], outs[1]):
        assert (a == b) if isinstance(a, (bytes, tuple)) else np.array_equal(a, b)


def test_a_refused_run_chain_leaves_every_stream_untouched(S, twin):
    """Round 5 (advisor): with the stages side by side ssdr_run_chain launches the audio stage FIRST, so the waterfall stage's
    shape rules (a whole number of lines at hop 1024) are checked before either is launched: a refused batch advances nothing,
    and the caller who fixes the batch and retries gets the stream it would have got without the mistake."""
    n_ch = 9
    iq = O.synth_iq(n_ch, 7 * 512, seed=515)
    ps, _ = mixed_params(S, n_ch)
    with S.SsdrEngine(n_ch) as eng, S.SsdrEngine(n_ch) as ref:
        for e in (eng, ref):
            e.set_params(0, ps)
            e.set_averaging(2)
            e.push_iq(iq[:, :4 * 512])
            e.run_chain()
        before = (eng.get_state()[0].tobytes(), eng.get_state()[1].tobytes(), eng.output_checksum())
        eng.push_iq(iq[:, 4 * 512:7 * 512])                  # three frames: one and a half lines
        with pytest.raises(S.SsdrError):
            eng.run_chain()
        assert (eng.get_state()[0].tobytes(), eng.get_state()[1].tobytes(), eng.output_checksum()) == before
        for e in (eng, ref):                                 # the retry with a well-formed batch continues where the stream stood
            e.push_iq(iq[:, 4 * 512:6 * 512])
            lines, _ = e.run_chain()
        assert np.array_equal(eng.fetch_wf(lines), ref.fetch_wf(lines))
        assert np.array_equal(eng.fetch_audio()[0], ref.fetch_audio()[0])
        assert eng.get_state()[0].tobytes() == ref.get_state()[0].tobytes()


def test_run_chain_on_a_callers_stream_leaves_both_stages_ordered_behind_it(S):
    """ssdr_set_stream: a caller that orders its own work behind its own stream (and never calls ssdr_sync) must see the audio
    stage too, although that one ran on the ctx's second stream: ssdr_run_chain joins it before returning.  The results are read
    with the caller's own copies on the caller's stream."""
    import ctypes
    from supersdr_amd import _lib as L
    hip = ctypes.CDLL("libamdhip64.so")
    n_ch, nf = 2048, 16
    iq = O.synth_iq(n_ch, nf * 512, seed=77)
    ps, _ = mixed_params(S, n_ch)
    with S.SsdrEngine(n_ch) as eng:                                 # the reference run: one stage after the other
        eng.set_overlap(0)
        eng.set_params(0, ps)
        eng.set_averaging(2)
        eng.push_iq(iq)
        eng.run_chain()
        want = eng.fetch_audio()[0].copy()
    stream = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1) == 0          # hipStreamNonBlocking
    try:
        with S.SsdrEngine(n_ch) as eng:
            eng.set_params(0, ps)
            eng.set_averaging(2)
            eng.set_stream(stream)
            eng.push_iq(iq)
            for rep in range(3):                                     # (state carries: only the first pass is compared)
                _, fused = eng.run_chain()
                assert not fused
                if rep == 0:
                    p, r = L._P(), L._P()
                    L.check(L.lib.ssdr_audio_device(eng._ctx, ctypes.byref(p), ctypes.byref(r)), "ssdr_audio_device")
                    got = np.empty_like(want)
                    assert hip.hipMemcpyAsync(ctypes.c_void_p(got.ctypes.data), p, ctypes.c_size_t(got.nbytes), 2, stream) == 0     # D2H
                    assert hip.hipStreamSynchronize(stream) == 0
                    assert np.array_equal(got, want)
            eng.set_stream(None)
    finally:
        hip.hipStreamDestroy(stream)


# ------------------------------------------------------------------ round 3
def test_set_concurrent_bits_are_bit_identical_to_the_default(S):
    """ssdr_set_concurrent: bit 0 (audio stage on a second stream beside the waterfall kernel, which then takes
    one workgroup per CU) and bit 1 (the per-path audio kernels one after the other) are scheduling only -- waterfall, PCM,
    RSSI, flags, carried state and FIR history of a mixed batch (all three frame paths, N = 3 groups straddling calls) are
    bit-identical to the default in every combination."""
    n_ch, calls = 37, [2, 6, 4]
    iq = O.synth_iq(n_ch, sum(calls) * 512, seed=77)
    iq[3, 700, 0] = 32767
    ps, _ = mixed_params(S, n_ch)
    outs = {}
    for mode in (0, 1, 2, 3):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_params(0, ps)
            eng.set_averaging(3)
            eng.set_concurrent(mode)
            assert all(n > 0 for n in eng.audio_paths())
            wfs, pcms, rssis, flags, pos = [], [], [], [], 0
            for nf in calls:
                eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
                wfs.append(eng.run_wf())
                p, r = eng.run_audio()
                pcms.append(p), rssis.append(r), flags.append(eng.audio_flags())
                pos += nf
            st, hist = eng.get_state()
            sums = eng.output_checksum()
        outs[mode] = (np.concatenate(wfs), np.concatenate(pcms, axis=1), np.concatenate(rssis, axis=1),
                      np.concatenate(flags, axis=1), st.tobytes(), hist, sums)
    for mode in (1, 2, 3):
        for a, b in zip(outs[0], outs[mode]):
            assert (a == b) if isinstance(a, (bytes, tuple)) else np.array_equal(a, b), mode
    assert outs[0][3].sum() == 1


def test_output_checksum_is_a_function_of_the_results_only(S):
    """ssdr_output_checksum (the multi-GPU parity hash): the same bytes give the same three sums whatever
    produced them (two kernels or the fused one, device-generated or pushed input, a second ctx), one changed input sample
    changes them, and the sums equal the host's own position-weighted sums of the fetched results."""
    n_ch, nf = 300, 8

    def host_sum(a):                                  # sum_i (word_i + 0x9E3779B9 mod 2^32) * (2 i + 1)  mod 2^64
        w = (np.ascontiguousarray(a).view(np.uint32).reshape(-1).astype(np.uint64) + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
        k = np.uint64(2) * np.arange(len(w), dtype=np.uint64) + np.uint64(1)
        with np.errstate(over="ignore"):
            return int((w * k).sum(dtype=np.uint64))    # uint64 arithmetic wraps mod 2^64

    with S.SsdrEngine(n_ch) as eng:
        eng.set_fused(False)
        eng.synth_iq(nf, seed=5, first_channel_id=1000)
        iq = eng.read_input()
        lines, fused = eng.run_chain()
        a = eng.output_checksum()
        wf, (pcm, rssi) = eng.fetch_wf(lines), eng.fetch_audio()
        assert not fused
    assert a == (host_sum(wf), host_sum(pcm), host_sum(rssi))
    with S.SsdrEngine(n_ch) as eng:                 # pushed instead of generated, fused kernel instead of two (the default)
        eng.push_iq(iq)
        lines, fused = eng.run_chain()
        assert fused and eng.output_checksum() == a
    iq2 = iq.copy()
    iq2[123, 2000, 1] ^= 1
    with S.SsdrEngine(n_ch) as eng:
        eng.push_iq(iq2)
        eng.run_wf(fetch=False), eng.run_audio(fetch=False)
        b = eng.output_checksum()
    assert b != a and (b[0] != a[0] or b[1] != a[1])


def test_reset_state_and_rate_change_clear_the_hop512_tail(S, twin):
    """at hop 512 the half-line carried from the previous batch is stream state.  ssdr_reset_state (and
    ssdr_set_decimation, which resets the streams) must return it to silence: the first line after a reset is then the
    line of a fresh ctx, not one built on a stale half-line of the old stream."""
    n_ch = 5
    iq = O.synth_iq(n_ch, 6 * 512, seed=31)
    with S.SsdrEngine(n_ch) as eng:
        eng.set_hop(512)
        eng.push_iq(iq[:, :2048])
        eng.run_wf()
        eng.reset_state()
        eng.push_iq(iq[:, 2048:])
        got = eng.run_wf()
        eng.push_iq(iq[:, :1024])
        eng.run_wf()
        eng.reset_state(1, 2)                        # a channel range: only those tails are cleared
        eng.push_iq(iq[:, 2048:])
        part = eng.run_wf()
        eng.set_decimation(2)
        eng.push_iq(iq[:, :2048])                    # 2 frames at D = 2
        dec = eng.run_wf()
        consts, _ = eng.get_consts()
    cal = consts["wf_cal_lin"]
    fresh = twin.wf_hop(np.concatenate([np.zeros((n_ch, 512, 2), np.int16), iq[:, 2048:]], axis=1), 512, 1, cal)
    assert np.array_equal(got, fresh)
    carried = twin.wf_hop(np.concatenate([iq[:, 512:1024], iq[:, 2048:]], axis=1), 512, 1, cal)
    for c in range(n_ch):
        assert np.array_equal(part[:, c], (fresh if c in (1, 2) else carried)[:, c]), c
    assert np.array_equal(dec, twin.wf_hop(np.concatenate([np.zeros((n_ch, 512, 2), np.int16), iq[:, :2048]], axis=1), 512, 1, cal))


def test_pipelined_feed_carries_flags_and_n_per_slot(S):
    """what belongs to a batch travels with its slot -- the ADC-overflow flags of ITS frames (not of the batch
    submitted after it) and the averaging N that was in force when it was submitted."""
    n_ch, nf = 4, 2
    iq = O.synth_iq(n_ch, 4 * nf * 512, seed=19)
    iq[2, 1 * nf * 512 + 600, 0] = -32768            # batch 1, frame 1, channel 2
    iq[0, 3 * nf * 512 + 5, 1] = 32767               # batch 3, frame 0, channel 0
    with S.SsdrEngine(n_ch) as eng:
        eng.feed_open(nf, depth=3)
        ns = [1, 1, 2, 2]
        got = []
        for b in range(4):
            eng.set_averaging(ns[b])
            eng.feed_slot()[:] = iq[:, b * nf * 512:(b + 1) * nf * 512]
            eng.feed_submit()
            if b >= 2:
                wf, pcm, rssi = eng.feed_collect()
                got.append((len(wf), eng.feed_n_avg, eng.feed_flags.copy()))
        while len(got) < 4:
            wf, pcm, rssi = eng.feed_collect()
            got.append((len(wf), eng.feed_n_avg, eng.feed_flags.copy()))
        eng.feed_close()
    assert [g[1] for g in got] == ns and [g[0] for g in got] == [1, 1, 0, 1]
    want = np.zeros((4, n_ch, nf), np.uint8)
    want[1, 2, 1] = 1
    want[3, 0, 0] = 1
    for b in range(4):
        assert np.array_equal(got[b][2], want[b]), b


@pytest.mark.parametrize("n_ch,n_avg,hop", [(64, 1, 1024), (33, 10, 1024), (7, 3, 512), (1, 1, 1024)])
def test_exact_bins_equal_the_float64_oracle_bit_for_bit(S, n_ch, n_avg, hop, seed=None):
    """ssdr_set_exact_bins (north_star: "bit-exact for the int16 waterfall bins"): with the waterfall stage
    evaluated in float64 there is NO guard band -- every int16 sum equals oracle/ssdr_oracle.py's (NumPy float64 FFT) on
    BASELINE configs[1]- and configs[3]-shaped batches (N = 1 and N = 10 time binning, mixed calibrations, groups
    straddling ragged calls, hop 512), while the default fp32 kernel differs from the same oracle in ~3e-4 of the bins."""
    calls = [4, 6, 2, 18] if hop == 1024 else [1, 2, 7, 3, 10]
    n_frames = sum(calls)
    if seed is None:
        iq = O.synth_iq(n_ch, n_frames * 512, seed=900 + n_ch, modes=[c % 4 for c in range(n_ch)])
    else:                                                                  # tools/soak_parity.py: levels from silence to the rails
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import random_params as RP
        iq = RP.signal(np.random.default_rng(seed), n_ch, n_frames * 512)
    iq[0, :1024] = 0                                                       # a silent line: p = 0 -> byte 0
    cal = np.linspace(-9, 9, n_ch) if n_ch > 1 else np.array([0.0])
    out = {}
    for exact in (True, False):
        with S.SsdrEngine(n_ch) as eng:
            eng.set_params(0, [S.default_params("am", wf_cal_db=float(cal[c])) for c in range(n_ch)])
            eng.set_hop(hop)
            eng.set_averaging(n_avg)
            eng.set_exact_bins(exact)
            got, pos = [], 0
            for nf in calls:
                eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
                got.append(eng.run_wf())
                pos += nf
        out[exact] = np.concatenate(got, axis=0)
    n_diff = 0
    for c in range(n_ch):
        if hop == 1024:
            ref = O.wf_sum_lines(iq[c].reshape(-1, 1024, 2), n_avg, cal[c])
        else:
            stream = np.concatenate([np.zeros((512, 2), np.int16), iq[c]])                   # silence in front
            b = O.wf_lines_hop(stream, 512, cal[c]).astype(np.int16)
            L = len(b) // n_avg
            ref = b[: L * n_avg].reshape(L, n_avg, 1024).sum(axis=1).astype(np.int16)
        assert out[True].shape[0] == len(ref)
        assert np.array_equal(out[True][:, c], ref), (c, int((out[True][:, c] != ref).sum()))
        n_diff += int((out[False][:, c] != ref).sum())
    if n_ch >= 33 and seed is None:
        assert 0 < n_diff < 3e-3 * out[False].size * n_avg                # what the mode exists for


def test_iq_chain_at_20250_hz_bit_exact_vs_twin_and_oracle(S, twin, seed=None):
    """ssdr_set_kiwi_rate(20250) (missing #3): the IQ of a three-channel KiwiSDR.  Every channel's constants are
    recompiled for the rate (NCO steps, taps, AGC decay, NBFM scale = the oracle's at that rate), the streams reset; random
    parameter sets over all modes, state crossing calls: PCM, RSSI, flags, state bit-exact vs the twin, vs the float64 oracle
    under the tolerance rule; tuning may reach +-10.125 kHz; back at 12 kHz the full-band AM shortcut paths return."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random_params as RP
    import tolerances as T
    rate = 20250
    rng = np.random.default_rng(2025 if seed is None else seed)
    n_ch, n_frames = 64, 6
    kw = [RP.draw(rng) for _ in range(n_ch)]
    kw[0]["f_shift_hz"], kw[0]["mode"], kw[0]["low_cut"], kw[0]["high_cut"] = 9800.0, "usb", 30.0, 300.0
    iq = RP.signal(rng, n_ch, n_frames * 512)
    ps = [S.default_params(k["mode"], f_shift_hz=k["f_shift_hz"], low_cut=k["low_cut"], high_cut=k["high_cut"],
                           agc_on=k["agc_on"], agc_hang=k["hang"], agc_thresh=k["thresh"], agc_slope=k["slope"],
                           agc_decay=k["decay"], agc_man_gain=k["man_gain"], wf_cal_db=k["wf_cal_db"],
                           smeter_cal_db=k["smeter_cal_db"]) for k in kw]
    with S.SsdrEngine(n_ch) as eng:
        with pytest.raises(S.SsdrError):
            eng.set_params(0, ps[:1])               # 9.8 kHz off centre does not exist in a 12 kHz band
        eng.set_params(1, ps[1:])                   # given at 12 kHz ...
        assert seed is not None or sum(eng.audio_paths()[1:]) > 0       # (some of them on the full-band shortcut paths there)
        eng.set_kiwi_rate(rate)                     # ... recompiled for 20.25 kHz
        eng.set_params(0, ps[:1])
        assert eng.audio_paths()[1:] == (0, 0) and eng.kiwi_rate == rate
        pcms, rssis, fl, wfs, pos = [], [], [], [], 0
        for nf in (2, 4):
            eng.push_iq(iq[:, pos * 512:(pos + nf) * 512])
            wfs.append(eng.run_wf())
            p, r = eng.run_audio()
            pcms.append(p), rssis.append(r), fl.append(eng.audio_flags())
            pos += nf
        consts, taps = eng.get_consts()
        st_g, hist_g = eng.get_state()
        with pytest.raises(S.SsdrError):
            eng.set_params(1, [S.default_params("am", f_shift_hz=10200.0)])
        with pytest.raises(S.SsdrError):
            eng.set_kiwi_rate(12000)                # channel 0 sits 9.8 kHz off centre: refused, and nothing changed
        assert eng.get_consts()[0]["kfm"][0] == consts["kfm"][0]
        eng.set_params(0, [S.default_params("am")])
        eng.set_kiwi_rate(12000)
        assert sum(eng.audio_paths()[1:]) > 0
    for c in (0, 5, 63):
        k = O.compile_params(O.ChanParams(**kw[c]), 1, rate)
        assert int(consts["dphi1"][c]) == int(k["dphi1"]) and int(consts["dphi2"][c]) == int(k["dphi2"]) and int(consts["ntap"][c]) == int(k["ntap"])
        assert np.array_equal(taps[c], k["taps"]) and consts["kfm"][c] == k["kfm"] and consts["agc_delta8"][c] == k["agc_delta8"]
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t, flags_t = twin.audio(iq, consts, taps, st, hist, want_flags=True)
    pcm = np.concatenate(pcms, axis=1)
    assert np.array_equal(pcm, pcm_t) and np.array_equal(np.concatenate(rssis, axis=1), rssi_t)
    assert np.array_equal(np.concatenate(fl, axis=1), flags_t) and st_g.tobytes() == st.tobytes() and np.array_equal(hist_g, hist)
    assert np.array_equal(np.concatenate(wfs), twin.wf(iq, 1, consts["wf_cal_lin"]))
    pcm_o, rssi_o, bound = T.oracle_with_bound(iq, [O.ChanParams(**k) for k in kw], 1, rate)
    T.assert_pcm_within_tolerance(pcm, pcm_o, bound)


@pytest.mark.parametrize("decim", [1, 4])
def test_mod_iq_bit_exact_vs_twin_and_oracle(S, twin, decim):
    """SSDR_MODE_IQ ("SET mod=iq", kiwi/client.py:217-249; VERDICT r2 missing #2): the tuned, filtered, gain-controlled baseband
    itself -- PCM row = I, ssdr_audio_iq = I,Q pairs -- on the general path and behind the decimating filter, among channels
    in other modes (whose IQ rows stay zero).  Bit-exact vs the twin, I and Q within the tolerance rule vs the float64 oracle;
    and with the default +-5 kHz passband a tone comes out as a rotating phasor of half-scale magnitude."""
    import tolerances as T
    n_ch, calls = 10, [2, 4]
    modes = ["iq", "usb", "iq", "am", "iq", "nbfm", "iq", "lsb", "iq", "cw"]
    rng = np.random.default_rng(60 + decim)
    kw = []
    for c, m in enumerate(modes):
        d = dict(mode=m, f_shift_hz=float(rng.integers(-3000, 3000)), agc_on=int(c != 4), man_gain=40.0, hang=int(c == 6),
                 decay=float(rng.choice([400, 4000])), thresh=-90.0, slope=0.0, wf_cal_db=0.0, smeter_cal_db=-13.0)
        lc, hc = {"iq": (-5000.0, 5000.0), "usb": (30.0, 3000.0), "lsb": (-3000.0, -30.0), "am": (-6000.0, 6000.0),
                  "nbfm": (-6000.0, 6000.0), "cw": (400.0, 800.0)}[m]
        if c == 8:
            lc, hc = -6000.0, 6000.0                    # full band: still the general path in IQ mode (no lane-shift shortcut)
        d.update(low_cut=lc, high_cut=hc)
        kw.append(d)
    iq = O.synth_iq(n_ch, sum(calls) * 512 * decim, seed=70 + decim, modes=[c % 4 for c in range(n_ch)])
    ps = [S.default_params(k["mode"], f_shift_hz=k["f_shift_hz"], low_cut=k["low_cut"], high_cut=k["high_cut"], agc_on=k["agc_on"],
                           agc_hang=k["hang"], agc_thresh=k["thresh"], agc_slope=k["slope"], agc_decay=k["decay"],
                           agc_man_gain=k["man_gain"]) for k in kw]
    assert S.default_params("iq").low_cut == -5000.0
    with S.SsdrEngine(n_ch) as eng:
        if decim > 1:
            eng.set_decimation(decim)
        eng.set_params(0, ps)
        if decim == 1:
            assert eng.audio_paths()[0] == 8            # every IQ channel (the full-band one too), USB, LSB, CW
        pcms, iqs, pos = [], [], 0
        for nf in calls:
            eng.push_iq(iq[:, pos * 512 * decim:(pos + nf) * 512 * decim])
            pcms.append(eng.run_audio()[0])
            iqs.append(eng.audio_iq())
            pos += nf
        consts, taps = eng.get_consts()
        eng.set_params(0, [S.default_params("am")] * n_ch)
        eng.push_iq(iq[:, :1024 * decim])
        eng.run_audio()
        with pytest.raises(S.SsdrError):
            eng.audio_iq()                               # no channel in IQ mode any more
    pcm, iqo = np.concatenate(pcms, axis=1), np.concatenate(iqs, axis=1)
    st, hist = twinlib.fresh_state(consts)
    pcm_t, rssi_t, iq_t = twin.audio(iq, consts, taps, st, hist, want_iq=True)
    assert np.array_equal(pcm, pcm_t) and np.array_equal(iqo, iq_t)
    is_iq = np.array([m == "iq" for m in modes])
    assert (iqo[~is_iq] == 0).all() and np.array_equal(iqo[is_iq][:, :, 0], pcm[is_iq])
    pcm_o, rssi_o, bound, q_o = O.audio_chain_with_bound(iq, [O.ChanParams(**k) for k in kw], T.EPS, decim, want_q=True)
    T.assert_pcm_within_tolerance(pcm, pcm_o, bound)
    T.assert_pcm_within_tolerance(iqo[is_iq][:, :, 1], q_o[is_iq], bound[is_iq])
    if decim == 1:                                      # AGC on, an AM carrier in the passband: the envelope's peaks sit at half scale
        mag = np.hypot(iqo[0, -512:, 0].astype(np.float64), iqo[0, -512:, 1].astype(np.float64))
        assert abs(mag.max() - 16384.0) < 16384.0 * 0.02 and mag.min() > 16384.0 * 0.25


@pytest.mark.parametrize("Z,hop,n_avg,decim", [(2, 1024, 1, 1), (4, 1024, 3, 1), (8, 512, 1, 1), (4, 1024, 1, 2)])
def test_waterfall_zoom_bit_exact_vs_twin_and_oracle(S, twin, Z, hop, n_avg, decim):
    """ssdr_set_wf_zoom / ssdr_set_wf_center (VERDICT r2 missing #4; "SET zoom= start=", utils_supersdr.py:741, 753-758, 839): a
    zoom stage (NCO at the zoom centre, the reference's tap formula at the new Nyquist, decimation by Z, int16 I,Q) in front of
    the unchanged waterfall kernel.  Zoomed stream and lines bit-exact vs the twin over several calls (phase, filter history,
    hop-512 tail and averaging groups carried), the stream within 1 LSB of the float64 oracle; the audio chain does not
    notice; a batch that does not hold whole zoomed lines is refused; a new centre restarts that channel only."""
    n_ch = 5
    fs_in = 12000.0 * decim
    unit = (1024 if hop == 1024 else 512) * Z // decim // 512       # frames per zoomed line
    unit = max(unit, 1)
    calls = [unit * k for k in (2, 1, 3)]
    n_frames = sum(calls)
    iq = O.synth_iq(n_ch, n_frames * 512 * decim, seed=300 + Z)
    offs = np.array([0.0, 1500.0, -2750.25, 0.45 * fs_in, -0.3 * fs_in])
    with S.SsdrEngine(n_ch) as eng:
        if decim > 1:
            eng.set_decimation(decim)
        eng.set_params(0, [S.default_params("usb" if decim > 1 else "am", wf_cal_db=float(c - 2)) for c in range(n_ch)])
        eng.set_hop(hop)
        eng.set_averaging(n_avg)
        eng.set_wf_zoom(Z)
        eng.set_wf_center(0, offs)
        with pytest.raises(S.SsdrError):
            eng.set_wf_center(0, [0.6 * fs_in])
        lines, zs, pcms, pos = [], [], [], 0
        for nf in calls:
            eng.push_iq(iq[:, pos * 512 * decim:(pos + nf) * 512 * decim])
            lines.append(eng.run_wf())
            zs.append(eng.read_zoom())
            pcms.append(eng.run_audio()[0])
            pos += nf
        consts, taps = eng.get_consts()
        if Z > 1 and hop == 1024 and decim == 1:
            eng.push_iq(iq[:, :1024])                                # 2 frames: not a whole zoomed line
            with pytest.raises(S.SsdrError):
                eng.run_wf()
    zoomed = np.concatenate(zs, axis=1)
    dphi = np.array([O._dphi(f, fs_in) for f in offs], np.uint32)
    ph, hist = np.zeros(n_ch, np.uint32), np.zeros((n_ch, 256, 2), np.int16)
    zt = twin.zoom(iq, Z, dphi, O.zoom_taps(Z), ph, hist)
    assert zoomed.shape == zt.shape == (n_ch, n_frames * 512 * decim // Z, 2) and np.array_equal(zoomed, zt)
    for c in range(n_ch):
        o = O.ZoomChannel(Z, offs[c], fs_in).process(iq[c])
        dd = np.abs(zoomed[c].astype(np.int32) - o.astype(np.int32))
        assert dd.max() <= 1 and (dd > 0).mean() < 0.01, c
    got = np.concatenate(lines, axis=0)
    cal = consts["wf_cal_lin"]
    if hop == 1024:
        ref = twin.wf(zt, n_avg, cal)
    else:
        ref = twin.wf_hop(np.concatenate([np.zeros((n_ch, 512, 2), np.int16), zt], axis=1), 512, n_avg, cal)
    assert got.shape == ref.shape and got.shape[0] > 0 and np.array_equal(got, ref)
    # the audio chain runs on the un-zoomed input as ever
    st, hist_a = twinlib.fresh_state(consts)
    pcm_t, _ = twin.audio(iq, consts, taps, st, hist_a)
    assert np.array_equal(np.concatenate(pcms, axis=1), pcm_t)
    # a new centre restarts that channel's zoom stream and nobody else's
    if Z == 2:
        with S.SsdrEngine(2) as eng:
            eng.set_wf_zoom(2)
            eng.set_wf_center(0, [1000.0, -1000.0])
            eng.push_iq(iq[:2, :4096])
            eng.run_wf()
            eng.set_wf_center(1, [2000.0])
            eng.push_iq(iq[:2, 4096:8192])
            eng.run_wf()
            z2 = eng.read_zoom()
            eng.reset_state()                                    # the zoomed streams restart with the others; the centres stay
            eng.push_iq(iq[:2, :4096])
            eng.run_wf()
            z3 = eng.read_zoom()
        d2 = np.array([O._dphi(1000.0, fs_in), O._dphi(2000.0, fs_in)], np.uint32)
        ph, hist = np.zeros(2, np.uint32), np.zeros((2, 256, 2), np.int16)
        t0 = twin.zoom(iq[:2, :8192], 2, np.array([d2[0], d2[0]], np.uint32), O.zoom_taps(2), ph, hist)[0, 2048:]
        ph, hist = np.zeros(1, np.uint32), np.zeros((1, 256, 2), np.int16)
        t1 = twin.zoom(iq[1:2, 4096:8192], 2, d2[1:], O.zoom_taps(2), ph, hist)[0]
        assert np.array_equal(z2[0], t0) and np.array_equal(z2[1], t1)
        ph, hist = np.zeros(2, np.uint32), np.zeros((2, 256, 2), np.int16)
        assert np.array_equal(z3, twin.zoom(iq[:2, :4096], 2, d2, O.zoom_taps(2), ph, hist))


def test_contexts_release_their_device_memory(S):
    """ssdr_destroy gives back everything a ctx took, whatever it was used for: contexts that each touch every lazily
    allocated buffer (pipelined feed with post-processing, zoom, exact bins, hop 512, decimation, wf_data ring, trace,
    playbuffer at both rates, wire input, pinned slots) are opened and closed in a loop; the device's free memory (hipMemGetInfo)
    does not drift."""
    import ctypes
    from supersdr_amd._lib import Db2colChan, PlayChan
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()

    def free_bytes():
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    n_ch = 512
    iq = O.synth_iq(n_ch, 4 * 512, seed=5)

    def one_life(k):
        with S.SsdrEngine(n_ch) as eng:
            ps, _ = mixed_params(S, n_ch)
            eng.set_params(0, ps)
            if k % 3 == 1:
                eng.set_hop(512)
            eng.set_averaging(1 + k % 3)
            eng.set_exact_bins(k % 2 == 1)
            eng.push_iq(iq)
            eng.run_chain()
            wf = eng.run_wf()
            eng.run_audio()
            eng.set_wfdata_rows(16)
            if len(wf):
                eng.run_db2col([Db2colChan(zoom=0, auto_scale=1, low_clip_db=-120.0, high_clip_db=-60.0, dynamic_range=40.0)] * n_ch, len(wf))
                eng.run_trace()
            eng.run_playbuffer([PlayChan(volume=100.0, balance=0.0)] * n_ch)
            eng.set_recording(True)
            eng.set_kiwi_rate(20250)
            eng.push_iq(iq)
            eng.run_audio()
            eng.run_playbuffer([PlayChan(volume=100.0, balance=0.0)] * n_ch)
            eng.set_kiwi_rate(12000)
            eng.set_exact_bins(False)
            eng.set_hop(1024)
            eng.set_wf_zoom(4)
            eng.push_iq(np.concatenate([iq, iq], axis=1))         # 4096 samples: one zoomed line
            eng.run_wf()
            eng.set_wf_zoom(1)
            eng.set_averaging(1)
            eng.feed_open(4, depth=3, post=True)
            buf = eng.host_alloc((n_ch, 4 * 512, 2), np.int16)
            buf[:] = iq
            for _ in range(4):
                eng.feed_submit_from(buf)
                eng.feed_collect()

    one_life(0)                                                   # the first ctx also pays for what the runtime keeps (code objects, pools)
    one_life(1)
    series = [free_bytes()]
    for k in range(36):
        one_life(k)
        series.append(free_bytes())
    print("free device memory after each ctx, MiB relative to the first:", [(x - series[0]) >> 20 for x in series])
    # the runtime grows its own pools in steps now and then; a leaked buffer would take its size with EVERY ctx
    assert abs(series[-1] - series[12]) <= 8 << 20 and abs(series[12] - series[0]) <= 32 << 20, series
