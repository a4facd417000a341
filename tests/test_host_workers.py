"""Host-side boundary logic (supersdr_amd/workers.py, iqstream.py, dist.py) on CPU.

The GPU engine is replaced by a test double that answers with the oracle twin, so what is
tested here is the host plumbing around the two seams: batching, queues, time binning by
division, spectrum_db2col / play_buffer against the reference's golden vectors, the
KiwiWorker retry policy, the wire <-> int16 conversion and the channel sharding."""
import os
import queue
import sys
import threading
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ssdr_oracle as O  # noqa: E402
import twinlib  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


class TwinEngine:
    """Test double with SsdrEngine's surface; numbers come from oracle/ssdr_twin.c."""

    def __init__(self, n_ch):
        import supersdr_amd as S
        self.S, self.n_ch, self.twin = S, n_ch, twinlib.load()
        self.consts = np.zeros(n_ch, twinlib.CONSTS_DTYPE)
        self.taps = np.zeros((n_ch, 128), np.float32)
        self.set_params(0, [S.default_params("am")] * n_ch)
        self.state, self.hist = twinlib.fresh_state(self.consts)
        self.n_avg, self.acc, self.phase = 1, None, 0
        self.param_log = []

    def set_params(self, first, params):
        for i, p in enumerate(params):
            k, t = self.S.compile_params(p)
            self.consts[first + i] = k
            self.taps[first + i] = t
            if hasattr(self, "param_log"):
                self.param_log.append((first + i, p))

    def set_averaging(self, n):
        if n != self.n_avg:
            self.n_avg, self.acc, self.phase = n, None, 0

    def push_iq(self, iq):
        self.iq = np.array(iq, np.int16)

    def run_wf(self):
        lines = self.twin.wf(self.iq, 1, self.consts["wf_cal_lin"])       # [L, ch, 1024]
        out = []
        for ln in lines:
            self.acc = ln.copy() if self.acc is None else self.acc + ln
            self.phase += 1
            if self.phase == self.n_avg:
                out.append(self.acc)
                self.acc, self.phase = None, 0
        return np.stack(out) if out else np.zeros((0, self.n_ch, 1024), np.int16)

    def run_audio(self):
        return self.twin.audio(self.iq, self.consts, self.taps, self.state, self.hist)

    def close(self):
        pass


class Disp:
    DISPLAY_WIDTH, WF_HEIGHT = 1024, 16


class Eibi:
    def __init__(self):
        self.calls = []

    def get_stations(self, a, b):
        self.calls.append((a, b))


def make_pair(n_ch=2, channel=1, zoom=10, freq=7100.0):
    from supersdr_amd.workers import IQHub, kiwi_waterfall, kiwi_sound
    hub = IQHub(n_ch, engine=TwinEngine(n_ch))
    wf = kiwi_waterfall("gpu", 0, "", zoom, freq, Eibi(), Disp(), hub=hub, channel=channel, timeout=0.2)
    snd = kiwi_sound(freq, "USB", 30, 3000, "", wf, 4)
    return hub, wf, snd


def test_seams_deliver_gpu_results_per_channel():
    hub, wf, snd = make_pair()
    iq = O.synth_iq(2, 3 * 1024, seed=8, modes=[0, 1])
    snd.freq = 7100.0 + ((1 * 37) % 97 - 48) * 0.1           # tune channel 1 onto its carrier (kHz)
    snd.set_mode_freq_pb()
    for c in range(2):                                         # ragged feeding: the hub re-blocks to superframes
        hub.feed(c, iq[c, :700])
        hub.feed(c, iq[c, 700:])
    assert hub.superframes == 3
    twin = twinlib.load()
    eng = hub.engine
    ref_wf = twin.wf(iq, 1, eng.consts["wf_cal_lin"])
    wf.receive_spectrum()
    assert wf.spectrum.dtype == np.float32 and np.array_equal(wf.spectrum, ref_wf[0, 1].astype(np.float32))
    st, hist = twinlib.fresh_state(eng.consts)
    ref_pcm, ref_rssi = twin.audio(iq, eng.consts, eng.taps, st, hist)
    for f in range(6):
        s = snd.process_audio_stream()
        assert s.dtype == np.int16 and np.array_equal(s, ref_pcm[1, f * 512:(f + 1) * 512])
        assert snd.rssi == pytest.approx(float(ref_rssi[1, f]))
    assert int(eng.consts["mode"][1]) == 2 and int(eng.consts["mode"][0]) == 0     # only channel 1 was retuned to USB
    with pytest.raises(queue.Empty):
        snd.process_audio_stream()
    assert snd.terminate and wf.terminate


def test_waterfall_run_loop_binning_scroll_and_db2col_vs_reference_golden():
    hub, wf, snd = make_pair(n_ch=1, channel=0, zoom=8)
    g = np.load(os.path.join(GOLD, "db2col.npz"))
    # db2col: feed the golden inputs through the restated method
    for i in range(int(g["count"])):
        zoom, auto, dlo, dhi = g["cfg_%d" % i]
        wf.zoom, wf.wf_auto_scaling, wf.delta_low_db, wf.delta_high_db = int(zoom), bool(auto), int(dlo), int(dhi)
        wf.low_clip_db, wf.high_clip_db, wf.dynamic_range = -120, -60, 40.0
        wf.spectrum = g["in_%d" % i].copy()
        wf.spectrum_db2col()
        assert np.array_equal(wf.wf_color, g["color_%d" % i])
        assert np.allclose([wf.low_clip_db, wf.high_clip_db, wf.dynamic_range, wf.wf_min_db, wf.wf_max_db],
                           g["scal_%d" % i], rtol=0, atol=0)
    # run loop with N = 3 time binning on the GPU side
    wf.zoom, wf.wf_auto_scaling, wf.delta_low_db, wf.delta_high_db = 8, True, 0, 0
    wf.averaging_n = 3
    iq = O.synth_iq(1, 15 * 1024, seed=9)
    hub.set_averaging(3)
    hub.feed(0, iq[0])
    lines = twinlib.load().wf(iq, 1)[:, 0].astype(np.float32)     # the byte lines the engine double produces
    for k in range(5):
        wf.step()
        assert np.array_equal(wf.spectrum, np.mean(list(lines[3 * k:3 * k + 3]), axis=0))   # == reference np.mean
    assert wf.run_index == 5
    # 3-deep delay buffer, newest line on row 0 (utils_supersdr.py:893-897)
    assert (wf.wf_data[2:] == 0).all() and (wf.wf_data[0] != 0).any() and (wf.wf_data[1] != 0).any()
    wf.set_white_flag()
    assert (wf.wf_data[0] == 255).all()


def test_freq_zoom_arithmetic():
    hub, wf, snd = make_pair(n_ch=1, channel=0, zoom=10, freq=7100.0)
    assert wf.span_khz == 30000 / 1024 and wf.start_f_khz == 7100 - wf.span_khz / 2
    assert wf.set_freq_zoom(14200.0, 0) == 15000 and wf.span_khz == 30000
    assert wf.set_freq_zoom(1.0, 10) == wf.span_khz / 2 and wf.start_f_khz == 0
    assert wf.set_freq_zoom(29999.0, 10) == 30000 - wf.span_khz / 2
    f = wf.set_freq_zoom(7100.0, 8)
    assert f == 7100.0 and wf.eibi.calls[-1] == (wf.start_f_khz, wf.end_f_khz)
    assert wf.bins_to_khz(512) == pytest.approx(7100.0) and wf.offset_to_bin(wf.span_khz) == 1024
    assert wf.deltabins_to_khz(1024) == pytest.approx(wf.span_khz)
    assert wf.counter == round(wf.start_f_khz / 30000 * 2 ** 14 * 1024)
    assert wf.div_list and wf.subdiv_list
    wf.radio_mode = "LSB"
    assert wf.change_passband(10, -20) == (-2980, -40)


def test_sound_control_plane_and_passbands():
    hub, wf, snd = make_pair(n_ch=1, channel=0)
    eng = hub.engine
    for mode, want in (("USB", (30, 3000)), ("LSB", (-3000, -30)), ("AM", (-6000, 6000)), ("CW", (400, 800))):
        snd.radio_mode = mode
        assert snd.change_passband(0, 0) == want
        snd.set_mode_freq_pb()
        assert int(eng.consts["mode"][0]) == {"USB": 2, "LSB": 1, "AM": 0, "CW": 3}[mode]
    assert snd.decay == 1000                                   # CW decay (utils_supersdr.py:1027)
    snd.change_agc_delay(-100)
    assert snd.decay == 900 and snd.decay_cw == 900
    snd.thresh = -100
    snd.set_agc_params()
    _, p = eng.param_log[-1]
    assert p.agc_thresh == -100 and p.agc_decay == 900 and p.mode == 3
    snd.freq = wf.freq + 1.5
    snd.set_mode_freq_pb()
    assert eng.param_log[-1][1].f_shift_hz == pytest.approx(1500.0)


def test_play_buffer_vs_reference_golden():
    hub, wf, snd = make_pair(n_ch=1, channel=0)
    g = np.load(os.path.join(GOLD, "playbuffer.npz"))
    for c in range(int(g["count"])):
        snd.volume, snd.audio_balance = g["cfg_%d" % c]
        snd.old_buffer = np.zeros(snd.n_tap - 1)
        snd.audio_buffer = queue.Queue()
        frames = g["in_%d" % c]
        for f in range(frames.shape[0]):
            snd.audio_buffer.put(frames[f])
            out = np.zeros((2048, 2), np.int16)
            snd.play_buffer(out, 2048, None, None)
            assert np.array_equal(out, g["out_%d" % c][f]), (c, f)
    snd.late_flag = True
    out = np.ones((2048, 2), np.int16)
    snd.play_buffer(out, 2048, None, None)
    assert (out == 0).all()
    snd.late_flag, snd.rssi = False, -10                       # TX mute (utils_supersdr.py:1142-1147)
    snd.audio_buffer.put(np.full(512, 1000, np.int16))
    out = np.ones((2048, 2), np.int16)
    snd.play_buffer(out, 2048, None, None)
    assert (out == 0).all() and snd.mute_counter == 15


def test_iq_wire_roundtrip_and_batcher_vs_reference_golden():
    from supersdr_amd.iqstream import IQBatcher, iq_body_to_int16, int16_to_wire
    g = np.load(os.path.join(GOLD, "frames.npz"))
    flags, seq, rssi, gps, iq = iq_body_to_int16(g["iq_body"].tobytes())
    assert np.array_equal(iq, g["iq_int16"]) and rssi == float(g["iq_rssi"]) and seq == int(g["iq_seq"])
    assert list(gps) == list(g["iq_gps"])
    assert int16_to_wire(iq, seq=seq, smeter=870, gps=gps) == g["iq_body"].tobytes()
    fed = []
    b = IQBatcher().attach(types.SimpleNamespace(feed=lambda ch, x: fed.append((ch, x.copy()))), 3)
    b._process_iq_samples(5, g["iq_complex64"], -40.0, {})     # what the reference hands to the hook
    b._process_iq_samples(7, g["iq_complex64"], -41.0, {})     # a sequence gap
    assert fed[0][0] == 3 and np.array_equal(fed[0][1], g["iq_int16"]) and b.dropped == 1 and b.last_rssi == -41.0


def test_worker_retry_policy():
    from supersdr_amd.iqstream import GpuKiwiWorker, KiwiServerTerminatedConnection, KiwiTooBusyError, KiwiTimeLimitError

    class Rec:
        def __init__(self, script):
            self.script, self.log = list(script), []

        def connect(self, h, p):
            self.log.append("connect")
            if self.script and self.script[0] == "refuse":
                self.script.pop(0)
                raise OSError("refused")

        def open(self):
            self.log.append("open")

        def run(self):
            step = self.script.pop(0) if self.script else "limit"
            self.log.append("run:" + step)
            if step == "term":
                raise KiwiServerTerminatedConnection("bye")
            if step == "busy":
                raise KiwiTooBusyError("busy")
            if step == "limit":
                raise KiwiTimeLimitError("limit")

        def close(self):
            self.log.append("close")

    def run(script, **opt):
        o = types.SimpleNamespace(connect_retries=2, connect_timeout=0, server_host="h", server_port=1,
                                  is_kiwi_tdoa=False, no_api=False, status=0)
        o.__dict__.update(opt)
        ev = threading.Event()
        ev.set()
        rec = Rec(script)
        w = GpuKiwiWorker(args=(rec, o, ev))
        w._event.wait = lambda timeout=None: None              # do not really sleep 5/15 s
        w.run()
        return rec.log, o, ev

    log, o, ev = run(["ok", "term", "ok", "limit"])
    assert log == ["connect", "open", "run:ok", "run:term", "close", "connect", "open", "run:ok", "run:limit", "close"]
    assert not ev.is_set()
    log, o, ev = run(["refuse", "refuse"])
    assert log == ["connect", "connect", "close"]              # connect_retries exhausted
    log, o, ev = run(["busy"], is_kiwi_tdoa=True)
    assert o.status == 2 and log[-1] == "close"
    log, o, ev = run(["term"], no_api=True)
    assert log == ["connect", "open", "run:term", "close", "close"]


def test_channel_blocks_partition():
    from supersdr_amd.dist import channel_block
    for total, world in ((1 << 20, 8), (65536, 3), (7, 8), (100, 1)):
        blocks = [channel_block(r, world, total) for r in range(world)]
        covered = np.concatenate([np.arange(f, f + n) for f, n in blocks])
        assert np.array_equal(covered, np.arange(total))
        assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1


def test_kiwi_wav_reader_vs_reference_golden():
    from supersdr_amd.iqstream import read_kiwi_iq_wav
    g = np.load(os.path.join(GOLD, "wavreader.npz"))
    blocks, stamps = read_kiwi_iq_wav(g["wav_bytes"].tobytes())
    assert np.array_equal(blocks, g["blocks"]) and blocks.dtype == np.int16
    z = blocks[2:].astype(np.float32).reshape(-1, 2).view(np.complex64).reshape(-1) / 65535      # wavreader.py:84
    assert np.array_equal(z.astype(np.complex64), g["z"])
    assert [s[3] for s in stamps] == [42666667 * i for i in range(5)]
    with pytest.raises(ValueError):
        read_kiwi_iq_wav(b"RIFX" + bytes(40))
