"""Host-side boundary logic (supersdr_amd/workers.py, iqstream.py, dist.py) on CPU.

The GPU engine is replaced by a test double that answers with the oracle (twin C for the two kernels,
NumPy restatements for db2col / play_buffer), so what is tested here is the host plumbing around the
seams: batching, queues, time binning by division, hand-out of the db2col / play_buffer results, the
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
        self.last_wf = np.stack(out) if out else np.zeros((0, self.n_ch, 1024), np.int16)
        return self.last_wf

    def run_audio(self):
        self.pcm, rssi = self.twin.audio(self.iq, self.consts, self.taps, self.state, self.hist)
        return self.pcm, rssi

    # the two post kernels, answered by the oracle's restatements of the reference
    def set_kiwi_rate(self, rate):
        self.kiwi_rate = rate
        self.players = None

    def playbuffer_frame_len(self):
        return 2048 if getattr(self, "kiwi_rate", 12000) == 12000 else 1213

    def run_db2col(self, chans, n_lines):
        out = np.empty((n_lines, self.n_ch, 1024), np.float32)
        for c, k in enumerate(chans):
            for i in range(n_lines):
                spec = self.last_wf[i, c].astype(np.float32) / np.float32(self.n_avg)
                col, lo, hi, dyn, mn, mx = O.spectrum_db2col(
                    spec, int(k.zoom), auto=bool(k.auto_scale), low_clip_db=k.low_clip_db, high_clip_db=k.high_clip_db,
                    dynamic_range=k.dynamic_range, delta_low_db=k.delta_low_db, delta_high_db=k.delta_high_db)
                out[i, c] = col
                k.low_clip_db, k.high_clip_db, k.dynamic_range, k.wf_min_db, k.wf_max_db = lo, hi, dyn, mn, mx
        return out

    def run_playbuffer(self, chans):
        wide = self.playbuffer_frame_len() != 2048
        if getattr(self, "players", None) is None:
            self.players = [O.PlayBufferResampled() if wide else O.PlayBuffer() for _ in range(self.n_ch)]
        L = self.playbuffer_frame_len()
        nf = self.pcm.shape[1] // 512
        out = np.empty((self.n_ch, nf * L, 2), np.int16)
        for c, k in enumerate(chans):
            for f in range(nf):
                out[c, f * L:(f + 1) * L] = self.players[c](self.pcm[c, f * 512:(f + 1) * 512], k.volume, k.balance)
        return out

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
    # db2col has no host implementation: without a result from ssdr_run_db2col the method refuses
    wf.spectrum = np.zeros(1024, np.float32)
    with pytest.raises(RuntimeError):
        wf.spectrum_db2col()
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


def test_play_buffer_hands_out_gpu_blocks():
    hub, wf, snd = make_pair(n_ch=1, channel=0)
    iq = O.synth_iq(1, 2 * 1024, seed=3)
    hub.feed(0, iq[0])
    ref = O.PlayBuffer()
    for f in range(4):                                         # frames arrive with their 48 kHz block attached
        s = snd.process_audio_stream()
        snd.audio_buffer.put(s)
        out = np.zeros((2048, 2), np.int16)
        snd.play_buffer(out, 2048, None, None)
        assert np.array_equal(out, ref(s, volume=snd.volume, balance=snd.audio_balance)), f
    snd.audio_buffer.put(np.zeros(512, np.int16))              # a frame that did not come from the hub: no host path,
    out = np.ones((2048, 2), np.int16)                         # and the PortAudio callback must not raise: silence + stop
    snd.play_buffer(out, 2048, None, None)
    assert (out == 0).all() and snd.terminate and isinstance(snd.error, RuntimeError)
    snd.terminate, snd.error = False, None
    snd.late_flag = True
    out = np.ones((2048, 2), np.int16)
    snd.play_buffer(out, 2048, None, None)
    assert (out == 0).all()
    snd.late_flag = False


def test_play_buffer_wide_rate_blocks_and_tx_mute():
    """20.25 kHz KiwiSDR: KIWI_RATE / SAMPLE_RATIO follow the hub, blocks are 1213 stereo samples (utils_supersdr.py:1211)"""
    from supersdr_amd.workers import IQHub, kiwi_waterfall, kiwi_sound
    hub = IQHub(1, engine=TwinEngine(1), kiwi_rate=20250)
    wf = kiwi_waterfall("gpu", 0, "", 10, 7100.0, Eibi(), Disp(), hub=hub, channel=0, timeout=0.2)
    snd = kiwi_sound(7100.0, "AM", -6000, 6000, "", wf, 4)
    assert snd.KIWI_RATE == 20250 and snd.SAMPLE_RATIO == 48000 / 20250 and int(512 * snd.SAMPLE_RATIO) == 1213
    hub.feed(0, O.synth_iq(1, 1024, seed=4)[0])
    ref = O.PlayBufferResampled()
    for f in range(2):
        s = snd.process_audio_stream()
        snd.audio_buffer.put(s)
        out = np.zeros((1213, 2), np.int16)
        snd.play_buffer(out, 1213, None, None)
        assert np.array_equal(out, ref(s, volume=snd.volume, balance=snd.audio_balance)), f
    hub.feed(0, O.synth_iq(1, 1024, seed=5)[0])
    s = snd.process_audio_stream()
    snd.rssi = -10                                             # TX mute (utils_supersdr.py:1142-1147)
    snd.audio_buffer.put(s)
    out = np.ones((1213, 2), np.int16)
    snd.play_buffer(out, 1213, None, None)
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
