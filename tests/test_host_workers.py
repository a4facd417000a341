"""Host-side boundary logic (supersdr_amd/workers.py, iqstream.py, dist.py) on CPU.

The GPU engine is replaced by a test double that answers with the oracle (twin C for the two kernels,
NumPy restatements for db2col / play_buffer), so what is tested here is the host plumbing around the
seams: batching, queues, time binning by division, hand-out of the db2col / play_buffer results, the
wire <-> int16 conversion, the channel sharding -- and the binding itself: every test that takes the `gpu` fixture runs
twice, with the seams bound over the package's bare `headless` classes and over the REAL reference's
`utils_supersdr.kiwi_waterfall / kiwi_sound` (where /root/reference exists: tests/refload.py); the tests that exercise the
reference's own code through the seams (zoom arithmetic, passband tables, pacing loop, recorder, KiwiWorker) take `ref`."""
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

import refload  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
NEEDS_REF = pytest.mark.skipif(not refload.available(), reason="the reference (utils_supersdr.py) is not on this box")


@pytest.fixture(params=["headless", pytest.param("reference", marks=NEEDS_REF)])
def gpu(request):
    """the seams bound over a pair of worker classes: namespace(kiwi_waterfall, kiwi_sound, module)"""
    from supersdr_amd.workers import bind, bind_headless
    return bind_headless() if request.param == "headless" else bind(refload.load()[0])


@pytest.fixture
def ref():
    """the seams bound over the real reference's classes"""
    if not refload.available():
        pytest.skip("the reference (utils_supersdr.py) is not on this box")
    from supersdr_amd.workers import bind
    return bind(refload.load()[0])


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
            k, t = self.S.compile_params(p, rate=getattr(self, "kiwi_rate", 12000))
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
        self.pcm, rssi, self.flags, self.iqo = self.twin.audio(self.iq, self.consts, self.taps, self.state, self.hist,
                                                               want_flags=True, want_iq=True)
        return self.pcm, rssi

    def audio_iq(self):
        return self.iqo

    def audio_flags(self):
        return self.flags

    def set_wf_lines(self, lines):
        self.last_wf = np.array(lines, np.int16)

    def db2col_line(self, wf_sum, n_avg, k):
        spec = np.asarray(wf_sum, np.int16).astype(np.float32) / np.float32(n_avg)
        col, lo, hi, dyn, mn, mx = O.spectrum_db2col(
            spec, int(k.zoom), auto=bool(k.auto_scale), low_clip_db=k.low_clip_db, high_clip_db=k.high_clip_db,
            dynamic_range=k.dynamic_range, delta_low_db=k.delta_low_db, delta_high_db=k.delta_high_db)
        k.low_clip_db, k.high_clip_db, k.dynamic_range, k.wf_min_db, k.wf_max_db = lo, hi, dyn, mn, mx
        return col

    def set_recording(self, on):
        self.recording = bool(on)

    def playbuffer_mono(self):
        return self.mono

    # the two post kernels, answered by the oracle's restatements of the reference
    def set_kiwi_rate(self, rate):
        self.kiwi_rate = rate
        self.players = None

    def playbuffer_frame_len(self):
        return 2048 if getattr(self, "kiwi_rate", 12000) == 12000 else 1213

    def set_post_channels(self, channels=None):
        """ssdr_set_post_channels: the post-processing works on these channels only, arrays in list order"""
        self.post_sel = None if channels is None else [int(c) for c in channels]

    def _post_list(self):
        sel = getattr(self, "post_sel", None)
        return list(range(self.n_ch)) if sel is None else sel

    def run_db2col(self, chans, n_lines):
        sel = self._post_list()
        out = np.empty((n_lines, len(sel), 1024), np.float32)
        for pos, k in enumerate(list(chans)[:len(sel)]):
            c = sel[pos]
            for i in range(n_lines):
                spec = self.last_wf[i, c].astype(np.float32) / np.float32(self.n_avg)
                col, lo, hi, dyn, mn, mx = O.spectrum_db2col(
                    spec, int(k.zoom), auto=bool(k.auto_scale), low_clip_db=k.low_clip_db, high_clip_db=k.high_clip_db,
                    dynamic_range=k.dynamic_range, delta_low_db=k.delta_low_db, delta_high_db=k.delta_high_db)
                out[i, pos] = col
                k.low_clip_db, k.high_clip_db, k.dynamic_range, k.wf_min_db, k.wf_max_db = lo, hi, dyn, mn, mx
        return out

    def run_playbuffer(self, chans):
        wide = self.playbuffer_frame_len() != 2048
        if getattr(self, "players", None) is None:
            self.players = [O.PlayBufferResampled() if wide else O.PlayBuffer() for _ in range(self.n_ch)]
        L = self.playbuffer_frame_len()
        nf = self.pcm.shape[1] // 512
        sel = self._post_list()
        out = np.empty((len(sel), nf * L, 2), np.int16)
        self.mono = np.empty((len(sel), nf * L), np.int16)
        for pos, k in enumerate(list(chans)[:len(sel)]):
            c = sel[pos]
            for f in range(nf):
                out[pos, f * L:(f + 1) * L] = self.players[c](self.pcm[c, f * 512:(f + 1) * 512], k.volume, k.balance)
                self.mono[pos, f * L:(f + 1) * L] = self.players[c].rec
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


def make_pair(gpu, n_ch=2, channel=1, zoom=10, freq=7100.0, **hub_kw):
    from supersdr_amd.workers import IQHub
    hub = IQHub(n_ch, engine=TwinEngine(n_ch), **hub_kw)
    wf = gpu.kiwi_waterfall("gpu", 0, "", zoom, freq, Eibi(), Disp(), hub=hub, channel=channel, timeout=0.2)
    snd = gpu.kiwi_sound(freq, "USB", 30, 3000, "", wf, 4)
    return hub, wf, snd


def test_the_bound_classes_are_the_maintainers_own(ref):
    """bind(): the seams sit in front of utils_supersdr's classes; everything that is not a seam is the reference's own
    code object (nothing restated in the package), and the constructor of the reference ran as it is."""
    U = ref.module
    assert issubclass(ref.kiwi_waterfall, U.kiwi_waterfall) and issubclass(ref.kiwi_sound, U.kiwi_sound)
    for name in ("set_freq_zoom", "gen_div", "zoom_to_span", "start_frequency_to_counter", "change_passband", "set_white_flag",
                 "bins_to_khz", "offset_to_bin", "keepalive", "start_stream"):
        assert getattr(ref.kiwi_waterfall, name) is getattr(U.kiwi_waterfall, name), name
    for name in ("run", "change_agc_delay", "change_passband", "set_agc_params", "set_mode_freq_pb", "get_audio_chunk", "keepalive"):
        assert getattr(ref.kiwi_sound, name) is getattr(U.kiwi_sound, name), name
    for name in ("receive_spectrum", "spectrum_db2col", "run"):
        assert getattr(ref.kiwi_waterfall, name) is not getattr(U.kiwi_waterfall, name), name
    for name in ("process_audio_stream", "play_buffer"):
        assert getattr(ref.kiwi_sound, name) is not getattr(U.kiwi_sound, name), name
    saved = (U.kiwi_sdr, U.socket, U.wsclient, U.Stream)
    hub, wf, snd = make_pair(ref, n_ch=1, channel=0)
    assert (U.kiwi_sdr, U.socket, U.wsclient, U.Stream) == saved          # the stand-ins were there for the constructors only
    assert wf.wf_stream is wf._gpu_stream and snd.stream is snd._gpu_stream
    assert isinstance(snd.kiwi_filter, U.filtering) and snd.n_tap == 33 and isinstance(snd.audio_rec, U.audio_recording)
    assert wf.wf_stream.zoom == 10                                        # "SET zoom=%d start=%d" of start_stream arrived
    assert hub.engine.param_log[-1][1].mode == 2 and hub.engine.param_log[-1][1].agc_thresh == -80    # "SET mod=usb", "SET agc="


def test_unmodified_reference_classes_run_on_the_gpu_stream(ref):
    """GpuStream alone, without the seams: the reference's OWN receive_spectrum / process_audio_stream parse the W/F and
    SND frames it builds from GPU results (wire format of utils_supersdr.py:782-784, 1065-1074) -- same numbers."""
    from supersdr_amd.workers import IQHub, GpuStream
    U = ref.module
    hub = IQHub(1, engine=TwinEngine(1), gpu_post=False)
    iq = O.synth_iq(1, 2 * 1024, seed=21)
    hub.feed(0, iq[0])
    wf = U.kiwi_waterfall.__new__(U.kiwi_waterfall)                       # no constructor: only the stream and what the method reads
    wf.wf_stream = GpuStream(hub, 0, "W/F", 7100.0, timeout=0.2)
    assert bytes(wf.wf_stream.receive_message()[:3]) == b"W/F"            # the constructor's greeting
    U.kiwi_waterfall.receive_spectrum(wf)
    want = twinlib.load().wf(iq, 1)
    assert wf.spectrum.dtype == np.float32 and np.array_equal(wf.spectrum, want[0, 0].astype(np.float32))
    snd = U.kiwi_sound.__new__(U.kiwi_sound)
    snd.stream = GpuStream(hub, 0, "SND", 7100.0, timeout=0.2)
    snd.run_index, snd.delta_t, snd.terminate, snd.kiwi_wf = 0, 0.0, False, wf
    assert b"MSG audio_init" in bytes(snd.stream.receive_message()) and bytes(snd.stream.receive_message()[:3]) == b"SND"
    eng = hub.engine
    st, hist = twinlib.fresh_state(eng.consts)
    ref_pcm, ref_rssi = twinlib.load().audio(iq, eng.consts, eng.taps, st, hist)
    for f in range(4):
        s = U.kiwi_sound.process_audio_stream(snd)
        assert s.dtype == np.int16 and np.array_equal(s, ref_pcm[0, f * 512:(f + 1) * 512])
        assert abs(snd.rssi - float(ref_rssi[0, f])) <= 0.05 + 1e-9      # the header carries rssi in 0.1 dB steps
        assert snd.adc_overflow_flag is False


def test_seams_deliver_gpu_results_per_channel(gpu):
    hub, wf, snd = make_pair(gpu)
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


def test_waterfall_run_loop_binning_scroll_and_db2col_vs_reference_golden(gpu):
    hub, wf, snd = make_pair(gpu, n_ch=1, channel=0, zoom=8)
    # db2col has no host implementation: without a result from ssdr_run_db2col the method refuses
    wf.spectrum = np.zeros(1024, np.float32)
    with pytest.raises(RuntimeError):
        wf.spectrum_db2col()
    # run loop with N = 3 time binning on the GPU side
    wf.zoom, wf.wf_auto_scaling, wf.delta_low_db, wf.delta_high_db = 8, True, 0, 0
    wf.averaging_n = 3
    iq = O.synth_iq(1, 16 * 1024, seed=9)
    hub.feed(0, iq[0][:1024])                                  # a first line at the old N = 1: stale for an N = 3 client
    hub.set_averaging(3)
    hub.feed(0, iq[0][1024:])
    iq = iq[:, 1024:]
    lines = twinlib.load().wf(iq, 1)[:, 0].astype(np.float32)     # the byte lines the engine double produces
    for k in range(5):
        wf.step()
        assert np.array_equal(wf.spectrum, np.mean(list(lines[3 * k:3 * k + 3]), axis=0))   # == reference np.mean
    assert wf.run_index == 5
    # 3-deep delay buffer, newest line on row 0 (utils_supersdr.py:893-897)
    assert (wf.wf_data[2:] == 0).all() and (wf.wf_data[0] != 0).any() and (wf.wf_data[1] != 0).any()
    if hasattr(wf, "set_white_flag"):                          # the reference's (:875-877)
        wf.set_white_flag()
        assert (wf.wf_data[0] == 255).all()


def test_freq_zoom_arithmetic(ref):
    """the reference's own zoom / tick arithmetic keeps running on the bound object (its "SET zoom= start=" lands in the
    GPU stream, which has nothing to retune)"""
    hub, wf, snd = make_pair(ref, n_ch=1, channel=0, zoom=10, freq=7100.0)
    assert wf.span_khz == 30000 / 1024 and wf.start_f_khz == 7100 - wf.span_khz / 2
    assert wf.set_freq_zoom(14200.0, 0) == 15000 and wf.span_khz == 30000
    assert wf.set_freq_zoom(1.0, 10) == wf.span_khz / 2 and wf.start_f_khz == 0
    assert wf.set_freq_zoom(29999.0, 10) == 30000 - wf.span_khz / 2
    f = wf.set_freq_zoom(7100.0, 8)
    assert f == 7100.0 and wf.eibi.calls[-1] == (wf.start_f_khz, wf.end_f_khz)
    assert wf.bins_to_khz(512) == pytest.approx(7100.0) and wf.offset_to_bin(wf.span_khz) == 1024
    assert wf.deltabins_to_khz(1024) == pytest.approx(wf.span_khz)
    assert wf.counter == round(wf.start_f_khz / 30000 * 2 ** 14 * 1024)
    assert (wf.wf_stream.zoom, wf.wf_stream.start) == (8, wf.counter)
    assert wf.div_list and wf.subdiv_list
    wf.radio_mode = "LSB"
    assert wf.change_passband(10, -20) == (-2980, -40)


def test_sound_control_plane_and_passbands(ref):
    """the reference's set_mode_freq_pb / set_agc_params / change_passband / change_agc_delay, unmodified: their SET commands
    become the channel's ssdr_chan_params"""
    hub, wf, snd = make_pair(ref, n_ch=1, channel=0)
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


def test_play_buffer_hands_out_gpu_blocks(gpu):
    hub, wf, snd = make_pair(gpu, n_ch=1, channel=0)
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


def test_play_buffer_wide_rate_blocks_and_tx_mute(gpu):
    """20.25 kHz KiwiSDR: the stream greets with "MSG audio_init audio_rate=20250" and the worker's own constructor takes
    KIWI_RATE / SAMPLE_RATIO from it (utils_supersdr.py:988-994); blocks are 1213 stereo samples (:1211)"""
    from supersdr_amd.workers import IQHub
    hub = IQHub(1, engine=TwinEngine(1), kiwi_rate=20250)
    wf = gpu.kiwi_waterfall("gpu", 0, "", 10, 7100.0, Eibi(), Disp(), hub=hub, channel=0, timeout=0.2)
    snd = gpu.kiwi_sound(7100.0, "AM", -6000, 6000, "", wf, 4)
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
    b._process_iq_samples(8, g["iq_complex64"], -41.0, {"gpssec": 5, "gpsnsec": 6})
    assert b.last_gps == {"gpssec": 5, "gpsnsec": 6}                       # the frame's GNSS stamp is kept, not dropped


def test_reference_kiwiworker_drives_an_iqbatcher_recorder():
    """kiwi/worker.py:10-79 unchanged: KiwiWorker(args=(recorder, options, run_event)) with a recorder whose IQ hook is
    IQBatcher -- connect / open / run / close and the retry table are the reference's, the IQ lands in the hub."""
    if not refload.available():
        pytest.skip("the reference (kiwi/worker.py) is not on this box")
    _, KW, KC = refload.load()
    from supersdr_amd.iqstream import IQBatcher
    g = np.load(os.path.join(GOLD, "frames.npz"))
    fed = []

    class Rec(IQBatcher):
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
            if step == "ok":
                self._process_iq_samples(len(self.log), g["iq_complex64"], -40.0, {})    # what KiwiSDRStream.run ends in
            if step == "term":
                raise KC.KiwiServerTerminatedConnection("bye")
            if step == "busy":
                raise KC.KiwiTooBusyError("busy")
            if step == "limit":
                raise KC.KiwiTimeLimitError("limit")

        def close(self):
            self.log.append("close")

    def run(script, **opt):
        o = types.SimpleNamespace(connect_retries=2, connect_timeout=0, server_host="h", server_port=1, rigctl_enabled=False,
                                  is_kiwi_tdoa=False, no_api=False, status=0)
        o.__dict__.update(opt)
        ev = threading.Event()
        ev.set()
        rec = Rec(script).attach(types.SimpleNamespace(feed=lambda ch, x: fed.append((ch, x.copy()))), 5)
        w = KW.KiwiWorker(args=(rec, o, ev))
        w._event.wait = lambda timeout=None: None              # do not really sleep 5/15 s
        w.run()
        return rec.log, o, ev

    log, o, ev = run(["ok", "term", "ok", "limit"])
    assert log == ["connect", "open", "run:ok", "run:term", "close", "connect", "open", "run:ok", "run:limit", "close"]
    assert not ev.is_set()
    assert len(fed) == 2 and fed[0][0] == 5 and np.array_equal(fed[0][1], g["iq_int16"])
    log, o, ev = run(["refuse", "refuse"])
    assert log == ["connect", "connect", "close"]              # connect_retries exhausted
    log, o, ev = run(["busy"], is_kiwi_tdoa=True)
    assert o.status == 2 and log[-1] == "close"


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
    # the reader's time axis (wavreader.py:86-99: per-block rate estimate from the GNSS stamps, 0.9 / 0.1 smoothing; VERDICT r3 missing #5)
    from supersdr_amd.iqstream import kiwi_iq_wav_time_axis
    t, rates = kiwi_iq_wav_time_axis(stamps, blocks.shape[1])
    assert t.shape == (3, 512) and np.array_equal(t.reshape(-1), g["t"]) and abs(rates[-1] - 12000.0) < 0.01


# ------------------------------------------------------------------ round 2: hardening of the host shim
@pytest.mark.timeout(60)
def test_sound_run_pacing_late_drop_and_refill_cycle(ref, monkeypatch):
    """the reference's kiwi_sound.run (utils_supersdr.py:1150-1186), unmodified, through a whole late / refill cycle on a scripted clock: frames on
    time are queued; a stall longer than (FULL_BUFF_LEN + 2) frames sets late_flag, frames read while late are dropped
    (with their 48 kHz blocks: nothing is left behind) and play_buffer plays silence; once the delay is worked off the
    queue is refilled to FULL_BUFF_LEN and playback resumes."""
    from supersdr_amd.workers import Frame
    hub, wf, snd = make_pair(ref, n_ch=1, channel=0)
    clock = {"ns": 10 ** 12}
    monkeypatch.setattr(ref.module.time, "time_ns", lambda: clock["ns"])
    ms = 512 / 12000 * 1000
    # (read duration in ms) per get_audio_chunk call: 8 on time, one 400 ms stall, then fast reads
    script = [ms] * 8 + [400.0] + [1.0] * 40
    log = []

    def get_audio_chunk():
        i = len(log)
        if i >= len(script):
            snd.terminate = True
            return None
        clock["ns"] += int(script[i] * 1e6)
        while snd.audio_buffer.full() and not snd.late_flag:     # the PortAudio callback drains the queue while playing
            snd.audio_buffer.get()
        f = Frame.make(np.full(512, i, np.int16), -60.0, play_block=np.full((2048, 2), i, np.int16))
        log.append((i, snd.late_flag, snd.audio_buffer.qsize()))
        return f

    snd.get_audio_chunk = get_audio_chunk
    seen_late = []
    orig_put = snd.audio_buffer.put
    snd.audio_buffer.put = lambda x, *a, **k: (seen_late.append((int(x[0]), snd.late_flag)), orig_put(x, *a, **k))[1]
    snd.run()
    late_at = [i for i, late, _ in log if late]
    assert late_at and late_at[0] == 9                            # the call after the 400 ms stall sees late_flag
    assert snd.FULL_BUFF_LEN == 4 and 400.0 > (snd.FULL_BUFF_LEN + 2) * ms
    queued = [i for i, _ in seen_late]
    dropped = [i for i in range(len(script)) if i not in queued]
    assert dropped and dropped[0] == 9 and dropped == list(range(9, 9 + len(dropped)))      # a contiguous run of dropped frames
    # (400 - 42.7) ms of delay is worked off at (42.7 - 1) ms per dropped frame: 8 or 9 frames
    assert 7 <= len(dropped) <= 10
    assert not snd.late_flag                                      # refilled and reset (utils_supersdr.py:1183-1186)
    assert all(q == snd.FULL_BUFF_LEN for i, late, q in log if late)    # nothing is consumed while late: the queue stays full
    first_after = queued[queued.index(8) + 1]
    assert first_after == dropped[-1] + 1                         # playback resumes with the frame after the dropped run
    # while late the callback plays silence and consumes nothing
    snd.late_flag = True
    out = np.ones((2048, 2), np.int16)
    n0 = snd.audio_buffer.qsize()
    snd.play_buffer(out, 2048, None, None)
    assert (out == 0).all() and snd.audio_buffer.qsize() == n0


def test_adc_overflow_flag_drift_skip_and_recording_branch(gpu, tmp_path, monkeypatch):
    """the SND header semantics at the seam: adc_overflow_flag follows the frame's flag (utils_supersdr.py:1066-1067), the
    sample-rate drift rule reads and discards one frame (:1049-1052), and while audio_rec records, play_buffer appends
    the mono block that the same GPU run produced (:1139-1140) -- written out as a 48 kHz mono WAV by stop()."""
    import wave
    hub, wf, snd = make_pair(gpu, n_ch=1, channel=0)
    iq = O.synth_iq(1, 3 * 1024, seed=3)
    iq[0, 1024 + 700, 0] = 32767                                  # frame 3 (second frame of superframe 1) clips
    hub.feed(0, iq[0][:2048])
    flags = []
    for f in range(4):
        snd.process_audio_stream()
        flags.append(snd.adc_overflow_flag)
    assert flags == [False, False, False, True]
    # drift: delta_t = KIWI_RATE_TRUE - KIWI_RATE; after run_index frames the surplus reaches one frame -> skip one
    hub.feed(0, iq[0][2048:])
    snd.delta_t, snd.run_index = 12.0, 1001                       # 1001 * 12 * 512 / 12000 = 512.5 >= 512
    a = snd.process_audio_stream()
    assert snd.run_index == 0
    twin = twinlib.load()
    st, hist = twinlib.fresh_state(hub.engine.consts)
    ref_pcm, _ = twin.audio(iq, hub.engine.consts, hub.engine.taps, st, hist)
    assert np.array_equal(a, ref_pcm[0, 5 * 512:6 * 512])         # frame 4 was read and discarded
    snd.delta_t = 0.0
    # recording
    monkeypatch.chdir(tmp_path)
    hub2, wf2, snd2 = make_pair(gpu, n_ch=1, channel=0)
    if hasattr(snd2.audio_rec, "start"):
        snd2.audio_rec.start()                                    # the reference's recorder (utils_supersdr.py:151-155)
    else:
        snd2.audio_rec.recording_flag = True
    assert snd2.audio_rec.recording_flag
    hub2.feed(0, iq[0][:2048])
    ref = O.PlayBuffer()
    want = []
    for f in range(4):
        s = snd2.process_audio_stream()
        snd2.audio_buffer.put(s)
        out = np.zeros((2048, 2), np.int16)
        snd2.play_buffer(out, 2048, None, None)
        ref(s, volume=snd2.volume, balance=snd2.audio_balance)
        want.append(ref.rec.copy())
    assert len(snd2.audio_rec.audio_buffer) == 4
    assert np.array_equal(np.concatenate(snd2.audio_rec.audio_buffer), np.concatenate(want))
    if not hasattr(snd2.audio_rec, "stop"):
        return
    snd2.audio_rec.stop()                                         # ... and its WAV writer (:157-172)
    with wave.open(snd2.audio_rec.filename, "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 48000)
        data = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    assert np.array_equal(data, np.concatenate(want))


def test_unknown_mode_and_out_of_band_tuning_are_errors(gpu):
    hub, wf, snd = make_pair(gpu, n_ch=1, channel=0)
    n0 = len(hub.engine.param_log)
    snd.radio_mode = "SAM"
    with pytest.raises(ValueError, match="no demodulator"):
        snd.set_mode_freq_pb()
    snd.radio_mode = "AM"
    snd.freq = wf.freq + 6.5                                       # 6.5 kHz from the centre of a 12 kHz band
    with pytest.raises(ValueError, match="outside"):
        snd.set_mode_freq_pb()
    assert len(hub.engine.param_log) == n0                         # neither reached the engine: nothing aliased silently
    snd.freq = wf.freq - 5.9
    snd.set_mode_freq_pb()
    assert hub.engine.param_log[-1][1].f_shift_hz == pytest.approx(-5900.0)
    # the library refuses it too
    import supersdr_amd as S
    with pytest.raises(S.SsdrError):
        S.compile_params(S.default_params("usb", f_shift_hz=6000.5))
    # the true axis of the GPU waterfall: 12 kHz around the channel centre, whatever the zoom says
    wf.set_freq_zoom(7100.0, 3)
    assert wf.iq_bin_to_khz(512) == 7100.0 and wf.iq_bin_to_khz(0) == 7094.0 and wf.iq_khz_to_bin(7106.0) == 1024


def test_two_waterfall_clients_with_different_averaging_do_not_disturb_each_other(gpu):
    """averaging is per kiwi_waterfall (utils_supersdr.py:881-886).  Client 0 wants N = 1, client 1 wants
    N = 3 on the same hub: the GPU then delivers single lines, client 1 takes the reference's mean of 3 of them and
    gets its db2col from the GPU for the binned line; nobody's group is restarted, nobody spins."""
    from supersdr_amd.workers import IQHub
    hub = IQHub(2, engine=TwinEngine(2), trace_rows=0)
    a = gpu.kiwi_waterfall("gpu", 0, "", 10, 7100.0, Eibi(), Disp(), hub=hub, channel=0, timeout=0.2)
    b = gpu.kiwi_waterfall("gpu", 0, "", 10, 7100.0, Eibi(), Disp(), hub=hub, channel=1, timeout=0.2)
    b.averaging_n = 3
    iq = O.synth_iq(2, 6 * 1024, seed=12)
    hub.set_averaging(1, 0)
    hub.set_averaging(3, 1)
    assert hub.averaging_n == 1                                    # they disagree: the GPU sums nothing
    for k in range(6):
        for c in range(2):
            hub.feed(c, iq[c, k * 1024:(k + 1) * 1024])
    lines = twinlib.load().wf(iq, 1).astype(np.float32)            # [6, 2, 1024]
    for k in range(6):
        a.step()
        assert np.array_equal(a.spectrum, lines[k, 0])
    for k in range(2):
        b.step()
        assert np.array_equal(b.spectrum, np.mean(list(lines[3 * k:3 * k + 3, 1]), axis=0))
        col = O.spectrum_db2col(b.spectrum, b.zoom)[0]
        assert np.array_equal(b.wf_color, col)
    assert hub.averaging_n == 1
    # when they agree the GPU does the summing
    a.averaging_n = 3
    hub.set_averaging(3, 0)
    assert hub.averaging_n == 3


def test_a_stalled_receiver_does_not_freeze_the_hub_and_backlogs_are_bounded():
    """one receiver that stops feeding (reconnect sleeps of 5-15 s in GpuKiwiWorker) must not freeze the
    others nor grow their buffers without bound: the hub runs once a healthy channel is 4 superframes ahead, the lagging
    channel gets zero-filled superframes, and its samples resume in order when it comes back."""
    from supersdr_amd.workers import IQHub
    hub = IQHub(2, engine=TwinEngine(2), gpu_post=False)
    iq = O.synth_iq(2, 12 * 1024, seed=14)
    hub.feed(0, iq[0, :1024])
    hub.feed(1, iq[1, :1024])
    assert hub.superframes == 1
    for k in range(1, 8):                                          # channel 1 goes quiet
        hub.feed(0, iq[0, k * 1024:(k + 1) * 1024])
    assert hub.superframes >= 4 and hub.stalled[1] == hub.superframes - 1 and hub.stalled[0] == 0
    assert hub.backlog(0) <= 4 * 1024                            # backlog bounded by the stall threshold
    got0 = np.concatenate([hub.snd_queue[0].get_nowait() for _ in range(2 * hub.superframes)])
    st, hist = twinlib.fresh_state(hub.engine.consts[:1])
    ref, _ = twinlib.load().audio(iq[:1, :hub.superframes * 1024], hub.engine.consts[:1], hub.engine.taps[:1], st, hist)
    assert np.array_equal(got0, ref[0])                            # the healthy channel's audio is uninterrupted
    # a burst far beyond the ring on a hub whose other channel stays silent: bounded memory, oldest dropped, counted
    hub2 = IQHub(2, engine=TwinEngine(2), gpu_post=False, backlog_superframes=4, stall_superframes=100)
    hub2.feed(0, iq[0])
    assert hub2.ring_capacity == 4 * 1024 and hub2.dropped[0] == 8 * 1024 and hub2.superframes == 0


def test_mod_iq_frames_reach_a_kiwiclient_through_the_gpu_stream():
    """"SET mod=iq" (kiwi/client.py:217-249): the channel's filtered, gain-controlled baseband as I,Q pairs.  The GpuStream
    builds mod=iq SND frames from them, and the reference's own KiwiSDRStream._process_aud decodes those into exactly the
    complex samples (and the GNSS dict) its _process_iq_samples hook expects."""
    from supersdr_amd.workers import IQHub, GpuStream
    from supersdr_amd import _lib as L
    hub = IQHub(2, engine=TwinEngine(2), gpu_post=False)
    iq = O.synth_iq(2, 2 * 1024, seed=33, modes=[1, 1])
    st0 = GpuStream(hub, 0, "SND", 7100.0, timeout=0.2)
    st0.send_message("SET mod=iq low_cut=-5000 high_cut=5000 freq=7100.500")
    assert hub.params(0).mode == L.MODE_IQ and hub.params(1).mode == L.MODE_AM
    for c in range(2):
        hub.feed(c, iq[c])
    eng = hub.engine
    st, hist = twinlib.fresh_state(eng.consts)
    pcm_t, rssi_t, iq_t = twinlib.load().audio(iq, eng.consts, eng.taps, st, hist, want_iq=True)
    assert np.array_equal(iq_t[0, :, 0], pcm_t[0]) and (iq_t[1] == 0).all() and np.abs(iq_t[0, :, 1]).max() > 1000
    st0.receive_message(), st0.receive_message()                          # the greeting
    frames = [st0.receive_message() for _ in range(4)]
    assert all(len(f) == 3 + 7 + 10 + 2048 for f in frames)
    if refload.available():
        _, _, KC = refload.load()
        got = []

        class Rec(KC.KiwiSDRStream):
            def _process_iq_samples(self, seq, samples, rssi, gps):
                got.append((seq, samples.copy(), rssi, gps))

        r = Rec()
        r._options = types.SimpleNamespace(ADC_OV=False, S_meter=-1, sdt=0, sound=True, raw=False, tstamp=False, stats=False)
        r._modulation, r._s_meter_valid, r._compression = "iq", True, False
        for f in frames:
            r._process_aud(f[3:])
        z = np.concatenate([g[1] for g in got])
        assert np.array_equal(z.real, iq_t[0, :, 0].astype(np.float32)) and np.array_equal(z.imag, iq_t[0, :, 1].astype(np.float32))
        assert set(got[0][3]) == {"last_gps_solution", "dummy", "gpssec", "gpsnsec"} and [g[0] for g in got] == [1, 2, 3, 4]
    # the other channel keeps sending ordinary PCM frames
    st1 = GpuStream(hub, 1, "SND", 7100.0, timeout=0.2)
    st1.receive_message(), st1.receive_message()
    assert len(st1.receive_message()) == 3 + 7 + 1024


# ---------------------------------------------------------------------------------------------------------------
# round 4: the hub's bulk ingest (slot ring, NumPy bookkeeping, queues only for attached channels)
# ---------------------------------------------------------------------------------------------------------------
class RecordingEngine:
    """Zero-cost engine double: remembers the batches it was pushed and answers with arrays that name them (so a test can
    tell which batch a result belongs to).  No arithmetic: what is tested is the hub."""

    def __init__(self, n_ch, keep=True):
        self.n_ch, self.keep, self.batches, self.runs = n_ch, keep, [], 0
        self.pcm = np.zeros((n_ch, 1024), np.int16)
        self.rssi = np.zeros((n_ch, 2), np.float32)
        self.fl = np.zeros((n_ch, 2), np.uint8)

    def set_kiwi_rate(self, rate):
        pass

    def set_averaging(self, n):
        pass

    def playbuffer_frame_len(self):
        return 2048

    def push_iq(self, batch):
        assert batch.flags.c_contiguous and batch.dtype == np.int16 and batch.shape[0] == self.n_ch
        self.runs += 1
        if self.keep:
            self.batches.append(batch.copy())
        self.frames = batch.shape[1] // 512

    def push_iq_wire(self, bodies):
        assert bodies.flags.c_contiguous and bodies.dtype == np.uint8 and bodies.shape[2] == 2065
        self.runs += 1
        self.batches.append(bodies.copy())
        self.frames = bodies.shape[1]
        return np.zeros((self.n_ch, self.frames), np.float32)

    def run_wf(self):
        if getattr(self, "wf", None) is None or len(self.wf) != self.frames // 2:
            self.wf = np.zeros((self.frames // 2, self.n_ch, 1024), np.int16)
        self.wf[:, :, 0] = self.runs
        return self.wf

    def run_audio(self):
        nf = self.frames
        if self.pcm.shape[1] != nf * 512:
            self.pcm, self.rssi, self.fl = np.zeros((self.n_ch, nf * 512), np.int16), np.zeros((self.n_ch, nf), np.float32), np.zeros((self.n_ch, nf), np.uint8)
        self.pcm[:, 0] = self.runs
        return self.pcm, self.rssi

    def audio_flags(self):
        return self.fl

    def close(self):
        pass


class ScanHub:
    """The hub's batching rule stated the slow way, one Python list per channel (rounds 1-3's implementation without the
    drop path): run when every channel has a superframe, or when one is `stall` superframes ahead -- then a channel that has
    none gets silence and keeps what it had."""

    def __init__(self, n_ch, stall=4, sf=1024):
        self.n, self.sf, self.stall = n_ch, sf, stall * sf
        self.buf = [np.zeros((0, 2), np.int16) for _ in range(n_ch)]
        self.batches, self.stalled = [], [0] * n_ch

    def feed(self, c, iq):
        pos = 0
        while pos < len(iq):
            m = min(len(iq) - pos, self.sf)
            self.buf[c] = np.concatenate([self.buf[c], iq[pos:pos + m]])
            pos += m
            self.pump()

    def pump(self):
        while True:
            avail = [len(b) for b in self.buf]
            if min(avail) < self.sf and max(avail) < self.stall + self.sf:
                return
            batch = np.zeros((self.n, self.sf, 2), np.int16)
            for c in range(self.n):
                if len(self.buf[c]) >= self.sf:
                    batch[c] = self.buf[c][:self.sf]
                    self.buf[c] = self.buf[c][self.sf:]
                else:
                    self.stalled[c] += 1
            self.batches.append(batch)


def test_hub_batches_equal_the_per_channel_scan_on_ragged_stalling_feeds():
    """The slot ring forms exactly the batches the per-channel scan forms: ragged feeds, receivers that fall behind and
    come back with half a superframe buffered, blocks of channels fed in step, in-place reserve / commit."""
    from supersdr_amd.workers import IQHub
    rng = np.random.default_rng(20)
    n = 7
    for trial in range(6):
        eng = RecordingEngine(n)
        hub = IQHub(n, engine=eng, gpu_post=False, backlog_superframes=8, stall_superframes=3)
        ref = ScanHub(n, stall=3)
        for step in range(120):
            kind = rng.integers(0, 10)
            m = int(rng.choice([1, 37, 512, 700, 1024, 1500, 2300]))
            if kind < 5:                                           # one channel; channel 6 is often silent for long
                c = int(rng.integers(0, n if step % 40 < 10 else n - 1))
                iq = rng.integers(-3000, 3000, (m, 2)).astype(np.int16)
                hub.feed(c, iq)
                ref.feed(c, iq)
            else:                                                  # a block of channels, whatever their positions
                f = int(rng.integers(0, n - 1))
                k = int(rng.integers(1, n - f))
                iq = rng.integers(-3000, 3000, (k, m, 2)).astype(np.int16)
                v = hub.reserve(f, k) if kind == 9 else None
                if v is not None and v.shape[1] >= m:
                    v[:, :m] = iq
                    hub.commit(f, k, m)
                else:
                    hub.feed_block(f, iq)
                for i in range(k):                                 # the scan hub takes the block channel by channel, run by run:
                    pass                                           # (see below: same order of pump calls per run of equal positions)
                _feed_block_like_the_hub(ref, hub, f, iq)
        assert eng.runs == len(ref.batches) and eng.runs > 20
        for a, b in zip(eng.batches, ref.batches):
            assert np.array_equal(a, b)
        assert list(hub.stalled) == ref.stalled and sum(ref.stalled) > 0 and not hub.dropped.any()


def _feed_block_like_the_hub(ref, hub, first, iq):
    """the reference scan has no block call: a block is its channels' feeds, piece by slot-sized piece, all channels of a
    run of equal write positions advancing together (which is what a block of receivers fed in step means)"""
    k, m = iq.shape[0], iq.shape[1]
    lens = [len(ref.buf[first + i]) for i in range(k)]
    runs, lo = [], 0
    for i in range(1, k + 1):
        if i == k or lens[i] != lens[lo]:
            runs.append((lo, i))
            lo = i
    for lo, hi in runs:
        pos = 0
        while pos < m:
            have = len(ref.buf[first + lo]) % ref.sf
            step = min(m - pos, ref.sf - have)
            for i in range(lo, hi):
                ref.buf[first + i] = np.concatenate([ref.buf[first + i], iq[i, pos:pos + step]])
            pos += step
            ref.pump()


def test_hub_scales_to_131072_channels_and_makes_objects_only_for_attached_channels():
    """VERDICT r3 'missing #1': one superframe of 131 072 receivers through the hub's own API inside the 85.3 ms that real
    time allows (zero-cost engine: what is timed is the host side).  In place (reserve / commit) the hub's share is
    bookkeeping; through feed_block it is one copy of the 512 MiB, held against a bare np.copyto of the same arrays."""
    import time
    from supersdr_amd.workers import IQHub
    n = 131072
    eng = RecordingEngine(n, keep=False)
    hub = IQHub(n, engine=eng, gpu_post=False, backlog_superframes=2, stall_superframes=1)
    assert hub.wf_queue.attached(7) is None and not hub._snd_att          # nobody listens: no queue, no Frame, anywhere
    q = hub.attach(77, wf=True, snd=True)
    blk = np.ones((n, 1024, 2), np.int16)
    for _ in range(2):
        hub.feed_block(0, blk)                                             # first touch of both slots
    t0 = time.perf_counter()
    np.copyto(hub._slots[0], blk)
    bare = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(3):
        hub.feed_block(0, blk)
    block_s = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        v = hub.reserve(0, n)
        assert v.shape == (n, 1024, 2)
        hub.commit(0, n, 1024)
    inplace_s = (time.perf_counter() - t0) / 3
    print("131072 ch: feed_block %.1f ms / superframe (bare copy %.1f ms), reserve+commit %.2f ms" % (block_s * 1e3, bare * 1e3, inplace_s * 1e3))
    assert eng.runs == 8 and hub.superframes == 8
    assert inplace_s < 0.0853 / 4, "the hub's own share of a superframe"
    assert block_s < 1.5 * bare + 0.010, "feed_block = one copy + bookkeeping"
    assert q["snd"].qsize() == 16 and q["wf"].qsize() == 8 and hub.last.pcm.shape == (n, 1024)
    # a block of 1000 receivers goes quiet: the others keep running, the quiet ones are counted, all of it vectorised
    t0 = time.perf_counter()
    for _ in range(3):
        hub.feed_block(0, blk[:50000])
        hub.feed_block(51000, blk[51000:])
    stall_s = (time.perf_counter() - t0) / 3
    assert hub.superframes == 8 + 2 and hub.stalled[50500] == 2 and hub.stalled[0] == 0 and stall_s < 0.5
    hub.close()


def test_wire_hub_batches_snd_bodies_and_lazy_queues_follow_clients(gpu):
    """a hub built with wire=True takes SND bodies (2065 B per frame) into its slots as they are; and on a lazy hub a
    channel gets queues exactly when a worker (or a GpuStream) attaches to it"""
    from supersdr_amd.workers import IQHub, GpuStream
    eng = RecordingEngine(6)
    hub = IQHub(6, engine=eng, gpu_post=False, wire=True, lazy=True)
    rng = np.random.default_rng(3)
    bodies = rng.integers(0, 256, (6, 4, 2065)).astype(np.uint8)
    hub.feed_wire_block(0, bodies[:, :1])
    assert eng.runs == 0
    hub.feed_wire_block(0, bodies[:4, 1:3])                              # four channels run ahead
    hub.feed_wire_block(4, bodies[4:, 1:2])
    assert eng.runs == 1 and np.array_equal(eng.batches[0], bodies[:, :2])
    with pytest.raises(ValueError):
        hub.feed(0, np.zeros((10, 2), np.int16))
    assert hub.snd_queue.attached(2) is None
    st = GpuStream(hub, 2, "SND", 7100.0, timeout=0.1)
    assert hub.snd_queue.attached(2) is not None and hub.wf_queue.attached(2) is None and hub._snd_att == [2]
    hub.feed_wire_block(4, bodies[4:, 2:4])
    hub.feed_wire_block(0, bodies[:4, 3:4])
    assert eng.runs == 2 and np.array_equal(eng.batches[1], bodies[:, 2:4])
    st.receive_message(), st.receive_message()
    assert st.receive_message()[:3] == b"SND" and hub.snd_queue[2].qsize() == 1
    # a worker pair on a lazy sample hub: its channel, and only its channel, is attached
    hub2 = IQHub(5, engine=TwinEngine(5), lazy=True)
    wf = gpu.kiwi_waterfall("gpu", 0, "", 10, 7100.0, Eibi(), Disp(), hub=hub2, channel=3, timeout=0.2)
    snd = gpu.kiwi_sound(7100.0, "USB", 30, 3000, "", wf, 4)
    assert hub2._wf_att == [3] and hub2._snd_att == [3] and hub2._n_wf_clients == 1 and hub2._n_snd_clients == 1
    iq = O.synth_iq(5, 2048, seed=5)
    hub2.feed_block(0, iq)
    wf.step(), wf.step()
    assert snd.process_audio_stream().shape == (512,) and wf.run_index == 2
    hub2.detach(3)
    assert hub2._wf_att == [] and hub2.wf_queue.attached(3) is None


@pytest.mark.parametrize("rate", [12000, 20250])
def test_start_audio_stream_opens_the_bound_worker_at_both_rates(ref, rate, monkeypatch):
    """VERDICT r3 'missing #3': the reference's own start_audio_stream (utils_supersdr.py:1188-1215), unmodified, on the GPU-backed
    kiwi_sound: its thread runs the reference's pacing loop over the GPU stream until the jitter buffer is full, then it opens
    sd.OutputStream with the block size of the server's rate (2048 at 12 kHz, 1213 = int(512 * 48000 / 20250) at 20.25 kHz)
    and the bound play_buffer as callback -- which a PortAudio stand-in then calls the way PortAudio does: outdata[blocksize, 2]
    int16, filled in place with the 48 kHz blocks the GPU path produced."""
    from supersdr_amd.workers import IQHub
    U = ref.module
    hub = IQHub(1, engine=TwinEngine(1), kiwi_rate=rate)
    wf = ref.kiwi_waterfall("gpu", 0, "", 10, 7100.0, Eibi(), Disp(), hub=hub, channel=0, timeout=0.5)
    snd = ref.kiwi_sound(7100.0, "AM", -6000, 6000, "", wf, 4)
    opened = []

    class OutputStream:                                            # what sounddevice.OutputStream is to start_audio_stream
        def __init__(self, **kw):
            self.kw, self.started = kw, False
            opened.append(self)

        def start(self):
            self.started = True

    fake_sd = types.SimpleNamespace(OutputStream=OutputStream,
                                    query_devices=lambda: [{"name": "pulse", "max_input_channels": 2}, {"name": "hw:0", "max_input_channels": 0}])
    monkeypatch.setattr(U, "sd", fake_sd)
    iq = O.synth_iq(1, 12 * 1024, seed=40)[0]
    hub.feed(0, iq)                                                # 24 frames wait in the channel's queue
    ok, stream = U.start_audio_stream(snd)
    assert ok is True and stream is opened[0] and stream.started and not snd.terminate
    L = 2048 if rate == 12000 else 1213
    kw = stream.kw
    assert kw["blocksize"] == L and kw["samplerate"] == 48000 and kw["channels"] == 2 and kw["dtype"] == snd.FORMAT
    assert kw["callback"].__func__ is type(snd).play_buffer and kw["callback"].__self__ is snd      # the GPU seam, bound
    assert snd.audio_buffer.qsize() >= snd.FULL_BUFF_LEN
    # PortAudio's side: block after block, in place
    eng = hub.engine
    st, hist = twinlib.fresh_state(eng.consts)
    pcm, _ = twinlib.load().audio(iq[None], eng.consts, eng.taps, st, hist)
    player = O.PlayBuffer() if rate == 12000 else O.PlayBufferResampled()
    for f in range(6):
        out = np.full((L, 2), 77, np.int16)
        kw["callback"](out, L, None, None)
        assert np.array_equal(out, player(pcm[0, f * 512:(f + 1) * 512], volume=snd.volume, balance=snd.audio_balance)), f
    snd.terminate = True                                           # the run thread ends with its next frame (or its queue time-out)


def test_lazy_hub_post_processes_its_listeners_only_and_they_see_what_they_always_saw(gpu):
    """the hub side of ssdr_set_post_channels on the CPU (engine double): a lazy hub names the channels with workers to the
    engine, keeps its display-state arrays in that order, and the workers get the same colours, clip levels and 48 kHz blocks
    as on a hub that post-processes everybody -- also after a third worker attaches mid-stream and one leaves."""
    from supersdr_amd.workers import IQHub
    n_ch = 9
    iq = O.synth_iq(n_ch, 6 * 1024, seed=91)
    seen = {}
    for lazy in (False, True):
        eng = TwinEngine(n_ch)
        hub = IQHub(n_ch, engine=eng, lazy=lazy)
        wfs = {c: gpu.kiwi_waterfall("gpu", 0, "", 6 + c, 7100.0, Eibi(), Disp(), hub=hub, channel=c, timeout=0.2) for c in (2, 7)}
        snds = {c: gpu.kiwi_sound(7100.0, "AM", -6000, 6000, "", wfs[c], 8) for c in (2, 7)}
        snds[7].volume, snds[7].audio_balance = 140, -0.5
        got = []
        for k in range(6):
            if k == 3:                                             # a third listener arrives, one leaves
                wfs[4] = gpu.kiwi_waterfall("gpu", 0, "", 3, 7100.0, Eibi(), Disp(), hub=hub, channel=4, timeout=0.2)
                snds[4] = gpu.kiwi_sound(7100.0, "AM", -6000, 6000, "", wfs[4], 8)
                hub.wf_clients[2] = None
                hub.snd_clients[2] = None
                if not lazy:                                       # (a hub that queues for everybody holds channel 4's earlier lines, made
                    while hub.wf_queue[4].qsize():                 #  before anybody looked: a viewer that arrives now starts with the next one)
                        hub.wf_queue[4].get_nowait()
                    while hub.snd_queue[4].qsize():
                        hub.snd_queue[4].get_nowait()
            hub.feed_block(0, iq[:, k * 1024:(k + 1) * 1024])
            if lazy:
                assert hub.post_channels == ([2, 7] if k < 3 else [4, 7]) and eng.post_sel == hub.post_channels
                assert hub.last.color.shape[1] == 2 and hub.last.play.shape[0] == 2
            for c in ((7,) if k < 3 else (4, 7)):
                wfs[c].step()
                got.append((wfs[c].wf_color.copy(), wfs[c].wf_min_db, wfs[c].wf_max_db))
                for f in range(2):
                    fr = snds[c].process_audio_stream()
                    # (a listener's very first block starts from play_buffer's zero history, utils_supersdr.py:1005, on the lazy hub -- as a
                    #  new kiwi_sound does -- and from three superframes of history on the hub that interpolated for nobody: not compared)
                    first_block = c == 4 and k == 3 and f == 0
                    got.append((np.asarray(fr).copy(), fr.play_block[64:].copy() if first_block else fr.play_block.copy()))
        seen[lazy] = got
    assert len(seen[False]) == len(seen[True]) > 20
    for a, b in zip(seen[False], seen[True]):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_lazy_hub_selection_follows_manual_attach_and_detach(gpu):
    """Round 5 (advisor): on a lazy hub the post selection was re-derived on client changes only.  A worker is set on a channel,
    the channel is detached before the next superframe (the selection is computed without it) and attached again by hand:
    _sync_display_state then looked the channel up in a selection that did not hold it (KeyError on the feeding thread).  The
    selection now follows attach() / detach() as well, and a channel outside it is skipped."""
    from supersdr_amd.workers import IQHub
    n_ch = 6
    iq = O.synth_iq(n_ch, 4 * 1024, seed=92)
    hub = IQHub(n_ch, engine=TwinEngine(n_ch), lazy=True)
    wf = gpu.kiwi_waterfall("gpu", 0, "", 5, 7100.0, Eibi(), Disp(), hub=hub, channel=3, timeout=0.2)
    snd = gpu.kiwi_sound(7100.0, "AM", -6000, 6000, "", wf, 8)
    hub.detach(3)                                              # gone before the first superframe: the selection is derived without it
    hub.feed_block(0, iq[:, :1024])
    assert hub.post_channels == []
    hub.attach(3, wf=True, snd=True)                           # back by hand: its workers are still the channel's clients
    hub.feed_block(0, iq[:, 1024:2048])                        # (used to raise KeyError: 3)
    assert hub.post_channels == [3] and hub.last.color.shape[1] == 1
    wf.step()
    assert wf.wf_color.shape == (1024,) and snd.process_audio_stream().play_block.shape[0] == 2048
    hub.close()


def test_hub_fed_from_many_threads_delivers_every_stream_in_order():
    """the reference's ingest is one thread per receiver (supersdr.py:121, utils_supersdr.py:1198): eight feeder threads, each with
    its own block of receivers and its own chunking, push into one hub at once; every receiver's samples come out of the
    batches in order, none lost, none doubled, and nobody was declared stalled (generous stall threshold)."""
    from supersdr_amd.workers import IQHub
    n_thr, per, n_sf = 8, 5, 12
    n_ch = n_thr * per
    eng = RecordingEngine(n_ch)
    hub = IQHub(n_ch, engine=eng, gpu_post=False, backlog_superframes=16, stall_superframes=14)
    # sample value = (channel, running index) so that order is checkable: I = channel, Q = index mod 2^15
    def stream(c):
        idx = np.arange(n_sf * 1024)
        return np.stack([np.full_like(idx, c), idx % 32768], axis=1).astype(np.int16)
    errs = []

    def feeder(t):
        try:
            rng = np.random.default_rng(t)
            first = t * per
            data = np.stack([stream(first + i) for i in range(per)])
            pos = 0
            while pos < n_sf * 1024:
                m = int(min(rng.choice([64, 512, 700, 1024, 1500]), n_sf * 1024 - pos))
                if rng.random() < 0.5:
                    hub.feed_block(first, data[:, pos:pos + m])
                else:
                    for i in range(per):
                        hub.feed(first + i, data[i, pos:pos + m])
                pos += m
        except Exception as e:                                     # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=feeder, args=(t,)) for t in range(n_thr)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(60)
    assert not errs and eng.runs == n_sf and not hub.stalled.any() and not hub.dropped.any()
    got = np.concatenate(eng.batches, axis=1)                      # [n_ch, n_sf * 1024, 2]
    for c in range(n_ch):
        assert np.array_equal(got[c], stream(c)), c


def test_a_reservation_overtaken_by_the_stall_rule_is_refused_not_misfiled():
    """reserve() hands out a view of the open slot; if the block then sits on it until the others are `stall_superframes` ahead,
    its superframe runs zero-filled and the view is a slot that has left: commit() says so (False, samples counted as dropped)
    instead of advancing the channel over samples that are not where the next run will read them."""
    from supersdr_amd.workers import IQHub
    eng = RecordingEngine(3)
    hub = IQHub(3, engine=eng, gpu_post=False, backlog_superframes=8, stall_superframes=2)
    rng = np.random.default_rng(9)
    v = hub.reserve(2, 1)
    assert v is not None and v.shape[0] == 1
    for _ in range(3):                                                   # channels 0 and 1 run ahead: the hub stops waiting for 2
        hub.feed_block(0, rng.integers(-900, 900, (2, 1024, 2)).astype(np.int16))
    assert eng.runs >= 1 and hub.stalled[2] >= 1 and not eng.batches[0][2].any()
    v[:, :512] = 7000
    assert hub.commit(2, 1, 512) is False and hub.dropped[2] == 512
    pos = hub.backlog(2)
    v2 = hub.reserve(2, 1)                                                # the block reserves again: now it is filed where the next run reads
    v2[:, :512] = 7000
    assert hub.commit(2, 1, 512) is True and hub.backlog(2) == pos + 512
    runs = eng.runs
    while eng.runs == runs:
        hub.feed_block(0, rng.integers(-900, 900, (3, 256, 2)).astype(np.int16))
    stream = np.concatenate([b[2] for b in eng.batches])                  # channel 2 as the engine saw it, run after run
    at = np.flatnonzero((stream == 7000).all(axis=1))
    assert len(at) == 512 and at[-1] - at[0] == 511                      # the second block, once, in one piece; the first nowhere


# ---------------------------------------------------------------------------------------------------------------
# round 6: a lazy_out hub whose submit is refused (the selection deque ran ahead of the engine's slots)
# ---------------------------------------------------------------------------------------------------------------
class LazyFeedDouble:
    """The pipelined feed of SsdrEngine as IQHub uses it with lazy_out (compact rows in the order of the selection in force at
    submit), no arithmetic: row r of batch b carries (b, channel) in its first two cells, so a result handed to the wrong
    listener or paired with the wrong batch's selection is visible.  `refuse` makes the next submit raise, as
    ssdr_feed_submit does with SSDR_ESTATE when the selection has more rows than SSDR_FEED_LAZY_MAX."""

    def __init__(self, n_ch):
        from collections import deque
        self.n_ch, self.sel, self.inflight, self.submitted, self.refuse = n_ch, None, deque(), 0, 0
        self.feed_n_avg, self.feed_flags, self.sel_log = 1, None, []

    def set_kiwi_rate(self, rate):
        pass

    def set_averaging(self, n):
        pass

    def playbuffer_frame_len(self):
        return 2048

    def set_post_channels(self, channels=None):
        self.sel = None if channels is None else list(channels)
        self.sel_log.append(self.sel)

    def feed_open(self, n_frames, depth=3, post=False, lazy_out=False):
        assert lazy_out and not post
        self.frames, self.depth = n_frames, depth

    def feed_submit_from(self, batch):
        if self.refuse:
            self.refuse -= 1
            from supersdr_amd import SsdrError
            raise SsdrError(-5, "ssdr_feed_submit_from")
        assert len(self.inflight) < self.depth
        self.submitted += 1
        self.inflight.append((self.submitted, list(range(self.n_ch)) if self.sel is None else list(self.sel)))

    def feed_collect(self):
        b, sel = self.inflight.popleft()
        wf = np.zeros((self.frames // 2, len(sel), 1024), np.int16)
        pcm = np.zeros((len(sel), self.frames * 512), np.int16)
        for r, c in enumerate(sel):
            wf[:, r, 0], wf[:, r, 1], pcm[r, 0], pcm[r, 1] = b, c, b, c
        self.feed_flags = np.zeros((len(sel), self.frames), np.uint8)
        return wf, pcm, np.zeros((len(sel), self.frames), np.float32)

    def close(self):
        pass


def test_a_refused_submit_leaves_the_lazy_hub_in_step():
    from supersdr_amd import SsdrError
    from supersdr_amd.workers import IQHub
    n = 16
    eng = LazyFeedDouble(n)
    hub = IQHub(n, engine=eng, pipeline=True, depth=3, lazy=True, lazy_out=True, gpu_post=False)
    q3, q7 = hub.attach(3, wf=True, snd=True), hub.attach(7, wf=True)
    sf = np.zeros((n, 1024, 2), np.int16)
    for _ in range(4):
        hub.feed_block(0, sf)
    eng.refuse = 2                                   # two submits in a row are refused: the feeding call raises, nothing is lost
    for _ in range(2):
        with pytest.raises(SsdrError):
            hub.feed_block(0, sf)
    assert len(hub._inflight_sel) == hub._inflight == len(eng.inflight)
    q9 = hub.attach(9, wf=True, snd=True)            # the selection changes while older batches are in flight
    for _ in range(5):
        hub.feed_block(0, sf)
    hub.flush()
    assert hub._inflight == 0 and not hub._inflight_sel and not eng.inflight

    def drain(q):
        out = []
        while not q.empty():
            out.append(q.get_nowait())
        return out
    for c, qs in ((3, q3), (7, q7), (9, q9)):
        lines = [e[0] for e in drain(qs["wf"])]          # (line, n_avg, colours)
        assert lines and all(int(ln[1]) == c for ln in lines), "channel %d was handed another receiver's line" % c
        batches = [int(ln[0]) for ln in lines]
        assert batches == sorted(set(batches)) and batches[-1] == eng.submitted
        if qs["snd"] is not None:
            frames = drain(qs["snd"])
            assert frames and all(int(f[1]) == c for f in frames[::2]), "channel %d was handed another receiver's audio" % c
    assert eng.sel_log[-1] == [3, 7, 9]


def test_lazy_out_attach_limit_and_argument_checks_come_before_the_engine(monkeypatch):
    from supersdr_amd import _lib as L
    from supersdr_amd.workers import IQHub
    assert L.FEED_LAZY_MAX == 4096
    hdr = open(os.path.join(ROOT, "include", "ssdr.h")).read()
    assert "#define SSDR_FEED_LAZY_MAX 4096u" in hdr
    monkeypatch.setattr(L, "FEED_LAZY_MAX", 4)
    hub = IQHub(16, engine=LazyFeedDouble(16), pipeline=True, depth=2, lazy=True, lazy_out=True, gpu_post=False)
    for c in (1, 2, 3):
        hub.attach(c, wf=True)
    hub.attach(3, snd=True)                           # a second queue of an attached channel is not a new row
    hub.attach(4, snd=True)
    with pytest.raises(ValueError, match="SSDR_FEED_LAZY_MAX"):
        hub.attach(5, wf=True)
    assert 5 not in hub.wf_queue._q and len(hub._att_count) == 4
    hub.detach(3, wf=True, snd=False)                 # still listening to its audio: the row stays
    with pytest.raises(ValueError):
        hub.attach(5, wf=True)
    hub.detach(3)
    hub.attach(5, wf=True)
    # impossible argument sets are refused before an engine is built (without a GPU the engine's constructor would raise SsdrError)
    with pytest.raises(ValueError, match="lazy_out"):
        IQHub(8, lazy=True, lazy_out=True)
    with pytest.raises(ValueError, match="zoom"):
        IQHub(8, pipeline=True, zoom=2)
