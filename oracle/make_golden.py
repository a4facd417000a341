#!/usr/bin/env python3
"""make_golden.py -- generate tests/golden/*.npz by running the REAL reference.

Run in the build container only (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference is pure Python and cannot travel, so its behaviour on the hot path is
captured here as data: inputs and the outputs the reference's own functions produce.
GUI/audio modules the image lacks (pygame, sounddevice, tkinter, xmltodict) are replaced
by inert stubs -- none of them is touched by the functions exercised below.  Instances are
built with cls.__new__ (the constructors open sockets) and given only the attributes the
exercised method reads.

Fixtures (all small):
  filtering.npz    filtering(fl, fs).h for the cut-offs the path uses        utils_supersdr.py:333-344
  binning.npz      np.mean time binning of N byte-valued lines               utils_supersdr.py:881-888
  db2col.npz       spectrum_db2col in/out at zoom 0/8/14, autoscale on/off   utils_supersdr.py:787-813
  playbuffer.npz   play_buffer over 4 consecutive frames, volume/pan cases   utils_supersdr.py:1106-1148
  display.npz      plot_spectrum's trace pixels; S-meter smoothing series             utils_supersdr.py:1669-1691;
                                                                              supersdr.py:936-947
  frames.npz       W/F, SND and IQ frame bytes and their decoded arrays      utils_supersdr.py:780-785,
                   ADPCM known answer                                        1065-1074; kiwi/client.py:33-87,384-482
  wavreader.npz    decode of a synthetic Kiwi IQ wav                         kiwi/wavreader.py:74-102
"""
import io
import os
import queue
import struct
import sys
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("pygame", "pygame.font", "pygame.event", "pygame.draw", "pygame.freetype", "sounddevice",
                 "xmltodict", "requests"):
        sys.modules.setdefault(name, mock.MagicMock())
    loc = types.ModuleType("pygame.locals")
    for k in (["K_%d" % i for i in range(10)] + ["K_KP%d" % i for i in range(10)] +
              ["K_BACKSPACE", "K_RETURN", "K_ESCAPE", "K_KP_ENTER"]):
        setattr(loc, k, hash(k) & 0xFFFF)
    sys.modules["pygame.locals"] = loc
    tk = types.ModuleType("tkinter")
    tk.__all__ = []
    sys.modules["tkinter"] = tk
    sys.modules["tkinter.ttk"] = mock.MagicMock()
    sys.modules["tkinter.messagebox"] = mock.MagicMock()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)              # the reference opens its .ttf fonts relative to CWD at import time
    try:
        import utils_supersdr as U
        from kiwi import client as KC
        from kiwi import wavreader as WR
    finally:
        os.chdir(cwd)
    return U, KC, WR


def gold_filtering(U):
    out = {}
    for fl, fs in ((6000, 48000), (3000, 12000), (6000, 12000), (1485, 12000), (200, 12000), (2700, 12000),
                   (500, 12000), (6000, 12000.5), (10125, 48000)):
        f = U.filtering(fl, fs)
        out["h_%g_%g" % (fl, fs)] = f.h
        out["n_%g_%g" % (fl, fs)] = np.int64(f.n_tap)
    x = np.random.default_rng(1).standard_normal(300)
    f = U.filtering(6000, 48000)
    out["lowpass_in"] = x
    out["lowpass_out"] = f.lowpass(x)
    return out


def gold_binning(U):
    from collections import deque
    rng = np.random.default_rng(2)
    out = {}
    for n in (1, 2, 3, 7, 10, 33, 100):
        lines = rng.integers(0, 256, (n, 1024)).astype(np.uint8)
        d = deque([], n)
        for i in range(n):
            d.append(lines[i].astype(np.float32))       # what receive_spectrum leaves in self.spectrum
        out["lines_%d" % n] = lines
        out["mean_%d" % n] = np.mean(d, axis=0)         # utils_supersdr.py:886
    return out


def gold_db2col(U):
    rng = np.random.default_rng(3)
    out = {}
    i = 0
    for zoom in (0, 8, 14):
        for auto in (True, False):
            for dlo, dhi in ((0, 0), (-10, 20)):
                wf = U.kiwi_waterfall.__new__(U.kiwi_waterfall)
                wf.zoom = zoom
                wf.wf_auto_scaling = auto
                wf.delta_low_db, wf.delta_high_db = dlo, dhi
                wf.dynamic_range = wf.MIN_DYN_RANGE
                base = rng.integers(120, 150, 1024).astype(np.float32)
                base[rng.integers(0, 1024, 20)] += rng.integers(20, 90, 20)
                if i % 3 == 2:
                    base = (rng.integers(0, 2551, 1024) / 10.0).astype(np.float32)      # averaged (k/10) values
                wf.spectrum = base.copy()
                wf.spectrum_db2col()
                out["in_%d" % i] = base
                out["cfg_%d" % i] = np.array([zoom, int(auto), dlo, dhi], np.float64)
                out["color_%d" % i] = np.asarray(wf.wf_color)
                out["scal_%d" % i] = np.array([wf.low_clip_db, wf.high_clip_db, wf.dynamic_range,
                                               wf.wf_min_db, wf.wf_max_db], np.float64)
                i += 1
    out["count"] = np.int64(i)
    return out


def gold_playbuffer(U):
    rng = np.random.default_rng(4)
    out = {}
    case = 0
    for volume, balance in ((100, 0.0), (150, 0.0), (70, -0.5), (100, 1.0), (100, -1.0)):
        snd = U.kiwi_sound.__new__(U.kiwi_sound)
        snd.audio_buffer = queue.Queue()
        snd.volume = volume
        snd.late_flag = False
        snd.SAMPLE_RATIO = 4
        snd.kiwi_filter = U.filtering(12000 / 2, 48000)
        snd.n_tap = snd.kiwi_filter.n_tap
        snd.lowpass = snd.kiwi_filter.lowpass
        snd.old_buffer = np.zeros((snd.n_tap - 1))
        snd.audio_balance = balance
        snd.rssi = -80
        snd.mute_counter = 0
        snd.max_rssi_before_mute = -20
        snd.muting_delay = 15
        # the recording branch (:1139-1140) runs too: what audio_rec.audio_buffer collects is the mono block before the pan
        snd.audio_rec = types.SimpleNamespace(recording_flag=True, audio_buffer=[])
        frames = (rng.standard_normal((4, 512)) * 9000).clip(-32768, 32767).astype(np.int16)
        frames[1, 100:110] = 32767                        # drives the truncating cast into wrap at volume 150
        outs = []
        for f in range(4):
            snd.audio_buffer.put(frames[f])
            o = np.zeros((2048, 2), np.int16)
            with np.errstate(invalid="ignore"):
                snd.play_buffer(o, 2048, None, None)
            outs.append(o.copy())
        out["in_%d" % case] = frames
        out["cfg_%d" % case] = np.array([volume, balance], np.float64)
        out["out_%d" % case] = np.stack(outs)
        out["rec_%d" % case] = np.stack(snd.audio_rec.audio_buffer)
        case += 1
    out["count"] = np.int64(case)
    out["n_tap"] = np.int64(U.filtering(6000, 48000).n_tap)

    # 20.25 kHz KiwiSDRs: SAMPLE_RATIO = 48000/20250 is fractional -> resample_poly(popped, 64, 27, padtype="line")[:-1]
    # (utils_supersdr.py:994, 1000-1001, 1125-1126); blocksize int(512 * SAMPLE_RATIO) = 1213 (:1211)
    case = 0
    for volume, balance in ((100, 0.0), (150, 0.3), (60, -1.0)):
        snd = U.kiwi_sound.__new__(U.kiwi_sound)
        snd.audio_buffer = queue.Queue()
        snd.volume = volume
        snd.late_flag = False
        snd.KIWI_RATE = 20250
        snd.SAMPLE_RATIO = snd.AUDIO_RATE / snd.KIWI_RATE
        gcd = np.gcd(snd.KIWI_RATE, snd.AUDIO_RATE)
        snd.n_low, snd.n_high = int(snd.KIWI_RATE / gcd), int(snd.AUDIO_RATE / gcd)
        snd.audio_balance = balance
        snd.rssi = -80
        snd.mute_counter = 0
        snd.max_rssi_before_mute = -20
        snd.muting_delay = 15
        snd.audio_rec = types.SimpleNamespace(recording_flag=True, audio_buffer=[])
        frames = (rng.standard_normal((3, 512)) * 9000).clip(-32768, 32767).astype(np.int16)
        frames[1, 200:210] = 32767
        frames[2, 0], frames[2, -1] = -30000, 30000      # a steep line through the end points
        n_out = int(512 * snd.SAMPLE_RATIO)
        outs = []
        for f in range(3):
            snd.audio_buffer.put(frames[f])
            o = np.zeros((n_out, 2), np.int16)
            with np.errstate(invalid="ignore"):
                snd.play_buffer(o, n_out, None, None)
            outs.append(o.copy())
        out["rs_in_%d" % case] = frames
        out["rs_cfg_%d" % case] = np.array([volume, balance], np.float64)
        out["rs_out_%d" % case] = np.stack(outs)
        out["rs_rec_%d" % case] = np.stack(snd.audio_rec.audio_buffer)
        case += 1
    out["rs_count"] = np.int64(case)
    out["rs_ratio"] = np.array([64, 27], np.int64)
    return out


def gold_display(U):
    """display reductions: the real display_stuff.plot_spectrum (utils_supersdr.py:1669-1691) drawn into a recording
    stand-in for pygame's PixelArray, and the S-meter lines of the main loop (supersdr.py:936-947), which are script
    text rather than a function: they are sliced out of the file between their first and last statement and executed
    on given inputs."""
    rng = np.random.default_rng(7)
    out = {}
    pygame = sys.modules["pygame"]
    case = 0
    for height, spec_h, n_lines, t_avg in ((40, 180, 30, 15), (40, 180, 6, 15), (24, 137, 20, 7)):
        wf = U.kiwi_waterfall.__new__(U.kiwi_waterfall)
        wf.wf_data = np.zeros((height, 1024))
        wf.wf_auto_scaling = True
        lines = []
        from collections import deque
        tmp, run_index = deque([], wf.wf_buffer_len), 0
        for i in range(n_lines):
            col = np.clip(rng.normal(90, 60, 1024), 0, 254).astype(np.float32)
            col[rng.integers(0, 1024, 5)] = np.float32(1e-7) * np.float32(rng.integers(1, 9))
            lines.append(col)
            run_index += 1
            tmp.appendleft(col)                                  # utils_supersdr.py:893-897
            if len(tmp) > 0 and run_index > wf.wf_buffer_len:
                wf.wf_data[1:, :] = wf.wf_data[0:-1, :]
                wf.wf_data[0, :] = tmp.pop()
        disp = U.display_stuff.__new__(U.display_stuff)
        disp.SPECTRUM_HEIGHT, disp.DISPLAY_WIDTH, disp.SPECTRUM_Y = spec_h, 1024, 0
        pix = mock.MagicMock()
        pygame.PixelArray = mock.MagicMock(return_value=pix)
        disp.plot_spectrum(mock.MagicMock(), wf, t_avg=t_avg)
        ys = np.full(1024, -1, np.int64)
        for call in pix.__setitem__.call_args_list:
            (x, y), _ = call.args
            ys[x] = y
        assert (ys >= 0).all()
        out["lines_%d" % case] = np.stack(lines)
        out["cfg_%d" % case] = np.array([height, spec_h, n_lines, t_avg], np.int64)
        out["y_%d" % case] = ys
        out["trace_%d" % case] = np.nanmean(wf.wf_data.T[:, :t_avg], axis=1)
        case += 1
    out["count"] = np.int64(case)

    src = open(os.path.join(REF, "supersdr.py")).read().split("\n")
    first = next(i for i, l in enumerate(src) if l.strip() == "rssi_last = rssi_hist[-1]")
    last = next(i for i, l in enumerate(src) if l.strip() == "rssi_smooth_slow = max(rssi_hist)")
    code = compile("\n".join(l[4:] if l.startswith("    ") else l for l in src[first:last + 1]), "supersdr.py:smeter", "exec")
    import math
    case = 0
    for decay, fps in ((4000, 30), (1000, 30), (400, 25), (8000, 60)):
        rssi_in = np.concatenate([np.full(30, -110.0), np.full(40, -53.5), rng.normal(-80, 12, 120), np.full(60, -127.0)])
        ns = dict(math=math, FPS=fps, kiwi_snd=types.SimpleNamespace(decay=decay),
                  rssi_hist=deque(10 * [float(rssi_in[0])], 10), rssi_smooth=float(rssi_in[0]),
                  rssi_smooth_slow=float(rssi_in[0]))
        sm, sl = [], []
        for run_index, r in enumerate(rssi_in):
            ns["rssi_hist"].append(float(r))                     # supersdr.py:190-191
            ns["run_index"] = run_index
            exec(code, ns)
            sm.append(ns["rssi_smooth"])
            sl.append(ns["rssi_smooth_slow"])
        out["sm_in_%d" % case] = rssi_in
        out["sm_cfg_%d" % case] = np.array([decay, fps], np.float64)
        out["sm_smooth_%d" % case] = np.array(sm)
        out["sm_slow_%d" % case] = np.array(sl)
        case += 1
    out["sm_count"] = np.int64(case)
    return out


def gold_frames(U, KC):
    rng = np.random.default_rng(5)
    out = {}
    # --- W/F frame: utils_supersdr.kiwi_waterfall.receive_spectrum
    bins = rng.integers(0, 256, 1024).astype(np.uint8)
    msg = bytearray(b"W/F" + b"\x00" + struct.pack("<III", 1234, 7, 42) + bins.tobytes())
    wf = U.kiwi_waterfall.__new__(U.kiwi_waterfall)
    wf.wf_stream = types.SimpleNamespace(receive_message=lambda: msg)
    wf.keepalive = lambda: None
    wf.receive_spectrum()
    out["wf_msg"] = np.frombuffer(bytes(msg), np.uint8)
    out["wf_spectrum"] = wf.spectrum
    # --- SND frame: utils_supersdr.kiwi_sound.process_audio_stream
    pcm = rng.integers(-32768, 32768, 512).astype(np.int16)
    smsg = bytearray(b"SND" + struct.pack("<BI", 2, 77) + struct.pack(">H", 1270 - 733) + pcm.astype(">i2").tobytes())
    snd = U.kiwi_sound.__new__(U.kiwi_sound)
    snd.stream = types.SimpleNamespace(receive_message=lambda: smsg)
    snd.run_index, snd.delta_t = 0, 0.0
    snd.KIWI_SAMPLES_PER_FRAME, snd.KIWI_RATE = 512, 12000
    samples = snd.process_audio_stream()
    out["snd_msg"] = np.frombuffer(bytes(smsg), np.uint8)
    out["snd_samples"] = samples
    out["snd_rssi"] = np.float64(snd.rssi)
    out["snd_adc_ovf"] = np.int64(snd.adc_overflow_flag)
    # --- IQ frame: kiwi.client.KiwiSDRStream._process_aud (IQ branch) -> _process_iq_samples
    iq = rng.integers(-32768, 32768, (512, 2)).astype(np.int16)
    body = bytearray(struct.pack("<BI", 0, 99) + struct.pack(">H", 870) + struct.pack("<BBII", 5, 0, 1234567, 891011) +
                     iq.astype(">i2").tobytes())
    got = {}

    class Rec(KC.KiwiSDRStream):
        def __init__(self):
            self._options = types.SimpleNamespace(ADC_OV=False, S_meter=-1, sdt=0, sound=True, raw=False, tstamp=False,
                                                  stats=False)
            self._s_meter_valid = False
            self._modulation = "iq"
            self._compression = False

        def _process_iq_samples(self, seq, samples, rssi, gps):
            got.update(seq=seq, samples=samples.copy(), rssi=rssi, gps=gps)

    Rec()._process_aud(body)
    out["iq_body"] = np.frombuffer(bytes(body), np.uint8)
    out["iq_int16"] = iq
    out["iq_complex64"] = got["samples"]
    out["iq_rssi"] = np.float64(got["rssi"])
    out["iq_seq"] = np.int64(got["seq"])
    out["iq_gps"] = np.array([got["gps"]["last_gps_solution"], got["gps"]["dummy"], got["gps"]["gpssec"],
                              got["gps"]["gpsnsec"]], np.int64)
    # --- IMA ADPCM decoder known answer (kiwi/client.py:58-87): state persists across two calls
    dec = KC.ImaAdpcmDecoder()
    data = rng.integers(0, 256, 600).astype(np.uint8)
    a = np.array(dec.decode(bytearray(data[:256].tobytes())), np.int16)
    b = np.array(dec.decode(bytearray(data[256:].tobytes())), np.int16)
    out["adpcm_in"] = data
    out["adpcm_out"] = np.concatenate([a, b])
    # --- W/F frame, compressed (kiwi/client.py:470-482): decoder reset per line, 10-sample tail dropped
    got_wf = {}

    class RecWf(KC.KiwiSDRStream):
        def __init__(self):
            self._options = types.SimpleNamespace(raw=False)
            self._compression = True
            self._decoder = KC.ImaAdpcmDecoder()

        def _process_waterfall_samples(self, seq, samples):
            got_wf.update(seq=seq, samples=np.array(samples, np.int16))

    cbody = bytearray(struct.pack("<III", 1, 2, 3) + data[:517].tobytes())
    RecWf()._process_wf(cbody)
    out["wfc_body"] = np.frombuffer(bytes(cbody), np.uint8)
    out["wfc_samples"] = got_wf["samples"]
    return out


def gold_wavreader(WR):
    """A synthetic Kiwi IQ wav: RIFF/WAVE, fmt (PCM, 2 ch), then [kiwi chunk][data chunk] x 5
    (512 samples per block, GNSS stamps 42.667 ms apart; the reader drops the first two blocks)."""
    rng = np.random.default_rng(6)
    blocks = [rng.integers(-32768, 32768, (512, 2)).astype("<i2") for _ in range(5)]
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 12000, 48000, 4, 16)
    for i, blk in enumerate(blocks):
        body += b"kiwi" + struct.pack("<IBBII", 10, 3, 0, 1000, 42666667 * i)
        body += b"data" + struct.pack("<I", blk.nbytes) + blk.tobytes()
    wav = b"RIFF" + struct.pack("<I", len(body)) + body
    path = "/tmp/ssdr_golden_kiwi.wav"
    with open(path, "wb") as f:
        f.write(wav)
    out = {"wav_bytes": np.frombuffer(wav, np.uint8), "blocks": np.stack(blocks).astype(np.int16)}
    try:
        t, z = WR.read_kiwi_iq_wav(path)
        out["t"] = np.asarray(t, np.float64)
        out["z"] = np.asarray(z, np.complex64)
        out["ok"] = np.int64(1)
    except Exception as e:          # keep the failure visible in the fixture rather than hiding it
        out["ok"] = np.int64(0)
        out["error"] = np.frombuffer(repr(e).encode(), np.uint8)
    os.remove(path)
    return out


def main():
    U, KC, WR = import_reference()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "filtering.npz"), **gold_filtering(U))
    np.savez_compressed(os.path.join(OUT, "binning.npz"), **gold_binning(U))
    np.savez_compressed(os.path.join(OUT, "db2col.npz"), **gold_db2col(U))
    np.savez_compressed(os.path.join(OUT, "playbuffer.npz"), **gold_playbuffer(U))
    np.savez_compressed(os.path.join(OUT, "display.npz"), **gold_display(U))
    np.savez_compressed(os.path.join(OUT, "frames.npz"), **gold_frames(U, KC))
    np.savez_compressed(os.path.join(OUT, "wavreader.npz"), **gold_wavreader(WR))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
