"""GPU-backed stand-ins for the reference's two receiver workers.

`kiwi_waterfall` and `kiwi_sound` keep the class names, constructor signatures, class
constants, methods and attributes that supersdr.py touches (usage
census of supersdr.py: 37 kiwi_wf.* and 32 kiwi_snd.* names), so the UI runs unmodified
when `utils_supersdr.kiwi_waterfall / kiwi_sound` are replaced by these.  What changes is
the producer behind the two seams:

    kiwi_waterfall.receive_spectrum()      utils_supersdr.py:780-785
        reference: one W/F websocket frame from the KiwiSDR server -> float32[1024] bytes
        here:      the next waterfall line of this channel from the GPU (ssdr_run_wf)
    kiwi_sound.process_audio_stream()      utils_supersdr.py:1044-1076
        reference: one SND websocket frame -> int16[512] + rssi
        here:      the next PCM frame of this channel from the GPU (ssdr_run_audio)

Both are fed by an `IQHub`, which owns one SsdrEngine (one GPU context) for a block of
receiver channels, batches the channels' IQ frames and runs the two kernels once per
superframe (1024 samples = 1 waterfall line + 2 audio frames).

What the reference does on the host AFTER those seams and is pure control flow stays here, with
citations (time binning by division of the GPU's integer sums, scrolling, pacing, TX mute).  The two
arithmetic steps -- spectrum_db2col (utils_supersdr.py:787-813) and the play_buffer interpolator
(:1106-1148, both the x4 and the 64/27 resample_poly branch) -- are HIP kernels (ssdr_run_db2col /
ssdr_run_playbuffer, bit-exact against golden vectors of the real reference): the hub runs them with every
superframe and the workers hand out their results.  There is no host implementation of either step:
`IQHub(gpu_post=False)` skips the two kernels for consumers that only want raw lines and PCM, and
spectrum_db2col() then raises and play_buffer() (a PortAudio callback, which must not raise) plays silence, stops
the worker and leaves the error in `kiwi_sound.error`.
"""
import logging
import queue
import threading
import time
from collections import deque

import numpy as np

from . import _lib as L
from ._lib import Db2colChan, PlayChan
from .engine import SsdrEngine, default_params

# module constants of the reference (utils_supersdr.py:42-50)
CW_PITCH = 0.6
LOW_CUT_SSB, HIGH_CUT_SSB = 30, 3000
LOW_CUT_CW, HIGH_CUT_CW = int(CW_PITCH * 1000 - 200), int(CW_PITCH * 1000 + 200)
HIGHLOW_CUT_AM = 6000


class IQHub:
    """Batches per-channel IQ into superframes and runs the GPU path for all channels at once.

    feed(channel, iq_int16[n,2]) appends samples of one channel (any n); whenever every
    channel has >= 1024 samples buffered, one superframe is pushed (ssdr_push_iq), both
    kernels run, and the results land in per-channel queues:
        wf_queue[c]  : int16[1024] sums of `averaging` byte lines (+ the N used)
        snd_queue[c] : (int16[512] pcm, float rssi) per audio frame
    """

    def __init__(self, n_channels, device=0, engine=None, max_queue=64, gpu_post=True, kiwi_rate=12000, trace_rows=0):
        self.n_ch = int(n_channels)
        self.engine = engine if engine is not None else SsdrEngine(self.n_ch, device)
        # spectrum_db2col and play_buffer run on the GPU with every superframe
        self.gpu_post = bool(gpu_post)
        self.kiwi_rate = int(kiwi_rate)              # kiwi_sound.KIWI_RATE: selects play_buffer's branch (:1125)
        self.play_len = 2048
        if self.gpu_post:
            self.engine.set_kiwi_rate(self.kiwi_rate)
            self.play_len = self.engine.playbuffer_frame_len()
        # display reductions: wf_data's newest rows stay on the device, fed by every db2col run
        self.trace_rows = int(trace_rows) if self.gpu_post else 0
        if self.trace_rows:
            self.engine.set_wfdata_rows(self.trace_rows)
        self._smeter = None
        self.wf_clients = [None] * self.n_ch        # kiwi_waterfall objects: display state for db2col
        self.snd_clients = [None] * self.n_ch       # kiwi_sound objects: volume / balance for play_buffer
        self._buf = [np.zeros((0, 2), np.int16) for _ in range(self.n_ch)]
        self.wf_queue = [queue.Queue(max_queue) for _ in range(self.n_ch)]
        self.snd_queue = [queue.Queue(2 * max_queue) for _ in range(self.n_ch)]
        self._params = [default_params("am") for _ in range(self.n_ch)]
        self.averaging_n = 1
        self._lock = threading.Lock()
        self.superframes = 0

    # ---- control plane (forwarded SET commands)
    def params(self, channel):
        return self._params[channel]

    def set_params(self, channel, p):
        with self._lock:
            self._params[channel] = p
            self.engine.set_params(channel, [p])

    def set_averaging(self, n):
        with self._lock:
            n = int(min(max(n, 1), 100))
            if n != self.averaging_n:
                self.averaging_n = n
                self.engine.set_averaging(n)

    # ---- data plane
    def feed(self, channel, iq):
        iq = np.asarray(iq, np.int16).reshape(-1, 2)
        with self._lock:
            self._buf[channel] = np.concatenate([self._buf[channel], iq])
            self._pump()

    def _pump(self):
        while all(len(b) >= L.NFFT for b in self._buf):
            batch = np.stack([b[: L.NFFT] for b in self._buf])
            self._buf = [b[L.NFFT:] for b in self._buf]
            self.engine.push_iq(batch)
            n_avg = self.averaging_n
            wf = self.engine.run_wf()                 # [lines, n_ch, 1024]
            color = chans = None
            if self.gpu_post and len(wf) and any(w is not None for w in self.wf_clients):
                chans = [self._db2col_chan(w) for w in self.wf_clients]
                color = self.engine.run_db2col(chans, len(wf))          # [lines, n_ch, 1024] float32 0..254
            pcm, rssi = self.engine.run_audio()       # [n_ch, 1024], [n_ch, 2]
            play = None
            if self.gpu_post and any(s is not None for s in self.snd_clients):
                play = self.engine.run_playbuffer([PlayChan(float(s.volume), float(s.audio_balance)) if s is not None
                                                   else PlayChan(100.0, 0.0) for s in self.snd_clients])
            self.superframes += 1
            for c in range(self.n_ch):
                for i, line in enumerate(wf):
                    post = None
                    if color is not None and self.wf_clients[c] is not None:
                        k = chans[c]
                        post = (color[i, c].copy(), k.low_clip_db, k.high_clip_db, k.dynamic_range, k.wf_min_db, k.wf_max_db)
                    _put_drop_oldest(self.wf_queue[c], (line[c].copy(), n_avg, post))
                for f in range(2):
                    blk = play[c, f * self.play_len:(f + 1) * self.play_len].copy() if play is not None else None
                    _put_drop_oldest(self.snd_queue[c], (pcm[c, f * L.FRAME:(f + 1) * L.FRAME].copy(), float(rssi[c, f]), blk))

    def spectrum_trace(self, t_avg=15, spectrum_height=0):
        """display_stuff.plot_spectrum's reduction for all channels (utils_supersdr.py:1678-1679): (float64 [n_ch, 1024]
        np.nanmean over the t_avg newest wf_data rows, int32 [n_ch, 1024] pixel rows).  The device copy of wf_data
        advances with the lines the hub produces (all channels in step), not with each worker's consumption."""
        with self._lock:
            return self.engine.run_trace(t_avg, spectrum_height)

    def smeter_step(self, fps, decay_ms=None):
        """One display frame of the main loop's S-meter smoothing (supersdr.py:936-947) for all channels, from the last
        audio frame's RSSI.  Returns (rssi_smooth [n_ch], rssi_smooth_slow [n_ch])."""
        from ._lib import SmeterChan
        with self._lock:
            if self._smeter is None:
                self._smeter = [SmeterChan.start(-127.0) for _ in range(self.n_ch)]      # kiwi_sound.rssi before any frame
            for c, s in enumerate(self.snd_clients):
                self._smeter[c].decay_ms = float(decay_ms if decay_ms is not None else (s.decay if s is not None else 4000))
            self.engine.run_smeter(self._smeter, fps)
            return (np.array([s.rssi_smooth for s in self._smeter]), np.array([s.rssi_smooth_slow for s in self._smeter]))

    @staticmethod
    def _db2col_chan(w):
        if w is None:
            return Db2colChan(auto_scale=1, low_clip_db=-120.0, high_clip_db=-60.0, dynamic_range=40.0)
        return Db2colChan(zoom=int(w.zoom), auto_scale=int(bool(w.wf_auto_scaling)), delta_low_db=int(w.delta_low_db),
                          delta_high_db=int(w.delta_high_db), low_clip_db=float(w.low_clip_db),
                          high_clip_db=float(w.high_clip_db), dynamic_range=float(w.dynamic_range))

    def close(self):
        self.engine.close()


def _put_drop_oldest(q, item):
    try:
        q.put_nowait(item)
    except queue.Full:
        try:
            q.get_nowait()
        except queue.Empty:
            pass
        q.put_nowait(item)


class kiwi_waterfall:
    """kiwi_waterfall (utils_supersdr.py:592-898) with the W/F websocket replaced by the GPU."""
    MAX_FREQ = 30000
    CENTER_FREQ = int(MAX_FREQ / 2)
    MAX_ZOOM = 14
    WF_BINS = 1024
    MAX_FPS = 23
    MIN_DYN_RANGE = 40.
    CLIP_LOWP, CLIP_HIGHP = 40., 100
    delta_low_db, delta_high_db = 0, 0
    low_clip_db, high_clip_db = -120, -60
    wf_min_db, wf_max_db = low_clip_db, low_clip_db + MIN_DYN_RANGE
    kiwi_wf_timestamp = None
    wf_buffer_len = 3

    def __init__(self, host_, port_, pass_, zoom_, freq_, eibi, disp, hub=None, channel=0, timeout=5.0):
        # attribute set-up mirrors utils_supersdr.py:606-645, 692-695
        self.eibi = eibi
        self.host, self.port, self.password = host_, port_, pass_
        self.zoom = zoom_
        self.freq = freq_
        self.averaging_n = 1
        self.wf_auto_scaling = True
        self.BINS2PIXEL_RATIO = disp.DISPLAY_WIDTH / self.WF_BINS
        self.old_averaging_n = self.averaging_n
        self.dynamic_range = self.MIN_DYN_RANGE
        self.wf_white_flag = False
        self.terminate = False
        self.run_index = 0
        if not self.freq:
            self.freq = 14200
        self.tune = self.freq
        self.radio_mode = "USB"
        self.span_khz = self.zoom_to_span()
        self.start_f_khz = self.start_freq()
        self.end_f_khz = self.end_freq()
        self.div_list, self.subdiv_list = [], []
        self.min_bin_spacing = 100
        self.space_khz = 10
        self.counter, self.actual_freq = self.start_frequency_to_counter(self.start_f_khz)
        self.wf_color = None
        self.freq_offset = 0
        self.kiwi_wf_timestamp = int(time.time())
        self.bins_per_khz = self.WF_BINS / self.span_khz
        self.wf_data = np.zeros((disp.WF_HEIGHT, self.WF_BINS))
        self.wf_data_tmp = deque([], self.wf_buffer_len)
        self.avg_spectrum_deque = deque([], self.averaging_n)
        self.spectrum = np.zeros(self.WF_BINS, np.float32)
        # the GPU side
        if hub is None:
            raise ValueError("the GPU-backed kiwi_waterfall needs an IQHub (there is no server-side FFT to fall back to)")
        self.hub, self.channel, self._timeout = hub, channel, timeout
        self._gpu_post = None
        if hasattr(hub, "wf_clients"):
            hub.wf_clients[channel] = self

    # ---- frequency / zoom arithmetic: utils_supersdr.py:747-778 (scalar UI math)
    def zoom_to_span(self):
        assert 0 <= self.zoom <= self.MAX_ZOOM
        self.span_khz = self.MAX_FREQ / 2 ** self.zoom
        return self.span_khz

    def start_frequency_to_counter(self, start_frequency_):
        assert 0 <= start_frequency_ <= self.MAX_FREQ
        self.counter = round(start_frequency_ / self.MAX_FREQ * 2 ** self.MAX_ZOOM * self.WF_BINS)
        return self.counter, self.counter * self.MAX_FREQ / self.WF_BINS / 2 ** self.MAX_ZOOM

    def start_freq(self):
        self.start_f_khz = self.freq - self.span_khz / 2
        return self.start_f_khz

    def end_freq(self):
        self.end_f_khz = self.freq + self.span_khz / 2
        return self.end_f_khz

    def offset_to_bin(self, offset_khz_):
        return self.WF_BINS / self.span_khz * offset_khz_

    def bins_to_khz(self, bins_):
        return bins_ / (self.WF_BINS / self.span_khz) + self.start_f_khz

    def deltabins_to_khz(self, bins_):
        return bins_ / (self.WF_BINS / self.span_khz)

    def gen_div(self):                                   # utils_supersdr.py:697-717
        self.space_khz = 10
        self.div_list, self.subdiv_list = [], []
        f_s, f_e = int(self.start_f_khz), int(self.end_f_khz)
        while self.div_list == [] and self.subdiv_list == []:
            if self.bins_per_khz * self.space_khz > self.min_bin_spacing:
                self.div_list = [int(self.offset_to_bin(f - self.start_f_khz)) for f in range(f_s, f_e + 1)
                                 if not f % self.space_khz]
            if self.bins_per_khz * self.space_khz / 10 > self.min_bin_spacing / 10:
                self.subdiv_list = [int(self.offset_to_bin(f - self.start_f_khz)) for f in range(f_s, f_e + 1)
                                    if not f % (self.space_khz / 10)]
            self.space_khz *= 10

    def set_freq_zoom(self, freq_, zoom_):               # utils_supersdr.py:815-845
        self.freq, self.zoom = freq_, zoom_
        self.zoom_to_span()
        self.start_freq()
        self.end_freq()
        if zoom_ == 0:
            self.freq = self.CENTER_FREQ
            self.start_freq()
            self.end_freq()
            self.span_khz = self.MAX_FREQ
        elif self.start_f_khz < 0:
            self.freq = self.zoom_to_span() / 2
            self.start_freq()
            self.end_freq()
        elif self.end_f_khz > self.MAX_FREQ:
            self.freq = self.MAX_FREQ - self.zoom_to_span() / 2
            self.start_freq()
            self.end_freq()
        self.counter, _ = self.start_frequency_to_counter(self.start_f_khz)
        if self.eibi is not None:
            self.eibi.get_stations(self.start_f_khz, self.end_f_khz)
        self.bins_per_khz = self.WF_BINS / self.span_khz
        self.gen_div()
        return self.freq

    def change_passband(self, delta_low_, delta_high_):  # utils_supersdr.py:859-873
        if self.radio_mode == "USB":
            lc_, hc_ = LOW_CUT_SSB + delta_low_, HIGH_CUT_SSB + delta_high_
        elif self.radio_mode == "LSB":
            lc_, hc_ = -HIGH_CUT_SSB - delta_high_, -LOW_CUT_SSB - delta_low_
        elif self.radio_mode == "AM":
            lc_, hc_ = -HIGHLOW_CUT_AM - delta_low_, HIGHLOW_CUT_AM + delta_high_
        else:
            lc_, hc_ = LOW_CUT_CW + delta_low_, HIGH_CUT_CW + delta_high_
        self.lc, self.hc = lc_, hc_
        return lc_, hc_

    def keepalive(self):
        pass                                             # no server to keep alive

    def close_connection(self):
        self.terminate = True

    # ---- the seam: utils_supersdr.py:780-785
    def receive_spectrum(self):
        """Leaves self.spectrum = float32[WF_BINS] in byte units (dBm = byte - 255)."""
        self.hub.set_averaging(1)                        # binning is done by run() exactly like the reference
        try:
            line, n, self._gpu_post = self.hub.wf_queue[self.channel].get(timeout=self._timeout)
        except queue.Empty:
            self.terminate = True
            return
        self.spectrum = line.astype(np.float32) / np.float32(n)

    def receive_binned_spectrum(self, n):
        """GPU time binning: one averaged line per N input lines; float32(sum)/float32(N) is
        bit-identical to the reference's np.mean over a deque of N lines (utils:881-886)."""
        self.hub.set_averaging(n)
        while not self.terminate:
            try:
                line, n_used, self._gpu_post = self.hub.wf_queue[self.channel].get(timeout=self._timeout)
            except queue.Empty:
                self.terminate = True
                return
            if n_used == n:                              # lines binned with a previous N are stale
                self.spectrum = line.astype(np.float32) / np.float32(n)
                return

    def spectrum_db2col(self):                           # utils_supersdr.py:787-813
        if self._gpu_post is not None:                   # computed by ssdr_run_db2col with this object's display state
            (self.wf_color, self.low_clip_db, self.high_clip_db, self.dynamic_range,
             self.wf_min_db, self.wf_max_db) = self._gpu_post
            self._gpu_post = None
            return
        raise RuntimeError("spectrum_db2col runs on the GPU (ssdr_run_db2col): no result came with this line -- "
                           "the hub was built with gpu_post=False or the line was already converted")

    def set_white_flag(self):                            # utils_supersdr.py:875-877
        self.wf_color = np.ones_like(self.wf_color) * 255
        self.wf_data[0, :] = self.wf_color

    def step(self):
        """One iteration of run() (utils_supersdr.py:879-897)."""
        if self.averaging_n > 1:
            self.receive_binned_spectrum(self.averaging_n)
        else:
            self.receive_spectrum()
        if self.terminate:
            return
        self.run_index += 1
        self.spectrum_db2col()
        self.wf_data_tmp.appendleft(self.wf_color)
        if len(self.wf_data_tmp) > 0 and self.run_index > self.wf_buffer_len:
            self.wf_data[1:, :] = self.wf_data[0:-1, :]
            self.wf_data[0, :] = self.wf_data_tmp.pop()

    def run(self):
        while not self.terminate:
            self.step()


class _NoRecording:
    recording_flag = False

    def start(self):
        pass

    def stop(self):
        pass


class kiwi_sound:
    """kiwi_sound (utils_supersdr.py:901-1186) with the SND websocket replaced by the GPU."""
    FORMAT = np.int16
    CHANNELS = 2
    AUDIO_RATE = 48000
    KIWI_RATE = 12000
    SAMPLE_RATIO = int(AUDIO_RATE / KIWI_RATE)
    CHUNKS = 1
    KIWI_SAMPLES_PER_FRAME = 512

    def __init__(self, freq_, mode_, lc_, hc_, password_, kiwi_wf, buffer_len, volume_=100, host_=None, port_=None,
                 subrx_=False, hub=None, channel=None, timeout=5.0):
        self.subrx = subrx_
        self.kiwi_wf = kiwi_wf
        self.host = host_ if host_ else kiwi_wf.host
        self.port = port_ if port_ else kiwi_wf.port
        self.FULL_BUFF_LEN = max(1, buffer_len)
        self.audio_buffer = queue.Queue(maxsize=self.FULL_BUFF_LEN)
        self.terminate = False
        self.volume = volume_
        self.max_rssi_before_mute = -20
        self.mute_counter = 0
        self.muting_delay = 15
        self.adc_overflow_flag = False
        self.status = None
        self.run_index = 0
        self.delta_t = 0.0
        self.rssi = -127
        self.freq = freq_
        self.radio_mode = mode_
        self.lc, self.hc = lc_, hc_
        # AGC parameter holders: utils_supersdr.py:936-945
        self.on, self.hang, self.thresh, self.slope = True, False, -80, 0
        self.decay_other, self.decay_cw, self.gain = 4000, 1000, 50
        self.min_agc_delay, self.max_agc_delay = 400, 8000
        self.decay = self.decay_other
        self.audio_balance = 0.0
        self.freq_offset = 0
        self.KIWI_RATE_TRUE = float(self.KIWI_RATE)
        self.late_flag = False
        # playback interpolator (utils_supersdr.py:999-1005): taps and history live in the GPU context
        self.n_tap = 33
        self.audio_rec = _NoRecording()
        self.hub = hub if hub is not None else kiwi_wf.hub
        self.channel = kiwi_wf.channel if channel is None else channel
        if getattr(self.hub, "kiwi_rate", self.KIWI_RATE) != self.KIWI_RATE:     # "audio_init audio_rate=" (:988-994)
            self.KIWI_RATE = int(self.hub.kiwi_rate)
            self.KIWI_RATE_TRUE = float(self.KIWI_RATE)
            self.SAMPLE_RATIO = self.AUDIO_RATE / self.KIWI_RATE
        self._timeout = timeout
        self.center_khz = float(kiwi_wf.freq)            # the IQ band's centre: tuning is relative to it
        self._play_blocks = {}
        self.error = None                                # set by play_buffer when it has to give up
        if hasattr(self.hub, "snd_clients"):
            self.hub.snd_clients[self.channel] = self
        self.set_mode_freq_pb()
        self.set_agc_params()

    # ---- control plane: the SET commands become ssdr_set_params
    def _push_params(self):
        mode = self.radio_mode.lower()
        p = default_params(mode if mode in L.MODE_BY_NAME else "am",
                           f_shift_hz=(self.freq - self.center_khz) * 1000.0, low_cut=float(self.lc), high_cut=float(self.hc),
                           agc_on=int(bool(self.on)), agc_hang=int(bool(self.hang)), agc_thresh=float(self.thresh),
                           agc_slope=float(self.slope), agc_decay=float(self.decay), agc_man_gain=float(self.gain))
        self.hub.set_params(self.channel, p)

    def change_agc_delay(self, delta):                   # utils_supersdr.py:1009-1020
        if delta < 0:
            if self.decay > self.min_agc_delay:
                self.decay += delta
        elif self.decay < self.max_agc_delay:
            self.decay += delta
        if self.radio_mode == "CW":
            self.decay_cw = self.decay
        else:
            self.decay_other = self.decay

    def set_agc_params(self):                            # "SET agc=..." utils_supersdr.py:1022-1024
        self._push_params()

    def set_mode_freq_pb(self):                          # "SET mod=..." utils_supersdr.py:1026-1029
        self.decay = self.decay_other if self.radio_mode != "CW" else self.decay_cw
        self._push_params()

    def change_passband(self, delta_low_, delta_high_):  # utils_supersdr.py:1078-1092
        if self.radio_mode == "USB":
            lc_, hc_ = LOW_CUT_SSB + delta_low_, HIGH_CUT_SSB + delta_high_
        elif self.radio_mode == "LSB":
            lc_, hc_ = -HIGH_CUT_SSB - delta_high_, -LOW_CUT_SSB - delta_low_
        elif self.radio_mode == "AM":
            lc_, hc_ = -HIGHLOW_CUT_AM - delta_low_, HIGHLOW_CUT_AM + delta_high_
        elif self.radio_mode == "CW":
            lc_, hc_ = LOW_CUT_CW + delta_low_, HIGH_CUT_CW + delta_high_
        else:
            lc_, hc_ = self.lc, self.hc
        self.lc, self.hc = lc_, hc_
        return lc_, hc_

    def keepalive(self):
        pass

    def close_connection(self):
        self.terminate = True

    # ---- the seam: utils_supersdr.py:1044-1076
    def process_audio_stream(self):
        try:
            samples, rssi, blk = self.hub.snd_queue[self.channel].get(timeout=self._timeout)
        except queue.Empty:
            self.terminate = True
            self.kiwi_wf.terminate = True
            raise
        self.rssi = rssi
        if blk is not None:
            self._play_blocks[id(samples)] = blk         # the frame's 48 kHz stereo block from ssdr_run_playbuffer
        return samples

    def get_audio_chunk(self):                           # utils_supersdr.py:1031-1042
        try:
            return self.process_audio_stream()
        except Exception:
            self.terminate = True
            return None

    # ---- playback stage: blocks interpolated, panned and packed by ssdr_run_playbuffer
    def play_buffer(self, outdata, frame_count, time_info, status):   # utils_supersdr.py:1106-1148
        self.status = status
        if self.late_flag:
            outdata[:] = 0
            return
        frames = [self.audio_buffer.get() for _ in range(self.CHUNKS)]
        blocks = [self._play_blocks.pop(id(f), None) for f in frames]
        if all(b is not None for b in blocks):           # interpolated, panned and packed on the GPU
            outdata[:] = np.concatenate(blocks)
            self._mute_logic(outdata)
            return
        # No host implementation exists.  This is the PortAudio callback, which must not raise: as the
        # reference does for its own stream errors (utils_supersdr.py:1031-1036), play silence, stop the worker and keep
        # the error where the owner finds it.
        outdata[:] = 0
        self.error = RuntimeError("play_buffer runs on the GPU (ssdr_run_playbuffer): a frame came without its 48 kHz "
                                  "block -- the hub was built with gpu_post=False")
        logging.error("%s", self.error)
        self.terminate = True

    def _mute_logic(self, outdata):                      # utils_supersdr.py:1142-1147
        if self.rssi > self.max_rssi_before_mute:
            self.mute_counter = self.muting_delay
        elif self.mute_counter > 0:
            self.mute_counter -= 1
        if self.mute_counter > 0:
            outdata *= 0

    def run(self):                                       # pacing loop: utils_supersdr.py:1150-1186
        self.total_delay_ms = 0.0
        delta_time_ms = 0.0
        self.ms_per_frame = (self.KIWI_SAMPLES_PER_FRAME / self.KIWI_RATE_TRUE) * 1000
        self.late_flag = False
        while not self.terminate:
            time_prev = time.time_ns() / 1000000
            snd_buf = self.get_audio_chunk()
            if snd_buf is not None and not self.late_flag:
                self.audio_buffer.put(snd_buf)
                self.run_index += 1
                self.total_delay_ms -= delta_time_ms
            else:
                self.total_delay_ms -= self.ms_per_frame
            delta_time_ms = time.time_ns() / 1000000 - time_prev
            self.total_delay_ms += delta_time_ms
            if not self.late_flag and self.total_delay_ms > (self.FULL_BUFF_LEN + 2) * self.ms_per_frame:
                self.late_flag = True
            if self.late_flag and self.total_delay_ms < self.ms_per_frame:
                while self.audio_buffer.qsize() < self.FULL_BUFF_LEN and not self.terminate:
                    snd_buf = self.get_audio_chunk()
                    if snd_buf is not None:
                        self.audio_buffer.put(snd_buf)
                self.late_flag = False
                self.total_delay_ms = 0.0
                delta_time_ms = 0.0
