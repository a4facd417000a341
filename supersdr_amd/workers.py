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

What the reference does on the host AFTER those seams and is pure control flow stays on the host, with
citations (time binning by division of the GPU's integer sums, scrolling, pacing, TX mute); the parts that are
statement-for-statement mirrors of the reference (constants, frequency / zoom arithmetic, passband tables, the pacing
loop, the recorder) live in `ref_surface.py`, this file holds the product's own logic.  The two
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
from .ref_surface import (WaterfallSurface, SoundSurface, audio_recording,            # noqa: F401  (re-exported names)
                          CW_PITCH, LOW_CUT_SSB, HIGH_CUT_SSB, LOW_CUT_CW, HIGH_CUT_CW, HIGHLOW_CUT_AM)

IQ_SPAN_KHZ = L.RATE / 1000.0          # what one channel's GPU waterfall covers: the 12 kHz IQ band around its centre


class Frame(np.ndarray):
    """One SND frame as kiwi_sound.process_audio_stream returns it (int16[512]) that also carries what the GPU
    produced with it: the 48 kHz stereo block of play_buffer, the mono block of its recording branch, the header
    fields.  A frame that is dropped (late stream) takes its blocks with it -- nothing is keyed on the side."""
    play_block = None
    rec_block = None
    adc_overflow = False
    rssi = -127.0

    def __array_finalize__(self, obj):
        pass

    @classmethod
    def make(cls, pcm, rssi, play_block=None, rec_block=None, adc_overflow=False):
        f = np.array(pcm, np.int16).view(cls)
        f.rssi, f.play_block, f.rec_block, f.adc_overflow = float(rssi), play_block, rec_block, bool(adc_overflow)
        return f


class IQHub:
    """Batches per-channel IQ into superframes and runs the GPU path for all channels at once.

    feed(channel, iq_int16[n,2]) appends samples of one channel (any n) to that channel's ring buffer; whenever
    every channel has >= 1024 samples buffered, one superframe is pushed, both kernels run, and the results land
    in per-channel queues:
        wf_queue[c]  : (int16[1024] sum of N byte lines, N, db2col result or None)
        snd_queue[c] : Frame (int16[512] pcm + rssi, ADC-overflow flag, 48 kHz blocks) per audio frame

    A receiver that stalls or reconnects (GpuKiwiWorker sleeps 5-15 s on its retry paths) does not stop the others:
    once a healthy channel is `stall_superframes` ahead, the hub runs anyway and the lagging channel's superframe is
    zero-filled (`stalled[c]` counts them).  A ring holds `backlog_superframes`; beyond that the oldest samples of
    that channel are dropped (`dropped[c]` counts samples).

    Time binning: every kiwi_waterfall asks for its own N (the reference keeps averaging_n per instance,
    utils_supersdr.py:881-886).  The GPU sums N lines when all clients agree; when they disagree it delivers single
    lines and the clients that want N > 1 take the reference's own mean of N of them (a group is never restarted by
    another client's call).

    pipeline=True (gpu_post=False only) sends the superframes through ssdr_feed_*: pinned slots, copy-in / kernels /
    copy-out of consecutive superframes overlapped; results then arrive `depth - 1` superframes late (flush() drains).
    """

    def __init__(self, n_channels, device=0, engine=None, max_queue=64, gpu_post=True, kiwi_rate=12000, trace_rows=0,
                 backlog_superframes=8, stall_superframes=4, pipeline=False, depth=3, hop=1024):
        self.n_ch = int(n_channels)
        self.engine = engine if engine is not None else SsdrEngine(self.n_ch, device)
        if hop != L.NFFT:                            # 512: two waterfall lines per superframe, 23.4 lines/s (MAX_FPS = 23, utils:597)
            self.engine.set_hop(hop)
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
        self._cap = max(2, int(backlog_superframes)) * L.NFFT
        self._stall = max(1, int(stall_superframes)) * L.NFFT
        self._ring = np.zeros((self.n_ch, self._cap, 2), np.int16)
        self._rd = [0] * self.n_ch                  # absolute sample counters; ring index = counter % cap
        self._wr = [0] * self.n_ch
        self._batch = np.zeros((self.n_ch, L.NFFT, 2), np.int16)
        self.dropped = [0] * self.n_ch
        self.stalled = [0] * self.n_ch
        self.wf_queue = [queue.Queue(max_queue) for _ in range(self.n_ch)]
        self.snd_queue = [queue.Queue(2 * max_queue) for _ in range(self.n_ch)]
        self._params = [default_params("am") for _ in range(self.n_ch)]
        self._want_n = [1] * self.n_ch              # averaging_n asked for by each channel's waterfall client
        self.averaging_n = 1                        # what the GPU sums right now
        self._recording = False
        self._lock = threading.Lock()
        self.superframes = 0
        self.pipeline = bool(pipeline)
        self._inflight, self._depth = 0, int(depth)
        if self.pipeline:
            if self.gpu_post:
                raise ValueError("pipeline=True needs gpu_post=False: db2col / play_buffer refer to the last un-pipelined batch")
            self.engine.feed_open(2, self._depth)

    # ---- control plane (forwarded SET commands)
    def params(self, channel):
        return self._params[channel]

    def set_params(self, channel, p):
        with self._lock:
            self.engine.set_params(channel, [p])     # raises for parameters the library refuses; the old ones stay
            self._params[channel] = p

    def set_averaging(self, n, channel=None):
        """channel=None: every channel wants N (one receiver, or a caller that owns the whole hub)."""
        n = int(min(max(n, 1), 100))
        with self._lock:
            if channel is None:
                self._want_n = [n] * self.n_ch
            else:
                self._want_n[channel] = n
            wants = {self._want_n[c] for c in range(self.n_ch) if self.wf_clients[c] is not None} or set(self._want_n)
            eff = wants.pop() if len(wants) == 1 else 1
            if eff != self.averaging_n:
                self.averaging_n = eff
                self.engine.set_averaging(eff)

    # ---- data plane
    def feed(self, channel, iq):
        iq = np.asarray(iq, np.int16).reshape(-1, 2)
        with self._lock:
            pos = 0
            while pos < len(iq):                                 # ring-sized pieces, pumping in between
                n = min(len(iq) - pos, L.NFFT)
                over = (self._wr[channel] - self._rd[channel]) + n - self._cap
                if over > 0:                                     # nobody consumes: drop-oldest, like the result queues
                    self._rd[channel] += over
                    self.dropped[channel] += over
                w = self._wr[channel] % self._cap
                first = min(n, self._cap - w)
                self._ring[channel, w:w + first] = iq[pos:pos + first]
                if first < n:
                    self._ring[channel, :n - first] = iq[pos + first:pos + n]
                self._wr[channel] += n
                pos += n
                self._pump()

    def _take(self, c):
        """next superframe of channel c into the batch; False (zero-filled) if the channel does not have one"""
        if self._wr[c] - self._rd[c] < L.NFFT:
            self._batch[c] = 0
            return False
        r = self._rd[c] % self._cap
        first = min(L.NFFT, self._cap - r)
        self._batch[c, :first] = self._ring[c, r:r + first]
        if first < L.NFFT:
            self._batch[c, first:] = self._ring[c, :L.NFFT - first]
        self._rd[c] += L.NFFT
        return True

    def _pump(self):
        while True:
            avail = [self._wr[c] - self._rd[c] for c in range(self.n_ch)]
            if min(avail) < L.NFFT and max(avail) < self._stall + L.NFFT:
                return                                           # wait for the slowest channel, but not for ever
            for c in range(self.n_ch):
                if not self._take(c):
                    self.stalled[c] += 1
            if self.pipeline:
                self._run_pipelined()
            else:
                self._run_superframe()

    def _run_superframe(self):
        eng = self.engine
        eng.push_iq(self._batch)
        n_avg = self.averaging_n
        wf = eng.run_wf()                             # [lines, n_ch, 1024]
        color = chans = None
        if self.gpu_post and len(wf) and any(w is not None for w in self.wf_clients):
            chans = [self._db2col_chan(w) for w in self.wf_clients]
            color = eng.run_db2col(chans, len(wf))    # [lines, n_ch, 1024] float32 0..254
        pcm, rssi = eng.run_audio()                   # [n_ch, 1024], [n_ch, 2]
        flags = eng.audio_flags()                     # [n_ch, 2] SND header bit 1 (utils_supersdr.py:1066-1067)
        play = mono = None
        if self.gpu_post and any(s is not None for s in self.snd_clients):
            rec = any(s is not None and s.audio_rec.recording_flag for s in self.snd_clients)
            if rec != self._recording:
                eng.set_recording(rec)
                self._recording = rec
            play = eng.run_playbuffer([PlayChan(float(s.volume), float(s.audio_balance)) if s is not None
                                       else PlayChan(100.0, 0.0) for s in self.snd_clients])
            if rec:
                mono = eng.playbuffer_mono()
        self.superframes += 1
        P = self.play_len
        for c in range(self.n_ch):
            for i, line in enumerate(wf):
                post = None
                if color is not None and self.wf_clients[c] is not None:
                    k = chans[c]
                    post = (color[i, c].copy(), k.low_clip_db, k.high_clip_db, k.dynamic_range, k.wf_min_db, k.wf_max_db)
                _put_drop_oldest(self.wf_queue[c], (line[c].copy(), n_avg, post))
            for f in range(2):
                _put_drop_oldest(self.snd_queue[c], Frame.make(
                    pcm[c, f * L.FRAME:(f + 1) * L.FRAME], rssi[c, f],
                    play[c, f * P:(f + 1) * P].copy() if play is not None else None,
                    mono[c, f * P:(f + 1) * P].copy() if mono is not None else None, flags[c, f]))

    def _run_pipelined(self):
        eng = self.engine
        eng.feed_slot()[:] = self._batch
        eng.feed_submit()
        self._inflight += 1
        self.superframes += 1
        if self._inflight == self._depth:
            self._collect()

    def _collect(self):
        wf, pcm, rssi = self.engine.feed_collect()[:3]
        self._inflight -= 1
        n_avg = self.averaging_n
        for c in range(self.n_ch):
            for line in wf:
                _put_drop_oldest(self.wf_queue[c], (line[c].copy(), n_avg, None))
            for f in range(2):
                _put_drop_oldest(self.snd_queue[c], Frame.make(pcm[c, f * L.FRAME:(f + 1) * L.FRAME], rssi[c, f]))

    def flush(self):
        """pipeline mode: wait for the superframes still in flight and hand their results out"""
        with self._lock:
            while self._inflight:
                self._collect()

    def db2col_line(self, channel, wf_sum, n):
        """spectrum_db2col (utils_supersdr.py:787-813) of ONE line of one client on the GPU -- for a client that binned N
        single lines itself because the hub's clients disagree on N.  Returns the tuple run_db2col results travel in."""
        with self._lock:
            if self.averaging_n != 1:
                raise RuntimeError("db2col_line is for clients that bin single lines themselves (the GPU runs at N = 1 then)")
            lines = np.zeros((1, self.n_ch, L.NFFT), np.int16)
            lines[0, channel] = wf_sum
            chans = [self._db2col_chan(w if c == channel else None) for c, w in enumerate(self.wf_clients)]
            self.engine.set_averaging(n)                  # the divisor of this one line; at N = 1 no partial sums exist to lose
            try:
                self.engine.set_wf_lines(lines)
                color = self.engine.run_db2col(chans, 1)
            finally:
                self.engine.set_averaging(1)
            k = chans[channel]
            return (color[0, channel].copy(), k.low_clip_db, k.high_clip_db, k.dynamic_range, k.wf_min_db, k.wf_max_db)

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
        if self.pipeline:
            try:
                self.flush()
                self.engine.feed_close()
            except Exception:
                pass
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


class kiwi_waterfall(WaterfallSurface):
    """kiwi_waterfall (utils_supersdr.py:592-898) with the W/F websocket replaced by the GPU.

    What the GPU waterfall shows is the channel's 12 kHz IQ band (IQ_SPAN_KHZ around `iq_center_khz`, bin 512 = centre,
    11.72 Hz per bin), not a zoomable 0-30 MHz span: there is no server-side DDC behind it.  The reference's zoom / span
    arithmetic (`zoom`, `span_khz`, `bins_to_khz`, `set_freq_zoom`, the +3*zoom dB of spectrum_db2col) is kept because
    supersdr.py drives it, but it only labels the display; `iq_bin_to_khz` / `iq_khz_to_bin` are the true axis of
    `spectrum` and `wf_data`, and `set_freq_zoom` does not retune anything."""

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
        self.iq_center_khz = float(self.freq)            # centre of the IQ band this channel receives
        self._gpu_post = None
        self._own_binning = None                         # (sum int32[1024], lines) while binning single lines itself
        if hasattr(hub, "wf_clients"):
            hub.wf_clients[channel] = self

    # ---- the true frequency axis of the GPU waterfall
    def iq_bin_to_khz(self, bin_):
        return self.iq_center_khz + (bin_ - self.WF_BINS / 2) * IQ_SPAN_KHZ / self.WF_BINS

    def iq_khz_to_bin(self, khz):
        return (khz - self.iq_center_khz) * self.WF_BINS / IQ_SPAN_KHZ + self.WF_BINS / 2

    def keepalive(self):
        pass                                             # no server to keep alive

    def close_connection(self):
        self.terminate = True

    def _next_line(self):
        try:
            return self.hub.wf_queue[self.channel].get(timeout=self._timeout)
        except queue.Empty:
            self.terminate = True
            return None

    # ---- the seam: utils_supersdr.py:780-785
    def receive_spectrum(self):
        """Leaves self.spectrum = float32[WF_BINS] in byte units (dBm = byte - 255)."""
        self.hub.set_averaging(1, self.channel)          # this client bins nothing; others keep their N
        while not self.terminate:
            item = self._next_line()
            if item is None:
                return
            line, n, self._gpu_post = item
            if n == 1:                                   # lines summed for a previous N of this client are stale
                self.spectrum = line.astype(np.float32)
                return

    def receive_binned_spectrum(self, n):
        """Time binning (utils_supersdr.py:881-886).  On the GPU when the hub's clients agree on N: one summed line per N
        input lines, float32(sum)/float32(N) bit-identical to the reference's np.mean over its deque.  When they disagree
        the hub delivers single lines and this client takes the reference's own mean of N of them."""
        self.hub.set_averaging(n, self.channel)
        single = deque([], n)
        while not self.terminate:
            item = self._next_line()
            if item is None:
                return
            line, n_used, post = item
            if n_used == n:
                self._gpu_post = post
                self.spectrum = line.astype(np.float32) / np.float32(n)
                return
            if n_used == 1:                              # utils_supersdr.py:881-886, on lines the GPU produced
                single.append(line.astype(np.float32))
                if len(single) == n:
                    self.spectrum = np.mean(single, axis=0)
                    self._gpu_post = None
                    self._own_binning = (np.sum([s.astype(np.int32) for s in single], axis=0).astype(np.int16), n)
                    return
            # anything else was summed for another N during a change-over: stale

    def spectrum_db2col(self):                           # utils_supersdr.py:787-813
        if self._gpu_post is None and self._own_binning is not None and getattr(self.hub, "gpu_post", False):
            wf_sum, n = self._own_binning                # a line this client binned itself: its own db2col run
            self._gpu_post = self.hub.db2col_line(self.channel, wf_sum, n)
        self._own_binning = None
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


class kiwi_sound(SoundSurface):
    """kiwi_sound (utils_supersdr.py:901-1186) with the SND websocket replaced by the GPU."""

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
        self.audio_rec = audio_recording(self)           # utils_supersdr.py:1006
        self.hub = hub if hub is not None else kiwi_wf.hub
        self.channel = kiwi_wf.channel if channel is None else channel
        if getattr(self.hub, "kiwi_rate", self.KIWI_RATE) != self.KIWI_RATE:     # "audio_init audio_rate=" (:988-994)
            self.KIWI_RATE = int(self.hub.kiwi_rate)
            self.KIWI_RATE_TRUE = float(self.KIWI_RATE)
            self.SAMPLE_RATIO = self.AUDIO_RATE / self.KIWI_RATE
        self._timeout = timeout
        self.center_khz = float(getattr(kiwi_wf, "iq_center_khz", kiwi_wf.freq))   # the IQ band's centre: tuning is relative to it
        self.error = None                                # set by play_buffer when it has to give up
        if hasattr(self.hub, "snd_clients"):
            self.hub.snd_clients[self.channel] = self
        self.set_mode_freq_pb()
        self.set_agc_params()

    # ---- control plane: the SET commands become ssdr_set_params
    def _push_params(self):
        mode = str(self.radio_mode).lower()
        if mode not in L.MODE_BY_NAME:                   # "SET mod=iq" and friends have no demodulator here: say so
            raise ValueError("radio_mode %r has no demodulator on the GPU path (am, lsb, usb, cw, nbfm)" % (self.radio_mode,))
        f_shift = (self.freq - self.center_khz) * 1000.0
        if abs(f_shift) > L.RATE / 2:
            raise ValueError("tuning %.3f kHz is outside the %g kHz IQ band around %.3f kHz that channel %d receives"
                             % (self.freq, IQ_SPAN_KHZ, self.center_khz, self.channel))
        p = default_params(mode, f_shift_hz=f_shift, low_cut=float(self.lc), high_cut=float(self.hc),
                           agc_on=int(bool(self.on)), agc_hang=int(bool(self.hang)), agc_thresh=float(self.thresh),
                           agc_slope=float(self.slope), agc_decay=float(self.decay), agc_man_gain=float(self.gain))
        self.hub.set_params(self.channel, p)

    def set_agc_params(self):                            # "SET agc=..." utils_supersdr.py:1022-1024
        self._push_params()

    def set_mode_freq_pb(self):                          # "SET mod=..." utils_supersdr.py:1026-1029
        self.decay = self.decay_other if self.radio_mode != "CW" else self.decay_cw
        self._push_params()

    def keepalive(self):
        pass

    def close_connection(self):
        self.terminate = True

    # ---- the seam: utils_supersdr.py:1044-1076
    def _next_frame(self):
        try:
            return self.hub.snd_queue[self.channel].get(timeout=self._timeout)
        except queue.Empty:
            self.terminate = True
            self.kiwi_wf.terminate = True
            raise

    def process_audio_stream(self):
        frame = self._next_frame()
        # sample-rate drift (utils_supersdr.py:1049-1052): when the accumulated difference between the stream's true rate
        # and the nominal one reaches a frame, one frame is read and thrown away
        if self.run_index * self.delta_t * self.KIWI_SAMPLES_PER_FRAME / self.KIWI_RATE >= self.KIWI_SAMPLES_PER_FRAME:
            frame = self._next_frame()
            self.run_index = 0
        self.adc_overflow_flag = True if frame.adc_overflow else False      # SND header flags & 2 (:1066-1067)
        self.rssi = frame.rssi                                              # :1068-1069
        return frame

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
        blocks = [getattr(f, "play_block", None) for f in frames]
        if all(b is not None for b in blocks):           # interpolated, panned and packed on the GPU
            outdata[:] = np.concatenate(blocks)
            if self.audio_rec.recording_flag:            # :1139-1140: the mono block before the pan, from the same kernel
                rec = [getattr(f, "rec_block", None) for f in frames]
                if all(r is not None for r in rec):      # (frames interpolated before start() carry none: skipped)
                    self.audio_rec.audio_buffer.append(np.concatenate(rec))
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
