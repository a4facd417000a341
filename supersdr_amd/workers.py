"""The GPU path behind the reference's two receiver workers.

The reference's `kiwi_waterfall` and `kiwi_sound` (utils_supersdr.py:592-898, 901-1186) are the boundary of this path
(SURVEY.md section 8b).  Nothing of them is restated here.  This module supplies what stands on the other side of them:

    IQHub           one GPU context for a block of receiver channels: batches the channels' IQ into superframes
                    (1024 samples = 1 waterfall line + 2 audio frames) and runs the kernels once per superframe
    GpuStream       what the workers hold where the reference holds a websocket `Stream` (self.wf_stream / self.stream):
                    send_message() takes the client's "SET ..." text commands and turns them into ssdr_set_params,
                    receive_message() returns W/F and SND frames in the KiwiSDR wire format, built from GPU results --
                    so even the maintainer's UNMODIFIED classes run on it
    WaterfallSeams, SoundSeams
                    mixins that go in front of the maintainer's own classes and replace exactly the seams:
                        receive_spectrum()       utils_supersdr.py:780-785   the next waterfall line from the GPU
                        spectrum_db2col()        :787-813                    result of ssdr_run_db2col, no host arithmetic
                        run()                    :879-897                    time binning on the GPU (int16 sums / N)
                        process_audio_stream()   :1044-1076                  the next PCM frame from the GPU
                        play_buffer()            :1106-1148                  48 kHz block of ssdr_run_playbuffer
                    and the socket part of the two constructors (:648-689, 946-994): while the maintainer's own
                    __init__ runs, the names it dials out with (kiwi_sdr, socket, wsclient, Stream) resolve to the GPU
                    stream.  Everything else -- frequency / zoom arithmetic, passband tables, the AGC stepper, the pacing
                    loop, the recorder, every attribute supersdr.py reads -- is the maintainer's code, inherited.
    bind(module)    -> namespace(kiwi_waterfall, kiwi_sound): the two mixins bound over `module`'s classes

There is no host implementation of spectrum_db2col or of the play_buffer interpolator in this package:
`IQHub(gpu_post=False)` skips the two kernels for consumers that only want raw lines and PCM; spectrum_db2col() then
raises, and play_buffer() (a PortAudio callback, which must not raise) plays silence, stops the worker and leaves the
error in `kiwi_sound.error`.
"""
import contextlib
import ctypes as C
import logging
import queue
import struct
import threading
import types
from collections import deque

import numpy as np

from . import _lib as L
from ._lib import Db2colChan, PlayChan
from .engine import SsdrEngine, default_params

IQ_SPAN_KHZ = L.RATE / 1000.0          # what one channel's GPU waterfall covers: the IQ band around its centre (12 kHz; hub.iq_span_khz)


class Frame(np.ndarray):
    """One SND frame as kiwi_sound.process_audio_stream returns it (int16[512]) that also carries what the GPU
    produced with it: the 48 kHz stereo block of play_buffer, the mono block of its recording branch, the header
    fields.  A frame that is dropped (late stream) takes its blocks with it -- nothing is keyed on the side."""
    play_block = None
    rec_block = None
    iq_block = None                     # "SET mod=iq": int16 [512, 2] I,Q of the frame (the PCM samples are its I column)
    adc_overflow = False
    rssi = -127.0

    def __array_finalize__(self, obj):
        pass

    @classmethod
    def make(cls, pcm, rssi, play_block=None, rec_block=None, adc_overflow=False, iq_block=None):
        f = np.array(pcm, np.int16).view(cls)
        f.rssi, f.play_block, f.rec_block, f.adc_overflow = float(rssi), play_block, rec_block, bool(adc_overflow)
        f.iq_block = iq_block
        return f


class SuperframeResult:
    """What one GPU run of the hub produced, as the arrays the engine returned (all channels, no per-channel objects):
    wf int16 [lines, n_ch, 1024] sums of n_avg byte lines, pcm int16 [n_ch, frames*512], rssi float32 [n_ch, frames],
    flags uint8 [n_ch, frames]; with gpu_post also color float32 [lines, n_ch, 1024], chans (Db2colChan per channel, as
    spectrum_db2col left them), play int16 [n_ch, frames*L, 2], mono (recording) -- or None where that stage did not run.
    On a lazy hub the post-processing runs for the channels with a worker only: post_channels lists them, and color / chans /
    play / mono have one entry per listed channel, in that order (post_channels None: one per channel).
    On a lazy_out hub (SSDR_FEED_LAZY_OUT) wf / pcm / rssi / flags / wire_rssi too have one row per channel of out_channels (the attached
    channels at the batch's submit) instead of one per channel; out_channels None: one per channel.
    In pipeline mode the arrays are views of the feed's pinned slots: valid until `depth - 1` further superframes have
    been collected (copy what must live longer)."""
    __slots__ = ("seq", "wf", "n_avg", "color", "chans", "pcm", "rssi", "flags", "play", "mono", "iq", "wire_rssi", "post_channels", "out_channels")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


class _ClientSlots(list):
    """hub.wf_clients / hub.snd_clients: the worker object of every channel (None: nobody).  Assigning an entry attaches
    the channel: from then on its results are queued for that worker."""

    def __init__(self, n, on_change):
        super().__init__([None] * n)
        self._on_change = on_change

    def __setitem__(self, c, obj):
        old = self[c]
        super().__setitem__(c, obj)
        self._on_change(int(c), old, obj)


class _Queues:
    """hub.wf_queue / hub.snd_queue: one bounded queue per ATTACHED channel.  A hub of a few receivers attaches every channel
    at start (`lazy=False`); a hub of 10^5 channels creates a queue -- and per-channel result objects -- only for the
    channels somebody listens to: indexing a channel attaches it (results produced from then on are queued)."""

    def __init__(self, hub, kind, maxsize):
        self._hub, self._kind, self._max, self._q = hub, kind, maxsize, {}

    def __getitem__(self, c):
        q = self._q.get(c)
        return q if q is not None else self._hub.attach(c, **{self._kind: True})[self._kind]

    def __len__(self):
        return self._hub.n_ch

    def attached(self, c):
        return self._q.get(c)


class IQHub:
    """Batches per-channel IQ into superframes and runs the GPU path for all channels at once.

    Ingest (KiwiSDRStream._process_iq_samples, kiwi/client.py:493-494, for a whole block of receivers):
        feed(channel, iq[n, 2])                      samples of one channel, any n
        feed_block(first_channel, iq[k, n, 2])       the same n samples for k consecutive channels: ONE strided copy
        reserve(first_channel, k) / commit(...)      the block's place in the open superframe slot, to be filled in place
        feed_wire_block(first_channel, bodies[k, f, 2065])   hubs built with wire=True: SND bodies as they come off the socket
    The ring is a short ring of SUPERFRAME SLOTS, each laid out as the batch the engine takes ([n_ch, samples, 2], pinned
    host memory when the hub is pipelined): a channel's samples are written once, where the H2D copy will read them.
    Per-channel state is NumPy (write positions, stall / drop counters); whether a superframe can run is decided from
    two incrementally kept numbers (channels that completed the open slot, the furthest write position), not by a scan.
    Whenever every channel has a superframe buffered, it is pushed, the kernels run, and the results are kept as the
    arrays the engine returned (`last`, `subscribe()`); per-channel queues and Frame objects exist only for ATTACHED
    channels -- those with a kiwi_waterfall / kiwi_sound / GpuStream on them (or attach()):
        wf_queue[c]  : (int16[1024] sum of N byte lines, N, db2col result or None)
        snd_queue[c] : Frame (int16[512] pcm + rssi, ADC-overflow flag, 48 kHz blocks) per audio frame
    `lazy=None` attaches every channel at start on hubs of up to 1024 channels (a receiver UI) and none above that.

    A receiver that stalls or reconnects (KiwiWorker sleeps 5-15 s on its retry paths, kiwi/worker.py:58, 66) does not
    stop the others: once a healthy channel is `stall_superframes` ahead, the hub runs anyway and the lagging channel's
    superframe is zero-filled (`stalled[c]` counts them; what it had buffered moves to the next slot).  A channel may
    run `backlog_superframes` ahead; beyond that its oldest buffered superframe is dropped (`dropped[c]` counts samples).

    Time binning: every kiwi_waterfall asks for its own N (the reference keeps averaging_n per instance,
    utils_supersdr.py:881-886).  The GPU sums N lines when all clients agree; when they disagree it delivers single
    lines and the clients that want N > 1 take the reference's own mean of N of them.

    pipeline=True sends the superframes through ssdr_feed_*: copy-in / kernels / copy-out of consecutive superframes
    overlapped, the copy-in straight from the hub's (pinned) slot (ssdr_feed_submit_from); with gpu_post the slot pipeline
    also runs spectrum_db2col and play_buffer (SSDR_FEED_POST) with the display state latched at submit.  Results arrive
    `depth - 1` superframes late (flush() drains) and are bit-identical to the synchronous hub's.
    `batch_superframes=K` runs K superframes per GPU call (K lines + 2K audio frames per channel and call): latency for
    launch efficiency at very large channel counts.  `exact_bins=True`: the waterfall stage in float64 (ssdr_set_exact_bins).  `copy_threads=T` splits feed_block's copy of a large block over T threads (default: up to 8 on hubs of 8192+ receivers).
    """

    LAZY_ABOVE = 1024

    def __init__(self, n_channels, device=0, engine=None, max_queue=64, gpu_post=True, kiwi_rate=12000, trace_rows=0,
                 backlog_superframes=8, stall_superframes=4, pipeline=False, depth=3, hop=1024, zoom=1, lazy=None,
                 batch_superframes=1, wire=False, copy_threads=None, exact_bins=False, lazy_out=False):
        self.n_ch = int(n_channels)
        # argument combinations that cannot work are refused BEFORE an engine (a GPU context) exists
        self._lazy = (self.n_ch > self.LAZY_ABOVE) if lazy is None else bool(lazy)
        if int(zoom) != 1 and pipeline:
            raise ValueError("zoom needs the synchronous hub (the pipelined feed's slots hold un-zoomed lines)")
        if lazy_out and not (pipeline and self._lazy):
            raise ValueError("lazy_out needs the pipelined feed and a lazy hub (pipeline=True, lazy=True)")
        self.engine = engine if engine is not None else SsdrEngine(self.n_ch, device)
        # waterfall zoom ("SET zoom=", utils_supersdr.py:741, 839): the lines then span 1/zoom of the IQ band around each
        # channel's zoom centre (set_wf_center) and one line needs `zoom` superframes: the hub batches that many per GPU run
        self.zoom = int(zoom)
        if self.zoom != 1:
            self.engine.set_wf_zoom(self.zoom)
        self.batch_superframes = max(1, int(batch_superframes))
        self._sf = L.NFFT * self.zoom * self.batch_superframes       # samples per channel and GPU run
        if hop != L.NFFT:                            # 512: two waterfall lines per superframe, 23.4 lines/s (MAX_FPS = 23, utils:597)
            self.engine.set_hop(hop)
        if exact_bins:                               # the waterfall stage in float64: lines equal the NumPy float64 path bit for bit
            self.engine.set_exact_bins(True)
        # spectrum_db2col and play_buffer run on the GPU with every superframe
        self.gpu_post = bool(gpu_post)
        # kiwi_sound.KIWI_RATE as the server announces it (:988-994): the rate of the IQ the channels receive (12000, or 20250 from
        # a three-channel KiwiSDR) -- the channels' constants are compiled for it -- and of play_buffer, whose branch it selects (:1125)
        self.kiwi_rate = int(kiwi_rate)
        self.iq_span_khz = self.kiwi_rate / 1000.0
        self.engine.set_kiwi_rate(self.kiwi_rate)
        self.play_len = 2048
        if self.gpu_post:
            self.play_len = self.engine.playbuffer_frame_len()
        # display reductions: wf_data's newest rows stay on the device, fed by every db2col run
        self.trace_rows = int(trace_rows) if self.gpu_post else 0
        if self.trace_rows:
            self.engine.set_wfdata_rows(self.trace_rows)
        self._smeter = None
        self.pipeline = bool(pipeline)
        self._inflight, self._depth = 0, int(depth)
        self.wire = bool(wire)                       # slots hold SND bodies (kiwi/client.py:443-454), unpacked on the device
        # feed_block's one copy, split over a few threads when the block is large (NumPy copies outside the GIL): a single core moves
        # ~10-20 GB/s, a superframe of 10^5 receivers is 0.5 GB
        self._copy_pool = None
        if copy_threads is None:                     # a hub of 10^4+ receivers moves hundreds of MB per superframe: a few threads by default
            import os
            copy_threads = min(8, max(1, (os.cpu_count() or 1) // 2)) if self.n_ch >= 8192 else 0
        if copy_threads and copy_threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._copy_pool = ThreadPoolExecutor(int(copy_threads))
            self._copy_threads = int(copy_threads)
        # ---- the ring of superframe slots.  Unit = what a channel's write position counts: samples, or SND frames (wire)
        if self.wire:
            self._U, self._row = self._sf // L.FRAME, (L.WIRE_BODY,)
            dtype = np.uint8
        else:
            self._U, self._row = self._sf, (2,)
            dtype = np.int16
        self._cap = max(2, int(backlog_superframes) // self.batch_superframes)          # slots a channel may run ahead (incl. the open one)
        self._stall = max(1, int(stall_superframes) // self.batch_superframes) * self._U
        # a pipelined hub may not rewrite a slot before the batch that left from it has been collected: depth - 1 more slots
        self._nslots = self._cap + (self._depth - 1 if self.pipeline else 0)
        alloc = getattr(self.engine, "host_alloc", None) if self.pipeline else None
        shape = (self.n_ch, self._U) + self._row
        self._slots = [alloc(shape, dtype) if alloc else np.zeros(shape, dtype) for _ in range(self._nslots)]
        if alloc:
            for sl in self._slots:
                sl[...] = 0
        self._base = 0                               # index of the next superframe to run == slot self._base % nslots
        self._gw = np.zeros(self.n_ch, np.int64)     # write positions on the hub's time axis, units (>= base * U)
        self._nfull = 0                              # channels with gw >= (base + 1) * U, kept incrementally
        self._gmax = 0                               # max(gw), kept incrementally
        self.dropped = np.zeros(self.n_ch, np.int64)
        self.stalled = np.zeros(self.n_ch, np.int64)
        self._reserved = {}                          # first channel of a reserve() -> the write position it was given at
        # ---- results
        self.last = None                             # SuperframeResult of the newest GPU run
        self._subscribers = []
        self._max_queue = int(max_queue)
        self.wf_queue = _Queues(self, "wf", self._max_queue)
        self.snd_queue = _Queues(self, "snd", 2 * self._max_queue)
        self._wf_att, self._snd_att = [], []         # attached channels, sorted
        self._n_wf_clients = self._n_snd_clients = 0
        self.wf_clients = _ClientSlots(self.n_ch, self._wf_client_changed)      # kiwi_waterfall objects: display state for db2col
        self.snd_clients = _ClientSlots(self.n_ch, self._snd_client_changed)    # kiwi_sound objects: volume / balance for play_buffer
        # spectrum_db2col / play_buffer are per viewer: a lazy hub runs them for the channels with a worker only
        # (ssdr_set_post_channels); a hub that attaches everybody keeps them on every channel
        self._post_select = self._lazy and hasattr(self.engine, "set_post_channels")
        # lazy_out (round 5, SSDR_FEED_LAZY_OUT): only the ATTACHED channels' lines / PCM / RSSI / flags are copied back from the GPU
        # (a hub of 10^5 receivers has a handful of listeners); the result arrays then have one row per attached channel
        # (SuperframeResult.out_channels) and every channel's results stay on the device (engine.feed_device())
        self._lazy_out = bool(lazy_out)
        if self._lazy_out and not self._post_select:
            raise ValueError("lazy_out needs an engine with set_post_channels")
        self._att_count = {}                         # lazy_out: channel -> queues attached (wf, snd): the rows that come back, at most L.FEED_LAZY_MAX
        self._post_sel, self._post_pos, self._post_dirty = None, None, self._post_select
        self._inflight_sel = deque()                 # pipelined: the selection each batch in flight was submitted with
        self._alloc_post_arrays(self.n_ch)
        self._params = {}                            # channel -> ChanParams, for the channels that were given any
        self._default_params = default_params("am")
        self._n_iq_mode = 0
        self._want_n = np.ones(self.n_ch, np.int32)  # averaging_n asked for by each channel's waterfall client
        self._want_all = {1: self.n_ch}              # N -> channels that want it (all channels / channels with a client)
        self._want_cli = {}
        self.averaging_n = 1                         # what the GPU sums right now
        self._recording = False
        self._lock = threading.RLock()
        self.superframes = 0
        if self.pipeline:
            self.engine.feed_open(2 * self.zoom * self.batch_superframes, self._depth, post=self.gpu_post, **({"wire": True} if self.wire else {}),
                                  **({"lazy_out": True} if self._lazy_out else {}))
        if not self._lazy:
            for c in range(self.n_ch):
                self.attach(c, wf=True, snd=True)

    # ---- who listens
    def attach(self, channel, wf=False, snd=False):
        """create the result queue(s) of a channel: its waterfall lines / audio frames are queued from now on"""
        c = int(channel)
        if not 0 <= c < self.n_ch:
            raise IndexError("channel %d of %d" % (c, self.n_ch))
        with self._lock:
            import bisect
            if self._lazy_out and (wf or snd) and c not in self._att_count and len(self._att_count) >= L.FEED_LAZY_MAX:
                # ssdr_feed_submit refuses a batch whose selection has more rows than the compact buffers hold (SSDR_FEED_LAZY_MAX):
                # say so HERE, where the listener asks, not on the feeding thread a superframe later
                raise ValueError("a lazy_out hub hands back at most %d channels (SSDR_FEED_LAZY_MAX); channel %d would be one more -- "
                                 "open the hub without lazy_out to copy every channel back" % (L.FEED_LAZY_MAX, c))
            if wf and c not in self.wf_queue._q:
                self.wf_queue._q[c] = queue.Queue(self._max_queue)
                bisect.insort(self._wf_att, c)
            if snd and c not in self.snd_queue._q:
                self.snd_queue._q[c] = queue.Queue(2 * self._max_queue)
                bisect.insort(self._snd_att, c)
            n_q = (c in self.wf_queue._q) + (c in self.snd_queue._q)
            if n_q:
                self._att_count[c] = n_q
            self._post_dirty = self._post_select            # the selection follows who is attached (re-derived before the next superframe)
        return {"wf": self.wf_queue._q.get(c), "snd": self.snd_queue._q.get(c)}

    def detach(self, channel, wf=True, snd=True):
        c = int(channel)
        with self._lock:
            if wf and self.wf_queue._q.pop(c, None) is not None:
                self._wf_att.remove(c)
            if snd and self.snd_queue._q.pop(c, None) is not None:
                self._snd_att.remove(c)
            n_q = (c in self.wf_queue._q) + (c in self.snd_queue._q)
            if n_q:
                self._att_count[c] = n_q
            else:
                self._att_count.pop(c, None)
            self._post_dirty = self._post_select

    def subscribe(self, fn):
        """fn(SuperframeResult) after every GPU run, on the feeding thread, with the hub's lock held: the bulk consumer's
        hook (a recorder, a detector over all channels); per-channel consumers use the queues"""
        self._subscribers.append(fn)

    def _alloc_post_arrays(self, n):
        self._db_arr = (Db2colChan * max(n, 1))()
        self._play_arr = (PlayChan * max(n, 1))()
        _fill_struct_array(self._db_arr, self._db2col_chan(None))
        _fill_struct_array(self._play_arr, PlayChan(100.0, 0.0))

    @property
    def post_channels(self):
        """the channels spectrum_db2col / play_buffer run for (None: all of them)"""
        return self._post_sel

    def _apply_post_selection(self):
        """lazy hub: the post kernels' channel list follows the workers that are attached"""
        self._post_dirty = False
        if self._lazy_out:                               # what comes back from the GPU: every attached channel (worker or bare queue)
            sel = sorted(set(self._wf_att) | set(self._snd_att))
        else:
            sel = sorted({c for c in self._wf_att if self.wf_clients[c] is not None} | {c for c in self._snd_att if self.snd_clients[c] is not None})
        if sel == self._post_sel:
            return
        self.engine.set_post_channels(sel)
        self._post_sel, self._post_pos = sel, {c: i for i, c in enumerate(sel)}
        self._alloc_post_arrays(len(sel))

    def _wf_client_changed(self, c, old, new):
        with self._lock:
            self._post_dirty = self._post_select
            self._n_wf_clients += (new is not None) - (old is not None)
            n = int(self._want_n[c])
            if (new is not None) != (old is not None):
                self._want_cli[n] = self._want_cli.get(n, 0) + (1 if new is not None else -1)
                if not self._want_cli[n]:
                    del self._want_cli[n]
            if new is not None:
                self.attach(c, wf=True)
            elif not self._post_select:
                self._db_arr[c] = self._db2col_chan(None)

    def _snd_client_changed(self, c, old, new):
        with self._lock:
            self._post_dirty = self._post_select
            self._n_snd_clients += (new is not None) - (old is not None)
            if new is not None:
                self.attach(c, snd=True)
            elif not self._post_select:
                self._play_arr[c] = PlayChan(100.0, 0.0)

    # ---- control plane (forwarded SET commands)
    def params(self, channel):
        return self._params.get(int(channel), self._default_params)

    def set_params(self, channel, p):
        if self.pipeline and p.mode == L.MODE_IQ:    # the feed's slots hand out PCM rows only (an IQ channel's row carries I)
            raise ValueError("mod=iq needs the synchronous hub: the pipelined feed does not return I,Q pairs")
        with self._lock:
            self.engine.set_params(channel, [p])     # raises for parameters the library refuses; the old ones stay
            self._n_iq_mode += (p.mode == L.MODE_IQ) - (self.params(channel).mode == L.MODE_IQ)
            self._params[int(channel)] = p

    def set_wf_center(self, channel, offset_hz):
        """zoom centre of one channel, Hz from the centre of its IQ band (restarts that channel's zoomed stream)"""
        with self._lock:
            self.engine.set_wf_center(channel, [float(offset_hz)])

    def set_averaging(self, n, channel=None):
        """channel=None: every channel wants N (one receiver, or a caller that owns the whole hub)."""
        n = int(min(max(n, 1), 100))
        with self._lock:
            if channel is None:
                self._want_n[:] = n
                self._want_all = {n: self.n_ch}
                self._want_cli = {n: self._n_wf_clients} if self._n_wf_clients else {}
            else:
                c, old = int(channel), int(self._want_n[channel])
                if old != n:
                    self._want_n[c] = n
                    for cnt in (self._want_all,) + ((self._want_cli,) if self.wf_clients[c] is not None else ()):
                        cnt[n] = cnt.get(n, 0) + 1
                        cnt[old] -= 1
                        if not cnt[old]:
                            del cnt[old]
            wants = self._want_cli or self._want_all
            eff = next(iter(wants)) if len(wants) == 1 else 1
            if eff != self.averaging_n:
                self.averaging_n = eff
                self.engine.set_averaging(eff)

    # ---- data plane: ingest
    def backlog(self, channel):
        """samples (wire hubs: frames) of `channel` that are buffered and not yet run"""
        return int(self._gw[channel]) - self._base * self._U

    @property
    def ring_capacity(self):
        """what a channel may have buffered at most, samples (wire hubs: frames)"""
        return self._cap * self._U

    def feed(self, channel, iq):
        """samples of one channel: int16 [n, 2] (any n)"""
        iq = np.asarray(iq, np.int16).reshape(1, -1, 2)
        if self.wire:
            raise ValueError("this hub takes SND bodies (feed_wire_block)")
        with self._lock:
            self._feed_run(int(channel), iq)

    def feed_block(self, first_channel, iq):
        """the same number of samples for k consecutive channels: int16 [k, n, 2].  Channels that stand at the same write
        position (receivers fed in step) take one strided copy per slot they touch."""
        iq = np.asarray(iq, np.int16)
        if self.wire or iq.ndim != 3 or iq.shape[2] != 2 or not 0 <= first_channel <= self.n_ch - iq.shape[0]:
            raise ValueError("feed_block takes int16 [k, n, 2] for channels [first, first + k) of a sample hub")
        with self._lock:
            self._feed_runs(int(first_channel), iq)

    def feed_wire_block(self, first_channel, bodies):
        """wire hubs: f SND bodies (7-byte header, 10-byte GNSS stamp, 512 big-endian I,Q pairs: kiwi/client.py:443-454)
        for k consecutive channels, uint8 [k, f, 2065]; header strip and byte swap run on the device"""
        bodies = np.asarray(bodies, np.uint8)
        if not self.wire or bodies.ndim != 3 or bodies.shape[2] != L.WIRE_BODY or not 0 <= first_channel <= self.n_ch - bodies.shape[0]:
            raise ValueError("feed_wire_block takes uint8 [k, f, %d] for channels [first, first + k) of a wire hub" % L.WIRE_BODY)
        with self._lock:
            self._feed_runs(int(first_channel), bodies)

    def reserve(self, first_channel, k):
        """-> writable view [k, room, 2] (wire: [k, room, 2065]) of where the next samples of channels [first, first + k) go,
        or None when they do not stand at one write position (use feed_block then).  Fill view[:, :n] and commit(first, k, n):
        the ingest writes where the H2D copy reads, no copy in between.  A block that sits on its reservation until the others are
        `stall_superframes` ahead is overtaken like any stalled receiver (its superframe runs zero-filled): commit() then returns
        False, the samples count as dropped, and the block reserves again."""
        with self._lock:
            g = self._gw[first_channel:first_channel + k]
            g0 = int(g[0])
            if k > 1 and (g != g0).any():
                return None
            b, off = divmod(g0, self._U)
            if b >= self._base + self._cap:
                return None
            self._await_slot(b)
            self._reserved[int(first_channel)] = g0
            return self._slots[b % self._nslots][first_channel:first_channel + k, off:]

    def commit(self, first_channel, k, n):
        """n samples (wire: frames) per channel were written into the reserved view -> True (False: the reservation was overtaken)"""
        with self._lock:
            g0 = int(self._gw[first_channel])
            if self._reserved.pop(int(first_channel), g0) != g0:     # the stall rule moved these channels on: the view was a slot that has run
                self.dropped[first_channel:first_channel + k] += n
                return False
            b, off = divmod(g0, self._U)
            if n < 0 or off + n > self._U:
                raise ValueError("commit beyond the reserved room")
            self._advance(first_channel, k, g0, n)
            self._pump()
            return True

    def _feed_runs(self, first, data):
        """split a block into runs of channels that stand at the same write position"""
        g = self._gw[first:first + data.shape[0]]
        if data.shape[0] == 1 or (g == g[0]).all():
            return self._feed_run(first, data)
        cuts = np.flatnonzero(g[1:] != g[:-1]) + 1
        lo = 0
        for hi in list(cuts) + [data.shape[0]]:
            self._feed_run(first + lo, data[lo:hi])
            lo = int(hi)

    def _advance(self, first, k, g0, n):
        U = self._U
        self._gw[first:first + k] += n
        if g0 < (self._base + 1) * U <= g0 + n:
            self._nfull += k
        if g0 + n > self._gmax:
            self._gmax = g0 + n

    def _feed_run(self, first, data):
        """data [k, n, ...] for k channels at ONE write position: slot-sized pieces, pumping in between"""
        k, n = data.shape[0], data.shape[1]
        U, pos = self._U, 0
        while pos < n:
            g0 = int(self._gw[first])
            b, off = divmod(g0, U)
            if b >= self._base + self._cap:                      # nobody consumes: the oldest buffered superframe goes
                self._drop_oldest(first, k)
                continue
            self._await_slot(b)
            m = min(n - pos, U - off)
            dst, src = self._slots[b % self._nslots][first:first + k, off:off + m], data[:, pos:pos + m]
            if self._copy_pool is not None and src.nbytes >= (32 << 20) and k >= 4 * self._copy_threads:
                step = -(-k // self._copy_threads)
                list(self._copy_pool.map(lambda lo: np.copyto(dst[lo:lo + step], src[lo:lo + step]), range(0, k, step)))
            else:
                dst[...] = src
            self._advance(first, k, g0, m)
            pos += m
            self._pump()

    def _await_slot(self, b):
        """pipelined hub: slot b % nslots last left as batch b - nslots; it may be rewritten once that batch was collected"""
        if self.pipeline:
            while self._inflight and b - self._nslots >= self.superframes - self._inflight:
                self._collect()

    def _drop_oldest(self, first, k):
        """channels [first, first + k) stand at the end of the ring: their buffered superframes move one slot down"""
        U, ns = self._U, self._nslots
        for j in range(self._cap - 1):
            self._slots[(self._base + j) % ns][first:first + k] = self._slots[(self._base + j + 1) % ns][first:first + k]
        self._gw[first:first + k] -= U
        self.dropped[first:first + k] += U
        self._gmax = int(self._gw.max())

    def _pump(self):
        U = self._U
        while True:
            if self._nfull < self.n_ch and self._gmax - self._base * U < self._stall + U:
                return                                           # wait for the slowest channel, but not for ever
            cur = self._slots[self._base % self._nslots]
            if self._nfull < self.n_ch:                          # somebody is `stall` ahead: the laggards get silence
                lim = (self._base + 1) * U
                lag = np.flatnonzero(self._gw < lim)
                fill = self._gw[lag] - self._base * U
                nxt = self._slots[(self._base + 1) % self._nslots]
                for c, f in zip(lag[fill > 0].tolist(), fill[fill > 0].tolist()):     # what they had buffered waits for the next run
                    nxt[c, :f] = cur[c, :f]
                cur[lag] = 0
                self._gw[lag] += U
                self.stalled[lag] += 1
            if self.pipeline:
                self._run_pipelined(cur)
            else:
                self._run_superframe(cur)
            self._base += 1
            self._nfull = int(np.count_nonzero(self._gw >= (self._base + 1) * U))

    # ---- data plane: the GPU run and what becomes of its results
    def _sync_display_state(self):
        """the attached workers' display state into the two arrays the post kernels read (one entry per channel, or per
        selected channel on a lazy hub)"""
        if self._post_dirty:
            self._apply_post_selection()
        pos = self._post_pos
        for c in self._wf_att:
            w = self.wf_clients[c]
            pc = c if pos is None else pos.get(c)
            if w is not None and pc is not None:
                self._db_arr[pc] = self._db2col_chan(w)
        for c in self._snd_att:
            s = self.snd_clients[c]
            pc = c if pos is None else pos.get(c)
            if s is not None and pc is not None:
                self._play_arr[pc] = PlayChan(float(s.volume), float(s.audio_balance))

    def _sync_recording(self):
        rec = any(self.snd_clients[c] is not None and self.snd_clients[c].audio_rec.recording_flag for c in self._snd_att)
        if rec != self._recording:
            self.engine.set_recording(rec)
            self._recording = rec
        return rec

    def _run_superframe(self, batch):
        eng = self.engine
        wire_rssi = eng.push_iq_wire(batch) if self.wire else eng.push_iq(batch)
        n_avg = self.averaging_n
        wf = eng.run_wf()                             # [lines, n_ch, 1024]
        color = chans = None
        if self.gpu_post and (self._n_wf_clients or self._n_snd_clients):
            self._sync_display_state()
        if self.gpu_post and len(wf) and self._n_wf_clients:
            chans = self._db_arr
            color = eng.run_db2col(chans, len(wf))    # [lines, n_ch, 1024] float32 0..254
        pcm, rssi = eng.run_audio()                   # [n_ch, frames*512], [n_ch, frames]
        flags = eng.audio_flags()                     # [n_ch, frames] SND header bit 1 (utils_supersdr.py:1066-1067)
        iqo = eng.audio_iq() if self._n_iq_mode else None                        # channels in "SET mod=iq"
        play = mono = None
        if self.gpu_post and self._n_snd_clients:
            rec = self._sync_recording()
            play = eng.run_playbuffer(self._play_arr)
            if rec:
                mono = eng.playbuffer_mono()
        self.superframes += 1
        self._hand_out(SuperframeResult(seq=self.superframes, wf=wf, n_avg=n_avg, color=color, chans=chans, pcm=pcm, rssi=rssi,
                                        flags=flags, play=play, mono=mono, iq=iqo, wire_rssi=wire_rssi, post_channels=self._post_sel))

    def _hand_out(self, r):
        self.last = r
        for fn in self._subscribers:
            fn(r)
        P, wf, pcm = self.play_len, r.wf, r.pcm
        pos = None if r.post_channels is None else {c: i for i, c in enumerate(r.post_channels)}      # row of a channel in the post results
        opos = None if r.out_channels is None else (pos if r.out_channels is r.post_channels else {c: i for i, c in enumerate(r.out_channels)})
        for c in self._wf_att:
            q = self.wf_queue._q[c]
            pc = c if pos is None else pos.get(c)
            oc = c if opos is None else opos.get(c)          # row of the channel's line (lazy_out: attached after the batch left -> not in it)
            if oc is None:
                continue
            has_post = r.color is not None and self.wf_clients[c] is not None and pc is not None
            for i in range(len(wf)):
                post = None
                if has_post:
                    k = r.chans[pc]
                    post = (r.color[i, pc].copy(), k.low_clip_db, k.high_clip_db, k.dynamic_range, k.wf_min_db, k.wf_max_db)
                _put_drop_oldest(q, (wf[i, oc].copy(), r.n_avg, post))
        for c in self._snd_att:
            q = self.snd_queue._q[c]
            iq_mode = r.iq is not None and self.params(c).mode == L.MODE_IQ
            pc = c if pos is None else pos.get(c)
            oc = c if opos is None else opos.get(c)
            if oc is None:
                continue
            has_play = r.play is not None and pc is not None
            for f in range(pcm.shape[1] // L.FRAME):
                _put_drop_oldest(q, Frame.make(
                    pcm[oc, f * L.FRAME:(f + 1) * L.FRAME], r.rssi[oc, f],
                    r.play[pc, f * P:(f + 1) * P].copy() if has_play else None,
                    r.mono[pc, f * P:(f + 1) * P].copy() if has_play and r.mono is not None else None, r.flags[oc, f],
                    r.iq[c, f * L.FRAME:(f + 1) * L.FRAME].copy() if iq_mode else None))

    def _run_pipelined(self, batch):
        eng = self.engine
        if self.gpu_post:                             # the display state this superframe is converted with, latched now
            self._sync_recording()
            self._sync_display_state()
            eng.feed_post(self._db_arr, self._play_arr)
        if self._post_dirty:                          # (without gpu_post nobody else re-derives it: lazy_out's rows follow the attached channels)
            self._apply_post_selection()
        if hasattr(eng, "feed_submit_from"):
            eng.feed_submit_from(batch)               # the H2D copy reads the hub's slot itself
        else:
            eng.feed_slot()[:] = batch
            eng.feed_submit()
        # only a batch that WAS submitted has a selection in flight (a refused submit raises above: the deque must stay in step
        # with the engine's slots, or every later result would be paired with another batch's rows)
        self._inflight_sel.append(self._post_sel)
        self._inflight += 1
        self.superframes += 1
        if self._inflight == self._depth:
            self._collect()

    def _collect(self):
        eng = self.engine
        got = eng.feed_collect()
        wf, pcm, rssi = got[:3]
        self._inflight -= 1
        n_avg, flags = eng.feed_n_avg, eng.feed_flags          # per slot: the N in force at submit, this batch's flags
        color = chans = play = mono = None
        sel = self._inflight_sel.popleft() if self._inflight_sel else self._post_sel
        if self.gpu_post:
            color, chans, play, mono = eng.feed_collect_post()
            if not self._n_wf_clients:
                color = None
            if not self._n_snd_clients:
                play = mono = None
        self._hand_out(SuperframeResult(seq=self.superframes - self._inflight, wf=wf, n_avg=n_avg, color=color, chans=chans, pcm=pcm,
                                        rssi=rssi, flags=flags, play=play, mono=mono, wire_rssi=got[3] if len(got) > 3 else None,
                                        post_channels=sel, out_channels=sel if self._lazy_out else None))

    def flush(self):
        """pipeline mode: wait for the superframes still in flight and hand their results out"""
        with self._lock:
            while self._inflight:
                self._collect()

    def db2col_line(self, channel, wf_sum, n):
        """spectrum_db2col (utils_supersdr.py:787-813) of ONE line of one client on the GPU -- for a client that binned N
        single lines itself because the hub's clients disagree on N.  Returns the tuple run_db2col results travel in.
        Only that client's line is converted (ssdr_db2col_line): the other channels' lines, the batch results and the
        device copy of wf_data are not touched."""
        with self._lock:
            k = self._db2col_chan(self.wf_clients[channel])
            color = self.engine.db2col_line(wf_sum, n, k)
            return (color, k.low_clip_db, k.high_clip_db, k.dynamic_range, k.wf_min_db, k.wf_max_db)

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
            if self._smeter is None:                 # one ctypes array for all channels, read and written through a NumPy view
                self._smeter = (SmeterChan * self.n_ch)()
                _fill_struct_array(self._smeter, SmeterChan.start(-127.0))               # kiwi_sound.rssi before any frame
                self._smeter_np = np.frombuffer(self._smeter, dtype=np.dtype(
                    [("rssi_smooth", "f8"), ("rssi_smooth_slow", "f8"), ("hist", "f8", (10,)), ("hist_pos", "u4"), ("run_index", "u4"), ("decay_ms", "f8")]))
            v = self._smeter_np
            v["decay_ms"] = 4000.0 if decay_ms is None else float(decay_ms)
            if decay_ms is None:
                for c in self._snd_att:              # the workers' own AGC decay (utils_supersdr.py:941), where there is a worker
                    s = self.snd_clients[c]
                    if s is not None:
                        v["decay_ms"][c] = float(s.decay)
            self.engine.run_smeter(self._smeter, fps)
            return v["rssi_smooth"].copy(), v["rssi_smooth_slow"].copy()

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
        self._slots = []
        if self._copy_pool is not None:
            self._copy_pool.shutdown()
        self.engine.close()


def _fill_struct_array(arr, proto):
    """every element of a ctypes array = proto, without a Python loop over the channels"""
    n, size = len(arr), C.sizeof(proto)
    if n:
        buf = (C.c_char * (n * size)).from_buffer(arr)
        buf[:] = bytes(proto) * n


def _put_drop_oldest(q, item):
    try:
        q.put_nowait(item)
    except queue.Full:
        try:
            q.get_nowait()
        except queue.Empty:
            pass
        q.put_nowait(item)


# ---------------------------------------------------------------------------------------------------------------
# the KiwiSDR end of the workers' websocket, played by the GPU
# ---------------------------------------------------------------------------------------------------------------
class GpuStream:
    """The object a worker holds where the reference holds `Stream(request, options)` (utils_supersdr.py:733, 964).

    send_message(text): the client -> server commands of SURVEY.md appendix A.  The two that select the DSP become the
    channel's ssdr_chan_params (a13):
        "SET mod=%s low_cut=%d high_cut=%d freq=%.3f"                    utils_supersdr.py:976, 1028
        "SET agc=%d hang=%d thresh=%d slope=%d decay=%d manGain=%d"      :979, 1023
    "SET zoom=%d start=%d" (:741, 839) is remembered (`zoom`, `start`); the rest (auth, keepalive, compression, ...)
    has no meaning without a server and is accepted.  A modulation without a demodulator here, or a frequency outside
    the channel's IQ band, raises ValueError instead of being demodulated as something else.  "SET mod=iq" selects the
    channel's filtered baseband itself (SSDR_MODE_IQ): its SND frames then carry I,Q pairs behind a GNSS stamp.

    receive_message(): server -> client frames, byte for byte in the wire format the reference parses
    (utils_supersdr.py:782-784, 1065-1074): first what the constructors wait for ("MSG audio_init audio_rate= sample_rate=",
    then one empty W/F resp. SND frame), after that one frame per GPU result of this channel."""

    def __init__(self, hub, channel, kind, center_khz, timeout=5.0):
        self.hub, self.channel, self.kind, self.center_khz, self.timeout = hub, int(channel), kind, float(center_khz), timeout
        self.zoom = self.start = None
        self.seq = 0
        self.closed = False
        self._greeting = deque()
        if hasattr(hub, "attach"):                   # this channel has a listener now: its results are queued from here on
            hub.attach(self.channel, wf=(kind != "SND"), snd=(kind == "SND"))
        if kind == "SND":
            rate = int(getattr(hub, "kiwi_rate", L.RATE))
            # every server announces its rate first; the reference takes KIWI_RATE, KIWI_RATE_TRUE, SAMPLE_RATIO from it (:988-994)
            self._greeting.append(bytearray(("MSG audio_init=0 audio_rate=%d sample_rate=%.6f" % (rate, float(rate))).encode()))
            self._greeting.append(bytearray(b"SND" + bytes(7)))
        else:
            self._greeting.append(bytearray(b"W/F" + bytes(13)))

    # ---- client -> server
    def send_message(self, msg, *a, **k):
        if isinstance(msg, (bytes, bytearray)):
            msg = bytes(msg).decode()
        words = msg.split()
        if len(words) < 2 or words[0] != "SET":
            return
        kv = dict(w.split("=", 1) for w in words[1:] if "=" in w)
        if "mod" in kv:
            self._retune(kv)
        elif "agc" in kv:
            p = self.hub.params(self.channel)
            q = _copy_params(p, agc_on=int(kv["agc"]), agc_hang=int(kv.get("hang", 0)), agc_thresh=float(kv.get("thresh", -80)),
                             agc_slope=float(kv.get("slope", 0)), agc_decay=float(kv.get("decay", 4000)),
                             agc_man_gain=float(kv.get("manGain", 50)))
            self.hub.set_params(self.channel, q)
        elif "zoom" in kv:
            self.zoom, self.start = int(kv["zoom"]), int(kv.get("start", 0))

    def _retune(self, kv):
        mode = kv["mod"].lower()
        if mode not in L.MODE_BY_NAME:               # "SET mod=sam", "drm", ... have no demodulator here: say so
            raise ValueError("radio_mode %r has no demodulator on the GPU path (am, lsb, usb, cw, nbfm, iq)" % (kv["mod"],))
        freq = float(kv.get("freq", self.center_khz))
        f_shift = (freq - self.center_khz) * 1000.0
        rate = float(getattr(self.hub, "kiwi_rate", L.RATE))
        if abs(f_shift) > rate / 2:
            raise ValueError("tuning %.3f kHz is outside the %g kHz IQ band around %.3f kHz that channel %d receives"
                             % (freq, rate / 1000.0, self.center_khz, self.channel))
        p = self.hub.params(self.channel)
        q = _copy_params(p, mode=L.MODE_BY_NAME[mode], f_shift_hz=f_shift, low_cut=float(kv.get("low_cut", p.low_cut)),
                         high_cut=float(kv.get("high_cut", p.high_cut)))
        self.hub.set_params(self.channel, q)

    # ---- server -> client
    def receive_message(self):
        if self._greeting:
            return self._greeting.popleft()
        if self.closed:
            return None
        try:
            if self.kind == "SND":
                f = self.hub.snd_queue[self.channel].get(timeout=self.timeout)
                if f.iq_block is not None:           # the channel is in "SET mod=iq": I,Q pairs behind a GNSS stamp
                    return snd_iq_frame(f.iq_block, f.rssi, self._next_seq(), adc_overflow=f.adc_overflow)
                return snd_frame(f, f.rssi, self._next_seq(), adc_overflow=f.adc_overflow)
            while True:                              # a line summed for some client's N > 1 is not a wire line: skip it
                line, n, _ = self.hub.wf_queue[self.channel].get(timeout=self.timeout)
                if n == 1:
                    return wf_frame(line, self._next_seq())
        except queue.Empty:
            return None                              # what a cleanly closed connection returns (:1053-1058)

    def _next_seq(self):
        self.seq = (self.seq + 1) & 0xFFFFFFFF
        return self.seq

    def close_connection(self, *a, **k):
        self.closed = True


def wf_frame(byte_line, seq=0, x_bin=0, flags_zoom=0):
    """One W/F message as the server sends it with "SET wf_comp=0": tag, skip, '<III' header, uint8[1024] (appendix A)"""
    return bytearray(b"W/F\x00" + struct.pack("<III", x_bin, flags_zoom, seq) + np.asarray(byte_line).astype(np.uint8).tobytes())


def snd_frame(pcm, rssi, seq=0, adc_overflow=False):
    """One SND message with "SET compression=0": tag, flags (bit 1 = ADC overflow), '<I' seq, '>H' smeter with
    rssi = 0.1 smeter - 127, big-endian int16[512] (utils_supersdr.py:1065-1074)"""
    smeter = int(min(max(round((float(rssi) + 127.0) * 10.0), 0), 65535))
    return bytearray(b"SND" + struct.pack("<BI", 2 if adc_overflow else 0, seq) + struct.pack(">H", smeter) +
                     np.asarray(pcm, np.int16).astype(">i2").tobytes())


def snd_iq_frame(iq, rssi, seq=0, adc_overflow=False, gps=(0, 0, 0, 0)):
    """One SND message in IQ mode: the same 7-byte header, '<BBII' last_gps_solution, dummy, gpssec, gpsnsec, then big-endian
    int16 I0, Q0, I1, Q1, ... (kiwi/client.py:443-454)"""
    smeter = int(min(max(round((float(rssi) + 127.0) * 10.0), 0), 65535))
    return bytearray(b"SND" + struct.pack("<BI", 2 if adc_overflow else 0, seq) + struct.pack(">H", smeter) +
                     struct.pack("<BBII", *gps) + np.asarray(iq, np.int16).astype(">i2").tobytes())


def _copy_params(p, **over):
    q = type(p)()
    for name, _ in p._fields_:
        setattr(q, name, getattr(p, name))
    for k, v in over.items():
        setattr(q, k, v)
    return q


class _Socket:                                       # socket.socket() of the constructors: nothing to dial
    def connect(self, *a, **k):
        pass

    def close(self):
        pass

    def settimeout(self, *a):
        pass


@contextlib.contextmanager
def _server_is_the_gpu(module, stream):
    """While the maintainer's constructor runs, the four names it reaches a KiwiSDR with resolve to the GPU stream:
    the /status probe `kiwi_sdr(host, port)` (utils_supersdr.py:648, 946), `socket.socket()` (:661, 957), the websocket
    handshake `wsclient.ClientHandshakeProcessor / ClientRequest` (:722-731, 962-963) and `Stream(request, options)`
    (:733, 964).  Constructors run on the thread that builds the UI, one at a time, as in supersdr.py:107-133, 774."""
    status = types.SimpleNamespace(users=0, users_max=4, offline=False, active=True, freq_offset=0.0, gps={}, antenna="",
                                   name="GPU hub", min_freq=0, max_freq=30000)
    ws = types.SimpleNamespace(ClientHandshakeProcessor=lambda *a, **k: types.SimpleNamespace(handshake=lambda uri: None),
                               ClientRequest=lambda *a, **k: types.SimpleNamespace(ws_version=None))
    repl = {"kiwi_sdr": lambda *a, **k: status, "socket": types.SimpleNamespace(socket=_Socket, create_connection=lambda *a, **k: _Socket()),
            "wsclient": ws, "Stream": lambda *a, **k: stream}
    saved = {k: getattr(module, k) for k in repl if hasattr(module, k)}
    try:
        for k, v in repl.items():
            setattr(module, k, v)
        yield
    finally:
        for k in repl:
            if k in saved:
                setattr(module, k, saved[k])
            else:
                delattr(module, k)


# ---------------------------------------------------------------------------------------------------------------
# the seams
# ---------------------------------------------------------------------------------------------------------------
class WaterfallSeams:
    """In front of the maintainer's kiwi_waterfall: the seams of the module docstring, nothing else.

    What the GPU waterfall shows is the channel's 12 kHz IQ band (IQ_SPAN_KHZ around `iq_center_khz`, bin 512 = centre,
    11.72 Hz per bin), not a zoomable 0-30 MHz span: there is no server-side DDC behind it.  The reference's zoom / span
    arithmetic keeps running because supersdr.py drives it, but it only labels the display; `iq_bin_to_khz` /
    `iq_khz_to_bin` are the true axis of `spectrum` and `wf_data`, and `set_freq_zoom` does not retune anything."""
    _ref_module = None                               # set by bind()

    def __init__(self, host_, port_, pass_, zoom_, freq_, eibi, disp, hub=None, channel=0, timeout=5.0):
        if hub is None:
            raise ValueError("the GPU-backed kiwi_waterfall needs an IQHub (there is no server-side FFT to fall back to)")
        self.hub, self.channel, self._timeout = hub, int(channel), timeout
        self.iq_center_khz = float(freq_ if freq_ else 14200)        # centre of the IQ band this channel receives
        self._gpu_post = None
        self._own_binning = None                     # (sum int16[1024], lines) while binning single lines itself
        self._gpu_stream = GpuStream(hub, channel, "W/F", self.iq_center_khz, timeout)
        with _server_is_the_gpu(self._ref_module, self._gpu_stream):
            # the maintainer's constructor and its start_stream() (:719-745) run as they are: their handshake lands on the
            # stand-ins, their Stream(...) is the GPU stream, their "SET zoom= start=" ... commands go to send_message()
            super().__init__(host_, port_, pass_, zoom_, freq_, eibi, disp)
        if hasattr(hub, "wf_clients"):
            hub.wf_clients[self.channel] = self

    # ---- the true frequency axis of the GPU waterfall
    def set_iq_zoom_center(self, khz):
        """centre of this channel's zoomed waterfall (hub built with zoom > 1), an absolute frequency inside its IQ band"""
        if getattr(self.hub, "zoom", 1) == 1:        # no zoom stage runs: the lines stay centred on iq_center_khz, and so must the axis
            raise ValueError("set_iq_zoom_center needs a hub built with zoom > 1")
        self.hub.set_wf_center(self.channel, (float(khz) - self.iq_center_khz) * 1000.0)
        self.iq_zoom_center_khz = float(khz)

    def _iq_axis(self):
        span = getattr(self.hub, "iq_span_khz", IQ_SPAN_KHZ) / getattr(self.hub, "zoom", 1)
        return getattr(self, "iq_zoom_center_khz", self.iq_center_khz), span

    def iq_bin_to_khz(self, bin_):
        centre, span = self._iq_axis()
        return centre + (bin_ - self.WF_BINS / 2) * span / self.WF_BINS

    def iq_khz_to_bin(self, khz):
        centre, span = self._iq_axis()
        return (khz - centre) * self.WF_BINS / span + self.WF_BINS / 2

    def close_connection(self):
        self.terminate = True
        self._gpu_stream.close_connection()

    def _next_line(self):
        try:
            return self.hub.wf_queue[self.channel].get(timeout=self._timeout)
        except queue.Empty:
            self.terminate = True
            return None

    # ---- seam: utils_supersdr.py:780-785
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

    # ---- seam: utils_supersdr.py:787-813
    def spectrum_db2col(self):
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

    # ---- seam: utils_supersdr.py:879-897 with the time binning on the GPU
    def step(self):
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


class SoundSeams:
    """In front of the maintainer's kiwi_sound: process_audio_stream, play_buffer and the constructor's socket part."""
    _ref_module = None

    def __init__(self, freq_, mode_, lc_, hc_, password_, kiwi_wf, buffer_len, volume_=100, host_=None, port_=None,
                 subrx_=False, hub=None, channel=None, timeout=5.0):
        self.hub = hub if hub is not None else kiwi_wf.hub
        self.channel = kiwi_wf.channel if channel is None else int(channel)
        self._timeout = timeout
        self.center_khz = float(getattr(kiwi_wf, "iq_center_khz", kiwi_wf.freq))   # the IQ band's centre: tuning is relative to it
        self.error = None                                # set by play_buffer when it has to give up
        self.late_flag = False                           # (the reference creates it in run(); play_buffer reads it)
        self._gpu_stream = GpuStream(self.hub, self.channel, "SND", self.center_khz, timeout)
        if hasattr(self.hub, "snd_clients"):
            self.hub.snd_clients[self.channel] = None
        with _server_is_the_gpu(self._ref_module, self._gpu_stream):
            # the maintainer's constructor sends "SET mod= ..." and "SET agc= ..." itself (:975-980): the channel is tuned by it
            super().__init__(freq_, mode_, lc_, hc_, password_, kiwi_wf, buffer_len, volume_, host_, port_, subrx_)
        if hasattr(self.hub, "snd_clients"):
            self.hub.snd_clients[self.channel] = self

    def close_connection(self):
        self.terminate = True
        self._gpu_stream.close_connection()

    # ---- seam: utils_supersdr.py:1044-1076
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

    # ---- seam: utils_supersdr.py:1106-1148 -- blocks interpolated, panned and packed by ssdr_run_playbuffer
    def play_buffer(self, outdata, frame_count, time_info, status):
        self.status = status
        if self.late_flag:
            outdata[:] = 0
            return
        frames = [self.audio_buffer.get() for _ in range(self.CHUNKS)]
        blocks = [getattr(f, "play_block", None) for f in frames]
        if all(b is not None for b in blocks):
            outdata[:] = np.concatenate(blocks)
            if self.audio_rec.recording_flag:            # :1139-1140: the mono block before the pan, from the same kernel
                rec = [getattr(f, "rec_block", None) for f in frames]
                if all(r is not None for r in rec):      # (frames interpolated before start() carry none: skipped)
                    self.audio_rec.audio_buffer.append(np.concatenate(rec))
            if self.rssi > self.max_rssi_before_mute:    # TX mute, :1142-1147
                self.mute_counter = self.muting_delay
            elif self.mute_counter > 0:
                self.mute_counter -= 1
            if self.mute_counter > 0:
                outdata *= 0
            return
        # No host implementation exists.  This is the PortAudio callback, which must not raise: as the
        # reference does for its own stream errors (utils_supersdr.py:1031-1036), play silence, stop the worker and keep
        # the error where the owner finds it.
        outdata[:] = 0
        self.error = RuntimeError("play_buffer runs on the GPU (ssdr_run_playbuffer): a frame came without its 48 kHz "
                                  "block -- the hub was built with gpu_post=False")
        logging.error("%s", self.error)
        self.terminate = True


def bind(module):
    """-> namespace(kiwi_waterfall, kiwi_sound, module): the seams in front of `module`'s own two classes.

        import utils_supersdr
        gpu = supersdr_amd.workers.bind(utils_supersdr)
        kiwi_wf = gpu.kiwi_waterfall(host, port, password, zoom, freq, eibi, disp, hub=hub, channel=0)
    """
    wf = type("kiwi_waterfall", (WaterfallSeams, module.kiwi_waterfall), {"_ref_module": module, "__doc__": WaterfallSeams.__doc__})
    snd = type("kiwi_sound", (SoundSeams, module.kiwi_sound), {"_ref_module": module, "__doc__": SoundSeams.__doc__})
    return types.SimpleNamespace(kiwi_waterfall=wf, kiwi_sound=snd, module=module)


def bind_headless():
    """bind() over supersdr_amd.headless: the seams on a bare pair of classes (no UI module needed)"""
    from . import headless
    return bind(headless)
