"""SsdrEngine -- thin NumPy-facing wrapper over the libssdr C-ABI (one ctx == one GPU).

Host code stays Python, as in the reference; every number is produced by the HIP
kernels behind include/ssdr.h.  NumPy is used for host buffers only.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import SmeterChan, ChanParams, ChanConsts, ChanState, Db2colChan, PlayChan, check, lib

CONSTS_DTYPE = np.dtype([("mode", "<u4"), ("ntap8", "<u4"), ("dphi1", "<u4"), ("dphi2", "<u4"),
                         ("wf_cal_lin", "<f4"), ("smeter_cal_db", "<f4"), ("agc_c0", "<f4"), ("agc_c1", "<f4"),
                         ("agc_knee", "<f4"), ("agc_delta8", "<f4"), ("hang_frames", "<u4"), ("ntap", "<u4"),
                         ("tap_groups", "<u4"), ("fir_flags", "<u4"), ("decim", "<u4"), ("kfm", "<f4")])
STATE_DTYPE = np.dtype([("phi1", "<u4"), ("phi2", "<u4"), ("dc", "<f4"), ("agc_d", "<f4"), ("agc_m", "<f4", (8,)),
                        ("prev_re", "<f4"), ("prev_im", "<f4"), ("pad", "<u4", (2,))])
assert CONSTS_DTYPE.itemsize == 64 and STATE_DTYPE.itemsize == 64


def default_params(mode="am", **over):
    """Reference defaults for a receiver in `mode` (utils_supersdr.py:42-50, 936-944)."""
    p = ChanParams()
    m = L.MODE_BY_NAME[mode.lower()] if isinstance(mode, str) else int(mode)
    check(lib.ssdr_default_params(m, C.byref(p)), "ssdr_default_params")
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError("ssdr_chan_params has no field %r" % k)
        setattr(p, k, v)
    return p


def compile_params(p, decim=1, rate=L.RATE):
    """ChanParams -> (consts record, float32[128] taps), computed by the library's host code (decim: ssdr_set_decimation,
    rate: ssdr_set_kiwi_rate)."""
    k = ChanConsts()
    taps = np.zeros(L.NTAP_MAX, np.float32)
    check(lib.ssdr_compile_params_rate(C.byref(p), int(decim), int(rate), C.byref(k), taps.ctypes.data), "ssdr_compile_params")
    rec = np.frombuffer(bytes(k), dtype=CONSTS_DTYPE)[0]
    return rec, taps


def table(which):
    n = {L.T_WINDOW: 1024, L.T_TWIDDLE_RE: 512, L.T_TWIDDLE_IM: 512, L.T_DB_THRESH: 256}[which]
    out = np.empty(n, np.float32)
    check(lib.ssdr_table(which, out.ctypes.data, n), "ssdr_table")
    return out


# run_chain's channel-count floors of every new SsdrEngine: None = the library's (from the device); (0, 0) = none -- what a test suite that wants the
# one-read kernels on its small batches sets (tests/conftest.py)
DEFAULT_CHAIN_FLOORS = None


class SsdrEngine:
    def __init__(self, n_channels, device=0, chain_floors=None):
        self.n_ch = int(n_channels)
        self._ctx = L._P()
        check(lib.ssdr_create(int(device), self.n_ch, L.NFFT, L.FRAME, C.byref(self._ctx)), "ssdr_create")
        floors = DEFAULT_CHAIN_FLOORS if chain_floors is None else chain_floors          # "library": leave the device-derived ones alone
        if floors is not None and floors != "library":
            self.set_chain_floors(*floors)
        self.in_frames = 0
        self.hop = L.NFFT
        self.decim = 1
        self.averaging = 1
        self.zoom = 1
        self.kiwi_rate = L.RATE
        self.audio_frames = 0          # frames of the last run_audio / set_pcm (extent of the device PCM / RSSI / flags)
        self._pinned = []              # host_alloc()
        self.n_post = self.n_ch        # channels the post-processing works on (set_post_channels)
        self._feed_n_post = []         # ... of the batches in flight in the pipelined feed, oldest first

    def close(self):
        if self._ctx:
            self.host_free_all()
            lib.ssdr_destroy(self._ctx)
            self._ctx = L._P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- control plane
    def set_params(self, first, params):
        params = list(params)
        arr = (ChanParams * len(params))(*params)
        check(lib.ssdr_set_params(self._ctx, int(first), len(params), arr), "ssdr_set_params")

    def reset_state(self, first=0, count=None):
        check(lib.ssdr_reset_state(self._ctx, int(first), self.n_ch - first if count is None else int(count)),
              "ssdr_reset_state")

    def set_averaging(self, n):
        check(lib.ssdr_set_averaging(self._ctx, int(n)), "ssdr_set_averaging")
        self.averaging = int(n)

    def set_decimation(self, decim):
        """D in {1, 2, 4}: the IQ arrives at D * 12 kHz (push_iq then takes [n_ch, n_frames*512*D, 2]); resets the streams"""
        check(lib.ssdr_set_decimation(self._ctx, int(decim)), "ssdr_set_decimation")
        self.decim = int(decim)

    def set_hop(self, hop):
        """samples between waterfall lines: 1024 (default) or 512 (lines overlap by half: 23.4 lines/s, the reference's rate)"""
        check(lib.ssdr_set_hop(self._ctx, int(hop)), "ssdr_set_hop")
        self.hop = int(hop)

    def set_wf_zoom(self, zoom):
        """waterfall span = the IQ band / zoom (1, 2, 4, 8) around each channel's zoom centre (set_wf_center); ctx-wide.
        A batch must then hold a whole number of zoomed lines (1024 * zoom input samples each, 512 * zoom at hop 512)."""
        check(lib.ssdr_set_wf_zoom(self._ctx, int(zoom)), "ssdr_set_wf_zoom")
        self.zoom = int(zoom)

    def set_wf_center(self, first, offsets_hz):
        """zoom centres, Hz from the IQ band's centre, for channels first .. first + len(offsets_hz) - 1"""
        off = np.ascontiguousarray(offsets_hz, np.float64)
        check(lib.ssdr_set_wf_center(self._ctx, int(first), len(off), off.ctypes.data), "ssdr_set_wf_center")

    def read_zoom(self, first=0, count=None):
        """-> int16 [count, n, 2]: the zoomed I,Q stream the last run_wf drew its lines from"""
        count = self.n_ch - first if count is None else int(count)
        n = C.c_uint32(0)
        check(lib.ssdr_read_zoom(self._ctx, int(first), count, None, C.byref(n)), "ssdr_read_zoom")
        out = np.empty((count, n.value, 2), np.int16)
        check(lib.ssdr_read_zoom(self._ctx, int(first), count, out.ctypes.data, C.byref(n)), "ssdr_read_zoom")
        return out

    def set_exact_bins(self, on):
        """waterfall stage in float64: the int16 sums then equal the float64 definition (NumPy float64 FFT) bit for bit; ~1.9x the fp32 kernel's time"""
        check(lib.ssdr_set_exact_bins(self._ctx, int(bool(on))), "ssdr_set_exact_bins")

    # ---- data plane
    def push_iq(self, iq):
        """iq: int16 [n_ch, n_frames*512, 2] host array (copied to the GPU)."""
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        if iq.ndim != 3 or iq.shape[0] != self.n_ch or iq.shape[2] != 2 or iq.shape[1] % (L.FRAME * self.decim):
            raise ValueError("iq must be int16[n_ch=%d, k*%d, 2], got %r" % (self.n_ch, L.FRAME * self.decim, iq.shape))
        self.in_frames = iq.shape[1] // (L.FRAME * self.decim)
        check(lib.ssdr_push_iq(self._ctx, iq.ctypes.data, self.in_frames, 0), "ssdr_push_iq")
        self.sync()             # the host buffer may be released by the caller after this returns

    def push_iq_device(self, dev_ptr, n_frames):
        self.in_frames = int(n_frames)
        check(lib.ssdr_push_iq(self._ctx, int(dev_ptr), self.in_frames, 1), "ssdr_push_iq")

    def synth_iq(self, n_frames, seed=0x5D5D, first_channel_id=0):
        self.in_frames = int(n_frames)
        check(lib.ssdr_synth_iq(self._ctx, self.in_frames, int(seed), int(first_channel_id)), "ssdr_synth_iq")

    def read_input(self, first=0, count=None):
        count = self.n_ch - first if count is None else int(count)
        out = np.empty((count, self.in_frames * L.FRAME * self.decim, 2), np.int16)
        check(lib.ssdr_read_input(self._ctx, int(first), count, out.ctypes.data), "ssdr_read_input")
        return out

    def run_wf(self, fetch=True):
        """-> int16 [lines_ready, n_ch, 1024] sums of `averaging` byte lines (or the line count if not fetch)."""
        n = C.c_uint32(0)
        if not fetch:
            check(lib.ssdr_run_wf(self._ctx, None, C.byref(n), 0), "ssdr_run_wf")
            return n.value
        halves = self.in_frames * self.decim // self.zoom
        total_lines = (halves if self.hop == L.NFFT // 2 else halves // 2) + 1    # upper bound incl. a carried partial group
        out = np.empty((total_lines, self.n_ch, L.NFFT), np.int16)
        check(lib.ssdr_run_wf(self._ctx, out.ctypes.data, C.byref(n), 0), "ssdr_run_wf")
        return out[: n.value]

    def run_audio(self, fetch=True):
        """-> (int16 [n_ch, n_frames*512] pcm, float32 [n_ch, n_frames] rssi dBm)."""
        self.audio_frames = self.in_frames
        if not fetch:
            check(lib.ssdr_run_audio(self._ctx, None, None, 0), "ssdr_run_audio")
            return None
        pcm = np.empty((self.n_ch, self.in_frames * L.FRAME), np.int16)
        rssi = np.empty((self.n_ch, self.in_frames), np.float32)
        check(lib.ssdr_run_audio(self._ctx, pcm.ctypes.data, rssi.ctypes.data, 0), "ssdr_run_audio")
        return pcm, rssi

    def set_fused(self, on):
        """0 / False: never a one-read kernel; 1 / True (default): the fused superframe kernel at hop 1024 (full-band AM batches) and the
        wave-specialised chain kernel for batches whose channels all run the general audio path; 2: the former at hop 512 / N > 1 as well;
        3: the latter for every batch it can take (any mix of audio paths at hop 1024)"""
        check(lib.ssdr_set_fused(self._ctx, int(on)), "ssdr_set_fused")

    def set_chain_floors(self, fused_am_min_channels, chain_ws_min_channels):
        """fewest channels for which run_chain's default takes the fused AM / the wave-specialised kernel (0: no floor)"""
        check(lib.ssdr_set_chain_floors(self._ctx, int(fused_am_min_channels), int(chain_ws_min_channels)), "ssdr_set_chain_floors")

    def chain_floors(self):
        a, w = C.c_uint32(0), C.c_uint32(0)
        check(lib.ssdr_get_chain_floors(self._ctx, C.byref(a), C.byref(w)), "ssdr_get_chain_floors")
        return a.value, w.value

    def set_overlap(self, on):
        """un-fused run_chain batches: the audio stage beside the waterfall kernel on a second stream (default on)"""
        check(lib.ssdr_set_overlap(self._ctx, int(bool(on))), "ssdr_set_overlap")

    def run_chain(self):
        """both stages on the current batch, results left on the device -> (lines ready, which way: 0 the two stages side by side,
        1 ssdr_fused_am_kernel, 2 ssdr_chain_ws_kernel -- an int, not a bool: test it with `!= 0`)"""
        n, fused = C.c_uint32(0), C.c_int(0)
        self.audio_frames = self.in_frames
        check(lib.ssdr_run_chain(self._ctx, C.byref(n), C.byref(fused)), "ssdr_run_chain")
        return n.value, fused.value

    def fetch_wf(self, lines):
        """device results of the last waterfall run -> int16 [lines, n_ch, 1024]"""
        out = np.empty((lines, self.n_ch, L.NFFT), np.int16)
        ptr, n = L._P(), C.c_uint32(0)
        check(lib.ssdr_wf_device(self._ctx, C.byref(ptr), C.byref(n)), "ssdr_wf_device")
        check(lib.ssdr_copy_from_device(self._ctx, out.ctypes.data, ptr, out.nbytes), "ssdr_copy_from_device")
        return out

    def fetch_audio(self):
        """device results of the last audio run -> (int16 [n_ch, n_frames*512], float32 [n_ch, n_frames])"""
        pcm = np.empty((self.n_ch, self.audio_frames * L.FRAME), np.int16)
        rssi = np.empty((self.n_ch, self.audio_frames), np.float32)
        p, r = L._P(), L._P()
        check(lib.ssdr_audio_device(self._ctx, C.byref(p), C.byref(r)), "ssdr_audio_device")
        check(lib.ssdr_copy_from_device(self._ctx, pcm.ctypes.data, p, pcm.nbytes), "ssdr_copy_from_device")
        check(lib.ssdr_copy_from_device(self._ctx, rssi.ctypes.data, r, rssi.nbytes), "ssdr_copy_from_device")
        return pcm, rssi

    def fetch_rows(self, channels, lines):
        """the device-resident results of the last run for a FEW channels, row by row (ssdr_copy_from_device) -- for batches whose
        whole result arrays (64 GiB at 2^20 channels x 16 superframes) must not be copied back:
        -> (wf int16 [lines, k, 1024], pcm int16 [k, n_frames*512], rssi float32 [k, n_frames])"""
        ch = [int(c) for c in channels]
        wf = np.empty((lines, len(ch), L.NFFT), np.int16)
        pcm = np.empty((len(ch), self.audio_frames * L.FRAME), np.int16)
        rssi = np.empty((len(ch), self.audio_frames), np.float32)
        wptr, n, p, r = L._P(), C.c_uint32(0), L._P(), L._P()
        check(lib.ssdr_wf_device(self._ctx, C.byref(wptr), C.byref(n)), "ssdr_wf_device")
        check(lib.ssdr_audio_device(self._ctx, C.byref(p), C.byref(r)), "ssdr_audio_device")
        if lines > n.value:
            raise ValueError("the last run left %d lines, not %d" % (n.value, lines))
        row_wf, row_pcm, row_rssi = L.NFFT * 2, self.audio_frames * L.FRAME * 2, self.audio_frames * 4
        for i, c in enumerate(ch):
            for ln in range(lines):
                check(lib.ssdr_copy_from_device(self._ctx, wf[ln, i].ctypes.data, C.c_void_p(wptr.value + (ln * self.n_ch + c) * row_wf), row_wf), "ssdr_copy_from_device")
            check(lib.ssdr_copy_from_device(self._ctx, pcm[i].ctypes.data, C.c_void_p(p.value + c * row_pcm), row_pcm), "ssdr_copy_from_device")
            check(lib.ssdr_copy_from_device(self._ctx, rssi[i].ctypes.data, C.c_void_p(r.value + c * row_rssi), row_rssi), "ssdr_copy_from_device")
        return wf, pcm, rssi

    def audio_flags(self):
        """-> uint8 [n_ch, n_frames]: the SND header's ADC-overflow bit (utils_supersdr.py:1066-1067) for every frame of the
        last run_audio."""
        out = np.empty((self.n_ch, self.audio_frames), np.uint8)
        check(lib.ssdr_audio_flags(self._ctx, out.ctypes.data, 0), "ssdr_audio_flags")
        return out

    def audio_iq(self):
        """-> int16 [n_ch, n_frames*512, 2]: I,Q of the channels in "iq" mode for the last run_audio (rows of other modes: 0)"""
        out = np.empty((self.n_ch, self.audio_frames * L.FRAME, 2), np.int16)
        check(lib.ssdr_audio_iq(self._ctx, out.ctypes.data, 0), "ssdr_audio_iq")
        return out

    def sync(self):
        check(lib.ssdr_sync(self._ctx), "ssdr_sync")

    # ---- the reference's post-processing on the GPU
    def set_post_channels(self, channels=None):
        """the channels the post-processing works on from now on (ascending, unique; None: all of them).  Every per-channel
        array of run_db2col / run_playbuffer / playbuffer_mono / feed_post / feed_collect_post / run_trace then has
        len(channels) entries, in this order."""
        if channels is None:
            check(lib.ssdr_set_post_channels(self._ctx, None, 0), "ssdr_set_post_channels")
            self.n_post = self.n_ch
            return
        sel = np.ascontiguousarray(channels, np.uint32)
        check(lib.ssdr_set_post_channels(self._ctx, sel.ctypes.data if len(sel) else (C.c_uint32 * 1)(), len(sel)), "ssdr_set_post_channels")
        self.n_post = len(sel)

    def run_db2col(self, chans, lines, fetch=True):
        """spectrum_db2col for the lines of the last run_wf.  chans: list of Db2colChan, or a ctypes array
        (Db2colChan * n_post) that goes to the library as it is (no per-channel Python work); updated in place.
        -> float32 [lines, n_post, 1024] wf_color (n_post = n_ch unless set_post_channels named a subset)."""
        n = self.n_post
        arr = chans if isinstance(chans, C.Array) else (Db2colChan * max(n, 1))(*chans)
        out = np.empty((lines, n, L.NFFT), np.float32) if fetch else None
        check(lib.ssdr_run_db2col(self._ctx, arr, out.ctypes.data if fetch and n else None, 0), "ssdr_run_db2col")
        if arr is not chans:
            for i in range(n):
                chans[i] = arr[i]
        return out

    # ---- pipelined host feed (copy-in / kernels / copy-out of consecutive batches overlap)
    def feed_open(self, n_frames, depth=3, wire=False, post=False, lazy_out=False):
        """wire=True: slots take SND bodies uint8 [n_ch, n_frames, 2065] (kiwi/client.py:443-454), unpacked on the device.
        post=True: every batch also goes through spectrum_db2col / play_buffer on the device (feed_post, feed_collect_post).
        lazy_out=True (SSDR_FEED_LAZY_OUT): only the channels of set_post_channels come back to the host -- feed_collect's arrays then
        have one row per SELECTED channel (the selection in force at the batch's submit); every channel's results stay on the
        device (feed_device)."""
        check(lib.ssdr_feed_open(self._ctx, int(n_frames), int(depth), (1 if wire else 0) | (2 if post else 0) | (4 if lazy_out else 0)), "ssdr_feed_open")
        self._feed_frames, self._feed_wire, self._feed_post = int(n_frames), bool(wire), bool(post)
        self._feed_lazy = bool(lazy_out)
        self._feed_lines = 0
        self._feed_n_post, self._feed_last_n_post = [], self.n_post

    def feed_post(self, chans=None, play=None):
        """display state for the batches submitted from now on: lists of Db2colChan / PlayChan (None keeps the previous)"""
        a = chans if chans is None or isinstance(chans, C.Array) else (Db2colChan * max(self.n_post, 1))(*chans)
        b = play if play is None or isinstance(play, C.Array) else (PlayChan * max(self.n_post, 1))(*play)
        check(lib.ssdr_feed_post(self._ctx, a, b), "ssdr_feed_post")

    def feed_collect_post(self):
        """of the batch feed_collect returned last -> (color float32 [lines, n_ch, 1024] or None, [Db2colChan] or None,
        play int16 [n_ch, n_frames*L, 2], mono int16 [n_ch, n_frames*L] or None), views of pinned memory"""
        col, ch, pl, mo = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.ssdr_feed_collect_post(self._ctx, C.byref(col), C.byref(ch), C.byref(pl), C.byref(mo)), "ssdr_feed_collect_post")
        nf, nl, P = self._feed_frames, self._feed_lines, self.playbuffer_frame_len()
        n = self._feed_last_n_post                  # what the batch was post-processed for: the selection at ITS submit
        color = chans = mono = play = None
        if n == 0:
            return None, None, None, None
        if nl and col.value:
            color = np.ctypeslib.as_array(C.cast(col, C.POINTER(C.c_float)), shape=(nl * n * L.NFFT,)).reshape(nl, n, L.NFFT)
            chans = (Db2colChan * n).from_address(ch.value)                # indexable view of the slot's pinned copy
        play = np.ctypeslib.as_array(C.cast(pl, C.POINTER(C.c_int16)), shape=(n * nf * P * 2,)).reshape(n, nf * P, 2)
        if mo.value:
            mono = np.ctypeslib.as_array(C.cast(mo, C.POINTER(C.c_int16)), shape=(n * nf * P,)).reshape(n, nf * P)
        return color, chans, play, mono

    def feed_slot(self):
        """-> int16 [n_ch, n_frames*512, 2] (or uint8 [n_ch, n_frames, 2065]) view of the next pinned slot (fill it, then
        feed_submit())."""
        p = C.c_void_p()
        check(lib.ssdr_feed_slot(self._ctx, C.byref(p)), "ssdr_feed_slot")
        if self._feed_wire:
            n = self.n_ch * self._feed_frames * 2065
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).reshape(self.n_ch, self._feed_frames, 2065)
        n = self.n_ch * self._feed_frames * L.FRAME * 2
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), shape=(n,)).reshape(self.n_ch, -1, 2)

    def feed_submit(self):
        check(lib.ssdr_feed_submit(self._ctx), "ssdr_feed_submit")
        self._feed_n_post.append(self.n_post)

    def feed_submit_from(self, batch):
        """queue a batch that lies in the caller's own host array (layout of feed_slot(); ideally from host_alloc): no copy
        into a slot.  The array must stay untouched until feed_collect has returned this batch."""
        if not (batch.flags.c_contiguous and batch.nbytes == self._feed_bytes()):
            raise ValueError("feed_submit_from: the batch must be C-contiguous and exactly one slot in size")
        check(lib.ssdr_feed_submit_from(self._ctx, batch.ctypes.data), "ssdr_feed_submit_from")
        self._feed_n_post.append(self.n_post)

    def _feed_bytes(self):
        return self.n_ch * self._feed_frames * (L.WIRE_BODY if self._feed_wire else L.FRAME * 4)

    def host_alloc(self, shape, dtype):
        """-> NumPy array over pinned host memory (hipHostMalloc): H2D copies from it are asynchronous.  Freed with the
        engine (close()) or by host_free(array)."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        check(lib.ssdr_host_alloc(self._ctx, n, C.byref(p)), "ssdr_host_alloc")
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).view(dtype).reshape(shape)
        self._pinned.append(p.value)
        return arr

    def host_free_all(self):
        for p in self._pinned:
            lib.ssdr_host_free(self._ctx, C.c_void_p(p))
        self._pinned = []

    def feed_collect(self):
        """Oldest submitted batch -> (wf int16 [lines, n_ch, 1024], pcm int16 [n_ch, n_frames*512], rssi float32
        [n_ch, n_frames]) as views of pinned memory, valid until that slot is handed out again; in wire mode a fourth
        item: the SND headers' rssi float32 [n_ch, n_frames]."""
        wf, pcm, rssi, wr, lines = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint32()
        fl, navg = C.c_void_p(), C.c_uint32()
        check(lib.ssdr_feed_collect(self._ctx, C.byref(wf), C.byref(lines), C.byref(pcm), C.byref(rssi), C.byref(wr),
                                    C.byref(fl), C.byref(navg)), "ssdr_feed_collect")
        nf, nl = self._feed_frames, int(lines.value)
        self._feed_lines = nl
        self._feed_last_n_post = self._feed_n_post.pop(0) if self._feed_n_post else self.n_post
        rows = self.n_ch                           # rows of the arrays: every channel, or (lazy_out) the channels selected at the batch's submit
        if getattr(self, "_feed_lazy", False):
            n_sel = C.c_uint32()
            check(lib.ssdr_feed_collect_lazy(self._ctx, C.byref(n_sel), None, None, None, None), "ssdr_feed_collect_lazy")
            rows = int(n_sel.value)
        self.feed_rows = rows

        def view(ptr, ctype, dtype, shape):
            n = int(np.prod(shape))
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).reshape(shape) if n else np.zeros(shape, dtype)

        self.feed_flags = view(fl, C.c_uint8, np.uint8, (rows, nf))
        self.feed_n_avg = int(navg.value)          # the N in force when this batch was submitted
        w = view(wf, C.c_int16, np.int16, (nl, rows, L.NFFT))
        p = view(pcm, C.c_int16, np.int16, (rows, nf * L.FRAME))
        r = view(rssi, C.c_float, np.float32, (rows, nf))
        if self._feed_wire:
            return w, p, r, view(wr, C.c_float, np.float32, (rows, nf))
        return w, p, r

    def feed_device(self):
        """device pointers (ints) of EVERY channel's results of the batch feed_collect returned last -> dict(wf, pcm, rssi, flags, rows):
        wf int16 [lines][n_ch][1024], pcm int16 [n_ch][n_frames*512], rssi float32 / flags uint8 [n_ch][n_frames]; valid until
        depth - 1 further batches have been submitted.  For device-side consumers of a lazy_out feed."""
        n_sel, a, b, c, d = C.c_uint32(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib.ssdr_feed_collect_lazy(self._ctx, C.byref(n_sel), C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "ssdr_feed_collect_lazy")
        return {"wf": a.value, "pcm": b.value, "rssi": c.value, "flags": d.value, "rows": int(n_sel.value), "lines": self._feed_lines}

    def feed_close(self):
        check(lib.ssdr_feed_close(self._ctx), "ssdr_feed_close")

    def set_wfdata_rows(self, rows):
        """Keep the `rows` newest rows of kiwi_waterfall.wf_data on the device (fed by run_db2col); 0 = off."""
        check(lib.ssdr_set_wfdata_rows(self._ctx, int(rows)), "ssdr_set_wfdata_rows")

    def push_color_line(self, color):
        """float32 [lines, n_ch, 1024] colour lines from elsewhere than run_db2col into the device copy of wf_data."""
        color = np.ascontiguousarray(color, np.float32)
        assert color.ndim == 3 and color.shape[1:] == (self.n_post, L.NFFT)
        check(lib.ssdr_push_color_lines(self._ctx, color.ctypes.data, color.shape[0], 0), "ssdr_push_color_lines")

    def white_flag(self, first=0, count=None):
        """kiwi_waterfall.set_white_flag (utils_supersdr.py:875-877) on the device copy of wf_data."""
        check(lib.ssdr_wfdata_white_flag(self._ctx, int(first), self.n_post - first if count is None else int(count)),
              "ssdr_wfdata_white_flag")

    def run_trace(self, t_avg=15, spectrum_height=0, want_y=True):
        """plot_spectrum's reduction (utils_supersdr.py:1678-1679) -> (float64 [n_ch, 1024] nanmean over the t_avg newest
        wf_data rows, int32 [n_ch, 1024] pixel rows or None)."""
        trace = np.empty((self.n_post, L.NFFT), np.float64)
        y = np.empty((self.n_post, L.NFFT), np.int32) if want_y else None
        check(lib.ssdr_run_trace(self._ctx, int(t_avg), int(spectrum_height), trace.ctypes.data,
                                 y.ctypes.data if want_y else None, 0), "ssdr_run_trace")
        return trace, y

    def run_smeter(self, chans, fps, rssi=None):
        """One display frame of the S-meter smoothing (supersdr.py:936-947) per channel; chans (list of SmeterChan, or a ctypes
        array SmeterChan * n_ch: no per-channel objects) is updated in place.  rssi float64 [n_ch] or None (= last frame of the
        last run_audio)."""
        arr = chans if isinstance(chans, C.Array) else (SmeterChan * self.n_ch)(*chans)
        r = None if rssi is None else np.ascontiguousarray(rssi, np.float64)
        check(lib.ssdr_run_smeter(self._ctx, arr, None if r is None else r.ctypes.data, float(fps)), "ssdr_run_smeter")
        if arr is not chans:
            for i in range(self.n_ch):
                chans[i] = arr[i]
        return chans

    def set_kiwi_rate(self, kiwi_rate):
        """kiwi_sound.KIWI_RATE (utils_supersdr.py:991-994): 12000, or 20250 -- the rate of the IQ the channels receive (their
        constants are recompiled for it, the streams reset) and of play_buffer, which then takes its resample_poly branch."""
        check(lib.ssdr_set_kiwi_rate(self._ctx, int(kiwi_rate)), "ssdr_set_kiwi_rate")
        self.kiwi_rate = int(kiwi_rate)

    def playbuffer_frame_len(self):
        n = C.c_uint32()
        check(lib.ssdr_playbuffer_frame_len(self._ctx, C.byref(n)), "ssdr_playbuffer_frame_len")
        return int(n.value)

    def run_playbuffer(self, chans, fetch=True):
        """play_buffer for the frames of the last run_audio -> int16 [n_ch, n_frames*L, 2], L = playbuffer_frame_len()
        (2048 at 12 kHz, 1213 at 20.25 kHz).  chans: list of PlayChan or a ctypes array (PlayChan * n_ch)."""
        arr = chans if isinstance(chans, C.Array) else (PlayChan * max(self.n_post, 1))(*chans)
        out = np.empty((self.n_post, self.audio_frames * self.playbuffer_frame_len(), 2), np.int16) if fetch else None
        check(lib.ssdr_run_playbuffer(self._ctx, arr, out.ctypes.data if fetch and self.n_post else None, 0), "ssdr_run_playbuffer")
        return out

    def set_recording(self, on):
        """audio_rec.recording_flag: run_playbuffer also keeps the mono block play_buffer appends to audio_rec.audio_buffer"""
        check(lib.ssdr_set_recording(self._ctx, int(bool(on))), "ssdr_set_recording")

    def playbuffer_mono(self):
        """-> int16 [n_ch, n_frames*L]: pyaudio_buffer.astype(np.int16) of the last run_playbuffer (utils_supersdr.py:1139-1140)"""
        out = np.empty((self.n_post, self.audio_frames * self.playbuffer_frame_len()), np.int16)
        check(lib.ssdr_playbuffer_mono(self._ctx, out.ctypes.data, 0), "ssdr_playbuffer_mono")
        return out

    def adpcm_decode(self, data, state=None):
        """data uint8 [n_streams, n_bytes]; state int32 [n_streams, 2] {index, prev} (updated in place)
        -> int16 [n_streams, 2*n_bytes].  IMA ADPCM of compressed SND / W-F payloads (kiwi/client.py:58-87)."""
        data = np.ascontiguousarray(data, np.uint8)
        if state is None:
            state = np.zeros((data.shape[0], 2), np.int32)
        assert state.dtype == np.int32 and state.flags.c_contiguous
        out = np.empty((data.shape[0], 2 * data.shape[1]), np.int16)
        check(lib.ssdr_adpcm_decode(self._ctx, data.ctypes.data, data.shape[0], data.shape[1], state.ctypes.data,
                                    out.ctypes.data), "ssdr_adpcm_decode")
        return out

    def set_wf_lines(self, wf_sum):
        """int16 [lines, n_ch, 1024]: stand in for the output of run_wf (golden-vector tests of run_db2col)."""
        wf_sum = np.ascontiguousarray(wf_sum, np.int16)
        check(lib.ssdr_set_wf_lines(self._ctx, wf_sum.ctypes.data, wf_sum.shape[0]), "ssdr_set_wf_lines")

    def set_pcm(self, pcm):
        """int16 [n_ch, n_frames*512]: stand in for the output of run_audio (golden-vector tests of run_playbuffer)."""
        pcm = np.ascontiguousarray(pcm, np.int16).reshape(self.n_ch, -1)
        self.audio_frames = pcm.shape[1] // L.FRAME
        check(lib.ssdr_set_pcm(self._ctx, pcm.ctypes.data, self.audio_frames), "ssdr_set_pcm")

    def push_iq_wire(self, bodies):
        """bodies: uint8 [n_ch, n_frames, 2065] SND bodies in IQ mode (kiwi/client.py:384-389, 443-454).
        -> float32 [n_ch, n_frames] rssi from the frame headers."""
        bodies = np.ascontiguousarray(bodies, np.uint8)
        if bodies.ndim != 3 or bodies.shape[0] != self.n_ch or bodies.shape[2] != L.WIRE_BODY:
            raise ValueError("bodies must be uint8[n_ch=%d, n_frames, %d]" % (self.n_ch, L.WIRE_BODY))
        self.in_frames = bodies.shape[1]
        rssi = np.empty((self.n_ch, self.in_frames), np.float32)
        check(lib.ssdr_push_iq_wire(self._ctx, bodies.ctypes.data, self.in_frames, rssi.ctypes.data), "ssdr_push_iq_wire")
        return rssi

    def wire_gps(self):
        """-> uint32 [n_ch, n_frames, 4]: last_gps_solution, dummy, gpssec, gpsnsec of every frame of the last push_iq_wire
        (the `gps` dict of kiwi/client.py:444-445)"""
        out = np.empty((self.n_ch, self.in_frames, 4), np.uint32)
        check(lib.ssdr_wire_gps(self._ctx, out.ctypes.data), "ssdr_wire_gps")
        return out

    # ---- measurement
    def set_profiling(self, on):
        check(lib.ssdr_set_profiling(self._ctx, int(bool(on))), "ssdr_set_profiling")

    def set_concurrent(self, on):
        check(lib.ssdr_set_concurrent(self._ctx, int(on)), "ssdr_set_concurrent")

    def kernel_stats(self, which, reset=False):
        ms, n = C.c_float(0), C.c_uint32(0)
        check(lib.ssdr_kernel_stats(self._ctx, int(which), C.byref(ms), C.byref(n), int(reset)), "ssdr_kernel_stats")
        return ms.value, n.value

    def audio_paths(self):
        """-> (general, shift, AM-shift) channel counts of the audio stage's frame paths (one kernel each)"""
        n = (C.c_uint32 * 3)()
        check(lib.ssdr_audio_paths(self._ctx, C.byref(n)), "ssdr_audio_paths")
        return tuple(int(x) for x in n)

    def elapsed_ms(self):
        ms = C.c_float(0)
        check(lib.ssdr_elapsed_ms(self._ctx, C.byref(ms)), "ssdr_elapsed_ms")
        return ms.value

    def set_stream(self, hip_stream):
        check(lib.ssdr_set_stream(self._ctx, hip_stream), "ssdr_set_stream")

    # ---- introspection (tests, checkpointing)
    def get_consts(self, first=0, count=None):
        count = self.n_ch - first if count is None else int(count)
        k = np.empty(count, CONSTS_DTYPE)
        taps = np.empty((count, L.NTAP_MAX), np.float32)
        check(lib.ssdr_get_consts(self._ctx, int(first), count, k.ctypes.data, taps.ctypes.data), "ssdr_get_consts")
        return k, taps

    def get_state(self, first=0, count=None):
        count = self.n_ch - first if count is None else int(count)
        st = np.empty(count, STATE_DTYPE)
        hist = np.empty((count, L.HIST, 2), np.int16)
        check(lib.ssdr_get_state(self._ctx, int(first), count, st.ctypes.data, hist.ctypes.data), "ssdr_get_state")
        return st, hist

    def set_state(self, first, state, hist):
        state = np.ascontiguousarray(state, STATE_DTYPE)
        hist = np.ascontiguousarray(hist, np.int16)
        check(lib.ssdr_set_state(self._ctx, int(first), len(state), state.ctypes.data, hist.ctypes.data), "ssdr_set_state")

    def checkpoint(self):
        """-> bytes: everything the streams carry between calls (ssdr_checkpoint_save)"""
        n = C.c_uint64(0)
        check(lib.ssdr_checkpoint_size(self._ctx, C.byref(n)), "ssdr_checkpoint_size")
        buf = (C.c_char * n.value)()
        check(lib.ssdr_checkpoint_save(self._ctx, buf), "ssdr_checkpoint_save")
        return bytes(buf)

    def restore(self, blob):
        """Load a checkpoint() blob; the blob's own hop, decimation and averaging N become the ctx's (and this object's)."""
        n = C.c_uint64(0)
        check(lib.ssdr_checkpoint_size(self._ctx, C.byref(n)), "ssdr_checkpoint_size")
        if len(blob) != n.value:
            raise ValueError("checkpoint blob has %d bytes, this ctx (%d channels) takes %d" % (len(blob), self.n_ch, n.value))
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        check(lib.ssdr_checkpoint_load(self._ctx, buf, len(blob)), "ssdr_checkpoint_load")
        self._refresh_config()
        self.in_frames = 0                 # a batch pushed before the load belongs to the old streams

    def _refresh_config(self):
        hop, decim, navg, rate = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib.ssdr_get_config(self._ctx, C.byref(hop), C.byref(decim), C.byref(navg), C.byref(rate)), "ssdr_get_config")
        self.hop, self.decim, self.averaging, self.kiwi_rate = hop.value, decim.value, navg.value, rate.value

    def output_checksum(self):
        """-> (wf, pcm, rssi) 64-bit position-weighted checksums of the device-resident results of the last run"""
        v = (C.c_uint64 * 3)()
        check(lib.ssdr_output_checksum(self._ctx, C.byref(v)), "ssdr_output_checksum")
        return tuple(int(x) for x in v)

    def db2col_line(self, wf_sum, n_avg, chan):
        """spectrum_db2col of one int16[1024] line (a sum of n_avg byte lines) with display state `chan` (Db2colChan,
        updated in place) -> float32[1024]; leaves the batch results and the device copy of wf_data alone."""
        wf_sum = np.ascontiguousarray(wf_sum, np.int16)
        assert wf_sum.shape == (L.NFFT,)
        out = np.empty(L.NFFT, np.float32)
        check(lib.ssdr_db2col_line(self._ctx, wf_sum.ctypes.data, int(n_avg), C.byref(chan), out.ctypes.data), "ssdr_db2col_line")
        return out

    def selftest_sqrt(self):
        n = C.c_uint64(0)
        check(lib.ssdr_selftest_sqrt(self._ctx, C.byref(n)), "ssdr_selftest_sqrt")
        return n.value

    def sqrt_values(self, x):
        """float32[n] -> (ssdr_sqrt_rn(x), ssdr_sqrt_rn_int(x)) as the device computes them"""
        x = np.ascontiguousarray(x, np.float32)
        a, b = np.empty_like(x), np.empty_like(x)
        check(lib.ssdr_selftest_sqrt_values(self._ctx, x.ctypes.data, a.ctypes.data, b.ctypes.data, len(x)), "ssdr_selftest_sqrt_values")
        return a, b

    def selftest_quantiser(self):
        n = C.c_uint64(0)
        check(lib.ssdr_selftest_quantiser(self._ctx, C.byref(n)), "ssdr_selftest_quantiser")
        return n.value
