"""ctypes wrapper of oracle/libssdr_twin.so (the fp32 CPU twin; test infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE, "libssdr_twin.so")

CONSTS_DTYPE = np.dtype([("mode", "<u4"), ("ntap8", "<u4"), ("dphi1", "<u4"), ("dphi2", "<u4"),
                         ("wf_cal_lin", "<f4"), ("smeter_cal_db", "<f4"), ("agc_c0", "<f4"), ("agc_c1", "<f4"),
                         ("agc_knee", "<f4"), ("agc_delta8", "<f4"), ("hang_frames", "<u4"), ("ntap", "<u4"),
                         ("tap_groups", "<u4"), ("fir_flags", "<u4"), ("decim", "<u4"), ("kfm", "<f4")])
STATE_DTYPE = np.dtype([("phi1", "<u4"), ("phi2", "<u4"), ("dc", "<f4"), ("agc_d", "<f4"), ("agc_m", "<f4", (8,)),
                        ("prev_re", "<f4"), ("prev_im", "<f4"), ("pad", "<u4", (2,))])


class Twin:
    def __init__(self, lib):
        self.lib = lib
        P = C.c_void_p
        lib.twin_make_tables.argtypes = [P, P, P, P]
        lib.twin_wf.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, P, P, P, P, P, P]
        lib.twin_wf_line.argtypes = [P, P, P, P, P, C.c_float, P]
        lib.twin_wf_lines.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, P, P, P, P, P, P]
        lib.twin_audio.argtypes = [P, C.c_uint32, C.c_uint32, P, P, P, P, P, P]
        lib.twin_audio2.argtypes = [P, C.c_uint32, C.c_uint32, P, P, P, P, P, P, P]
        lib.twin_audio3.argtypes = [P, C.c_uint32, C.c_uint32, P, P, P, P, P, P, P, P]
        lib.twin_zoom.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, P, P, C.c_uint32, P, P, P]
        lib.twin_quantise.argtypes = [C.c_float, P]
        lib.twin_quantise.restype = C.c_int
        for f in ("twin_log2p", "twin_exp2p"):
            getattr(lib, f).argtypes = [C.c_float]
            getattr(lib, f).restype = C.c_float
        lib.twin_atan2p.argtypes = [C.c_float, C.c_float]
        lib.twin_atan2p.restype = C.c_float
        lib.twin_sincos20.argtypes = [C.c_uint32, P, P]
        lib.twin_phasor32.argtypes = [C.c_uint32, P, P]
        self.win = np.empty(1024, np.float32)
        self.wr = np.empty(512, np.float32)
        self.wi = np.empty(512, np.float32)
        self.thr = np.empty(256, np.float32)
        lib.twin_make_tables(self.win.ctypes.data, self.wr.ctypes.data, self.wi.ctypes.data, self.thr.ctypes.data)

    def wf(self, iq, n_avg, cal_lin=None):
        """iq int16[n_ch, n_lines*1024, 2] -> int16[n_lines//n_avg, n_ch, 1024]"""
        iq = np.ascontiguousarray(iq, np.int16)
        n_ch, n_lines = iq.shape[0], iq.shape[1] // 1024
        cal = np.ones(n_ch, np.float32) if cal_lin is None else np.ascontiguousarray(cal_lin, np.float32)
        out = np.zeros((n_lines // n_avg, n_ch, 1024), np.int16)
        self.lib.twin_wf(iq.ctypes.data, n_ch, n_lines, n_avg, cal.ctypes.data, self.win.ctypes.data,
                         self.wr.ctypes.data, self.wi.ctypes.data, self.thr.ctypes.data, out.ctypes.data)
        return out

    def wf_hop(self, stream, hop, n_avg=1, cal_lin=None):
        """stream int16[n_ch, n, 2] (for hop 512: the carried 512-sample tail in front) -> int16[(lines // n_avg), n_ch, 1024]
        sums of n_avg consecutive overlapping lines; line k covers samples [k*hop, k*hop + 1024)"""
        stream = np.ascontiguousarray(stream, np.int16)
        n_ch = stream.shape[0]
        n_lines = (stream.shape[1] - 1024) // hop + 1
        stream = np.ascontiguousarray(stream[:, :(n_lines - 1) * hop + 1024])
        cal = np.ones(n_ch, np.float32) if cal_lin is None else np.ascontiguousarray(cal_lin, np.float32)
        b = np.zeros((n_lines, n_ch, 1024), np.uint8)
        self.lib.twin_wf_lines(stream.ctypes.data, n_ch, n_lines, hop, cal.ctypes.data, self.win.ctypes.data,
                               self.wr.ctypes.data, self.wi.ctypes.data, self.thr.ctypes.data, b.ctypes.data)
        L = n_lines // n_avg
        return b[: L * n_avg].astype(np.int16).reshape(L, n_avg, n_ch, 1024).sum(axis=1).astype(np.int16)

    def audio(self, iq, consts, taps, state, hist, want_flags=False, want_iq=False):
        """iq int16[n_ch, n_frames*512, 2]; state/hist updated in place -> (pcm, rssi[, adc-overflow flags uint8][, iq_out
        int16 [n_ch, n_frames*512, 2]: I,Q of the channels in mode 5, zero elsewhere])"""
        iq = np.ascontiguousarray(iq, np.int16)
        consts = np.ascontiguousarray(consts, CONSTS_DTYPE)
        n_ch, n_frames = iq.shape[0], iq.shape[1] // (512 * max(1, int(consts["decim"][0])))
        taps = np.ascontiguousarray(taps, np.float32)
        assert state.dtype == STATE_DTYPE and state.flags.c_contiguous and hist.flags.c_contiguous
        pcm = np.zeros((n_ch, n_frames * 512), np.int16)
        rssi = np.zeros((n_ch, n_frames), np.float32)
        flags = np.zeros((n_ch, n_frames), np.uint8)
        iqo = np.zeros((n_ch, n_frames * 512, 2), np.int16)
        self.lib.twin_audio3(iq.ctypes.data, n_ch, n_frames, consts.ctypes.data, taps.ctypes.data,
                             state.ctypes.data, hist.ctypes.data, pcm.ctypes.data, rssi.ctypes.data, flags.ctypes.data,
                             iqo.ctypes.data)
        out = (pcm, rssi) + ((flags,) if want_flags else ()) + ((iqo,) if want_iq else ())
        return out

    def zoom(self, iq, Z, dphi, taps, phase, hist):
        """iq int16[n_ch, n_in, 2]; dphi uint32[n_ch]; taps float32[ntap]; phase uint32[n_ch] and hist int16[n_ch, 256, 2]
        updated in place -> int16[n_ch, n_in // Z, 2] the zoomed stream"""
        iq = np.ascontiguousarray(iq, np.int16)
        n_ch, n_in = iq.shape[0], iq.shape[1]
        dphi = np.ascontiguousarray(dphi, np.uint32)
        taps = np.ascontiguousarray(taps, np.float32)
        assert phase.dtype == np.uint32 and hist.dtype == np.int16 and hist.shape == (n_ch, 256, 2) and n_in >= 256
        out = np.zeros((n_ch, n_in // Z, 2), np.int16)
        self.lib.twin_zoom(iq.ctypes.data, n_ch, n_in, Z, dphi.ctypes.data, taps.ctypes.data, len(taps), phase.ctypes.data,
                           hist.ctypes.data, out.ctypes.data)
        return out

    def phasor32(self, ph):
        c, s = C.c_float(), C.c_float()
        self.lib.twin_phasor32(int(ph), C.byref(c), C.byref(s))
        return c.value, s.value

    def sincos20(self, ph):
        c, s = C.c_float(), C.c_float()
        self.lib.twin_sincos20(int(ph), C.byref(c), C.byref(s))
        return c.value, s.value


def load():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(ORACLE, "ssdr_twin.c")):
        subprocess.check_call(["make", "-C", ORACLE], stdout=subprocess.DEVNULL)
    return Twin(C.CDLL(SO))


def fresh_state(consts):
    st = np.zeros(len(consts), STATE_DTYPE)
    st["agc_d"] = consts["agc_knee"]
    st["agc_m"] = -1000.0
    hist = np.zeros((len(consts), 128, 2), np.int16)
    return st, hist
