"""ctypes binding of libssdr.so (include/ssdr.h).

The HIP library is the product: if it is missing this module raises ImportError
loudly -- there is no CPU or NumPy fallback anywhere in supersdr_amd.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSDR_LIB_PATH") or os.path.join(_HERE, "libssdr.so")   # override: A/B builds only

NFFT, FRAME, RATE, NTAP_MAX, HIST = 1024, 512, 12000, 128, 128
OK, EINVAL, ENOMEM, EHIP, ENODEV, ESTATE = 0, -1, -2, -3, -4, -5
MODE_AM, MODE_LSB, MODE_USB, MODE_CW, MODE_NBFM, MODE_IQ = range(6)
MODE_BY_NAME = {"am": 0, "lsb": 1, "usb": 2, "cw": 3, "nbfm": 4, "nfm": 4, "iq": 5}
K_WF, K_AUDIO, K_SYNTH, K_DB2COL, K_PLAY, K_WIRE, K_TRACE, K_SMETER, K_FUSED, K_ZOOM = range(10)
T_WINDOW, T_TWIDDLE_RE, T_TWIDDLE_IM, T_DB_THRESH = range(4)


class ChanParams(C.Structure):
    """ssdr_chan_params: "SET mod=/low_cut=/high_cut=/freq=" and "SET agc=..." of
    utils_supersdr.py:1023,1028."""
    _fields_ = [("mode", C.c_int32), ("agc_on", C.c_int32), ("agc_hang", C.c_int32), ("reserved", C.c_int32),
                ("f_shift_hz", C.c_double), ("low_cut", C.c_double), ("high_cut", C.c_double),
                ("agc_thresh", C.c_double), ("agc_slope", C.c_double), ("agc_decay", C.c_double),
                ("agc_man_gain", C.c_double), ("wf_cal_db", C.c_double), ("smeter_cal_db", C.c_double)]


class ChanConsts(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("ntap8", C.c_uint32), ("dphi1", C.c_uint32), ("dphi2", C.c_uint32),
                ("wf_cal_lin", C.c_float), ("smeter_cal_db", C.c_float),
                ("agc_c0", C.c_float), ("agc_c1", C.c_float), ("agc_knee", C.c_float), ("agc_delta8", C.c_float),
                ("hang_frames", C.c_uint32), ("ntap", C.c_uint32), ("tap_groups", C.c_uint32), ("fir_flags", C.c_uint32),
                ("decim", C.c_uint32), ("kfm", C.c_float)]


class Db2colChan(C.Structure):
    """ssdr_db2col_chan: display state of kiwi_waterfall.spectrum_db2col (utils_supersdr.py:787-813)."""
    _fields_ = [("zoom", C.c_int32), ("auto_scale", C.c_int32), ("delta_low_db", C.c_int32), ("delta_high_db", C.c_int32),
                ("low_clip_db", C.c_float), ("high_clip_db", C.c_float), ("dynamic_range", C.c_float),
                ("wf_min_db", C.c_float), ("wf_max_db", C.c_float), ("pad", C.c_uint32 * 3)]


class PlayChan(C.Structure):
    """ssdr_play_chan: kiwi_sound.volume / audio_balance (utils_supersdr.py:921, 945)."""
    _fields_ = [("volume", C.c_double), ("balance", C.c_double)]


class SmeterChan(C.Structure):          # ssdr_smeter_chan, 112 B
    _fields_ = [("rssi_smooth", C.c_double), ("rssi_smooth_slow", C.c_double), ("hist", C.c_double * 10),
                ("hist_pos", C.c_uint32), ("run_index", C.c_uint32), ("decay_ms", C.c_double)]

    @classmethod
    def start(cls, rssi0, decay_ms=4000.0):
        """state before the first frame: rssi_hist = deque(10*[rssi0], 10), smooth = slow = rssi0 (supersdr.py:164-167)"""
        return cls(float(rssi0), float(rssi0), (C.c_double * 10)(*([float(rssi0)] * 10)), 0, 0, float(decay_ms))


class ChanState(C.Structure):
    _fields_ = [("phi1", C.c_uint32), ("phi2", C.c_uint32), ("dc", C.c_float), ("agc_d", C.c_float),
                ("agc_m", C.c_float * 8), ("prev_re", C.c_float), ("prev_im", C.c_float), ("pad", C.c_uint32 * 2)]


assert C.sizeof(ChanConsts) == 64 and C.sizeof(ChanState) == 64 and C.sizeof(ChanParams) == 88
assert C.sizeof(Db2colChan) == 48 and C.sizeof(PlayChan) == 16
WIRE_BODY = 17 + FRAME * 4
FEED_LAZY_MAX = 4096        # SSDR_FEED_LAZY_MAX (include/ssdr.h): rows a SSDR_FEED_LAZY_OUT feed copies back per batch

_P = C.c_void_p
_SIGS = {
    "ssdr_create": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "ssdr_destroy": (None, [_P]),
    "ssdr_set_params": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(ChanParams)]),
    "ssdr_default_params": (C.c_int, [C.c_int, C.POINTER(ChanParams)]),
    "ssdr_reset_state": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "ssdr_set_averaging": (C.c_int, [_P, C.c_uint32]),
    "ssdr_set_hop": (C.c_int, [_P, C.c_uint32]),
    "ssdr_set_exact_bins": (C.c_int, [_P, C.c_int]),
    "ssdr_set_wf_zoom": (C.c_int, [_P, C.c_uint32]),
    "ssdr_set_wf_center": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "ssdr_read_zoom": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint32)]),
    "ssdr_set_decimation": (C.c_int, [_P, C.c_uint32]),
    "ssdr_compile_params_decim": (C.c_int, [C.POINTER(ChanParams), C.c_uint32, C.POINTER(ChanConsts), _P]),
    "ssdr_compile_params_rate": (C.c_int, [C.POINTER(ChanParams), C.c_uint32, C.c_uint32, C.POINTER(ChanConsts), _P]),
    "ssdr_push_iq": (C.c_int, [_P, _P, C.c_uint32, C.c_int]),
    "ssdr_run_wf": (C.c_int, [_P, _P, C.POINTER(C.c_uint32), C.c_int]),
    "ssdr_run_audio": (C.c_int, [_P, _P, _P, C.c_int]),
    "ssdr_audio_flags": (C.c_int, [_P, _P, C.c_int]),
    "ssdr_audio_iq": (C.c_int, [_P, _P, C.c_int]),
    "ssdr_run_chain": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "ssdr_set_fused": (C.c_int, [_P, C.c_int]),
    "ssdr_set_chain_floors": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "ssdr_get_chain_floors": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ssdr_set_overlap": (C.c_int, [_P, C.c_int]),
    "ssdr_sync": (C.c_int, [_P]),
    "ssdr_set_post_channels": (C.c_int, [_P, _P, C.c_uint32]),
    "ssdr_run_db2col": (C.c_int, [_P, C.POINTER(Db2colChan), _P, C.c_int]),
    "ssdr_run_playbuffer": (C.c_int, [_P, C.POINTER(PlayChan), _P, C.c_int]),
    "ssdr_set_wfdata_rows": (C.c_int, [_P, C.c_uint32]),
    "ssdr_push_color_lines": (C.c_int, [_P, _P, C.c_uint32, C.c_int]),
    "ssdr_wfdata_white_flag": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "ssdr_run_trace": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P, C.c_int]),
    "ssdr_run_smeter": (C.c_int, [_P, C.POINTER(SmeterChan), _P, C.c_double]),
    "ssdr_set_kiwi_rate": (C.c_int, [_P, C.c_uint32]),
    "ssdr_playbuffer_frame_len": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "ssdr_set_recording": (C.c_int, [_P, C.c_int]),
    "ssdr_playbuffer_mono": (C.c_int, [_P, _P, C.c_int]),
    "ssdr_push_iq_wire": (C.c_int, [_P, _P, C.c_uint32, _P]),
    "ssdr_wire_gps": (C.c_int, [_P, _P]),
    "ssdr_adpcm_decode": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "ssdr_feed_open": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "ssdr_feed_slot": (C.c_int, [_P, C.POINTER(_P)]),
    "ssdr_feed_submit": (C.c_int, [_P]),
    "ssdr_feed_submit_from": (C.c_int, [_P, _P]),
    "ssdr_host_alloc": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "ssdr_host_free": (C.c_int, [_P, _P]),
    "ssdr_feed_collect": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint32), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                    C.POINTER(_P), C.POINTER(C.c_uint32)]),
    "ssdr_feed_post": (C.c_int, [_P, C.POINTER(Db2colChan), C.POINTER(PlayChan)]),
    "ssdr_feed_collect_post": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "ssdr_feed_close": (C.c_int, [_P]),
    "ssdr_copy_from_device": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "ssdr_wf_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint32)]),
    "ssdr_audio_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "ssdr_set_stream": (C.c_int, [_P, _P]),
    "ssdr_set_profiling": (C.c_int, [_P, C.c_int]),
    "ssdr_set_concurrent": (C.c_int, [_P, C.c_int]),
    "ssdr_kernel_stats": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_int]),
    "ssdr_audio_paths": (C.c_int, [_P, C.POINTER(C.c_uint32 * 3)]),
    "ssdr_elapsed_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "ssdr_synth_iq": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "ssdr_read_input": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "ssdr_table": (C.c_int, [C.c_int, _P, C.c_uint32]),
    "ssdr_compile_params": (C.c_int, [C.POINTER(ChanParams), C.POINTER(ChanConsts), _P]),
    "ssdr_get_consts": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P]),
    "ssdr_get_state": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P]),
    "ssdr_set_state": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P]),
    "ssdr_checkpoint_size": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ssdr_checkpoint_save": (C.c_int, [_P, _P]),
    "ssdr_checkpoint_load": (C.c_int, [_P, _P, C.c_uint64]),
    "ssdr_get_config": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ssdr_db2col_line": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(Db2colChan), _P]),
    "ssdr_feed_collect_lazy": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "ssdr_output_checksum": (C.c_int, [_P, C.POINTER(C.c_uint64 * 3)]),
    "ssdr_set_wf_lines": (C.c_int, [_P, _P, C.c_uint32]),
    "ssdr_set_pcm": (C.c_int, [_P, _P, C.c_uint32]),
    "ssdr_selftest_quantiser": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ssdr_selftest_sqrt": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ssdr_selftest_sqrt_values": (C.c_int, [_P, _P, _P, _P, C.c_uint32]),
    "ssdr_strerror": (C.c_char_p, [C.c_int]),
    "ssdr_last_hip_error": (C.c_char_p, []),
    "ssdr_version": (C.c_char_p, []),
}
EXPORTS = tuple(_SIGS)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "supersdr_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C supersdr_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class SsdrError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        detail = lib.ssdr_last_hip_error().decode() if code == EHIP else ""
        super().__init__("%s: %s (%d) %s" % (where, lib.ssdr_strerror(code).decode(), code, detail))


def check(code, where):
    if code != OK:
        raise SsdrError(code, where)
