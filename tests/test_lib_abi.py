"""The C-ABI library on a machine WITHOUT a GPU: it loads, exports every symbol that
include/ssdr.h declares, its host-only entry points work, and device entry points fail
with an error code (never a crash, never a CPU fallback)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ssdr_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def S():
    import supersdr_amd
    return supersdr_amd


def declared_functions():
    src = open(os.path.join(ROOT, "include", "ssdr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ssdr_[a-z0-9_]+)\s*\(", src)))


def test_header_and_library_agree(S):
    from supersdr_amd import _lib as L
    names = declared_functions()
    assert len(names) >= 27
    raw = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libssdr.so does not export %s" % n
    assert sorted(L.EXPORTS) == names          # the ctypes binding covers exactly the header


def test_struct_layouts(S):
    from supersdr_amd import _lib as L
    assert C.sizeof(L.ChanParams) == 88 and C.sizeof(L.ChanConsts) == 64 and C.sizeof(L.ChanState) == 64
    assert L.ChanParams.f_shift_hz.offset == 16 and L.ChanParams.smeter_cal_db.offset == 80


def test_host_only_entry_points(S):
    from supersdr_amd import _lib as L
    assert b"gfx950" in L.lib.ssdr_version()
    assert L.lib.ssdr_strerror(L.ENODEV) == b"no such GPU device"
    assert L.lib.ssdr_strerror(-99) == b"unknown error"
    p = S.default_params("cw")
    assert (p.low_cut, p.high_cut, p.agc_decay, p.agc_thresh, p.agc_man_gain) == (400.0, 800.0, 1000.0, -80.0, 50.0)
    p = S.default_params("lsb")
    assert (p.low_cut, p.high_cut, p.agc_decay, p.smeter_cal_db) == (-3000.0, -30.0, 4000.0, -13.0)
    bad = L.ChanParams()
    assert L.lib.ssdr_default_params(9, C.byref(bad)) == L.EINVAL
    # the waterfall calibration must stay within +-200 dB (the quantiser scales its factor by 2^-48), here and in the oracle
    for db, ok in ((200.0, True), (-200.0, True), (200.5, False), (-1e9, False)):
        p = S.default_params("am", wf_cal_db=db)
        if ok:
            assert S.compile_params(p)[0]["wf_cal_lin"] == np.float32(10.0 ** (db / 10.0))
            O.compile_params(O.ChanParams(mode="am", wf_cal_db=db))
        else:
            with pytest.raises(S.SsdrError):
                S.compile_params(p)
            with pytest.raises(ValueError):
                O.compile_params(O.ChanParams(mode="am", wf_cal_db=db))


def test_tables_equal_oracle(S):
    assert np.array_equal(S.table(0), O.hann_window())
    wr, wi = O.twiddles()
    assert np.array_equal(S.table(1), wr) and np.array_equal(S.table(2), wi)
    assert np.array_equal(S.table(3), O.db_thresholds())
    assert S.table(1)[0] == 1.0 and S.table(2)[256] == -1.0 and S.table(1)[256] == 0.0


@pytest.mark.parametrize("mode,over", [("am", {}), ("usb", {}), ("lsb", dict(f_shift_hz=-1234.5)), ("cw", {}),
                                       ("nbfm", {}), ("usb", dict(agc_on=0, agc_man_gain=61.0)),
                                       ("am", dict(agc_hang=1, agc_decay=2600.0, agc_slope=7.0, agc_thresh=-95.0)),
                                       ("usb", dict(low_cut=300.0, high_cut=2700.0, wf_cal_db=-3.5, f_shift_hz=5999.9))])
def test_param_compile_equals_oracle(S, mode, over):
    """ssdr_compile_params (library host code) == ssdr_oracle.compile_params, field by field;
    FIR taps use the reference's design formula (utils_supersdr.py:334-344)"""
    p = S.default_params(mode, **over)
    k, taps = S.compile_params(p)
    op = O.ChanParams(mode=mode, f_shift_hz=p.f_shift_hz, low_cut=p.low_cut, high_cut=p.high_cut, agc_on=p.agc_on,
                      hang=p.agc_hang, thresh=p.agc_thresh, slope=p.agc_slope, decay=p.agc_decay,
                      man_gain=p.agc_man_gain, wf_cal_db=p.wf_cal_db, smeter_cal_db=p.smeter_cal_db)
    ok = O.compile_params(op)
    for f in ("mode", "ntap", "dphi1", "dphi2", "hang_frames", "tap_groups", "fir_flags"):
        assert int(k[f]) == int(ok[f]), f
    assert int(k["ntap8"]) == (int(ok["ntap"]) + 7) // 8 * 8
    for f in ("wf_cal_lin", "smeter_cal_db", "agc_c0", "agc_c1", "agc_knee", "agc_delta8"):
        assert np.float32(k[f]) == np.float32(ok[f]), f
    assert np.array_equal(taps, ok["taps"])


def test_param_compile_equals_oracle_on_random_parameters(S):
    """400 seeded random parameter sets: the library's host code and the oracle compile to the same constants and the
    same float32 taps (two libms, one rounding to float32)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random_params as RP
    rng = np.random.default_rng(5)
    for i in range(400):
        kw = RP.draw(rng)
        p = S.default_params(kw["mode"], f_shift_hz=kw["f_shift_hz"], low_cut=kw["low_cut"], high_cut=kw["high_cut"],
                             agc_on=kw["agc_on"], agc_hang=kw["hang"], agc_thresh=kw["thresh"], agc_slope=kw["slope"],
                             agc_decay=kw["decay"], agc_man_gain=kw["man_gain"], wf_cal_db=kw["wf_cal_db"],
                             smeter_cal_db=kw["smeter_cal_db"])
        k, taps = S.compile_params(p)
        ok = O.compile_params(O.ChanParams(**kw))
        for f in ("mode", "ntap", "dphi1", "dphi2", "hang_frames", "tap_groups", "fir_flags"):
            assert int(k[f]) == int(ok[f]), (i, f, kw)
        for f in ("wf_cal_lin", "smeter_cal_db", "agc_c0", "agc_c1", "agc_knee", "agc_delta8"):
            assert np.float32(k[f]) == np.float32(ok[f]), (i, f, kw)
        assert np.array_equal(taps, ok["taps"]), (i, kw, np.nonzero(taps != ok["taps"])[0][:5])


@pytest.mark.parametrize("rate,decim", [(12000, 1), (20250, 1), (20250, 2), (20250, 4)])
def test_param_compile_over_rates_and_the_iq_mode_equals_oracle(S, rate, decim):
    """The parameter surface grew in round 3 (the review: "keep that test honest when parameters grow"): the IQ rate
    (12 / 20.25 kHz, ssdr_set_kiwi_rate) and "SET mod=iq".  The twin takes its constants from the product, so the product's
    host compile is held to the oracle's here on 200 random sets per case, every field incl. the NBFM scale `kfm`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random_params as RP
    rng = np.random.default_rng(rate + decim)
    for i in range(200):
        kw = RP.draw(rng)
        if i % 5 == 0:
            kw.update(mode="iq", low_cut=-float(rng.choice([5000, 6000, 2500])), high_cut=float(rng.choice([5000, 6000, 2500])))
        kw["f_shift_hz"] *= decim * rate / 12000.0
        p = S.default_params(kw["mode"], f_shift_hz=kw["f_shift_hz"], low_cut=kw["low_cut"], high_cut=kw["high_cut"],
                             agc_on=kw["agc_on"], agc_hang=kw["hang"], agc_thresh=kw["thresh"], agc_slope=kw["slope"],
                             agc_decay=kw["decay"], agc_man_gain=kw["man_gain"], wf_cal_db=kw["wf_cal_db"],
                             smeter_cal_db=kw["smeter_cal_db"])
        k, taps = S.compile_params(p, decim, rate)
        ok = O.compile_params(O.ChanParams(**kw), decim, rate)
        for f in ("mode", "ntap", "ntap8", "dphi1", "dphi2", "hang_frames", "tap_groups", "fir_flags"):
            assert int(k[f]) == int(ok[f]), (i, f, kw)
        for f in ("wf_cal_lin", "smeter_cal_db", "agc_c0", "agc_c1", "agc_knee", "agc_delta8", "kfm"):
            assert np.float32(k[f]) == np.float32(ok[f]), (i, f, kw)
        assert np.array_equal(taps, ok["taps_streams"] if decim > 1 else ok["taps"]), (i, kw)
        if kw["mode"] == "iq":
            assert int(k["mode"]) == 5 and int(k["fir_flags"]) == 0       # never a lane-shift path: the filter output itself is the product
    with pytest.raises(S.SsdrError):
        S.compile_params(S.default_params("am"), 1, 16000)                 # rates a KiwiSDR does not have


def test_device_entry_points_fail_cleanly_without_gpu(S):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the -m gpu tests")
    from supersdr_amd import _lib as L
    ctx = L._P()
    assert L.lib.ssdr_create(0, 4, 1024, 512, C.byref(ctx)) == L.ENODEV
    assert not ctx
    assert L.lib.ssdr_create(0, 4, 512, 512, C.byref(ctx)) == L.EINVAL
    with pytest.raises(S.SsdrError):
        S.SsdrEngine(4)                                   # the product refuses to run: no CPU path
    assert L.lib.ssdr_sync(None) == L.EINVAL and L.lib.ssdr_run_wf(None, None, None, 0) == L.EINVAL
    L.lib.ssdr_destroy(None)


def test_missing_library_is_a_loud_import_error(tmp_path):
    import subprocess
    code = ("import os,sys; os.environ['SSDR_LIB_PATH']=%r; sys.path.insert(0,%r)\n"
            "try:\n import supersdr_amd\nexcept ImportError as e:\n print('IMPORTERROR', 'no CPU fallback' in str(e).lower() or 'missing' in str(e))\n"
            % (str(tmp_path / "nope.so"), ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "IMPORTERROR True" in out.stdout


def test_product_never_touches_the_oracle():
    """supersdr_amd/ must not import, load or link anything under oracle/ (or any CPU twin)"""
    bad = re.compile(r"oracle|ssdr_twin|twinlib|ssdr_oracle")
    for dp, _, files in os.walk(os.path.join(ROOT, "supersdr_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f)).read()
                assert not bad.search(txt), os.path.join(dp, f)


@pytest.mark.parametrize("decim", [2, 4])
def test_param_compile_for_a_decimating_front_end_equals_oracle(S, decim):
    """ssdr_set_decimation(D): the reference's tap formula at D * 12 kHz (capped at 127 / 125 taps), NCO step at the input
    rate, taps laid out as D polyphase streams -- the library's host code and the oracle agree on 150 random sets, and
    the stream layout is a permutation of the filter (stream q >= 1 behind one zero tap)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random_params as RP
    rng = np.random.default_rng(50 + decim)
    for i in range(150):
        kw = RP.draw(rng)
        kw["f_shift_hz"] *= decim                                   # the band is D times as wide
        p = S.default_params(kw["mode"], f_shift_hz=kw["f_shift_hz"], low_cut=kw["low_cut"], high_cut=kw["high_cut"],
                             agc_on=kw["agc_on"], agc_hang=kw["hang"], agc_thresh=kw["thresh"], agc_slope=kw["slope"],
                             agc_decay=kw["decay"], agc_man_gain=kw["man_gain"], wf_cal_db=kw["wf_cal_db"],
                             smeter_cal_db=kw["smeter_cal_db"])
        k, taps = S.compile_params(p, decim)
        ok = O.compile_params(O.ChanParams(**kw), decim)
        for f in ("mode", "ntap", "ntap8", "dphi1", "dphi2", "hang_frames", "tap_groups", "fir_flags", "decim"):
            assert int(k[f]) == int(ok[f]), (i, f, kw)
        assert np.array_equal(taps, ok["taps_streams"]), (i, kw)
        h, slots = ok["taps"][: ok["ntap"]], 128 // decim
        back = np.zeros(int(ok["ntap"]), np.float32)
        for kk in range(int(ok["ntap"])):
            pph, ii = kk % decim, kk // decim
            q, pos = (decim - pph, ii + 1) if pph else (0, ii)
            back[kk] = taps[q * slots + pos]
        assert np.array_equal(back, h) and np.count_nonzero(taps) == np.count_nonzero(h)
        assert int(ok["ntap"]) <= (125 if decim == 4 else 127) and int(k["ntap8"]) <= slots
    with pytest.raises(S.SsdrError):
        S.compile_params(S.default_params("am"), 3)
    assert S.compile_params(S.default_params("usb", f_shift_hz=11000.0), 2)[0]["decim"] == 2      # inside +-12 kHz
    with pytest.raises(S.SsdrError):
        S.compile_params(S.default_params("usb", f_shift_hz=12000.5), 2)


# ---- include/ssdr.h:11-14 "never throws or aborts" (the reference's own policy: errors become a flag, utils_supersdr.py:1031-1036) ----

def test_every_int_entry_point_is_a_function_try_block():
    """every `int ssdr_*` definition inside csrc/ssdr_api.cpp's extern "C" block carries SSDR_GUARD ... SSDR_UNGUARD
    (catch std::bad_alloc -> SSDR_ENOMEM, anything else -> SSDR_EHIP), and the header's functions are all defined there"""
    src = open(os.path.join(ROOT, "supersdr_amd", "csrc", "ssdr_api.cpp")).read()
    body = src[src.index('extern "C" {'):]
    defs = re.findall(r"^int (ssdr_\w+)\([^;{]*?\)( SSDR_GUARD)?\n\{", body, flags=re.M)
    assert len(defs) >= 70
    unguarded = [n for n, g in defs if not g]
    assert not unguarded, unguarded
    assert body.count("} SSDR_UNGUARD") == len(defs)
    returning_int = set(n for n, _ in defs)
    hdr = open(os.path.join(ROOT, "include", "ssdr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared_int = set(re.findall(r"^int (ssdr_\w+)\s*\(", hdr, flags=re.M))
    assert declared_int == returning_int, declared_int ^ returning_int


def build_fault_harness(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_fault_harness")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-rdynamic", "-o", exe,
                           os.path.join(ROOT, "tests", "abi_fault_harness.cpp"), "-ldl"])
    return exe


def test_failing_host_allocations_come_back_as_codes_cpu(tmp_path):
    """tests/abi_fault_harness.cpp: a process whose operator new fails for every allocation made from libssdr.so still gets
    return codes from the entry points that need no GPU (and SSDR_ENODEV / SSDR_ENOMEM, no half-built ctx, from ssdr_create)"""
    import subprocess
    from supersdr_amd import _lib as L
    out = subprocess.run([build_fault_harness(tmp_path), L.LIB_PATH, "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_failing_host_allocations_come_back_as_codes_gpu(tmp_path):
    """the fault-injection sweep on a live ctx: in ssdr_create, ssdr_set_params, ssdr_reset_state, the run calls (channel-list
    rebuild), get / set state, checkpoint save / load, the post-processing selection and kernels, the feed, zoom, exact bins, rate
    and decimation changes, the 1st, 2nd, 3rd ... host allocation fails in turn: every time the call RETURNS SSDR_ENOMEM (a
    std::bad_alloc crossing the C boundary would be SIGABRT), the ctx stays usable and the same call then succeeds"""
    import subprocess
    from supersdr_amd import _lib as L
    out = subprocess.run([build_fault_harness(tmp_path), L.LIB_PATH, "gpu"], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr
    assert "ssdr_checkpoint_load" in out.stdout and "ssdr_create" in out.stdout
