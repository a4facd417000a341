"""The oracle against golden vectors produced by the REAL reference (oracle/make_golden.py).

These pin every stage of the path that exists in the reference: FIR design and convolution,
time binning, spectrum_db2col, play_buffer, and the W/F / SND / IQ / ADPCM / wav formats.
CPU only; nothing here touches the product.
"""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ssdr_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def test_filtering_taps_bit_exact():
    """design_lowpass == filtering(fl, fs).h (utils_supersdr.py:334-344), float64 bit-for-bit"""
    g = gold("filtering.npz")
    keys = [k for k in g.files if k.startswith("h_")]
    assert len(keys) >= 8
    for k in keys:
        _, fl, fs = k.split("_")
        h = O.design_lowpass(float(fl), float(fs))
        assert len(h) == int(g["n_" + fl + "_" + fs])
        assert np.array_equal(h, g[k]), k
    # the probes of SURVEY.md 8a row a8
    assert int(g["n_6000_48000"]) == 33 and int(g["n_3000_12000"]) == 17 and int(g["n_500_12000"]) == 97
    assert np.array_equal(np.convolve(g["lowpass_in"], O.design_lowpass(6000, 48000), "valid"), g["lowpass_out"])


def test_time_binning_integer_sum_is_exact():
    """np.mean of N float32 byte lines (utils_supersdr.py:881-886) == float32(int sum)/float32(N), bit-for-bit:
    the int16 sums the GPU emits lose nothing."""
    g = gold("binning.npz")
    for n in (1, 2, 3, 7, 10, 33, 100):
        lines = g["lines_%d" % n]
        ref = g["mean_%d" % n]
        assert ref.dtype == np.float32
        s = lines.astype(np.int16).sum(axis=0, dtype=np.int16)
        assert lines.astype(np.int64).sum(axis=0).max() < 2 ** 15
        assert np.array_equal(O.wf_mean_from_sum(s, n), ref), n
        assert np.array_equal(O.wf_time_binning([l.astype(np.float32) for l in lines]), ref)


def test_spectrum_db2col_matches_reference():
    g = gold("db2col.npz")
    for i in range(int(g["count"])):
        zoom, auto, dlo, dhi = g["cfg_%d" % i]
        col, lo, hi, dyn, mn, mx = O.spectrum_db2col(g["in_%d" % i].copy(), int(zoom), bool(auto),
                                                     delta_low_db=int(dlo), delta_high_db=int(dhi))
        assert np.array_equal(col, g["color_%d" % i]), i
        assert np.allclose([lo, hi, dyn, mn, mx], g["scal_%d" % i], rtol=0, atol=0), i


def test_play_buffer_matches_reference():
    """4 consecutive frames per case: history carry, volume 150 (int16 wrap), pan -1/-0.5/0/+1"""
    g = gold("playbuffer.npz")
    assert int(g["n_tap"]) == 33
    for c in range(int(g["count"])):
        volume, balance = g["cfg_%d" % c]
        pb = O.PlayBuffer()
        frames = g["in_%d" % c]
        for f in range(frames.shape[0]):
            out = pb(frames[f], volume=volume, balance=balance)
            assert np.array_equal(out, g["out_%d" % c][f]), (c, f)
            assert np.array_equal(pb.rec, g["rec_%d" % c][f]), (c, f)          # the recording branch (:1139-1140)


def test_play_buffer_resampled_matches_reference():
    """20.25 kHz KiwiSDRs: resample_poly(popped, 64, 27, padtype="line")[:-1] per frame (utils_supersdr.py:1125-1126)"""
    g = gold("playbuffer.npz")
    assert list(g["rs_ratio"]) == [O.RS_UP, O.RS_DOWN]
    pb = O.PlayBufferResampled()
    for c in range(int(g["rs_count"])):
        volume, balance = g["rs_cfg_%d" % c]
        frames = g["rs_in_%d" % c]
        for f in range(frames.shape[0]):
            out = pb(frames[f], volume=volume, balance=balance)
            assert out.shape == (1213, 2)
            assert np.array_equal(out, g["rs_out_%d" % c][f]), (c, f)
            assert np.array_equal(pb.rec, g["rs_rec_%d" % c][f]), (c, f)


def test_resample_tap_table_is_scipys():
    """the committed tap table (csrc/ssdr_resample_taps.h, tools/gen_resample_taps.py) equals what scipy designs"""
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "supersdr_amd", "csrc", "ssdr_resample_taps.h")
    text = open(path).read()
    body = text[text.index("SSDR_RS_TAPS["):]
    vals = np.array([float.fromhex(v) for v in re.findall(r"-?0x[0-9a-f.]+p[+-]?\d+", body)])
    tf, n_pre_remove = O.resample_taps_64_27()
    assert np.array_equal(vals, tf)
    assert "#define SSDR_RS_PRE_REMOVE %d " % n_pre_remove in text and "#define SSDR_RS_HPP %d\n" % (len(tf) // 64) in text


def test_display_reductions_match_reference():
    """plot_spectrum's trace (real reference drawing into a recording PixelArray) and the S-meter lines of the main loop"""
    g = gold("display.npz")
    for c in range(int(g["count"])):
        height, spec_h, n_lines, t_avg = (int(v) for v in g["cfg_%d" % c])
        wd = O.WfData(height)
        for line in g["lines_%d" % c]:
            wd.push(line)
        v = O.spectrum_trace(wd.wf_data, t_avg)
        assert np.array_equal(v, g["trace_%d" % c]), c
        assert np.array_equal(O.trace_pixels(v, spec_h), g["y_%d" % c]), c
    for c in range(int(g["sm_count"])):
        decay, fps = g["sm_cfg_%d" % c]
        rssi = g["sm_in_%d" % c]
        sm = O.SMeter(float(rssi[0]))
        got = np.array([sm.step(float(r), decay, fps, i) for i, r in enumerate(rssi)])
        assert np.array_equal(got[:, 0], g["sm_smooth_%d" % c]) and np.array_equal(got[:, 1], g["sm_slow_%d" % c]), c


def test_frame_decoders_match_reference():
    g = gold("frames.npz")
    assert np.array_equal(O.decode_wf_frame(g["wf_msg"].tobytes()), g["wf_spectrum"])
    ovf, seq, rssi, samples = O.decode_snd_frame(g["snd_msg"].tobytes())
    assert np.array_equal(samples, g["snd_samples"]) and samples.dtype == np.int16
    assert rssi == float(g["snd_rssi"]) and int(ovf) == int(g["snd_adc_ovf"]) and seq == 77
    flags, seq, rssi, gps, cs = O.decode_iq_frame(g["iq_body"].tobytes())
    assert np.array_equal(cs, g["iq_complex64"]) and cs.dtype == np.complex64
    assert np.array_equal(cs.real.astype(np.int16), g["iq_int16"][:, 0])
    assert np.array_equal(cs.imag.astype(np.int16), g["iq_int16"][:, 1])
    assert rssi == float(g["iq_rssi"]) and seq == int(g["iq_seq"])
    assert [gps["last_gps_solution"], gps["dummy"], gps["gpssec"], gps["gpsnsec"]] == list(g["iq_gps"])


def test_ima_adpcm_matches_reference():
    g = gold("frames.npz")
    data = g["adpcm_in"].tobytes()
    a, idx, prev = O.ima_adpcm_decode(data[:256])
    b, _, _ = O.ima_adpcm_decode(data[256:], idx, prev)       # state persists across SND frames
    assert np.array_equal(np.concatenate([a, b]), g["adpcm_out"])
    # compressed W/F line: decoder reset per line, last 10 samples dropped (kiwi/client.py:476-479)
    body = g["wfc_body"].tobytes()
    s, _, _ = O.ima_adpcm_decode(body[12:])
    assert np.array_equal(s[:-10], g["wfc_samples"])


def test_kiwi_wav_decode_matches_reference():
    g = gold("wavreader.npz")
    assert int(g["ok"]) == 1
    blocks, stamps = O.decode_kiwi_wav(g["wav_bytes"].tobytes())
    assert np.array_equal(blocks, g["blocks"])
    # the reference returns blocks 2.. as complex64 / 65535 (kiwi/wavreader.py:84, 92-99)
    z = (blocks[2:].astype(np.float32).reshape(-1, 2).view(np.complex64).reshape(-1)) / 65535
    assert np.array_equal(z.astype(np.complex64), g["z"])
    assert len(stamps) == 5 and stamps[1][2] == 1000 and stamps[1][3] == 42666667
