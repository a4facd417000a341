#!/usr/bin/env python3
"""Replay a KiwiSDR IQ recording (kiwirecorder .wav, kiwi/wavreader.py format) through the GPU path, the way supersdr.py
would see a live receiver: IQ blocks -> IQHub (ssdr_push_iq, both kernels, db2col, play_buffer) -> kiwi_waterfall /
kiwi_sound -> a 48 kHz stereo .wav of what PortAudio would have played and the waterfall rows as .npy.

    python tools/replay_iq_wav.py rec.wav --mode usb --tune-khz 0.0 --audio out.wav --waterfall wf.npy [--averaging 3]
"""
import argparse
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


class _Disp:
    DISPLAY_WIDTH, WF_HEIGHT = 1024, 400


class _Eibi:
    def get_stations(self, a, b):
        pass


def write_wav(path, stereo_i16, rate=48000):
    data = np.ascontiguousarray(stereo_i16, "<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " +
                struct.pack("<IHHIIHH", 16, 1, 2, rate, rate * 4, 4, 16) + b"data" + struct.pack("<I", len(data)) + data)


def replay(wav, mode="AM", tune_khz=0.0, averaging=1, volume=100, zoom=10, center_khz=7100.0, device=0, probe=None):
    """-> (int16 [n, 2] 48 kHz stereo, float64 [rows, 1024] waterfall rows oldest first, float [frames] rssi)"""
    from supersdr_amd.iqstream import read_kiwi_iq_wav
    from supersdr_amd.workers import IQHub, bind, bind_headless
    try:                                     # next to supersdr.py: the maintainer's own classes; anywhere else: the bare pair
        import utils_supersdr
        gpu = bind(utils_supersdr)
    except Exception:
        gpu = bind_headless()
    kiwi_waterfall, kiwi_sound = gpu.kiwi_waterfall, gpu.kiwi_sound
    blocks, _ = read_kiwi_iq_wav(wav)
    hub = IQHub(1, device=device)
    wf = kiwi_waterfall("replay", 0, "", zoom, center_khz, _Eibi(), _Disp(), hub=hub, channel=0, timeout=0.05)
    lc, hc = {"AM": (-6000, 6000), "USB": (30, 3000), "LSB": (-3000, -30), "CW": (400, 800), "NBFM": (-6000, 6000)}[mode.upper()]
    snd = kiwi_sound(center_khz + tune_khz, mode.upper(), lc, hc, "", wf, 8, volume_=volume)
    wf.averaging_n = averaging
    hub.set_averaging(averaging)
    audio, rows, rssi = [], [], []
    for blk in blocks:
        hub.feed(0, blk)
        while hub.snd_queue[0].qsize():                   # what kiwi_snd.run / the PortAudio callback would do
            s = snd.process_audio_stream()
            rssi.append(snd.rssi)
            snd.audio_buffer.put(s)
            out = np.zeros((hub.play_len, 2), np.int16)
            snd.play_buffer(out, hub.play_len, None, None)
            audio.append(out)
        while hub.wf_queue[0].qsize():                    # what kiwi_wf.run would do
            wf.step()
            rows.append(np.asarray(wf.wf_color, np.float64).copy())
    if probe is not None:                                 # tests: the constants the channel ran with
        probe["consts"], probe["taps"] = hub.engine.get_consts()
    hub.close()
    return (np.concatenate(audio) if audio else np.zeros((0, 2), np.int16),
            np.stack(rows) if rows else np.zeros((0, 1024)), np.array(rssi))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("wav")
    ap.add_argument("--mode", default="am", choices=["am", "usb", "lsb", "cw", "nbfm"])
    ap.add_argument("--tune-khz", type=float, default=0.0, help="receiver frequency relative to the recording's centre")
    ap.add_argument("--averaging", type=int, default=1)
    ap.add_argument("--volume", type=int, default=100)
    ap.add_argument("--audio", default="replay_audio.wav")
    ap.add_argument("--waterfall", default="replay_wf.npy")
    a = ap.parse_args()
    audio, rows, rssi = replay(a.wav, a.mode, a.tune_khz, a.averaging, a.volume)
    write_wav(a.audio, audio)
    np.save(a.waterfall, rows)
    print("%d audio frames -> %s (%.2f s), %d waterfall rows -> %s, rssi %.1f .. %.1f dBm"
          % (len(rssi), a.audio, len(audio) / 48000.0, len(rows), a.waterfall, rssi.min() if len(rssi) else 0, rssi.max() if len(rssi) else 0))


if __name__ == "__main__":
    main()
