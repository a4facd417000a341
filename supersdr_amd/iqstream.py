"""IQ ingest: the kiwiclient side of the boundary (kiwi/client.py, kiwi/worker.py).

  IQBatcher       mixin for the reference's KiwiSDRStream: overrides the empty hook
                  _process_iq_samples(seq, samples, rssi, gps) (kiwi/client.py:493-494) and hands
                  every IQ frame to an IQHub, i.e. to ssdr_push_iq.  Use as
                      class Recorder(IQBatcher, KiwiSDRStream): pass
                  and drive it with the reference's own KiwiWorker(args=(recorder, options, run_event))
                  (kiwi/worker.py:10-79) -- its connect / open / run loop and retry table need nothing from here.
  iq_body_to_int16 / int16_to_wire   the SND IQ frame payload (kiwi/client.py:443-454) <-> the
                  little-endian int16 [n,2] layout the kernels read.
"""
import struct

import numpy as np


def iq_body_to_int16(body):
    """SND body in IQ mode (after the 3-byte 'SND' tag): '<BI' flags,seq; '>H' smeter; '<BBII' GPS;
    then big-endian int16 I0,Q0,I1,Q1...  Returns (flags, seq, rssi_dbm, gps tuple, int16[n,2] LE)."""
    flags, seq = struct.unpack("<BI", bytes(body[0:5]))
    smeter, = struct.unpack(">H", bytes(body[5:7]))
    gps = struct.unpack("<BBII", bytes(body[7:17]))
    iq = np.frombuffer(bytes(body[17:]), dtype=">i2").astype(np.int16).reshape(-1, 2)
    return flags, seq, 0.1 * smeter - 127, gps, iq


def int16_to_wire(iq, seq=0, smeter=0, flags=0, gps=(0, 0, 0, 0)):
    """Inverse of iq_body_to_int16 (test helper and loop-back source)."""
    iq = np.asarray(iq, np.int16).reshape(-1, 2)
    return (struct.pack("<BI", flags, seq) + struct.pack(">H", smeter) + struct.pack("<BBII", *gps) +
            iq.astype(">i2").tobytes())


def read_kiwi_iq_wav(path_or_bytes):
    """Kiwi IQ recording (kiwi/wavreader.py:29-65, 74-85): RIFF/WAVE, 'fmt ' (PCM, 2 channels, block align 4),
    then repeating ['kiwi' chunk '<BBII' GNSS stamp]['data' chunk little-endian int16 I,Q].  Returns
    (int16 [n_blocks, n, 2], [(last_gps_solution, dummy, gpssec, gpsnsec), ...]).  The samples are already in
    the kernels' layout, so blocks go straight to IQHub.feed / ssdr_push_iq (the reference scales by 1/65535
    for its complex64 view, wavreader.py:84; the int16 values are what was recorded)."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if data[0:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, blocks, stamps = 12, [], []
    while pos + 8 <= len(data):
        name, size = bytes(data[pos:pos + 4]), struct.unpack("<I", bytes(data[pos + 4:pos + 8]))[0]
        body = data[pos + 8:pos + 8 + size]
        if name == b"fmt ":
            tag, nch, _, _, align = struct.unpack("<HHLLH", bytes(body[:14]))
            if not (tag == 1 and nch == 2 and align == 4):
                raise ValueError("this is not a KiwiSDR IQ wav file")
        elif name == b"kiwi":
            stamps.append(struct.unpack("<BBII", bytes(body[:10])))
        elif name == b"data":
            blocks.append(np.frombuffer(bytes(body), dtype="<i2").reshape(-1, 2).astype(np.int16))
        pos += 8 + size + (size & 1)
    return np.stack(blocks), stamps


def kiwi_iq_wav_time_axis(stamps, block_len, nominal_rate=12000.0):
    """The time axis kiwi/wavreader.py builds for a Kiwi IQ recording (wavreader.py:86-99): every block is stamped with the GNSS
    time of its first sample; the sample rate is re-estimated from consecutive stamps -- taken as measured for the first blocks,
    then smoothed 0.9 / 0.1 -- and a block's samples sit at stamp + k / rate.  The first two blocks only prime the estimate
    (the reader hands out no time for them).  stamps: the (last_gps_solution, dummy, gpssec, gpsnsec) tuples read_kiwi_iq_wav
    returns; -> (t float64 [n_blocks - 2, block_len] GNSS seconds, rate estimates float64 [n_blocks])."""
    rate, last, primed = float(nominal_rate), -1.0, 0
    rates, rows = [], []
    for _, _, sec, nsec in stamps:
        now = sec + 1e-9 * nsec
        if last >= 0:
            measured = block_len / (now - last)
            rate = measured if primed < 3 else 0.9 * rate + 0.1 * measured
        if primed >= 2:
            rows.append(np.arange(start=now, stop=now + (block_len - 0.5) / rate, step=1 / rate, dtype=np.float64))
        rates.append(rate)
        last = now
        primed += primed < 3
    return (np.stack(rows) if rows else np.zeros((0, block_len))), np.array(rates)


class IQBatcher:
    """Mixin: IQ frames -> IQHub.  `samples` arrives as the reference builds it (kiwi/client.py:
    449-453): complex64 with unscaled int16 values in re/im, so the conversion back is exact."""
    hub = None
    channel = 0
    last_rssi = -127.0
    last_seq = -1
    last_gps = None                 # the GNSS stamp of the newest frame, as _process_aud built it (kiwi/client.py:444-445)
    dropped = 0

    def attach(self, hub, channel):
        self.hub, self.channel = hub, int(channel)
        return self

    def _process_iq_samples(self, seq, samples, rssi, gps):
        if self.hub is None:
            return
        if self.last_seq >= 0 and seq != ((self.last_seq + 1) & 0xFFFFFFFF):
            self.dropped += 1                       # sequence gap: the hub keeps streaming, history stays continuous
        self.last_seq, self.last_rssi, self.last_gps = seq, rssi, gps
        z = np.asarray(samples)
        iq = np.empty((len(z), 2), np.int16)
        iq[:, 0] = z.real
        iq[:, 1] = z.imag
        self.hub.feed(self.channel, iq)

    def _process_audio_samples(self, seq, samples, rssi):
        pass                                        # demodulation happens on the GPU, not on the server

    def _process_waterfall_samples(self, seq, samples):
        pass
