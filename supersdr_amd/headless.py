"""A minimal pair of worker classes for running the GPU path WITHOUT supersdr's UI module (`utils_supersdr` needs pygame,
sounddevice and tkinter): the replay tool, the GPU-box tests, headless services.

    gpu = supersdr_amd.workers.bind(supersdr_amd.headless)        # instead of bind(utils_supersdr)

Self-written and deliberately bare: two classes that hold the attributes the seams read and reach their "server" the way
the reference's constructors do -- through the module-level names kiwi_sdr, socket, wsclient, Stream and the "SET ..."
text commands of SURVEY.md appendix A -- so that bind() goes through exactly the code path it takes over the real module.
None of the reference's own logic (zoom / tick / passband arithmetic, pacing loop, WAV recorder) is here: with the UI,
bind the maintainer's module and inherit it.
"""
import queue
import socket                # noqa: F401  (a name workers._server_is_the_gpu replaces while a constructor runs)
import time
from collections import deque

import numpy as np

wsclient = None              # the reference has `from kiwi import wsclient` here
Stream = None                # ... and mod_pywebsocket's Stream


def kiwi_sdr(host, port, verbose=False):
    raise OSError("no network in the tests")


class audio_recording:
    def __init__(self, kiwi_snd):
        self.audio_buffer, self.recording_flag, self.kiwi_snd = [], False, kiwi_snd


class kiwi_waterfall:
    WF_BINS = 1024
    MIN_DYN_RANGE = 40.
    delta_low_db, delta_high_db = 0, 0
    low_clip_db, high_clip_db = -120, -60
    wf_min_db, wf_max_db = -120, -80
    wf_buffer_len = 3

    def __init__(self, host_, port_, pass_, zoom_, freq_, eibi, disp):
        self.eibi, self.host, self.port, self.password = eibi, host_, port_, pass_
        self.zoom, self.freq = zoom_, freq_ if freq_ else 14200
        self.averaging_n, self.wf_auto_scaling = 1, True
        self.dynamic_range = self.MIN_DYN_RANGE
        self.terminate, self.run_index, self.counter = False, 0, 0
        self.radio_mode = "USB"
        self.wf_color = None
        status = kiwi_sdr(host_, port_, True)
        self.freq_offset = status.freq_offset / 1000.0
        self.socket = socket.socket()
        self.socket.connect((self.host, self.port))
        self.kiwi_wf_timestamp = int(time.time())
        wsclient.ClientHandshakeProcessor(self.socket, self.host, self.port).handshake("/%d/W/F" % self.kiwi_wf_timestamp)
        self.wf_stream = Stream(wsclient.ClientRequest(self.socket), None)
        self.wf_stream.send_message("SET zoom=%d start=%d" % (self.zoom, self.counter))
        while bytes(self.wf_stream.receive_message()[0:3]) != b"W/F":
            pass
        self.wf_data = np.zeros((disp.WF_HEIGHT, self.WF_BINS))
        self.wf_data_tmp = deque([], self.wf_buffer_len)

    def set_freq_zoom(self, freq_, zoom_):
        self.freq, self.zoom = freq_, zoom_
        self.wf_stream.send_message("SET zoom=%d start=%d" % (self.zoom, self.counter))
        return self.freq


class kiwi_sound:
    FORMAT = np.int16
    CHANNELS = 2
    AUDIO_RATE = 48000
    KIWI_RATE = 12000
    SAMPLE_RATIO = int(AUDIO_RATE / KIWI_RATE)
    CHUNKS = 1
    KIWI_SAMPLES_PER_FRAME = 512

    def __init__(self, freq_, mode_, lc_, hc_, password_, kiwi_wf, buffer_len, volume_=100, host_=None, port_=None, subrx_=False):
        self.subrx, self.kiwi_wf = subrx_, kiwi_wf
        self.host, self.port = host_ if host_ else kiwi_wf.host, port_ if port_ else kiwi_wf.port
        self.FULL_BUFF_LEN = max(1, buffer_len)
        self.audio_buffer = queue.Queue(maxsize=self.FULL_BUFF_LEN)
        self.terminate, self.volume = False, volume_
        self.max_rssi_before_mute, self.mute_counter, self.muting_delay = -20, 0, 15
        self.adc_overflow_flag, self.status, self.run_index, self.delta_t, self.rssi = False, None, 0, 0.0, -127
        self.freq, self.radio_mode, self.lc, self.hc = freq_, mode_, lc_, hc_
        self.on, self.hang, self.thresh, self.slope, self.decay, self.gain = True, False, -80, 0, 4000, 50
        self.audio_balance = 0.0
        kiwi_sdr(self.host, self.port)
        self.socket = socket.socket()
        self.socket.connect((self.host, self.port))
        wsclient.ClientHandshakeProcessor(self.socket, self.host, self.port).handshake("/%d/SND" % kiwi_wf.kiwi_wf_timestamp)
        self.stream = Stream(wsclient.ClientRequest(self.socket), None)
        self.set_mode_freq_pb()
        self.set_agc_params()
        while True:
            msg = bytes(self.stream.receive_message())
            if msg[:3] == b"SND":
                break
            if b"MSG audio_init" in msg:
                els = msg[4:].decode().split()
                self.KIWI_RATE = int(els[1].split("=")[1])
                self.KIWI_RATE_TRUE = float(els[2].split("=")[1])
                self.SAMPLE_RATIO = self.AUDIO_RATE / self.KIWI_RATE
        self.audio_rec = audio_recording(self)

    def set_agc_params(self):
        self.stream.send_message("SET agc=%d hang=%d thresh=%d slope=%d decay=%d manGain=%d"
                                 % (self.on, self.hang, self.thresh, self.slope, self.decay, self.gain))

    def set_mode_freq_pb(self):
        self.stream.send_message("SET mod=%s low_cut=%d high_cut=%d freq=%.3f" % (self.radio_mode.lower(), self.lc, self.hc, self.freq))
