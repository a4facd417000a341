"""The reference's own scalar control surface, kept statement for statement.

SURVEY.md section 8b requires that supersdr.py runs unmodified against the GPU-backed workers, and Appendix B5
says to keep the pacing loop verbatim.  Everything in this module is therefore a transcription of
`/root/reference/utils_supersdr.py` -- class constants, frequency / zoom / tick arithmetic, passband tables, the AGC
delay stepper, the TX-mute counter, the jitter-buffer pacing loop, the WAV recorder -- with the file:line it mirrors.
None of it is DSP and none of it is on the hot path; it lives here, apart from `workers.py`, so that the product's
own logic (the GPU seams, the hub, the hand-out of kernel results) and the transcribed surface can be told apart.
"""
import time
import wave
from datetime import datetime

import numpy as np

# module constants of the reference (utils_supersdr.py:42-50)
CW_PITCH = 0.6
LOW_CUT_SSB, HIGH_CUT_SSB = 30, 3000
LOW_CUT_CW, HIGH_CUT_CW = int(CW_PITCH * 1000 - 200), int(CW_PITCH * 1000 + 200)
HIGHLOW_CUT_AM = 6000


class WaterfallSurface:
    """kiwi_waterfall's constants and scalar UI arithmetic (utils_supersdr.py:593-604, 697-717, 747-778, 815-845, 859-873)."""
    MAX_FREQ = 30000
    CENTER_FREQ = int(MAX_FREQ / 2)
    MAX_ZOOM = 14
    WF_BINS = 1024
    MAX_FPS = 23
    MIN_DYN_RANGE = 40.
    CLIP_LOWP, CLIP_HIGHP = 40., 100
    delta_low_db, delta_high_db = 0, 0
    low_clip_db, high_clip_db = -120, -60
    wf_min_db, wf_max_db = low_clip_db, low_clip_db + MIN_DYN_RANGE
    kiwi_wf_timestamp = None
    wf_buffer_len = 3

    # ---- utils_supersdr.py:747-778
    def zoom_to_span(self):
        assert 0 <= self.zoom <= self.MAX_ZOOM
        self.span_khz = self.MAX_FREQ / 2 ** self.zoom
        return self.span_khz

    def start_frequency_to_counter(self, start_frequency_):
        assert 0 <= start_frequency_ <= self.MAX_FREQ
        self.counter = round(start_frequency_ / self.MAX_FREQ * 2 ** self.MAX_ZOOM * self.WF_BINS)
        return self.counter, self.counter * self.MAX_FREQ / self.WF_BINS / 2 ** self.MAX_ZOOM

    def start_freq(self):
        self.start_f_khz = self.freq - self.span_khz / 2
        return self.start_f_khz

    def end_freq(self):
        self.end_f_khz = self.freq + self.span_khz / 2
        return self.end_f_khz

    def offset_to_bin(self, offset_khz_):
        return self.WF_BINS / self.span_khz * offset_khz_

    def bins_to_khz(self, bins_):
        return bins_ / (self.WF_BINS / self.span_khz) + self.start_f_khz

    def deltabins_to_khz(self, bins_):
        return bins_ / (self.WF_BINS / self.span_khz)

    def gen_div(self):                                   # utils_supersdr.py:697-717
        self.space_khz = 10
        self.div_list, self.subdiv_list = [], []
        f_s, f_e = int(self.start_f_khz), int(self.end_f_khz)
        while self.div_list == [] and self.subdiv_list == []:
            if self.bins_per_khz * self.space_khz > self.min_bin_spacing:
                self.div_list = [int(self.offset_to_bin(f - self.start_f_khz)) for f in range(f_s, f_e + 1)
                                 if not f % self.space_khz]
            if self.bins_per_khz * self.space_khz / 10 > self.min_bin_spacing / 10:
                self.subdiv_list = [int(self.offset_to_bin(f - self.start_f_khz)) for f in range(f_s, f_e + 1)
                                    if not f % (self.space_khz / 10)]
            self.space_khz *= 10

    def set_freq_zoom(self, freq_, zoom_):               # utils_supersdr.py:815-845
        self.freq, self.zoom = freq_, zoom_
        self.zoom_to_span()
        self.start_freq()
        self.end_freq()
        if zoom_ == 0:
            self.freq = self.CENTER_FREQ
            self.start_freq()
            self.end_freq()
            self.span_khz = self.MAX_FREQ
        elif self.start_f_khz < 0:
            self.freq = self.zoom_to_span() / 2
            self.start_freq()
            self.end_freq()
        elif self.end_f_khz > self.MAX_FREQ:
            self.freq = self.MAX_FREQ - self.zoom_to_span() / 2
            self.start_freq()
            self.end_freq()
        self.counter, _ = self.start_frequency_to_counter(self.start_f_khz)
        if self.eibi is not None:
            self.eibi.get_stations(self.start_f_khz, self.end_f_khz)
        self.bins_per_khz = self.WF_BINS / self.span_khz
        self.gen_div()
        return self.freq

    def change_passband(self, delta_low_, delta_high_):  # utils_supersdr.py:859-873
        if self.radio_mode == "USB":
            lc_, hc_ = LOW_CUT_SSB + delta_low_, HIGH_CUT_SSB + delta_high_
        elif self.radio_mode == "LSB":
            lc_, hc_ = -HIGH_CUT_SSB - delta_high_, -LOW_CUT_SSB - delta_low_
        elif self.radio_mode == "AM":
            lc_, hc_ = -HIGHLOW_CUT_AM - delta_low_, HIGHLOW_CUT_AM + delta_high_
        else:
            lc_, hc_ = LOW_CUT_CW + delta_low_, HIGH_CUT_CW + delta_high_
        self.lc, self.hc = lc_, hc_
        return lc_, hc_


class audio_recording:
    """utils_supersdr.py:144-172: collects the mono 48 kHz blocks play_buffer appends and writes them as a WAV file."""
    CHANNELS = 1

    def __init__(self, kiwi_snd):
        self.filename = ""
        self.audio_buffer = []
        self.kiwi_snd = kiwi_snd
        self.frames = []
        self.recording_flag = False

    def start(self):
        self.filename = "supersdr_%sUTC.wav" % datetime.utcnow().isoformat().split(".")[0].replace(":", "_")
        self.audio_buffer = []
        self.recording_flag = True

    def stop(self):
        self.recording_flag = False
        self.save()

    def save(self):
        self.wave = wave.open(self.filename, "wb")
        self.wave.setnchannels(self.CHANNELS)
        self.wave.setsampwidth(2)                        # two bytes per sample (int16)
        self.wave.setframerate(self.kiwi_snd.AUDIO_RATE)
        self.wave.writeframes(b"".join(np.asarray(b, np.int16).tobytes() for b in self.audio_buffer))
        self.wave.close()
        self.recording = False


class SoundSurface:
    """kiwi_sound's constants and host control flow (utils_supersdr.py:902-909, 1009-1020, 1078-1092, 1142-1147, 1150-1186)."""
    FORMAT = np.int16
    CHANNELS = 2
    AUDIO_RATE = 48000
    KIWI_RATE = 12000
    SAMPLE_RATIO = int(AUDIO_RATE / KIWI_RATE)
    CHUNKS = 1
    KIWI_SAMPLES_PER_FRAME = 512

    def change_agc_delay(self, delta):                   # utils_supersdr.py:1009-1020
        if delta < 0:
            if self.decay > self.min_agc_delay:
                self.decay += delta
        elif self.decay < self.max_agc_delay:
            self.decay += delta
        if self.radio_mode == "CW":
            self.decay_cw = self.decay
        else:
            self.decay_other = self.decay

    def change_passband(self, delta_low_, delta_high_):  # utils_supersdr.py:1078-1092
        if self.radio_mode == "USB":
            lc_, hc_ = LOW_CUT_SSB + delta_low_, HIGH_CUT_SSB + delta_high_
        elif self.radio_mode == "LSB":
            lc_, hc_ = -HIGH_CUT_SSB - delta_high_, -LOW_CUT_SSB - delta_low_
        elif self.radio_mode == "AM":
            lc_, hc_ = -HIGHLOW_CUT_AM - delta_low_, HIGHLOW_CUT_AM + delta_high_
        elif self.radio_mode == "CW":
            lc_, hc_ = LOW_CUT_CW + delta_low_, HIGH_CUT_CW + delta_high_
        else:
            lc_, hc_ = self.lc, self.hc
        self.lc, self.hc = lc_, hc_
        return lc_, hc_

    def _mute_logic(self, outdata):                      # utils_supersdr.py:1142-1147
        if self.rssi > self.max_rssi_before_mute:
            self.mute_counter = self.muting_delay
        elif self.mute_counter > 0:
            self.mute_counter -= 1
        if self.mute_counter > 0:
            outdata *= 0

    def run(self):                                       # pacing loop: utils_supersdr.py:1150-1186
        self.total_delay_ms = 0.0
        delta_time_ms = 0.0
        self.ms_per_frame = (self.KIWI_SAMPLES_PER_FRAME / self.KIWI_RATE_TRUE) * 1000
        self.late_flag = False
        while not self.terminate:
            time_prev = time.time_ns() / 1000000
            snd_buf = self.get_audio_chunk()
            if snd_buf is not None and not self.late_flag:
                self.audio_buffer.put(snd_buf)
                self.run_index += 1
                self.total_delay_ms -= delta_time_ms
            else:
                self.total_delay_ms -= self.ms_per_frame
            delta_time_ms = time.time_ns() / 1000000 - time_prev
            self.total_delay_ms += delta_time_ms
            if not self.late_flag and self.total_delay_ms > (self.FULL_BUFF_LEN + 2) * self.ms_per_frame:
                self.late_flag = True
            if self.late_flag and self.total_delay_ms < self.ms_per_frame:
                while self.audio_buffer.qsize() < self.FULL_BUFF_LEN and not self.terminate:
                    snd_buf = self.get_audio_chunk()
                    if snd_buf is not None:
                        self.audio_buffer.put(snd_buf)
                self.late_flag = False
                self.total_delay_ms = 0.0
                delta_time_ms = 0.0
