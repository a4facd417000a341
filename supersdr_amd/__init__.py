"""supersdr_amd -- MI355X (gfx950) implementation of SuperSDR's DSP hot path.

Waterfall (windowed 1024-pt FFT -> 1-dB bytes -> N-line time binning) and the 12 kHz
IQ audio chain (NCO, FIR, AM/SSB/CW/NBFM, AGC) as hand-written HIP kernels behind the
C-ABI in include/ssdr.h; the Python host goes in front of the reference's own kiwi_waterfall /
kiwi_sound classes (supersdr_amd.workers: bind(), IQHub, GpuStream) and its kiwiclient hook
(supersdr_amd.iqstream: IQBatcher).
"""
from ._lib import (NFFT, FRAME, RATE, NTAP_MAX, HIST, MODE_AM, MODE_LSB, MODE_USB, MODE_CW, MODE_NBFM,
                   MODE_BY_NAME, ChanParams, SsdrError, LIB_PATH)
from .engine import SsdrEngine, default_params, compile_params, table

__all__ = ["SsdrEngine", "default_params", "compile_params", "table", "ChanParams", "SsdrError",
           "NFFT", "FRAME", "RATE", "NTAP_MAX", "HIST", "MODE_BY_NAME", "LIB_PATH"]
