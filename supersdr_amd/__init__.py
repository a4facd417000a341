"""supersdr_amd -- MI355X (gfx950) implementation of SuperSDR's DSP hot path.

Waterfall (windowed 1024-pt FFT -> 1-dB bytes -> N-line time binning) and the 12 kHz
IQ audio chain (NCO, FIR, AM/SSB/CW/NBFM, AGC) as hand-written HIP kernels behind the
C-ABI in include/ssdr.h; the Python host keeps the reference's kiwi_waterfall /
kiwi_sound worker surface (supersdr_amd.waterfall / supersdr_amd.sound).
"""
from ._lib import (NFFT, FRAME, RATE, NTAP_MAX, HIST, MODE_AM, MODE_LSB, MODE_USB, MODE_CW, MODE_NBFM,
                   MODE_BY_NAME, ChanParams, SsdrError, LIB_PATH)
from .engine import SsdrEngine, default_params, compile_params, table

__all__ = ["SsdrEngine", "default_params", "compile_params", "table", "ChanParams", "SsdrError",
           "NFFT", "FRAME", "RATE", "NTAP_MAX", "HIST", "MODE_BY_NAME", "LIB_PATH"]
