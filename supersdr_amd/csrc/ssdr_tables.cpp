// ssdr_tables.cpp -- host-side tables and per-channel parameter compilation.
//
// Every table is a float64 formula rounded once to float32.  The FIR design is the
// reference's only filter design, filtering.__init__ (utils_supersdr.py:334-344):
//   b = fl/fs; N = ceil(4/b) forced odd; h = sinc(2 fl/fs (n-(N-1)/2)) * blackman(N); h /= sum(h)
#include "ssdr_kernels.h"
#include <cmath>
#include <cstring>
#include <algorithm>

static const double kPi = 3.14159265358979323846;

void ssdr_make_window(float *win)
{
    for (int n = 0; n < SSDR_NFFT; n++) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * kPi * n / SSDR_NFFT));
}

void ssdr_make_twiddles(float *wr, float *wi)
{
    for (int m = 0; m < 512; m++) {
        wr[m] = (float)std::cos(2.0 * kPi * m / SSDR_NFFT);
        wi[m] = (float)(-std::sin(2.0 * kPi * m / SSDR_NFFT));
    }
    wr[0] = 1.0f; wi[0] = 0.0f;
    wr[256] = 0.0f; wi[256] = -1.0f;
}

// stage t (FFT stage 6+t), sub-block jl, lane l -> W_1024[(l + 32 jl) << (4 - t)]
void ssdr_make_tw_stage(float2 *tw)
{
    float wr[512], wi[512];
    ssdr_make_twiddles(wr, wi);
    for (int t = 0; t < 5; t++) {
        const int half = 1 << t, off = 32 * (half - 1);
        for (int jl = 0; jl < half; jl++)
            for (int l = 0; l < 32; l++) {
                const int m = (l + 32 * jl) << (4 - t);
                tw[off + jl * 32 + l] = make_float2(wr[m], wi[m]);
            }
    }
}

// Twiddles of the float64 waterfall kernel's stages 5..10 (ssdr_wf_exact.hip), W_1024^m = exp(-2 pi j m / 1024) in double,
// laid out the way the lanes read them (lo4 = a-index bits 3..0, b4 / b5 = lane bits 4 / 5):
//   stage 5: T5[lo4]            m = 32 lo4
//   stage 6: T6[q][lo4]         m = 16 (lo4 + 16 q), q < 2          stage 7: m = 8 (lo4 + 16 q), q < 4
//   stage 8: T8[q][lo4]         m = 4 (lo4 + 16 q), q < 8
//   stage 9: T9[mr][b4][lo4]    m = 2 (16 (mr + 8 b4) + lo4), mr < 8
//   stage 10: T10[mm][b4][b5][lo4]  m = 16 (mm + 4 b5 + 8 b4) + lo4, mm < 4   (the partner W^(m + 256) = -j W^m is not stored)
void ssdr_make_tw64(double *tw)
{
    int n = 0;
    auto put = [&](int m) { tw[2 * n] = std::cos(2.0 * kPi * m / SSDR_NFFT); tw[2 * n + 1] = -std::sin(2.0 * kPi * m / SSDR_NFFT);
                            if (m == 0) { tw[2 * n] = 1.0; tw[2 * n + 1] = 0.0; }
                            if (m == 256) { tw[2 * n] = 0.0; tw[2 * n + 1] = -1.0; }
                            n++; };
    for (int s = 5; s <= 8; s++)
        for (int q = 0; q < (1 << (s - 5)); q++)
            for (int lo4 = 0; lo4 < 16; lo4++) put((lo4 + 16 * q) << (10 - s));
    for (int mr = 0; mr < 8; mr++)
        for (int b4 = 0; b4 < 2; b4++)
            for (int lo4 = 0; lo4 < 16; lo4++) put(2 * (16 * (mr + 8 * b4) + lo4));
    for (int mm = 0; mm < 4; mm++)
        for (int b4 = 0; b4 < 2; b4++)
            for (int b5 = 0; b5 < 2; b5++)
                for (int lo4 = 0; lo4 < 16; lo4++) put(16 * (mm + 4 * b5 + 8 * b4) + lo4);
}

void ssdr_make_thresholds(float *thr)
{
    for (int k = 0; k < 256; k++) thr[k] = (float)(std::pow(10.0, (k - 255) / 10.0) * 281474976710656.0 /* 2^48 */);
}

// Quantiser segments (see ssdr_wf.hip:quantise).  The kernel looks at p' = p * 2^-48 clamped to [0, 1]; segment i holds
// the floats with (bits(p') >> SSDR_LUT_SHIFT) == i.  With base = #{k>=1 : T'[k] <= lower edge} (T' = T * 2^-48, exact)
// and t = the bits of the one threshold inside the segment (or of its upper edge if there is none), the word is
//     (base << 24) + 2^24 - t        (mod 2^32)
// so that bits(p') + word = ((base + 1) << 24) + (bits(p') - t): the top byte is base + 1 from the threshold on and
// base below it (|bits(p') - t| <= 2^21 inside the segment).  A segment is narrower than 1 dB, so it never holds two
// thresholds (checked).
int ssdr_make_quant_lut(uint32_t *lut)
{
    float thr[256];
    ssdr_make_thresholds(thr);
    for (int k = 0; k < 256; k++) thr[k] *= SSDR_LUT_SCALE;
    for (int i = 0; i < SSDR_LUT_N; i++) {
        const uint32_t lo_bits = (uint32_t)i << SSDR_LUT_SHIFT;
        const uint32_t hi_bits = lo_bits + (1u << SSDR_LUT_SHIFT);
        float lo, hi;
        std::memcpy(&lo, &lo_bits, 4);
        std::memcpy(&hi, &hi_bits, 4);
        uint32_t base = 0;
        for (int k = 1; k < 256; k++) base += (thr[k] <= lo) ? 1u : 0u;
        uint32_t t_bits = hi_bits;                              // no threshold inside: never reached
        if (base < 255 && thr[base + 1] < hi) std::memcpy(&t_bits, &thr[base + 1], 4);
        if (base < 254 && thr[base + 2] < hi) return -1;
        lut[i] = (base << 24) + (1u << 24) - t_bits;
    }
    return 0;
}

static double sinc_pi(double x)
{
    if (x == 0.0) return 1.0;
    const double y = kPi * x;
    return std::sin(y) / y;
}

// utils_supersdr.py:334-344, with the tap count capped at n_max (odd); n_min > 0: at least that many taps (odd)
static int design_lowpass_n(double fl, double fs, int n_max, int n_min, double *h)
{
    const double b = fl / fs;
    int N = (int)std::ceil(4.0 / b);
    if (N % 2 == 0) N += 1;
    if (N < n_min) N = (n_min % 2) ? n_min : n_min + 1;
    if (N > n_max) N = (n_max % 2) ? n_max : n_max - 1;
    double sum = 0.0;
    for (int n = 0; n < N; n++) {
        const double w = (N == 1) ? 1.0
                                  : 0.42 - 0.5 * std::cos(2.0 * kPi * n / (N - 1)) + 0.08 * std::cos(4.0 * kPi * n / (N - 1));
        h[n] = sinc_pi(2.0 * fl / fs * (n - (N - 1) / 2.0)) * w;
        sum += h[n];
    }
    for (int n = 0; n < N; n++) h[n] /= sum;
    return N;
}
int ssdr_design_lowpass(double fl, double fs, int n_max, double *h) { return design_lowpass_n(fl, fs, n_max, 0, h); }
int ssdr_design_lowpass_exact(double fl, double fs, int n, double *h) { return design_lowpass_n(fl, fs, n, n, h); }

static uint32_t dphi_of(double f_hz, double rate)
{
    const double x = std::nearbyint(f_hz / rate * 4294967296.0);
    long long v = (long long)x % 4294967296ll;
    if (v < 0) v += 4294967296ll;
    return (uint32_t)v;
}

static uint32_t dphi_at(double f_hz, double fs)
{
    const double x = std::nearbyint(f_hz / fs * 4294967296.0);
    long long v = (long long)x % 4294967296ll;
    if (v < 0) v += 4294967296ll;
    return (uint32_t)v;
}

// decim = D in {1, 2, 4}: the IQ arrives at D * rate and the channel filter decimates to `rate`;
// rate = 12000, or 20250 for a three-channel KiwiSDR (utils_supersdr.py:988-994)
int ssdr_compile_params_host(const ssdr_chan_params *p, ssdr_chan_consts *c, float *taps, uint32_t decim, uint32_t rate_hz)
{
    if (!p || !c || !taps) return SSDR_EINVAL;
    if (p->mode < SSDR_MODE_AM || p->mode > SSDR_MODE_IQ) return SSDR_EINVAL;
    if (decim != 1 && decim != 2 && decim != 4) return SSDR_EINVAL;
    if (rate_hz != SSDR_RATE && rate_hz != SSDR_RATE_WIDE) return SSDR_EINVAL;
    const double rate = (double)rate_hz;
    const double fs_in = rate * decim;
    // the tuning offset must lie inside the IQ band: beyond +-fs/2 the NCO step wraps mod 2^32 and the channel would
    // silently demodulate an alias
    if (!(std::fabs(p->f_shift_hz) <= fs_in / 2.0)) return SSDR_EINVAL;
    // the waterfall quantiser scales the calibration factor by 2^-48 (ssdr_wf.hip:quantise): it must stay a normal float
    if (!(std::fabs(p->wf_cal_db) <= 200.0)) return SSDR_EINVAL;
    std::memset(c, 0, sizeof *c);
    double f_bc, fl;
    if (p->mode >= SSDR_MODE_LSB && p->mode <= SSDR_MODE_CW) {
        f_bc = 0.5 * (p->low_cut + p->high_cut);
        fl = 0.5 * std::fabs(p->high_cut - p->low_cut);
    } else {
        f_bc = 0.0;
        fl = std::max(std::fabs(p->low_cut), std::fabs(p->high_cut));
    }
    fl = std::min(std::max(fl, 50.0), rate / 2.0);          // the OUTPUT rate bounds the passband: this is the anti-alias filter too
    double h[SSDR_NTAP_MAX];
    // The reference's tap formula (Blackman-windowed sinc, cut-off fl) at the input rate.  Its length rule N = ceil(4 fs / fl)
    // sizes an interpolation filter; in front of a decimator it would leave a transition band of 5.5 fs_in / N that reaches
    // far into what folds onto the 12 kHz output (33 taps for the full-band AM default at D = 4: +-4 kHz around 6 kHz).  A
    // decimating channel filter therefore always takes the whole tap budget -- 127 taps at D = 2, 125 at D = 4 (each of the
    // four polyphase streams gets 32 slots, one of them the stream's leading delay tap): transition +-0.52 / +-1.06 kHz
    // around the cut-off, >= 74 dB beyond it.
    const int n_cap = decim == 4 ? 125 : SSDR_NTAP_MAX - 1;
    const int ntap = design_lowpass_n(fl, fs_in, n_cap, decim > 1 ? n_cap : 0, h);
    for (int i = 0; i < SSDR_NTAP_MAX; i++) taps[i] = (i < ntap) ? (float)h[i] : 0.0f;
    // The windowed-sinc formula leaves numerical dust where a tap is mathematically zero (sinc at integers,
    // Blackman end points: 1e-17 .. 1e-34 against a peak of ~1).  Taps below 2^-40 of the largest are set to
    // exactly zero, and the kernel skips 4-tap groups that are all zero (AM at the full 12 kHz band is a
    // pure delay: one group instead of three).
    float hmax = 0.0f;
    for (int i = 0; i < ntap; i++) hmax = std::max(hmax, std::fabs(taps[i]));
    uint32_t groups = 0;
    for (int i = 0; i < ntap; i++) {
        if (std::fabs(taps[i]) < hmax * 0x1p-40f) taps[i] = 0.0f;
        if (taps[i] != 0.0f) groups |= 1u << (i >> 2);
    }
    c->mode = (uint32_t)p->mode;
    c->ntap = (uint32_t)ntap;
    c->ntap8 = (uint32_t)((ntap + 7) & ~7);
    c->tap_groups = groups;
    // exactly one tap, a unit one, at index 4: the filter is a 4-sample delay (fl = 6 kHz: sinc vanishes at every other tap)
    {
        int nz = 0;
        for (int i = 0; i < ntap; i++) nz += (taps[i] != 0.0f);
        c->fir_flags = (nz == 1 && ntap > 4 && taps[4] == 1.0f && p->mode != SSDR_MODE_IQ) ? SSDR_FIR_DELAY4 : 0u;
    }
    c->dphi1 = dphi_at(p->f_shift_hz + f_bc, fs_in);            // the mixer runs at the input rate,
    c->dphi2 = dphi_of(f_bc, rate);                             // the SSB re-mixer at the output rate
    c->kfm = (float)(16384.0 * rate / (2.0 * kPi * 5000.0));    // NBFM: 5 kHz deviation <-> half scale
    c->decim = decim;
    if (decim > 1) {
        // Polyphase streams v_q[m] = z[D m + q], q = 0..D-1 (what a lane's D*8 consecutive inputs de-interleave into):
        //   y[m] = sum_k h[k] z[D m - k] = sum_i h[D i] v_0[m - i] + sum_{q>=1} sum_i h[D i + (D - q)] v_q[m - 1 - i]
        // so stream q >= 1 carries its taps behind one zero tap.  Layout: stream q at taps[q * 128/D ...].
        float g[SSDR_NTAP_MAX];
        std::memset(g, 0, sizeof g);
        const int slots = SSDR_NTAP_MAX / (int)decim;
        int longest = 0;
        for (int k = 0; k < ntap; k++) {
            const int pph = k % (int)decim, i = k / (int)decim;
            const int q = pph ? (int)decim - pph : 0, pos = pph ? i + 1 : i;
            if (pos >= slots) return SSDR_EINVAL;
            g[q * slots + pos] = taps[k];
            longest = std::max(longest, pos + 1);
        }
        std::memcpy(taps, g, sizeof g);
        c->ntap8 = (uint32_t)((longest + 7) & ~7);              // per stream
        c->fir_flags = 0;
        c->tap_groups = 0;
    }
    c->wf_cal_lin = (float)std::pow(10.0, p->wf_cal_db / 10.0);
    c->smeter_cal_db = (float)p->smeter_cal_db;
    const double log2_10 = std::log2(10.0);
    double c0, c1;
    if (p->mode == SSDR_MODE_NBFM) {
        c0 = 0.0; c1 = 0.0;
    } else if (p->agc_on) {
        const double gs = std::min(std::max(p->agc_slope, 0.0), 10.0) / 100.0;
        c1 = (gs - 1.0) / 2.0;
        c0 = -1.0 - 30.0 * c1;
    } else {
        c1 = 0.0;
        c0 = (p->agc_man_gain - 50.0) * log2_10 / 20.0;
    }
    const double knee = (p->agc_thresh - p->smeter_cal_db) * (log2_10 / 10.0) + 30.0;
    const double tau = std::max(p->agc_decay, 1.0) / 1000.0;
    const double delta8 = 2.0 * std::log2(std::exp(1.0)) * 8.0 / (tau * rate);
    c->agc_c0 = (float)c0;
    c->agc_c1 = (float)c1;
    c->agc_knee = (float)knee;
    c->agc_delta8 = (float)delta8;
    c->hang_frames = p->agc_hang ? (uint32_t)std::min(std::max(std::nearbyint(p->agc_decay / 500.0), 1.0), 8.0) : 0u;
    return SSDR_OK;
}
