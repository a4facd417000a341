/*
 * ssdr_twin.c -- fp32 CPU restatement ("twin") of the SuperSDR DSP hot path.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Built into oracle/libssdr_twin.so by
 * oracle/Makefile.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under supersdr_amd/ links or calls it.
 *
 * Role: oracle/ssdr_oracle.py (NumPy float64) is the normative definition of the
 * path; this file restates the SAME algorithm in float32 with every rounding
 * step spelled out (explicit fmaf, -ffp-contract=off, no libm transcendentals),
 * so that a correct HIP implementation is bit-identical to it.  It is written
 * as plain scalar loops (textbook radix-2 FFT, direct-form FIR, sequential
 * scans "as a 64-lane machine would"), independently of the kernels' tiling.
 *
 * What is pinned and what is not:
 *   - time binning == integer sum of byte lines: reference utils_supersdr.py:881-888
 *     (np.mean of float32 byte values == int sum / N; golden: tests/golden/binning.npz)
 *   - byte semantics dBm = byte - 255: utils_supersdr.py:788-789
 *   - FIR tap formula: utils_supersdr.py:334-344 (taps arrive pre-designed)
 *   - FFT / log-mag / NCO / demod / AGC arithmetic (incl. the decimating channel filter of twin_consts.decim > 1 and the
 *     overlapping lines of twin_wf_lines): ABSENT from the reference
 *     (server-side) -> PARITY UNPINNED, defined by
 *     ssdr_oracle.py and checked against it with the guard-band / 1e-5 RMS rule.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define NFFT 1024
#define FRAME 512
#define HIST 128
#define NTAP_MAX 128
#define NLANE 64

/* ---------------------------------------------------------------- helpers */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* sin/cos of 2*pi*(phase>>12)/2^20 : quadrant reduction + cephes-style minimax
 * polynomials on [-pi/4, pi/4].  No libm. */
static void sincos20(uint32_t phase, float *c_out, float *s_out)
{
    const float C_2PI_20 = 0x1.921fb6p-18f;        /* float32(2*pi/2^20) */
    const float S1 = -1.6666654611e-1f, S2 = 8.3321608736e-3f, S3 = -1.9515295891e-4f;
    const float C1 = 4.166664568298827e-2f, C2 = -1.388731625493765e-3f, C3 = 2.443315711809948e-5f;
    uint32_t p20 = phase >> 12;
    uint32_t k = (p20 + (1u << 17)) >> 18;
    int32_t ri = (int32_t)p20 - (int32_t)(k << 18);
    float th = (float)ri * C_2PI_20;
    float t2 = th * th;
    float u = fmaf(t2, S3, S2);
    u = fmaf(t2, u, S1);
    float s = fmaf(th * t2, u, th);
    float v = fmaf(t2, C3, C2);
    v = fmaf(t2, v, C1);
    float c = fmaf(t2 * t2, v, fmaf(-0.5f, t2, 1.0f));
    switch (k & 3u) {
    case 0: *c_out = c;  *s_out = s;  break;
    case 1: *c_out = -s; *s_out = c;  break;
    case 2: *c_out = -c; *s_out = -s; break;
    default: *c_out = s; *s_out = -c; break;
    }
}

/* 1/d for d in [1, 2.42]: cubic seed + two Newton steps, seven fused multiply-adds (no divide). */
static float rcp_1to2p42(float d)
{
    float r = fmaf(fmaf(fmaf(-0.1340303272008896f, d, 0.9271132946014404f), d, -2.337818145751953f), d, 2.539294958114624f);
    float e = fmaf(-d, r, 1.0f);
    r = fmaf(r, e, r);
    e = fmaf(-d, r, 1.0f);
    r = fmaf(r, e, r);
    return r;
}

/* log2(x), x > 0 normal.  atanh series in s=(m-1)/(m+1), m in [0.707,1.414]. */
static float log2p(float x)
{
    const float K2 = 0x1.715476p+1f;                 /* float32(2/ln 2) */
    const float L1 = 0.33333333333f, L2 = 0.2f, L3 = 0.14285714286f, L4 = 0.11111111111f;
    uint32_t I = f2u(x);
    int32_t e = (int32_t)(I >> 23) - 127;
    float m = u2f((I & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float s = (m - 1.0f) * rcp_1to2p42(m + 1.0f);
    float z = s * s;
    float t = fmaf(z, L4, L3);
    t = fmaf(z, t, L2);
    t = fmaf(z, t, L1);
    t = z * t;
    float sk = s * K2;
    return (float)e + fmaf(sk, t, sk);
}

/* 2^y for |y| <= 126 */
static float exp2p(float y)
{
    const float E1 = 0.69314718056f, E2 = 0.24022650696f, E3 = 0.055504108665f,
                E4 = 0.0096181291076f, E5 = 0.0013333558146f, E6 = 0.00015403530393f,
                E7 = 0.000015252733805f;
    y = fminf(fmaxf(y, -126.0f), 126.0f);
    float n = rintf(y);
    float f = y - n;
    float r = fmaf(E7, f, E6);
    r = fmaf(r, f, E5);
    r = fmaf(r, f, E4);
    r = fmaf(r, f, E3);
    r = fmaf(r, f, E2);
    r = fmaf(r, f, E1);
    r = fmaf(r, f, 1.0f);
    return u2f(f2u(r) + ((uint32_t)(int32_t)n << 23));
}

/* atan2(y, x): t = min/max through rcp_1to2p42, atan(t) = t + t^3 Q(t^2) on [0, 1], octant by sign bits; -0 counts as +0. */
static float atan2p(float y, float x)
{
    const float Q0 = -3.3331659436e-01f, Q1 = 1.9962704182e-01f, Q2 = -1.3976582885e-01f, Q3 = 9.7942389548e-02f,
                Q4 = -5.7773657143e-02f, Q5 = 2.3040184751e-02f, Q6 = -4.3554198928e-03f;
    const float PI_2 = 1.5707963267948966f, PI_1 = 3.14159265358979323f;
    x = x + 0.0f;
    y = y + 0.0f;
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float mxc = fmaxf(mx, 1e-30f);
    const float sc = u2f(0x7F000000u - (f2u(mxc) & 0x7F800000u));       /* 2^-exponent(mx): mx*sc in [1, 2) */
    float t = (mn * sc) * rcp_1to2p42(mxc * sc);
    float z = t * t;
    float q = fmaf(Q6, z, Q5);
    q = fmaf(q, z, Q4);
    q = fmaf(q, z, Q3);
    q = fmaf(q, z, Q2);
    q = fmaf(q, z, Q1);
    q = fmaf(q, z, Q0);
    float r = fmaf(t * z, q, t);
    float d = ax - ay;
    uint32_t flip = (f2u(d) ^ f2u(x)) & 0x80000000u;
    float rs = u2f(f2u(r) ^ flip);
    float base = (x < 0.0f) ? PI_1 : 0.0f;
    base = (d < 0.0f) ? PI_2 : base;
    float a = rs + base;
    return u2f(f2u(a) | (f2u(y) & 0x80000000u));
}

/* byte = #{k in 1..255 : T[k] <= p} */
static int quantise(float p, const float *T)
{
    int lo = 0, hi = 255;          /* invariant: T[lo] <= p (lo=0 is a sentinel), T[hi+1] > p */
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (T[mid] <= p) lo = mid; else hi = mid - 1;
    }
    return lo;
}

int twin_quantise(float p, const float *T) { return quantise(p, T); }
float twin_log2p(float x) { return log2p(x); }
float twin_exp2p(float x) { return exp2p(x); }
float twin_atan2p(float y, float x) { return atan2p(y, x); }
void twin_sincos20(uint32_t ph, float *c, float *s) { sincos20(ph, c, s); }
static void phasor32(uint32_t ph, float *c, float *s);
void twin_phasor32(uint32_t ph, float *c, float *s) { phasor32(ph, c, s); }

/* ------------------------------------------------------------------ tables */
/* Restates the table definitions (double libm, rounded once to float32). */
void twin_make_tables(float *win /*1024*/, float *wr /*512*/, float *wi /*512*/, float *thr /*256*/)
{
    const double PI = 3.14159265358979323846;
    for (int n = 0; n < NFFT; n++) win[n] = (float)(0.5 - 0.5 * cos(2.0 * PI * n / NFFT));
    for (int m = 0; m < 512; m++) {
        wr[m] = (float)cos(2.0 * PI * m / NFFT);
        wi[m] = (float)(-sin(2.0 * PI * m / NFFT));
    }
    wr[0] = 1.0f; wi[0] = 0.0f; wr[256] = 0.0f; wi[256] = -1.0f;
    for (int k = 0; k < 256; k++) thr[k] = (float)(pow(10.0, (k - 255) / 10.0) * 281474976710656.0);
}

/* --------------------------------------------------------------- waterfall */
static uint32_t bitrev10(uint32_t v)
{
    uint32_t r = 0;
    for (int b = 0; b < 10; b++) r |= ((v >> b) & 1u) << (9 - b);
    return r;
}

/* textbook iterative radix-2 DIT.  Butterfly (u, v, w) -> (u + w v, u - w v) in its 6-FMA form
 * (Linzer-Feig / Goedecker):
 *     s = (fma(-wi, vi, ur), fma(wi, vr, ui))
 *     a = (fma( wr, vr, sr), fma(wr, vi, si))        = u + w v
 *     b = (fma(  2, ur, -ar), fma(2, ui, -ai))       = 2u - a
 * Stages 1..5 use the exact forms a = u + v, b = u - v for w = 1 and a = u + (vi, -vr), b = u - (vi, -vr)
 * for w = -j; stages 6..10 always use the general form (table values (1,0) and (0,-1) are exact).
 * Stage 1 comes fused with the window: it pairs sample n with sample n + 512, and with x, x' the raw samples and w, w'
 * their window values  a = fma(x', w', x w),  b = fma(-x', w', x w)  (the second product is not rounded on its own). */
static void fft1024_windowed(const int16_t *iq, const float *win, float *re, float *im, const float *wr, const float *wi)
{
    for (uint32_t i = 0; i < NFFT; i += 2) {           /* bit-reversed order: positions i, i+1 hold samples n, n + 512 */
        const uint32_t n = bitrev10(i);
        const float tr = (float)iq[2 * n] * win[n], ti = (float)iq[2 * n + 1] * win[n];
        const float xr = (float)iq[2 * (n + 512)], xi = (float)iq[2 * (n + 512) + 1], w2 = win[n + 512];
        re[i] = fmaf(xr, w2, tr);      im[i] = fmaf(xi, w2, ti);
        re[i + 1] = fmaf(-xr, w2, tr); im[i + 1] = fmaf(-xi, w2, ti);
    }
    for (int s = 2; s <= 10; s++) {
        int half = 1 << (s - 1), step = NFFT >> s;
        for (int blk = 0; blk < NFFT; blk += 2 * half) {
            for (int k = 0; k < half; k++) {
                int m = k * step, i = blk + k, j = i + half;
                float ur = re[i], ui = im[i], vr = re[j], vi = im[j];
                if (s <= 5 && m == 0) {
                    re[i] = ur + vr; im[i] = ui + vi;
                    re[j] = ur - vr; im[j] = ui - vi;
                } else if (s <= 5 && m == 256) {
                    re[i] = ur + vi; im[i] = ui - vr;
                    re[j] = ur - vi; im[j] = ui + vr;
                } else {
                    float a = wr[m], b = wi[m];
                    float sr = fmaf(-b, vi, ur), si = fmaf(b, vr, ui);
                    float ar = fmaf(a, vr, sr), ai = fmaf(a, vi, si);
                    re[i] = ar; im[i] = ai;
                    re[j] = fmaf(2.0f, ur, -ar); im[j] = fmaf(2.0f, ui, -ai);
                }
            }
        }
    }
}
/* the plain transform of already windowed data (kept for reference: twin_fft_plain) */
static void fft1024(float *re, float *im, const float *wr, const float *wi)
{
    for (uint32_t i = 0; i < NFFT; i++) {
        uint32_t j = bitrev10(i);
        if (j > i) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int s = 1; s <= 10; s++) {
        int half = 1 << (s - 1), step = NFFT >> s;
        for (int blk = 0; blk < NFFT; blk += 2 * half) {
            for (int k = 0; k < half; k++) {
                int m = k * step, i = blk + k, j = i + half;
                float ur = re[i], ui = im[i], vr = re[j], vi = im[j];
                if (s <= 5 && m == 0) {
                    re[i] = ur + vr; im[i] = ui + vi;
                    re[j] = ur - vr; im[j] = ui - vi;
                } else if (s <= 5 && m == 256) {
                    re[i] = ur + vi; im[i] = ui - vr;
                    re[j] = ur - vi; im[j] = ui + vr;
                } else {
                    float a = wr[m], b = wi[m];
                    float sr = fmaf(-b, vi, ur), si = fmaf(b, vr, ui);
                    float ar = fmaf(a, vr, sr), ai = fmaf(a, vi, si);
                    re[i] = ar; im[i] = ai;
                    re[j] = fmaf(2.0f, ur, -ar); im[j] = fmaf(2.0f, ui, -ai);
                }
            }
        }
    }
}

/* the plain radix-2 transform of caller-supplied (already windowed) data, in place */
void twin_fft_plain(float *re, float *im, const float *wr, const float *wi) { fft1024(re, im, wr, wi); }

/* one line: int16 IQ[1024][2] -> bytes[1024], ascending frequency (fftshift) */
void twin_wf_line(const int16_t *iq, const float *win, const float *wr, const float *wi,
                  const float *thr, float cal_lin, uint8_t *out)
{
    float re[NFFT], im[NFFT];
    fft1024_windowed(iq, win, re, im, wr, wi);
    for (int k = 0; k < NFFT; k++) {
        float p = fmaf(re[k], re[k], im[k] * im[k]) * cal_lin;
        out[(k + 512) & 1023] = (uint8_t)quantise(p, thr);
    }
}

/* diagnostic: the fp32 scaled powers of one line in FFT order (guard-band calibration in tests) */
void twin_wf_power(const int16_t *iq, const float *win, const float *wr, const float *wi, float cal_lin, float *p_out)
{
    float re[NFFT], im[NFFT];
    fft1024_windowed(iq, win, re, im, wr, wi);
    for (int k = 0; k < NFFT; k++) p_out[k] = fmaf(re[k], re[k], im[k] * im[k]) * cal_lin;
}

/* batch: iq[n_ch][n_lines*1024][2] -> out[n_lines/n_avg][n_ch][1024] int16 sums */
void twin_wf(const int16_t *iq, uint32_t n_ch, uint32_t n_lines, uint32_t n_avg,
             const float *cal_lin /*[n_ch]*/, const float *win, const float *wr, const float *wi,
             const float *thr, int16_t *out)
{
    uint32_t n_out = n_lines / n_avg;
    uint8_t b[NFFT];
    for (uint32_t c = 0; c < n_ch; c++) {
        for (uint32_t o = 0; o < n_out; o++) {
            int16_t *dst = out + ((size_t)o * n_ch + c) * NFFT;
            memset(dst, 0, NFFT * sizeof(int16_t));
            for (uint32_t a = 0; a < n_avg; a++) {
                const int16_t *src = iq + ((size_t)c * n_lines + (size_t)o * n_avg + a) * NFFT * 2;
                twin_wf_line(src, win, wr, wi, thr, cal_lin[c], b);
                for (int k = 0; k < NFFT; k++) dst[k] = (int16_t)(dst[k] + b[k]);
            }
        }
    }
}

/* overlapping lines: line k of channel c covers samples [k*hop, k*hop + 1024) of that channel's stream
 * iq[n_ch][(n_lines-1)*hop + 1024][2]  ->  bytes[n_lines][n_ch][1024]  (hop 512: 23.4 lines/s, the reference's MAX_FPS = 23,
 * utils_supersdr.py:597; hop 1024 is twin_wf's framing) */
void twin_wf_lines(const int16_t *iq, uint32_t n_ch, uint32_t n_lines, uint32_t hop, const float *cal_lin /*[n_ch]*/,
                   const float *win, const float *wr, const float *wi, const float *thr, uint8_t *out)
{
    const size_t per_ch = (size_t)(n_lines - 1) * hop + NFFT;
    for (uint32_t c = 0; c < n_ch; c++)
        for (uint32_t k = 0; k < n_lines; k++)
            twin_wf_line(iq + (c * per_ch + (size_t)k * hop) * 2, win, wr, wi, thr, cal_lin[c],
                         out + ((size_t)k * n_ch + c) * NFFT);
}

/* -------------------------------------------------------------------- zoom */
/* The zoom stage in front of the waterfall (ssdr_set_wf_zoom): per channel z[n] = x[n] conj(P(phi0 + n dphi)) with the phasor
 * of EVERY sample taken from its absolute phase (no recurrence), y[m] = sum_k h[k] z[Z m - k] as an fma chain from zero, k
 * ascending, out = saturate(rint(y)) as int16 I,Q.  hist = the ZHIST raw samples before the call's first (oldest first). */
#define ZHIST 256
void twin_zoom(const int16_t *iq /*[n_ch][n_in][2]*/, uint32_t n_ch, uint32_t n_in, uint32_t Z, const uint32_t *dphi,
               const float *taps, uint32_t ntap, uint32_t *phase, int16_t *hist /*[n_ch][ZHIST][2]*/, int16_t *out /*[n_ch][n_in/Z][2]*/)
{
    float *zr = (float *)malloc((ZHIST + (size_t)n_in) * sizeof(float)), *zi = (float *)malloc((ZHIST + (size_t)n_in) * sizeof(float));
    for (uint32_t c = 0; c < n_ch; c++) {
        const int16_t *x = iq + (size_t)c * n_in * 2;
        int16_t *h = hist + (size_t)c * ZHIST * 2;
        for (int64_t n = -ZHIST; n < (int64_t)n_in; n++) {
            const int16_t *s = n < 0 ? h + 2 * (ZHIST + n) : x + 2 * n;
            float co, si;
            phasor32(phase[c] + (uint32_t)(int32_t)n * dphi[c], &co, &si);
            const float xr = (float)s[0], xi = (float)s[1];
            zr[ZHIST + n] = fmaf(xr, co, xi * si);
            zi[ZHIST + n] = fmaf(xi, co, -(xr * si));
        }
        for (uint32_t m = 0; m < n_in / Z; m++) {
            float ar = 0.0f, ai = 0.0f;
            for (uint32_t k = 0; k < ntap; k++) {
                ar = fmaf(taps[k], zr[ZHIST + (size_t)Z * m - k], ar);
                ai = fmaf(taps[k], zi[ZHIST + (size_t)Z * m - k], ai);
            }
            float yr = fminf(fmaxf(rintf(ar), -32768.0f), 32767.0f), yi = fminf(fmaxf(rintf(ai), -32768.0f), 32767.0f);
            out[((size_t)c * (n_in / Z) + m) * 2] = (int16_t)(int32_t)yr;
            out[((size_t)c * (n_in / Z) + m) * 2 + 1] = (int16_t)(int32_t)yi;
        }
        memcpy(h, x + 2 * ((size_t)n_in - ZHIST), ZHIST * 2 * sizeof(int16_t));          /* n_in >= ZHIST */
        phase[c] += n_in * dphi[c];
    }
    free(zr); free(zi);
}

/* ------------------------------------------------------------------- audio */
typedef struct {            /* per-channel kernel constants; same layout as the product's */
    uint32_t mode;          /* 0 am, 1 lsb, 2 usb, 3 cw, 4 nbfm */
    uint32_t ntap8;         /* taps in use, rounded up to a multiple of 8 (zero padded) */
    uint32_t dphi1, dphi2;  /* NCO steps: input mixer, SSB re-mixer */
    float wf_cal_lin, smeter_cal_db;
    float agc_c0, agc_c1, agc_knee, agc_delta8;
    uint32_t hang_frames, ntap;
    uint32_t tap_groups;    /* unused here: zero taps are exact no-ops in the fma chain */
    uint32_t fir_flags;     /* bit 0: the filter is a pure 4-sample delay (one unit tap at index 4) */
    uint32_t decim;         /* D (0 or 1: none): the IQ arrives at D * 12 kHz; taps are stream-major, ntap8 per stream */
    float kfm;              /* NBFM: output per radian, 16384 * rate / (2 pi 5000) */
} twin_consts;              /* 64 bytes */
#define FIR_DELAY4 1u

typedef struct {
    uint32_t phi1, phi2;
    float dc, agc_d;
    float agc_m[8];
    float prev_re, prev_im;
    uint32_t pad[2];
} twin_state;               /* 64 bytes */

/* P(x) = cos/sin(2 pi x / 2^32) at all 32 bits of the phase: 20-bit evaluation + first-order term for the low 12 bits */
static void phasor32(uint32_t ph, float *c, float *s)
{
    const float C_2PI_32 = 0x1.921fb6p-30f;         /* float32(2*pi/2^32) */
    float c20, s20;
    sincos20(ph, &c20, &s20);
    float eps = (float)(ph & 0xFFFu) * C_2PI_32;
    *c = fmaf(-s20, eps, c20);
    *s = fmaf(c20, eps, s20);
}

/* The NCO: the phasor of sample 8 b + j of a frame that starts at phase phi is P(phi) * P(8 b dphi) * S^j, S = P(dphi)
 * -- the ideal oscillator in real arithmetic, three fp32 phasors multiplied here.  block_phasor = the first two. */
static void block_phasor_n(uint32_t frame_phase, uint32_t dphi, int first_sample, float *c, float *s)
{
    float fc, fs, qc, qs;
    phasor32(frame_phase, &fc, &fs);
    phasor32((uint32_t)first_sample * dphi, &qc, &qs);
    *c = fmaf(fc, qc, -(fs * qs));
    *s = fmaf(fs, qc, fc * qs);
}
static void block_phasor(uint32_t frame_phase, uint32_t dphi, int b /*0..63*/, float *c, float *s)
{
    float fc, fs, qc, qs;
    phasor32(frame_phase, &fc, &fs);
    phasor32((uint32_t)(8 * b) * dphi, &qc, &qs);
    *c = fmaf(fc, qc, -(fs * qs));
    *s = fmaf(fs, qc, fc * qs);
}

static void mixn(const int16_t *x /*[n][2]*/, int n, float c, float s, float cs, float ss, float *zr, float *zi)
{
    for (int j = 0; j < n; j++) {
        float xr = (float)x[2 * j], xi = (float)x[2 * j + 1];
        zr[j] = fmaf(xr, c, xi * s);
        zi[j] = fmaf(xi, c, -(xr * s));
        float cn = fmaf(c, cs, -(s * ss)), sn = fmaf(s, cs, c * ss);
        c = cn; s = sn;
    }
}
static void mix8(const int16_t *x /*[8][2]*/, float c, float s, float cs, float ss, float *zr, float *zi)
{
    for (int j = 0; j < 8; j++) {
        float xr = (float)x[2 * j], xi = (float)x[2 * j + 1];
        zr[j] = fmaf(xr, c, xi * s);
        zi[j] = fmaf(xi, c, -(xr * s));
        float cn = fmaf(c, cs, -(s * ss)), sn = fmaf(s, cs, c * ss);
        c = cn; s = sn;
    }
}

/* The 64-lane inclusive scan "as the machine does it": Kogge-Stone inside each row of 16 lanes
 * (distances 1, 2, 4, 8), then lane 15 of rows 0/2 into every lane of rows 1/3, then lane 31 into
 * every lane of rows 2 and 3.  scan_src(step, lane) = source lane, or -1 if the lane sits out
 * (it then combines with the identity, which leaves its value unchanged exactly). */
static int scan_src(int step, int l)
{
    if (step < 4) { int d = 1 << step; return ((l & 15) >= d) ? l - d : -1; }
    if (step == 4) return (((l >> 4) & 1) == 1) ? (l & ~15) - 1 : -1;
    return (l >= 32) ? 31 : -1;
}

static const float DC_A = 0.9921875f, DC_AL = 0.0078125f;
/* float32((127/128)^(j+1)), j = 0..7 */
static const float DC_APOW[8] = { 0x1.fcp-1f, 0x1.f808p-1f, 0x1.f417fp-1f, 0x1.f02fcp-1f,
                                  0x1.ec4f6p-1f, 0x1.e876c2p-1f, 0x1.e4a5d4p-1f, 0x1.e0dc88p-1f };
static const float P_FLOOR = 9.5367431640625e-07f;   /* 2^-20 */

static void audio_frame(const int16_t *iq /*[512][2]*/, const twin_consts *c, const float *taps,
                        twin_state *st, int16_t *hist /*[HIST][2]*/, int16_t *pcm, float *rssi, uint8_t *flag,
                        int16_t *iq_out /*[512][2] or NULL: mode 5, I,Q of the filtered baseband under the AGC gain*/)
{
    static _Thread_local float z1r[HIST + FRAME * 4], z1i[HIST + FRAME * 4];
    float z2r[FRAME], z2i[FRAME], p[FRAME], aud[FRAME];
    const int D = c->decim > 1 ? (int)c->decim : 1;
    /* 1. NCO mix of history + frame (the history blocks are blocks 48..63 of the previous frame, mixed as that frame did) */
    float cs1, ss1, cs2, ss2;
    phasor32(c->dphi1, &cs1, &ss1);
    phasor32(c->dphi2, &cs2, &ss2);
    if (D > 1) {
        /* Decimating front end: the IQ arrives at D * 12 kHz, a frame is 512 D inputs, "lane" l owns inputs 8 D l ..: one
         * block phasor and a rotation chain per lane.  y[m] = sum_k h[k] z[D m - k], summed stream by stream: v_q[m] =
         * z[D m + q], stream q's taps at taps[q * 128/D ...] (stream q >= 1 behind one zero tap), taps ascending. */
        const int NB = 8 * D, n_in = FRAME * D, slots = NTAP_MAX / D, tail_lanes = HIST / NB;
        for (int t = 0; t < tail_lanes; t++) {
            float bc, bs;
            block_phasor_n(st->phi1 - (uint32_t)n_in * c->dphi1, c->dphi1, NB * (64 - tail_lanes + t), &bc, &bs);
            mixn(hist + 2 * NB * t, NB, bc, bs, cs1, ss1, z1r + NB * t, z1i + NB * t);
        }
        for (int l = 0; l < NLANE; l++) {
            float bc, bs;
            block_phasor_n(st->phi1, c->dphi1, NB * l, &bc, &bs);
            mixn(iq + 2 * NB * l, NB, bc, bs, cs1, ss1, z1r + HIST + NB * l, z1i + HIST + NB * l);
        }
        for (int m = 0; m < FRAME; m++) {
            float ar = 0.0f, ai = 0.0f;
            for (int q = 0; q < D; q++)
                for (uint32_t i = 0; i < c->ntap8; i++) {
                    const int idx = HIST + D * (m - (int)i) + q;
                    ar = fmaf(taps[q * slots + i], z1r[idx], ar);
                    ai = fmaf(taps[q * slots + i], z1i[idx], ai);
                }
            z2r[m] = ar; z2i[m] = ai;
            p[m] = fmaf(ar, ar, ai * ai);
        }
    }
    for (int b = -HIST / 8; b < FRAME / 8 && D == 1; b++) {
        const int16_t *x = (b < 0) ? hist + 2 * (HIST + 8 * b) : iq + 2 * 8 * b;
        float bc, bs;
        if (b < 0) block_phasor(st->phi1 - (uint32_t)FRAME * c->dphi1, c->dphi1, 64 + b, &bc, &bs);
        else block_phasor(st->phi1, c->dphi1, b, &bc, &bs);
        mix8(x, bc, bs, cs1, ss1, z1r + HIST + 8 * b, z1i + HIST + 8 * b);
    }
    /* 2. FIR, taps ascending, fma chain from zero */
    for (int n = 0; n < FRAME && D == 1; n++) {
        float ar = 0.0f, ai = 0.0f;
        for (uint32_t k = 0; k < c->ntap8; k++) {
            ar = fmaf(taps[k], z1r[HIST + n - (int)k], ar);
            ai = fmaf(taps[k], z1i[HIST + n - (int)k], ai);
        }
        z2r[n] = ar; z2i[n] = ai;
        p[n] = fmaf(ar, ar, ai * ai);
    }
    /* AM behind a filter that is a pure delay: |x e^{j phi}| = |x|, so envelope, AGC level and RSSI do not depend on
     * the NCO.  The power of a sample is then taken exactly in integers, I*I + Q*Q (< 2^32), and rounded once.
     * (z2 keeps its mixed value: its last sample is the discriminator memory carried in the state.) */
    const int am_raw = (c->mode == 0) && (c->fir_flags & FIR_DELAY4) && D == 1;
    if (am_raw) {
        for (int n = 0; n < FRAME; n++) {
            const int16_t *x = (n < 4) ? hist + 2 * (HIST + n - 4) : iq + 2 * (n - 4);
            const uint32_t q = (uint32_t)((int32_t)x[0] * x[0]) + (uint32_t)((int32_t)x[1] * x[1]);
            p[n] = (float)q;
        }
    }
    /* ADC overflow (SND header flags bit 1, utils_supersdr.py:1066-1067): a sample of this frame at the rails */
    {
        int ovf = 0;
        for (int n = 0; n < 2 * FRAME * D; n++) ovf |= (iq[n] >= 32767) || (iq[n] <= -32767);
        *flag = (uint8_t)ovf;
    }
    /* 3. demod */
    if (c->mode == 0) {
        float env[FRAME], loc[NLANE][8], A[NLANE], B[NLANE], An[NLANE], Bn[NLANE];
        for (int n = 0; n < FRAME; n++) env[n] = sqrtf(p[n]);
        for (int l = 0; l < NLANE; l++) {
            float s = 0.0f;
            for (int j = 0; j < 8; j++) { s = fmaf(DC_A, s, DC_AL * env[8 * l + j]); loc[l][j] = s; }
            B[l] = s; A[l] = DC_APOW[7];
        }
        for (int step = 0; step < 6; step++) {      /* affine maps m -> A m + B, composed lane after source */
            for (int l = 0; l < NLANE; l++) {
                int src = scan_src(step, l);
                float Al = (src < 0) ? 1.0f : A[src], Bl = (src < 0) ? 0.0f : B[src];
                Bn[l] = fmaf(A[l], Bl, B[l]);
                An[l] = A[l] * Al;
            }
            memcpy(A, An, sizeof A); memcpy(B, Bn, sizeof B);
        }
        for (int l = 0; l < NLANE; l++) {
            float carry = (l == 0) ? fmaf(1.0f, st->dc, 0.0f) : fmaf(A[l - 1], st->dc, B[l - 1]);
            for (int j = 0; j < 8; j++) {
                float m = fmaf(DC_APOW[j], carry, loc[l][j]);
                aud[8 * l + j] = env[8 * l + j] - m;
                if (l == NLANE - 1 && j == 7) st->dc = m;
            }
        }
    } else if (c->mode <= 3) {
        for (int b = 0; b < FRAME / 8; b++) {
            float co, si;
            block_phasor(st->phi2, c->dphi2, b, &co, &si);
            for (int j = 0; j < 8; j++) {
                int n = 8 * b + j;
                aud[n] = fmaf(z2r[n], co, -(z2i[n] * si));
                float cn = fmaf(co, cs2, -(si * ss2)), sn = fmaf(si, cs2, co * ss2);
                co = cn; si = sn;
            }
        }
    } else if (c->mode == 5) {                       /* "iq": no demodulator, the PCM row carries I */
        for (int n = 0; n < FRAME; n++) aud[n] = z2r[n];
    } else {
        float pr = st->prev_re, pi = st->prev_im;
        for (int n = 0; n < FRAME; n++) {
            float dr = fmaf(z2r[n], pr, z2i[n] * pi);
            float di = fmaf(z2i[n], pr, -(z2r[n] * pi));
            aud[n] = atan2p(di, dr) * c->kfm;       /* 12 kHz: float32(16384*12000/(2*pi*5000)) = 6258.227 */
            pr = z2r[n]; pi = z2i[n];
        }
    }
    st->prev_re = z2r[FRAME - 1]; st->prev_im = z2i[FRAME - 1];
    /* 4. AGC per 8-sample block */
    float a[NLANE], e[NLANE], psum[NLANE];
    float amax = -3.0e38f;
    for (int l = 0; l < NLANE; l++) {
        float pm = p[8 * l], s = p[8 * l];
        for (int j = 1; j < 8; j++) { pm = fmaxf(pm, p[8 * l + j]); s = s + p[8 * l + j]; }
        psum[l] = s;
        a[l] = log2p(fmaxf(pm, P_FLOOR));
        amax = fmaxf(amax, a[l]);
    }
    uint32_t K = c->hang_frames;
    float d8 = c->agc_delta8;
    if (K == 0) {
        float P = -3.0e38f;
        for (int l = 0; l < NLANE; l++) {
            P = fmaxf(P, fmaf((float)l, d8, a[l]));
            e[l] = fmaxf(fmaf(-(float)l, d8, P), fmaf(-(float)(l + 1), d8, st->agc_d));
        }
        st->agc_d = e[NLANE - 1];
    } else {
        float maxM = st->agc_m[0], P = -3.0e38f;
        for (uint32_t i = 1; i < K; i++) maxM = fmaxf(maxM, st->agc_m[i]);
        for (int l = 0; l < NLANE; l++) {
            P = fmaxf(P, a[l]);
            e[l] = fmaxf(fmaxf(P, maxM), fmaf(-(float)(l + 1), d8, st->agc_d));
        }
        st->agc_d = fmaxf(fmaf(-64.0f, d8, st->agc_d), st->agc_m[K - 1]);
        for (int i = 7; i > 0; i--) st->agc_m[i] = st->agc_m[i - 1];
        st->agc_m[0] = amax;
    }
    for (int l = 0; l < NLANE; l++) {
        float g = exp2p(fmaf(c->agc_c1, fmaxf(e[l], c->agc_knee), c->agc_c0));
        for (int j = 0; j < 8; j++) {
            float y = rintf(aud[8 * l + j] * g);
            y = fminf(fmaxf(y, -32768.0f), 32767.0f);
            pcm[8 * l + j] = (int16_t)(int32_t)y;
            if (iq_out && c->mode == 5) {
                float q = rintf(z2i[8 * l + j] * g);
                q = fminf(fmaxf(q, -32768.0f), 32767.0f);
                iq_out[2 * (8 * l + j)] = pcm[8 * l + j];
                iq_out[2 * (8 * l + j) + 1] = (int16_t)(int32_t)q;
            }
        }
    }
    /* 5. rssi: inclusive sum scan over lanes (same six steps), total = lane 63 */
    for (int step = 0; step < 6; step++) {
        float t[NLANE];
        for (int l = 0; l < NLANE; l++) {
            int src = scan_src(step, l);
            t[l] = psum[l] + ((src < 0) ? 0.0f : psum[src]);
        }
        memcpy(psum, t, sizeof t);
    }
    *rssi = fmaf(log2p(fmaxf(psum[NLANE - 1], 1e-20f)) - 39.0f, 0x1.815182p+1f /* 10*log10(2) */, c->smeter_cal_db);
    /* 6. state carry */
    st->phi1 += (uint32_t)(FRAME * D) * c->dphi1;
    st->phi2 += (uint32_t)FRAME * c->dphi2;
    /* HIST <= FRAME: the new history is the frame tail */
    memcpy(hist, iq + 2 * (FRAME * D - HIST), HIST * 2 * sizeof(int16_t));
}

/* batch: iq[n_ch][n_frames*512][2]; consts[n_ch]; taps[n_ch][128]; state[n_ch]; hist[n_ch][128][2]
 * -> pcm[n_ch][n_frames*512], rssi[n_ch][n_frames]; state and hist updated in place. */
void twin_audio3(const int16_t *iq, uint32_t n_ch, uint32_t n_frames, const twin_consts *consts,
                 const float *taps, twin_state *state, int16_t *hist, int16_t *pcm, float *rssi, uint8_t *flags,
                 int16_t *iq_out /*[n_ch][n_frames*512][2] or NULL; rows of channels not in mode 5 are left alone*/)
{
    uint8_t dummy;
    const size_t D = consts[0].decim > 1 ? consts[0].decim : 1;     /* one input rate per batch */
    for (uint32_t c = 0; c < n_ch; c++)
        for (uint32_t f = 0; f < n_frames; f++)
            audio_frame(iq + ((size_t)c * n_frames + f) * FRAME * D * 2, consts + c,
                        taps + (size_t)c * NTAP_MAX, state + c, hist + (size_t)c * HIST * 2,
                        pcm + ((size_t)c * n_frames + f) * FRAME, rssi + (size_t)c * n_frames + f,
                        flags ? flags + (size_t)c * n_frames + f : &dummy,
                        iq_out ? iq_out + ((size_t)c * n_frames + f) * FRAME * 2 : NULL);
}

void twin_audio2(const int16_t *iq, uint32_t n_ch, uint32_t n_frames, const twin_consts *consts,
                 const float *taps, twin_state *state, int16_t *hist, int16_t *pcm, float *rssi, uint8_t *flags)
{
    twin_audio3(iq, n_ch, n_frames, consts, taps, state, hist, pcm, rssi, flags, NULL);
}

void twin_audio(const int16_t *iq, uint32_t n_ch, uint32_t n_frames, const twin_consts *consts,
                const float *taps, twin_state *state, int16_t *hist, int16_t *pcm, float *rssi)
{
    twin_audio2(iq, n_ch, n_frames, consts, taps, state, hist, pcm, rssi, NULL);
}
