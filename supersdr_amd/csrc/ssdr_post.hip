// ssdr_post.hip -- the reference's own post-processing of the two streams, on the GPU:
//
//   ssdr_db2col_kernel   kiwi_waterfall.spectrum_db2col (utils_supersdr.py:787-813): byte line -> dBm ->
//                        40th-percentile / max autoscale -> palette index 0..254, float32 op for op as
//                        NumPy evaluates it on a float32 line
//   ssdr_play_kernel     kiwi_sound.play_buffer (utils_supersdr.py:1106-1148): volume, x4 zero-stuffing
//                        interpolation with the reference's 33-tap filter (:999-1005), pan^2, truncating
//                        int16 stereo pack -- float64 like the reference
//   ssdr_play_rs_kernel  the same for 20.25 kHz KiwiSDRs (:1125-1126): resample_poly(popped, 64, 27, "line")[:-1]
//   ssdr_trace_kernel    display_stuff.plot_spectrum's reduction (utils_supersdr.py:1678-1679) over the device copy
//                        of kiwi_waterfall.wf_data's newest rows; ssdr_smeter_kernel: the S-meter smoothing of the
//                        main loop (supersdr.py:936-947)
//   ssdr_iqwire_kernel   KiwiSDRStream._process_aud, IQ branch (kiwi/client.py:443-454): strips the 17-byte
//                        SND/GPS header of each frame and turns big-endian int16 I,Q into the kernels' layout
//
// These are pinned by golden vectors produced by the real reference (tests/golden/*.npz).
#include "ssdr_math.h"
#include "ssdr_kernels.h"
#include "ssdr_resample_taps.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTRL, int ROWMASK>
SSDR_DEV int dpp_i(int identity, int x)
{
    return __builtin_amdgcn_update_dpp(identity, x, CTRL, ROWMASK, 0xF, false);
}
SSDR_DEV int wave_sum_i(int x)          // total over the 64 lanes, returned to every lane (uniform)
{
    x += dpp_i<0x111, 0xF>(0, x);
    x += dpp_i<0x112, 0xF>(0, x);
    x += dpp_i<0x114, 0xF>(0, x);
    x += dpp_i<0x118, 0xF>(0, x);
    x += dpp_i<0x142, 0xA>(0, x);
    x += dpp_i<0x143, 0xC>(0, x);
    return __builtin_amdgcn_readlane(x, 63);
}
SSDR_DEV int wave_max_i(int x)
{
    x = max(x, dpp_i<0x111, 0xF>(x, x));
    x = max(x, dpp_i<0x112, 0xF>(x, x));
    x = max(x, dpp_i<0x114, 0xF>(x, x));
    x = max(x, dpp_i<0x118, 0xF>(x, x));
    x = max(x, dpp_i<0x142, 0xA>(x, x));
    x = max(x, dpp_i<0x143, 0xC>(x, x));
    return __builtin_amdgcn_readlane(x, 63);
}
SSDR_DEV int wave_min_i(int x) { return -wave_max_i(-x); }

// One wave per (channel, line) -- lines are independent of one another: with auto-scaling a line's clip levels come from the
// line itself, without it they are the caller's and stay what they are.
//
// NumPy on a float32 line (restated, utils_supersdr.py:787-813):
//   wf_db   = (-(255 - s) - 13) + 3 zoom               three float32 roundings
//   wf_db[0] = wf_db[1]
//   low     = a + (b - a) * G,  a, b = sorted[409], sorted[410], G = float32(0.2000122)   (np.percentile(.,40)
//             on float32: q = 40/float32(100), virtual index 1024 q + (1 - q) - 1 in float32 -> gamma G)
//   high    = max;  dyn = max(high - low, 40)
//   color   = clip(clip((wf_db - (low + dlo)) / ((dyn + dhi) - dlo), 0, 1) * 254, 0, 255)
// The order statistics are taken on the int16 sums (the map sum -> wf_db is monotone), by bisection on the value over
// [0, 255 N] -- 8 steps at N = 1, 12 at N = 10 -- with the count taken by wave-wide compares (v_cmp writes a lane mask to
// scalar registers, s_bcnt1 counts it: no cross-lane reduction, the decision is scalar): exact, no sort.
// The 1024 divisions of a line share their divisor: where it is an ordinary number the quotients are formed with the
// correctly rounded division's own refinement steps on ONE reciprocal (the sequence a float32 divide compiles to, minus
// its per-quotient reciprocal and scaling); a zero, huge, tiny or non-finite divisor takes the plain divide.
// num / den correctly rounded, given r = refined_rcp(den): the refinement steps of the float32 divide (what `/` compiles to on
// gfx950: rcp, one Newton step, quotient, two residual corrections) with the reciprocal shared by all quotients of one divisor.
// Valid where nothing leaves the normal range: |den| in [2^-40, 2^40], num = 0 or 2^-20 <= |num| < 2^24.
SSDR_DEV float refined_rcp(float den)
{
    const float r = __builtin_amdgcn_rcpf(den);
    return fmaf(fmaf(-den, r, 1.0f), r, r);
}
SSDR_DEV float div_shared(float num, float den, float r)
{
    float c = num * r;
    c = fmaf(fmaf(-den, c, num), r, c);
    return fmaf(fmaf(-den, c, num), r, c);
}

SSDR_DEV int wave_count_le(const int (&s)[16], int mid)
{
    int c = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) c += __builtin_popcountll(__ballot(s[i] <= mid));
    return c;
}

__global__ __launch_bounds__(256) void ssdr_db2col_kernel(SsdrDb2colArgs a)
{
    const int l = threadIdx.x & 63;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (uint64_t)a.n_sel * a.n_lines) return;
    const uint32_t line = (uint32_t)(item / a.n_sel), pos = (uint32_t)(item - (uint64_t)line * a.n_sel);      // line-major, like the data
    const uint32_t ch = a.sel ? a.sel[pos] : pos;
    ssdr_db2col_chan st = a.chans[pos];
    const float z3 = (float)(3 * st.zoom), dlo = (float)st.delta_low_db, dhi = (float)st.delta_high_db;
    const float fn = (float)a.n_avg;
    const float G = 0x1.99ap-3f;

    // lane l owns bins 256 q + 4 l .. + 3, q = 0..3: every load instruction reads 512 contiguous bytes of the line, every store
    // writes 1 KB of the colour line (the order statistics do not care who holds which bin)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 *src = reinterpret_cast<const u32x2 *>(a.wf + ((uint64_t)line * a.n_ch + ch) * SSDR_NFFT) + l;
    int s[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const u32x2 r = __builtin_nontemporal_load(src + 64 * q);
        s[4 * q] = (int)(r.x & 0xFFFFu); s[4 * q + 1] = (int)(r.x >> 16);
        s[4 * q + 2] = (int)(r.y & 0xFFFFu); s[4 * q + 3] = (int)(r.y >> 16);
    }
    if (l == 0) s[0] = s[1];                                    // "first bin is broken" (:791)

    const float rn = refined_rcp(fn);                            // N in 1..100, sums in 0..25500: float32(sum) / float32(N) == np.mean
    float wf_db[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float sp = div_shared((float)s[i], fn, rn);
        wf_db[i] = (-(255.0f - sp) - 13.0f) + z3;
    }

    if (st.auto_scale) {
        // smallest v with #{s <= v} >= 410  ==  sorted[409]; the sums of N byte lines lie in [0, 255 N]
        int lo = 0, hi = min(255 * (int)a.n_avg, 32767);
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (wave_count_le(s, mid) >= 410) hi = mid; else lo = mid + 1;
        }
        const int v409 = lo;
        int above = 0x7FFF, mx = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            above = min(above, (s[i] > v409) ? s[i] : 0x7FFF);
            mx = max(mx, s[i]);
        }
        const int v410 = (wave_count_le(s, v409) >= 411) ? v409 : wave_min_i(above);
        const int vmax = wave_max_i(mx);
        const float fa = (-(255.0f - div_shared((float)v409, fn, rn)) - 13.0f) + z3;
        const float fb = (-(255.0f - div_shared((float)v410, fn, rn)) - 13.0f) + z3;
        const float fh = (-(255.0f - div_shared((float)vmax, fn, rn)) - 13.0f) + z3;
        const float d = fb - fa;
        const float t = d * G;
        st.low_clip_db = fa + t;
        st.high_clip_db = fh;
        const float span = fh - st.low_clip_db;
        st.dynamic_range = (span > 40.0f) ? span : 40.0f;
    }
    const float lo2 = st.low_clip_db + dlo;
    const float nf = st.dynamic_range + dhi;
    const float den = nf - dlo;
    st.wf_min_db = lo2 - z3;
    st.wf_max_db = (st.low_clip_db + nf) - z3;

    f32x4 *dst0 = reinterpret_cast<f32x4 *>(a.color + ((uint64_t)line * a.n_sel + pos) * SSDR_NFFT) + l;
    const float aden = fabsf(den);
    if (aden >= 0x1p-40f && aden <= 0x1p40f) {                  // wave-uniform
        // num / den, correctly rounded: r = rcp refined once; q = num r refined twice against the exact residual.  |num| < 2^20 and is
        // 0 or >= 2^-17, so no step leaves the normal range and the divide's scaling (v_div_scale / v_div_fixup) has nothing to do
        const float r = refined_rcp(den);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float c = div_shared(wf_db[4 * q + i] - lo2, den, r);
                c = fminf(fmaxf(c, 0.0f), 1.0f) * 254.0f;
                o[i] = fminf(fmaxf(c, 0.0f), 255.0f);
            }
            __builtin_nontemporal_store(o, dst0 + 64 * q);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float c = (wf_db[4 * q + i] - lo2) / den;
                c = fminf(fmaxf(c, 0.0f), 1.0f) * 254.0f;
                o[i] = fminf(fmaxf(c, 0.0f), 255.0f);
            }
            __builtin_nontemporal_store(o, dst0 + 64 * q);
        }
    }
    // the display state as the LAST line leaves it (utils_supersdr.py:795-808); only the fields spectrum_db2col writes.
    // INVARIANT (the waves of a channel's other lines read a.chans[pos] at their start, unordered against this store): every field
    // written here is either recomputed by each wave from its own line (auto_scale: low / high / dynamic_range, wf_min / wf_max) or
    // -- without auto_scale -- written back with the value it was read with; the fields a wave reads and does not recompute (zoom,
    // auto_scale, the deltas) are never written.  So a wave sees the same inputs whichever side of the store it starts on.  State
    // that is CARRIED from line to line would break this: it needs a second array (as ssdr_play_kernel's hist / hist_out).
    if (l == 0 && line + 1 == a.n_lines) {
        ssdr_db2col_chan *o = a.chans + pos;
        o->low_clip_db = st.low_clip_db; o->high_clip_db = st.high_clip_db; o->dynamic_range = st.dynamic_range;
        o->wf_min_db = st.wf_min_db; o->wf_max_db = st.wf_max_db;
    }
}

// ---------------------------------------------------------------------------------------------------------
// play_buffer: out[4m + r] for r = 0..3.  With zero stuffing only every 4th term of the 33-tap convolution is non-zero:
//   y[4m + r] = 4 * sum_t h[r + 4t] X[m + 8 - t],  X = [8 carried samples | frame], all scaled by the volume --
// phase 0 has nine terms, phases 1..3 eight -- summed in ascending sample order like np.convolve, multiply then add in
// float64.  One wave per (channel, frame): the carried samples of a frame are the previous frame's last eight (the ctx's
// history for the first frame of a call; the last frame's tail goes to `hist_out`, a different buffer).  Lane l takes input
// positions m = 64 k + l: the nine samples it needs are nine conflict-free LDS reads shared by its four outputs, which
// leave as ONE 16-byte store -- every store instruction of the wave writes 1 KB of contiguous output.
// PAN: which of the two pan factors is not 1.0 (0: neither, 1: left, 2: right, 3: both).  min(1 -+ balance, 1)^2 leaves at most one
// of them below 1, and x * 1.0 == x: the unscaled side (and the recording branch's block) is the truncated sum itself.
// The taps arrive multiplied by SAMPLE_RATIO = 4 (`h4`): a power of two commutes with every rounding of the sum (np.convolve's
// products and partial sums are nowhere near the subnormals), so "* self.SAMPLE_RATIO" (:1134) costs nothing per output.
template <int PAN>
SSDR_DEV void play_frame(const double *s_x, const double (&h4)[33], double l2, double r2, int l, u32x4 *dst, int16_t *mono)
{
#pragma unroll 2
    for (int k = 0; k < 8; k++) {
        const int m = 64 * k + l;
        double x[9];
#pragma unroll
        for (int i = 0; i < 9; i++) x[i] = s_x[m + i];
        uint32_t o[4];
        int mo[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double acc = 0.0;
#pragma unroll
            for (int t = 8; t >= 0; t--) {
                const int j = r + 4 * t;
                if (j <= 32) acc += h4[j] * x[8 - t];
            }
            const int mi = (int)acc;                                      // trunc toward zero, then wrap to int16
            const int li = (PAN & 1) ? (int)(acc * l2) : mi, ri = (PAN & 2) ? (int)(acc * r2) : mi;
            o[r] = ((uint32_t)li & 0xFFFFu) | ((uint32_t)ri << 16);
            mo[r] = mi;
        }
        __builtin_nontemporal_store(u32x4{o[0], o[1], o[2], o[3]}, dst + m);
        if (mono) {                                                       // recording branch (:1139-1140)
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = {((uint32_t)mo[0] & 0xFFFFu) | ((uint32_t)mo[1] << 16), ((uint32_t)mo[2] & 0xFFFFu) | ((uint32_t)mo[3] << 16)};
            *reinterpret_cast<u32x2 *>(mono + 4 * m) = v;
        }
    }
}

__global__ __launch_bounds__(256) void ssdr_play_kernel(SsdrPlayArgs a)
{
    __shared__ double s_xw[4][8 + SSDR_FRAME];                      // per wave: 8 carried samples + the frame, volume applied
    const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + wave;
    if (item >= (uint64_t)a.n_sel * a.n_frames) return;
    const uint32_t pos = (uint32_t)(item / a.n_frames), f = (uint32_t)(item - (uint64_t)pos * a.n_frames);
    const uint32_t ch = a.sel ? a.sel[pos] : pos;
    double *s_x = s_xw[wave];
    const ssdr_play_chan pc = a.chans[pos];
    const double vol = pc.volume / 100.0;
    const double lv = fmin(1.0 - pc.balance, 1.0), rv = fmin(1.0 + pc.balance, 1.0);
    const double l2 = lv * lv, r2 = rv * rv;
    double h4[33];
#pragma unroll
    for (int j = 0; j < 33; j++) h4[j] = a.taps[j];                // uniform address: scalar loads (the host uploads 4 h)

    const int16_t *src = a.pcm + ((uint64_t)ch * a.n_frames + f) * SSDR_FRAME;
#pragma unroll
    for (int k = 0; k < 8; k++) s_x[8 + 64 * k + l] = (double)src[64 * k + l] * vol;
    if (l < 8) s_x[l] = f ? (double)src[l - 8] * vol : a.hist[(size_t)ch * 8 + l];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    u32x4 *dst = reinterpret_cast<u32x4 *>(a.out + ((uint64_t)pos * a.n_frames + f) * 2048 * 2);
    int16_t *mono = a.mono ? a.mono + ((uint64_t)pos * a.n_frames + f) * 2048 : nullptr;
    const int pan = __builtin_amdgcn_readfirstlane((l2 != 1.0 ? 1 : 0) | (r2 != 1.0 ? 2 : 0));      // per channel: wave-uniform
    if (pan == 0) play_frame<0>(s_x, h4, l2, r2, l, dst, mono);
    else if (pan == 1) play_frame<1>(s_x, h4, l2, r2, l, dst, mono);
    else if (pan == 2) play_frame<2>(s_x, h4, l2, r2, l, dst, mono);
    else play_frame<3>(s_x, h4, l2, r2, l, dst, mono);
    if (f + 1 == a.n_frames && l < 8) a.hist_out[(size_t)ch * 8 + l] = s_x[SSDR_FRAME + l];
}

// ---------------------------------------------------------------------------------------------------------
// play_buffer for a fractional SAMPLE_RATIO (20.25 kHz KiwiSDRs, utils_supersdr.py:1125-1126): every 512-sample
// frame goes through scipy.signal.resample_poly(popped, 64, 27, padtype="line") on its own (no history) and the
// last output is dropped: 1213 stereo samples per frame.  Restated from scipy 1.15.3 (_upfirdn_apply.pyx,
// _apply_impl with MODE_LINE): output y of the full polyphase convolution uses phase t = 27 y mod 64 ending at
// sample 27 y div 64; its 21 taps are applied oldest sample first, one multiply and one add each in float64;
// samples outside the frame lie on the line through the frame's first and last sample; outputs
// y = 24 .. 24 + 1212 are kept.
// One wave per (channel, frame) at a time, persistent grid.  Output k = p + 64 j of lane p has phase 27 (p + 24) mod 64 for
// every j: a lane's 21 taps are constants of the launch and live in registers; the frame, extended along the line on both
// sides, lies in LDS as doubles, so the tap loop is 21 branch-free read / multiply / add steps; the lanes' outputs of one j
// are 64 consecutive samples (a permutation of them): each store instruction fills 256 contiguous bytes.
constexpr int RS_LEFT = SSDR_RS_HPP - 1;                                                     // 20 samples before the frame
constexpr int RS_XMAX = ((SSDR_RS_OUT_PER_FRAME - 1 + SSDR_RS_PRE_REMOVE) * SSDR_RS_DOWN) / SSDR_RS_UP;   // last sample index touched: 521
constexpr int RS_EXT = RS_LEFT + RS_XMAX + 1;                                                // 542 doubles

// The frame lies in LDS twice, the second copy shifted by one sample: whatever the parity of a window's first sample, one of
// the two copies holds it 16-byte aligned, and the 21 samples arrive as eleven ds_read_b128 (left to itself the compiler pairs
// 8-byte reads into ds_read2_b64, which moves half as many bytes per LDS cycle -- the kernel was bound by exactly that).
constexpr int RS_PITCH = 560;                                                               // doubles per copy: even, >= RS_EXT + 1, and 2 * 560 dwords = 32 (mod 64 banks):
                                                                                            // the two copies of a sample sit half the banks apart
static_assert(RS_PITCH >= RS_EXT + 1 && RS_PITCH % 2 == 0, "copy pitch");

__global__ __launch_bounds__(256) void ssdr_play_rs_kernel(SsdrPlayArgs a)
{
    __shared__ __attribute__((aligned(16))) double s_xw[4][2 * RS_PITCH];
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int p = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *s_a = s_xw[wave], *s_b = s_a + RS_PITCH;               // s_a[i] = X[i], s_b[i] = X[i + 1]
    const int total0 = (p + SSDR_RS_PRE_REMOVE) * SSDR_RS_DOWN;
    const int t = total0 % SSDR_RS_UP, x0_idx = total0 / SSDR_RS_UP;       // phase of all of this lane's outputs; end sample of its first
    double h[SSDR_RS_HPP];
#pragma unroll
    for (int m = 0; m < SSDR_RS_HPP; m++) h[m] = a.rs_taps[t * SSDR_RS_HPP + m];
    const uint64_t n_items = (uint64_t)a.n_sel * a.n_frames;
    auto put = [&](int i, double v) { s_a[i] = v; if (i) s_b[i - 1] = v; };
    auto frame_of = [&](uint64_t it) -> uint64_t {                  // item (position, frame) -> its PCM frame (channel, frame)
        const uint32_t ps = (uint32_t)(it / a.n_frames), fr = (uint32_t)(it - (uint64_t)ps * a.n_frames);
        return (uint64_t)(a.sel ? a.sel[ps] : ps) * a.n_frames + fr;
    };
    const uint64_t stride = (uint64_t)gridDim.x * 4;
    uint64_t item = (uint64_t)blockIdx.x * 4 + wave;
    // the next frame's samples are fetched while this one is computed (a frame is 1 KB: one memory round trip per frame
    // would otherwise stand in front of every 1.7 us of arithmetic)
    int16_t cur[8], nxt[8];
    if (item < n_items) {
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = a.pcm[frame_of(item) * SSDR_FRAME + 64 * k + p];
    }
    for (; item < n_items; item += stride) {
        const uint32_t pos = (uint32_t)(item / a.n_frames), f = (uint32_t)(item - (uint64_t)pos * a.n_frames);
        const uint64_t nframe = frame_of(item + stride < n_items ? item + stride : item);
#pragma unroll
        for (int k = 0; k < 8; k++) nxt[k] = a.pcm[nframe * SSDR_FRAME + 64 * k + p];
        const ssdr_play_chan pc = a.chans[pos];
        const double vol = pc.volume / 100.0;
        const double lv = fmin(1.0 - pc.balance, 1.0), rv = fmin(1.0 + pc.balance, 1.0);
        const double l2 = lv * lv, r2 = rv * rv;
#pragma unroll
        for (int k = 0; k < 8; k++) put(RS_LEFT + 64 * k + p, (double)cur[k] * vol);
#pragma unroll
        for (int k = 0; k < 8; k++) cur[k] = nxt[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double x0 = s_a[RS_LEFT], xl = s_a[RS_LEFT + SSDR_FRAME - 1];
        const double slope = (xl - x0) / (double)(SSDR_FRAME - 1);
        if (p < RS_LEFT) put(p, x0 + (double)(p - RS_LEFT) * slope);                              // xi = p - 20 < 0
        else if (p < RS_LEFT + RS_XMAX + 1 - SSDR_FRAME) {                                         // xi = 512 .. 521
            const int xi = SSDR_FRAME + (p - RS_LEFT);
            put(RS_LEFT + xi, xl + (double)(xi - SSDR_FRAME + 1) * slope);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t *dst = reinterpret_cast<uint32_t *>(a.out + ((uint64_t)pos * a.n_frames + f) * SSDR_RS_OUT_PER_FRAME * 2);
        int16_t *mono = a.mono ? a.mono + ((uint64_t)pos * a.n_frames + f) * SSDR_RS_OUT_PER_FRAME : nullptr;
        constexpr int NJ = (SSDR_RS_OUT_PER_FRAME + 63) / 64;                                        // 19
        // per output: all eleven reads are issued, then the 21-term chain runs (left to the scheduler every read is followed by
        // its own wait, and the wave sits out the LDS latency eleven times per output)
#pragma unroll 1
        for (int j = 0; j < NJ; j++) {
            const int k = p + 64 * j;
            // window of output k: samples x_idx - 20 .. x_idx, x_idx = x0_idx + 27 j  ->  X[s + m], m = 0..20, s = x0_idx + 27 j
            const int s = min(x0_idx + SSDR_RS_DOWN * j, RS_XMAX);
            const f64x2 *xw = reinterpret_cast<const f64x2 *>((s & 1) ? s_b + (s - 1) : s_a + s);
            double x[22];
#pragma unroll
            for (int m = 0; m < 11; m++) { const f64x2 v = xw[m]; x[2 * m] = v.x; x[2 * m + 1] = v.y; }
            __builtin_amdgcn_sched_barrier(0);
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < SSDR_RS_HPP; m++) acc = acc + x[m] * h[m];
            __builtin_amdgcn_sched_barrier(0);
            if (k < SSDR_RS_OUT_PER_FRAME) {
                const int li = (int)(acc * l2), ri = (int)(acc * r2);                                // trunc toward zero, then wrap to int16
                dst[k] = ((uint32_t)li & 0xFFFFu) | ((uint32_t)ri << 16);
                if (mono) mono[k] = (int16_t)(int)acc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                                            // the next frame overwrites the copies
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Spectrum trace: np.nanmean(wf_data.T[:, :t_avg], axis=1) and y = H-1-int(v/255*H) (utils_supersdr.py:1678-1679)
// over the device copy of wf_data's newest rows.  NumPy sums the rows left to right in float64 starting with row 0
// and divides by the number of non-NaN rows.
__global__ __launch_bounds__(256) void ssdr_trace_kernel(SsdrTraceArgs a)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t plane = (uint64_t)a.n_ch * SSDR_NFFT;
    if (idx >= plane) return;
    double tot = 0.0;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < a.t_avg; k++) {
        const double v = (double)a.ring[(uint64_t)((a.head + k) % a.rows) * plane + idx];
        if (v == v) { tot = tot + v; cnt++; }
    }
    const double mean = tot / (double)cnt;                          // 0/0 = NaN, as np.nanmean of an all-NaN column
    a.trace[idx] = mean;
    if (a.y) {
        const double s = mean / 255.0 * (double)a.spectrum_height;
        a.y[idx] = (s == s) ? (int32_t)a.spectrum_height - 1 - (int32_t)s : INT32_MIN;   // int() of a NaN raises in the reference
    }
}

// S-meter smoothing, one display frame per call (supersdr.py:164-168, 190-191, 936-947).
__global__ __launch_bounds__(256) void ssdr_smeter_kernel(SsdrSmeterArgs a)
{
    const uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= a.n_ch) return;
    ssdr_smeter_chan *g = a.chans + ch;             // the ten-deep history is indexed by a run-time position: it stays in global
    const double rssi = a.rssi_in ? a.rssi_in[ch] : (double)a.rssi[(uint64_t)ch * a.n_frames + (a.n_frames - 1)];
    const uint32_t pos = g->hist_pos;               // memory (a private copy would live in scratch)
    g->hist[pos] = rssi;                                             // deque(maxlen=10).append
    g->hist_pos = (pos + 1) % 10;
    double smooth = g->rssi_smooth;
    if (fabs(rssi) > fabs(smooth)) {
        const double v0 = -20 + 135;
        const double t = log(v0 / (smooth + 135));
        smooth += -v0 / (g->decay_ms / (1000 / (2 * a.fps))) * exp(-t);
    } else {
        smooth += fmin((rssi - smooth) / 5, 3.0);
    }
    g->rssi_smooth = smooth;
    const uint32_t run_index = g->run_index;
    if (run_index % 20 == 0) {
        double m = pos == 0 ? rssi : g->hist[0];
        for (uint32_t i = 1; i < 10; i++) m = fmax(m, i == pos ? rssi : g->hist[i]);
        g->rssi_smooth_slow = m;
    }
    g->run_index = run_index + 1;
}

// SND body in IQ mode: 7 bytes (flags, seq, smeter) + 10 bytes GPS + 512 x (I,Q) big-endian int16.
// One wave per (channel, frame): lane l converts samples 8l .. 8l+7 (32 payload bytes at byte offset 17 + 32 l).  A body is
// 2065 bytes long, so the payload sits at any byte alignment -- the same one for all lanes of a frame: each lane fetches the
// nine aligned dwords that cover its 32 bytes (two 16-byte loads + one), v_alignbyte_b32 shifts neighbours together by the
// frame's misalignment and one v_perm_b32 per sample swaps the bytes of I and of Q.
__global__ __launch_bounds__(256) void ssdr_iqwire_kernel(SsdrWireArgs a)
{
    const int l = threadIdx.x & 63;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (uint64_t)a.n_ch * a.n_frames) return;
    const uint32_t ch = (uint32_t)(item / a.n_frames), f = (uint32_t)(item - (uint64_t)ch * a.n_frames);
    const uint8_t *body = a.bodies + ((uint64_t)ch * a.n_frames + f) * SSDR_WIRE_BODY;
    const uint8_t *p = body + 17 + 32 * l;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);                     // wave-uniform
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p - sh);
    uint32_t d[9];
    {
        typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const u32x4u v0 = *reinterpret_cast<const u32x4u *>(q), v1 = *reinterpret_cast<const u32x4u *>(q + 4);
        d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
        // the ninth dword is only needed (and, at the very end of the buffer, only there) when the payload is not dword-aligned
        d[8] = sh ? q[8] : 0u;
    }
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t be = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);                  // bytes I_hi I_lo Q_hi Q_lo, lowest address first
        w[i] = __builtin_amdgcn_perm(0u, be, 0x02030001u);                                   // -> I | Q << 16, little-endian halves
    }
    u32x4 *dst = reinterpret_cast<u32x4 *>(a.iq + (uint64_t)ch * a.ch_stride + (uint64_t)f * SSDR_FRAME) + 2 * l;
    dst[0] = u32x4{w[0], w[1], w[2], w[3]};
    dst[1] = u32x4{w[4], w[5], w[6], w[7]};
    if (l == 0 && a.rssi) {
        const uint32_t smeter = ((uint32_t)body[5] << 8) | body[6];
        a.rssi[(uint64_t)ch * a.n_frames + f] = 0.1f * (float)smeter - 127.0f;
    }
    if (l == 1 && a.gps) {                   // the GNSS stamp of the frame: '<BBII' at body[7..16] (unaligned: byte by byte)
        auto le32 = [&](int o) { return (uint32_t)body[o] | ((uint32_t)body[o + 1] << 8) | ((uint32_t)body[o + 2] << 16) | ((uint32_t)body[o + 3] << 24); };
        uint32_t *g = a.gps + ((uint64_t)ch * a.n_frames + f) * 4;
        g[0] = body[7];                      // last_gps_solution
        g[1] = body[8];                      // dummy
        g[2] = le32(9);                      // gpssec
        g[3] = le32(13);                     // gpsnsec
    }
}

// ---------------------------------------------------------------------------------------------------------
// IMA ADPCM (kiwi/client.py:33-87): sequential within a stream, parallel across streams: one lane per stream.
// Low nibble first; diff = step>>3 (+step>>2, +step>>1, +step by code bits 0..2), negated by bit 3; sample and
// index clamped; (index, prev) persist across calls for SND audio, are reset per line for W/F (:476-477).
__constant__ int c_ima_step[89] = {
    7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97, 107, 118,
    130, 143, 157, 173, 190, 209, 230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060,
    1166, 1282, 1411, 1552, 1707, 1878, 2066, 2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484,
    7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};

__global__ void ssdr_adpcm_kernel(const uint8_t *data, uint32_t n_streams, uint32_t n_bytes, int32_t *state /*[n][2]*/,
                                  int16_t *out /*[n][2*n_bytes]*/)
{
    const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= n_streams) return;
    int index = state[2 * sidx], prev = state[2 * sidx + 1];
    const uint8_t *p = data + (uint64_t)sidx * n_bytes;
    int16_t *o = out + (uint64_t)sidx * 2 * n_bytes;
    for (uint32_t i = 0; i < n_bytes; i++) {
        const int byte = p[i];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int code = half ? (byte >> 4) : (byte & 0x0F);
            const int step = c_ima_step[index];
            const int adj = (code & 4) ? 2 * (code & 3) + 2 : -1;           // -1,-1,-1,-1,2,4,6,8 (twice)
            index = min(max(index + adj, 0), 88);
            int diff = step >> 3;
            if (code & 1) diff += step >> 2;
            if (code & 2) diff += step >> 1;
            if (code & 4) diff += step;
            if (code & 8) diff = -diff;
            prev = min(max(prev + diff, -32768), 32767);
            o[2 * i + half] = (int16_t)prev;
        }
    }
    state[2 * sidx] = index;
    state[2 * sidx + 1] = prev;
}

// position-weighted sum of 32-bit words mod 2^64 (ssdr_output_checksum): integer arithmetic only, so the value does not
// depend on the grid, the order of the atomics or the GPU
__global__ __launch_bounds__(256) void ssdr_checksum_kernel(const uint32_t *data, uint64_t n_words, unsigned long long *out)
{
    __shared__ unsigned long long part[256];
    unsigned long long h = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride)
        h += (unsigned long long)(data[i] + 0x9E3779B9u) * (2ull * i + 1ull);
    part[threadIdx.x] = h;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, part[0]);
}

// SSDR_FEED_LAZY_OUT: the results of the channels somebody listens to, gathered into compact rows (position in the selection) for the
// copy back to the host; everything else stays on the device.  One workgroup per selected channel; 16 bytes per lane where the rows allow it.
__global__ __launch_bounds__(256) void ssdr_gather_kernel(SsdrGatherArgs a)
{
    const uint32_t pos = blockIdx.x;
    const uint32_t ch = a.sel ? a.sel[pos] : pos;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    for (uint32_t line = 0; line < a.n_lines; line++) {                        // [line][ch][1024] int16 -> [line][pos][1024]
        const u4 *src = reinterpret_cast<const u4 *>(a.wf + ((uint64_t)line * a.n_ch + ch) * SSDR_NFFT);
        u4 *dst = reinterpret_cast<u4 *>(a.wf_out + ((uint64_t)line * a.n_sel + pos) * SSDR_NFFT);
        for (uint32_t i = threadIdx.x; i < SSDR_NFFT * 2 / 16; i += blockDim.x) dst[i] = src[i];
    }
    {                                                                          // [ch][n_frames * 512] int16 -> [pos][...]
        const u4 *src = reinterpret_cast<const u4 *>(a.pcm + (uint64_t)ch * a.n_frames * SSDR_FRAME);
        u4 *dst = reinterpret_cast<u4 *>(a.pcm_out + (uint64_t)pos * a.n_frames * SSDR_FRAME);
        for (uint32_t i = threadIdx.x; i < a.n_frames * (SSDR_FRAME * 2 / 16); i += blockDim.x) dst[i] = src[i];
    }
    for (uint32_t f = threadIdx.x; f < a.n_frames; f += blockDim.x) {
        a.rssi_out[(uint64_t)pos * a.n_frames + f] = a.rssi[(uint64_t)ch * a.n_frames + f];
        a.flags_out[(uint64_t)pos * a.n_frames + f] = a.flags[(uint64_t)ch * a.n_frames + f];
        if (a.wire_rssi) a.wire_rssi_out[(uint64_t)pos * a.n_frames + f] = a.wire_rssi[(uint64_t)ch * a.n_frames + f];
    }
}

} // namespace

hipError_t ssdr_launch_gather(const SsdrGatherArgs &a, hipStream_t stream)
{
    if (!a.n_sel) return hipSuccess;
    hipLaunchKernelGGL(ssdr_gather_kernel, dim3(a.n_sel), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t ssdr_launch_checksum(const void *data, uint64_t n_words, unsigned long long *out, hipStream_t stream)
{
    const uint64_t blocks = (n_words + 255) / 256;
    hipLaunchKernelGGL(ssdr_checksum_kernel, dim3((uint32_t)(blocks < 4096 ? (blocks ? blocks : 1) : 4096)), dim3(256), 0, stream,
                       static_cast<const uint32_t *>(data), n_words, out);
    return hipGetLastError();
}

hipError_t ssdr_launch_adpcm(const uint8_t *data, uint32_t n_streams, uint32_t n_bytes, int32_t *state, int16_t *out,
                             hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_adpcm_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, stream, data, n_streams, n_bytes, state, out);
    return hipGetLastError();
}

hipError_t ssdr_launch_db2col(const SsdrDb2colArgs &a, hipStream_t stream)
{
    const uint64_t items = (uint64_t)a.n_sel * a.n_lines;
    if (!items) return hipSuccess;
    hipLaunchKernelGGL(ssdr_db2col_kernel, dim3((uint32_t)((items + 3) / 4)), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t ssdr_launch_play_rs(const SsdrPlayArgs &a, hipStream_t stream)
{
    const uint64_t items = (uint64_t)a.n_sel * a.n_frames;
    if (!items) return hipSuccess;
    static uint32_t resident = 0;                    // persistent grid: the lanes' taps are loaded once per wave
    if (!resident) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ssdr_play_rs_kernel, 256, 0) == hipSuccess && per_cu > 0)
            resident = (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
        else
            resident = 1024;
    }
    const uint64_t need = (items + 3) / 4;
    hipLaunchKernelGGL(ssdr_play_rs_kernel, dim3((uint32_t)(need < resident ? need : resident)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t ssdr_launch_play(const SsdrPlayArgs &a, hipStream_t stream)
{
    const uint64_t items = (uint64_t)a.n_sel * a.n_frames;
    if (!items) return hipSuccess;
    hipLaunchKernelGGL(ssdr_play_kernel, dim3((uint32_t)((items + 3) / 4)), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t ssdr_launch_trace(const SsdrTraceArgs &a, hipStream_t stream)
{
    const uint64_t n = (uint64_t)a.n_ch * SSDR_NFFT;
    hipLaunchKernelGGL(ssdr_trace_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t ssdr_launch_smeter(const SsdrSmeterArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_smeter_kernel, dim3((a.n_ch + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t ssdr_launch_iqwire(const SsdrWireArgs &a, hipStream_t stream)
{
    const uint64_t items = (uint64_t)a.n_ch * a.n_frames;
    if (!items) return hipSuccess;
    hipLaunchKernelGGL(ssdr_iqwire_kernel, dim3((uint32_t)((items + 3) / 4)), dim3(256), 0, stream, a);
    return hipGetLastError();
}
