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

// One wave per channel, lines in order; lane l owns bins 16l .. 16l+15.
//
// NumPy on a float32 line (restated, utils_supersdr.py:787-813):
//   wf_db   = (-(255 - s) - 13) + 3 zoom               three float32 roundings
//   wf_db[0] = wf_db[1]
//   low     = a + (b - a) * G,  a, b = sorted[409], sorted[410], G = float32(0.2000122)   (np.percentile(.,40)
//             on float32: q = 40/float32(100), virtual index 1024 q + (1 - q) - 1 in float32 -> gamma G)
//   high    = max;  dyn = max(high - low, 40)
//   color   = clip(clip((wf_db - (low + dlo)) / ((dyn + dhi) - dlo), 0, 1) * 254, 0, 255)
// The order statistics are taken on the int16 sums (the map sum -> wf_db is monotone), by bisection on the
// value with wave-wide counting: exact, no sort.
__global__ __launch_bounds__(64) void ssdr_db2col_kernel(SsdrDb2colArgs a)
{
    const int l = threadIdx.x;
    const uint32_t ch = blockIdx.x;
    if (ch >= a.n_ch) return;
    ssdr_db2col_chan st = a.chans[ch];
    const float z3 = (float)(3 * st.zoom), dlo = (float)st.delta_low_db, dhi = (float)st.delta_high_db;
    const float fn = (float)a.n_avg;
    const float G = 0x1.99ap-3f;

    for (uint32_t line = 0; line < a.n_lines; line++) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(a.wf + ((uint64_t)line * a.n_ch + ch) * SSDR_NFFT) + 2 * l;
        const u32x4 r0 = src[0], r1 = src[1];
        const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        int s[16];
#pragma unroll
        for (int i = 0; i < 8; i++) { s[2 * i] = (int)(rw[i] & 0xFFFFu); s[2 * i + 1] = (int)(rw[i] >> 16); }
        if (l == 0) s[0] = s[1];                                    // "first bin is broken" (:791)

        float wf_db[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float sp = (float)s[i] / fn;                       // float32(sum) / float32(N) == np.mean
            wf_db[i] = (-(255.0f - sp) - 13.0f) + z3;
        }

        if (st.auto_scale) {
            // smallest v with #{s <= v} >= 410  ==  sorted[409]
            int lo = 0, hi = 32767;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                int c = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) c += (s[i] <= mid) ? 1 : 0;
                if (wave_sum_i(c) >= 410) hi = mid; else lo = mid + 1;
            }
            const int v409 = lo;
            int c = 0, above = 0x7FFF, mx = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                c += (s[i] <= v409) ? 1 : 0;
                above = min(above, (s[i] > v409) ? s[i] : 0x7FFF);
                mx = max(mx, s[i]);
            }
            const int v410 = (wave_sum_i(c) >= 411) ? v409 : wave_min_i(above);
            const int vmax = wave_max_i(mx);
            const float fa = (-(255.0f - (float)v409 / fn) - 13.0f) + z3;
            const float fb = (-(255.0f - (float)v410 / fn) - 13.0f) + z3;
            const float fh = (-(255.0f - (float)vmax / fn) - 13.0f) + z3;
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

        f32x4 *dst = reinterpret_cast<f32x4 *>(a.color + ((uint64_t)line * a.n_ch + ch) * SSDR_NFFT) + 4 * l;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float c = (wf_db[4 * q + i] - lo2) / den;
                c = fminf(fmaxf(c, 0.0f), 1.0f) * 254.0f;
                o[i] = fminf(fmaxf(c, 0.0f), 255.0f);
            }
            dst[q] = o;
        }
    }
    if (l == 0) a.chans[ch] = st;
}

// ---------------------------------------------------------------------------------------------------------
// play_buffer: out[4n + r] for r = 0..3.  With zero stuffing only every 4th term of the 33-tap convolution
// is non-zero:  y[i] = 4 * sum_j h[j] U[i + 32 - j],  U = [history(32) | stuffed frame],  U[m] != 0 only for
// m = 0 mod 4.  One wave per channel; lane l produces outputs 32l .. 32l+31 of each 2048-sample block.
__global__ __launch_bounds__(64) void ssdr_play_kernel(SsdrPlayArgs a)
{
    __shared__ double s_x[8 + SSDR_FRAME];                          // 8 carried samples + the frame, volume applied
    const int l = threadIdx.x;
    const uint32_t ch = blockIdx.x;
    if (ch >= a.n_ch) return;
    const ssdr_play_chan pc = a.chans[ch];
    const double vol = pc.volume / 100.0;
    const double lv = fmin(1.0 - pc.balance, 1.0), rv = fmin(1.0 + pc.balance, 1.0);
    const double l2 = lv * lv, r2 = rv * rv;
    double h[33];
#pragma unroll
    for (int j = 0; j < 33; j++) h[j] = a.taps[j];
    if (l < 8) s_x[l] = a.hist[(size_t)ch * 8 + l];
    __syncthreads();

    for (uint32_t f = 0; f < a.n_frames; f++) {
        const int16_t *src = a.pcm + ((uint64_t)ch * a.n_frames + f) * SSDR_FRAME;
#pragma unroll
        for (int i = 0; i < 8; i++) s_x[8 + 8 * l + i] = (double)src[8 * l + i] * vol;
        __syncthreads();
        // stuffed index m = 4 * (k) holds X[k], X = s_x (k = 0..519; k < 8 is history)
        uint32_t *dst = reinterpret_cast<uint32_t *>(a.out + (((uint64_t)ch * a.n_frames + f) * 2048 + 32 * l) * 2);
        for (int o4 = 0; o4 < 32; o4 += 4) {
#pragma unroll
            for (int r = 0; r < 4; r++) {                           // i = 32 l + o4 + r, so i mod 4 == r (static)
                const int i = 32 * l + o4 + r;
                // terms j with (i + 32 - j) = 0 mod 4  ->  j = r + 4 t (j <= 32), summed in ascending sample order
                double acc = 0.0;
#pragma unroll
                for (int t = 8; t >= 0; t--) {
                    const int j = r + 4 * t;
                    if (j <= 32) acc += h[j] * s_x[(i + 32 - j) >> 2];
                }
                acc *= 4.0;
                const int li = (int)(acc * l2), ri = (int)(acc * r2);    // trunc toward zero, then wrap to int16
                dst[o4 + r] = ((uint32_t)li & 0xFFFFu) | ((uint32_t)ri << 16);
                if (a.mono) a.mono[((uint64_t)ch * a.n_frames + f) * 2048 + i] = (int16_t)(int)acc;    // recording branch (:1139-1140)
            }
        }
        __syncthreads();
        if (l < 8) s_x[l] = s_x[SSDR_FRAME + l];
        __syncthreads();
    }
    if (l < 8) a.hist[(size_t)ch * 8 + l] = s_x[l];
}

// ---------------------------------------------------------------------------------------------------------
// play_buffer for a fractional SAMPLE_RATIO (20.25 kHz KiwiSDRs, utils_supersdr.py:1125-1126): every 512-sample
// frame goes through scipy.signal.resample_poly(popped, 64, 27, padtype="line") on its own (no history) and the
// last output is dropped: 1213 stereo samples per frame.  Restated from scipy 1.15.3 (_upfirdn_apply.pyx,
// _apply_impl with MODE_LINE): output y of the full polyphase convolution uses phase t = 27 y mod 64 ending at
// sample 27 y div 64; its 21 taps are applied oldest sample first, one multiply and one add each in float64;
// samples outside the frame lie on the line through the frame's first and last sample; outputs
// y = 24 .. 24 + 1212 are kept.  One workgroup per (channel, frame).
__global__ __launch_bounds__(256) void ssdr_play_rs_kernel(SsdrPlayArgs a)
{
    __shared__ double s_x[SSDR_FRAME];
    const uint32_t ch = blockIdx.x / a.n_frames, f = blockIdx.x - ch * a.n_frames;
    const ssdr_play_chan pc = a.chans[ch];
    const double vol = pc.volume / 100.0;
    const double lv = fmin(1.0 - pc.balance, 1.0), rv = fmin(1.0 + pc.balance, 1.0);
    const double l2 = lv * lv, r2 = rv * rv;
    const int16_t *src = a.pcm + ((uint64_t)ch * a.n_frames + f) * SSDR_FRAME;
    for (int i = threadIdx.x; i < SSDR_FRAME; i += blockDim.x) s_x[i] = (double)src[i] * vol;
    __syncthreads();
    const double x0 = s_x[0], xl = s_x[SSDR_FRAME - 1];
    const double slope = (xl - x0) / (double)(SSDR_FRAME - 1);
    uint32_t *dst = reinterpret_cast<uint32_t *>(a.out + ((uint64_t)ch * a.n_frames + f) * SSDR_RS_OUT_PER_FRAME * 2);
    for (int k = threadIdx.x; k < SSDR_RS_OUT_PER_FRAME; k += blockDim.x) {
        const int total = (k + SSDR_RS_PRE_REMOVE) * SSDR_RS_DOWN;
        const int x_idx = total / SSDR_RS_UP, t = total % SSDR_RS_UP;
        const double *h = a.rs_taps + t * SSDR_RS_HPP;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < SSDR_RS_HPP; m++) {
            const int xi = x_idx - SSDR_RS_HPP + 1 + m;
            double xv;
            if (xi < 0) xv = x0 + (double)xi * slope;
            else if (xi >= SSDR_FRAME) xv = xl + (double)(xi - SSDR_FRAME + 1) * slope;
            else xv = s_x[xi];
            acc = acc + xv * h[m];
        }
        const int li = (int)(acc * l2), ri = (int)(acc * r2);            // trunc toward zero, then wrap to int16
        dst[k] = ((uint32_t)li & 0xFFFFu) | ((uint32_t)ri << 16);
        if (a.mono) a.mono[((uint64_t)ch * a.n_frames + f) * SSDR_RS_OUT_PER_FRAME + k] = (int16_t)(int)acc;
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
// One wave per (channel, frame): lane l converts samples 8l .. 8l+7 (32 payload bytes at byte offset 17 + 32 l).
__global__ __launch_bounds__(64) void ssdr_iqwire_kernel(SsdrWireArgs a)
{
    const int l = threadIdx.x;
    const uint32_t f = blockIdx.x % a.n_frames, ch = blockIdx.x / a.n_frames;
    if (ch >= a.n_ch) return;
    const uint8_t *body = a.bodies + ((uint64_t)ch * a.n_frames + f) * SSDR_WIRE_BODY;
    const uint8_t *p = body + 17 + 32 * l;
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        // I = p[0..1] big-endian, Q = p[2..3] big-endian -> dword I | Q << 16 (little-endian halves)
        const uint32_t b0 = p[4 * i], b1 = p[4 * i + 1], b2 = p[4 * i + 2], b3 = p[4 * i + 3];
        w[i] = ((b0 << 8) | b1) | (((b2 << 8) | b3) << 16);
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

} // namespace

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
    hipLaunchKernelGGL(ssdr_db2col_kernel, dim3(a.n_ch), dim3(64), 0, stream, a);
    return hipGetLastError();
}
hipError_t ssdr_launch_play_rs(const SsdrPlayArgs &a, hipStream_t stream)
{
    if (a.n_ch == 0 || a.n_frames == 0) return hipSuccess;
    hipLaunchKernelGGL(ssdr_play_rs_kernel, dim3(a.n_ch * a.n_frames), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t ssdr_launch_play(const SsdrPlayArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_play_kernel, dim3(a.n_ch), dim3(64), 0, stream, a);
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
    hipLaunchKernelGGL(ssdr_iqwire_kernel, dim3(a.n_ch * a.n_frames), dim3(64), 0, stream, a);
    return hipGetLastError();
}
