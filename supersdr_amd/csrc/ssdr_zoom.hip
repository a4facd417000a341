// ssdr_zoom.hip -- the zoom stage in front of the waterfall kernel (ssdr_set_wf_zoom): what the KiwiSDR server's DDC does for
// "SET zoom=%d start=%d" (utils_supersdr.py:741, 753-758, 839).  Per channel:
//     z[n] = x[n] * conj(P(phi0 + n dphi))            P(x) = e^{j 2 pi x / 2^32} at all 32 bits (ssdr_phasor32), per sample
//     y[m] = sum_k h[k] z[Z m - k]                     k ascending, fma chain from zero; h = the reference's tap formula
//     out[m] = saturate(rint(y[m])) as int16 I, Q      the zoomed stream, 1/Z of the input rate
// One 256-thread workgroup per channel walks the call in chunks of 512 outputs: the chunk's 512 Z inputs and the 256 before
// them are mixed into LDS once (each sample's phasor from its absolute phase: no recurrence, so any thread can mix any
// sample and the CPU twin is a plain loop), then every thread forms two outputs.  Not a hot path: a display feature for
// the receivers somebody is looking at; ~(64 + 30) multiply-adds per input sample.
#include "ssdr_math.h"
#include "ssdr_kernels.h"

namespace {

constexpr int ZCHUNK = 512;                  // outputs per chunk

__global__ __launch_bounds__(256) void ssdr_zoom_kernel(SsdrZoomArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char zsm[];
    float2 *s_z = reinterpret_cast<float2 *>(zsm);                                   // [SSDR_ZOOM_HIST + ZCHUNK * Z]
    float *s_taps = reinterpret_cast<float *>(zsm + (size_t)(SSDR_ZOOM_HIST + ZCHUNK * a.zoom) * sizeof(float2));
    const uint32_t t = threadIdx.x, ch = blockIdx.x, Z = a.zoom;
    if (ch >= a.n_ch) return;
    for (uint32_t i = t; i < SSDR_ZOOM_TAPS_MAX + 1; i += 256) s_taps[i] = i < a.ntap ? a.taps[i] : 0.0f;
    const uint32_t dphi = a.dphi[ch], phi0 = a.phase[ch];
    const uint32_t *src = a.iq + (uint64_t)ch * a.ch_stride;
    uint32_t *hist = a.hist + (uint64_t)ch * SSDR_ZOOM_HIST;
    const uint32_t n_out = a.n_in / Z;
    uint32_t *dst = a.out + (uint64_t)ch * n_out;
    for (uint32_t m0 = 0; m0 < n_out; m0 += ZCHUNK) {
        const uint32_t n_here = min((uint32_t)ZCHUNK, n_out - m0);
        const int64_t in0 = (int64_t)m0 * Z - SSDR_ZOOM_HIST;                        // input index of LDS slot 0 (may be negative)
        const uint32_t n_stage = SSDR_ZOOM_HIST + n_here * Z;
        __syncthreads();
        for (uint32_t i = t; i < n_stage; i += 256) {
            const int64_t n = in0 + i;                                               // sample index relative to the call's first
            const uint32_t raw = n < 0 ? hist[SSDR_ZOOM_HIST + n] : src[n];
            float c, s;
            ssdr_phasor32(phi0 + (uint32_t)(int32_t)n * dphi, c, s);
            const float xr = (float)(int16_t)(raw & 0xFFFFu), xi = (float)((int32_t)raw >> 16);
            s_z[i] = make_float2(fmaf(xr, c, xi * s), fmaf(xi, c, -(xr * s)));       // x * (c - j s)
        }
        __syncthreads();
        for (uint32_t q = t; q < n_here; q += 256) {
            // newest input of output m: index Z m -> LDS slot SSDR_ZOOM_HIST + Z q + (Z - 1)?  y[m] = sum h[k] z[Z m - k]: slot HIST + Z q - k
            const float2 *w = s_z + SSDR_ZOOM_HIST + Z * q;
            float ar = 0.0f, ai = 0.0f;
            for (uint32_t k = 0; k < a.ntap; k++) {
                const float h = s_taps[k];
                const float2 v = *(w - k);
                ar = fmaf(h, v.x, ar);
                ai = fmaf(h, v.y, ai);
            }
            const int ir = __float2int_rn(ar), ii = __float2int_rn(ai);              // saturating conversions, then saturating pack
            dst[m0 + q] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(ir, ii));
        }
    }
    __syncthreads();
    // carry: the call's last SSDR_ZOOM_HIST raw samples (n_in >= 512 always), the phase of the next call's first sample
    {
        const uint32_t v = src[a.n_in - SSDR_ZOOM_HIST + t];
        __syncthreads();
        hist[t] = v;
    }
    if (t == 0) a.phase[ch] = phi0 + a.n_in * dphi;
}

} // namespace

hipError_t ssdr_launch_zoom(const SsdrZoomArgs &a, hipStream_t stream)
{
    const size_t lds = (size_t)(SSDR_ZOOM_HIST + ZCHUNK * a.zoom) * sizeof(float2) + (SSDR_ZOOM_TAPS_MAX + 1) * sizeof(float);
    hipLaunchKernelGGL(ssdr_zoom_kernel, dim3(a.n_ch), dim3(256), lds, stream, a);
    return hipGetLastError();
}
