// ssdr_wf_exact.hip -- the waterfall stage in float64 (ssdr_set_exact_bins): the same definition as ssdr_wf.hip
// (Hann window -> 1024-pt FFT -> |X|^2 cal -> byte = #{k : T[k] <= p} -> fftshift -> sum of N lines), evaluated the way the
// normative float64 definition (NumPy) evaluates it: samples times the float32 window table in float64 (exact products), a float64 FFT,
// float64 power, float32 thresholds compared in float64.  An fp32 FFT lands ~3e-4 of the bins one step off where |X| sits
// within its rounding error of a 1-dB threshold (the guard band of DESIGN.md section 3); a float64 FFT's error (1e-15
// relative) is ten orders of magnitude below the spacing of anything that can sit there, so these bins equal the
// float64 definition's bit for bit -- north_star's "bit-exact int16 waterfall bins" taken literally.
//
// Not the fast path: one 256-thread workgroup per (channel, averaging group), the line in LDS as 1024 double complex,
// textbook radix-2 DIT with __syncthreads between stages.  ~25x slower than ssdr_wf_kernel (profiles/README.md); opt-in.
#include "ssdr_kernels.h"

namespace {

__device__ __forceinline__ uint32_t brev10(uint32_t v) { return __builtin_bitreverse32(v) >> 22; }

__global__ __launch_bounds__(256) void ssdr_wf_exact_kernel(SsdrWfArgs a, const double2 *tw /*[512] e^{-2 pi j m/1024}*/,
                                                           const float *thr /*[256]*/)
{
    __shared__ double2 z[SSDR_NFFT];
    __shared__ double s_thr[256];
    const uint32_t t = threadIdx.x;
    const uint32_t ch = blockIdx.x % a.n_ch, grp = blockIdx.x / a.n_ch;
    s_thr[t] = (double)thr[t];
    const int64_t g0 = (int64_t)grp * a.n_avg - a.phase;
    const uint32_t l0 = g0 < 0 ? 0u : (uint32_t)g0;
    const uint32_t l1 = min((uint32_t)(g0 + a.n_avg), a.n_lines);
    const bool carry_in = (grp == 0) && (a.phase != 0);
    const bool complete = (g0 + (int64_t)a.n_avg) <= (int64_t)a.n_lines;
    const double cal = (double)a.consts[ch].wf_cal_lin;
    const uint32_t step = a.tail ? SSDR_NFFT / 2 : SSDR_NFFT;
    const uint32_t *base = a.iq + (uint64_t)ch * a.ch_stride;
    int acc[4] = {0, 0, 0, 0};
    for (uint32_t line = l0; line < l1; line++) {
        __syncthreads();
        for (uint32_t i = t; i < SSDR_NFFT; i += 256) {
            uint32_t raw;
            if (a.tail) {                            // hop 512: half-line (line - 1) then half-line (line); half-line -1 is the carried tail
                const uint32_t half = i >> 9, o = i & 511u;
                raw = (half == 0) ? (line ? base[(uint64_t)(line - 1) * 512 + o] : a.tail[(uint64_t)ch * 512 + o])
                                  : base[(uint64_t)line * 512 + o];
            } else {
                raw = base[(uint64_t)line * step + i];
            }
            const double w = (double)a.win[i];
            z[brev10(i)] = make_double2((double)(int16_t)(raw & 0xFFFFu) * w, (double)((int32_t)raw >> 16) * w);
        }
        for (uint32_t s = 1; s <= 10; s++) {
            __syncthreads();
            const uint32_t half = 1u << (s - 1);
            for (uint32_t b = t; b < SSDR_NFFT / 2; b += 256) {
                const uint32_t k = b & (half - 1), i = ((b >> (s - 1)) << s) + k, j = i + half;
                const double2 w = tw[k << (10 - s)], u = z[i], v = z[j];
                const double tr = w.x * v.x - w.y * v.y, ti = w.x * v.y + w.y * v.x;
                z[i] = make_double2(u.x + tr, u.y + ti);
                z[j] = make_double2(u.x - tr, u.y - ti);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t j = 4 * t + q;            // output bin (ascending frequency) <- FFT bin (j + 512) mod 1024
            const double2 x = z[(j + 512) & 1023];
            const double p = (x.x * x.x + x.y * x.y) * cal;
            int lo = 0, hi = 255;                    // byte = #{k in 1..255 : T[k] <= p}
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_thr[mid] <= p) lo = mid; else hi = mid - 1;
            }
            acc[q] += lo;
        }
    }
    int16_t *dst = complete ? a.out + ((uint64_t)grp * a.n_ch + ch) * SSDR_NFFT : a.acc_out + (uint64_t)ch * SSDR_NFFT;
    const int16_t *cin = a.acc_in + (uint64_t)ch * SSDR_NFFT;
#pragma unroll
    for (int q = 0; q < 4; q++) dst[4 * t + q] = (int16_t)(acc[q] + (carry_in ? (int)cin[4 * t + q] : 0));
}

} // namespace

hipError_t ssdr_launch_wf_exact(const SsdrWfArgs &a, const double2 *tw, const float *thr, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_wf_exact_kernel, dim3(a.n_ch * a.n_groups), dim3(256), 0, stream, a, tw, thr);
    return hipGetLastError();
}
