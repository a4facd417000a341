// ssdr_wf.hip -- waterfall stage K1 for gfx950 (MI355X):
//   int16 IQ -> Hann window -> 1024-pt complex FFT -> |X|^2 -> 1-dB byte quantise ->
//   fftshift -> sum of N consecutive lines (int16)
//
// Stands in for the KiwiSDR server's W/F producer whose output the reference consumes
// in kiwi_waterfall.receive_spectrum (utils_supersdr.py:780-785), fused with the
// reference's time binning (utils_supersdr.py:881-888; integer sum == np.mean * N).
//
// Mapping (wave64, one FFT per 32-lane half, 32 points per lane):
//   * lane l of a half loads samples n = 32*r + l, r = 0..31: every load instruction
//     covers one 128-byte line per FFT (coalesced), no LDS staging on the way in.
//   * DIT stages 1..5 run entirely in registers on the bit-reversed group this lane
//     owns (compile-time W_32 twiddles), ONE transpose through LDS (stride-33 padded,
//     conflict-free both ways, re then im through the same 4.1 KB), stages 6..10 in
//     registers again with per-lane twiddles from a per-stage LDS table.
//   * power, quantiser (bit-pattern estimate + one LDS threshold compare), N-line
//     accumulation in registers, then the line is staged through the (now free)
//     exchange buffer so every lane stores 16 contiguous bytes (512 B per half-wave
//     instruction) with the fftshift folded into the LDS address.
//   The butterflies are exactly those of a textbook radix-2 DIT FFT (same operand
//   pairs, same fma pattern), only regrouped -- results are bit-identical to it.
#include "ssdr_math.h"
#include "ssdr_kernels.h"

namespace {

constexpr int XPAD = 33;                       // row stride (floats) of the transpose buffer
constexpr int XCH_FLOATS = 32 * XPAD;          // per FFT: 4224 B
constexpr int WAVES = SSDR_WF_BLOCK / 64;

__device__ constexpr int brev5(int v)
{
    return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

SSDR_DEV void bfly(float &ur, float &ui, float &vr, float &vi, float wr, float wi)
{
    float tr = fmaf(wr, vr, -(wi * vi));
    float ti = fmaf(wr, vi, wi * vr);
    float xr = ur, xi = ui;
    ur = xr + tr; ui = xi + ti;
    vr = xr - tr; vi = xi - ti;
}
SSDR_DEV void bfly_1(float &ur, float &ui, float &vr, float &vi)      // w = 1
{
    float tr = vr, ti = vi, xr = ur, xi = ui;
    ur = xr + tr; ui = xi + ti;
    vr = xr - tr; vi = xi - ti;
}
SSDR_DEV void bfly_mj(float &ur, float &ui, float &vr, float &vi)     // w = -j
{
    float tr = vi, ti = -vr, xr = ur, xi = ui;
    ur = xr + tr; ui = xi + ti;
    vr = xr - tr; vi = xi - ti;
}

// stages 1..5 on a[0..31] (a-index order), twiddle W_1024[k * (1024 >> s)] = W32[k * (32 >> s)]
template <int S>
SSDR_DEV void stage_const(float (&re)[32], float (&im)[32])
{
    constexpr float W32R[16] = SSDR_W32R_INIT;
    constexpr float W32I[16] = SSDR_W32I_INIT;
    constexpr int half = 1 << (S - 1);
#pragma unroll
    for (int blk = 0; blk < 32; blk += 2 * half) {
#pragma unroll
        for (int k = 0; k < half; k++) {
            const int m = k * (32 >> S);              // index into W32 (0..15)
            const int i = blk + k, j = i + half;
            if (m == 0) bfly_1(re[i], im[i], re[j], im[j]);
            else if (m == 8) bfly_mj(re[i], im[i], re[j], im[j]);
            else bfly(re[i], im[i], re[j], im[j], W32R[m], W32I[m]);
        }
    }
}

// stages 6..10 (T = s - 6) on x[j] = a[32 j + lane]; twiddle W_1024[(lane + 32 (j mod 2^T)) << (4 - T)]
template <int T>
SSDR_DEV void stage_lane(float (&re)[32], float (&im)[32], const float2 *tw_lane)
{
    constexpr int half = 1 << T;
    constexpr int off = 32 * (half - 1);
#pragma unroll
    for (int jl = 0; jl < half; jl++) {
        const float2 w = tw_lane[off + jl * 32];
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * half) {
            const int i = blk + jl, j = i + half;
            bfly(re[i], im[i], re[j], im[j], w.x, w.y);
        }
    }
}

SSDR_DEV void wave_lds_sync()
{
    // One wave owns its LDS region and DS instructions of a wave execute in order, so
    // no s_barrier is needed -- only a compiler fence so that cross-lane LDS traffic is
    // not reordered (per-thread alias analysis would otherwise be allowed to).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one windowed FFT + power + quantise for the 32-lane half this lane belongs to
SSDR_DEV void fft_line_bytes(const uint32_t *__restrict__ src /* + lane */, const float *s_win_lane,
                             const float2 *s_tw_lane, const float *s_thr, float *xch, int l, float cal,
                             int (&acc)[32])
{
    float re[32], im[32];
    uint32_t raw[32];
#pragma unroll
    for (int r = 0; r < 32; r++) raw[r] = src[32 * r];
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const float w = s_win_lane[32 * r];
        const float xr = (float)(int16_t)(raw[r] & 0xFFFFu);
        const float xi = (float)((int32_t)raw[r] >> 16);
        re[brev5(r)] = xr * w;
        im[brev5(r)] = xi * w;
    }
    stage_const<1>(re, im);
    stage_const<2>(re, im);
    stage_const<3>(re, im);
    stage_const<4>(re, im);
    stage_const<5>(re, im);

    // transpose: element (g = brev5(l), r) -> lane r, register g
    const int g = __builtin_bitreverse32((uint32_t)l) >> 27;
#pragma unroll
    for (int r = 0; r < 32; r++) xch[r * XPAD + g] = re[r];
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; j++) re[j] = xch[l * XPAD + j];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 32; r++) xch[r * XPAD + g] = im[r];
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; j++) im[j] = xch[l * XPAD + j];
    wave_lds_sync();

    stage_lane<0>(re, im, s_tw_lane);
    stage_lane<1>(re, im, s_tw_lane);
    stage_lane<2>(re, im, s_tw_lane);
    stage_lane<3>(re, im, s_tw_lane);
    stage_lane<4>(re, im, s_tw_lane);

#pragma unroll
    for (int j = 0; j < 32; j++) {
        const float p = fmaf(re[j], re[j], im[j] * im[j]) * cal;
        acc[j] += ssdr_quantise(p, s_thr);
    }
}

__global__ __launch_bounds__(SSDR_WF_BLOCK) void ssdr_wf_kernel(SsdrWfArgs a)
{
    __shared__ float s_win[SSDR_NFFT];
    __shared__ float2 s_tw[SSDR_TW_STAGE_N];
    __shared__ float s_thr[256];
    __shared__ __attribute__((aligned(16))) float s_xch[WAVES][2][XCH_FLOATS];

    for (int i = threadIdx.x; i < SSDR_NFFT; i += SSDR_WF_BLOCK) s_win[i] = a.win[i];
    for (int i = threadIdx.x; i < SSDR_TW_STAGE_N; i += SSDR_WF_BLOCK) s_tw[i] = a.tw_stage[i];
    for (int i = threadIdx.x; i < 256; i += SSDR_WF_BLOCK) s_thr[i] = a.thr[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l = lane & 31;
    float *xch = &s_xch[wave][h][0];
    const uint32_t n_pairs = (a.n_ch + 1) >> 1;
    const uint64_t n_items = (uint64_t)n_pairs * a.n_groups;
    const uint64_t wave_stride = (uint64_t)gridDim.x * WAVES;

    for (uint64_t item = (uint64_t)blockIdx.x * WAVES + wave; item < n_items; item += wave_stride) {
        const uint32_t grp = (uint32_t)(item / n_pairs);
        const uint32_t pair = (uint32_t)(item - (uint64_t)grp * n_pairs);
        const uint32_t ch_raw = 2 * pair + h;
        const bool ch_ok = ch_raw < a.n_ch;
        const uint32_t ch = ch_ok ? ch_raw : a.n_ch - 1;
        // lines [l0, l1) of this batch belong to averaging group `grp`
        const int64_t g0 = (int64_t)grp * a.n_avg - a.phase;
        const uint32_t l0 = g0 < 0 ? 0u : (uint32_t)g0;
        const uint32_t l1 = min((uint32_t)(g0 + a.n_avg), a.n_lines);
        const bool carry_in = (grp == 0) && (a.phase != 0);
        const bool complete = (g0 + (int64_t)a.n_avg) <= (int64_t)a.n_lines;
        const float cal = a.consts[ch].wf_cal_lin;

        int acc[32];
#pragma unroll
        for (int j = 0; j < 32; j++) acc[j] = 0;
        const uint32_t *src = a.iq + (uint64_t)ch * a.ch_stride + (uint64_t)l0 * SSDR_NFFT + l;
        for (uint32_t line = l0; line < l1; line++, src += SSDR_NFFT)
            fft_line_bytes(src, s_win + l, s_tw + l, s_thr, xch, l, cal, acc);

        // stage the int16 line through LDS with the fftshift folded into the address
        int16_t *x16 = reinterpret_cast<int16_t *>(xch);
#pragma unroll
        for (int j = 0; j < 32; j++) x16[(32 * ((j + 16) & 31) + l)] = (int16_t)acc[j];
        wave_lds_sync();
        int16_t *dst = complete ? a.out + ((uint64_t)grp * a.n_ch + ch) * SSDR_NFFT
                                : a.acc_out + (uint64_t)ch * SSDR_NFFT;
        const int16_t *cin = a.acc_in + (uint64_t)ch * SSDR_NFFT;
        const uint4 *x128 = reinterpret_cast<const uint4 *>(xch);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 v = x128[q * 32 + l];
            if (carry_in) {                     // wave-uniform; sums stay < 2^15 so a 32-bit add is a packed 2x16 add
                const uint4 c = reinterpret_cast<const uint4 *>(cin)[q * 32 + l];
                v.x += c.x;
                v.y += c.y;
                v.z += c.z;
                v.w += c.w;
            }
            if (ch_ok) reinterpret_cast<uint4 *>(dst)[q * 32 + l] = v;
        }
        wave_lds_sync();
    }
}

// exhaustive quantiser self-test: every positive finite float against a binary search
__global__ void ssdr_quant_selftest_kernel(const float *thr_g, unsigned long long *mismatch)
{
    __shared__ float s_thr[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_thr[i] = thr_g[i];
    __syncthreads();
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < 0x7F800000ull; u += stride) {
        const float p = __uint_as_float((uint32_t)u);
        int lo = 0, hi = 255;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_thr[mid] <= p) lo = mid; else hi = mid - 1;
        }
        bad += (ssdr_quantise(p, s_thr) != lo);
    }
    if (bad) atomicAdd(mismatch, bad);
}

} // namespace

hipError_t ssdr_launch_wf(const SsdrWfArgs &a, uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_wf_kernel, dim3(grid), dim3(SSDR_WF_BLOCK), 0, stream, a);
    return hipGetLastError();
}

hipError_t ssdr_launch_quant_selftest(const float *thr, unsigned long long *mismatch, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_quant_selftest_kernel, dim3(2048), dim3(256), 0, stream, thr, mismatch);
    return hipGetLastError();
}
