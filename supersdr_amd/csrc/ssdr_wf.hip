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
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The register budget only holds if the phases of a line stay phases: without these fences
// the machine scheduler hoists later phases' LDS table reads across the whole FFT and spills.
// (Scheduling fence only; emits no instruction.)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ constexpr int brev5(int v)
{
    return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Radix-2 DIT butterfly in its 6-FMA form (Linzer-Feig / Goedecker):
//     s = u - j*wi*(j v)...   precisely, with w = (wr, wi), v = (vr, vi), u = (ur, ui):
//     sr = fma(-wi, vi, ur)     si = fma(wi, vr, ui)          s = u + j*wi*v  (imaginary part of w)
//     ar = fma( wr, vr, sr)     ai = fma(wr, vi, si)          a = u + w*v
//     br = fma( 2, ur, -ar)     bi = fma(2, ui, -ai)          b = 2u - a = u - w*v
// 6 full-rate fp32 ops instead of 8 (mul, mul, fma, fma, 4 adds); on gfx950 only fma/add/mul/mov
// issue at 2 cycles per wave, so op count IS the cost (profiles/r01_valu_issue_rate_ubench.txt).
// The twin (oracle/ssdr_twin.c) states the same six roundings.
SSDR_DEV void bfly(f32x2 &u, f32x2 &v, float wr, float wi)
{
    const float sr = fmaf(-wi, v.y, u.x), si = fmaf(wi, v.x, u.y);
    const float ar = fmaf(wr, v.x, sr), ai = fmaf(wr, v.y, si);
    const float br = fmaf(2.0f, u.x, -ar), bi = fmaf(2.0f, u.y, -ai);
    u = f32x2{ar, ai};
    v = f32x2{br, bi};
}
SSDR_DEV void bfly_1(f32x2 &u, f32x2 &v)                          // w = 1 (stages 1..5 only)
{
    const f32x2 t = v, x = u;
    u = x + t; v = x - t;
}
SSDR_DEV void bfly_mj(f32x2 &u, f32x2 &v)                         // w = -j: t = (vi, -vr) (stages 1..5 only)
{
    const f32x2 t = {v.y, -v.x}, x = u;
    u = x + t; v = x - t;
}

// stages 1..5 on a[0..31] (a-index order), twiddle W_1024[k * (1024 >> s)] = W32[k * (32 >> s)]
template <int S>
SSDR_DEV void stage_const(f32x2 (&z)[32])
{
    constexpr float W32R[16] = SSDR_W32R_INIT;
    constexpr float W32I[16] = SSDR_W32I_INIT;
    constexpr int half = 1 << (S - 1);
#pragma unroll
    for (int k = 0; k < half; k++) {
        const int mi = k * (32 >> S);                 // index into W32 (0..15)
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * half) {
            const int i = blk + k, j = i + half;
            if (mi == 0) bfly_1(z[i], z[j]);
            else if (mi == 8) bfly_mj(z[i], z[j]);
            else bfly(z[i], z[j], W32R[mi], W32I[mi]);
        }
    }
}

// stages 6..10 (T = s - 6) on x[j] = a[32 j + lane]; twiddle W_1024[(lane + 32 (j mod 2^T)) << (4 - T)].
// Every butterfly takes the general form here, also where a lane's twiddle happens to be 1 or -j.
template <int T>
SSDR_DEV void stage_lane(f32x2 (&z)[32], const f32x2 *tw_lane)
{
    constexpr int half = 1 << T;
    constexpr int off = 32 * (half - 1);
#pragma unroll
    for (int jl = 0; jl < half; jl++) {
        const f32x2 w = tw_lane[off + jl * 32];
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * half) {
            const int i = blk + jl, j = i + half;
            bfly(z[i], z[j], w.x, w.y);
        }
        if ((jl & 3) == 3) SCHED_FENCE();
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

// raw int16 IQ dwords of one line -> windowed complex samples in a-index (bit-reversed) order
SSDR_DEV void load_line(const uint32_t *__restrict__ src /* + lane */, uint32_t (&raw)[32])
{
#pragma unroll
    for (int r = 0; r < 32; r++) raw[r] = __builtin_nontemporal_load(src + 32 * r);
}

SSDR_DEV void window_line(const uint32_t (&raw)[32], const float *s_win_lane, f32x2 (&z)[32])
{
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const float w = s_win_lane[32 * r];
        const f32x2 x = {(float)(int16_t)(raw[r] & 0xFFFFu), (float)((int32_t)raw[r] >> 16)};
        z[brev5(r)] = x * w;
        if ((r & 7) == 7) SCHED_FENCE();
    }
}

// 1024-pt FFT of the windowed line held by this 32-lane half; on return z[j] = X[32 j + l]
SSDR_DEV void fft_line(f32x2 (&z)[32], const f32x2 *s_tw_lane, float *xch, int l)
{
    stage_const<1>(z);
    stage_const<2>(z);
    stage_const<3>(z);
    stage_const<4>(z);
    stage_const<5>(z);
    SCHED_FENCE();

    // transpose: element (g = brev5(l), r) -> lane r, register g; re then im through the same buffer.
    // Rows are written with stride 33 across lanes and read along rows: conflict-free both ways, and the
    // reads of one lane are 132 B apart so they stay single ds_read_b32 landing in the right pair half.
    const int g = __builtin_bitreverse32((uint32_t)l) >> 27;
#pragma unroll
    for (int r = 0; r < 32; r++) xch[g * XPAD + r] = z[r].x;
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; j++) z[j].x = xch[j * XPAD + l];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 32; r++) xch[g * XPAD + r] = z[r].y;
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; j++) z[j].y = xch[j * XPAD + l];
    wave_lds_sync();

    SCHED_FENCE();
    stage_lane<0>(z, s_tw_lane);
    stage_lane<1>(z, s_tw_lane);
    stage_lane<2>(z, s_tw_lane);
    SCHED_FENCE();
    stage_lane<3>(z, s_tw_lane);
    SCHED_FENCE();
    stage_lane<4>(z, s_tw_lane);
    SCHED_FENCE();
}

struct WfItem {                 // one (channel pair, averaging group) work item, wave-uniform except ch/ch_ok
    uint32_t ch, l0, l1, grp;
    bool ch_ok, carry_in, complete;
};

SSDR_DEV WfItem wf_item(const SsdrWfArgs &a, uint64_t item, uint32_t n_pairs, int h)
{
    WfItem it;
    it.grp = (uint32_t)(item / n_pairs);
    const uint32_t pair = (uint32_t)(item - (uint64_t)it.grp * n_pairs);
    const uint32_t ch_raw = 2 * pair + h;
    it.ch_ok = ch_raw < a.n_ch;
    it.ch = it.ch_ok ? ch_raw : a.n_ch - 1;
    // lines [l0, l1) of this batch belong to averaging group `grp`
    const int64_t g0 = (int64_t)it.grp * a.n_avg - a.phase;
    it.l0 = g0 < 0 ? 0u : (uint32_t)g0;
    it.l1 = min((uint32_t)(g0 + a.n_avg), a.n_lines);
    it.carry_in = (it.grp == 0) && (a.phase != 0);
    it.complete = (g0 + (int64_t)a.n_avg) <= (int64_t)a.n_lines;
    return it;
}

// AVG == false: averaging N == 1, every line is an output line (no accumulators at all).
// Lines stream through a software pipeline: the next line's 32 loads per lane are issued as
// soon as the current line has been converted to float, and fly under the whole FFT.
template <bool AVG>
__global__ __launch_bounds__(SSDR_WF_BLOCK, SSDR_WF_WAVES_PER_EU) void ssdr_wf_kernel(SsdrWfArgs a)
{
    __shared__ float s_win[SSDR_NFFT];
    __shared__ f32x2 s_tw[SSDR_TW_STAGE_N];
    __shared__ float s_thr[256];
    __shared__ __attribute__((aligned(16))) float s_xch[WAVES][2][XCH_FLOATS];

    for (int i = threadIdx.x; i < SSDR_NFFT; i += SSDR_WF_BLOCK) s_win[i] = a.win[i];
    for (int i = threadIdx.x; i < SSDR_TW_STAGE_N; i += SSDR_WF_BLOCK) s_tw[i] = f32x2{a.tw_stage[i].x, a.tw_stage[i].y};
    for (int i = threadIdx.x; i < 256; i += SSDR_WF_BLOCK) s_thr[i] = a.thr[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, l = lane & 31;
    float *xch = &s_xch[wave][h][0];
    int16_t *x16 = reinterpret_cast<int16_t *>(xch);
    const u32x4 *x128 = reinterpret_cast<const u32x4 *>(xch);
    const uint32_t n_pairs = (a.n_ch + 1) >> 1;
    const uint64_t n_items = (uint64_t)n_pairs * a.n_groups;
    const uint64_t wave_stride = (uint64_t)gridDim.x * WAVES;

    uint64_t item = (uint64_t)blockIdx.x * WAVES + wave;
    if (item >= n_items) return;
    WfItem it = wf_item(a, item, n_pairs, h);
    uint32_t line = it.l0;
    float cal = a.consts[it.ch].wf_cal_lin;

    uint32_t raw[32];
#if SSDR_WF_PREFETCH
    load_line(a.iq + (uint64_t)it.ch * a.ch_stride + (uint64_t)line * SSDR_NFFT + l, raw);
#endif
    uint32_t acc[AVG ? 16 : 1];
#pragma unroll
    for (int j = 0; j < (AVG ? 16 : 1); j++) acc[j] = 0;

    for (;;) {
        f32x2 z[32];
#if !SSDR_WF_PREFETCH
#if SSDR_WF_ABLATE == 2      // ablation: no global loads
#pragma unroll
        for (int r = 0; r < 32; r++) raw[r] = (uint32_t)(line * 2654435761u + r * 40503u + lane * 97u) & 0x1FFF1FFFu;
#else
        load_line(a.iq + (uint64_t)it.ch * a.ch_stride + (uint64_t)line * SSDR_NFFT + l, raw);
#endif
#endif
#if SSDR_WF_ABLATE != 1
        window_line(raw, s_win + l, z);
        SCHED_FENCE();
#endif

        // what comes next: the following line of this group, or the first line of the next item
        const bool group_end = (line + 1 == it.l1);
        uint64_t nitem = item;
        WfItem nit = it;
        uint32_t nline = line + 1;
        if (group_end) {
            nitem = item + wave_stride;
            if (nitem < n_items) { nit = wf_item(a, nitem, n_pairs, h); nline = nit.l0; }
        }
        const bool has_next = !group_end || nitem < n_items;
#if SSDR_WF_PREFETCH
        if (has_next)
            load_line(a.iq + (uint64_t)nit.ch * a.ch_stride + (uint64_t)nline * SSDR_NFFT + l, raw);
        SCHED_FENCE();
#endif

#if SSDR_WF_ABLATE == 1      // ablation: memory traffic only
#pragma unroll
        for (int j = 0; j < 32; j++) x16[32 * ((j + 16) & 31) + l] = (int16_t)(raw[j] & 0xFF);
#else
        fft_line(z, s_tw + l, xch, l);

        // |X|^2 -> 1-dB byte; bin k = 32 j + l lands at fftshifted position 32 ((j+16)&31) + l
        if (AVG) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const float p0 = fmaf(z[j].x, z[j].x, z[j].y * z[j].y) * cal;
                const float p1 = fmaf(z[j + 16].x, z[j + 16].x, z[j + 16].y * z[j + 16].y) * cal;
                acc[j] += (uint32_t)ssdr_quantise(p0, s_thr) | ((uint32_t)ssdr_quantise(p1, s_thr) << 16);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const float p = fmaf(z[j].x, z[j].x, z[j].y * z[j].y) * cal;
                x16[32 * ((j + 16) & 31) + l] = (int16_t)ssdr_quantise(p, s_thr);
                if ((j & 7) == 7) SCHED_FENCE();
            }
        }

#endif   // SSDR_WF_ABLATE == 1
        if (group_end) {
            if (AVG) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    x16[32 * (j + 16) + l] = (int16_t)(acc[j] & 0xFFFFu);
                    x16[32 * j + l] = (int16_t)(acc[j] >> 16);
                    acc[j] = 0;
                }
            }
            wave_lds_sync();
            int16_t *dst = it.complete ? a.out + ((uint64_t)it.grp * a.n_ch + it.ch) * SSDR_NFFT
                                       : a.acc_out + (uint64_t)it.ch * SSDR_NFFT;
            const int16_t *cin = a.acc_in + (uint64_t)it.ch * SSDR_NFFT;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                u32x4 v = x128[q * 32 + l];
                if (AVG && it.carry_in)         // wave-uniform; sums stay < 2^15 so a 32-bit add is a packed 2x16 add
                    v += reinterpret_cast<const u32x4 *>(cin)[q * 32 + l];
                if (it.ch_ok && (SSDR_WF_ABLATE != 3 || v.x == 0x12345u)) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst) + q * 32 + l);
            }
            wave_lds_sync();
            if (!has_next) break;
            cal = a.consts[nit.ch].wf_cal_lin;
        }
        item = nitem;
        it = nit;
        line = nline;
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
    if (a.n_avg > 1) hipLaunchKernelGGL(ssdr_wf_kernel<true>, dim3(grid), dim3(SSDR_WF_BLOCK), 0, stream, a);
    else hipLaunchKernelGGL(ssdr_wf_kernel<false>, dim3(grid), dim3(SSDR_WF_BLOCK), 0, stream, a);
    return hipGetLastError();
}

// workgroups of the waterfall kernel that are resident per CU (min over both instances)
hipError_t ssdr_wf_blocks_per_cu(int *blocks)
{
    int b0 = 0, b1 = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b0, ssdr_wf_kernel<false>, SSDR_WF_BLOCK, 0);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b1, ssdr_wf_kernel<true>, SSDR_WF_BLOCK, 0);
    if (e != hipSuccess) return e;
    *blocks = b0 < b1 ? b0 : b1;
    return hipSuccess;
}

hipError_t ssdr_launch_quant_selftest(const float *thr, unsigned long long *mismatch, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_quant_selftest_kernel, dim3(2048), dim3(256), 0, stream, thr, mismatch);
    return hipGetLastError();
}
