// ssdr_wf_exact.hip -- the waterfall stage in float64 (ssdr_set_exact_bins): the same definition as ssdr_wf.hip
// (Hann window -> 1024-pt FFT -> |X|^2 cal -> byte = #{k : T[k] <= p} -> fftshift -> sum of N lines), evaluated the way the
// normative float64 definition (NumPy) evaluates it: samples times the float32 window table in float64 (exact products), a
// float64 FFT, float64 power, float32 thresholds compared exactly.  An fp32 FFT lands ~3e-4 of the bins one step off where
// |X| sits within its rounding error of a 1-dB threshold (the guard band of DESIGN.md section 3); a float64 FFT's error
// (1e-15 relative) is ten orders of magnitude below the spacing of anything that can sit there, so these bins equal the
// float64 definition's bit for bit -- north_star's "bit-exact int16 waterfall bins" taken literally.  (Nothing here has to
// follow NumPy's butterfly order for that: any float64 FFT with 1e-15 accuracy gives the same bytes.)
//
// Round 4: the kernel has the fp32 kernel's structure instead of a textbook LDS radix-2.
//   * one wave64 per line, 16 complex doubles per lane (64 VGPRs); lane L loads samples 64 q + L, q = 0..15: every load
//     instruction covers 256 contiguous bytes of the line;
//   * DIT stages 1-4 in registers on the bit-reversed group the lane owns (stage 1 fused with the window, compile-time W_16
//     twiddles), ONE transpose through LDS (re, then im through the same 8.1 KB: row stride 65 doubles and a slot
//     permutation lo4 ^ hi2 make the b64 writes and the b64 reads conflict-free), stages 5-8 in registers with per-lane
//     twiddles from per-stage LDS tables (lanes that share a twiddle read one address: broadcast);
//   * stages 9 and 10 pair lanes 16 and 32 apart: v_permlane16_swap / v_permlane32_swap (new on gfx950) exchange half of
//     the registers so that every lane holds both operands of eight butterflies -- no second trip through the LDS;
//     stage 10's twiddles W^k and W^(k+256) = -j W^k share one table entry (the -j is a different static FMA form);
//   * quantiser: p = (re^2 + im^2) cal in double, truncated TOWARD ZERO to float32 (clear the low 29 mantissa bits, then the
//     conversion is exact): for float32 thresholds T[k] <= p  <=>  T[k] <= trunc32(p), so the fp32 kernel's carry-into-the-
//     count table (ssdr_wf.hip:quantise, exhaustively verified) gives the exact count;
//   * the int16 line is staged through the transpose buffer so that every lane stores 2 x 16 contiguous bytes.
// 256-thread workgroups, three per CU (LDS: 4 x 8.1 KB transpose buffers + 16 KB of tables each), persistent grid.
#include <map>
#include <mutex>
#include <utility>
#include "ssdr_math.h"
#include "ssdr_kernels.h"
#include "ssdr_audio_dev.h"

#ifndef SSDR_WFX_BLOCK
#define SSDR_WFX_BLOCK 256
#endif
#ifndef SSDR_WFX_WAVES_PER_EU
#define SSDR_WFX_WAVES_PER_EU 3
#endif

#ifndef SSDR_PRIO_EXACT
#define SSDR_PRIO_EXACT 1            // wave priority by phase (ssdr_wf.hip: prio_latency_phase)
#endif

namespace {

#define XDEV __device__ __forceinline__
#define XFENCE() __builtin_amdgcn_sched_barrier(0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int WAVES = SSDR_WFX_BLOCK / 64;
constexpr int XROW = 65;                                     // row stride of the transpose buffer, doubles
constexpr int XCH_BYTES = 16 * XROW * 8;                     // 8320 per wave
// LDS map: window (513 doubles), quantiser words, twiddle tables (double2), per-wave transpose buffers
constexpr int LDS_WIN = 0;                                   // 513 doubles: float32 window table x 2^-16 (exact)
constexpr int LDS_LUT0 = 4112;
constexpr int LDS_LUT_END = LDS_LUT0 + SSDR_LUT_N * 4;
constexpr int LDS_TW = (LDS_LUT_END + 15) & ~15;
constexpr int LDS_XCH = LDS_TW + SSDR_TW64_N * 16;
constexpr int LDS_TOTAL = LDS_XCH + WAVES * XCH_BYTES;
static_assert(LDS_XCH % 16 == 0, "alignment");
static_assert(LDS_TOTAL * 3 <= 163840 || SSDR_WFX_BLOCK != 256, "three workgroups per CU");
// table offsets in entries (ssdr_make_tw64)
constexpr int T5 = 0, T6 = 16, T7 = 48, T8 = 112, T9 = 240, T10 = 496;
static_assert(T10 + 256 == SSDR_TW64_N, "table size");

XDEV int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
XDEV void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct cd { double r, i; };

// a = u + w v, b = u - w v in the 6-FMA form (b = 2u - a)
XDEV void bfly(cd &u, cd &v, double wr, double wi)
{
    const double sr = fma(-wi, v.i, u.r), si = fma(wi, v.r, u.i);
    const double ar = fma(wr, v.r, sr), ai = fma(wr, v.i, si);
    const double br = fma(2.0, u.r, -ar), bi = fma(2.0, u.i, -ai);
    u = cd{ar, ai};
    v = cd{br, bi};
}
// the same with the twiddle -j w: (wr', wi') = (wi, -wr)
XDEV void bfly_mjw(cd &u, cd &v, double wr, double wi)
{
    const double sr = fma(wr, v.i, u.r), si = fma(-wr, v.r, u.i);
    const double ar = fma(wi, v.r, sr), ai = fma(wi, v.i, si);
    const double br = fma(2.0, u.r, -ar), bi = fma(2.0, u.i, -ai);
    u = cd{ar, ai};
    v = cd{br, bi};
}
XDEV void bfly_1(cd &u, cd &v)
{
    const cd x = u, t = v;
    u = cd{x.r + t.r, x.i + t.i};
    v = cd{x.r - t.r, x.i - t.i};
}
XDEV void bfly_mj(cd &u, cd &v)                               // w = -j: t = (v.i, -v.r)
{
    const cd x = u, t = v;
    u = cd{x.r + t.i, x.i - t.r};
    v = cd{x.r - t.i, x.i + t.r};
}

__device__ constexpr int brev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }

// stages 2..4 on the lane's 16 registers, twiddle W_16^(k (16 >> S))
template <int S>
XDEV void stage_const(cd (&z)[16])
{
    constexpr double C1 = 0x1.d906bcf328d46p-1, S1 = 0x1.87de2a6aea963p-2, C2 = 0x1.6a09e667f3bcdp-1;      // cos pi/8, sin pi/8, sqrt 1/2
    constexpr double WR[8] = {1.0, C1, C2, S1, 0.0, -S1, -C2, -C1};
    constexpr double WI[8] = {0.0, -S1, -C2, -C1, -1.0, -C1, -C2, -S1};
    constexpr int half = 1 << (S - 1);
#pragma unroll
    for (int k = 0; k < half; k++) {
        const int mi = k * (16 >> S);
#pragma unroll
        for (int blk = 0; blk < 16; blk += 2 * half) {
            const int i = blk + k, j = i + half;
            if (mi == 0) bfly_1(z[i], z[j]);
            else if (mi == 4) bfly_mj(z[i], z[j]);
            else bfly(z[i], z[j], WR[mi], WI[mi]);
        }
    }
}

// stages 5..8 (T = s - 5) on x[rho], rho = bits 7..4 of the a-index; the lane's twiddles w[q] = W_(2^s)^(lo4 + 16 q)
template <int T>
XDEV void stage_lane(cd (&z)[16], const f64x2 (&w)[1 << T])
{
    constexpr int half = 1 << T;
#pragma unroll
    for (int q = 0; q < half; q++) {
#pragma unroll
        for (int blk = 0; blk < 16; blk += 2 * half) {
            const int i = blk + q, j = i + half;
            bfly(z[i], z[j], w[q].x, w[q].y);
        }
    }
}

XDEV void swap16(double &a, double &b)                        // rows of 16 lanes: a's odd rows <-> b's even rows
{
    uint32_t alo = (uint32_t)__double_as_longlong(a), ahi = (uint32_t)(__double_as_longlong(a) >> 32);
    uint32_t blo = (uint32_t)__double_as_longlong(b), bhi = (uint32_t)(__double_as_longlong(b) >> 32);
    asm("v_permlane16_swap_b32 %0, %1" : "+v"(alo), "+v"(blo));
    asm("v_permlane16_swap_b32 %0, %1" : "+v"(ahi), "+v"(bhi));
    a = __longlong_as_double((long long)(((uint64_t)ahi << 32) | alo));
    b = __longlong_as_double((long long)(((uint64_t)bhi << 32) | blo));
}
XDEV void swap32(double &a, double &b)                        // a's lanes 32..63 <-> b's lanes 0..31
{
    uint32_t alo = (uint32_t)__double_as_longlong(a), ahi = (uint32_t)(__double_as_longlong(a) >> 32);
    uint32_t blo = (uint32_t)__double_as_longlong(b), bhi = (uint32_t)(__double_as_longlong(b) >> 32);
    asm("v_permlane32_swap_b32 %0, %1" : "+v"(alo), "+v"(blo));
    asm("v_permlane32_swap_b32 %0, %1" : "+v"(ahi), "+v"(bhi));
    a = __longlong_as_double((long long)(((uint64_t)ahi << 32) | alo));
    b = __longlong_as_double((long long)(((uint64_t)bhi << 32) | blo));
}

// the fp32 kernel's threshold counter (ssdr_wf.hip:quantise) on a float that is already scaled by 2^-48 and clamped
XDEV uint32_t quant_addr(float pc) { return (__float_as_uint(pc) >> (SSDR_LUT_SHIFT - 2)) & ~3u; }
XDEV uint32_t quant_pair(uint32_t w0, uint32_t w1) { return __builtin_amdgcn_perm(w1, w0, 0x0C070C03u); }

// |X|^2 cal in double -> the largest float32 not above it, scaled by 2^-48 and clamped to [0, 1]
// (cal arrives times 2^-48, an exact scaling: the product is the table's argument already; the clamp to [0, 1] rides on the conversion)
XDEV float scaled_power_trunc(cd x, double cal_scaled)
{
    const double p = fma(x.r, x.r, x.i * x.i) * cal_scaled;
    const long long bits = __double_as_longlong(p) & ~0x1FFFFFFFll;       // 23 mantissa bits stay: the conversion below is exact
    const double pt = __longlong_as_double(bits);
    float pc;
    asm("v_cvt_f32_f64_e64 %0, %1 clamp" : "=v"(pc) : "v"(pt));
    return pc;
}

XDEV double dbl_i16lo(uint32_t raw) { return (double)(int32_t)(raw << 16); }            // I * 2^16
XDEV double dbl_i16hi(uint32_t raw) { return (double)(int32_t)(raw & 0xFFFF0000u); }    // Q * 2^16

// One line: raw[q] = sample 64 q + lane (I | Q << 16) -> q01[4 t + mm] = byte of bin k (c = 0) | byte of bin k + 512 (c = 1) << 16,
// k = 256 t + 128 b4 + 64 b5 + 16 mm + lo4 for this lane (b4, b5, lo4 = lane bits 4, 5, 0..3).  Window, FFT, power, threshold count.
XDEV void exact_line_bytes(const uint32_t (&raw)[16], double cal, const unsigned char *smem, unsigned char *xch, const unsigned char *lut,
                           int lane, uint32_t (&q01)[8])
{
    // ---- window (float32 table, products exact in double) with stage 1 folded in: sample n = 64 q + L pairs with
    //      n + 512; w[n + 512] = w[512 - n] (symmetric table of 513)
    cd z[16];
    {
        const int ll = opaque(lane);
        const double *win_up = reinterpret_cast<const double *>(smem + LDS_WIN) + ll;
        const double *win_dn = reinterpret_cast<const double *>(smem + LDS_WIN) + 512 - ll;
        double wu[8], wd[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { wu[q] = win_up[64 * q]; wd[q] = win_dn[-64 * q]; }
        XFENCE();
#pragma unroll
        for (int q = 0; q < 8; q++) {
            // I and Q converted where they sit in the dword, i.e. times 2^16 (one shift / one mask instead of a sign
            // extension each); the window table carries the 2^-16: the products are the same doubles
            const double w0 = wu[q], w1 = wd[q];
            const double xr = dbl_i16lo(raw[q]), xi = dbl_i16hi(raw[q]);
            const double yr = dbl_i16lo(raw[q + 8]), yi = dbl_i16hi(raw[q + 8]);
            const double tr = xr * w0, ti = xi * w0;
            z[brev4(q)] = cd{fma(yr, w1, tr), fma(yi, w1, ti)};
            z[brev4(q) + 1] = cd{fma(-yr, w1, tr), fma(-yi, w1, ti)};
        }
    }
    XFENCE();
    stage_const<2>(z);
    stage_const<3>(z);
    stage_const<4>(z);
    XFENCE();
    // ---- transpose.  Register r of lane L is a-index 16 G + r, G = brev6(L) = hi2 << 4 | rho.  It goes to lane
    //      L' = hi2 << 4 | lo4 (lo4 = r), register rho.  Slot of (rho, hi2, lo4): rho * 65 + hi2 * 16 + (lo4 ^ hi2).
    {
        const int lx = opaque(lane);
        const int G = (int)(__builtin_bitreverse32((uint32_t)lx) >> 26);
        const int hi2w = G >> 4, rho_w = G & 15;
        double *wbase[4];
#pragma unroll
        for (int x = 0; x < 4; x++)
            wbase[x] = reinterpret_cast<double *>(xch) + rho_w * XROW + hi2w * 16 + (x ^ hi2w);
        const int hi2r = lx >> 4, lo4r = lx & 15;
        const double *rbase = reinterpret_cast<const double *>(xch) + hi2r * 16 + (lo4r ^ hi2r);
#pragma unroll
        for (int r = 0; r < 16; r++) wbase[r & 3][r & 12] = z[r].r;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 16; j++) z[j].r = rbase[j * XROW];
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 16; r++) wbase[r & 3][r & 12] = z[r].i;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 16; j++) z[j].i = rbase[j * XROW];
        wave_lds_sync();
    }
    // ---- stages 5..8: per-lane twiddles T_s[q][lo4]
    {
        const f64x2 *twl = reinterpret_cast<const f64x2 *>(smem + LDS_TW) + (opaque(lane) & 15);
        f64x2 w5[1], w6[2], w7[4], w8[8];
        w5[0] = twl[T5];
#pragma unroll
        for (int q = 0; q < 2; q++) w6[q] = twl[T6 + 16 * q];
#pragma unroll
        for (int q = 0; q < 4; q++) w7[q] = twl[T7 + 16 * q];
        XFENCE();
        stage_lane<0>(z, w5);
        stage_lane<1>(z, w6);
        XFENCE();
#pragma unroll
        for (int q = 0; q < 8; q++) w8[q] = twl[T8 + 16 * q];
        XFENCE();
        stage_lane<2>(z, w7);
        XFENCE();
        stage_lane<3>(z, w8);
        XFENCE();
    }
    // ---- stage 9: pairs a-index bit 8 = lane bit 4.  After the swap lanes with bit 4 clear hold u, v of elements
    //      rho = m (registers m, m + 8), the others of rho = m + 8; twiddle W_512^(16 rho + lo4)
    {
        const int lx = opaque(lane);
        const f64x2 *t9 = reinterpret_cast<const f64x2 *>(smem + LDS_TW) + T9 + (lx & 31);     // [m][b4][lo4]
        f64x2 w9[8];
#pragma unroll
        for (int m = 0; m < 8; m++) w9[m] = t9[32 * m];
#pragma unroll
        for (int m = 0; m < 8; m++) { swap16(z[m].r, z[m + 8].r); swap16(z[m].i, z[m + 8].i); }
        XFENCE();
#pragma unroll
        for (int m = 0; m < 8; m++) bfly(z[m], z[m + 8], w9[m].x, w9[m].y);
        XFENCE();
    }
    // ---- stage 10: pairs bit 9 = lane bit 5; register pairs (8 t + mm, 8 t + 4 + mm).  Afterwards the lane holds
    //      elements rho = mm + 4 b5 + 8 b4 of both t; twiddle W_1024^(256 t + 16 rho + lo4) = (-j)^t W^(16 rho + lo4)
    {
        const int lx = opaque(lane);
        const int e = ((lx >> 4) & 1) * 32 + (lx >> 5) * 16 + (lx & 15);                         // [mm][b4][b5][lo4]
        const f64x2 *t10 = reinterpret_cast<const f64x2 *>(smem + LDS_TW) + T10 + e;
        f64x2 w10[4];
#pragma unroll
        for (int mm = 0; mm < 4; mm++) w10[mm] = t10[64 * mm];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int mm = 0; mm < 4; mm++) {
                swap32(z[8 * t + mm].r, z[8 * t + 4 + mm].r);
                swap32(z[8 * t + mm].i, z[8 * t + 4 + mm].i);
            }
        XFENCE();
#pragma unroll
        for (int mm = 0; mm < 4; mm++) {
            bfly(z[mm], z[4 + mm], w10[mm].x, w10[mm].y);
            bfly_mjw(z[8 + mm], z[12 + mm], w10[mm].x, w10[mm].y);
        }
        XFENCE();
    }
    // ---- power, exact threshold count.  Register 8 t + 4 c + mm holds FFT bin
    //      k = 512 c + 256 t + 128 b4 + 64 b5 + 16 mm + lo4; pairs (c = 0, c = 1) share a dword: byte_c0 | byte_c1 << 16
    {
        float pc[16];
        uint32_t e[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            pc[r] = scaled_power_trunc(z[r], cal);
            e[r] = *reinterpret_cast<const uint32_t *>(lut + quant_addr(pc[r]));
        }
        XFENCE();
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int mm = 0; mm < 4; mm++) {
                const int r0 = 8 * t + mm, r1 = r0 + 4;
                q01[4 * t + mm] = quant_pair(__float_as_uint(pc[r0]) + e[r0], __float_as_uint(pc[r1]) + e[r1]);
            }
    }
}

// AVG: averaging N > 1 (accumulators); HOP: lines overlap by half (hop 512)
template <bool AVG, bool HOP>
__global__ __launch_bounds__(SSDR_WFX_BLOCK, SSDR_WFX_WAVES_PER_EU) void ssdr_wf_exact_kernel(SsdrWfArgs a, const double2 *tw_g)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];
    {
        double *s_win = reinterpret_cast<double *>(smem + LDS_WIN);
        uint32_t *s_lut = reinterpret_cast<uint32_t *>(smem + LDS_LUT0);
        f64x2 *s_tw = reinterpret_cast<f64x2 *>(smem + LDS_TW);
        for (int i = threadIdx.x; i < 513; i += blockDim.x) s_win[i] = (double)a.win[i] * 0x1p-16;       // (the samples arrive times 2^16)
        for (int i = threadIdx.x; i < SSDR_LUT_N; i += blockDim.x) s_lut[i] = a.lut[i];
        for (int i = threadIdx.x; i < SSDR_TW64_N; i += blockDim.x) s_tw[i] = f64x2{tw_g[i].x, tw_g[i].y};
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *xch = smem + LDS_XCH + wave * XCH_BYTES;                      // wave-uniform
    const unsigned char *lut = smem + LDS_LUT0;
    const uint32_t run = HOP ? a.grp_run : 1u;
    const uint32_t n_runs = (a.n_groups + run - 1) / run;
    const uint32_t n_items = a.n_ch * n_runs;
    const uint32_t wave_stride = gridDim.x * WAVES;

    constexpr uint32_t LINE_STEP = HOP ? SSDR_NFFT / 2 : SSDR_NFFT;
    // work items: hop 1024 -- one (group, channel) each, group-major; hop 512 -- a run of consecutive groups of one channel
    auto decode = [&](uint32_t it, uint32_t &ch_, uint32_t &gb, uint32_t &ge) {
        if (HOP) {
            ch_ = it / n_runs;
            gb = (it - ch_ * n_runs) * run;
            ge = min(gb + run, a.n_groups);
        } else {
            gb = it / a.n_ch;
            ch_ = it - gb * a.n_ch;
            ge = gb + 1;
        }
    };
    // lane L fetches samples 64 q + L of line `ln` of channel `ch_` (hop 512: the older half, then the new one)
    auto load_raw = [&](uint32_t ch_, uint32_t ln, uint32_t (&raw_)[16]) {
        const uint32_t *src_ = a.iq + (uint64_t)ch_ * a.ch_stride + (uint64_t)ln * LINE_STEP + lane;
        if (HOP) {
            const uint32_t *older = ln ? src_ - SSDR_NFFT / 2 : a.tail + (uint64_t)ch_ * (SSDR_NFFT / 2) + lane;
#pragma unroll
            for (int q = 0; q < 8; q++) raw_[q] = __builtin_nontemporal_load(older + 64 * q);
#pragma unroll
            for (int q = 0; q < 8; q++) raw_[8 + q] = src_[64 * q];
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) raw_[q] = __builtin_nontemporal_load(src_ + 64 * q);
        }
    };
    uint32_t raw[16];

    for (uint32_t item = blockIdx.x * WAVES + wave; item < n_items; item += wave_stride) {
        uint32_t ch, g_begin, g_end;
        decode(item, ch, g_begin, g_end);
        for (uint32_t grp = g_begin; grp < g_end; grp++) {
            uint32_t ch_now = __builtin_amdgcn_readfirstlane(ch);
            asm volatile("" : "+s"(ch_now));
            const int64_t g0 = (int64_t)grp * a.n_avg - a.phase;
            const uint32_t l0 = g0 < 0 ? 0u : (uint32_t)g0;
            const uint32_t l1 = min((uint32_t)(g0 + a.n_avg), a.n_lines);
            const bool carry_in = (grp == 0) && (a.phase != 0);
            const bool complete = (g0 + (int64_t)a.n_avg) <= (int64_t)a.n_lines;
            const double cal = (double)a.consts[ch_now].wf_cal_lin * (double)SSDR_LUT_SCALE;
            uint32_t acc[AVG ? 8 : 1];
#pragma unroll
            for (int j = 0; j < (AVG ? 8 : 1); j++) acc[j] = 0;
            for (uint32_t line = l0; line < l1; line++) {
                // ---- the line: lane L holds samples 64 q + L
                load_raw(ch_now, line, raw);
                if (SSDR_PRIO_EXACT) __builtin_amdgcn_s_setprio(0);        // the butterflies yield to waves that load, look up or store
                uint32_t q01[8];
                exact_line_bytes(raw, cal, smem, xch, lut, lane, q01);
                if (SSDR_PRIO_EXACT) __builtin_amdgcn_s_setprio(3);
                if (AVG) {
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += q01[j];
                }
                if (!AVG || line + 1 == l1) {
                    // ---- the int16 line through the transpose buffer: output position j = k ^ 512 (fftshift)
                    const int lx = opaque(lane);
                    int16_t *x16 = reinterpret_cast<int16_t *>(xch) + ((lx >> 4) & 1) * 128 + (lx >> 5) * 64 + (lx & 15);
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int mm = 0; mm < 4; mm++) {
                            const uint32_t v = AVG ? acc[4 * t + mm] : q01[4 * t + mm];
                            x16[512 + 256 * t + 16 * mm] = (int16_t)(v & 0xFFFFu);       // c = 0: bin k < 512 -> upper half of the line
                            x16[256 * t + 16 * mm] = (int16_t)(v >> 16);                // c = 1 -> lower half
                        }
                    wave_lds_sync();
                    const u32x4 *x128 = reinterpret_cast<const u32x4 *>(xch);
                    int16_t *dst = complete ? a.out + ((uint64_t)grp * a.n_ch + ch_now) * SSDR_NFFT : a.acc_out + (uint64_t)ch_now * SSDR_NFFT;
                    const int16_t *cin = a.acc_in + (uint64_t)ch_now * SSDR_NFFT;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        u32x4 v = x128[q * 64 + lx];
                        if (AVG && carry_in) v += reinterpret_cast<const u32x4 *>(cin)[q * 64 + lx];     // sums < 2^15: a packed 2 x 16 add
                        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst) + q * 64 + lx);
                    }
                    wave_lds_sync();
                }
            }
        }
    }
}

// ---- both stages on one read of the input, float64 bins (the counterpart of ssdr_wf.hip:ssdr_fused_am_kernel<false, false>) -----------
// ssdr_run_chain's kernel when ssdr_set_exact_bins is on and the batch is the metric's configuration: every channel on the full-band AM
// path, N = 1, hop 1024.  A wave owns a channel for the whole call (the audio chain is sequential in time) and walks its lines.  Per
// line: 16 streaming loads per lane bring the 4 KB line in the FFT's layout (lane L: samples 64 q + L) and stay in registers for the
// FFT; a copy parks in the wave's transpose buffer, from which every lane takes its eight consecutive samples of each of the two
// frames and runs exactly the stand-alone AM kernel's code (integer powers, demod_am, agc_pack_store, rssi_flag_step: same scans, same
// orders); then exact_line_bytes as in the kernel above.  The audio chain runs at wave priority 3, the FFT at 0: the float64
// butterflies of the other waves fill the time the scans wait.  Bit-identical to ssdr_wf_exact_kernel + ssdr_audio_kernel<2>.
__global__ __launch_bounds__(SSDR_WFX_BLOCK, SSDR_WFX_WAVES_PER_EU) void ssdr_fused_exact_am_kernel(SsdrFusedArgs fa, const double2 *tw_g)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];
    const SsdrWfArgs &a = fa.wf;
    const SsdrAudioArgs &u = fa.au;
    {
        double *s_win = reinterpret_cast<double *>(smem + LDS_WIN);
        uint32_t *s_lut = reinterpret_cast<uint32_t *>(smem + LDS_LUT0);
        f64x2 *s_tw = reinterpret_cast<f64x2 *>(smem + LDS_TW);
        for (int i = threadIdx.x; i < 513; i += blockDim.x) s_win[i] = (double)a.win[i] * 0x1p-16;
        for (int i = threadIdx.x; i < SSDR_LUT_N; i += blockDim.x) s_lut[i] = a.lut[i];
        for (int i = threadIdx.x; i < SSDR_TW64_N; i += blockDim.x) s_tw[i] = f64x2{tw_g[i].x, tw_g[i].y};
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *xch = smem + LDS_XCH + wave * XCH_BYTES;
    const unsigned char *lut = smem + LDS_LUT0;
    uint32_t *q32 = reinterpret_cast<uint32_t *>(xch);                           // the raw line, sample n at q32[n]
    const uint32_t wave_stride = gridDim.x * WAVES;
    const uint32_t n_frames = u.n_frames;

    for (uint32_t ch_v = blockIdx.x * WAVES + wave; ch_v < a.n_ch; ch_v += wave_stride) {
        uint32_t ch = __builtin_amdgcn_readfirstlane(ch_v);
        asm volatile("" : "+s"(ch));
        const ssdr_chan_consts &kc = u.consts[ch];
        const AgcK agc_c = {kc.agc_c0, kc.agc_c1, kc.agc_knee, kc.agc_delta8, kc.hang_frames};
        const float cal_c = kc.smeter_cal_db;
        const double cal = (double)kc.wf_cal_lin * (double)SSDR_LUT_SCALE;
        // the audio chain's carried state (wave-uniform) and the two per-lane keepers
        float dc, agc_d, agc_m[8];
        uint32_t tail_q[4];
        {
            const ssdr_chan_state st = u.state[ch];
            dc = st.dc; agc_d = st.agc_d;
#pragma unroll
            for (int i = 0; i < 8; i++) agc_m[i] = st.agc_m[i];
            const uint4 t = *reinterpret_cast<const uint4 *>(u.hist + (size_t)ch * SSDR_HIST + SSDR_HIST - 4);
            tail_q[0] = iq_power(t.x); tail_q[1] = iq_power(t.y); tail_q[2] = iq_power(t.z); tail_q[3] = iq_power(t.w);
        }
        float rssi_sum = 0.0f;
        uint32_t flag_keep = 0u;
        uint32_t raw[16];
        const uint32_t *src = a.iq + (uint64_t)ch * a.ch_stride + lane;
        if (SSDR_PRIO_EXACT) __builtin_amdgcn_s_setprio(3);
        for (uint32_t line = 0; line < a.n_lines; line++, src += SSDR_NFFT) {
#pragma unroll
            for (int q = 0; q < 16; q++) raw[q] = __builtin_nontemporal_load(src + 64 * q);
            XFENCE();
            {
                uint32_t *qw = q32 + opaque(lane);
#pragma unroll
                for (int q = 0; q < 16; q++) qw[64 * q] = raw[q];
            }
            wave_lds_sync();
            // ---- audio: two frames, lane l on samples 8 l .. 8 l + 7 of each
#pragma unroll
            for (int f = 0; f < 2; f++) {
                const uint32_t frame = 2 * line + f;
                const u32x4 *qp = reinterpret_cast<const u32x4 *>(q32 + SSDR_FRAME * f) + 2 * opaque(lane);
                const u32x4 q0 = qp[0], q1 = qp[1];
                const uint32_t rw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                uint32_t qv[8], d[8];
#pragma unroll
                for (int j = 0; j < 8; j++) qv[j] = iq_power(rw[j]);
                float p[8], aud[8];
#pragma unroll
                for (int j = 0; j < 4; j++) { d[j] = from_prev_lane_u(tail_q[j], qv[4 + j]); d[4 + j] = qv[j]; }
#pragma unroll
                for (int j = 0; j < 4; j++) tail_q[j] = lane63_u(qv[4 + j]);
#pragma unroll
                for (int j = 0; j < 8; j++) p[j] = (float)d[j];
                const float pmx = block_peak(p);
                const bool trig = wave_any(pmx >= 1073676160.0f) || tail_q[0] >= 0x3FFF0001u || tail_q[1] >= 0x3FFF0001u ||
                                  tail_q[2] >= 0x3FFF0001u || tail_q[3] >= 0x3FFF0001u;
                const bool clip = trig ? wave_any(raw_clipped(rw)) : false;
                demod_am<true>(p, dc, aud);
                agc_pack_store(p, aud, lane, agc_c, agc_d, agc_m, u.pcm + ((uint64_t)ch * n_frames + frame) * SSDR_FRAME + 8 * lane, pmx);
                rssi_flag_step(p, clip, frame, n_frames, lane, cal_c, rssi_sum, flag_keep, u.rssi + (uint64_t)ch * n_frames, u.flags + (uint64_t)ch * n_frames);
            }
            wave_lds_sync();
            // ---- waterfall: the line is still in registers
            if (SSDR_PRIO_EXACT) __builtin_amdgcn_s_setprio(0);
            uint32_t q01[8];
            exact_line_bytes(raw, cal, smem, xch, lut, lane, q01);
            if (SSDR_PRIO_EXACT) __builtin_amdgcn_s_setprio(3);
            {
                const int lx = opaque(lane);
                int16_t *x16 = reinterpret_cast<int16_t *>(xch) + ((lx >> 4) & 1) * 128 + (lx >> 5) * 64 + (lx & 15);
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int mm = 0; mm < 4; mm++) {
                        const uint32_t v = q01[4 * t + mm];
                        x16[512 + 256 * t + 16 * mm] = (int16_t)(v & 0xFFFFu);
                        x16[256 * t + 16 * mm] = (int16_t)(v >> 16);
                    }
                wave_lds_sync();
                const u32x4 *x128 = reinterpret_cast<const u32x4 *>(xch);
                int16_t *dst = a.out + ((uint64_t)line * a.n_ch + ch) * SSDR_NFFT;
#pragma unroll
                for (int q = 0; q < 2; q++) __builtin_nontemporal_store(x128[q * 64 + lx], reinterpret_cast<u32x4 *>(dst) + q * 64 + lx);
                wave_lds_sync();
            }
        }
        // ---- state back to HBM
        if (a.n_lines) {
            // the raw tail of the call's last frame (its samples 384..511 = the line's 896..1023 = q 14, 15) is the next call's history
#pragma unroll
            for (int q = 14; q < 16; q++) u.hist[(size_t)ch * SSDR_HIST + 64 * (q - 14) + lane] = raw[q];
            // the discriminator memory an AM channel leaves behind: y[511] = z1[507] of the last frame, mixed as the twin does (block 63 of
            // the frame, element 3).  Sample 507 of that frame is the line's sample 1019 = raw[15] of lane 59.
            ssdr_chan_state st = u.state[ch];
            const uint32_t phi_last = st.phi1 + (uint32_t)(SSDR_FRAME * (n_frames - 1)) * kc.dphi1;
            float fc, fs, qc, qs, bc, bs, cs, ss;
            ssdr_phasor32(phi_last, fc, fs);
            ssdr_phasor32((uint32_t)(8 * 63) * kc.dphi1, qc, qs);
            ssdr_phasor32(kc.dphi1, cs, ss);
            phasor_mul(fc, fs, qc, qs, bc, bs);
#pragma unroll
            for (int j = 0; j < 3; j++) { const float cn = fmaf(bc, cs, -(bs * ss)), sn = fmaf(bs, cs, bc * ss); bc = cn; bs = sn; }
            const float xr = (float)(int16_t)(raw[15] & 0xFFFFu), xi = (float)((int32_t)raw[15] >> 16);
            const float zr = fmaf(xr, bc, xi * bs) + 0.0f, zi = fmaf(xi, bc, -(xr * bs)) + 0.0f;
            st.prev_re = lane_f(zr, 59);
            st.prev_im = lane_f(zi, 59);
            st.phi1 += (uint32_t)(SSDR_FRAME * n_frames) * kc.dphi1;
            st.phi2 += (uint32_t)(SSDR_FRAME * n_frames) * kc.dphi2;
            st.dc = dc; st.agc_d = agc_d;
#pragma unroll
            for (int i = 0; i < 8; i++) st.agc_m[i] = agc_m[i];
            if (lane == 0) u.state[ch] = st;
        }
    }
}

} // namespace

// The persistent grids' sizes are a property of the DEVICE a launch goes to: cached per device id, under a lock (a process may hold
// contexts on several GPUs and make its first launches from several threads).  which: 0 the waterfall kernels, 1 the fused kernel.
static bool resident_cached(int which, int dev, uint32_t *blocks, bool store)
{
    static std::mutex m;
    static std::map<std::pair<int, int>, uint32_t> cache;
    std::lock_guard<std::mutex> lock(m);
    if (store) { cache[{which, dev}] = *blocks; return true; }
    auto it = cache.find({which, dev});
    if (it == cache.end()) return false;
    *blocks = it->second;
    return true;
}

// resident workgroups of the kernel on the current device (persistent grid)
static hipError_t wfx_resident(uint32_t *blocks)
{
    int dev = 0;
    {
        hipError_t e0 = hipGetDevice(&dev);
        if (e0 != hipSuccess) return e0;
    }
    uint32_t cached = 0;
    if (!resident_cached(0, dev, &cached, false)) {
        int per_cu = 0, b = 0;
        hipError_t e;
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        per_cu = 1 << 30;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ssdr_wf_exact_kernel<false, false>, SSDR_WFX_BLOCK, 0)) != hipSuccess) return e;
        per_cu = b < per_cu ? b : per_cu;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ssdr_wf_exact_kernel<true, false>, SSDR_WFX_BLOCK, 0)) != hipSuccess) return e;
        per_cu = b < per_cu ? b : per_cu;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ssdr_wf_exact_kernel<false, true>, SSDR_WFX_BLOCK, 0)) != hipSuccess) return e;
        per_cu = b < per_cu ? b : per_cu;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ssdr_wf_exact_kernel<true, true>, SSDR_WFX_BLOCK, 0)) != hipSuccess) return e;
        per_cu = b < per_cu ? b : per_cu;
        cached = (uint32_t)prop.multiProcessorCount * (uint32_t)(per_cu < 1 ? 1 : per_cu);
        resident_cached(0, dev, &cached, true);
    }
    *blocks = cached;
    return hipSuccess;
}

hipError_t ssdr_launch_wf_exact(const SsdrWfArgs &a_in, const double2 *tw, hipStream_t stream)
{
    SsdrWfArgs a = a_in;
    if (a.n_groups == 0 || a.n_ch == 0) return hipSuccess;
    uint32_t resident = 0;
    hipError_t e = wfx_resident(&resident);
    if (e != hipSuccess) return e;
    constexpr uint32_t waves = SSDR_WFX_BLOCK / 64;
    uint64_t items = (uint64_t)a.n_ch * a.n_groups;
    a.grp_run = 1;
    if (a.tail) {            // hop 512: runs of consecutive groups per wave (the shared half-line is re-read by the wave that fetched it)
        uint64_t run = items / (8ull * resident * waves);
        run = run < 1 ? 1 : (run > a.n_groups ? a.n_groups : run);
        a.grp_run = (uint32_t)run;
        items = (uint64_t)a.n_ch * ((a.n_groups + run - 1) / run);
    }
    const uint64_t need = (items + waves - 1) / waves;
    const dim3 g((uint32_t)(need < resident ? need : resident)), b(SSDR_WFX_BLOCK);
    if (a.tail) {
        if (a.n_avg > 1) hipLaunchKernelGGL((ssdr_wf_exact_kernel<true, true>), g, b, 0, stream, a, tw);
        else hipLaunchKernelGGL((ssdr_wf_exact_kernel<false, true>), g, b, 0, stream, a, tw);
    } else {
        if (a.n_avg > 1) hipLaunchKernelGGL((ssdr_wf_exact_kernel<true, false>), g, b, 0, stream, a, tw);
        else hipLaunchKernelGGL((ssdr_wf_exact_kernel<false, false>), g, b, 0, stream, a, tw);
    }
    return hipGetLastError();
}

// the fused float64 kernel: one wave per channel, persistent grid
hipError_t ssdr_launch_fused_exact_am(const SsdrFusedArgs &a, const double2 *tw, hipStream_t stream)
{
    if (a.wf.n_ch == 0 || a.wf.n_lines == 0) return hipSuccess;
    uint32_t resident = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!resident_cached(1, dev, &resident, false)) {
        int b = 0;
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ssdr_fused_exact_am_kernel, SSDR_WFX_BLOCK, 0)) != hipSuccess) return e;
        resident = (uint32_t)prop.multiProcessorCount * (uint32_t)(b < 1 ? 1 : b);
        resident_cached(1, dev, &resident, true);
    }
    constexpr uint32_t waves = SSDR_WFX_BLOCK / 64;
    const uint64_t need = ((uint64_t)a.wf.n_ch + waves - 1) / waves;
    hipLaunchKernelGGL(ssdr_fused_exact_am_kernel, dim3((uint32_t)(need < resident ? need : resident)), dim3(SSDR_WFX_BLOCK), 0, stream, a, tw);
    return hipGetLastError();
}
