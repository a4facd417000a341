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
//   * power, quantiser (one LDS table look-up + one compare, see quantise()), N-line
//     accumulation in registers, then the line is staged through the (now free)
//     exchange buffer so every lane stores 16 contiguous bytes (512 B per half-wave
//     instruction) with the fftshift folded into the LDS address.
//   The butterflies are exactly those of a textbook radix-2 DIT FFT (same operand
//   pairs, same six fused multiply-adds), only regrouped -- results are bit-identical.
//
// What bounds it (profiles/): VALU lane-operations and, at steady state, the board's power cap -- not HBM.  On gfx950
// a wave64 v_fma/add/mul/mov_f32 or simple integer add/and takes 2 cycles of its SIMD; shifts, conversions, med3,
// compares and packed-fp32 ops (two lane-operations) take 4.  Hence: 6-FMA butterflies (packing them buys nothing),
// a quantiser with a single slow op chain, and two 512-thread workgroups per CU (16 waves, 4 per SIMD, <= 128 VGPRs)
// that share the 160 KB of LDS.
#include "ssdr_math.h"
#include "ssdr_kernels.h"
#include "ssdr_audio_dev.h"

#ifndef SSDR_FUSED_ABLATE
#define SSDR_FUSED_ABLATE 0                  // timing ablations of the fused kernel only (1: no FFT, 2: no audio chain)
#endif
#ifndef SSDR_FUSED_WIDE_LOADS
#define SSDR_FUSED_WIDE_LOADS 1              // the fused kernel fetches a line 16 bytes per lane (A/B: 0 = 4 bytes per lane in the FFT's layout)
#endif
#ifndef SSDR_WF_PAIR_MAJOR
#define SSDR_WF_PAIR_MAJOR 0
#endif
#ifndef SSDR_WF_BLOCKED_ITEMS
#define SSDR_WF_BLOCKED_ITEMS 0              // A/B: each wave takes a contiguous range of work items instead of a strided one
#endif

namespace {

constexpr int XPAD = 33;                       // row stride (floats) of the transpose buffer
constexpr int XCH_FLOATS = 32 * XPAD;          // per FFT: 4224 B
constexpr int WAVES = SSDR_WF_BLOCK / 64;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS map of the waterfall kernel (one allocation starting at LDS address 0):
//   window (513 floats: first half + midpoint, w[n] = w[1024-n]), quantiser table (its constant base rides in the
//   DS instruction's offset field), per-stage twiddles for FFT stages 6..10 (992 float2), then the per-wave
//   transpose / staging buffers (2 x 4224 B).
constexpr int LDS_WIN = 0;                                      // [0, 2064)
constexpr int LDS_LUT0 = 2064;                                  // [2064, 4100)
constexpr int LDS_LUT_END = LDS_LUT0 + SSDR_LUT_N * 4;
constexpr int LDS_TW = (LDS_LUT_END + 15) & ~15;                // [4112, 12048)
constexpr int LDS_XCH = LDS_TW + SSDR_TW_STAGE_N * 8;
constexpr int LDS_TOTAL = LDS_XCH + WAVES * 2 * XCH_FLOATS * 4;
static_assert(LDS_XCH % 16 == 0 && LDS_TW % 8 == 0, "alignment");
static_assert(LDS_TOTAL <= 163840, "LDS budget");

// The register budget only holds if the phases of a line stay phases: without these fences
// the machine scheduler hoists later phases' LDS table reads across the whole FFT and spills.
// (Scheduling fence only; emits no instruction.)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Wave priority by phase (s_setprio): a SIMD issues from its highest-priority ready wave.  The butterfly stages have six independent
// FMAs per butterfly and 16 butterflies per stage to pick from -- they can always issue; the audio chain's scans (DPP, dependent
// chains), the quantiser's table look-ups, the loads and the stores mostly wait.  A wave in such a phase gets the issue slot the
// moment it can use it, waves in the FFT take what is left: the four waves of a SIMD spread over the phases instead of
// convoying through them.  +7 % on the fused kernel (profiles/r04_ab_wave_priority.txt); any level above 0 does it.
#ifndef SSDR_PRIO
#define SSDR_PRIO 1
#endif
#ifndef SSDR_PRIO_WF
#define SSDR_PRIO_WF 1
#endif
SSDR_DEV void prio_latency_phase() { if (SSDR_PRIO) __builtin_amdgcn_s_setprio(3); }
SSDR_DEV void prio_compute_phase() { if (SSDR_PRIO) __builtin_amdgcn_s_setprio(0); }

__device__ constexpr int brev5(int v)
{
    return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

// Radix-2 DIT butterfly in its 6-FMA form (Linzer-Feig / Goedecker), w = (wr, wi):
//     sr = fma(-wi, vi, ur)     si = fma(wi, vr, ui)
//     ar = fma( wr, vr, sr)     ai = fma(wr, vi, si)          a = u + w*v
//     br = fma(  2, ur, -ar)    bi = fma( 2, ui, -ai)         b = 2u - a = u - w*v
// 6 full-rate fp32 ops instead of 8 (DESIGN.md section 3 spells out the six roundings).
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
    // two packed adds whose operand modifiers swap and negate v's halves: a = (ur + vi, ui - vr), b = (ur - vi, ui + vr)
    // (written as x + t, x - t with t = {v.y, -v.x} the compiler builds t with two moves and a sign flip per butterfly)
    f32x2 a, b;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(a) : "v"(u), "v"(v));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(b) : "v"(u), "v"(v));
    u = a; v = b;
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
// The twiddles are passed in registers: the caller loads them from the LDS table one stage (or half a
// stage) AHEAD of their use, so that no butterfly ever waits for an LDS round trip.
template <int T, int JL0, int NJL>
SSDR_DEV void stage_lane(f32x2 (&z)[32], const f32x2 (&w)[NJL])
{
    constexpr int half = 1 << T;
#pragma unroll
    for (int q = 0; q < NJL; q++) {
        const int jl = JL0 + q;
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * half) {
            const int i = blk + jl, j = i + half;
            bfly(z[i], z[j], w[q].x, w[q].y);
        }
    }
}

template <int T, int JL0, int NJL>
SSDR_DEV void load_tw(f32x2 (&w)[NJL], const f32x2 *tw_lane)
{
    constexpr int off = 32 * ((1 << T) - 1);
#pragma unroll
    for (int q = 0; q < NJL; q++) w[q] = tw_lane[off + (JL0 + q) * 32];
}

// Per-lane LDS base addresses are all cheap functions of the lane id.  Left alone, the compiler keeps
// a dozen of them live across the whole line (and spills them at the 128-VGPR budget); laundering
// the lane id through an empty asm makes each phase recompute its own base in one or two fast ops.
SSDR_DEV int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
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

// dB quantiser: byte = #{k in 1..255 : T[k] <= p}, exactly, without a logarithm, a compare or a select.
// p arrives scaled by 2^-48 (exact; the calibration factor carries it) and clamped to [0, 1] by the multiply that
// produced it: T[255] = 2^48 is 1.0 there, everything below T[1] (zero and denormals included) sits in segments that
// count 0.  A float's top bits (exponent, SSDR_LUT_BITS mantissa bits) name a segment narrower than 1 dB; a segment
// contains at most one 1-dB threshold, and the host tabulates per segment one word (ssdr_tables.cpp:ssdr_make_quant_lut)
// such that
//     byte = (bits(p') + word[segment]) >> 24
// -- the distance of p' from the threshold carries into the count.  Split in two so that the table reads of a whole
// batch are in flight together (and the next batch's are issued before this batch's adds): quant_addr -> load ->
// quant_word; the callers take the top bytes out pairwise with one v_perm_b32.
SSDR_DEV float quant_scaled_power(f32x2 z, float calq)
{
    float p = fmaf(z.x, z.x, z.y * z.y), pc;
    asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(pc) : "v"(p), "v"(calq));
    return pc;
}
SSDR_DEV uint32_t quant_addr(float pc) { return (__float_as_uint(pc) >> (SSDR_LUT_SHIFT - 2)) & ~3u; }
SSDR_DEV uint32_t quant_word(float pc, uint32_t w) { return __float_as_uint(pc) + w; }
// bytes of two words -> byte(w0) | byte(w1) << 16
SSDR_DEV uint32_t quant_pair(uint32_t w0, uint32_t w1) { return __builtin_amdgcn_perm(w1, w0, 0x0C070C03u); }
SSDR_DEV uint32_t quantise(float p_scaled_clamped, const unsigned char *lut)
{
    return quant_word(p_scaled_clamped, *reinterpret_cast<const uint32_t *>(lut + quant_addr(p_scaled_clamped))) >> 24;
}

// power + quantiser for the 32 bins of a lane, in 4 batches of 8 (bins 0-7, 16-23, 8-15, 24-31) with the table
// reads software-pipelined one batch ahead and no LDS store in between, so nothing orders one look-up behind
// another.  `sink(j, byte_j | byte_{j+16} << 16)` receives the results pairwise (j = 0..15).
template <typename Sink>
SSDR_DEV void quantise32(const f32x2 (&z)[32], float calq, const unsigned char *lut, Sink sink)
{
    constexpr int ORDER[4] = {0, 16, 8, 24};
    float pc[2][8];
    uint32_t e[2][8];
    uint32_t lo[8];
#pragma unroll
    for (int b = 0; b < 5; b++) {
        if (b < 4) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int j = ORDER[b] + i;
                pc[b & 1][i] = quant_scaled_power(z[j], calq);
                e[b & 1][i] = *reinterpret_cast<const uint32_t *>(lut + quant_addr(pc[b & 1][i]));
            }
        }
        SCHED_FENCE();
        if (b > 0) {
            const int pb = b - 1;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t r = quant_word(pc[pb & 1][i], e[pb & 1][i]);
                if ((pb & 1) == 0) lo[i] = r;                       // bins j (batch 0 or 2)
                else sink(ORDER[pb - 1] + i, quant_pair(lo[i], r)); // bins j+16 arrive one batch later
            }
        }
    }
}

SSDR_DEV void load_line(const uint32_t *__restrict__ src /* + lane */, uint32_t (&raw)[32])
{
#pragma unroll
    for (int r = 0; r < 32; r++) raw[r] = SSDR_NT_LOAD(src + 32 * r);
}
// hop 512: a line is the previous half-line followed by a new one.  The older half is the previous line's newer half,
// which this very wave fetched one line earlier (a wave walks a run of consecutive lines of its channel pair, see the
// item loop of ssdr_wf_kernel): the second read is served by the L2 -- plain loads, not non-temporal ones.
SSDR_DEV void load_line_halves(const uint32_t *__restrict__ older, const uint32_t *__restrict__ newer, uint32_t (&raw)[32])
{
#pragma unroll
    for (int r = 0; r < 16; r++) raw[r] = SSDR_NT_LOAD(older + 32 * r);        // its last use: do not keep it
#pragma unroll
    for (int r = 0; r < 16; r++) raw[16 + r] = newer[32 * r];
}

// raw int16 IQ dwords of one line -> windowed complex samples in a-index (bit-reversed) order, with FFT stage 1 folded in.
// The window is symmetric, w[n] = w[1024-n]: samples of the second half read the same 513-entry table
// backwards from a second per-lane base.
// Stage 1 pairs sample n with sample n + 512 (registers r and r + 16 of a lane), twiddle 1: a = x w + x' w', b = x w - x' w'.
// The second product is not rounded on its own: t = x w, a = fma(x', w', t), b = fma(-x', w', t) -- three operations per
// pair and component instead of four (the twin states the same).
// Packed multiply / multiply-add of a complex sample by ONE real factor that sits in half H of a register pair: the
// operand modifiers broadcast that half to both lanes of the packed operation (the compiler only knows the broadcast
// of a pair's low half and moves a factor there first: one v_mov per window value).
template <int H>
SSDR_DEV f32x2 pk_mul_half(f32x2 x, f32x2 wpair)
{
    f32x2 r;
    if (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(wpair));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(wpair));
    return r;
}
template <int H, bool NEG>
SSDR_DEV f32x2 pk_fma_half(f32x2 x, f32x2 wpair, f32x2 t)          // (NEG ? -x : x) * wpair[H] + t
{
    f32x2 r;
    if (H == 0 && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(wpair), "v"(t));
    if (H == 1 && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "v"(wpair), "v"(t));
    if (H == 0 && NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(wpair), "v"(t));
    if (H == 1 && NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(wpair), "v"(t));
    return r;
}

SSDR_DEV void window_line(const uint32_t (&raw)[32], const unsigned char *smem, int l, f32x2 (&z)[32])
{
    const int ll = opaque(l);
    const float *win_up = reinterpret_cast<const float *>(smem + LDS_WIN) + ll;
    const float *win_dn = reinterpret_cast<const float *>(smem + LDS_WIN) - ll;
    // all 32 window values first: their LDS latency hides under the HBM latency of the line's samples.  Pairs of rows
    // (2k, 2k + 1) share a register pair (one ds_read2_b32 each): wu for samples n < 512, wd for their partners n + 512
    // (the mirrored half of the table: its pair is held in address order, row 2k + 1 in the low half).
    f32x2 wu[8], wd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        wu[k] = f32x2{win_up[32 * (2 * k)], win_up[32 * (2 * k + 1)]};
        wd[k] = f32x2{win_dn[32 * (16 - (2 * k + 1))], win_dn[32 * (16 - 2 * k)]};      // ascending addresses: row 2k+1 first
    }
    SCHED_FENCE();
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int r = 2 * k + hh;
            const f32x2 xa = {(float)(int16_t)(raw[r] & 0xFFFFu), (float)((int32_t)raw[r] >> 16)};
            const f32x2 xb = {(float)(int16_t)(raw[r + 16] & 0xFFFFu), (float)((int32_t)raw[r + 16] >> 16)};
            if (hh == 0) {
                const f32x2 t = pk_mul_half<0>(xa, wu[k]);
                z[brev5(r)] = pk_fma_half<1, false>(xb, wd[k], t);
                z[brev5(r) + 1] = pk_fma_half<1, true>(xb, wd[k], t);
            } else {
                const f32x2 t = pk_mul_half<1>(xa, wu[k]);
                z[brev5(r)] = pk_fma_half<0, false>(xb, wd[k], t);
                z[brev5(r) + 1] = pk_fma_half<0, true>(xb, wd[k], t);
            }
        }
        if (k & 1) SCHED_FENCE();
    }
}

// 1024-pt FFT of the windowed line held by this 32-lane half; on return z[j] = X[32 j + l]
template <bool TIGHT>
SSDR_DEV void fft_line(f32x2 (&z)[32], const unsigned char *smem, float *xch_wave, int h, int l)
{
    stage_const<2>(z);                             // stage 1 came with the window (window_line)
    stage_const<3>(z);
    stage_const<4>(z);
    stage_const<5>(z);
    SCHED_FENCE();

    // transpose: element (g = brev5(l), r) -> lane r, register g; re then im through the same buffer.
    // Rows are written with stride 33 across lanes and read along rows: conflict-free both ways.
    const int lx = opaque(l);
    float *xch = xch_wave + opaque(h) * XCH_FLOATS;
    const int g = __builtin_bitreverse32((uint32_t)lx) >> 27;
#pragma unroll
    for (int r = 0; r < 32; r++) xch[g * XPAD + r] = z[r].x;
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; j++) z[j].x = xch[j * XPAD + lx];
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 32; r++) xch[g * XPAD + r] = z[r].y;
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 32; j++) z[j].y = xch[j * XPAD + lx];
    wave_lds_sync();

    // stages 6..10.  Twiddle loads are issued well ahead of the butterflies that use them: stages 6-8 (7 values)
    // right behind the transpose reads, stage 9 under stage 8's arithmetic, stage 10 in two halves under stage 9
    // and under its own first half.
    const f32x2 *s_tw_lane = reinterpret_cast<const f32x2 *>(smem + LDS_TW) + opaque(l);
    f32x2 w0[1], w1[2], w2[4];
    load_tw<0, 0, 1>(w0, s_tw_lane);
    load_tw<1, 0, 2>(w1, s_tw_lane);
    load_tw<2, 0, 4>(w2, s_tw_lane);
    SCHED_FENCE();
    stage_lane<0, 0, 1>(z, w0);
    stage_lane<1, 0, 2>(z, w1);
    if (!TIGHT) {
        f32x2 w3[8];
        load_tw<3, 0, 8>(w3, s_tw_lane);
        SCHED_FENCE();
        stage_lane<2, 0, 4>(z, w2);
        SCHED_FENCE();
        f32x2 w4a[8];
        load_tw<4, 0, 8>(w4a, s_tw_lane);
        SCHED_FENCE();
        stage_lane<3, 0, 8>(z, w3);
        SCHED_FENCE();
        f32x2 w4b[8];
        load_tw<4, 8, 8>(w4b, s_tw_lane);
        SCHED_FENCE();
        stage_lane<4, 0, 8>(z, w4a);
        SCHED_FENCE();
        stage_lane<4, 8, 8>(z, w4b);
        SCHED_FENCE();
    } else {
        // the averaging kernel also carries 16 accumulator registers: twiddles arrive in groups of four, one group ahead
        f32x2 wa[4], wb[4];
        load_tw<3, 0, 4>(wa, s_tw_lane);
        SCHED_FENCE();
        stage_lane<2, 0, 4>(z, w2);
        SCHED_FENCE();
        load_tw<3, 4, 4>(wb, s_tw_lane);
        SCHED_FENCE();
        stage_lane<3, 0, 4>(z, wa);
        SCHED_FENCE();
        load_tw<4, 0, 4>(wa, s_tw_lane);
        SCHED_FENCE();
        stage_lane<3, 4, 4>(z, wb);
        SCHED_FENCE();
        load_tw<4, 4, 4>(wb, s_tw_lane);
        SCHED_FENCE();
        stage_lane<4, 0, 4>(z, wa);
        SCHED_FENCE();
        load_tw<4, 8, 4>(wa, s_tw_lane);
        SCHED_FENCE();
        stage_lane<4, 4, 4>(z, wb);
        SCHED_FENCE();
        load_tw<4, 12, 4>(wb, s_tw_lane);
        SCHED_FENCE();
        stage_lane<4, 8, 4>(z, wa);
        SCHED_FENCE();
        stage_lane<4, 12, 4>(z, wb);
        SCHED_FENCE();
    }
}

struct WfItem {                 // one (channel pair, averaging group) work item, wave-uniform except ch/ch_ok
    uint32_t ch, l0, l1, grp;
    bool ch_ok, carry_in, complete;
};

SSDR_DEV WfItem wf_item(const SsdrWfArgs &a, uint32_t pair, uint32_t grp, int h)
{
    WfItem it;
    it.grp = grp;
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

SSDR_DEV void load_tables(unsigned char *smem, const float *win, const float2 *tw, const uint32_t *lut)
{
    float *s_win = reinterpret_cast<float *>(smem + LDS_WIN);
    f32x2 *s_tw = reinterpret_cast<f32x2 *>(smem + LDS_TW);
    uint32_t *s_lut = reinterpret_cast<uint32_t *>(smem + LDS_LUT0);
    for (int i = threadIdx.x; i < 513; i += blockDim.x) s_win[i] = win[i];
    for (int i = threadIdx.x; i < SSDR_TW_STAGE_N; i += blockDim.x) s_tw[i] = f32x2{tw[i].x, tw[i].y};
    for (int i = threadIdx.x; i < SSDR_LUT_N; i += blockDim.x) s_lut[i] = lut[i];
    __syncthreads();
}

// AVG == false: averaging N == 1, every line is an output line (no accumulators at all).
// HOP == true: lines overlap by half (hop 512 samples = 23.4 lines/s, the reference's waterfall rate, utils_supersdr.py:597):
// line k of the batch covers half-lines k-1 and k, half-line -1 being the tail the previous call left behind.
template <bool AVG, bool HOP>
__global__ __launch_bounds__(SSDR_WF_BLOCK, SSDR_WF_WAVES_PER_EU) void ssdr_wf_kernel(SsdrWfArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];     // the kernel's only LDS object: address 0
    load_tables(smem, a.win, a.tw_stage, a.lut);

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, l = lane & 31;
    float *xch_wave = reinterpret_cast<float *>(smem + LDS_XCH) + wave * 2 * XCH_FLOATS;      // wave-uniform
    const unsigned char *lut = smem + LDS_LUT0;
    const uint32_t n_pairs = (a.n_ch + 1) >> 1;
    // work items: hop 1024 -- one (group, channel pair) each, group-major; hop 512 -- a run of a.grp_run consecutive groups
    // of a channel pair, pair-major: the wave walks the pair's lines in order
    const uint32_t run = (HOP || SSDR_WF_PAIR_MAJOR) ? a.grp_run : 1u;
    const uint32_t n_runs = (a.n_groups + run - 1) / run;
    const uint32_t n_items = n_pairs * n_runs;
    const uint32_t wave_stride = gridDim.x * WAVES;

#if SSDR_WF_BLOCKED_ITEMS
    const uint32_t per_wave = (n_items + wave_stride - 1) / wave_stride;
    const uint32_t item_begin = (blockIdx.x * WAVES + wave) * per_wave;
    const uint32_t item_end = min(item_begin + per_wave, n_items);
    for (uint32_t item = item_begin; item < item_end; item++) {
#else
    for (uint32_t item = blockIdx.x * WAVES + wave; item < n_items; item += wave_stride) {
#endif
      uint32_t pair, g_begin, g_end;
      if (HOP || SSDR_WF_PAIR_MAJOR) {
          pair = item / n_runs;
          g_begin = (item - pair * n_runs) * run;
          g_end = min(g_begin + run, a.n_groups);
      } else {
          g_begin = item / n_pairs;
          pair = item - g_begin * n_pairs;
          g_end = g_begin + 1;
      }
      const uint32_t n_trips = (HOP || SSDR_WF_PAIR_MAJOR) ? g_end - g_begin : 1u;
      for (uint32_t trip = 0; trip < n_trips; trip++) {
        const uint32_t grp = g_begin + trip;
        uint32_t pair_now = __builtin_amdgcn_readfirstlane(pair);   // everything derived from the pair is recomputed per group
        asm volatile("" : "+s"(pair_now));      // (a few scalar ops) rather than hoisted into VGPRs that live across the whole FFT
        const WfItem it = wf_item(a, pair_now, grp, h);
        const float calq = a.consts[it.ch].wf_cal_lin * SSDR_LUT_SCALE;
        uint32_t acc[AVG ? 16 : 1];
#pragma unroll
        for (int j = 0; j < (AVG ? 16 : 1); j++) acc[j] = 0;
        constexpr uint32_t LINE_STEP = HOP ? SSDR_NFFT / 2 : SSDR_NFFT;
        const uint32_t *src = a.iq + (uint64_t)it.ch * a.ch_stride + (uint64_t)it.l0 * LINE_STEP + l;

        for (uint32_t line = it.l0; line < it.l1; line++, src += LINE_STEP) {
            f32x2 z[32];
            uint32_t raw[32];
#if SSDR_WF_ABLATE == 2      // ablation: no global loads
#pragma unroll
            for (int r = 0; r < 32; r++) raw[r] = (uint32_t)(line * 2654435761u + r * 40503u + lane * 97u) & 0x1FFF1FFFu;
#else
            if (HOP) load_line_halves(line ? src - SSDR_NFFT / 2 : a.tail + (uint64_t)it.ch * (SSDR_NFFT / 2) + l, src, raw);
            else load_line(src, raw);
#endif
#if SSDR_WF_ABLATE == 1      // ablation: memory traffic only
            int16_t *x16 = reinterpret_cast<int16_t *>(xch_wave + opaque(h) * XCH_FLOATS) + opaque(l);
#pragma unroll
            for (int j = 0; j < 32; j++) x16[32 * ((j + 16) & 31)] = (int16_t)(raw[j] & 0xFF);
#else
            if (SSDR_PRIO_WF) prio_compute_phase();
            window_line(raw, smem, l, z);
            SCHED_FENCE();
            fft_line<AVG>(z, smem, xch_wave, h, l);
            if (SSDR_PRIO_WF) prio_latency_phase();          // quantiser look-ups, the store, the next line's loads

            // |X|^2 -> 1-dB byte; bin k = 32 j + l lands at fftshifted position 32 ((j+16)&31) + l
            if (AVG) {
                quantise32(z, calq, lut, [&](int j, uint32_t q01) { acc[j] += q01; });
            } else {
                uint32_t q[16];                                     // bins j | j+16, written out after the last look-up
                quantise32(z, calq, lut, [&](int j, uint32_t q01) { q[j] = q01; });
                int16_t *x16 = reinterpret_cast<int16_t *>(xch_wave + opaque(h) * XCH_FLOATS) + opaque(l);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    x16[32 * (j + 16)] = (int16_t)(q[j] & 0xFFFFu);
                    x16[32 * j] = (int16_t)(q[j] >> 16);
                }
            }
#endif
        }

        float *xch = xch_wave + opaque(h) * XCH_FLOATS;
        if (AVG) {
            int16_t *x16 = reinterpret_cast<int16_t *>(xch) + opaque(l);
#pragma unroll
            for (int j = 0; j < 16; j++) {
                x16[32 * (j + 16)] = (int16_t)(acc[j] & 0xFFFFu);
                x16[32 * j] = (int16_t)(acc[j] >> 16);
            }
        }
        wave_lds_sync();
        const u32x4 *x128 = reinterpret_cast<const u32x4 *>(xch);
        int16_t *dst = it.complete ? a.out + ((uint64_t)it.grp * a.n_ch + it.ch) * SSDR_NFFT
                                   : a.acc_out + (uint64_t)it.ch * SSDR_NFFT;
        const int16_t *cin = a.acc_in + (uint64_t)it.ch * SSDR_NFFT;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            u32x4 v = x128[q * 32 + l];
            if (AVG && it.carry_in)         // wave-uniform; sums stay < 2^15 so a 32-bit add is a packed 2x16 add
                v += reinterpret_cast<const u32x4 *>(cin)[q * 32 + l];
            if (it.ch_ok) SSDR_NT_STORE(v, reinterpret_cast<u32x4 *>(dst) + q * 32 + l);
        }
        wave_lds_sync();
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused superframe kernel for the metric's configuration (N = 1, hop 1024, every channel on the full-band AM path):
// one read of a channel's 4 KB line feeds the FFT and both 512-sample audio frames.
//
// A wave owns a channel pair for the whole call (the audio chain is sequential in time).  Per superframe: the line is
// loaded once in the FFT's layout (lane l of a half: samples 32 r + l); each lane turns its 32 samples into integer
// powers I*I + Q*Q (all the full-band AM chain needs, ssdr_audio.hip) and parks them in the wave's transpose buffer,
// which the FFT does not need yet; then the whole wave runs the audio chain of channel A, then of channel B -- 64 lanes
// x 8 consecutive samples per frame read back from LDS, i.e. exactly the stand-alone kernel's layout and code (bit-identical
// results, the same scan orders); then the FFT proceeds on the raw samples it still holds.  The vector ALU work is the
// sum of the two kernels'; what is saved is the second read of the input.
// HOP (round 4): the waterfall at the reference's line rate (hop 512, utils_supersdr.py:597, 742).  A step is then one half-line:
// the 512 new samples of each channel are its next audio frame, and the line is the previous half (re-read -- the wave fetched
// it one step earlier, and only that read is left to the L2: plain load, everything else streams) followed by the new one.
// AVG (round 4): time binning N > 1 -- the wave owns its channel pair for the whole call, so the N-line sums stay in 16 registers
// per lane across the lines of a group (carry in / out of partial groups as in ssdr_wf_kernel<true, .>).
template <bool HOP, bool AVG = false>
__global__ __launch_bounds__(SSDR_WF_BLOCK, SSDR_WF_WAVES_PER_EU) void ssdr_fused_am_kernel(SsdrFusedArgs fa)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];
    const SsdrWfArgs &a = fa.wf;
    const SsdrAudioArgs &u = fa.au;
    load_tables(smem, a.win, a.tw_stage, a.lut);

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, l = lane & 31;
    float *xch_wave = reinterpret_cast<float *>(smem + LDS_XCH) + wave * 2 * XCH_FLOATS;
    uint32_t *qbuf = reinterpret_cast<uint32_t *>(xch_wave);                 // [2][XCH_FLOATS]: powers of the two channels' line
    const unsigned char *lut = smem + LDS_LUT0;
    const uint32_t n_pairs = (a.n_ch + 1) >> 1;
    const uint32_t wave_stride = gridDim.x * WAVES;
    const uint32_t n_frames = u.n_frames;

    for (uint32_t pair = blockIdx.x * WAVES + wave; pair < n_pairs; pair += wave_stride) {
        const uint32_t ch_raw = 2 * pair + h;
        const bool ch_ok = ch_raw < a.n_ch;
        const uint32_t ch = ch_ok ? ch_raw : a.n_ch - 1;
        const float cal_wf = a.consts[ch].wf_cal_lin * SSDR_LUT_SCALE;
        const uint32_t n_sub = (2 * pair + 1 < a.n_ch) ? 2u : 1u;           // channels of this pair that exist

        // The audio chain's carried state of both channels (wave-uniform: 14 words per channel) does not stay in registers
        // across the FFT, which needs nearly all of them: between two audio phases it rests in the pad column of the wave's
        // transpose buffer (index 33 i + 32, i >= 16: written by neither the transpose nor the line staging).  Only the two per-lane
        // keepers (RSSI sum, flag of frame f mod 64) stay in registers.
        float rssi_sum[2] = {0.0f, 0.0f};
        uint32_t flag_keep[2] = {0u, 0u};
        // (rows 16..31 of the pad column: the line staging reuses the first 2 KB of the buffer)
        auto pad = [&](int c, int i) -> float & { return xch_wave[c * XCH_FLOATS + 33 * (i + 16) + 32]; };
        if (lane < 2) {
            const uint32_t cc = min(2 * pair + (uint32_t)lane, a.n_ch - 1);
            const ssdr_chan_state st = u.state[cc];
            pad(lane, 0) = st.dc;
            pad(lane, 1) = st.agc_d;
#pragma unroll
            for (int i = 0; i < 8; i++) pad(lane, 2 + i) = st.agc_m[i];
            const uint4 t = *reinterpret_cast<const uint4 *>(u.hist + (size_t)cc * SSDR_HIST + SSDR_HIST - 4);
            pad(lane, 10) = __uint_as_float(iq_power(t.x)); pad(lane, 11) = __uint_as_float(iq_power(t.y));
            pad(lane, 12) = __uint_as_float(iq_power(t.z)); pad(lane, 13) = __uint_as_float(iq_power(t.w));
        }
        wave_lds_sync();

        const uint32_t *src = a.iq + (uint64_t)ch * a.ch_stride + l;
        uint32_t last_raw31 = 0;
        uint32_t acc[AVG ? 16 : 1];                            // AVG: sums of the group's lines so far (two bins per register)
#pragma unroll
        for (int j = 0; j < (AVG ? 16 : 1); j++) acc[j] = 0;
        constexpr uint32_t LINE_STEP = HOP ? SSDR_NFFT / 2 : SSDR_NFFT;
        constexpr int FRAMES_PER_LINE = HOP ? 1 : 2;
        for (uint32_t line = 0; line < a.n_lines; line++, src += LINE_STEP) {
            // ---- the carried state out of its resting place (the powers below overwrite it)
            float dc[2], agc_d[2], agc_m[2][8];
            uint32_t tail_q[2][4];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                dc[c] = pad(c, 0); agc_d[c] = pad(c, 1);
#pragma unroll
                for (int i = 0; i < 8; i++) agc_m[c][i] = pad(c, 2 + i);
#pragma unroll
                for (int i = 0; i < 4; i++) tail_q[c][i] = __float_as_uint(pad(c, 10 + i));
            }
            wave_lds_sync();
            // ---- audio, phase 1: the line's raw samples into the (still unused) transpose buffer, in natural order.  Nobody needs them
            // in registers yet, so they come in the widest form: 16 bytes per lane, 1 KB contiguous per instruction, the whole wave on one
            // channel at a time (8 load and 8 LDS store instructions per line pair instead of 32 and 16).  The audio chain reads them back
            // eight consecutive samples per lane, the FFT takes them out of the LDS in its own layout afterwards: one read from HBM
            // (streaming), none from the L2, and no 32 registers held across the audio chain.
#if SSDR_FUSED_WIDE_LOADS
            {
                const uint32_t lw = opaque(lane);
                u32x4 t[2][4];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    uint32_t pair_now = __builtin_amdgcn_readfirstlane(pair);
                    asm volatile("" : "+s"(pair_now));
                    const uint32_t cc = min(2 * pair_now + (uint32_t)c, a.n_ch - 1);
                    const uint32_t *row = a.iq + (uint64_t)cc * a.ch_stride + (uint64_t)line * LINE_STEP;       // the line's newest LINE_STEP samples
                    if (HOP) {
                        const uint32_t *older = line ? row - SSDR_NFFT / 2 : a.tail + (uint64_t)cc * (SSDR_NFFT / 2);
#pragma unroll
                        for (int i = 0; i < 2; i++) t[c][i] = SSDR_NT_LOAD(reinterpret_cast<const u32x4 *>(older) + 64 * i + lw);   // its last use
#pragma unroll
                        for (int i = 0; i < 2; i++) t[c][2 + i] = reinterpret_cast<const u32x4 *>(row)[64 * i + lw];      // read again one line later: L2
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; i++) t[c][i] = SSDR_NT_LOAD(reinterpret_cast<const u32x4 *>(row) + 64 * i + lw);
                    }
                }
                SCHED_FENCE();
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int i = 0; i < 4; i++) reinterpret_cast<u32x4 *>(qbuf + c * XCH_FLOATS)[64 * i + lw] = t[c][i];
                SCHED_FENCE();
            }
#else
            {
                uint32_t *q = qbuf + opaque(h) * XCH_FLOATS + opaque(l);
                uint32_t t[32];                     // all 32 loads in flight, then the stores: left alone the compiler issues
                if (HOP) load_line_halves(line ? src - SSDR_NFFT / 2 : a.tail + (uint64_t)ch * (SSDR_NFFT / 2) + l, src, t);
                else load_line(src, t);             // load, wait, store one sample at a time -- 32 round trips to memory per line
                SCHED_FENCE();
#pragma unroll
                for (int r = 0; r < 32; r++) q[32 * r] = t[r];
                SCHED_FENCE();
            }
#endif
            wave_lds_sync();
            // ---- audio, phase 2: channel A, then channel B, two frames each, all 64 lanes on one channel
            prio_latency_phase();                                             // (the call's first line; later ones arrive with it)
#pragma unroll
            for (int c = 0; c < (SSDR_FUSED_ABLATE == 2 ? 0 : 2); c++) {      // (timing ablation 2: no audio chain)
                if ((uint32_t)c >= n_sub) continue;                           // wave-uniform
                uint32_t pair_now = __builtin_amdgcn_readfirstlane(pair);     // per line: the channel's constants and output rows are
                asm volatile("" : "+s"(pair_now));                              // fetched again (scalar loads) rather than kept across the FFT
                const uint32_t cc = 2 * pair_now + c;
                const ssdr_chan_consts &kc = u.consts[cc];
                const AgcK agc_c = {kc.agc_c0, kc.agc_c1, kc.agc_knee, kc.agc_delta8, kc.hang_frames};
                const float cal_c = kc.smeter_cal_db;
#pragma unroll
                for (int f = 0; f < FRAMES_PER_LINE; f++) {
                    const uint32_t frame = HOP ? line : 2 * line + f;               // hop 512: the new half of the line is the step's frame
                    const u32x4 *qp = reinterpret_cast<const u32x4 *>(qbuf + c * XCH_FLOATS + SSDR_FRAME * (HOP ? 1 : f)) + 2 * opaque(lane);
                    const u32x4 q0 = qp[0], q1 = qp[1];
                    const uint32_t rw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                    uint32_t qv[8], d[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) qv[j] = iq_power(rw[j]);
                    float p[8], aud[8];
#pragma unroll
                    for (int j = 0; j < 4; j++) { d[j] = from_prev_lane_u(tail_q[c][j], qv[4 + j]); d[4 + j] = qv[j]; }
#pragma unroll
                    for (int j = 0; j < 4; j++) tail_q[c][j] = lane63_u(qv[4 + j]);
#pragma unroll
                    for (int j = 0; j < 8; j++) p[j] = (float)d[j];
                    const float pmx = block_peak(p);                          // (also what the AGC takes as the block's peak)
                    const bool trig = wave_any(pmx >= 1073676160.0f) || tail_q[c][0] >= 0x3FFF0001u || tail_q[c][1] >= 0x3FFF0001u ||
                                      tail_q[c][2] >= 0x3FFF0001u || tail_q[c][3] >= 0x3FFF0001u;
                    const bool clip = trig ? wave_any(raw_clipped(rw)) : false;    // the exact check, only then
                    demod_am<true>(p, dc[c], aud);
                    agc_pack_store(p, aud, lane, agc_c, agc_d[c], agc_m[c], u.pcm + ((uint64_t)cc * n_frames + frame) * SSDR_FRAME + 8 * lane, pmx);
                    rssi_flag_step(p, clip, frame, n_frames, lane, cal_c, rssi_sum[c], flag_keep[c],
                                   u.rssi + (uint64_t)cc * n_frames, u.flags + (uint64_t)cc * n_frames);
                }
            }
            wave_lds_sync();
            prio_compute_phase();
            // ---- waterfall: the line out of the LDS (before the carried state takes its resting place in it again)
            uint32_t raw[32];
            {
                const uint32_t *q = qbuf + opaque(h) * XCH_FLOATS + opaque(l);
#pragma unroll
                for (int r = 0; r < 32; r++) raw[r] = q[32 * r];
            }
            wave_lds_sync();
            if (lane < 2) {                                                    // ... and back to rest
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    if (lane != c) continue;
                    pad(c, 0) = dc[c]; pad(c, 1) = agc_d[c];
#pragma unroll
                    for (int i = 0; i < 8; i++) pad(c, 2 + i) = agc_m[c][i];
#pragma unroll
                    for (int i = 0; i < 4; i++) pad(c, 10 + i) = __uint_as_float(tail_q[c][i]);
                }
            }
            wave_lds_sync();
            SCHED_FENCE();
            // ---- ... from here on exactly as ssdr_wf_kernel<false, false>
            // the raw tail of the call's last frame (its samples 384..511 = this line's 896..1023) is the next call's history
            if (line + 1 == a.n_lines && ch_ok) {
#pragma unroll
                for (int r = 28; r < 32; r++) u.hist[(size_t)ch * SSDR_HIST + 32 * (r - 28) + l] = raw[r];
            }
            last_raw31 = raw[31];
            uint32_t qn[16];
#if SSDR_FUSED_ABLATE == 1              // timing ablation: no FFT (the audio phase, the loads and the stores remain)
#pragma unroll
            for (int j = 0; j < 16; j++) qn[j] = raw[j] ^ raw[j + 16];
#else
            f32x2 z[32];
            window_line(raw, smem, l, z);
            SCHED_FENCE();
            fft_line<true>(z, smem, xch_wave, h, l);          // the averaging kernel's twiddle schedule: fewer registers in flight
            prio_latency_phase();                             // quantiser look-ups, the line's store, the next line's loads and audio phase
            if (AVG) quantise32(z, cal_wf, lut, [&](int j, uint32_t q01) { acc[j] += q01; });
            else quantise32(z, cal_wf, lut, [&](int j, uint32_t q01) { qn[j] = q01; });
#endif
            // AVG: line `line` is line (phase + line) of the stream of groups; a group leaves when its N-th line is in, the last
            // (partial) one of the call goes to acc_out
            const uint32_t pos = a.phase + line;
            const bool group_done = !AVG || (pos + 1) % a.n_avg == 0;
            const bool last_line = line + 1 == a.n_lines;
            if (group_done || last_line) {
                float *xch = xch_wave + opaque(h) * XCH_FLOATS;
                int16_t *x16 = reinterpret_cast<int16_t *>(xch) + opaque(l);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const uint32_t v = AVG ? acc[j] : qn[j];
                    x16[32 * (j + 16)] = (int16_t)(v & 0xFFFFu);
                    x16[32 * j] = (int16_t)(v >> 16);
                }
                wave_lds_sync();
                const u32x4 *x128 = reinterpret_cast<const u32x4 *>(xch);
                const uint32_t grp = AVG ? pos / a.n_avg : line;
                int16_t *dst = group_done ? a.out + ((uint64_t)grp * a.n_ch + ch) * SSDR_NFFT : a.acc_out + (uint64_t)ch * SSDR_NFFT;
                const bool carry_in = AVG && grp == 0 && a.phase != 0;           // wave-uniform
                const int16_t *cin = a.acc_in + (uint64_t)ch * SSDR_NFFT;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u32x4 v = x128[q * 32 + l];
                    if (carry_in) v += reinterpret_cast<const u32x4 *>(cin)[q * 32 + l];      // sums stay < 2^15: a packed 2 x 16 add
                    if (ch_ok) SSDR_NT_STORE(v, reinterpret_cast<u32x4 *>(dst) + q * 32 + l);
                }
                wave_lds_sync();
                if (AVG) {
#pragma unroll
                    for (int j = 0; j < 16; j++) acc[j] = 0;
                }
            }
        }

        // ---- state back to HBM
        if (a.n_lines) {
            // the discriminator memory an AM channel leaves behind: y[511] = z1[507] of the last frame, mixed as the twin does
            // (block 63 of the frame, element 3).  Sample 507 of that frame is the line's sample 1019 = raw[31] of lane 27.
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if ((uint32_t)c >= n_sub) continue;
                const uint32_t cc = 2 * pair + c;
                const ssdr_chan_consts &kc = u.consts[cc];
                ssdr_chan_state st = u.state[cc];
                const uint32_t phi_last = st.phi1 + (uint32_t)(SSDR_FRAME * (n_frames - 1)) * kc.dphi1;
                float fc, fs, qc, qs, bc, bs, cs, ss;
                ssdr_phasor32(phi_last, fc, fs);
                ssdr_phasor32((uint32_t)(8 * 63) * kc.dphi1, qc, qs);
                ssdr_phasor32(kc.dphi1, cs, ss);
                phasor_mul(fc, fs, qc, qs, bc, bs);
#pragma unroll
                for (int j = 0; j < 3; j++) { const float cn = fmaf(bc, cs, -(bs * ss)), sn = fmaf(bs, cs, bc * ss); bc = cn; bs = sn; }
                const float xr = (float)(int16_t)(last_raw31 & 0xFFFFu), xi = (float)((int32_t)last_raw31 >> 16);
                const float zr = fmaf(xr, bc, xi * bs) + 0.0f, zi = fmaf(xi, bc, -(xr * bs)) + 0.0f;
                st.prev_re = lane_f(zr, 32 * c + 27);
                st.prev_im = lane_f(zi, 32 * c + 27);
                st.phi1 += (uint32_t)(SSDR_FRAME * n_frames) * kc.dphi1;
                st.phi2 += (uint32_t)(SSDR_FRAME * n_frames) * kc.dphi2;
                st.dc = pad(c, 0); st.agc_d = pad(c, 1);
#pragma unroll
                for (int i = 0; i < 8; i++) st.agc_m[i] = pad(c, 2 + i);
                if (lane == 0) u.state[cc] = st;
            }
        }
    }
}

// exhaustive quantiser self-test: every positive finite float p, scaled and clamped as the kernel does it, against a
// binary search over T[] with the unscaled p
__global__ __launch_bounds__(256) void ssdr_quant_selftest_kernel(const float *thr_g, const uint32_t *lut_g,
                                                                    unsigned long long *mismatch)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_LUT_END + 16 + 1024];
    float *s_thr = reinterpret_cast<float *>(smem + ((LDS_LUT_END + 15) & ~15));
    uint32_t *s_lut = reinterpret_cast<uint32_t *>(smem + LDS_LUT0);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_thr[i] = thr_g[i];
    for (int i = threadIdx.x; i < SSDR_LUT_N; i += blockDim.x) s_lut[i] = lut_g[i];
    __syncthreads();
    unsigned long long bad = 0;
    const unsigned char *lut = smem + LDS_LUT0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < 0x7F800000ull; u += stride) {
        const float p = __uint_as_float((uint32_t)u);
        int lo = 0, hi = 255;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_thr[mid] <= p) lo = mid; else hi = mid - 1;
        }
        float pc;
        asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(pc) : "v"(p), "v"(SSDR_LUT_SCALE));
        bad += (quantise(pc, lut) != (uint32_t)lo);
    }
    if (bad) atomicAdd(mismatch, bad);
}

} // namespace

hipError_t ssdr_launch_wf(const SsdrWfArgs &a, uint32_t grid, hipStream_t stream)
{
    const dim3 g(grid), b(SSDR_WF_BLOCK);
    if (a.tail) {
        if (a.n_avg > 1) hipLaunchKernelGGL((ssdr_wf_kernel<true, true>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((ssdr_wf_kernel<false, true>), g, b, 0, stream, a);
    } else {
        if (a.n_avg > 1) hipLaunchKernelGGL((ssdr_wf_kernel<true, false>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((ssdr_wf_kernel<false, false>), g, b, 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t ssdr_launch_fused_am(const SsdrFusedArgs &a, uint32_t grid, hipStream_t stream)
{
    if (a.wf.n_avg > 1) {
        if (a.wf.tail) hipLaunchKernelGGL((ssdr_fused_am_kernel<true, true>), dim3(grid), dim3(SSDR_WF_BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((ssdr_fused_am_kernel<false, true>), dim3(grid), dim3(SSDR_WF_BLOCK), 0, stream, a);
    } else {
        if (a.wf.tail) hipLaunchKernelGGL((ssdr_fused_am_kernel<true, false>), dim3(grid), dim3(SSDR_WF_BLOCK), 0, stream, a);
        else hipLaunchKernelGGL((ssdr_fused_am_kernel<false, false>), dim3(grid), dim3(SSDR_WF_BLOCK), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t ssdr_fused_blocks_per_cu(int *blocks)
{
    int b[4] = {0, 0, 0, 0};
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b[0], ssdr_fused_am_kernel<false, false>, SSDR_WF_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b[1], ssdr_fused_am_kernel<true, false>, SSDR_WF_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b[2], ssdr_fused_am_kernel<false, true>, SSDR_WF_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b[3], ssdr_fused_am_kernel<true, true>, SSDR_WF_BLOCK, 0);
    *blocks = b[0];
    for (int i = 1; i < 4; i++) *blocks = b[i] < *blocks ? b[i] : *blocks;
    return e;
}

// workgroups of the waterfall kernel that are resident per CU (min over both instances)
hipError_t ssdr_wf_blocks_per_cu(int *blocks)
{
    int b[4] = {0, 0, 0, 0};
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b[0], ssdr_wf_kernel<false, false>, SSDR_WF_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b[1], ssdr_wf_kernel<true, false>, SSDR_WF_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b[2], ssdr_wf_kernel<false, true>, SSDR_WF_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b[3], ssdr_wf_kernel<true, true>, SSDR_WF_BLOCK, 0);
    if (e != hipSuccess) return e;
    *blocks = b[0];
    for (int i = 1; i < 4; i++) *blocks = b[i] < *blocks ? b[i] : *blocks;
    return hipSuccess;
}

hipError_t ssdr_launch_quant_selftest(const float *thr, const uint32_t *lut, unsigned long long *mismatch, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_quant_selftest_kernel, dim3(2048), dim3(256), 0, stream, thr, lut, mismatch);
    return hipGetLastError();
}
