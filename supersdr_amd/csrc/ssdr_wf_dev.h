// ssdr_wf_dev.h -- device pieces of the waterfall stage shared by the kernels that contain a 1024-point line FFT
// (ssdr_wf.hip: ssdr_wf_kernel, ssdr_fused_am_kernel; ssdr_chain_ws.hip: the FFT waves of ssdr_chain_ws_kernel): the LDS map of the tables,
// the 6-FMA butterflies and register stages, the one-transpose FFT of a 32-lane half, the window folded into stage 1, the
// threshold-count dB quantiser.  "wave64 == two FFTs, 32 points per lane"; see ssdr_wf.hip for the mapping.
// The including file defines WAVES (waves per workgroup) and LDS_TOTAL after this header.
#pragma once
#include "ssdr_math.h"
#include "ssdr_kernels.h"

namespace {

constexpr int XPAD = 33;                       // row stride (floats) of the transpose buffer
constexpr int XCH_FLOATS = 32 * XPAD;          // per FFT: 4224 B
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
static_assert(LDS_XCH % 16 == 0 && LDS_TW % 8 == 0, "alignment");

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

} // namespace
