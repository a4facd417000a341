// EXPERIMENT, NOT BUILT (profiles/r04_ab_one_channel_per_wave.txt: 15 % slower than the shipped kernel; apply ssdr_wf64_wiring.patch and copy this file to
// supersdr_amd/csrc/ to rebuild it).  Stage 5 is in the general form here, so its bins are not the twin's bit for bit.
// ssdr_wf64.hip -- the fused superframe kernel of the metric's configuration with ONE CHANNEL PER WAVE (round 4, last):
//   ssdr_fused_am64_kernel = ssdr_wf.hip:ssdr_fused_am_kernel<false, false> in the float64 kernel's layout (ssdr_wf_exact.hip).
//
// Why: the two-channels-per-wave kernel (32 lanes x 32 points per FFT) needs 64 registers for a line and 8.4 KB of LDS per wave: 128
// VGPRs, 4 waves per SIMD -- and its rate is proportional to the waves per SIMD (3 -> 4 waves: +35 %, profiles/r04_ab_occupancy.txt).
// With 64 lanes x 16 points a line is 32 registers and 4.1 KB: 6 waves per SIMD.
//
// The butterflies are those of the textbook radix-2 DIT FFT the fp32 twin states (same operand pairs, same twiddle table values, the
// same exact / 6-FMA forms per stage), regrouped:
//   * lane L loads samples 64 q + L (q = 0..15): 256 contiguous bytes per load instruction; register brev4(q) of lane L is element
//     e = 16 brev6(L) + brev4(q) of the bit-reversed array;
//   * stage 1 with the window, stages 2-4 in registers (compile-time twiddles), ONE transpose through LDS (re, then im, through the same
//     4.1 KB: row stride 65, conflict-free both ways), stages 5-8 in registers with per-lane twiddles from LDS tables;
//   * stages 9 and 10 pair lanes 16 and 32 apart: v_permlane16_swap / v_permlane32_swap exchange half of the registers so that every
//     lane holds both operands of eight butterflies;
//   * power, table quantiser (ssdr_wf.hip:quantise), the int16 line staged through the transpose buffer: 2 x 16 contiguous bytes per lane.
// The audio chain is the stand-alone AM kernel's code on the raw line parked in the transpose buffer (as in ssdr_fused_exact_am_kernel).
#include "ssdr_math.h"
#include "ssdr_kernels.h"
#include "ssdr_audio_dev.h"
#include "ssdr_consts.h"
#include <cstdio>
#include <cstdlib>

#ifndef SSDR_W64_BLOCK
#define SSDR_W64_BLOCK 512
#endif
#ifndef SSDR_W64_WAVES_PER_EU
#define SSDR_W64_WAVES_PER_EU 6
#endif
#ifndef SSDR_W64_ABLATE
#define SSDR_W64_ABLATE 0                    // timing ablations (wrong results): 1 no lane swaps, 2 no stages 9 / 10, 3 no audio chain, 4 no FFT, 5 no twiddle reads
#endif

namespace {

#define YFENCE() __builtin_amdgcn_sched_barrier(0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVES = SSDR_W64_BLOCK / 64;
constexpr int XROW = 65;                                     // row stride of the transpose buffer, floats
constexpr int XCH_BYTES = 16 * XROW * 4;                     // 4160 per wave (>= the 4096 B of a parked line)
constexpr int LDS_WIN = 0;                                   // 513 floats
constexpr int LDS_LUT0 = 2064;
constexpr int LDS_LUT_END = LDS_LUT0 + SSDR_LUT_N * 4;
constexpr int LDS_TW = (LDS_LUT_END + 15) & ~15;
constexpr int LDS_XCH = LDS_TW + SSDR_TW64F_N * 8;
constexpr int LDS_TOTAL = LDS_XCH + WAVES * XCH_BYTES;
static_assert(LDS_XCH % 16 == 0 && XCH_BYTES % 16 == 0, "alignment");
static_assert(LDS_TOTAL * (SSDR_W64_WAVES_PER_EU * 4 / WAVES) <= 163840, "LDS budget");
// table offsets in entries (ssdr_make_tw64f)
constexpr int T5 = 0, T6 = 16, T7 = 48, T8 = 112, T9 = 240, T10 = 496, T10B = 752;
static_assert(T10B + 256 == SSDR_TW64F_N, "table size");

SSDR_DEV int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
SSDR_DEV void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the butterflies of ssdr_wf.hip (6-FMA general form; exact forms for w = 1 and w = -j in stages 1..4)
SSDR_DEV void bfly(f32x2 &u, f32x2 &v, float wr, float wi)
{
    const float sr = fmaf(-wi, v.y, u.x), si = fmaf(wi, v.x, u.y);
    const float ar = fmaf(wr, v.x, sr), ai = fmaf(wr, v.y, si);
    const float br = fmaf(2.0f, u.x, -ar), bi = fmaf(2.0f, u.y, -ai);
    u = f32x2{ar, ai};
    v = f32x2{br, bi};
}
SSDR_DEV void bfly_1(f32x2 &u, f32x2 &v)
{
    const f32x2 t = v, x = u;
    u = x + t; v = x - t;
}
SSDR_DEV void bfly_mj(f32x2 &u, f32x2 &v)
{
    f32x2 a, b;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(a) : "v"(u), "v"(v));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(b) : "v"(u), "v"(v));
    u = a; v = b;
}

__device__ constexpr int brev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }

// stages 2..4 on the lane's 16 registers, twiddle W_1024[k (1024 >> S)] = W32[k (32 >> S)]
template <int S>
SSDR_DEV void stage_const(f32x2 (&z)[16])
{
    constexpr float W32R[16] = SSDR_W32R_INIT;
    constexpr float W32I[16] = SSDR_W32I_INIT;
    constexpr int half = 1 << (S - 1);
#pragma unroll
    for (int k = 0; k < half; k++) {
        const int mi = k * (32 >> S);
#pragma unroll
        for (int blk = 0; blk < 16; blk += 2 * half) {
            const int i = blk + k, j = i + half;
            if (mi == 0) bfly_1(z[i], z[j]);
            else if (mi == 8) bfly_mj(z[i], z[j]);
            else bfly(z[i], z[j], W32R[mi], W32I[mi]);
        }
    }
}

// stages 5..8 (T = s - 5) on x[rho], rho = bits 7..4 of the element index; the lane's twiddles w[q] = W_(2^s)^(lo4 + 16 q)
template <int T>
SSDR_DEV void stage_lane(f32x2 (&z)[16], const f32x2 (&w)[1 << T])
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

SSDR_DEV void swap16(f32x2 &a, f32x2 &b)                      // rows of 16 lanes: a's odd rows <-> b's even rows
{
    asm("v_permlane16_swap_b32 %0, %1" : "+v"(a.x), "+v"(b.x));
    asm("v_permlane16_swap_b32 %0, %1" : "+v"(a.y), "+v"(b.y));
}
SSDR_DEV void swap32(f32x2 &a, f32x2 &b)                      // a's lanes 32..63 <-> b's lanes 0..31
{
    asm("v_permlane32_swap_b32 %0, %1" : "+v"(a.x), "+v"(b.x));
    asm("v_permlane32_swap_b32 %0, %1" : "+v"(a.y), "+v"(b.y));
}

// the quantiser of ssdr_wf.hip
SSDR_DEV float quant_scaled_power(f32x2 z, float calq)
{
    float p = fmaf(z.x, z.x, z.y * z.y), pc;
    asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(pc) : "v"(p), "v"(calq));
    return pc;
}
SSDR_DEV uint32_t quant_addr(float pc) { return (__float_as_uint(pc) >> (SSDR_LUT_SHIFT - 2)) & ~3u; }
SSDR_DEV uint32_t quant_pair(uint32_t w0, uint32_t w1) { return __builtin_amdgcn_perm(w1, w0, 0x0C070C03u); }

// One line: raw[q] = sample 64 q + lane (I | Q << 16) -> q01[4 t + mm] = byte of bin k (c = 0) | byte of bin k + 512 (c = 1) << 16,
// k = 256 t + 128 b4 + 64 b5 + 16 mm + lo4 for this lane (b4, b5, lo4 = lane bits 4, 5, 0..3).
SSDR_DEV void line_bytes64(const uint32_t (&raw)[16], float calq, const unsigned char *smem, unsigned char *xch, const unsigned char *lut,
                           int lane, uint32_t (&q01)[8])
{
    f32x2 z[16];
    {
        // window with stage 1 folded in: sample n = 64 q + L pairs with n + 512; w[n + 512] = w[512 - n] (symmetric table of 513);
        // t = x w, a = fma(x', w', t), b = fma(-x', w', t)
        const int ll = opaque(lane);
        const float *win_up = reinterpret_cast<const float *>(smem + LDS_WIN) + ll;
        const float *win_dn = reinterpret_cast<const float *>(smem + LDS_WIN) + 512 - ll;
        float wu[8], wd[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { wu[q] = win_up[64 * q]; wd[q] = win_dn[-64 * q]; }
        YFENCE();
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float xr = (float)(int16_t)(raw[q] & 0xFFFFu), xi = (float)((int32_t)raw[q] >> 16);
            const float yr = (float)(int16_t)(raw[q + 8] & 0xFFFFu), yi = (float)((int32_t)raw[q + 8] >> 16);
            const float tr = xr * wu[q], ti = xi * wu[q];
            z[brev4(q)] = f32x2{fmaf(yr, wd[q], tr), fmaf(yi, wd[q], ti)};
            z[brev4(q) + 1] = f32x2{fmaf(-yr, wd[q], tr), fmaf(-yi, wd[q], ti)};
        }
    }
    YFENCE();
    stage_const<2>(z);
    stage_const<3>(z);
    stage_const<4>(z);
    YFENCE();
    // transpose.  Register r of lane L is element 16 G + r, G = brev6(L) = hi2 << 4 | rho.  It goes to lane L' = hi2 << 4 | lo4
    // (lo4 = r), register rho.  Slot of (rho, hi2, lo4): rho * 65 + hi2 * 16 + lo4 -- 64 different banks per instruction both ways.
    {
        const int lx = opaque(lane);
        const int G = (int)(__builtin_bitreverse32((uint32_t)lx) >> 26);
        float *wbase = reinterpret_cast<float *>(xch) + (G & 15) * XROW + (G >> 4) * 16;
        const float *rbase = reinterpret_cast<const float *>(xch) + lx;          // hi2 * 16 + lo4 == lane
#pragma unroll
        for (int r = 0; r < 16; r++) wbase[r] = z[r].x;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 16; j++) z[j].x = rbase[j * XROW];
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 16; r++) wbase[r] = z[r].y;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 16; j++) z[j].y = rbase[j * XROW];
        wave_lds_sync();
    }
    // stages 5..8: per-lane twiddles T_s[q][lo4]
    {
        const f32x2 *twl = reinterpret_cast<const f32x2 *>(smem + LDS_TW) + (opaque(lane) & 15);
        f32x2 w5[1], w6[2], w7[4], w8[8];
        w5[0] = twl[T5];
#pragma unroll
        for (int q = 0; q < 2; q++) w6[q] = twl[T6 + 16 * q];
#pragma unroll
        for (int q = 0; q < 4; q++) w7[q] = twl[T7 + 16 * q];
        YFENCE();
        stage_lane<0>(z, w5);
        stage_lane<1>(z, w6);
        YFENCE();
#pragma unroll
        for (int q = 0; q < 8; q++) w8[q] = twl[T8 + 16 * q];
        YFENCE();
        stage_lane<2>(z, w7);
        YFENCE();
        stage_lane<3>(z, w8);
        YFENCE();
    }
    if (SSDR_W64_ABLATE != 2) {
    // stage 9: pairs element bit 8 = lane bit 4.  After the swap lanes with bit 4 clear hold u, v of elements rho = m (registers m,
    // m + 8), the others of rho = m + 8; twiddle W_512^(16 rho + lo4)
    {
        const int lx = opaque(lane);
        const f32x2 *t9 = reinterpret_cast<const f32x2 *>(smem + LDS_TW) + T9 + (lx & 31);       // [m][b4][lo4]
        f32x2 w9[8];
#pragma unroll
        for (int m = 0; m < 8; m++) w9[m] = t9[32 * m];
#pragma unroll
        for (int m = 0; m < ((SSDR_W64_ABLATE == 1) ? 0 : 8); m++) swap16(z[m], z[m + 8]);
        YFENCE();
#pragma unroll
        for (int m = 0; m < 8; m++) bfly(z[m], z[m + 8], w9[m].x, w9[m].y);
        YFENCE();
    }
    // stage 10: pairs bit 9 = lane bit 5; register pairs (8 t + mm, 8 t + 4 + mm).  Afterwards the lane holds elements
    // rho = mm + 4 b5 + 8 b4 of both t; twiddle W_1024^(256 t + 16 rho + lo4)
    {
        const int lx = opaque(lane);
        const int e = ((lx >> 4) & 1) * 32 + (lx >> 5) * 16 + (lx & 15);                           // [mm][b4][b5][lo4]
        const f32x2 *t10 = reinterpret_cast<const f32x2 *>(smem + LDS_TW) + T10 + e;
        f32x2 w10[8];
#pragma unroll
        for (int mm = 0; mm < 4; mm++) { w10[mm] = t10[64 * mm]; w10[4 + mm] = t10[(T10B - T10) + 64 * mm]; }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int mm = 0; mm < ((SSDR_W64_ABLATE == 1) ? 0 : 4); mm++) swap32(z[8 * t + mm], z[8 * t + 4 + mm]);
        YFENCE();
#pragma unroll
        for (int mm = 0; mm < 4; mm++) {
            bfly(z[mm], z[4 + mm], w10[mm].x, w10[mm].y);
            bfly(z[8 + mm], z[12 + mm], w10[4 + mm].x, w10[4 + mm].y);
        }
        YFENCE();
    }
    }
    // power, threshold count.  Register 8 t + 4 c + mm holds FFT bin k = 512 c + 256 t + 128 b4 + 64 b5 + 16 mm + lo4
    {
        float pc[16];
        uint32_t e[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            pc[r] = quant_scaled_power(z[r], calq);
            e[r] = *reinterpret_cast<const uint32_t *>(lut + quant_addr(pc[r]));
        }
        YFENCE();
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int mm = 0; mm < 4; mm++) {
                const int r0 = 8 * t + mm, r1 = r0 + 4;
                q01[4 * t + mm] = quant_pair(__float_as_uint(pc[r0]) + e[r0], __float_as_uint(pc[r1]) + e[r1]);
            }
    }
}

__global__ __launch_bounds__(SSDR_W64_BLOCK, SSDR_W64_WAVES_PER_EU) void ssdr_fused_am64_kernel(SsdrFusedArgs fa, const float2 *tw_g)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];
    const SsdrWfArgs &a = fa.wf;
    const SsdrAudioArgs &u = fa.au;
    {
        float *s_win = reinterpret_cast<float *>(smem + LDS_WIN);
        uint32_t *s_lut = reinterpret_cast<uint32_t *>(smem + LDS_LUT0);
        f32x2 *s_tw = reinterpret_cast<f32x2 *>(smem + LDS_TW);
        for (int i = threadIdx.x; i < 513; i += blockDim.x) s_win[i] = a.win[i];
        for (int i = threadIdx.x; i < SSDR_LUT_N; i += blockDim.x) s_lut[i] = a.lut[i];
        for (int i = threadIdx.x; i < SSDR_TW64F_N; i += blockDim.x) s_tw[i] = f32x2{tw_g[i].x, tw_g[i].y};
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
        const float calq = kc.wf_cal_lin * SSDR_LUT_SCALE;
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
        __builtin_amdgcn_s_setprio(3);
        for (uint32_t line = 0; line < a.n_lines; line++, src += SSDR_NFFT) {
#pragma unroll
            for (int q = 0; q < 16; q++) raw[q] = __builtin_nontemporal_load(src + 64 * q);
            YFENCE();
            {
                uint32_t *qw = q32 + opaque(lane);
#pragma unroll
                for (int q = 0; q < 16; q++) qw[64 * q] = raw[q];
            }
            wave_lds_sync();
            // ---- audio: two frames, lane l on samples 8 l .. 8 l + 7 of each
#pragma unroll
            for (int f = 0; f < ((SSDR_W64_ABLATE == 3) ? 0 : 2); f++) {
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
            __builtin_amdgcn_s_setprio(0);
            uint32_t q01[8];
            if (SSDR_W64_ABLATE == 4) {
#pragma unroll
                for (int j = 0; j < 8; j++) q01[j] = raw[j] ^ raw[j + 8];
            } else
            line_bytes64(raw, calq, smem, xch, lut, lane, q01);
            __builtin_amdgcn_s_setprio(3);
            {
                const int lx = opaque(lane);
                int16_t *x16 = reinterpret_cast<int16_t *>(xch) + ((lx >> 4) & 1) * 128 + (lx >> 5) * 64 + (lx & 15);
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int mm = 0; mm < 4; mm++) {
                        const uint32_t v = q01[4 * t + mm];
                        x16[512 + 256 * t + 16 * mm] = (int16_t)(v & 0xFFFFu);       // c = 0: bin k < 512 -> upper half of the line (fftshift)
                        x16[256 * t + 16 * mm] = (int16_t)(v >> 16);                // c = 1 -> lower half
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

// the one-channel-per-wave fused kernel: persistent grid
hipError_t ssdr_launch_fused_am64(const SsdrFusedArgs &a, const float2 *tw, hipStream_t stream)
{
    if (a.wf.n_ch == 0 || a.wf.n_lines == 0) return hipSuccess;
    static uint32_t resident = 0;
    if (!resident) {
        int dev = 0, b = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, ssdr_fused_am64_kernel, SSDR_W64_BLOCK, 0)) != hipSuccess) return e;
        resident = (uint32_t)prop.multiProcessorCount * (uint32_t)(b < 1 ? 1 : b);
        if (getenv("SSDR_DEBUG_OCC")) fprintf(stderr, "ssdr_fused_am64_kernel: %d workgroups of %d per CU\n", b, SSDR_W64_BLOCK);
    }
    constexpr uint32_t waves = SSDR_W64_BLOCK / 64;
    const uint64_t need = ((uint64_t)a.wf.n_ch + waves - 1) / waves;
    hipLaunchKernelGGL(ssdr_fused_am64_kernel, dim3((uint32_t)(need < resident ? need : resident)), dim3(SSDR_W64_BLOCK), 0, stream, a, tw);
    return hipGetLastError();
}
