// ssdr_fused_gen.hip -- both stages on ONE read of the input for ANY mix of audio frame paths (round 5):
//   the general-mode counterpart of ssdr_wf.hip:ssdr_fused_am_kernel.  configs[3] of BASELINE.json (AM / USB / LSB / NBFM by channel,
//   10x time binning) ran its audio kernels and the waterfall kernel over the same 4 KB line twice (10.4 KB of HBM traffic per
//   channel-superframe against SURVEY.md 8d's fused budget of 6.35 KB); here a wave reads each line once.
//
// Stands where the reference receives W/F lines and SND frames of the same receiver from its server (utils_supersdr.py:780-785,
// 1044-1076); tap formula of the channel filter: utils_supersdr.py:334-344 (ssdr_tables.cpp).
//
// Shape: the fused AM kernel's -- a wave owns a channel PAIR for the whole call (the audio chain is sequential in time) and walks its
// lines; per line the audio chain of one channel, then of the other (all 64 lanes on one channel: exactly the stand-alone kernels'
// layout, code and scan orders -- ssdr_audio_dev.h -- so the results are theirs bit for bit), then both channels' FFTs side by side in
// the two half-waves (ssdr_wf_dev.h).  What is new is where everything rests while the other phase has the registers and the LDS:
//
//   * LDS per wave (9456 B, 16 waves = ONE 1024-thread workgroup per CU, so the tables are held once per CU):
//       work area 8720 B = [ R0: slot 0's raw line 4096 | ................................. ]   R1 = [4624, 8720): slot 1's raw line
//                          [ ........ S_lo: [0, 4624) ....|..... S_hi: [4096, 8720) ........ ]   S  = the FIR's work area of a general-path
//       channel: 4 history octets + the frame's 64 octets of mixed samples (float2), 64 B per octet + 16 B of padding after every fourth
//       (conflict-free ds_read_b128 at a 64-byte lane stride; the stand-alone kernel's 80-byte stride does not fit).  S overlaps the OTHER
//       slot's raw line, so a pair with one general channel runs that channel first, before the other line is fetched; a pair of two
//       general channels runs slot 1 from per-frame loads and files its line afterwards (a second read, served by the L2).
//       The FFT's transposes take the first 8448 B of the work area once both lines are in registers.
//       Behind the work area: 2 x 256 B filter history (the last 32 mixed samples of each general channel's previous line) and
//       2 x 28 words of carried state (phases, DC, AGC follower, discriminator memory, the shift paths' 4-sample tails).
//   * registers: nothing of the audio chain lives across the FFT, and nothing of the FFT across the audio chain.  The per-lane NCO constants
//     P(8 l dphi) and the line's two frame phasors are re-evaluated per line (two polynomials per oscillator: the same arguments, hence
//     the same bits, as the stand-alone kernel's per-call / per-64-frames tables); the carried state comes back from the LDS through
//     v_readfirstlane (wave-uniform, so it sits in scalar registers as in the stand-alone kernels); the per-frame RSSI sums and flags go
//     straight to their output rows and are converted to dBm once per call; with N > 1 the N-line sums rest in 4 KB of global memory per
//     wave while the audio chain runs; the kernel's arguments are re-read from the kernarg segment per phase.
//   * MEASURED (profiles/r05_ab_fused_general.txt): bit-identical to the two kernels and 45 % slower than running them side by side --
//     the register and LDS squeeze costs +21 % instructions, 31-36 spilled registers and more memory traffic than the second read saves.
//     Opt-in: ssdr_set_fused(ctx, 3).
//   * channel filters of up to 33 taps (4 history octets: every passband of the reference's mode table except CW; a ctx with a longer
//     filter runs the two kernels); SSDR_MODE_IQ channels (a second output row) likewise.
#include "ssdr_math.h"
#include "ssdr_kernels.h"
#include "ssdr_audio_dev.h"
#include "ssdr_wf_dev.h"

#ifndef SSDR_GEN_ABLATE
#define SSDR_GEN_ABLATE 0                    // timing ablations only (1: no FFT, 2: no audio chain)
#endif

namespace {

constexpr int WAVES = SSDR_GEN_BLOCK / 64;
constexpr int HMAX = SSDR_GEN_HIST_OCT;                          // history octets of a general-path channel
constexpr int SOCT = 64 + HMAX;
constexpr int S_BYTES = SOCT * 64 + ((SOCT + 3) / 4) * 16;       // 4624
constexpr int R_BYTES = SSDR_NFFT * 4;                           // a raw line
constexpr int WORK_BYTES = R_BYTES + S_BYTES;                    // 8720
constexpr int R1_OFF = S_BYTES;                                  // slot 1's raw line: the top 4096 B of the work area
constexpr int HIST_BYTES = HMAX * 64;
constexpr int STATE_WORDS = 28;
constexpr int WAVE_BYTES = WORK_BYTES + 2 * HIST_BYTES + 2 * STATE_WORDS * 4;
constexpr int LDS_TOTAL = LDS_XCH + WAVES * WAVE_BYTES;
static_assert(WORK_BYTES >= 2 * XCH_FLOATS * 4, "the FFT's transposes live in the work area");
static_assert(WAVE_BYTES % 16 == 0 && R1_OFF % 16 == 0, "alignment");
static_assert(LDS_TOTAL <= 163840, "LDS budget");
static_assert(HMAX * 8 >= SSDR_GEN_NTAP_MAX - 1, "history covers the longest filter");

enum { PATH_GENERAL = SSDR_PATH_GENERAL, PATH_DELAY4 = SSDR_PATH_DELAY4, PATH_AM_RAW = SSDR_PATH_AM_RAW };

// the FIR's work area: octet q (0 .. SOCT-1; q < HMAX is history) at 64 q + 16 (q >> 2)
SSDR_DEV int s_addr(int q) { return (q << 6) + ((q >> 2) << 4); }
template <typename T> SSDR_DEV T *al16(const void *p) { return reinterpret_cast<T *>(__builtin_assume_aligned(const_cast<void *>(p), 16)); }
// ... and the 16 bytes of padding behind octets 4 j .. 4 j + 3 hold the channel's taps 4 j .. 4 j + 3 (staged per line: audio_line)
SSDR_DEV int s_tap_addr(int j) { return 272 * j + 256; }
SSDR_DEV void s_load_oct(const unsigned char *S, int q, float2 (&v)[8])
{
    const float4 *p = al16<const float4>(S + s_addr(q));
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float4 t = p[i];
        v[2 * i] = make_float2(t.x, t.y);
        v[2 * i + 1] = make_float2(t.z, t.w);
    }
}
SSDR_DEV void s_store_oct(unsigned char *S, int q, const float2 (&v)[8])
{
    float4 *p = al16<float4>(S + s_addr(q));
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = make_float4(v[2 * i].x, v[2 * i].y, v[2 * i + 1].x, v[2 * i + 1].y);
}

SSDR_DEV float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
SSDR_DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// carried state of one slot: wave-uniform, in scalar registers while its audio phase runs, in the LDS otherwise
struct Slot {
    uint32_t phi1, phi2;
    float dc, agc_d, agc_m[8], prev_re, prev_im;
    float cs1, ss1, cs2, ss2;                   // S = P(dphi) of the two oscillators
    float2 tail_z[4];                           // PATH_DELAY4: mixed samples -4 .. -1
    uint32_t tail_q[4];                         // PATH_AM_RAW: I*I + Q*Q of samples -4 .. -1
};
// words: 0 phi1, 1 phi2, 2 dc, 3 agc_d, 4..11 agc_m, 12 prev_re, 13 prev_im, 14..17 cs1 ss1 cs2 ss2, 18..25 tail (tail_q in 18..21)
SSDR_DEV void slot_load(const float *sp, Slot &s, int path)
{
    s.phi1 = uni(__float_as_uint(sp[0])); s.phi2 = uni(__float_as_uint(sp[1]));
    s.dc = uni(sp[2]); s.agc_d = uni(sp[3]);
#pragma unroll
    for (int i = 0; i < 8; i++) s.agc_m[i] = uni(sp[4 + i]);
    s.prev_re = uni(sp[12]); s.prev_im = uni(sp[13]);
    s.cs1 = uni(sp[14]); s.ss1 = uni(sp[15]); s.cs2 = uni(sp[16]); s.ss2 = uni(sp[17]);
    if (path == PATH_DELAY4) {
#pragma unroll
        for (int i = 0; i < 4; i++) s.tail_z[i] = make_float2(uni(sp[18 + 2 * i]), uni(sp[19 + 2 * i]));
    } else if (path == PATH_AM_RAW) {
#pragma unroll
        for (int i = 0; i < 4; i++) s.tail_q[i] = uni(__float_as_uint(sp[18 + i]));
    }
}
SSDR_DEV void slot_store(float *sp, const Slot &s, int path, int lane)
{
    if (lane == 0) {
        sp[0] = __uint_as_float(s.phi1); sp[1] = __uint_as_float(s.phi2);
        sp[2] = s.dc; sp[3] = s.agc_d;
#pragma unroll
        for (int i = 0; i < 8; i++) sp[4 + i] = s.agc_m[i];
        sp[12] = s.prev_re; sp[13] = s.prev_im;
        if (path == PATH_DELAY4) {
#pragma unroll
            for (int i = 0; i < 4; i++) { sp[18 + 2 * i] = s.tail_z[i].x; sp[19 + 2 * i] = s.tail_z[i].y; }
        } else if (path == PATH_AM_RAW) {
#pragma unroll
            for (int i = 0; i < 4; i++) sp[18 + i] = __uint_as_float(s.tail_q[i]);
        }
    }
}

// one oscillator for the two frames of a line: P(8 l dphi) of this lane and, in lanes 0 / 1, the phasors of the line's frames --
// the values the stand-alone kernel holds per call (nco_setup) and per 64 frames (nco_frame_table), from the same arguments
SSDR_DEV void nco_line(Nco &n, uint32_t dphi, uint32_t phase_of_line, float cs, float ss, int l)
{
    n.dphi = dphi; n.cs = cs; n.ss = ss;
    ssdr_phasor32((uint32_t)(8 * l) * dphi, n.qc, n.qs);
    ssdr_phasor32(phase_of_line + (uint32_t)(SSDR_FRAME * l) * dphi, n.tc, n.ts);
}

SSDR_DEV int dev_audio_path(const ssdr_chan_consts &k)          // ssdr_kernels.h:ssdr_audio_path
{
    if (!(k.fir_flags & SSDR_FIR_DELAY4) || k.mode == SSDR_MODE_IQ) return PATH_GENERAL;
    return k.mode == SSDR_MODE_AM ? PATH_AM_RAW : PATH_DELAY4;
}

struct LineCtx {                                // what the audio phase of one slot needs to know (all wave-uniform)
    uint32_t cc;                                // channel
    uint32_t line, n_frames;
    const uint32_t *raw_lds;                    // the slot's raw line in the LDS (null: fetch the frames from `raw_glb`)
    const uint32_t *raw_glb;                    // the line in global memory
    unsigned char *S;                           // FIR work area (general path)
    float4 *hist;                               // the slot's filter history (general path): HMAX octets, 4 float4 each
    bool last_line;
};

// Per-frame RSSI and ADC-overflow flag without a per-lane keeper that would have to live across the FFT: lane 0 leaves the frame's power
// sum (the same scan, the same order) and the flag in the output rows; the conversion to dBm runs once per call (rssi_finish).
SSDR_DEV void rssi_flag_raw(const float (&p)[8], bool clip, uint32_t f, int l, float *rssi_row, uint8_t *flag_row)
{
    float ps = p[0];
#pragma unroll
    for (int j = 1; j < 8; j++) ps = ps + p[j];
    const float tot = lane63(scan_sum(ps));
    if (l == 0) { rssi_row[f] = tot; flag_row[f] = clip ? (uint8_t)1 : (uint8_t)0; }
}
SSDR_DEV void rssi_finish(float *rssi_row, uint32_t n_frames, float cal, int l)
{
    // lane 0's stores of the call have left the wave (written through to this XCD's L2; no agent-scope release: that would write the
    // whole L2 back, once per channel) ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t f0 = 0; f0 < n_frames; f0 += 64) {
        const uint32_t f = f0 + (uint32_t)l;
        if (f < n_frames) {
            const float sum = __hip_atomic_load(rssi_row + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ... and are read from there, past the L1
            rssi_row[f] = fmaf(ssdr_log2p(fmaxf(sum, 1e-20f)) - 39.0f, SSDR_DB_PER_LOG2, cal);
        }
    }
}

// The audio chain of one channel for the two frames of a line.  PATH as in ssdr_audio.hip:channel_frames, whose per-frame code this
// is, statement for statement, with the carried state handed in and out.
template <int PATH>
SSDR_DEV void audio_line(const SsdrAudioArgs &u, const LineCtx &x, Slot &s, const ssdr_chan_consts &kc, const int l)
{
    const uint32_t mode = kc.mode;
    const uint32_t tap_groups = kc.tap_groups;
    const uint32_t nblk = (kc.ntap + 7) >> 3;
    const uint32_t dphi1 = kc.dphi1, dphi2 = kc.dphi2;
    const AgcK agc = {kc.agc_c0, kc.agc_c1, kc.agc_knee, kc.agc_delta8, kc.hang_frames};
    const bool ssb = mode >= SSDR_MODE_LSB && mode <= SSDR_MODE_CW;
    const bool untuned = dphi1 == 0 && s.phi1 == 0;         // stays what it is for the whole call (phi1 += 512 * 0)
    Nco n1, n2;
    n1.qc = n1.qs = n1.tc = n1.ts = 0.0f; n2.qc = n2.qs = n2.tc = n2.ts = 0.0f;
    if (PATH != PATH_AM_RAW) {
        if (!untuned) nco_line(n1, dphi1, s.phi1, s.cs1, s.ss1, l);
        if (ssb) nco_line(n2, dphi2, s.phi2, s.cs2, s.ss2, l);
    }
    if (PATH == PATH_GENERAL) {
        // the channel's taps into the work area's padding (lane k: tap k; 8 (HMAX + 1) = 40 >= ntap + 7 of them), the FIR reads them back as
        // broadcasts; and the previous line's last HMAX octets in front of the frame
        if (l < 8 * (HMAX + 1)) *reinterpret_cast<float *>(x.S + s_tap_addr(l >> 2) + 4 * (l & 3)) = u.taps[(size_t)x.cc * SSDR_NTAP_MAX + l];
        if (l < HMAX) {
            float4 *d = al16<float4>(x.S + s_addr(l));
            const float4 *hs = al16<const float4>(x.hist + 4 * l);
#pragma unroll
            for (int i = 0; i < 4; i++) d[i] = hs[i];
        }
    }
    int16_t *dst = u.pcm + ((uint64_t)x.cc * x.n_frames + 2 * x.line) * SSDR_FRAME + 8 * l;
    float *rssi_row = u.rssi + (uint64_t)x.cc * x.n_frames;
    uint8_t *flag_row = u.flags + (uint64_t)x.cc * x.n_frames;
    u32x4 raw0 = {0, 0, 0, 0}, raw1 = {0, 0, 0, 0};

#pragma unroll 1
    for (int f = 0; f < 2; f++, dst += SSDR_FRAME) {
        const uint32_t frame = 2 * x.line + f;
        if (x.raw_lds) {
            const u32x4 *qp = al16<const u32x4>(x.raw_lds + SSDR_FRAME * f) + 2 * l;
            raw0 = qp[0]; raw1 = qp[1];
        } else {                                // plain loads: the line is read once more when it is filed for the FFT (the L2 has it)
            const u32x4 *gp = reinterpret_cast<const u32x4 *>(x.raw_glb + SSDR_FRAME * f + 8 * l);
            raw0 = gp[0]; raw1 = gp[1];
        }
        const uint32_t rw[8] = {raw0.x, raw0.y, raw0.z, raw0.w, raw1.x, raw1.y, raw1.z, raw1.w};
        float p[8], aud[8];
        float yr[8], yi[8];
        bool clip;
        float pm_am = -1.0f;

        if constexpr (PATH == PATH_AM_RAW) {
            uint32_t q[8], d[8];
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = iq_power(rw[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) { d[j] = from_prev_lane_u(s.tail_q[j], q[4 + j]); d[4 + j] = q[j]; }
#pragma unroll
            for (int j = 0; j < 4; j++) s.tail_q[j] = lane63_u(q[4 + j]);
#pragma unroll
            for (int j = 0; j < 8; j++) p[j] = (float)d[j];
            pm_am = block_peak(p);
            const bool trig = wave_any(pm_am >= 1073676160.0f) || s.tail_q[0] >= 0x3FFF0001u || s.tail_q[1] >= 0x3FFF0001u ||
                              s.tail_q[2] >= 0x3FFF0001u || s.tail_q[3] >= 0x3FFF0001u;
            clip = trig ? wave_any(raw_clipped(rw)) : false;
            demod_am<true>(p, s.dc, aud);
        } else {
            float amax = 0.0f;
            float2 A[8], B[8];
            if (untuned) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float xr = (float)(int16_t)(rw[j] & 0xFFFFu), xi = (float)((int32_t)rw[j] >> 16);
                    amax = vmax3_abs(amax, xr, xi);
                    A[j] = make_float2(xr, xi);
                }
            } else {
                float bc, bs;
                nco_block(n1, (uint32_t)f, bc, bs);
                mix8<true>(rw, bc, bs, s.cs1, s.ss1, A, amax);
            }
            clip = wave_any(amax >= 32767.0f);
            if constexpr (PATH == PATH_DELAY4) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    yr[j] = from_prev_lane(s.tail_z[j].x, A[4 + j].x);
                    yi[j] = from_prev_lane(s.tail_z[j].y, A[4 + j].y);
                    yr[4 + j] = A[j].x;
                    yi[4 + j] = A[j].y;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) s.tail_z[j] = make_float2(lane63(A[4 + j].x), lane63(A[4 + j].y));
            } else {
                s_store_oct(x.S, HMAX + l, A);
                lds_sync();
                // FIR: ssdr_audio.hip's block loop.  The oldest octet it touches is l - 5 (block 4's second half: taps 33 .. 39, all
                // zero for the <= 33 taps this kernel takes); lane 0's would lie in front of the history: any finite samples do for a
                // product with +0 (the sums never hold a -0), so the index stops at 0.
#pragma unroll
                for (int j = 0; j < 8; j++) { yr[j] = 0.0f; yi[j] = 0.0f; }
                uint32_t a_oct = 0;
                for (uint32_t b = 0; b < nblk; b += 2) {
                    const uint32_t m4 = (tap_groups >> (2 * b)) & 15u;
                    if (m4 == 0) continue;
                    const unsigned char *hq = x.S + s_tap_addr(2 * (int)b);        // taps 8 b .. : padding chunks 2 b, 2 b + 1 (, + 2, + 3)
                    if (m4 & 3u) {
                        if (a_oct != b) s_load_oct(x.S, HMAX + l - (int)b, A);
                        s_load_oct(x.S, max(HMAX + l - 1 - (int)b, 0), B);
                        const float4 h0 = *al16<const float4>(hq), h1 = *al16<const float4>(hq + 272);
                        const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                        if (m4 & 1u) fir_taps<0, 4>(h, A, B, yr, yi);
                        if (m4 & 2u) fir_taps<4, 4>(h, A, B, yr, yi);
                    }
                    if (m4 & 12u) {
                        if (!(m4 & 3u)) s_load_oct(x.S, max(HMAX + l - 1 - (int)b, 0), B);
                        s_load_oct(x.S, max(HMAX + l - 2 - (int)b, 0), A);
                        const float4 h0 = *al16<const float4>(hq + 544), h1 = *al16<const float4>(hq + 816);
                        const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                        if (m4 & 4u) fir_taps<0, 4>(h, B, A, yr, yi);
                        if (m4 & 8u) fir_taps<4, 4>(h, B, A, yr, yi);
                        a_oct = b + 2;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) p[j] = fmaf(yr[j], yr[j], yi[j] * yi[j]);
            if (mode == SSDR_MODE_AM) demod_am<false>(p, s.dc, aud);
            else if (mode <= SSDR_MODE_CW) {
                float b2c, b2s;
                nco_block(n2, (uint32_t)f, b2c, b2s);
                demod_ssb(yr, yi, b2c, b2s, s.cs2, s.ss2, aud);
            } else demod_fm(yr, yi, s.prev_re, s.prev_im, kc.kfm, aud);     // SSDR_MODE_NBFM (SSDR_MODE_IQ never gets here)
            s.prev_re = lane63(yr[7]);
            s.prev_im = lane63(yi[7]);
            if constexpr (PATH == PATH_DELAY4) { s.prev_re = s.prev_re + 0.0f; s.prev_im = s.prev_im + 0.0f; }
        }

        agc_pack_store(p, aud, l, agc, s.agc_d, s.agc_m, dst, pm_am);
        rssi_flag_raw(p, clip, frame, l, rssi_row, flag_row);
        s.phi1 += (uint32_t)SSDR_FRAME * dphi1;
        s.phi2 += (uint32_t)SSDR_FRAME * dphi2;
        if constexpr (PATH == PATH_GENERAL) {   // the frame's tail: the next frame's history (frame 1's waits in `hist` for the next line)
            lds_sync();
            if (l < HMAX) {
                const float4 *t = al16<const float4>(x.S + s_addr(64 + l));
                float4 *d = al16<float4>(x.S + s_addr(l));
                float4 *hd = al16<float4>(x.hist + 4 * l);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 v = t[i];
                    if (f == 0) d[i] = v; else hd[i] = v;
                }
            }
            lds_sync();
        }
    }
    if constexpr (PATH == PATH_AM_RAW) {
        if (x.last_line) {      // the discriminator memory an AM channel leaves behind: y[511] = z1[507] of the call's last frame, mixed as
            const uint32_t rw[8] = {raw0.x, raw0.y, raw0.z, raw0.w, raw1.x, raw1.y, raw1.z, raw1.w};      // the stand-alone kernel mixes it
            float2 Z[8];
            float unused = 0.0f, fc, fs, qc, qs, bc, bs;
            ssdr_phasor32(s.phi1 - (uint32_t)SSDR_FRAME * dphi1, fc, fs);
            ssdr_phasor32((uint32_t)(8 * l) * dphi1, qc, qs);
            phasor_mul(fc, fs, qc, qs, bc, bs);
            mix8<false>(rw, bc, bs, s.cs1, s.ss1, Z, unused);
            s.prev_re = lane63(Z[3].x) + 0.0f;
            s.prev_im = lane63(Z[3].y) + 0.0f;
        }
    }
}

// the carried state of one channel at the start of a call: ssdr_audio.hip:channel_frames' prologue, results into the slot's LDS words
// (and the general path's filter history: the raw tail's last HMAX octets re-mixed exactly as the previous frame mixed them)
SSDR_DEV void slot_begin(const SsdrAudioArgs &u, uint32_t cc, const ssdr_chan_consts &kc, int path, float *sp, float4 *hist, int l)
{
    const ssdr_chan_state st = u.state[cc];
    const uint32_t dphi1 = kc.dphi1, dphi2 = kc.dphi2;
    float cs1, ss1, cs2, ss2;
    ssdr_phasor32(dphi1, cs1, ss1);
    ssdr_phasor32(dphi2, cs2, ss2);
    const uint32_t *hraw = u.hist + (size_t)cc * SSDR_HIST;
    if (path == PATH_GENERAL) {
        if (l < HMAX) {
            const uint4 *hp = reinterpret_cast<const uint4 *>(hraw + 8 * (HOCT - HMAX + l));
            const uint4 h0 = hp[0], h1 = hp[1];
            const uint32_t rw[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            float2 H[8];
            float unused = 0.0f, fc, fs, qc, qs, bc, bs;
            ssdr_phasor32(st.phi1 - (uint32_t)SSDR_FRAME * dphi1, fc, fs);
            ssdr_phasor32((uint32_t)(8 * (64 - HMAX + l)) * dphi1, qc, qs);
            phasor_mul(fc, fs, qc, qs, bc, bs);
            mix8<false>(rw, bc, bs, cs1, ss1, H, unused);
#pragma unroll
            for (int i = 0; i < 4; i++) al16<float4>(hist + 4 * l)[i] = make_float4(H[2 * i].x, H[2 * i].y, H[2 * i + 1].x, H[2 * i + 1].y);
        }
    }
    float tail[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (path != PATH_GENERAL) {
        const uint4 *hp = reinterpret_cast<const uint4 *>(hraw + SSDR_HIST - 8);
        const uint4 h0 = hp[0], h1 = hp[1];
        const uint32_t rw[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        if (path == PATH_DELAY4) {
            float2 H[8];
            float unused = 0.0f, fc, fs, qc, qs, bc, bs;
            ssdr_phasor32(st.phi1 - (uint32_t)SSDR_FRAME * dphi1, fc, fs);
            ssdr_phasor32((uint32_t)(8 * 63) * dphi1, qc, qs);
            phasor_mul(fc, fs, qc, qs, bc, bs);
            mix8<false>(rw, bc, bs, cs1, ss1, H, unused);
#pragma unroll
            for (int j = 0; j < 4; j++) { tail[2 * j] = H[4 + j].x; tail[2 * j + 1] = H[4 + j].y; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) tail[j] = __uint_as_float(iq_power(rw[4 + j]));
        }
    }
    if (l == 0) {
        sp[0] = __uint_as_float(st.phi1); sp[1] = __uint_as_float(st.phi2);
        sp[2] = st.dc; sp[3] = st.agc_d;
#pragma unroll
        for (int i = 0; i < 8; i++) sp[4 + i] = st.agc_m[i];
        sp[12] = st.prev_re; sp[13] = st.prev_im;
        sp[14] = cs1; sp[15] = ss1; sp[16] = cs2; sp[17] = ss2;
#pragma unroll
        for (int i = 0; i < 8; i++) sp[18 + i] = tail[i];
    }
}

// a channel's 4 KB line, 16 bytes per lane and instruction, into its place in the work area (natural order)
template <bool LAST_USE>
SSDR_DEV void file_line(const uint32_t *row, unsigned char *dst, int lane)
{
    u32x4 t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(row) + 64 * i + lane;
        t[i] = LAST_USE ? SSDR_NT_LOAD(p) : *p;
    }
    SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < 4; i++) al16<u32x4>(dst)[64 * i + lane] = t[i];
    SCHED_FENCE();
}

// The kernel's arguments are ~45 scalar registers' worth of pointers and counts; held for the whole kernel they, the carried state of a slot
// and the channel constants do not fit the scalar file, and what spills goes through vector registers.  Each phase therefore reads what it
// needs from the kernarg segment again (scalar loads, cached) through a laundered pointer, and nothing of it lives across the other phase.
typedef const __attribute__((address_space(4))) SsdrFusedArgs *KArgs;
#ifndef SSDR_GEN_KARGS
#define SSDR_GEN_KARGS 1
#endif
SSDR_DEV KArgs args_now()
{
    KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    if (SSDR_GEN_KARGS) asm volatile("" : "+s"(p));         // (0: A/B -- the arguments are loaded once and kept)
    return p;
}

template <bool AVG>
__global__ __launch_bounds__(SSDR_GEN_BLOCK, SSDR_GEN_WAVES_PER_EU) void ssdr_fused_gen_kernel(SsdrFusedArgs fa_unused)
{
#if defined(__HIP_DEVICE_COMPILE__)             // (the host pass only needs the kernel's symbol; address space 4 exists on the device side)
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];
    (void)fa_unused;
    {
        const SsdrWfArgs a0 = args_now()->wf;
        load_tables(smem, a0.win, a0.tw_stage, a0.lut);
    }
    const SsdrWfArgs a = args_now()->wf;                // (what the loop bounds and the call-start / call-end code use)
    const SsdrAudioArgs u = args_now()->au;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, l = lane & 31;
    unsigned char *work = smem + LDS_XCH + wave * WAVE_BYTES;
    float *xch_wave = reinterpret_cast<float *>(work);
    float4 *hist_lds = reinterpret_cast<float4 *>(work + WORK_BYTES);
    float *state_lds = reinterpret_cast<float *>(work + WORK_BYTES + 2 * HIST_BYTES);
    const unsigned char *lut = smem + LDS_LUT0;
    const uint32_t n_pairs = (a.n_ch + 1) >> 1;
    const uint32_t wave_stride = gridDim.x * WAVES;
    const uint32_t n_frames = u.n_frames;

    for (uint32_t pair = blockIdx.x * WAVES + wave; pair < n_pairs; pair += wave_stride) {
        const uint32_t ch_raw = 2 * pair + h;
        const bool ch_ok = ch_raw < a.n_ch;
        const uint32_t ch = ch_ok ? ch_raw : a.n_ch - 1;
        const float cal_wf = a.consts[ch].wf_cal_lin * SSDR_LUT_SCALE;
        const uint32_t n_sub = (2 * pair + 1 < a.n_ch) ? 2u : 1u;

        // ---- call start: both slots' carried state into the LDS
        int path0 = PATH_AM_RAW, path1 = PATH_AM_RAW;
#pragma unroll 1
        for (uint32_t sidx = 0; sidx < n_sub; sidx++) {
            const uint32_t cc = 2 * pair + sidx;
            const ssdr_chan_consts &kc = u.consts[cc];
            const int path = __builtin_amdgcn_readfirstlane(dev_audio_path(kc));
            if (sidx == 0) path0 = path; else path1 = path;
            slot_begin(u, cc, kc, path, state_lds + sidx * STATE_WORDS, hist_lds + sidx * (HIST_BYTES / 16), lane);
        }
        wave_lds_sync();
        // a pair with one general-path channel runs it FIRST: its work area lies where the other slot's line will be filed
        const uint32_t first = (n_sub == 2 && path1 == PATH_GENERAL && path0 != PATH_GENERAL) ? 1u : 0u;
        // AVG: the N-line sums rest in global memory (16 dwords per lane, the wave's own 4 KB: L2-resident) while the audio chain has the registers
        uint32_t *park = args_now()->park + ((size_t)(blockIdx.x * WAVES + wave) * 16) * 64 + lane;
        uint32_t acc[AVG ? 16 : 1];
#pragma unroll
        for (int j = 0; j < (AVG ? 16 : 1); j++) acc[j] = 0;
        if (AVG) {
#pragma unroll
            for (int j = 0; j < 16; j++) park[64 * j] = 0u;
        }

        for (uint32_t line = 0; line < a.n_lines; line++) {
            prio_latency_phase();
            // ---- audio: one slot after the other, all 64 lanes on one channel
#pragma unroll 1
            for (uint32_t k = 0; k < (SSDR_GEN_ABLATE == 2 ? 0u : 2u); k++) {
                const uint32_t sidx = first ^ k;
                if (sidx >= n_sub) continue;                                      // wave-uniform
                uint32_t pair_now = __builtin_amdgcn_readfirstlane(pair);
                asm volatile("" : "+s"(pair_now));      // everything derived from the pair and from the lane id is recomputed per phase (a few
                const int ln = opaque(lane);            // fast ops) rather than hoisted out of the line loop into registers that live across the FFT
                const uint32_t cc = 2 * pair_now + sidx;
                const KArgs kp = args_now();
                const SsdrAudioArgs u = kp->au;                                  // (shadows the outer copy: this phase's own scalar loads)
                const ssdr_chan_consts &kc = u.consts[cc];
                const int path = sidx ? path1 : path0;
                const uint32_t *row = kp->wf.iq + (uint64_t)cc * kp->wf.ch_stride + (uint64_t)line * SSDR_NFFT;
                unsigned char *R = work + (sidx ? R1_OFF : 0);
                // two general channels: slot 1's work area is where its own line belongs -- frames straight from memory, the line filed afterwards
                const bool late = sidx == 1 && path == PATH_GENERAL && path0 == PATH_GENERAL;
                if (!late) { file_line<true>(row, R, ln); wave_lds_sync(); }
                LineCtx x;
                x.cc = cc; x.line = line; x.n_frames = u.n_frames;
                x.raw_lds = late ? nullptr : reinterpret_cast<const uint32_t *>(R);
                x.raw_glb = row;
                x.S = work + ((sidx == 1 && path0 != PATH_GENERAL) ? 0 : R_BYTES);
                x.hist = hist_lds + sidx * (HIST_BYTES / 16);
                x.last_line = line + 1 == kp->wf.n_lines;
                float *sp = state_lds + sidx * STATE_WORDS;
                Slot s;
                slot_load(sp, s, path);
                if (path == PATH_GENERAL) audio_line<PATH_GENERAL>(u, x, s, kc, ln);
                else if (path == PATH_DELAY4) audio_line<PATH_DELAY4>(u, x, s, kc, ln);
                else audio_line<PATH_AM_RAW>(u, x, s, kc, ln);
                slot_store(sp, s, path, ln);
                wave_lds_sync();
                if (late) { file_line<true>(row, R, ln); wave_lds_sync(); }
            }
#if SSDR_GEN_ABLATE == 2
            for (uint32_t sidx = 0; sidx < n_sub; sidx++)
                file_line<true>(a.iq + (uint64_t)(2 * pair + sidx) * a.ch_stride + (uint64_t)line * SSDR_NFFT, work + (sidx ? R1_OFF : 0), lane);
            wave_lds_sync();
#endif
            prio_compute_phase();
            // ---- waterfall: both lines out of the work area, then exactly ssdr_wf_kernel<AVG, false>
            const SsdrWfArgs a = args_now()->wf;                                  // (this phase's own scalar loads)
            uint32_t *hist_out = args_now()->au.hist;
            uint32_t raw[32];
            {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(work + opaque(h) * R1_OFF) + opaque(l);
#pragma unroll
                for (int r = 0; r < 32; r++) raw[r] = q[32 * r];
            }
            wave_lds_sync();
            SCHED_FENCE();
            // the raw tail of the call's last frame (its samples 384..511 = this line's 896..1023) is the next call's history
            if (line + 1 == a.n_lines && ch_ok) {
#pragma unroll
                for (int r = 28; r < 32; r++) hist_out[(size_t)ch * SSDR_HIST + 32 * (r - 28) + l] = raw[r];
            }
            uint32_t qn[16];
            if (AVG) {
                uint32_t *pk = park;
                asm volatile("" : "+v"(pk));                 // (a laundered pointer: the 16 loads go out together and are not forwarded from the stores)
#pragma unroll
                for (int j = 0; j < 16; j++) acc[j] = pk[64 * j];
            }
#if SSDR_GEN_ABLATE == 1
#pragma unroll
            for (int j = 0; j < 16; j++) qn[j] = raw[j] ^ raw[j + 16];
            if (AVG) {
#pragma unroll
                for (int j = 0; j < 16; j++) acc[j] += qn[j] & 0x00FF00FFu;
            }
#else
            f32x2 z[32];
            window_line(raw, smem, l, z);
            SCHED_FENCE();
            fft_line<true>(z, smem, xch_wave, h, l);
            prio_latency_phase();
            if (AVG) quantise32(z, cal_wf, lut, [&](int j, uint32_t q01) { acc[j] += q01; });
            else quantise32(z, cal_wf, lut, [&](int j, uint32_t q01) { qn[j] = q01; });
#endif
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
                const bool carry_in = AVG && grp == 0 && a.phase != 0;
                const int16_t *cin = a.acc_in + (uint64_t)ch * SSDR_NFFT;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u32x4 v = x128[q * 32 + l];
                    if (carry_in) v += reinterpret_cast<const u32x4 *>(cin)[q * 32 + l];
                    if (ch_ok) SSDR_NT_STORE(v, reinterpret_cast<u32x4 *>(dst) + q * 32 + l);
                }
                wave_lds_sync();
                if (AVG) {
#pragma unroll
                    for (int j = 0; j < 16; j++) acc[j] = 0;
                }
            }
            if (AVG) {
                uint32_t *pk = park;
                asm volatile("" : "+v"(pk));
#pragma unroll
                for (int j = 0; j < 16; j++) pk[64 * j] = acc[j];
            }
        }

        // ---- state back to HBM
        if (a.n_lines) {
#pragma unroll 1
            for (uint32_t sidx = 0; sidx < n_sub; sidx++) {
                const uint32_t cc = 2 * pair + sidx;
                const float *sp = state_lds + sidx * STATE_WORDS;
                ssdr_chan_state st = u.state[cc];
                st.phi1 = __float_as_uint(sp[0]); st.phi2 = __float_as_uint(sp[1]);
                st.dc = sp[2]; st.agc_d = sp[3];
#pragma unroll
                for (int i = 0; i < 8; i++) st.agc_m[i] = sp[4 + i];
                st.prev_re = sp[12]; st.prev_im = sp[13];
                if (lane == 0) u.state[cc] = st;
                rssi_finish(u.rssi + (uint64_t)cc * n_frames, n_frames, u.consts[cc].smeter_cal_db, lane);
            }
        }
        wave_lds_sync();
    }
#endif
}

} // namespace

hipError_t ssdr_launch_fused_gen(const SsdrFusedArgs &a, uint32_t grid, hipStream_t stream)
{
    if (a.wf.n_avg > 1) hipLaunchKernelGGL((ssdr_fused_gen_kernel<true>), dim3(grid), dim3(SSDR_GEN_BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((ssdr_fused_gen_kernel<false>), dim3(grid), dim3(SSDR_GEN_BLOCK), 0, stream, a);
    return hipGetLastError();
}

hipError_t ssdr_fused_gen_blocks_per_cu(int *blocks)
{
    int b0 = 0, b1 = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b0, ssdr_fused_gen_kernel<false>, SSDR_GEN_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b1, ssdr_fused_gen_kernel<true>, SSDR_GEN_BLOCK, 0);
    *blocks = b0 < b1 ? b0 : b1;
    return e;
}
