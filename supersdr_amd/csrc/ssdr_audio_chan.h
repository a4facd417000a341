// ssdr_audio_chan.h -- the audio chain of ONE receiver channel for all frames of a call, as one wave64 runs it: shared by the
// stand-alone audio kernels (ssdr_audio.hip) and the wave-specialised chain kernel (ssdr_chain_ws.hip), whose audio waves ARE this
// function -- carried state in registers for the whole call -- and additionally hand every raw frame to an FFT wave (the `tap`).
// Stands in for the KiwiSDR server's SND producer whose frames the reference consumes in kiwi_sound.process_audio_stream
// (utils_supersdr.py:1044-1076); tap formula of the channel filter: utils_supersdr.py:334-344 (ssdr_tables.cpp).
#pragma once
#include "ssdr_audio_dev.h"

// A lane's eight samples of a frame are 32 bytes, fetched as two 16-byte loads: a 128-byte line is completed by two load instructions of the
// wave.  PLAIN loads, not non-temporal ones: with the hint a line could be dropped between the two (the kernels read 1.02-1.03 x their input);
// without it the full-band AM kernel runs 3 % faster, the others 0-0.4 % (profiles/r06_ab_plain_loads.txt).
#define SSDR_AUDIO_LOAD(p) (*(p))

namespace {

// `tap(f, raw0, raw1)` sees the lane's eight raw samples of frame f right after they were loaded; the stand-alone kernels pass NoTap.
// Tap::PREFETCH: the next frame's samples are requested before this frame's arithmetic starts (8 more registers; for kernels whose
// occupancy does not hide the load by itself)
struct NoTap {
    static constexpr bool PREFETCH = false;
    SSDR_DEV void operator()(uint32_t, const u32x4 &, const u32x4 &) const {}
};

enum { PATH_GENERAL = SSDR_PATH_GENERAL, PATH_DELAY4 = SSDR_PATH_DELAY4, PATH_AM_RAW = SSDR_PATH_AM_RAW };

// One receiver channel, all frames of the call.
//   PATH_GENERAL  NCO -> LDS -> FIR (any tap set)
//   PATH_DELAY4   the channel filter is a pure 4-sample delay (the reference's full-band +-6 kHz passband at 12 kHz:
//                 one unit tap): the "FIR" is a lane shift by DPP, no LDS
//   PATH_AM_RAW   PATH_DELAY4 and mode AM: |x e^{j phi}| = |x|, so the envelope, the AGC level and the RSSI do not
//                 depend on the NCO at all; the power of a sample is taken exactly in integers (I*I + Q*Q, one
//                 v_dot2) and rounded once
template <int PATH, typename Tap = NoTap>
SSDR_DEV void channel_frames(const SsdrAudioArgs &a, const uint32_t ch, const int l, const ssdr_chan_consts &kc,
                             float2 *s_z, float *s_taps, const Tap &tap = Tap())
{
    const uint32_t mode = kc.mode;
    const uint32_t tap_groups = kc.tap_groups;           // fma(0, z, acc) == acc exactly: all-zero 4-tap groups are skipped
    const uint32_t nblk = (kc.ntap + 7) >> 3;
    const uint32_t dphi1 = kc.dphi1, dphi2 = kc.dphi2;
    const AgcK agc = {kc.agc_c0, kc.agc_c1, kc.agc_knee, kc.agc_delta8, kc.hang_frames};
    const float cal = kc.smeter_cal_db;
    Nco n1, n2;
    nco_setup(n1, dphi1, l);
    nco_setup(n2, dphi2, l);
    const float cs1 = n1.cs, ss1 = n1.ss, cs2 = n2.cs, ss2 = n2.ss;

    ssdr_chan_state st = a.state[ch];
    uint32_t phi1 = st.phi1, phi2 = st.phi2;
    float dc = st.dc, agc_d = st.agc_d, prev_re = st.prev_re, prev_im = st.prev_im;
    float agc_m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) agc_m[i] = st.agc_m[i];

    const bool untuned = dphi1 == 0 && phi1 == 0;         // wave-uniform; stays true for the whole call (phi1 += 512 * 0)
    const uint32_t *hist = a.hist + (size_t)ch * SSDR_HIST;
    float2 tail_z[4];                                   // PATH_DELAY4: mixed samples -4..-1 (wave-uniform)
    uint32_t tail_q[4];                                 // PATH_AM_RAW: I*I + Q*Q of samples -4..-1 (wave-uniform)
    if constexpr (PATH == PATH_GENERAL) {
        {   // the channel's taps go to LDS once; the FIR reads them back as broadcasts
            const float2 t = reinterpret_cast<const float2 *>(a.taps + (size_t)ch * SSDR_NTAP_MAX)[l];
            s_taps[2 * l] = t.x;
            s_taps[2 * l + 1] = t.y;
            if (l < 8) s_taps[SSDR_NTAP_MAX + l] = 0.0f;
        }
        // history z1[-128..-1]: re-mix the raw tail kept in HBM exactly as the previous frame mixed it
        // (block t of the tail was block 48+t of that frame): lanes 0..15, one octet each
        if (l < HOCT) {
            const uint4 *hp = reinterpret_cast<const uint4 *>(hist + 8 * l);
            const uint4 h0 = hp[0], h1 = hp[1];
            const uint32_t rw[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            float2 H[8];
            float unused = 0.0f, fc, fs, qc, qs, bc, bs;
            ssdr_phasor32(phi1 - (uint32_t)SSDR_FRAME * dphi1, fc, fs);                 // the previous frame's phasor
            ssdr_phasor32((uint32_t)(8 * (64 - HOCT + l)) * dphi1, qc, qs);               // its block 48 + l
            phasor_mul(fc, fs, qc, qs, bc, bs);
            mix8<false>(rw, bc, bs, cs1, ss1, H, unused);
            store_oct(s_z, l, H);
        }
    } else {
        // the last block of the raw tail (samples -8..-1), the same bytes in every lane
        const uint4 *hp = reinterpret_cast<const uint4 *>(hist + SSDR_HIST - 8);
        const uint4 h0 = hp[0], h1 = hp[1];
        const uint32_t rw[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        if constexpr (PATH == PATH_DELAY4) {
            float2 H[8];
            float unused = 0.0f, fc, fs, bc, bs;
            ssdr_phasor32(phi1 - (uint32_t)SSDR_FRAME * dphi1, fc, fs);                 // block 63 of the previous frame
            phasor_mul(fc, fs, lane63(n1.qc), lane63(n1.qs), bc, bs);
            mix8<false>(rw, bc, bs, cs1, ss1, H, unused);
#pragma unroll
            for (int j = 0; j < 4; j++) tail_z[j] = H[4 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) tail_q[j] = iq_power(rw[4 + j]);
        }
    }

    const uint32_t *src = a.iq + (uint64_t)ch * a.ch_stride + 8 * l;
    int16_t *dst = a.pcm + (uint64_t)ch * a.n_frames * SSDR_FRAME + 8 * l;
    float *rssi_row = a.rssi + (uint64_t)ch * a.n_frames;
    uint8_t *flag_row = a.flags + (uint64_t)ch * a.n_frames;
    u32x4 raw0, raw1;
    u32x4 nxt0 = {0, 0, 0, 0}, nxt1 = {0, 0, 0, 0};
    if constexpr (Tap::PREFETCH) {
        if (a.n_frames) {
            nxt0 = SSDR_AUDIO_LOAD(reinterpret_cast<const u32x4 *>(src));
            nxt1 = SSDR_AUDIO_LOAD(reinterpret_cast<const u32x4 *>(src) + 1);
        }
    }
    float rssi_sum = 0.0f;
    uint32_t flag_keep = 0;

    for (uint32_t f = 0; f < a.n_frames; f++, src += SSDR_FRAME, dst += SSDR_FRAME) {
        if ((f & 63u) == 0) {                               // the next 64 frames' phasors, one per lane
            if constexpr (PATH != PATH_AM_RAW) nco_frame_table(n1, phi1, l);
            if constexpr (PATH != PATH_AM_RAW) if (mode >= SSDR_MODE_LSB && mode <= SSDR_MODE_CW) nco_frame_table(n2, phi2, l);
        }
        if constexpr (Tap::PREFETCH) {
            raw0 = nxt0;
            raw1 = nxt1;
            if (f + 1 < a.n_frames) {
                nxt0 = SSDR_AUDIO_LOAD(reinterpret_cast<const u32x4 *>(src + SSDR_FRAME));
                nxt1 = SSDR_AUDIO_LOAD(reinterpret_cast<const u32x4 *>(src + SSDR_FRAME) + 1);
            }
        } else {
            raw0 = SSDR_AUDIO_LOAD(reinterpret_cast<const u32x4 *>(src));
            raw1 = SSDR_AUDIO_LOAD(reinterpret_cast<const u32x4 *>(src) + 1);
        }
        tap(f, raw0, raw1);
        const uint32_t rw[8] = {raw0.x, raw0.y, raw0.z, raw0.w, raw1.x, raw1.y, raw1.z, raw1.w};
        float p[8], aud[8];
        float yr[8], yi[8];                                 // the channel filter's output (unused on the full-band AM path)
        bool clip;
        float pm_am = -1.0f;                                // full-band AM path: the block peak, known before the AGC asks for it

        if constexpr (PATH == PATH_AM_RAW) {
            uint32_t q[8], d[8];
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = iq_power(rw[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) { d[j] = from_prev_lane_u(tail_q[j], q[4 + j]); d[4 + j] = q[j]; }
#pragma unroll
            for (int j = 0; j < 4; j++) tail_q[j] = lane63_u(q[4 + j]);
#pragma unroll
            for (int j = 0; j < 8; j++) p[j] = (float)d[j];
            // ADC overflow: a component at the rails makes I*I + Q*Q >= 32767^2; the exact check runs only then
            // (the delayed window of the lanes misses this frame's last four samples: those are in tail_q, scalar)
            pm_am = block_peak(p);                          // (the AGC's block peak: the floor is far below the trigger)
            const bool trig = wave_any(pm_am >= 1073676160.0f) || tail_q[0] >= 0x3FFF0001u || tail_q[1] >= 0x3FFF0001u ||
                              tail_q[2] >= 0x3FFF0001u || tail_q[3] >= 0x3FFF0001u;
            clip = trig ? wave_any(raw_clipped(rw)) : false;
            demod_am<true>(p, dc, aud);
        } else {
            float amax = 0.0f;
            float2 A[8], B[8];
            if (untuned) {
                // the channel sits at the centre of its IQ band and its phase never left zero: every phasor of the NCO is
                // exactly (1, 0) and x * (1 - j0) == x bit for bit -- convert, do not mix
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float xr = (float)(int16_t)(rw[j] & 0xFFFFu), xi = (float)((int32_t)rw[j] >> 16);
                    amax = vmax3_abs(amax, xr, xi);
                    A[j] = make_float2(xr, xi);
                }
            } else {
                float bc, bs;
                nco_block(n1, f, bc, bs);
                mix8<true>(rw, bc, bs, cs1, ss1, A, amax);
            }
            clip = wave_any(amax >= 32767.0f);
            if constexpr (PATH == PATH_DELAY4) {
                // y[n] = z1[n - 4]: the previous lane's last four samples, then this lane's first four
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    yr[j] = from_prev_lane(tail_z[j].x, A[4 + j].x);
                    yi[j] = from_prev_lane(tail_z[j].y, A[4 + j].y);
                    yr[4 + j] = A[j].x;
                    yi[4 + j] = A[j].y;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) tail_z[j] = make_float2(lane63(A[4 + j].x), lane63(A[4 + j].y));
            } else {
                // 1. NCO mix of this lane's 8 samples -> LDS
                store_oct(s_z, HOCT + l, A);
                lds_sync();
                // 2. FIR: y[n] = sum_k h[k] z1[n-k], k ascending, fma chain from zero.  Blocks of 8 taps go in pairs with
                //    the two register octets swapping roles (newer, older) -> (older, newer), so no window is ever copied.
#pragma unroll
                for (int j = 0; j < 8; j++) { yr[j] = 0.0f; yi[j] = 0.0f; }
                uint32_t a_oct = 0;                                      // A currently holds octet (l - a_oct)
                for (uint32_t b = 0; b < nblk; b += 2) {
                    const uint32_t m4 = (tap_groups >> (2 * b)) & 15u;   // wave-uniform
                    if (m4 == 0) continue;
                    const float4 *hq = reinterpret_cast<const float4 *>(s_taps + 8 * b);
                    if (m4 & 3u) {
                        if (a_oct != b) load_oct(s_z, HOCT + l - (int)b, A);
                        load_oct(s_z, HOCT + l - 1 - (int)b, B);
                        const float4 h0 = hq[0], h1 = hq[1];             // same address in all lanes: LDS broadcast
                        const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                        if (m4 & 1u) fir_taps<0, 4>(h, A, B, yr, yi);
                        if (m4 & 2u) fir_taps<4, 4>(h, A, B, yr, yi);
                    }
                    if (m4 & 12u) {
                        if (!(m4 & 3u)) load_oct(s_z, HOCT + l - 1 - (int)b, B);
                        load_oct(s_z, HOCT + l - 2 - (int)b, A);
                        const float4 h0 = hq[2], h1 = hq[3];
                        const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                        if (m4 & 4u) fir_taps<0, 4>(h, B, A, yr, yi);
                        if (m4 & 8u) fir_taps<4, 4>(h, B, A, yr, yi);
                        a_oct = b + 2;
                    }
                }
            }
            // 3. power, demodulation
#pragma unroll
            for (int j = 0; j < 8; j++) p[j] = fmaf(yr[j], yr[j], yi[j] * yi[j]);
            if (mode == SSDR_MODE_AM) demod_am<false>(p, dc, aud);
            else if (mode <= SSDR_MODE_CW) {
                float b2c, b2s;
                nco_block(n2, f, b2c, b2s);
                demod_ssb(yr, yi, b2c, b2s, cs2, ss2, aud);
            }
            else if (mode == SSDR_MODE_NBFM) demod_fm(yr, yi, prev_re, prev_im, kc.kfm, aud);
            else {                                          // SSDR_MODE_IQ: no demodulator, the PCM row carries I
#pragma unroll
                for (int j = 0; j < 8; j++) aud[j] = yr[j];
            }
            // the filter output is an fma chain that ends in "+ 0": a -0 can only come out of the shift path
            prev_re = lane63(yr[7]);
            prev_im = lane63(yi[7]);
            if constexpr (PATH == PATH_DELAY4) { prev_re = prev_re + 0.0f; prev_im = prev_im + 0.0f; }
        }

        // 4./5. AGC, pack, store; 6. RSSI and overflow flag
        const float g = agc_pack_store(p, aud, l, agc, agc_d, agc_m, dst, pm_am);
        if constexpr (PATH == PATH_GENERAL) {
            if (mode == SSDR_MODE_IQ && a.iq_out)           // wave-uniform: I,Q pairs of the filtered baseband under the same gain
                iq_pack_store(yr, yi, g, a.iq_out + ((uint64_t)ch * a.n_frames + f) * SSDR_FRAME + 8 * l);
        }
        rssi_flag_step(p, clip, f, a.n_frames, l, cal, rssi_sum, flag_keep, rssi_row, flag_row);

        // 7. carry: phases advance one frame; the frame tail becomes the FIR history
        phi1 += (uint32_t)SSDR_FRAME * dphi1;
        phi2 += (uint32_t)SSDR_FRAME * dphi2;
        if constexpr (PATH == PATH_GENERAL) {
            lds_sync();
            if (l < HOCT) {
                float2 T[8];
                load_oct(s_z, NOCT - HOCT + l, T);
                store_oct(s_z, l, T);
            }
            lds_sync();
        }
    }

    // state back to HBM (raw tail of the last frame: lanes 48..63 hold it)
    if (a.n_frames) {
        if constexpr (PATH == PATH_AM_RAW) {
            // the discriminator memory is the last filter output, y[511] = z1[507]: mix that one block of the last frame
            const uint32_t rw[8] = {raw0.x, raw0.y, raw0.z, raw0.w, raw1.x, raw1.y, raw1.z, raw1.w};
            float2 Z[8];
            float unused = 0.0f, fc, fs, bc, bs;
            ssdr_phasor32(phi1 - (uint32_t)SSDR_FRAME * dphi1, fc, fs);                 // the last frame's phasor
            phasor_mul(fc, fs, n1.qc, n1.qs, bc, bs);
            mix8<false>(rw, bc, bs, cs1, ss1, Z, unused);
            prev_re = lane63(Z[3].x) + 0.0f;
            prev_im = lane63(Z[3].y) + 0.0f;
        }
        if (l >= 64 - HOCT) {
            u32x4 *hp = reinterpret_cast<u32x4 *>(a.hist + (size_t)ch * SSDR_HIST + 8 * (l - (64 - HOCT)));
            hp[0] = raw0;
            hp[1] = raw1;
        }
        if (l == 0) {
            st.phi1 = phi1; st.phi2 = phi2; st.dc = dc; st.agc_d = agc_d;
#pragma unroll
            for (int i = 0; i < 8; i++) st.agc_m[i] = agc_m[i];
            st.prev_re = prev_re; st.prev_im = prev_im;
            a.state[ch] = st;
        }
    }
}

} // namespace
