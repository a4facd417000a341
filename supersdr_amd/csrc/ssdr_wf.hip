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
#include "ssdr_wf_dev.h"

namespace {

constexpr int WAVES = SSDR_WF_BLOCK / 64;
constexpr int LDS_TOTAL = LDS_XCH + WAVES * 2 * XCH_FLOATS * 4;
static_assert(LDS_TOTAL <= 163840, "LDS budget");

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
    const uint32_t run = HOP ? a.grp_run : 1u;
    const uint32_t n_runs = (a.n_groups + run - 1) / run;
    const uint32_t n_items = n_pairs * n_runs;
    const uint32_t wave_stride = gridDim.x * WAVES;

    for (uint32_t item = blockIdx.x * WAVES + wave; item < n_items; item += wave_stride) {
      uint32_t pair, g_begin, g_end;
      if (HOP) {
          pair = item / n_runs;
          g_begin = (item - pair * n_runs) * run;
          g_end = min(g_begin + run, a.n_groups);
      } else {
          g_begin = item / n_pairs;
          pair = item - g_begin * n_pairs;
          g_end = g_begin + 1;
      }
      const uint32_t n_trips = HOP ? g_end - g_begin : 1u;
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
            if (HOP) load_line_halves(line ? src - SSDR_NFFT / 2 : a.tail + (uint64_t)it.ch * (SSDR_NFFT / 2) + l, src, raw);
            else load_line(src, raw);
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
            wave_lds_sync();
            // ---- audio, phase 2: channel A, then channel B, two frames each, all 64 lanes on one channel
            prio_latency_phase();                                             // (the call's first line; later ones arrive with it)
#pragma unroll
            for (int c = 0; c < 2; c++) {
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
            f32x2 z[32];
            window_line(raw, smem, l, z);
            SCHED_FENCE();
            fft_line<true>(z, smem, xch_wave, h, l);          // the averaging kernel's twiddle schedule: fewer registers in flight
            prio_latency_phase();                             // quantiser look-ups, the line's store, the next line's loads and audio phase
            if (AVG) quantise32(z, cal_wf, lut, [&](int j, uint32_t q01) { acc[j] += q01; });
            else quantise32(z, cal_wf, lut, [&](int j, uint32_t q01) { qn[j] = q01; });
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
