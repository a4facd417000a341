// ssdr_audio.hip -- 12 kHz IQ audio chain K2 for gfx950 (MI355X), every stage in one kernel per frame path:
//   int16 IQ -> NCO frequency shift -> FIR low-pass (reference tap formula,
//   utils_supersdr.py:334-344) -> AM envelope / SSB-CW product / NBFM discriminator ->
//   AGC -> int16 PCM + per-frame RSSI
//
// Stands in for the KiwiSDR server's SND producer whose frames the reference consumes
// in kiwi_sound.process_audio_stream (utils_supersdr.py:1044-1076): 512 int16 samples
// per frame plus an S-meter value.
//
// Mapping: one wave64 == one receiver channel (mode, tap count and AGC law are
// wave-uniform, so mixed-mode batches never diverge); lane l owns the 8 consecutive
// samples 8l..8l+7 of the frame.  Frames of a channel are processed in order by the same
// wave with all carried state (NCO phases, DC, AGC envelope, discriminator memory) in
// registers; only the first/last frame of a call touch the state in HBM.
//   * loads: 2 x 16 B per lane, the whole 2 KB frame contiguous per wave
//   * the mixed samples go to LDS once (80 B lane stride: ds_write_b128/ds_read_b128
//     conflict-free); the FIR walks a 16-sample register window per 8-tap block, taps
//     are staged in LDS once per call and read back as broadcasts, 128 FMAs per 6 LDS reads
//   * the NCO is a product of three full-precision phasors (frame, lane block, sample): no polynomial in the frame loop
//     (ssdr_audio_dev.h); full-band channels skip the FIR (lane shift) and, in AM, the NCO as well (channel_frames)
//   * ssdr_set_decimation: IQ at D * 12 kHz, the FIR decimates as D polyphase stream filters (channel_frames_dec)
//   * the recurrences along time are exact or order-defined scans across lanes, done with DPP
//     (row_shr / row_bcast / wave_shr: no LDS, no address arithmetic): AM DC block = affine scan,
//     AGC envelope = (max,+) prefix max (exact), RSSI = sum scan
//   * store: 8 int16 = 16 B per lane, 1 KB contiguous per wave
#include "ssdr_audio_dev.h"
#include "ssdr_audio_chan.h"

namespace {

// The decimating kernels' loads are PLAIN ones: a lane's 8 D consecutive inputs are 64 or 128 bytes, fetched 16 at a time, so a 128-byte line is
// completed by two to eight load instructions of the wave.  Marked non-temporal, a line was dropped between two of them and fetched again: the
// kernel read 1.22 x its input at D = 4 (profiles/r05_traffic_decim4.json); plain loads read 1.01 x and the workload runs 7 % faster
// (profiles/r06_ab_dec_plain_loads.txt).
#define SSDR_DEC_LOAD(p) (*(p))
#ifndef SSDR_DEC_PHASED
#define SSDR_DEC_PHASED 1
#endif
__host__ __device__ constexpr bool dec_phased(int D) { return SSDR_DEC_PHASED && D > 2; }
__host__ __device__ constexpr int dec_streams_per_phase(int D) { return dec_phased(D) ? 2 : D; }

// ---- decimating front end (ssdr_set_decimation: the IQ arrives at D * 12 kHz) -------------------------------------
// Lane l owns the 8 D consecutive inputs that produce its 8 outputs.  Mixed, they de-interleave into D polyphase streams
// v_q[m] = z[D m + q]; y[m] = sum_k h[k] z[D m - k] is then the sum of D ordinary FIRs, one per stream, each on the taps
// the host laid out for it (stream q >= 1 carries its taps behind one zero tap: ssdr_tables.cpp) -- the same register-window
// FIR as the 12 kHz path, run D times on D regions of LDS, 2 FMAs per tap and output sample in total.  Everything behind
// the filter (demodulators, AGC, PCM, RSSI) is the 12 kHz chain.
//
// D = 4 runs the streams in TWO PHASES of two (SSDR_DEC_PHASED): the D regions of LDS are what holds the kernel to 7 waves per CU,
// so the work area is two regions only -- streams 0, 1 are filed and filtered while the mixed samples of streams 2, 3 wait in
// registers, then those take the same two regions -- and every stream's history (its last HOCT_S octets) lives in a small area of
// its own, copied in front of the stream when its phase begins and refreshed by the lanes that hold the frame's tail when they file
// it.  Same sums in the same order: stream 0 first, taps ascending.
template <int D>
SSDR_DEV void channel_frames_dec(const SsdrAudioArgs &a, const uint32_t ch, const int l, const ssdr_chan_consts &kc,
                                 float2 *s_z, float2 *s_h, float *s_taps)
{
    constexpr int SLOTS = SSDR_NTAP_MAX / D;             // tap slots per stream
    constexpr int HOCT_S = SSDR_HIST / D / 8;            // history octets per stream
    constexpr int ROCT = HOCT_S + 64;                    // octets per stream region
    constexpr bool PHASED = dec_phased(D);
    constexpr int SPP = dec_streams_per_phase(D);        // streams per phase (= regions of the work area)
    constexpr int NB = 8 * D;                            // inputs per lane and frame
    constexpr int TAIL_LANES = SSDR_HIST / NB;           // lanes of a frame whose inputs form the 128-sample raw tail
    const uint32_t mode = kc.mode;
    const uint32_t nblk = kc.ntap8 >> 3;                 // per stream
    const uint32_t dphi1 = kc.dphi1, dphi2 = kc.dphi2;
    const AgcK agc = {kc.agc_c0, kc.agc_c1, kc.agc_knee, kc.agc_delta8, kc.hang_frames};
    const float cal = kc.smeter_cal_db;
    Nco n1, n2;
    nco_setup(n1, dphi1, l, NB);
    nco_setup(n2, dphi2, l);
    const float cs1 = n1.cs, ss1 = n1.ss, cs2 = n2.cs, ss2 = n2.ss;

    ssdr_chan_state st = a.state[ch];
    uint32_t phi1 = st.phi1, phi2 = st.phi2;
    float dc = st.dc, agc_d = st.agc_d, prev_re = st.prev_re, prev_im = st.prev_im;
    float agc_m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) agc_m[i] = st.agc_m[i];
    {
        const float2 t = reinterpret_cast<const float2 *>(a.taps + (size_t)ch * SSDR_NTAP_MAX)[l];
        s_taps[2 * l] = t.x;
        s_taps[2 * l + 1] = t.y;
        if (l < 8) s_taps[SSDR_NTAP_MAX + l] = 0.0f;
    }
    uint32_t *hist = a.hist + (size_t)ch * SSDR_HIST;
    // history: the raw tail is the input of the previous frame's last TAIL_LANES lanes; lane l re-mixes the block of lane
    // 64 - TAIL_LANES + l exactly as that frame did and files it in front of each stream
    if (l < TAIL_LANES) {
        uint32_t rw[NB];
        const uint4 *hp = reinterpret_cast<const uint4 *>(hist + NB * l);
#pragma unroll
        for (int i = 0; i < NB / 4; i++) { const uint4 v = hp[i]; rw[4 * i] = v.x; rw[4 * i + 1] = v.y; rw[4 * i + 2] = v.z; rw[4 * i + 3] = v.w; }
        float2 Z[NB];
        float unused = 0.0f, fc, fs, qc, qs, bc, bs;
        ssdr_phasor32(phi1 - (uint32_t)(SSDR_FRAME * D) * dphi1, fc, fs);
        ssdr_phasor32((uint32_t)(NB * (64 - TAIL_LANES + l)) * dphi1, qc, qs);
        phasor_mul(fc, fs, qc, qs, bc, bs);
        mix8<false, NB>(rw, bc, bs, cs1, ss1, Z, unused);
#pragma unroll
        for (int q = 0; q < D; q++) {
            float2 V[8];
#pragma unroll
            for (int j = 0; j < 8; j++) V[j] = Z[D * j + q];
            if constexpr (PHASED) store_oct(s_h + q * HOCT_S * OCT, l, V);
            else store_oct(s_z + q * ROCT * OCT, l, V);
        }
    }

    const uint32_t *src = a.iq + (uint64_t)ch * a.ch_stride + NB * l;
    int16_t *dst = a.pcm + (uint64_t)ch * a.n_frames * SSDR_FRAME + 8 * l;
    float *rssi_row = a.rssi + (uint64_t)ch * a.n_frames;
    uint8_t *flag_row = a.flags + (uint64_t)ch * a.n_frames;
    float rssi_sum = 0.0f;
    uint32_t flag_keep = 0;

    for (uint32_t f = 0; f < a.n_frames; f++, src += SSDR_FRAME * D, dst += SSDR_FRAME) {
        if ((f & 63u) == 0) {
            nco_frame_table(n1, phi1, l, SSDR_FRAME * D);
            if (mode >= SSDR_MODE_LSB && mode <= SSDR_MODE_CW) nco_frame_table(n2, phi2, l);
        }
        uint32_t rw[NB];
#pragma unroll
        for (int i = 0; i < NB / 4; i++) {
            const u32x4 v = SSDR_DEC_LOAD(reinterpret_cast<const u32x4 *>(src) + i);
            rw[4 * i] = v.x; rw[4 * i + 1] = v.y; rw[4 * i + 2] = v.z; rw[4 * i + 3] = v.w;
        }
        float p[8], aud[8], yr[8], yi[8];
        {
            float amax = 0.0f, bc, bs;
            nco_block(n1, f, bc, bs);
            // eight inputs at a time: input 8 sb + t belongs to stream (t mod D), element (8 sb + t) / D of the lane's octet there;
            // a sub-block fills 8 / D consecutive elements of every stream (only 8 mixed samples are ever live: 4 waves per SIMD)
            constexpr int EPS = 8 / D;                       // elements per stream and sub-block (2 at D = 4, 4 at D = 2)
            float2 held[PHASED ? (D - SPP) * 8 : 1];         // the later phases' streams, mixed, until their turn
            if constexpr (PHASED) {                          // phase 0: the streams' history in front of them (read before the tail lanes refresh it)
                if (l < HOCT_S * SPP) {
                    const int q = l / HOCT_S, o = l - q * HOCT_S;
                    float2 T[8];
                    load_oct(s_h + q * HOCT_S * OCT, o, T);
                    store_oct(s_z + q * ROCT * OCT, o, T);
                }
            }
#pragma unroll
            for (int sb = 0; sb < D; sb++) {
                float2 Z8[8];
                mix8_carry<true>(rw + 8 * sb, bc, bs, cs1, ss1, Z8, amax);
#pragma unroll
                for (int q = 0; q < D; q++) {
                    if (!PHASED || q < SPP) {
                        float4 *dstq = reinterpret_cast<float4 *>(s_z + q * ROCT * OCT + (HOCT_S + l) * OCT + EPS * sb);
#pragma unroll
                        for (int u = 0; u < EPS; u += 2)
                            dstq[u >> 1] = make_float4(Z8[q + D * u].x, Z8[q + D * u].y, Z8[q + D * (u + 1)].x, Z8[q + D * (u + 1)].y);
                    } else {
#pragma unroll
                        for (int u = 0; u < EPS; u++) held[(q - SPP) * 8 + EPS * sb + u] = Z8[q + D * u];
                    }
                }
            }
            if constexpr (PHASED) {                          // the frame's tail is the next frame's history
                if (l >= 64 - HOCT_S) {
#pragma unroll
                    for (int q = 0; q < SPP; q++) {
                        float2 T[8];
                        load_oct(s_z + q * ROCT * OCT, HOCT_S + l, T);
                        store_oct(s_h + q * HOCT_S * OCT, l - (64 - HOCT_S), T);
                    }
                }
            }
            lds_sync();
            const bool clip = wave_any(amax >= 32767.0f);
#pragma unroll
            for (int j = 0; j < 8; j++) { yr[j] = 0.0f; yi[j] = 0.0f; }
            // the D stream filters, one after the other into the same accumulators: stream 0 first, taps ascending
            for (int q = 0; q < D; q++) {
                if constexpr (PHASED) {
                    if (q == SPP) {                          // phase 1: streams SPP.. take the work area over
                        lds_sync();
                        if (l < HOCT_S * SPP) {
                            const int q2 = l / HOCT_S, o = l - q2 * HOCT_S;
                            float2 T[8];
                            load_oct(s_h + (SPP + q2) * HOCT_S * OCT, o, T);
                            store_oct(s_z + q2 * ROCT * OCT, o, T);
                        }
#pragma unroll
                        for (int q2 = 0; q2 < D - SPP; q2++) {
                            float2 V[8];
#pragma unroll
                            for (int j = 0; j < 8; j++) V[j] = held[q2 * 8 + j];
                            store_oct(s_z + q2 * ROCT * OCT, HOCT_S + l, V);
                            if (l >= 64 - HOCT_S) store_oct(s_h + (SPP + q2) * HOCT_S * OCT, l - (64 - HOCT_S), V);
                        }
                        lds_sync();
                    }
                }
                const float2 *zq = s_z + (PHASED ? q % SPP : q) * ROCT * OCT;
                const float *gq = s_taps + q * SLOTS;
                float2 A[8], B[8];
                for (uint32_t b = 0; b < nblk; b += 2) {
                    const float4 *hq = reinterpret_cast<const float4 *>(gq + 8 * b);
                    load_oct(zq, HOCT_S + l - (int)b, A);
                    load_oct(zq, HOCT_S + l - 1 - (int)b, B);
                    {
                        const float4 h0 = hq[0], h1 = hq[1];
                        const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                        fir_taps<0, 8>(h, A, B, yr, yi);
                    }
                    if (b + 1 < nblk) {                          // (an odd block count: the next slots are the next stream's)
                        load_oct(zq, HOCT_S + l - 2 - (int)b, A);
                        const float4 h0 = hq[2], h1 = hq[3];
                        const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
                        fir_taps<0, 8>(h, B, A, yr, yi);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) p[j] = fmaf(yr[j], yr[j], yi[j] * yi[j]);
            if (mode == SSDR_MODE_AM) demod_am<false>(p, dc, aud);
            else if (mode <= SSDR_MODE_CW) {
                float b2c, b2s;
                nco_block(n2, f, b2c, b2s);
                demod_ssb(yr, yi, b2c, b2s, cs2, ss2, aud);
            } else if (mode == SSDR_MODE_NBFM) demod_fm(yr, yi, prev_re, prev_im, kc.kfm, aud);
            else {
#pragma unroll
                for (int j = 0; j < 8; j++) aud[j] = yr[j];
            }
            prev_re = lane63(yr[7]);
            prev_im = lane63(yi[7]);
            const float g = agc_pack_store(p, aud, l, agc, agc_d, agc_m, dst);
            if (mode == SSDR_MODE_IQ && a.iq_out) iq_pack_store(yr, yi, g, a.iq_out + ((uint64_t)ch * a.n_frames + f) * SSDR_FRAME + 8 * l);
            rssi_flag_step(p, clip, f, a.n_frames, l, cal, rssi_sum, flag_keep, rssi_row, flag_row);
        }
        phi1 += (uint32_t)(SSDR_FRAME * D) * dphi1;
        phi2 += (uint32_t)SSDR_FRAME * dphi2;
        lds_sync();
        if constexpr (!PHASED) {
            if (l < HOCT_S * D) {                                // each stream's tail becomes its history
                const int q = l / HOCT_S, o = l - q * HOCT_S;
                float2 T[8];
                load_oct(s_z + q * ROCT * OCT, 64 + o, T);
                store_oct(s_z + q * ROCT * OCT, o, T);
            }
            lds_sync();
        }
    }

    if (a.n_frames) {
        if (l >= 64 - TAIL_LANES) {                              // raw tail of the last frame: 128 input samples, read once more
            // (not kept in the frame loop's registers: 8 D of them per lane would be live across the whole filter)
            const u32x4 *lp = reinterpret_cast<const u32x4 *>(src - SSDR_FRAME * D);
            u32x4 *hp = reinterpret_cast<u32x4 *>(hist + NB * (l - (64 - TAIL_LANES)));
#pragma unroll
            for (int i = 0; i < NB / 4; i++) hp[i] = lp[i];
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

template <int D>
__global__ __launch_bounds__(SSDR_AUDIO_BLOCK) __attribute__((amdgpu_waves_per_eu(dec_phased(D) ? 3 : 1, 8))) void ssdr_audio_dec_kernel(SsdrAudioArgs a)
{
    constexpr int HOCT_S = SSDR_HIST / D / 8;
    __shared__ __attribute__((aligned(16))) float2 s_z[dec_streams_per_phase(D) * (HOCT_S + 64) * OCT];     // 11.5 KB (D = 2), 10.9 KB (D = 4: two of four streams)
    __shared__ __attribute__((aligned(16))) float2 s_h[dec_phased(D) ? D * HOCT_S * OCT : 1];              // D = 4: every stream's history, 1280 B
    __shared__ __attribute__((aligned(16))) float s_taps[SSDR_NTAP_MAX + 8];
    const int l = threadIdx.x;
    const uint32_t ch = blockIdx.x;
    if (ch >= a.n_ch) return;
    channel_frames_dec<D>(a, ch, l, a.consts[ch], s_z, s_h, s_taps);
}

// One kernel per frame path: the paths differ by a factor of two in registers (the general path holds a 16-sample FIR
// window, the AM shift path fits 64 VGPRs and needs no LDS), and a wave's register file share is fixed per kernel.  The
// host keeps the channels of a ctx sorted by path (chan_list) and launches each non-empty group.
#ifndef SSDR_AUDIO_WAVES
#define SSDR_AUDIO_WAVES 1
#endif
template <int PATH>
__global__ __launch_bounds__(SSDR_AUDIO_BLOCK) __attribute__((amdgpu_waves_per_eu(SSDR_AUDIO_WAVES, 8))) void ssdr_audio_kernel(SsdrAudioArgs a)
{
    constexpr bool FIR = PATH == PATH_GENERAL;
    __shared__ __attribute__((aligned(16))) float2 s_z[FIR ? NOCT * OCT : 1];            // 6400 B
    __shared__ __attribute__((aligned(16))) float s_taps[FIR ? SSDR_NTAP_MAX + 8 : 1];   // + one block of padding for odd block counts
    const int l = threadIdx.x;
    if (blockIdx.x >= a.list_n) return;
    const uint32_t ch = a.chan_list[blockIdx.x];
    channel_frames<PATH>(a, ch, l, a.consts[ch], s_z, s_taps);
}

// ---------------------------------------------------------------- synthetic IQ (bench input)
SSDR_DEV uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
SSDR_DEV float hash_noise(uint32_t h)        // ~N(0,1): Irwin-Hall sum of 4 bytes
{
    const int s = (int)(h & 0xFF) + (int)((h >> 8) & 0xFF) + (int)((h >> 16) & 0xFF) + (int)(h >> 24) - 510;
    return (float)s * 0.0067659f;             // 1/sqrt(4*(256^2-1)/12)
}

__global__ __launch_bounds__(256) void ssdr_synth_kernel(SsdrSynthArgs a)
{
    const uint64_t total = (uint64_t)a.n_ch * a.n_samples;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const uint32_t c = (uint32_t)(idx / a.n_samples);
        const uint32_t n = (uint32_t)(idx - (uint64_t)c * a.n_samples);
        const uint32_t cid = c + a.first_channel_id;
        const uint32_t t = (uint32_t)(a.sample0 + n);
        // carrier f_c = ((cid*37) mod 97 - 48) * 100 Hz; modulation by cid mod 4
        const int fc = ((int)((cid * 37u) % 97u) - 48) * 100;
        const uint32_t m = cid & 3u;
        const int f_tone = fc + (m == 1 ? 1000 : (m == 2 ? -1000 : 0));
        const uint32_t dphi = (uint32_t)(int32_t)((int64_t)f_tone * 357913941ll / 1000ll);   // 2^32/12000 = 357913.941
        uint32_t ph = t * dphi;
        float amp = 8000.0f;
        if (m == 0) {
            float c1, s1;
            ssdr_sincos20(t * 357913941u, c1, s1);                 // 1 kHz
            amp = 8000.0f * fmaf(0.5f, s1, 1.0f);
        } else if (m == 3) {
            float c1, s1;
            ssdr_sincos20(t * 286331153u, c1, s1);                 // 800 Hz
            ph += (uint32_t)(int32_t)(s1 * 1708913189.0f);         // 2.5 rad * 2^32 / (2 pi)
        }
        float cc, ss;
        ssdr_sincos20(ph, cc, ss);
        const uint32_t h = fmix32(a.seed ^ (cid * 0x9E3779B1u) ^ fmix32(t * 2u + 1u));
        const float gi = hash_noise(fmix32(h ^ 0x68E31DA4u)) * 200.0f;
        const float gq = hash_noise(fmix32(h ^ 0xB5297A4Du)) * 200.0f;
        const int vi = (int)rintf(fmaf(amp, cc, gi)), vq = (int)rintf(fmaf(amp, ss, gq));
        a.iq[(uint64_t)c * a.ch_stride + n] = ((uint32_t)vi & 0xFFFFu) | ((uint32_t)vq << 16);
    }
}

// exhaustive check of ssdr_sqrt_rn against the compiler's IEEE sqrtf over its whole domain [0, 2^62)
__global__ void ssdr_sqrt_selftest_kernel(unsigned long long *mismatch)
{
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // every float from the smallest denormal up to 2^62 (the documented domain)
    for (uint64_t u = 1ull + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < 0x5E800000ull; u += stride) {
        const float p = __uint_as_float((uint32_t)u);
        bad += (__float_as_uint(ssdr_sqrt_rn(p)) != __float_as_uint(sqrtf(p)));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bad += (__float_as_uint(ssdr_sqrt_rn(0.0f)) != 0u);
    // the unscaled form of the integer-power path: every float in [1, 2^33), and zero
    for (uint64_t u = 0x3F800000ull + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < 0x50000000ull; u += stride) {
        const float p = __uint_as_float((uint32_t)u);
        bad += (__float_as_uint(ssdr_sqrt_rn_int(p)) != __float_as_uint(sqrtf(p)));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) bad += (__float_as_uint(ssdr_sqrt_rn_int(0.0f)) != 0u);
    if (bad) atomicAdd(mismatch, bad);
}

// the two envelope square roots on caller-chosen arguments (checked on the host against an independent IEEE sqrt)
__global__ void ssdr_sqrt_values_kernel(const float *in, float *out_scaled, float *out_int, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_scaled[i] = ssdr_sqrt_rn(in[i]);
    out_int[i] = ssdr_sqrt_rn_int(in[i]);
}

} // namespace

hipError_t ssdr_launch_sqrt_values(const float *in, float *out_scaled, float *out_int, uint32_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_sqrt_values_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, out_scaled, out_int, n);
    return hipGetLastError();
}

hipError_t ssdr_launch_sqrt_selftest(unsigned long long *mismatch, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_sqrt_selftest_kernel, dim3(2048), dim3(256), 0, stream, mismatch);
    return hipGetLastError();
}

hipError_t ssdr_launch_audio(const SsdrAudioArgs &a, int path, hipStream_t stream)
{
    if (!a.list_n) return hipSuccess;
    switch (path) {
    case SSDR_PATH_GENERAL: hipLaunchKernelGGL(ssdr_audio_kernel<PATH_GENERAL>, dim3(a.list_n), dim3(SSDR_AUDIO_BLOCK), 0, stream, a); break;
    case SSDR_PATH_DELAY4: hipLaunchKernelGGL(ssdr_audio_kernel<PATH_DELAY4>, dim3(a.list_n), dim3(SSDR_AUDIO_BLOCK), 0, stream, a); break;
    case SSDR_PATH_AM_RAW: hipLaunchKernelGGL(ssdr_audio_kernel<PATH_AM_RAW>, dim3(a.list_n), dim3(SSDR_AUDIO_BLOCK), 0, stream, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t ssdr_launch_audio_dec(const SsdrAudioArgs &a, uint32_t decim, hipStream_t stream)
{
    if (decim == 2) hipLaunchKernelGGL(ssdr_audio_dec_kernel<2>, dim3(a.n_ch), dim3(SSDR_AUDIO_BLOCK), 0, stream, a);
    else if (decim == 4) hipLaunchKernelGGL(ssdr_audio_dec_kernel<4>, dim3(a.n_ch), dim3(SSDR_AUDIO_BLOCK), 0, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t ssdr_launch_synth(const SsdrSynthArgs &a, hipStream_t stream)
{
    const uint64_t total = (uint64_t)a.n_ch * a.n_samples;
    const uint32_t grid = (uint32_t)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(ssdr_synth_kernel, dim3(grid ? grid : 1), dim3(256), 0, stream, a);
    return hipGetLastError();
}
