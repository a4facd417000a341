// ssdr_audio.hip -- 12 kHz IQ audio chain K2 for gfx950 (MI355X), one fused kernel:
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
//     come through the scalar cache (wave-uniform), 128 FMAs per 4 LDS reads
//   * the recurrences along time are exact or order-defined scans across lanes:
//     AM DC block = affine Kogge-Stone scan, AGC envelope = (max,+) prefix max (exact),
//     RSSI = xor-butterfly sum
//   * store: 8 int16 = 16 B per lane, 1 KB contiguous per wave
#include "ssdr_math.h"
#include "ssdr_kernels.h"

namespace {

constexpr int OCT = 10;                         // LDS slots (float2) per 8 samples: 8 + 2 pad
constexpr int NOCT = (SSDR_HIST + SSDR_FRAME) / 8;   // 80 octets
constexpr int HOCT = SSDR_HIST / 8;             // 16 history octets

SSDR_DEV void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

SSDR_DEV float2 mix(uint32_t raw, uint32_t phase)
{
    const float xr = (float)(int16_t)(raw & 0xFFFFu);
    const float xi = (float)((int32_t)raw >> 16);
    float c, s;
    ssdr_sincos20(phase, c, s);
    return make_float2(fmaf(xr, c, xi * s), fmaf(xi, c, -(xr * s)));
}

SSDR_DEV void load_oct(const float2 *z, int q, float2 (&v)[8])
{
    const float4 *p = reinterpret_cast<const float4 *>(z + q * OCT);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float4 t = p[i];
        v[2 * i] = make_float2(t.x, t.y);
        v[2 * i + 1] = make_float2(t.z, t.w);
    }
}

SSDR_DEV void store_oct(float2 *z, int q, const float2 (&v)[8])
{
    float4 *p = reinterpret_cast<float4 *>(z + q * OCT);
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = make_float4(v[2 * i].x, v[2 * i].y, v[2 * i + 1].x, v[2 * i + 1].y);
}

__global__ __launch_bounds__(SSDR_AUDIO_BLOCK) void ssdr_audio_kernel(SsdrAudioArgs a)
{
    __shared__ __attribute__((aligned(16))) float2 s_z[NOCT * OCT];      // 6400 B
    constexpr float DC_APOW[8] = SSDR_DC_APOW_INIT;

    const int l = threadIdx.x;
    const uint32_t ch = blockIdx.x;
    if (ch >= a.n_ch) return;

    const ssdr_chan_consts &kc = a.consts[ch];
    const uint32_t mode = kc.mode, nblk = kc.ntap8 >> 3;
    const uint32_t dphi1 = kc.dphi1, dphi2 = kc.dphi2;
    const float c0 = kc.agc_c0, c1 = kc.agc_c1, knee = kc.agc_knee, d8 = kc.agc_delta8;
    const uint32_t K = kc.hang_frames;
    const float cal = kc.smeter_cal_db;
    const float *taps = a.taps + (size_t)ch * SSDR_NTAP_MAX;

    ssdr_chan_state st = a.state[ch];
    uint32_t phi1 = st.phi1, phi2 = st.phi2;
    float dc = st.dc, agc_d = st.agc_d, prev_re = st.prev_re, prev_im = st.prev_im;
    float agc_m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) agc_m[i] = st.agc_m[i];

    // history z1[-128..-1] from the raw tail kept in HBM: 2 samples per lane
    {
        const uint2 hr = reinterpret_cast<const uint2 *>(a.hist + (size_t)ch * SSDR_HIST)[l];
        const int i0 = -SSDR_HIST + 2 * l;
        const float2 z0 = mix(hr.x, phi1 + (uint32_t)i0 * dphi1);
        const float2 z1 = mix(hr.y, phi1 + (uint32_t)(i0 + 1) * dphi1);
        const int s = 2 * l;                                   // slot index s = i + 128
        *reinterpret_cast<float4 *>(&s_z[(s >> 3) * OCT + (s & 7)]) = make_float4(z0.x, z0.y, z1.x, z1.y);
    }

    const uint32_t *src = a.iq + (uint64_t)ch * a.ch_stride + 8 * l;
    int16_t *dst = a.pcm + (uint64_t)ch * a.n_frames * SSDR_FRAME + 8 * l;
    uint4 raw0, raw1;

    for (uint32_t f = 0; f < a.n_frames; f++, src += SSDR_FRAME, dst += SSDR_FRAME) {
        raw0 = reinterpret_cast<const uint4 *>(src)[0];
        raw1 = reinterpret_cast<const uint4 *>(src)[1];
        const uint32_t rw[8] = {raw0.x, raw0.y, raw0.z, raw0.w, raw1.x, raw1.y, raw1.z, raw1.w};

        // 1. NCO mix of this lane's 8 samples -> LDS
        float2 A[8], B[8];
        {
            const uint32_t ph0 = phi1 + (uint32_t)(8 * l) * dphi1;
#pragma unroll
            for (int j = 0; j < 8; j++) A[j] = mix(rw[j], ph0 + (uint32_t)j * dphi1);
            store_oct(s_z, HOCT + l, A);
        }
        lds_sync();

        // 2. FIR: y[n] = sum_k h[k] z1[n-k], k ascending, fma chain from zero
        float yr[8], yi[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { yr[j] = 0.0f; yi[j] = 0.0f; }
        for (uint32_t b = 0; b < nblk; b++) {
            load_oct(s_z, HOCT + l - 1 - (int)b, B);
            float h[8];
#pragma unroll
            for (int kk = 0; kk < 8; kk++) h[kk] = taps[8 * b + kk];
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float2 v = (j - kk >= 0) ? A[(j - kk) & 7] : B[(8 + j - kk) & 7];
                    yr[j] = fmaf(h[kk], v.x, yr[j]);
                    yi[j] = fmaf(h[kk], v.y, yi[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) A[j] = B[j];
        }

        // 3. power, demodulation
        float p[8], aud[8];
#pragma unroll
        for (int j = 0; j < 8; j++) p[j] = fmaf(yr[j], yr[j], yi[j] * yi[j]);

        if (mode == SSDR_MODE_AM) {
            float env[8], loc[8];
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                env[j] = sqrtf(p[j]);
                s = fmaf(SSDR_DC_A, s, SSDR_DC_AL * env[j]);
                loc[j] = s;
            }
            // inclusive Kogge-Stone scan of the affine maps m -> A*m + B over lanes
            float Asc = DC_APOW[7], Bsc = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float Ap = __shfl_up(Asc, d, 64), Bp = __shfl_up(Bsc, d, 64);
                if (l >= d) { Bsc = fmaf(Asc, Bp, Bsc); Asc = Asc * Ap; }
            }
            const float Ae = __shfl_up(Asc, 1, 64), Be = __shfl_up(Bsc, 1, 64);
            const float carry = (l == 0) ? dc : fmaf(Ae, dc, Be);
            float m = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                m = fmaf(DC_APOW[j], carry, loc[j]);
                aud[j] = env[j] - m;
            }
            dc = __shfl(m, 63, 64);
        } else if (mode <= SSDR_MODE_CW) {
            const uint32_t ph0 = phi2 + (uint32_t)(8 * l) * dphi2;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float c, s;
                ssdr_sincos20(ph0 + (uint32_t)j * dphi2, c, s);
                aud[j] = fmaf(yr[j], c, -(yi[j] * s));
            }
        } else {
            float pr = __shfl_up(yr[7], 1, 64), pi = __shfl_up(yi[7], 1, 64);
            if (l == 0) { pr = prev_re; pi = prev_im; }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float dr = fmaf(yr[j], pr, yi[j] * pi);
                const float di = fmaf(yi[j], pr, -(yr[j] * pi));
                aud[j] = ssdr_atan2p(di, dr) * SSDR_KFM;
                pr = yr[j]; pi = yi[j];
            }
        }
        prev_re = __shfl(yr[7], 63, 64);
        prev_im = __shfl(yi[7], 63, 64);

        // 4. AGC: block peak -> log2 -> (max,+) follower across lanes -> gain
        float pm = p[0], ps = p[0];
#pragma unroll
        for (int j = 1; j < 8; j++) { pm = fmaxf(pm, p[j]); ps = ps + p[j]; }
        const float al = ssdr_log2p(fmaxf(pm, SSDR_P_FLOOR));
        const float fl = (float)l;
        float e;
        if (K == 0) {
            float P = fmaf(fl, d8, al);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float t = __shfl_up(P, d, 64);
                if (l >= d) P = fmaxf(P, t);
            }
            e = fmaxf(fmaf(-fl, d8, P), fmaf(-(fl + 1.0f), d8, agc_d));
            agc_d = __shfl(e, 63, 64);
        } else {
            float P = al;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float t = __shfl_up(P, d, 64);
                if (l >= d) P = fmaxf(P, t);
            }
            float maxM = agc_m[0], mK = agc_m[0];
#pragma unroll
            for (int i = 1; i < 8; i++)
                if ((uint32_t)i < K) { maxM = fmaxf(maxM, agc_m[i]); mK = agc_m[i]; }
            e = fmaxf(fmaxf(P, maxM), fmaf(-(fl + 1.0f), d8, agc_d));
            agc_d = fmaxf(fmaf(-64.0f, d8, agc_d), mK);
#pragma unroll
            for (int i = 7; i > 0; i--) agc_m[i] = agc_m[i - 1];
            agc_m[0] = __shfl(P, 63, 64);
        }
        const float g = ssdr_exp2p(fmaf(c1, fmaxf(e, knee), c0));

        // 5. round-half-even, saturate, pack 8 x int16 = 16 B, store
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            float y0 = rintf(aud[j] * g), y1 = rintf(aud[j + 1] * g);
            y0 = fminf(fmaxf(y0, -32768.0f), 32767.0f);
            y1 = fminf(fmaxf(y1, -32768.0f), 32767.0f);
            w[j >> 1] = ((uint32_t)(int32_t)y0 & 0xFFFFu) | ((uint32_t)(int32_t)y1 << 16);
        }
        *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);

        // 6. RSSI
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ps = ps + __shfl_xor(ps, d, 64);
        if (l == 0)
            a.rssi[(uint64_t)ch * a.n_frames + f] =
                fmaf(ssdr_log2p(fmaxf(ps, 1e-20f)) - 39.0f, SSDR_DB_PER_LOG2, cal);

        // 7. carry: phases advance one frame; the frame tail becomes the FIR history
        phi1 += (uint32_t)SSDR_FRAME * dphi1;
        phi2 += (uint32_t)SSDR_FRAME * dphi2;
        lds_sync();
        if (l < HOCT) {
            load_oct(s_z, NOCT - HOCT + l, B);
            store_oct(s_z, l, B);
        }
        lds_sync();
    }

    // state back to HBM (raw tail of the last frame: lanes 48..63 hold it)
    if (a.n_frames) {
        if (l >= 64 - HOCT) {
            uint4 *hp = reinterpret_cast<uint4 *>(a.hist + (size_t)ch * SSDR_HIST + 8 * (l - (64 - HOCT)));
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

} // namespace

hipError_t ssdr_launch_audio(const SsdrAudioArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(ssdr_audio_kernel, dim3(a.n_ch), dim3(SSDR_AUDIO_BLOCK), 0, stream, a);
    return hipGetLastError();
}

hipError_t ssdr_launch_synth(const SsdrSynthArgs &a, hipStream_t stream)
{
    const uint64_t total = (uint64_t)a.n_ch * a.n_samples;
    const uint32_t grid = (uint32_t)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(ssdr_synth_kernel, dim3(grid ? grid : 1), dim3(256), 0, stream, a);
    return hipGetLastError();
}
