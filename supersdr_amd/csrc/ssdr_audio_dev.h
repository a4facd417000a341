// ssdr_audio_dev.h -- device pieces of the audio chain shared by the audio kernels (ssdr_audio.hip) and the fused
// superframe kernel (ssdr_wf.hip): cross-lane primitives (DPP scans in the order the twin defines), the NCO, the
// demodulators, the AGC / pack / store tail, the per-frame RSSI and ADC-overflow bookkeeping.  All of it works on
// "one wave64 == one receiver channel, lane l owns samples 8l .. 8l+7 of the frame".
#pragma once
#include "ssdr_math.h"
#include "ssdr_kernels.h"

constexpr int OCT = 10;                         // LDS slots (float2) per 8 samples: 8 + 2 pad
constexpr int NOCT = (SSDR_HIST + SSDR_FRAME) / 8;   // 80 octets
constexpr int HOCT = SSDR_HIST / 8;             // 16 history octets
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

SSDR_DEV void lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- cross-lane primitives: DPP, no LDS traffic and no address arithmetic.
// dpp<CTRL, ROWMASK>(identity, x): lanes whose source is out of range or masked off receive `identity`.
template <int CTRL, int ROWMASK>
SSDR_DEV float dpp(float identity, float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(x), CTRL, ROWMASK, 0xF, false));
}
// fmaxf() makes the compiler quiet signalling NaNs first (an extra v_max per operand); none can occur here
SSDR_DEV float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
SSDR_DEV float vmax3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max(a, |b|, |c|): the absolute values are source modifiers, no extra instruction
SSDR_DEV float vmax3_abs(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
SSDR_DEV uint32_t lane63_u(uint32_t x) { return (uint32_t)__builtin_amdgcn_readlane((int)x, 63); }
SSDR_DEV uint32_t from_prev_lane_u(uint32_t lane0_value, uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0_value, (int)x, 0x138, 0xF, 0xF, false);     // wave_shr:1
}
// I*I + Q*Q of one raw sample, exactly, in one instruction (v_dot2_i32_i16; both components -32768 give 2^31, read unsigned)
typedef short s16x2 __attribute__((ext_vector_type(2)));
SSDR_DEV uint32_t iq_power(uint32_t raw)
{
    uint32_t p;                                     // the three-operand form with a literal 0: the compiler's choice, the
    asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(p) : "v"(raw));      // accumulating v_dot2c, needs a v_mov 0 per sample
    return p;
}
SSDR_DEV float lane63(float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63)); }
SSDR_DEV float from_prev_lane(float lane0_value, float x) { return dpp<0x138, 0xF>(lane0_value, x); }   // wave_shr:1

// Inclusive scan over the 64 lanes in six steps: Kogge-Stone inside each row of 16 (row_shr 1,2,4,8),
// then lane 15 of rows 0/2 into rows 1/3 (row_bcast:15), then lane 31 into rows 2,3 (row_bcast:31).
// STEP(dpp control, row mask) is expanded once per step; the twin walks the same six steps.
// Each step is ONE instruction: the DPP operand is the scanned register itself, read from the source lane
// before anything is written; lanes whose source is out of range or masked off are not written and keep
// their value (which is what combining with the identity would give).  "s_nop 1": a VALU write needs two
// wait states before a DPP read of the same register.
#define SSDR_SCAN6(STEP) STEP("row_shr:1", "0xf") STEP("row_shr:2", "0xf") STEP("row_shr:4", "0xf") STEP("row_shr:8", "0xf") \
                         STEP("row_bcast:15", "0xa") STEP("row_bcast:31", "0xc")
#define SSDR_DPP(C, M) " " C " row_mask:" M " bank_mask:0xf"

SSDR_DEV float scan_max(float x)
{
#define STEP(C, M) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0" SSDR_DPP(C, M) : "+v"(x));
    SSDR_SCAN6(STEP)
#undef STEP
    return x;
}
SSDR_DEV float scan_sum(float x)
{
#define STEP(C, M) asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0" SSDR_DPP(C, M) : "+v"(x));
    SSDR_SCAN6(STEP)
#undef STEP
    return x;
}
// affine maps m -> A m + B: (A, B) of a lane := (A, B) of the lane composed after its source's:
// B = fma(A, B_src, B), then A = A * A_src
SSDR_DEV void scan_affine(float &A, float &B)
{
#define STEP(C, M) asm("s_nop 1\n\tv_fmac_f32_dpp %1, %1, %0" SSDR_DPP(C, M) "\n\tv_mul_f32_dpp %0, %0, %0" SSDR_DPP(C, M) : "+v"(A), "+v"(B));
    SSDR_SCAN6(STEP)
#undef STEP
}

// The NCO.  The phasor of sample n = 8 b + j of a frame that starts at phase phi is
//     P(phi) * P(8 b dphi) * S^j,     P(x) = e^{j 2 pi x / 2^32} at all 32 bits (ssdr_phasor32), S = P(dphi):
// mathematically the ideal oscillator e^{j 2 pi (phi + n dphi) / 2^32}; in fp32 a product of three phasors of
// ~1e-7 error each.  P(8 b dphi) depends on the channel only: lane b keeps it for the whole call.  P(phi) is one value per
// frame: lane i evaluates it for frame i of the call (64 frames per polynomial evaluation), the frame loop broadcasts it
// with two v_readlane.  What is left per frame and lane: one complex multiply for the block phasor and a rotation per
// sample -- no polynomial in the frame loop.
struct Nco {
    uint32_t dphi;
    float cs, ss;               // S
    float qc, qs;               // P(8 l dphi) of this lane
    float tc, ts;               // lane i: P(phase of frame (f & ~63) + i)
};
// block = samples per lane (8; 8 D in front of a decimating filter), frame = samples per frame at this NCO's rate
SSDR_DEV void nco_setup(Nco &n, uint32_t dphi, int l, int block = 8)
{
    n.dphi = dphi;
    ssdr_phasor32(dphi, n.cs, n.ss);
    ssdr_phasor32((uint32_t)(block * l) * dphi, n.qc, n.qs);
    n.tc = 1.0f; n.ts = 0.0f;
}
SSDR_DEV void nco_frame_table(Nco &n, uint32_t phase_of_frame, int l, int frame = SSDR_FRAME)       // every 64 frames
{
    ssdr_phasor32(phase_of_frame + (uint32_t)(frame * l) * n.dphi, n.tc, n.ts);
}
SSDR_DEV float lane_f(float x, uint32_t lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (int)lane)); }
// block phasor of this lane for frame f: (frame phasor) * (lane's block offset phasor)
SSDR_DEV void nco_block(const Nco &n, uint32_t f, float &c, float &s)
{
    const float fc = lane_f(n.tc, f & 63u), fs = lane_f(n.ts, f & 63u);
    c = fmaf(fc, n.qc, -(fs * n.qs));
    s = fmaf(fs, n.qc, fc * n.qs);
}
SSDR_DEV void phasor_mul(float ac, float as, float bc, float bs, float &c, float &s)
{
    c = fmaf(ac, bc, -(as * bs));
    s = fmaf(as, bc, ac * bs);
}

// Mix eight samples with the block phasor (c, s) and the step S: x * conj(P S^j).
// CLIP: amax = max(amax, |I|, |Q|) over the block (ADC-overflow detection on the samples as they arrive).
template <bool CLIP, int N = 8>
SSDR_DEV void mix8(const uint32_t (&rw)[N], float c, float s, float cs, float ss, float2 (&z)[N], float &amax)
{
#pragma unroll
    for (int j = 0; j < N; j++) {
        const float xr = (float)(int16_t)(rw[j] & 0xFFFFu);
        const float xi = (float)((int32_t)rw[j] >> 16);
        if (CLIP) amax = vmax3_abs(amax, xr, xi);
        z[j] = make_float2(fmaf(xr, c, xi * s), fmaf(xi, c, -(xr * s)));          // x * (c - j s)
        const float cn = fmaf(c, cs, -(s * ss)), sn = fmaf(s, cs, c * ss);
        c = cn; s = sn;
    }
}

// the same on eight samples with the running phasor handed back: a lane's 8 D inputs in front of a decimating filter are
// mixed eight at a time (the rotation chain is one and the same, so the values are those of mix8<CLIP, 8 D>)
template <bool CLIP>
SSDR_DEV void mix8_carry(const uint32_t *rw, float &c, float &s, float cs, float ss, float2 (&z)[8], float &amax)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float xr = (float)(int16_t)(rw[j] & 0xFFFFu);
        const float xi = (float)((int32_t)rw[j] >> 16);
        if (CLIP) amax = vmax3_abs(amax, xr, xi);
        z[j] = make_float2(fmaf(xr, c, xi * s), fmaf(xi, c, -(xr * s)));
        const float cn = fmaf(c, cs, -(s * ss)), sn = fmaf(s, cs, c * ss);
        c = cn; s = sn;
    }
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

// taps [k0, k0 + NK) of the FIR against the 16-sample register window (A = newer octet, B = older)
template <int K0, int NK>
SSDR_DEV void fir_taps(const float *h, const float2 (&A)[8], const float2 (&B)[8], float (&yr)[8], float (&yi)[8])
{
#pragma unroll
    for (int kk = K0; kk < K0 + NK; kk++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float2 v = (j - kk >= 0) ? A[(j - kk) & 7] : B[(8 + j - kk) & 7];
            yr[j] = fmaf(h[kk], v.x, yr[j]);
            yi[j] = fmaf(h[kk], v.y, yi[j]);
        }
    }
}

// ---- per-frame pieces shared by the three frame paths of the kernel ----------------------------------

struct AgcK { float c0, c1, knee, d8; uint32_t K; };

// AM: envelope minus a one-pole DC estimate.  The envelope is the correctly rounded sqrt of the power; the
// recurrence along time is an affine scan across lanes in a defined order (the twin walks the same six steps).
template <bool INTEGER_POWER>
SSDR_DEV void demod_am(const float (&p)[8], float &dc, float (&aud)[8])
{
    constexpr float DC_APOW[8] = SSDR_DC_APOW_INIT;
    float env[8], loc[8];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        env[j] = INTEGER_POWER ? ssdr_sqrt_rn_int(p[j]) : ssdr_sqrt_rn(p[j]);
        s = fmaf(SSDR_DC_A, s, SSDR_DC_AL * env[j]);
        loc[j] = s;
    }
    float Asc = DC_APOW[7], Bsc = s;                 // this lane's 8 samples as the map m -> A m + B
    scan_affine(Asc, Bsc);
    const float Ae = from_prev_lane(1.0f, Asc), Be = from_prev_lane(0.0f, Bsc);
    const float carry = fmaf(Ae, dc, Be);            // lane 0: identity map -> dc
    float m = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        m = fmaf(DC_APOW[j], carry, loc[j]);
        aud[j] = env[j] - m;
    }
    dc = lane63(m);
}

// SSB / CW product detector: Re{y * e^{+j phi2}}, the second NCO with the same structure as the first
SSDR_DEV void demod_ssb(const float (&yr)[8], const float (&yi)[8], float c, float s, float cs2, float ss2, float (&aud)[8])
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        aud[j] = fmaf(yr[j], c, -(yi[j] * s));       // Re{y * (c + j s)}
        const float cn = fmaf(c, cs2, -(s * ss2)), sn = fmaf(s, cs2, c * ss2);
        c = cn; s = sn;
    }
}

// NBFM discriminator: angle of y[n] * conj(y[n-1])
SSDR_DEV void demod_fm(const float (&yr)[8], const float (&yi)[8], float prev_re, float prev_im, float kfm, float (&aud)[8])
{
    float pr = from_prev_lane(prev_re, yr[7]), pi = from_prev_lane(prev_im, yi[7]);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float dr = fmaf(yr[j], pr, yi[j] * pi);
        const float di = fmaf(yi[j], pr, -(yr[j] * pi));
        aud[j] = ssdr_atan2p(di, dr) * kfm;
        pr = yr[j]; pi = yi[j];
    }
}

SSDR_DEV float block_peak(const float (&p)[8])
{
    return vmax3(vmax3(vmax3(p[0], p[1], p[2]), p[3], p[4]), vmax3(p[5], p[6], p[7]), SSDR_P_FLOOR);
}

// AGC (block peak -> log2 -> (max,+) follower across lanes -> gain), round-half-even, saturate, pack, store
// (pm_known >= 0: the caller already holds max(p[0..7], SSDR_P_FLOOR))
SSDR_DEV float agc_pack_store(const float (&p)[8], const float (&aud)[8], int l, const AgcK &k, float &agc_d,
                              float (&agc_m)[8], int16_t *dst, float pm_known = -1.0f)
{
    // max of the eight powers and the floor in four three-input maxima (max is exact: any grouping gives the same value)
    const float pm = pm_known >= 0.0f ? pm_known : block_peak(p);
    const float al = ssdr_log2p(pm);
    const float fl = (float)l;
    const float d8 = k.d8;
    float e;
    if (k.K == 0) {
        const float P = scan_max(fmaf(fl, d8, al));
        e = vmax(fmaf(-fl, d8, P), fmaf(-(fl + 1.0f), d8, agc_d));
        agc_d = lane63(e);
    } else {
        const float P = scan_max(al);
        float maxM = agc_m[0], mK = agc_m[0];
#pragma unroll
        for (int i = 1; i < 8; i++)
            if ((uint32_t)i < k.K) { maxM = fmaxf(maxM, agc_m[i]); mK = agc_m[i]; }
        e = vmax(vmax(P, maxM), fmaf(-(fl + 1.0f), d8, agc_d));
        agc_d = fmaxf(fmaf(-64.0f, d8, agc_d), mK);
#pragma unroll
        for (int i = 7; i > 0; i--) agc_m[i] = agc_m[i - 1];
        agc_m[0] = lane63(P);
    }
    const float g = ssdr_exp2p(fmaf(k.c1, vmax(e, k.knee), k.c0));
    u32x4 w;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        // v_cvt_i32_f32 saturates, v_cvt_pk_i16_i32 saturates again to int16: same as clamp(rint(y))
        const int i0 = __float2int_rn(aud[j] * g), i1 = __float2int_rn(aud[j + 1] * g);
        w[j >> 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(i0, i1));
    }
    SSDR_NT_STORE(w, reinterpret_cast<u32x4 *>(dst));
    return g;
}

// SSDR_MODE_IQ: the lane's eight filtered samples times the AGC gain as I | Q << 16 (round-half-even, saturating), 32 B per lane
SSDR_DEV void iq_pack_store(const float (&yr)[8], const float (&yi)[8], float g, uint32_t *dst)
{
    u32x4 w0, w1;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t v = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(__float2int_rn(yr[j] * g), __float2int_rn(yi[j] * g)));
        if (j < 4) w0[j] = v; else w1[j - 4] = v;
    }
    SSDR_NT_STORE(w0, reinterpret_cast<u32x4 *>(dst));
    SSDR_NT_STORE(w1, reinterpret_cast<u32x4 *>(dst) + 1);
}

// Per-frame RSSI (sum over the frame = last lane of the inclusive sum scan) and ADC-overflow flag.  Lane (f mod 64)
// keeps both; the conversion to dBm (one log2) and the stores run once per 64 frames (or at the end of the call) for
// all kept frames together, instead of once per frame on a single lane.
SSDR_DEV void rssi_flag_step(const float (&p)[8], bool clip, uint32_t f, uint32_t n_frames, int l, float cal,
                             float &rssi_sum, uint32_t &flag_keep, float *rssi_row, uint8_t *flag_row)
{
    float ps = p[0];
#pragma unroll
    for (int j = 1; j < 8; j++) ps = ps + p[j];
    const float tot = lane63(scan_sum(ps));
    if ((f & 63u) == (uint32_t)l) { rssi_sum = tot; flag_keep = clip ? 1u : 0u; }
    if ((f & 63u) == 63u || f + 1 == n_frames) {
        if ((uint32_t)l <= (f & 63u)) {
            rssi_row[(f & ~63u) + l] = fmaf(ssdr_log2p(fmaxf(rssi_sum, 1e-20f)) - 39.0f, SSDR_DB_PER_LOG2, cal);
            flag_row[(f & ~63u) + l] = (uint8_t)flag_keep;
        }
    }
}

SSDR_DEV bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }

// |component| >= 32767 for any of a lane's eight raw samples (the rare exact check behind the cheap triggers)
SSDR_DEV bool raw_clipped(const uint32_t (&rw)[8])
{
    bool c = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int lo = (int16_t)(rw[j] & 0xFFFFu), hi = (int32_t)rw[j] >> 16;
        c = c || lo >= 32767 || lo <= -32767 || hi >= 32767 || hi <= -32767;
    }
    return c;
}
