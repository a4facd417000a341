// ssdr_chain_ws.hip -- both stages on ONE read of the input for ANY mix of audio frame paths, by WAVE SPECIALISATION inside a
// workgroup (round 6; successor of round 5's ssdr_fused_gen_kernel, which time-shared one wave between the two stages and lost 45 %
// to the register and LDS squeeze: profiles/r05_ab_fused_general.txt; deleted, its source is in the history at 848c201).
//
// Stands where the reference receives W/F lines and SND frames of the same receiver from its server (utils_supersdr.py:780-785,
// 1044-1076); tap formula of the channel filter: utils_supersdr.py:334-344 (ssdr_tables.cpp); what a listener's passband change
// does to the filter: utils_supersdr.py:1078-1092.
//
// A workgroup = NA / 2 TRIOS of two audio waves and one FFT wave.  A trio takes one channel pair of the ctx's chain list at a time,
// for the whole call, drawn from a ticket counter (the FFT wave draws the next pair while the current one runs); the host lays the
// list out with the frame paths interleaved, so that filter-heavy pairs (audio-bound) and full-band pairs (FFT-bound) share a
// SIMD at any time instead of following each other.
//   * an AUDIO wave IS ssdr_audio.hip's kernel for its channel -- ssdr_audio_chan.h:channel_frames<PATH>, carried state in registers
//     for the whole call, the general path's FIR work area at the stand-alone kernel's conflict-free 80-byte lane stride -- and does
//     one thing more: the 32 bytes of raw samples a lane has just loaded for a frame go into the channel's RING (three frames
//     of 2 KB per channel, natural order) as well.  Two frames make a line.
//   * an FFT wave is ssdr_wf.hip's kernel body for a channel pair (one FFT per 32-lane half, ssdr_wf_dev.h), except that a line's
//     samples come out of the two rings (ds_read_b32, lane-consecutive) instead of out of HBM.  It owns its pair for the whole call,
//     so with N > 1 the N-line sums stay in its registers across the group's lines.
//   * hand-shake per channel slot: two words of LDS, `prod` = frames written, `cons` = frames taken (monotonic over the whole kernel).
//     The audio wave publishes a frame after its ring write (release); before it writes into a place again it checks that the FFT
//     wave took what sat there (it did so -- 16 LDS reads per frame -- a frame's worth of audio work earlier unless it has fallen
//     behind).  The FFT wave waits for both producers of a frame, copies it into registers and hands the place back at once.
//     No s_barrier after the tables are loaded; waiting waves sleep (s_sleep) and give the issue port to the others.
//     Progress: a workgroup's twelve waves are resident together; a producer only ever waits for its consumer to have taken the frame
//     RF places back, the consumer only for frames its producers file without waiting on it again -- no cycle; a slot without a channel
//     (odd channel count) publishes its frames at once; the ticket that runs past the list ends a trio.
//   * the audio waves ask for frame f + 1 before they work on frame f (8 more registers; at 3 waves per SIMD nothing else hides
//     the HBM latency of a frame: -4 % time on all-filtering batches).
// The input is read from HBM once; nothing of one stage lives in the other's registers; the instruction count is the two
// kernels' plus the ring writes and the polls.  Results are the two kernels' bit for bit (same code, same scan orders).
// Batches it takes: hop 1024, whole lines, 12 kHz IQ (D = 1), no waterfall zoom, fp32 bins; any mix of modes incl. SSDR_MODE_IQ,
// any filter length, any N.
// MEASURED (profiles/r06_ab_chain_ws.txt): 145 VGPRs, no scratch, 150.7 KB of LDS, 84 % of the SIMDs' quad-cycles issue a vector
// instruction, HBM traffic 1.02 x the fused budget; +2 % against the two stages side by side where every channel filters, a tie on
// all-SSB, -0.6 ... -1.5 % with full-band channels in the batch: both schedules sit at the board's power cap and spend the same joules.
// ssdr_run_chain takes it by default where every channel runs the general path, from 32 768 channels on (ssdr_api.cpp).
#include "ssdr_math.h"
#include "ssdr_kernels.h"
#include "ssdr_audio_dev.h"
#include "ssdr_audio_chan.h"
#include "ssdr_wf_dev.h"

namespace {

constexpr int NA = SSDR_WS_AUDIO_WAVES;                          // audio waves = channels of a group
constexpr int NF = NA / 2;                                       // FFT waves = channel pairs
constexpr int WAVES = NA + NF;
static_assert(NA % 2 == 0 && WAVES * 64 == SSDR_WS_BLOCK, "workgroup shape");
constexpr int A_BYTES = NOCT * OCT * 8 + (SSDR_NTAP_MAX + 8) * 4; // FIR work area 6400 + taps 544 per audio wave
constexpr int RF = 3;                                            // 512-sample frames of raw IQ a channel's ring holds (6 KB; four do not fit the LDS)
constexpr int NAP_SHORT = 1, NAP_LONG = 8, SHORT_LOOKS = 3;      // s_sleep arguments of a waiting wave (units of 64 clocks): its first looks, the later ones
constexpr int FRAME_BYTES = SSDR_FRAME * 4;
constexpr int RING_BYTES = RF * FRAME_BYTES;
constexpr int LDS_FFT = LDS_XCH;                                 // [NF] transposes, 2 x 4224 B each
constexpr int LDS_AUD = LDS_FFT + NF * 2 * XCH_FLOATS * 4;       // [NA] FIR work areas
constexpr int LDS_RING = LDS_AUD + NA * A_BYTES;                 // [NA] raw lines
constexpr int LDS_FLAGS = LDS_RING + NA * RING_BYTES;            // prod[NA], cons[NA], item_seq[NF], item_idx[NF][2]
constexpr int FLAG_WORDS = 2 * NA + 3 * NF;
constexpr int LDS_TOTAL = LDS_FLAGS + FLAG_WORDS * 4;
static_assert(LDS_TOTAL <= 163840, "LDS budget");
static_assert(LDS_AUD % 16 == 0 && A_BYTES % 16 == 0 && LDS_RING % 16 == 0, "alignment");

SSDR_DEV uint32_t flag_read(const uint32_t *p)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// sleep until the counter has reached `need` (counters are monotonic; compared as a signed difference)
SSDR_DEV void flag_wait(const uint32_t *p, uint32_t need)
{
    // (a look costs an LDS read and a vector instruction -- v_readfirstlane -- on a port the other waves want: a few short naps for the
    //  hand-over that is almost there, long ones for a consumer that is a frame ahead of its producers)
    for (uint32_t looks = 0; (int32_t)(flag_read(p) - need) < 0; looks++) {
        if (looks < SHORT_LOOKS) __builtin_amdgcn_s_sleep(NAP_SHORT);
        else __builtin_amdgcn_s_sleep(NAP_LONG);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
SSDR_DEV void flag_publish(uint32_t *p, uint32_t v, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the wave's ring accesses first (DS operations of a wave run in order)
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// what an audio wave does with a frame's raw samples besides using them
struct RingTap {
    static constexpr bool PREFETCH = true;
    unsigned char *ring;                    // the channel's RF frames of raw samples
    uint32_t *prod;
    const uint32_t *cons;
    uint32_t k0;                            // frames this slot had produced before the call's first
    int lane;
    SSDR_DEV void operator()(uint32_t f, const u32x4 &raw0, const u32x4 &raw1) const
    {
        const uint32_t k = k0 + f;
        flag_wait(cons, k + 1 - RF);                             // the frame that sat in this place has been taken
        u32x4 *dst = reinterpret_cast<u32x4 *>(ring + (k % RF) * FRAME_BYTES) + 2 * lane;
        dst[0] = raw0;
        dst[1] = raw1;
        flag_publish(prod, k + 1, lane);
    }
};

SSDR_DEV int dev_audio_path(const ssdr_chan_consts &k)          // ssdr_kernels.h:ssdr_audio_path
{
    if (!(k.fir_flags & SSDR_FIR_DELAY4) || k.mode == SSDR_MODE_IQ) return PATH_GENERAL;
    return k.mode == SSDR_MODE_AM ? PATH_AM_RAW : PATH_DELAY4;
}

template <bool AVG>
__global__ __launch_bounds__(SSDR_WS_BLOCK) void ssdr_chain_ws_kernel(SsdrFusedArgs fa)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_TOTAL];
    const SsdrWfArgs &a = fa.wf;
    const SsdrAudioArgs &u = fa.au;
    uint32_t *prod = reinterpret_cast<uint32_t *>(smem + LDS_FLAGS), *cons = prod + NA;
    uint32_t *item_seq = cons + NA, *item_idx = item_seq + NF;   // per trio: items announced so far; the last two of them
    if (threadIdx.x < FLAG_WORDS) prod[threadIdx.x] = 0u;
    load_tables(smem, a.win, a.tw_stage, a.lut);                 // (ends in the kernel's only __syncthreads)

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_items = (u.list_n + 1) >> 1;                // channel pairs of the list, handed out by a ticket counter
    const uint32_t n_lines = a.n_lines, n_fr = 2 * a.n_lines;    // the hand-shake counts frames

    if (wave < NA) {
        // ------------------------------------------------------------------ audio wave: one side of trio `wave / 2`
        unsigned char *mine = smem + LDS_AUD + wave * A_BYTES;
        float2 *s_z = reinterpret_cast<float2 *>(mine);
        float *s_taps = reinterpret_cast<float *>(mine + NOCT * OCT * 8);
        const int t = wave >> 1;
        uint32_t k0 = 0;
        for (uint32_t n = 0;; n++, k0 += n_fr) {
            flag_wait(item_seq + t, n + 1);
            const uint32_t item = flag_read(item_idx + 2 * t + (n & 1u));
            if (item >= n_items) break;
            const uint32_t idx = 2 * item + (wave & 1u);
            if (idx >= u.list_n) {                               // the list's last pair has one channel: the frames count as delivered
                flag_publish(prod + wave, k0 + n_fr, lane);
                continue;
            }
            const uint32_t ch = u.chan_list[idx];
            const ssdr_chan_consts &kc = u.consts[ch];
            const RingTap tap = {smem + LDS_RING + wave * RING_BYTES, prod + wave, cons + wave, k0, lane};
            const int path = __builtin_amdgcn_readfirstlane(dev_audio_path(kc));
            if (path == PATH_GENERAL) channel_frames<PATH_GENERAL>(u, ch, lane, kc, s_z, s_taps, tap);
            else if (path == PATH_DELAY4) channel_frames<PATH_DELAY4>(u, ch, lane, kc, s_z, s_taps, tap);
            else channel_frames<PATH_AM_RAW>(u, ch, lane, kc, s_z, s_taps, tap);
            lds_sync();                                          // (the next channel's prologue writes the work area again)
        }
        return;
    }

    // ---------------------------------------------------------------------- FFT wave of trio `wave - NA`: draws the trio's pairs
    const int pw = wave - NA;
    const int h = lane >> 5, l = lane & 31;
    float *xch_wave = reinterpret_cast<float *>(smem + LDS_FFT) + pw * 2 * XCH_FLOATS;
    const unsigned char *lut = smem + LDS_LUT0;
    uint32_t *prod_a = prod + 2 * pw, *cons_a = cons + 2 * pw;
    auto draw = [&]() -> uint32_t {                              // (the value is not needed before the first line is out)
        uint32_t v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(fa.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - fa.ticket_base;
        return v;
    };
    auto announce = [&](uint32_t n, uint32_t item) {             // item n of this trio
        if (lane == 0) __hip_atomic_store(item_idx + 2 * pw + (n & 1u), item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        flag_publish(item_seq + pw, n + 1, lane);
    };
    uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)draw());
    announce(0, item);
    uint32_t k0 = 0;
    for (uint32_t n = 0; item < n_items; n++, k0 += n_fr) {
        const uint32_t next_v = draw();                          // the pair after this one: asked for now, announced behind line 0
        const uint32_t ia = 2 * item;
        const bool has_b = ia + 1 < u.list_n;                    // wave-uniform
        const bool ch_ok = h == 0 || has_b;
        const uint32_t ch = u.chan_list[ia + ((h && has_b) ? 1u : 0u)];
        const float calq = a.consts[ch].wf_cal_lin * SSDR_LUT_SCALE;
        const unsigned char *ring = smem + LDS_RING + (2 * pw + ((h && has_b) ? 1 : 0)) * RING_BYTES;
        uint32_t acc[AVG ? 16 : 1];
#pragma unroll
        for (int j = 0; j < (AVG ? 16 : 1); j++) acc[j] = 0;

        for (uint32_t line = 0; line < n_lines; line++) {
            if (SSDR_PRIO_WF) prio_latency_phase();
            uint32_t raw[32];
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {                     // the line's two frames, each as soon as both producers have filed it
                const uint32_t k = k0 + 2 * line + hf;
                flag_wait(prod_a, k + 1);
                if (has_b) flag_wait(prod_a + 1, k + 1);
                const uint32_t *q = reinterpret_cast<const uint32_t *>(ring + (k % RF) * FRAME_BYTES) + opaque(l);
#pragma unroll
                for (int r = 0; r < 16; r++) raw[16 * hf + r] = q[32 * r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the reads have returned: the place is the producers' again
                if (lane < 2) __hip_atomic_store(cons_a + lane, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            SCHED_FENCE();
            if (SSDR_PRIO_WF) prio_compute_phase();
            f32x2 z[32];
            window_line(raw, smem, l, z);
            SCHED_FENCE();
            fft_line<AVG>(z, smem, xch_wave, h, l);
            if (SSDR_PRIO_WF) prio_latency_phase();              // quantiser look-ups, the line's store, the next line's hand-shake
            uint32_t qn[16];
            if (AVG) quantise32(z, calq, lut, [&](int j, uint32_t q01) { acc[j] += q01; });
            else quantise32(z, calq, lut, [&](int j, uint32_t q01) { qn[j] = q01; });
            // AVG: line `line` is line (phase + line) of the stream of groups; a group leaves when its N-th line is in, the last
            // (partial) one of the call goes to acc_out (ssdr_wf.hip:ssdr_fused_am_kernel)
            const uint32_t pos = a.phase + line;
            const bool group_done = !AVG || (pos + 1) % a.n_avg == 0;
            const bool last_line = line + 1 == n_lines;
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
            if (line == 0) announce(n + 1, (uint32_t)__builtin_amdgcn_readfirstlane((int)next_v));
        }
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_v);
    }
}

} // namespace

hipError_t ssdr_launch_chain_ws(const SsdrFusedArgs &a, uint32_t grid, hipStream_t stream)
{
    if (a.wf.n_avg > 1) hipLaunchKernelGGL((ssdr_chain_ws_kernel<true>), dim3(grid), dim3(SSDR_WS_BLOCK), 0, stream, a);
    else hipLaunchKernelGGL((ssdr_chain_ws_kernel<false>), dim3(grid), dim3(SSDR_WS_BLOCK), 0, stream, a);
    return hipGetLastError();
}

hipError_t ssdr_chain_ws_blocks_per_cu(int *blocks)
{
    int b0 = 0, b1 = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b0, ssdr_chain_ws_kernel<false>, SSDR_WS_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b1, ssdr_chain_ws_kernel<true>, SSDR_WS_BLOCK, 0);
    *blocks = b0 < b1 ? b0 : b1;
    return e;
}
