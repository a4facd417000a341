// ssdr_kernels.h -- internal interface between the C-ABI host code and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ssdr.h"

#ifndef SSDR_WF_BLOCK
#define SSDR_WF_BLOCK 512                    // two workgroups per CU: 16 waves = 4 per SIMD, <= 128 VGPRs
#endif
#ifndef SSDR_WF_WAVES_PER_EU
#define SSDR_WF_WAVES_PER_EU 4
#endif
#define SSDR_TW_STAGE_N 992                  // per-stage twiddle table entries: 32*(1+2+4+8+16)
// dB quantiser table: one 32-bit word per quarter-octave segment of [0, 1] (0.75 dB < 1 dB: at most one threshold
// inside a segment), indexed by bits(p') >> (23 - SSDR_LUT_BITS) where p' = p * 2^-48 clamped to [0, 1] (the top
// threshold T[255] = 2^48 lands on 1.0, the clamp rides on the multiply): byte = (bits(p') + word) >> 24, see
// ssdr_wf.hip:quantise.  2 KB.  (A 16-segment-per-octave table indexed by one SDWA op was measured slower: 9x LDS
// bank conflicts, profiles/README.md.)
#define SSDR_LUT_BITS 2
#define SSDR_LUT_SCALE 0x1p-48f
#define SSDR_LUT_SHIFT (23 - SSDR_LUT_BITS)
#define SSDR_LUT_N ((127 << SSDR_LUT_BITS) + 1)                  // bits(1.0) >> SHIFT, + 1
#define SSDR_AUDIO_BLOCK 64                  // one wave == one receiver channel

struct SsdrWfArgs {
    const uint32_t *iq;                      // [n_ch][ch_stride] dwords, each = I | Q << 16
    uint64_t ch_stride;                      // dwords between channels
    uint32_t n_ch, n_lines;                  // lines in this batch (one per 1024 samples; with `tail` one per 512)
    const uint32_t *tail;                    // hop 512: [n_ch][512] the half-line before the batch (null: hop 1024)
    uint32_t n_avg, phase;                   // averaging N; lines already summed in `acc`
    uint32_t n_groups;                       // averaging groups touched by this batch
    uint32_t grp_run;                        // hop 512: consecutive groups of a channel pair one wave works through (>= 1)
    int16_t *out;                            // [n_complete_groups][n_ch][1024]
    const int16_t *acc_in;                   // [n_ch][1024] partial sums carried in from the previous call
    int16_t *acc_out;                        // [n_ch][1024] partial sums carried out (a different buffer: other
                                             // workgroups may still be reading acc_in)
    const ssdr_chan_consts *consts;          // [n_ch] (wf_cal_lin)
    const float *win;                        // [513]  first half of the symmetric window + midpoint
    const float2 *tw_stage;                  // [992]
    const uint32_t *lut;                     // [SSDR_LUT_N] quantiser segment words (ssdr_make_quant_lut)
};

struct SsdrAudioArgs {
    const uint32_t *iq;
    uint64_t ch_stride;
    uint32_t n_ch, n_frames;
    const ssdr_chan_consts *consts;
    const float *taps;                       // [n_ch][128]
    ssdr_chan_state *state;                  // [n_ch]
    uint32_t *hist;                          // [n_ch][128] raw IQ dwords (oldest first)
    int16_t *pcm;                            // [n_ch][n_frames*512]
    float *rssi;                             // [n_ch][n_frames]
    uint8_t *flags;                          // [n_ch][n_frames] ADC overflow per frame
    uint32_t *iq_out;                        // [n_ch][n_frames*512] I | Q << 16 of channels in SSDR_MODE_IQ, or null
    const uint32_t *chan_list;               // channels of this launch (one frame path), list_n of them
    uint32_t list_n;
};
// frame paths of the audio kernel (ssdr_audio.hip): chosen per channel from its compiled constants
enum { SSDR_PATH_GENERAL = 0, SSDR_PATH_DELAY4 = 1, SSDR_PATH_AM_RAW = 2, SSDR_PATH_COUNT = 3 };
static inline int ssdr_audio_path(const ssdr_chan_consts &k)
{
    if (!(k.fir_flags & SSDR_FIR_DELAY4) || k.mode == SSDR_MODE_IQ) return SSDR_PATH_GENERAL;
    return k.mode == SSDR_MODE_AM ? SSDR_PATH_AM_RAW : SSDR_PATH_DELAY4;
}

struct SsdrSynthArgs {
    uint32_t *iq;
    uint64_t ch_stride;
    uint32_t n_ch, n_samples;
    uint32_t seed, first_channel_id;
    uint64_t sample0;                        // absolute index of the first sample (phase continuity)
};

#define SSDR_WIRE_BODY (17 + SSDR_FRAME * 4)  // SND body in IQ mode: 7 B header + 10 B GPS + 512 x (I,Q) int16 BE

// Post-processing works on `n_sel` channels: all of the ctx (sel == null, n_sel == n_ch) or the ones ssdr_set_post_channels named.
// Inputs (waterfall sums, PCM, play_buffer history) are indexed by the channel, display state and outputs by the position in the list.
struct SsdrDb2colArgs {
    const int16_t *wf;                       // [n_lines][n_ch][1024] sums of n_avg byte lines
    uint32_t n_ch, n_lines, n_avg;
    ssdr_db2col_chan *chans;                 // [n_sel] in/out
    float *color;                            // [n_lines][n_sel][1024]
    const uint32_t *sel;                     // [n_sel] channel of every position, or null
    uint32_t n_sel;
};

struct SsdrPlayArgs {
    const int16_t *pcm;                      // [n_ch][n_frames*512]
    uint32_t n_ch, n_frames;
    const ssdr_play_chan *chans;             // [n_ch]
    const double *taps;                      // [33] filtering(KIWI_RATE/2, AUDIO_RATE).h times SAMPLE_RATIO = 4 (exact)
    const double *hist;                      // [n_ch][8] last 8 volume-scaled samples (the non-zero part of old_buffer) before the call
    double *hist_out;                        // [n_ch][8] ... after it (another buffer: frames of a channel run side by side)
    int16_t *out;                            // [n_ch][n_frames*2048][2]   (resampled path: [n_ch][n_frames*1213][2])
    const double *rs_taps;                   // resampled path: [64*21] polyphase taps (ssdr_resample_taps.h)
    int16_t *mono;                           // [n_ch][n_frames*L] the block before the pan, truncated (the recording branch,
                                             // utils_supersdr.py:1139-1140), or null
    const uint32_t *sel;                     // as in SsdrDb2colArgs: chans / out / mono are [n_sel]..., pcm and hist per channel
    uint32_t n_sel;
};

struct SsdrTraceArgs {
    const float *ring;                       // [rows][n_ch][1024] device copy of wf_data's newest rows; row k at slot (head + k) % rows
    uint32_t n_ch, rows, head, t_avg, spectrum_height;
    double *trace;                           // [n_ch][1024]
    int32_t *y;                              // [n_ch][1024] pixel rows (may be null)
};

struct SsdrSmeterArgs {
    ssdr_smeter_chan *chans;                 // [n_ch] in/out
    const float *rssi;                       // [n_ch][n_frames] of the last audio run (used when rssi_in is null)
    const double *rssi_in;                   // [n_ch] or null
    uint32_t n_ch, n_frames;
    double fps;
};

struct SsdrWireArgs {
    const uint8_t *bodies;                   // [n_ch][n_frames][SSDR_WIRE_BODY]
    uint32_t n_ch, n_frames;
    uint32_t *iq;                            // [n_ch][ch_stride] dwords
    uint64_t ch_stride;
    float *rssi;                             // [n_ch][n_frames] or null: 0.1*smeter - 127 of each frame header
    uint32_t *gps;                           // [n_ch][n_frames][4] or null: '<BBII' of the IQ branch (kiwi/client.py:444-445):
                                             // last_gps_solution, dummy, gpssec, gpsnsec; and the header's flags / seq ride along: see ssdr.h
};

struct SsdrGatherArgs {                      // SSDR_FEED_LAZY_OUT: rows of the selected channels -> compact rows
    const int16_t *wf; const int16_t *pcm; const float *rssi; const uint8_t *flags; const float *wire_rssi;     // whole-batch results (wire_rssi may be null)
    int16_t *wf_out; int16_t *pcm_out; float *rssi_out; uint8_t *flags_out; float *wire_rssi_out;                // [..][n_sel]..
    const uint32_t *sel;                     // [n_sel] channel of every position, or null (position == channel)
    uint32_t n_sel, n_ch, n_lines, n_frames;
};
hipError_t ssdr_launch_gather(const SsdrGatherArgs &a, hipStream_t stream);
hipError_t ssdr_launch_db2col(const SsdrDb2colArgs &a, hipStream_t stream);
hipError_t ssdr_launch_play(const SsdrPlayArgs &a, hipStream_t stream);
hipError_t ssdr_launch_play_rs(const SsdrPlayArgs &a, hipStream_t stream);
hipError_t ssdr_launch_iqwire(const SsdrWireArgs &a, hipStream_t stream);
hipError_t ssdr_launch_trace(const SsdrTraceArgs &a, hipStream_t stream);
hipError_t ssdr_launch_smeter(const SsdrSmeterArgs &a, hipStream_t stream);
hipError_t ssdr_launch_checksum(const void *data, uint64_t n_words, unsigned long long *out, hipStream_t stream);
hipError_t ssdr_launch_adpcm(const uint8_t *data, uint32_t n_streams, uint32_t n_bytes, int32_t *state, int16_t *out,
                             hipStream_t stream);
#define SSDR_ZOOM_HIST 256                   // raw input samples carried per channel (>= 32 Z - 2 for Z <= 8)
#define SSDR_ZOOM_TAPS_MAX 255
struct SsdrZoomArgs {
    const uint32_t *iq;                      // [n_ch][ch_stride] input dwords
    uint64_t ch_stride;
    uint32_t n_ch, n_in, zoom, ntap;         // n_in input samples per channel in this call (multiple of zoom)
    const float *taps;                       // [ntap]
    const uint32_t *dphi;                    // [n_ch] NCO step of the zoom centre
    uint32_t *phase;                         // [n_ch] in/out: phase of the call's first sample
    uint32_t *hist;                          // [n_ch][SSDR_ZOOM_HIST] in/out: the raw samples before the call's first
    uint32_t *out;                           // [n_ch][n_in / zoom] I | Q << 16
};
hipError_t ssdr_launch_zoom(const SsdrZoomArgs &a, hipStream_t stream);
struct SsdrFusedArgs { SsdrWfArgs wf; SsdrAudioArgs au; uint32_t *ticket; uint32_t ticket_base; };
// ticket: ssdr_chain_ws_kernel's pair counter; it stands at ticket_base at launch and is never reset: every trio draws its pairs and one ticket
// beyond the last pair, so a launch of `grid` workgroups leaves it at ticket_base + pairs + grid * SSDR_WS_AUDIO_WAVES / 2 (ssdr_api.cpp)
// the wave-specialised chain kernel (ssdr_chain_ws.hip): a workgroup of SSDR_WS_AUDIO_WAVES audio waves (one receiver each) and half as
// many FFT waves (one channel pair each); one workgroup per CU.  (The kernel's tuning knobs -- ring depth, prefetch, poll naps, priorities -- are
// constants in ssdr_chain_ws.hip; the A/B trail of each is in profiles/r06_ab_chain_ws.txt, the switchable version in
// tools/experiments/ssdr_chain_ws_knobs.patch.)
#define SSDR_WS_AUDIO_WAVES 8
#define SSDR_WS_BLOCK (64 * (SSDR_WS_AUDIO_WAVES + SSDR_WS_AUDIO_WAVES / 2))
hipError_t ssdr_launch_chain_ws(const SsdrFusedArgs &a, uint32_t grid, hipStream_t stream);
hipError_t ssdr_chain_ws_blocks_per_cu(int *blocks);
hipError_t ssdr_launch_fused_am(const SsdrFusedArgs &a, uint32_t grid, hipStream_t stream);
hipError_t ssdr_fused_blocks_per_cu(int *blocks);
hipError_t ssdr_launch_fused_exact_am(const SsdrFusedArgs &a, const double2 *tw, hipStream_t stream);   // ssdr_wf_exact.hip: float64 bins; chooses its grid
hipError_t ssdr_launch_wf(const SsdrWfArgs &a, uint32_t grid, hipStream_t stream);
// float64 waterfall stage (ssdr_wf_exact.hip): twiddle tables of FFT stages 5..10, double2 entries (ssdr_make_tw64):
//   T5[lo4] 16 | T6[q][lo4] 32 | T7[q][lo4] 64 | T8[q][lo4] 128 | T9[m][b4][lo4] 256 | T10[mm][b4][b5][lo4] 256
#define SSDR_TW64_N 752
hipError_t ssdr_launch_wf_exact(const SsdrWfArgs &a, const double2 *tw, hipStream_t stream);   // ssdr_wf_exact.hip; chooses grid and a.grp_run itself
hipError_t ssdr_wf_blocks_per_cu(int *blocks);
hipError_t ssdr_launch_audio(const SsdrAudioArgs &a, int path, hipStream_t stream);
hipError_t ssdr_launch_audio_dec(const SsdrAudioArgs &a, uint32_t decim, hipStream_t stream);
hipError_t ssdr_launch_synth(const SsdrSynthArgs &a, hipStream_t stream);
hipError_t ssdr_launch_sqrt_selftest(unsigned long long *mismatch, hipStream_t stream);
hipError_t ssdr_launch_sqrt_values(const float *in, float *out_scaled, float *out_int, uint32_t n, hipStream_t stream);
hipError_t ssdr_launch_quant_selftest(const float *thr, const uint32_t *lut, unsigned long long *mismatch, hipStream_t stream);

// host-side tables and parameter compilation (ssdr_tables.cpp)
void ssdr_make_window(float *win);                    // [1024]
void ssdr_make_twiddles(float *wr, float *wi);        // [512] each
void ssdr_make_tw_stage(float2 *tw);                  // [992]
void ssdr_make_thresholds(float *thr);                // [256]
void ssdr_make_tw64(double *tw);                      // [SSDR_TW64_N][2] (re, im)
int ssdr_make_quant_lut(uint32_t *lut);               // [SSDR_LUT_N]; returns 0, or -1 if a segment held two thresholds
int ssdr_compile_params_host(const ssdr_chan_params *p, ssdr_chan_consts *c, float *taps, uint32_t decim, uint32_t rate_hz = SSDR_RATE);
int ssdr_design_lowpass(double fl, double fs, int n_max, double *h);   // utils_supersdr.py:334-344; returns tap count
int ssdr_design_lowpass_exact(double fl, double fs, int n, double *h);  // the same window and sinc with exactly n (odd) taps
