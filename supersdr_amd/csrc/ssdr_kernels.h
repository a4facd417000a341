// ssdr_kernels.h -- internal interface between the C-ABI host code and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ssdr.h"

#ifndef SSDR_WF_BLOCK
#define SSDR_WF_BLOCK 256                    // threads per workgroup of the waterfall kernel
#endif
#ifndef SSDR_WF_ABLATE
#define SSDR_WF_ABLATE 0                     // profiling ablations only (1 memory-only, 2 no loads, 3 no stores)
#endif
#ifndef SSDR_WF_PREFETCH
#define SSDR_WF_PREFETCH 0                   // 1: register software pipeline (costs 32 VGPRs)
#endif
#ifndef SSDR_WF_WAVES_PER_EU
#define SSDR_WF_WAVES_PER_EU 3                // register budget: 3 waves/SIMD -> <= 168 VGPRs
#endif
#define SSDR_TW_STAGE_N 992                  // per-stage twiddle table entries: 32*(1+2+4+8+16)
#define SSDR_AUDIO_BLOCK 64                  // one wave == one receiver channel

struct SsdrWfArgs {
    const uint32_t *iq;                      // [n_ch][ch_stride] dwords, each = I | Q << 16
    uint64_t ch_stride;                      // dwords between channels
    uint32_t n_ch, n_lines;                  // lines (1024 samples) in this batch
    uint32_t n_avg, phase;                   // averaging N; lines already summed in `acc`
    uint32_t n_groups;                       // averaging groups touched by this batch
    int16_t *out;                            // [n_complete_groups][n_ch][1024]
    const int16_t *acc_in;                   // [n_ch][1024] partial sums carried in from the previous call
    int16_t *acc_out;                        // [n_ch][1024] partial sums carried out (a different buffer: other
                                             // workgroups may still be reading acc_in)
    const ssdr_chan_consts *consts;          // [n_ch] (wf_cal_lin)
    const float *win;                        // [1024]
    const float2 *tw_stage;                  // [992]
    const float *thr;                        // [256]
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
};

struct SsdrSynthArgs {
    uint32_t *iq;
    uint64_t ch_stride;
    uint32_t n_ch, n_samples;
    uint32_t seed, first_channel_id;
    uint64_t sample0;                        // absolute index of the first sample (phase continuity)
};

hipError_t ssdr_launch_wf(const SsdrWfArgs &a, uint32_t grid, hipStream_t stream);
hipError_t ssdr_wf_blocks_per_cu(int *blocks);
hipError_t ssdr_launch_audio(const SsdrAudioArgs &a, hipStream_t stream);
hipError_t ssdr_launch_synth(const SsdrSynthArgs &a, hipStream_t stream);
hipError_t ssdr_launch_quant_selftest(const float *thr, unsigned long long *mismatch, hipStream_t stream);

// host-side tables and parameter compilation (ssdr_tables.cpp)
void ssdr_make_window(float *win);                    // [1024]
void ssdr_make_twiddles(float *wr, float *wi);        // [512] each
void ssdr_make_tw_stage(float2 *tw);                  // [992]
void ssdr_make_thresholds(float *thr);                // [256]
int ssdr_compile_params_host(const ssdr_chan_params *p, ssdr_chan_consts *c, float *taps);
