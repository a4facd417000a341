// ssdr_api.cpp -- C-ABI of libssdr.so (see include/ssdr.h).  Host side only: owns the
// device buffers, the per-channel state and the stream; launches the HIP kernels.
// Never throws, never aborts: every failure is a negative return code (SSDR_GUARD / SSDR_UNGUARD below).
#include "ssdr_kernels.h"
#include "ssdr_resample_taps.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

static thread_local char g_hip_err[256] = "";

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", #expr, hipGetErrorString(e_)); \
            return e_ == hipErrorOutOfMemory ? SSDR_ENOMEM : SSDR_EHIP;                 \
        }                                                                               \
    } while (0)

// "Never throws": every `int` entry point is a function-try-block.  A host allocation that fails (std::vector, std::bad_alloc)
// or anything else thrown below the C boundary comes back as a return code, like the reference's own policy of turning errors into
// a flag the caller polls (utils_supersdr.py:1031-1036) -- the maintainer's process is never terminated from inside the library.
static int ssdr_caught(bool nomem) noexcept
{
    snprintf(g_hip_err, sizeof g_hip_err, nomem ? "host allocation failed (std::bad_alloc)" : "C++ exception below the C boundary");
    return nomem ? SSDR_ENOMEM : SSDR_EHIP;
}
#define SSDR_GUARD try
#define SSDR_UNGUARD                                               \
    catch (const std::bad_alloc &) { return ssdr_caught(true); }   \
    catch (...) { return ssdr_caught(false); }

struct ssdr_ctx {
    int device = 0;
    uint32_t n_ch = 0;
    uint32_t n_avg = 1, wf_phase = 0;
    uint32_t decim = 1;                                 // D: the IQ arrives at D * 12 kHz, the audio chain decimates to 12 kHz
    std::vector<ssdr_chan_params> h_params;             // the parameters the channels were last given (recompiled when D changes)
    uint32_t hop = SSDR_NFFT;                           // samples between waterfall lines: 1024, or 512 (lines overlap by half)
    uint32_t *d_wf_tail = nullptr;                      // hop 512: [n_ch][512] the last half-line of the previous batch
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t stream2 = nullptr;                      // audio kernel when running concurrently with the waterfall
    hipEvent_t ev_in = nullptr, ev_a = nullptr;
    bool concurrent = false, audio_pending = false;
    // tables
    float *d_win = nullptr, *d_thr = nullptr;
    bool exact_bins = false;                            // ssdr_set_exact_bins: the waterfall stage in float64
    // zoom stage in front of the waterfall kernel (ssdr_set_wf_zoom / ssdr_set_wf_center)
    uint32_t zoom = 1, zoom_ntap = 0;
    std::vector<double> h_zoom_offset;                  // [n_ch] zoom centre, Hz from the IQ band's centre
    float *d_zoom_taps = nullptr;
    uint32_t *d_zoom_dphi = nullptr, *d_zoom_phase = nullptr, *d_zoom_hist = nullptr, *d_zoom_out = nullptr;
    size_t zoom_out_samples = 0;                        // capacity of d_zoom_out per channel
    uint32_t zoom_run_samples = 0;                      // zoomed samples per channel of the last ssdr_run_wf
    double2 *d_tw64 = nullptr;                          // [SSDR_TW64_N] stage twiddles of the float64 waterfall kernel (ssdr_make_tw64)
    float2 *d_tw = nullptr;
    uint32_t *d_lut = nullptr;
    // per-channel
    ssdr_chan_consts *d_consts = nullptr;
    float *d_taps = nullptr;
    ssdr_chan_state *d_state = nullptr;
    uint32_t *d_hist = nullptr;
    std::vector<ssdr_chan_consts> h_consts;             // host mirror of d_consts
    uint32_t *d_chan_list = nullptr;                    // channels sorted by audio frame path (ssdr_audio_path)
    uint32_t *d_ws_list = nullptr;                      // [n_ch] + 1 ticket word: the same channels as pairs, the paths interleaved (ssdr_chain_ws_kernel)
    uint32_t ws_ticket = 0;                             // where the ticket word stands (it only counts up: SsdrFusedArgs)
    uint32_t path_off[SSDR_PATH_COUNT] = {}, path_n[SSDR_PATH_COUNT] = {};
    bool chan_list_dirty = true;
    bool summary_dirty = true;                          // path counts / any channel in IQ mode: recounted after the constants change
    uint32_t sum_paths[SSDR_PATH_COUNT] = {0, 0, 0};
    bool sum_any_iq = false;
    hipStream_t path_stream[SSDR_PATH_COUNT - 1] = {};  // the audio kernels of different paths run side by side
    hipEvent_t ev_fork = nullptr, ev_path[SSDR_PATH_COUNT - 1] = {};
    int fused_enabled = 1;                              // ssdr_set_fused: 0 never, 1 at hop 1024 (default), 2 at hop 512 as well, 3 + the wave-specialised kernel
    bool fuse_ws_next = false;                          // ... and that kernel is ssdr_chain_ws_kernel (any mix of audio paths)
    uint32_t ws_grid = 0;
    uint32_t am_floor = 0, ws_floor = 0;                // ssdr_set_chain_floors: fewest channels for which ssdr_run_chain's default takes a one-read kernel
    bool overlap_enabled = true;                        // ssdr_set_overlap: un-fused ssdr_run_chain batches run the audio stage beside the waterfall kernel
    bool fuse_next = false;                             // ssdr_run_chain: run_wf parks its arguments, run_audio launches the fused kernel
    SsdrWfArgs fused_wf;
    uint32_t fused_grid = 0;
    bool audio_serial = false;                          // measurement: one path kernel after the other on one stream
    int16_t *d_wf_acc[2] = {nullptr, nullptr};          // ping-pong: carry-in / carry-out of partial groups
    int wf_acc_cur = 0;
    // input batch
    uint32_t *d_iq_own = nullptr;
    size_t iq_own_frames = 0;
    const uint32_t *d_iq = nullptr;
    uint32_t in_frames = 0;
    bool have_input = false;
    bool audio_started = false;              // an audio kernel has run since create / full reset: set_params leaves the state alone
    uint64_t synth_sample0 = 0;
    // outputs
    int16_t *d_wf_out = nullptr;
    size_t wf_out_lines = 0;
    uint32_t wf_lines_ready = 0;
    int16_t *d_pcm = nullptr;
    float *d_rssi = nullptr;
    size_t audio_frames = 0;
    uint32_t *d_iq_out = nullptr;             // [n_ch][n_frames*512] I | Q << 16 of the channels in SSDR_MODE_IQ (allocated when one exists)
    size_t iq_out_frames = 0;
    bool iq_out_valid = false;
    uint8_t *d_flags = nullptr;               // ADC-overflow flag per frame of the last audio run
    size_t flags_frames = 0;
    uint32_t audio_run_frames = 0;            // frames the last audio run (or ssdr_set_pcm) produced: extent and stride of d_pcm / d_rssi
    // pipelined host feed (ssdr_feed_*): slots of pinned host memory + their own device buffers
    struct FeedSlot {
        void *h_in = nullptr;                            // int16 IQ, or SND bodies in wire mode
        int16_t *h_wf = nullptr, *h_pcm = nullptr;
        float *h_rssi = nullptr, *h_wire_rssi = nullptr;
        uint8_t *d_wire = nullptr;
        float *d_wire_rssi = nullptr;
        uint32_t *d_in = nullptr;
        int16_t *d_wf = nullptr, *d_pcm = nullptr;
        float *d_rssi = nullptr;
        hipEvent_t ev_in = nullptr, ev_run = nullptr, ev_out = nullptr;
        uint32_t lines = 0;
        uint32_t n_avg = 1;                              // averaging N in force when the batch was submitted
        uint8_t *d_flags = nullptr, *h_flags = nullptr;  // ADC-overflow flag per frame of this batch
        // SSDR_FEED_POST: spectrum_db2col / play_buffer of this batch
        float *d_color = nullptr, *h_color = nullptr;
        ssdr_db2col_chan *d_dbchan = nullptr, *h_dbchan = nullptr;
        ssdr_play_chan *h_playchan = nullptr;
        int16_t *d_play = nullptr, *h_play = nullptr, *d_mono = nullptr, *h_mono = nullptr;
        bool has_mono = false;
        uint32_t n_post = 0;                             // channels the batch was post-processed for (ssdr_set_post_channels at submit)
        // SSDR_FEED_LAZY_OUT: compact rows of the selected channels (what is copied back); the whole-batch d_wf / d_pcm / ... stay on the device
        int16_t *d_sel_wf = nullptr, *d_sel_pcm = nullptr;
        float *d_sel_rssi = nullptr, *d_sel_wire_rssi = nullptr;
        uint8_t *d_sel_flags = nullptr;
        uint32_t n_sel = 0;
    };
    std::vector<FeedSlot> feed;
    uint32_t feed_frames = 0, feed_head = 0, feed_tail = 0, feed_inflight = 0;
    bool feed_taken = false;                             // slot at feed_head handed to the caller, not yet submitted
    bool feed_wire = false;                              // slots hold SND bodies (kiwi/client.py:443-454), unpacked on the device
    bool feed_post = false;                              // SSDR_FEED_POST: db2col + play_buffer in the slot pipeline
    bool feed_lazy = false;                              // SSDR_FEED_LAZY_OUT: only the selected channels' results are copied back
    uint32_t feed_lazy_max = 0;                          // rows the compact buffers hold
    std::vector<ssdr_db2col_chan> feed_dbchan;           // display state for the next submits (ssdr_feed_post)
    std::vector<ssdr_play_chan> feed_playchan;
    int feed_last = -1;                                  // slot ssdr_feed_collect returned last
    int16_t *d_line1 = nullptr;                          // ssdr_db2col_line: one line, its display state, its colours
    ssdr_db2col_chan *d_dbchan1 = nullptr;
    float *d_color1 = nullptr;
    hipStream_t feed_s_in = nullptr, feed_s_out = nullptr;
    // measurement
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;            // last launch (ssdr_elapsed_ms)
    struct Pending { hipEvent_t e0, e1; int which; };
    std::vector<Pending> pending;                       // profiling: resolved lazily, no sync per launch
    std::vector<hipEvent_t> free_events;
    float k_ms[SSDR_K_COUNT] = {};
    uint32_t k_n[SSDR_K_COUNT] = {};
    float last_ms = 0.0f;
    uint32_t wf_grid = 0, wf_grid_1 = 0;
    unsigned long long *d_scratch = nullptr;
    // post-processing
    uint32_t *d_post_sel = nullptr;         // ssdr_set_post_channels: the channels the post kernels work on (null: all)
    uint32_t n_post = 0;                    // their number (n_ch when no selection is set)
    ssdr_db2col_chan *d_db2col = nullptr;
    float *d_color = nullptr;
    size_t color_lines = 0;
    ssdr_play_chan *d_play = nullptr;
    double *d_play_taps = nullptr, *d_play_hist = nullptr, *d_play_rs_taps = nullptr;
    double *d_play_hist_alt = nullptr;      // the kernel reads d_play_hist and writes this one; swapped after every launch
    float *d_wfdata = nullptr;              // [wfdata_rows][n_ch][1024] newest rows of wf_data, row k at slot (head + k) % rows
    float *d_wfpend = nullptr;              // [3][n_ch][1024] wf_data_tmp: deque(maxlen = wf_buffer_len = 3) in front of it
    uint32_t wfdata_rows = 0, wfdata_head = 0, wfpend_n = 0;
    uint64_t wfdata_seen = 0, wfpend_first = 0;     // run_index of kiwi_waterfall.run; arrival index of the oldest queued line
    double *d_trace = nullptr;
    int32_t *d_trace_y = nullptr;
    ssdr_smeter_chan *d_smeter = nullptr;
    double *d_smeter_in = nullptr;
    uint32_t kiwi_rate = SSDR_RATE;         // kiwi_sound.KIWI_RATE: 12000, or 20250 (fractional SAMPLE_RATIO path)
    int16_t *d_play_out = nullptr;
    size_t play_frames = 0;
    std::vector<double> pending_play_hist;  // ssdr_checkpoint_load before the first ssdr_run_playbuffer
    bool recording = false;                 // audio_rec.recording_flag: play_buffer also keeps the mono block (:1139-1140)
    int16_t *d_play_mono = nullptr;
    size_t play_mono_frames = 0;
    uint32_t play_run_frames = 0, play_run_len = 0;
    uint8_t *d_wire = nullptr;
    size_t wire_frames = 0;
    float *d_wire_rssi = nullptr;
    uint32_t *d_wire_gps = nullptr;
    uint32_t wire_run_frames = 0;
};

static int get_event(ssdr_ctx *c, hipEvent_t *e)
{
    if (!c->free_events.empty()) { *e = c->free_events.back(); c->free_events.pop_back(); return SSDR_OK; }
    HIP_TRY(hipEventCreate(e));
    return SSDR_OK;
}
// HIP events on the stream the kernel is launched on, bracketing exactly one launch.
static int timed_begin(ssdr_ctx *c, hipStream_t s = nullptr)
{
    if (!s) s = c->stream;
    if (c->profiling) {
        ssdr_ctx::Pending p{nullptr, nullptr, -1};
        int rc;
        if ((rc = get_event(c, &p.e0)) != SSDR_OK) return rc;
        if ((rc = get_event(c, &p.e1)) != SSDR_OK) return rc;
        c->pending.push_back(p);
        HIP_TRY(hipEventRecord(p.e0, s));
    } else {
        HIP_TRY(hipEventRecord(c->ev0, s));
    }
    return SSDR_OK;
}
static int timed_end(ssdr_ctx *c, int which, hipStream_t s = nullptr)
{
    if (!s) s = c->stream;
    if (c->profiling) {
        c->pending.back().which = which;
        HIP_TRY(hipEventRecord(c->pending.back().e1, s));
    } else {
        HIP_TRY(hipEventRecord(c->ev1, s));
    }
    return SSDR_OK;
}
static int resolve_pending(ssdr_ctx *c)
{
    for (auto &p : c->pending) {
        float ms = 0.0f;
        HIP_TRY(hipEventSynchronize(p.e1));
        HIP_TRY(hipEventElapsedTime(&ms, p.e0, p.e1));
        if (p.which >= 0) { c->k_ms[p.which] += ms; c->k_n[p.which] += 1; c->last_ms = ms; }
        c->free_events.push_back(p.e0);
        c->free_events.push_back(p.e1);
    }
    c->pending.clear();
    return SSDR_OK;
}

extern "C" {

const char *ssdr_version(void) { return "supersdr_amd 0.1 (gfx950)"; }
const char *ssdr_last_hip_error(void) { return g_hip_err; }

const char *ssdr_strerror(int code)
{
    switch (code) {
    case SSDR_OK: return "ok";
    case SSDR_EINVAL: return "invalid argument";
    case SSDR_ENOMEM: return "out of memory";
    case SSDR_EHIP: return "HIP runtime error";
    case SSDR_ENODEV: return "no such GPU device";
    case SSDR_ESTATE: return "call out of order";
    default: return "unknown error";
    }
}

void ssdr_destroy(ssdr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)ssdr_feed_close(c);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    void *ptrs[] = {c->d_win, c->d_thr, c->d_tw, c->d_lut, c->d_consts, c->d_taps, c->d_state, c->d_hist, c->d_chan_list, c->d_ws_list, c->d_wf_tail, c->d_wf_acc[0], c->d_wf_acc[1],
                    c->d_iq_own, c->d_wf_out, c->d_pcm, c->d_rssi, c->d_flags, c->d_scratch, c->d_db2col, c->d_color, c->d_play,
                    c->d_play_taps, c->d_play_hist, c->d_play_hist_alt, c->d_play_rs_taps, c->d_play_out, c->d_wfdata, c->d_wfpend, c->d_trace, c->d_trace_y, c->d_smeter,
                    c->d_smeter_in, c->d_post_sel, c->d_wire, c->d_wire_rssi, c->d_play_mono, c->d_line1, c->d_dbchan1, c->d_color1, c->d_tw64, c->d_wire_gps, c->d_iq_out, c->d_zoom_taps, c->d_zoom_dphi, c->d_zoom_phase, c->d_zoom_hist, c->d_zoom_out};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto &p : c->pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int i = 0; i < SSDR_PATH_COUNT - 1; i++) {
        if (c->path_stream[i]) { (void)hipStreamSynchronize(c->path_stream[i]); (void)hipStreamDestroy(c->path_stream[i]); }
        if (c->ev_path[i]) (void)hipEventDestroy(c->ev_path[i]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int ssdr_default_params(int mode, ssdr_chan_params *p) SSDR_GUARD
{
    if (!p || mode < SSDR_MODE_AM || mode > SSDR_MODE_IQ) return SSDR_EINVAL;
    memset(p, 0, sizeof *p);
    p->mode = mode;
    p->agc_on = 1;                    // utils_supersdr.py:937
    p->agc_hang = 0;                  // :938
    p->agc_thresh = -80.0;            // :939
    p->agc_slope = 0.0;               // :940
    p->agc_decay = (mode == SSDR_MODE_CW) ? 1000.0 : 4000.0;   // :941-942
    p->agc_man_gain = 50.0;           // :943
    p->wf_cal_db = 0.0;
    p->smeter_cal_db = -13.0;         // :790
    switch (mode) {                   // passbands: utils_supersdr.py:46-50, kiwi/client.py:217-249
    case SSDR_MODE_AM: p->low_cut = -6000.0; p->high_cut = 6000.0; break;
    case SSDR_MODE_LSB: p->low_cut = -3000.0; p->high_cut = -30.0; break;
    case SSDR_MODE_USB: p->low_cut = 30.0; p->high_cut = 3000.0; break;
    case SSDR_MODE_CW: p->low_cut = 400.0; p->high_cut = 800.0; break;
    case SSDR_MODE_IQ: p->low_cut = -5000.0; p->high_cut = 5000.0; break;       // kiwi/client.py:244-246
    default: p->low_cut = -6000.0; p->high_cut = 6000.0; break;
    }
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_compile_params(const ssdr_chan_params *p, ssdr_chan_consts *consts, float *taps) SSDR_GUARD
{
    return ssdr_compile_params_host(p, consts, taps, 1);
} SSDR_UNGUARD

int ssdr_table(int which, float *out, uint32_t n) SSDR_GUARD
{
    if (!out) return SSDR_EINVAL;
    float wr[512], wi[512];
    switch (which) {
    case SSDR_T_WINDOW:
        if (n != SSDR_NFFT) return SSDR_EINVAL;
        ssdr_make_window(out);
        return SSDR_OK;
    case SSDR_T_TWIDDLE_RE:
    case SSDR_T_TWIDDLE_IM:
        if (n != 512) return SSDR_EINVAL;
        ssdr_make_twiddles(wr, wi);
        memcpy(out, which == SSDR_T_TWIDDLE_RE ? wr : wi, sizeof wr);
        return SSDR_OK;
    case SSDR_T_DB_THRESH:
        if (n != 256) return SSDR_EINVAL;
        ssdr_make_thresholds(out);
        return SSDR_OK;
    default: return SSDR_EINVAL;
    }
} SSDR_UNGUARD

static int zoom_restart(ssdr_ctx *c, uint32_t first, uint32_t count, bool restart_group = true);
static int join_audio(ssdr_ctx *c);
static int drain_audio(ssdr_ctx *c);

int ssdr_reset_state(ssdr_ctx *c, uint32_t first, uint32_t count) SSDR_GUARD
{
    if (!c || (uint64_t)first + count > c->n_ch) return SSDR_EINVAL;
    if (!count) return SSDR_OK;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    std::vector<ssdr_chan_consts> k(count);
    HIP_TRY(hipMemcpyAsync(k.data(), c->d_consts + first, count * sizeof(ssdr_chan_consts), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<ssdr_chan_state> st(count);
    for (uint32_t i = 0; i < count; i++) {
        memset(&st[i], 0, sizeof st[i]);
        st[i].agc_d = k[i].agc_knee;          // envelope starts at the knee: full gain, no pop
        for (int j = 0; j < 8; j++) st[i].agc_m[j] = -1000.0f;
    }
    HIP_TRY(hipMemcpyAsync(c->d_state + first, st.data(), count * sizeof(ssdr_chan_state), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_hist + (size_t)first * SSDR_HIST, 0, (size_t)count * SSDR_HIST * 4, c->stream));
    for (int i = 0; i < 2; i++)
        HIP_TRY(hipMemsetAsync(c->d_wf_acc[i] + (size_t)first * SSDR_NFFT, 0, (size_t)count * SSDR_NFFT * 2, c->stream));
    if (c->d_wf_tail)        // hop 512: the half-line before the stream is silence again (also after a change of the input rate)
        HIP_TRY(hipMemsetAsync(c->d_wf_tail + (size_t)first * (SSDR_NFFT / 2), 0, (size_t)count * (SSDR_NFFT / 2) * 4, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (first == 0 && count == c->n_ch) { c->wf_phase = 0; c->synth_sample0 = 0; c->audio_started = false; }
    return zoom_restart(c, first, count, false);         // the zoomed streams of these channels start over as well
} SSDR_UNGUARD

int ssdr_set_params(ssdr_ctx *c, uint32_t first, uint32_t count, const ssdr_chan_params *p) SSDR_GUARD
{
    if (!c || !p || (uint64_t)first + count > c->n_ch) return SSDR_EINVAL;
    if (!count) return SSDR_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<ssdr_chan_consts> k(count);
    std::vector<float> taps((size_t)count * SSDR_NTAP_MAX);
    for (uint32_t i = 0; i < count; i++) {
        const int rc = ssdr_compile_params_host(p + i, &k[i], taps.data() + (size_t)i * SSDR_NTAP_MAX, c->decim, c->kiwi_rate);
        if (rc != SSDR_OK) return rc;
    }
    for (uint32_t i = 0; i < count; i++) c->h_params[first + i] = p[i];
    for (uint32_t i = 0; i < count; i++) c->h_consts[first + i] = k[i];
    c->chan_list_dirty = true;
    c->summary_dirty = true;
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    HIP_TRY(hipMemcpyAsync(c->d_consts + first, k.data(), count * sizeof(ssdr_chan_consts), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_taps + (size_t)first * SSDR_NTAP_MAX, taps.data(), taps.size() * sizeof(float),
                           hipMemcpyHostToDevice, c->stream));
    std::vector<ssdr_chan_state> st;
    if (!c->audio_started) {                 // a channel that has not produced audio yet starts at ITS knee (full gain, no pop)
        st.resize(count);
        for (uint32_t i = 0; i < count; i++) {
            memset(&st[i], 0, sizeof st[i]);
            st[i].agc_d = k[i].agc_knee;
            for (int j = 0; j < 8; j++) st[i].agc_m[j] = -1000.0f;
        }
        HIP_TRY(hipMemcpyAsync(c->d_state + first, st.data(), count * sizeof(ssdr_chan_state), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_create(int device_id, uint32_t n_channels, uint32_t nfft, uint32_t frame, ssdr_ctx **out) SSDR_GUARD
{
    if (!out) return SSDR_EINVAL;
    *out = nullptr;
    if (nfft != SSDR_NFFT || frame != SSDR_FRAME || n_channels == 0) return SSDR_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) {
        snprintf(g_hip_err, sizeof g_hip_err, "hipGetDeviceCount: %d device(s), asked for %d", ndev, device_id);
        return SSDR_ENODEV;
    }
    ssdr_ctx *c = new (std::nothrow) ssdr_ctx;
    if (!c) return SSDR_ENOMEM;
    c->device = device_id;
    c->n_ch = n_channels;
    c->n_post = n_channels;
    struct Owner { ssdr_ctx *c; ~Owner() { if (c) ssdr_destroy(c); } } owner{c};     // an error return or an exception below frees the half-built ctx
    int rc = [&]() -> int {
        HIP_TRY(hipSetDevice(device_id));
        HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        HIP_TRY(hipEventCreate(&c->ev0));
        HIP_TRY(hipEventCreate(&c->ev1));
        HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming));
        HIP_TRY(hipMalloc(&c->d_win, SSDR_NFFT * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_thr, 256 * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_tw, SSDR_TW_STAGE_N * sizeof(float2)));
        HIP_TRY(hipMalloc(&c->d_lut, SSDR_LUT_N * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_consts, (size_t)n_channels * sizeof(ssdr_chan_consts)));
        HIP_TRY(hipMalloc(&c->d_taps, (size_t)n_channels * SSDR_NTAP_MAX * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_state, (size_t)n_channels * sizeof(ssdr_chan_state)));
        HIP_TRY(hipMalloc(&c->d_chan_list, (size_t)n_channels * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_ws_list, ((size_t)n_channels + 1) * sizeof(uint32_t)));
        c->h_consts.resize(n_channels);
        c->h_params.resize(n_channels);
        HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < SSDR_PATH_COUNT - 1; i++) {
            HIP_TRY(hipStreamCreateWithFlags(&c->path_stream[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_path[i], hipEventDisableTiming));
        }
        HIP_TRY(hipMalloc(&c->d_hist, (size_t)n_channels * SSDR_HIST * 4));
        HIP_TRY(hipMalloc(&c->d_wf_acc[0], (size_t)n_channels * SSDR_NFFT * 2));
        HIP_TRY(hipMalloc(&c->d_wf_acc[1], (size_t)n_channels * SSDR_NFFT * 2));
        HIP_TRY(hipMalloc(&c->d_scratch, 64));
        std::vector<float> win(SSDR_NFFT), thr(256);
        std::vector<float2> tw(SSDR_TW_STAGE_N);
        ssdr_make_window(win.data());
        ssdr_make_thresholds(thr.data());
        ssdr_make_tw_stage(tw.data());
        HIP_TRY(hipMemcpy(c->d_win, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_thr, thr.data(), thr.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
        std::vector<uint32_t> lut(SSDR_LUT_N);
        if (ssdr_make_quant_lut(lut.data()) != 0) return SSDR_EINVAL;
        HIP_TRY(hipMemcpy(c->d_lut, lut.data(), lut.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_id));
        int per_cu = 0;
        HIP_TRY(ssdr_wf_blocks_per_cu(&per_cu));
        if (per_cu < 1) per_cu = 1;
        // persistent grid == exactly the resident workgroups (a larger grid would run a ragged second round)
        c->wf_grid = (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
        c->wf_grid_1 = (uint32_t)prop.multiProcessorCount;        // one workgroup per CU (concurrent mode)
        int fused_per_cu = 0;
        HIP_TRY(ssdr_fused_blocks_per_cu(&fused_per_cu));
        c->fused_grid = (uint32_t)prop.multiProcessorCount * (uint32_t)(fused_per_cu < 1 ? 1 : fused_per_cu);
        int ws_per_cu = 0;
        HIP_TRY(ssdr_chain_ws_blocks_per_cu(&ws_per_cu));
        c->ws_grid = (uint32_t)prop.multiProcessorCount * (uint32_t)(ws_per_cu < 0 ? 0 : ws_per_cu);     // 0: not resident on this device, never chosen
        // Below these batch sizes the two stages side by side are FASTER than a one-read kernel (profiles/r06_ab_small_batches.txt): a one-read kernel
        // walks all lines of a channel pair in one wave, the waterfall kernel spreads them over the chip.  Fused AM kernel: one channel pair per
        // resident wave (8192 channels on an MI355X: -3 % at 6144, +10 % at 8192; 2.8 x slower at 64); wave-specialised kernel: 16 pairs per trio
        // (32768 channels: +1.2 % on narrowed AM; 2 x slower at 1024).
        c->am_floor = 2u * c->fused_grid * (SSDR_WF_BLOCK / 64);
        c->ws_floor = 2u * 16u * c->ws_grid * (SSDR_WS_AUDIO_WAVES / 2);
        return SSDR_OK;
    }();
    if (rc == SSDR_OK) {
        // every channel starts as the reference's default receiver: AM, AGC on (utils:936-944)
        ssdr_chan_params dp;
        ssdr_default_params(SSDR_MODE_AM, &dp);
        std::vector<ssdr_chan_params> all(n_channels, dp);
        rc = ssdr_set_params(c, 0, n_channels, all.data());
    }
    if (rc == SSDR_OK) rc = ssdr_reset_state(c, 0, n_channels);
    if (rc != SSDR_OK) return rc;
    owner.c = nullptr;
    *out = c;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_averaging(ssdr_ctx *c, uint32_t n) SSDR_GUARD
{
    if (!c || n < 1 || n > 100) return SSDR_EINVAL;      // supersdr.py:376-385: averaging_n in 1..100
    if (n != c->n_avg) {
        // like the reference (a new deque per output line, utils:882), a change restarts the group
        HIP_TRY(hipSetDevice(c->device));
        // (partial sums are only read when wf_phase != 0, so no clearing is needed)
        c->wf_phase = 0;
        c->n_avg = n;
    }
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_compile_params_decim(const ssdr_chan_params *p, uint32_t decim, ssdr_chan_consts *consts, float *taps) SSDR_GUARD
{
    return ssdr_compile_params_host(p, consts, taps, decim);
} SSDR_UNGUARD

int ssdr_compile_params_rate(const ssdr_chan_params *p, uint32_t decim, uint32_t rate, ssdr_chan_consts *consts, float *taps) SSDR_GUARD
{
    return ssdr_compile_params_host(p, consts, taps, decim, rate);
} SSDR_UNGUARD

int ssdr_set_decimation(ssdr_ctx *c, uint32_t decim) SSDR_GUARD
{
    if (!c || (decim != 1 && decim != 2 && decim != 4)) return SSDR_EINVAL;
    if (!c->feed.empty()) return SSDR_ESTATE;
    if (decim == c->decim) return SSDR_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<ssdr_chan_params> all = c->h_params;              // recompile every channel for the new input rate
    const uint32_t keep = c->decim;
    c->decim = decim;
    int rc = ssdr_set_params(c, 0, c->n_ch, all.data());
    if (rc == SSDR_OK) {
        c->have_input = false;                                    // a batch pushed at the old rate has the wrong extent
        rc = ssdr_reset_state(c, 0, c->n_ch);                     // phases and histories of the old rate mean nothing now
        if (rc == SSDR_OK) rc = zoom_restart(c, 0, c->n_ch);
    }
    if (rc != SSDR_OK) {                                          // all or nothing: back to the old rate, streams restarted
        c->decim = keep;
        (void)ssdr_set_params(c, 0, c->n_ch, all.data());
        (void)ssdr_reset_state(c, 0, c->n_ch);
    }
    return rc;
} SSDR_UNGUARD

int ssdr_set_hop(ssdr_ctx *c, uint32_t hop) SSDR_GUARD
{
    if (!c || (hop != SSDR_NFFT && hop != SSDR_NFFT / 2)) return SSDR_EINVAL;
    if (!c->feed.empty()) return SSDR_ESTATE;                 // the feed's slots are sized for the hop they were opened with
    if (hop == c->hop) return SSDR_OK;
    HIP_TRY(hipSetDevice(c->device));
    if (hop == SSDR_NFFT / 2) {
        if (!c->d_wf_tail) HIP_TRY(hipMalloc(&c->d_wf_tail, (size_t)c->n_ch * (SSDR_NFFT / 2) * 4));
        HIP_TRY(hipMemsetAsync(c->d_wf_tail, 0, (size_t)c->n_ch * (SSDR_NFFT / 2) * 4, c->stream));   // silence before the stream
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->hop = hop;
    c->wf_phase = 0;                                          // a change of framing restarts the averaging group
    return SSDR_OK;
} SSDR_UNGUARD

// the zoom centres as NCO steps at the current input rate; the zoom streams restart (phase, history, averaging group)
static int zoom_restart(ssdr_ctx *c, uint32_t first, uint32_t count, bool restart_group)
{
    if (c->zoom <= 1 || !c->d_zoom_dphi || !count) return SSDR_OK;
    const double fs_in = (double)c->kiwi_rate * c->decim;
    std::vector<uint32_t> dphi(count);
    for (uint32_t i = 0; i < count; i++) {
        const double x = std::nearbyint(c->h_zoom_offset[first + i] / fs_in * 4294967296.0);
        long long v = (long long)x % 4294967296ll;
        if (v < 0) v += 4294967296ll;
        dphi[i] = (uint32_t)v;
    }
    HIP_TRY(hipMemcpyAsync(c->d_zoom_dphi + first, dphi.data(), count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_zoom_phase + first, 0, count * sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_zoom_hist + (size_t)first * SSDR_ZOOM_HIST, 0, (size_t)count * SSDR_ZOOM_HIST * 4, c->stream));
    if (c->d_wf_tail)        // hop 512: the carried half-line belongs to the old span / centre
        HIP_TRY(hipMemsetAsync(c->d_wf_tail + (size_t)first * (SSDR_NFFT / 2), 0, (size_t)count * (SSDR_NFFT / 2) * 4, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (restart_group) c->wf_phase = 0;
    return SSDR_OK;
}

int ssdr_set_wf_zoom(ssdr_ctx *c, uint32_t zoom) SSDR_GUARD
{
    if (!c || (zoom != 1 && zoom != 2 && zoom != 4 && zoom != 8)) return SSDR_EINVAL;
    if (!c->feed.empty()) return SSDR_ESTATE;              // the feed's slots are sized for un-zoomed lines
    HIP_TRY(hipSetDevice(c->device));
    if (zoom > 1 && !c->d_zoom_dphi) {
        HIP_TRY(hipMalloc(&c->d_zoom_taps, (SSDR_ZOOM_TAPS_MAX + 1) * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_zoom_dphi, (size_t)c->n_ch * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_zoom_phase, (size_t)c->n_ch * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->d_zoom_hist, (size_t)c->n_ch * SSDR_ZOOM_HIST * 4));
    }
    if (c->h_zoom_offset.size() != c->n_ch) c->h_zoom_offset.assign(c->n_ch, 0.0);
    const bool changed = zoom != c->zoom;
    c->zoom = zoom;
    c->wf_phase = 0;
    if (changed && zoom == 1 && c->d_wf_tail) {             // back to the full span: the carried half-line was a zoomed one
        HIP_TRY(hipMemsetAsync(c->d_wf_tail, 0, (size_t)c->n_ch * (SSDR_NFFT / 2) * 4, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (zoom > 1) {
        // the reference's tap formula (utils_supersdr.py:334-344) with the cut-off at the zoomed stream's Nyquist frequency,
        // fl / fs = 1 / (2 Z), and 32 Z - 1 taps: the transition takes the outer ~17 % of the span on either side at every Z
        double h[SSDR_ZOOM_TAPS_MAX + 1];
        const int n = (int)(32 * zoom - 1);
        if (ssdr_design_lowpass_exact(1.0 / (2.0 * zoom), 1.0, n, h) != n) return SSDR_EINVAL;
        float hf[SSDR_ZOOM_TAPS_MAX + 1] = {};
        for (int i = 0; i < n; i++) hf[i] = (float)h[i];
        c->zoom_ntap = (uint32_t)n;
        HIP_TRY(hipMemcpy(c->d_zoom_taps, hf, sizeof hf, hipMemcpyHostToDevice));
        return zoom_restart(c, 0, c->n_ch);
    }
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_wf_center(ssdr_ctx *c, uint32_t first, uint32_t count, const double *offset_hz) SSDR_GUARD
{
    if (!c || !offset_hz || (uint64_t)first + count > c->n_ch) return SSDR_EINVAL;
    const double half = 0.5 * (double)c->kiwi_rate * c->decim;
    for (uint32_t i = 0; i < count; i++)
        if (!(std::fabs(offset_hz[i]) <= half)) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    if (c->h_zoom_offset.size() != c->n_ch) c->h_zoom_offset.assign(c->n_ch, 0.0);
    for (uint32_t i = 0; i < count; i++) c->h_zoom_offset[first + i] = offset_hz[i];
    return zoom_restart(c, first, count);
} SSDR_UNGUARD

int ssdr_read_zoom(ssdr_ctx *c, uint32_t first, uint32_t count, int16_t *iq_out, uint32_t *samples_per_channel) SSDR_GUARD
{
    if (!c || (uint64_t)first + count > c->n_ch) return SSDR_EINVAL;
    if (c->zoom <= 1 || !c->d_zoom_out || c->zoom_run_samples == 0) return SSDR_ESTATE;
    if (samples_per_channel) *samples_per_channel = c->zoom_run_samples;
    if (!iq_out) return SSDR_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(iq_out, c->d_zoom_out + (size_t)first * c->zoom_run_samples, (size_t)count * c->zoom_run_samples * 4,
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_exact_bins(ssdr_ctx *c, int on) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    if (on && !c->d_tw64) {
        std::vector<double> tw(2 * SSDR_TW64_N);
        ssdr_make_tw64(tw.data());
        HIP_TRY(hipMalloc(&c->d_tw64, SSDR_TW64_N * sizeof(double2)));
        HIP_TRY(hipMemcpy(c->d_tw64, tw.data(), SSDR_TW64_N * sizeof(double2), hipMemcpyHostToDevice));
    }
    c->exact_bins = on != 0;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_get_config(ssdr_ctx *c, uint32_t *hop, uint32_t *decim, uint32_t *averaging, uint32_t *kiwi_rate) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (hop) *hop = c->hop;
    if (decim) *decim = c->decim;
    if (averaging) *averaging = c->n_avg;
    if (kiwi_rate) *kiwi_rate = c->kiwi_rate;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_stream(ssdr_ctx *c, void *hip_stream) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_concurrent(ssdr_ctx *c, int on) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream2));
    c->concurrent = (on & 1) != 0;
    c->audio_serial = (on & 2) != 0;
    c->audio_pending = false;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_profiling(ssdr_ctx *c, int on) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    c->profiling = on != 0;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_kernel_stats(ssdr_ctx *c, int which, float *total_ms, uint32_t *launches, int reset) SSDR_GUARD
{
    if (!c || which < 0 || which >= SSDR_K_COUNT) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    int rc = resolve_pending(c);
    if (rc != SSDR_OK) return rc;
    if (total_ms) *total_ms = c->k_ms[which];
    if (launches) *launches = c->k_n[which];
    if (reset) { c->k_ms[which] = 0.0f; c->k_n[which] = 0; }
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_elapsed_ms(ssdr_ctx *c, float *ms) SSDR_GUARD
{
    if (!c || !ms) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    if (c->profiling) {
        int rc = resolve_pending(c);
        if (rc != SSDR_OK) return rc;
        *ms = c->last_ms;
        return SSDR_OK;
    }
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_sync(ssdr_ctx *c) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->concurrent || c->audio_pending) HIP_TRY(hipStreamSynchronize(c->stream2));
    return SSDR_OK;
} SSDR_UNGUARD

// An audio stage that ran beside the waterfall kernel (stream2) and has not been joined yet: everything that follows on the
// main stream and touches what it reads or writes (input, constants, state, PCM, RSSI, flags) waits for it first.
static int join_audio(ssdr_ctx *c)
{
    if (c->audio_pending) { HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_a, 0)); c->audio_pending = false; }
    return SSDR_OK;
}
// ... and before a buffer it uses is freed on the host side
static int drain_audio(ssdr_ctx *c)
{
    if (c->audio_pending) { HIP_TRY(hipStreamSynchronize(c->stream2)); c->audio_pending = false; }
    return SSDR_OK;
}

// what the per-call decisions need to know about the channels' constants, counted once per change of them
static void chan_summary(ssdr_ctx *c)
{
    if (!c->summary_dirty) return;
    for (int p = 0; p < SSDR_PATH_COUNT; p++) c->sum_paths[p] = 0;
    c->sum_any_iq = false;
    for (uint32_t ch = 0; ch < c->n_ch; ch++) {
        const int path = ssdr_audio_path(c->h_consts[ch]);
        c->sum_paths[path]++;
        c->sum_any_iq = c->sum_any_iq || c->h_consts[ch].mode == SSDR_MODE_IQ;
    }
    c->summary_dirty = false;
}

// channels sorted by audio frame path; rebuilt after ssdr_set_params.  The only host allocation of a run: made before a stage
// of the call has been launched or any bookkeeping advanced (ssdr_run_chain calls it first), so a failure leaves the streams untouched.
static int ensure_chan_list(ssdr_ctx *c, hipStream_t s)
{
    if (!c->chan_list_dirty) return SSDR_OK;
    std::vector<uint32_t> list(c->n_ch);
    uint32_t pos = 0, off[SSDR_PATH_COUNT], cnt[SSDR_PATH_COUNT];
    for (int p = 0; p < SSDR_PATH_COUNT; p++) {
        off[p] = pos;
        for (uint32_t ch = 0; ch < c->n_ch; ch++)
            if (ssdr_audio_path(c->h_consts[ch]) == p) list[pos++] = ch;
        cnt[p] = pos - off[p];
    }
    HIP_TRY(hipMemcpyAsync(c->d_chan_list, list.data(), (size_t)c->n_ch * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    // the chain list of the wave-specialised kernel: consecutive channels of the sorted list form pairs (so a pair is of one path,
    // except where two paths meet); the pairs of the three paths are dealt out evenly over the list -- pair m of a path with p pairs
    // sits at (m + 1/2) / p of the way -- so that at any time the trios of a workgroup work on the ctx's mix of paths
    std::vector<uint32_t> ws((size_t)c->n_ch + 1, 0u);      // (+ the ticket word: it starts over at zero with every new list)
    {
        const uint32_t n_pairs = (c->n_ch + 1) / 2;
        std::vector<uint32_t> first_of[SSDR_PATH_COUNT];                   // pairs by the path of their first channel
        for (uint32_t j = 0; j < n_pairs; j++) first_of[ssdr_audio_path(c->h_consts[list[2 * j]])].push_back(j);
        std::vector<std::pair<double, uint32_t>> order;
        order.reserve(n_pairs);
        for (int p = 0; p < SSDR_PATH_COUNT; p++)
            for (size_t m = 0; m < first_of[p].size(); m++)
                if (first_of[p][m] != n_pairs - 1 || !(c->n_ch & 1u))          // (a single last channel stays the list's last pair)
                    order.emplace_back(((double)m + 0.5) / (double)first_of[p].size(), first_of[p][m]);
        std::stable_sort(order.begin(), order.end(), [](const std::pair<double, uint32_t> &x, const std::pair<double, uint32_t> &y) { return x.first < y.first; });
        if (c->n_ch & 1u) order.emplace_back(2.0, n_pairs - 1);
        uint32_t w = 0;
        for (const auto &o : order) {
            ws[w++] = list[2 * o.second];
            if (2 * o.second + 1 < c->n_ch) ws[w++] = list[2 * o.second + 1];
        }
    }
    HIP_TRY(hipMemcpyAsync(c->d_ws_list, ws.data(), ((size_t)c->n_ch + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    c->ws_ticket = 0;
    HIP_TRY(hipStreamSynchronize(s));    // `list` goes out of scope
    for (int p = 0; p < SSDR_PATH_COUNT; p++) { c->path_off[p] = off[p]; c->path_n[p] = cnt[p]; }
    c->chan_list_dirty = false;
    return SSDR_OK;
}

// input samples (dwords) per channel of a batch of n_frames frames: a frame yields 512 PCM samples and takes 512 * D of IQ
static inline size_t in_len(const ssdr_ctx *c, uint32_t n_frames) { return (size_t)n_frames * SSDR_FRAME * c->decim; }

static int ensure_input(ssdr_ctx *c, uint32_t n_frames)
{
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    const size_t need = (size_t)n_frames * c->decim;             // capacity is kept in 512-sample units
    if (c->iq_own_frames < need) {
        if (c->d_iq_own) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_iq_own)); c->d_iq_own = nullptr; c->iq_own_frames = 0; }
        HIP_TRY(hipMalloc(&c->d_iq_own, (size_t)c->n_ch * need * SSDR_FRAME * 4));
        c->iq_own_frames = need;
    }
    return SSDR_OK;
}

int ssdr_push_iq(ssdr_ctx *c, const int16_t *iq, uint32_t n_frames, int is_device) SSDR_GUARD
{
    if (!c || !iq || n_frames == 0) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    if (is_device) {
        c->d_iq = reinterpret_cast<const uint32_t *>(iq);
    } else {
        int rc = ensure_input(c, n_frames);
        if (rc != SSDR_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_iq_own, iq, (size_t)c->n_ch * in_len(c, n_frames) * 4, hipMemcpyHostToDevice, c->stream));
        c->d_iq = c->d_iq_own;
    }
    c->in_frames = n_frames;
    c->have_input = true;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_synth_iq(ssdr_ctx *c, uint32_t n_frames, uint32_t seed, uint32_t first_channel_id) SSDR_GUARD
{
    if (!c || n_frames == 0) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_input(c, n_frames);
    if (rc != SSDR_OK) return rc;
    SsdrSynthArgs a;
    a.iq = c->d_iq_own;
    a.ch_stride = (uint64_t)in_len(c, n_frames);
    a.n_ch = c->n_ch;
    a.n_samples = (uint32_t)in_len(c, n_frames);
    a.seed = seed;
    a.first_channel_id = first_channel_id;
    a.sample0 = c->synth_sample0;
    if ((rc = timed_begin(c)) != SSDR_OK) return rc;
    HIP_TRY(ssdr_launch_synth(a, c->stream));
    if ((rc = timed_end(c, SSDR_K_SYNTH)) != SSDR_OK) return rc;
    c->synth_sample0 += a.n_samples;
    c->d_iq = c->d_iq_own;
    c->in_frames = n_frames;
    c->have_input = true;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_read_input(ssdr_ctx *c, uint32_t first, uint32_t count, int16_t *iq_out) SSDR_GUARD
{
    if (!c || !iq_out || (uint64_t)first + count > c->n_ch) return SSDR_EINVAL;
    if (!c->have_input) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    const size_t per_ch = in_len(c, c->in_frames);
    HIP_TRY(hipMemcpyAsync(iq_out, c->d_iq + (size_t)first * per_ch, (size_t)count * per_ch * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

// The shape rules of a waterfall batch, checked before ANY stage of a call is launched (ssdr_run_chain runs its audio stage first
// when the stages go side by side: a batch the waterfall stage would refuse must not have advanced the audio state by then).
static int validate_wf_batch(const ssdr_ctx *c)
{
    const bool hop512 = c->hop == SSDR_NFFT / 2;
    const bool zoomed = c->zoom > 1;
    if (zoomed && (c->fuse_next || (c->in_frames * c->decim) % c->zoom)) return SSDR_EINVAL;
    const uint32_t halves = c->in_frames * c->decim / c->zoom;       // 512-sample half-lines the batch yields (of the zoomed stream)
    if (!hop512 && (halves & 1u)) return SSDR_EINVAL;                // a batch must hold a whole number of lines
    if (zoomed && halves == 0) return SSDR_EINVAL;
    return SSDR_OK;
}

int ssdr_run_wf(ssdr_ctx *c, int16_t *wf_sum_out, uint32_t *lines_ready, int out_is_device) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (!c->have_input) return SSDR_ESTATE;
    const bool hop512 = c->hop == SSDR_NFFT / 2;
    const bool zoomed = c->zoom > 1;
    { const int rcv = validate_wf_batch(c); if (rcv != SSDR_OK) return rcv; }
    const uint32_t halves = c->in_frames * c->decim / c->zoom;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t *wf_src = c->d_iq;                                // what the waterfall kernel reads: the input, or the zoomed stream
    uint64_t wf_stride = (uint64_t)in_len(c, c->in_frames);
    if (zoomed) {
        const uint32_t n_in = (uint32_t)in_len(c, c->in_frames), n_out = n_in / c->zoom;
        if (c->zoom_out_samples < n_out) {
            if (c->d_zoom_out) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_zoom_out)); c->d_zoom_out = nullptr; c->zoom_out_samples = 0; }
            HIP_TRY(hipMalloc(&c->d_zoom_out, (size_t)c->n_ch * n_out * 4));
            c->zoom_out_samples = n_out;
        }
        SsdrZoomArgs z;
        z.iq = c->d_iq; z.ch_stride = wf_stride; z.n_ch = c->n_ch; z.n_in = n_in; z.zoom = c->zoom; z.ntap = c->zoom_ntap;
        z.taps = c->d_zoom_taps; z.dphi = c->d_zoom_dphi; z.phase = c->d_zoom_phase; z.hist = c->d_zoom_hist; z.out = c->d_zoom_out;
        int rcz;
        if ((rcz = timed_begin(c)) != SSDR_OK) return rcz;
        HIP_TRY(ssdr_launch_zoom(z, c->stream));
        if ((rcz = timed_end(c, SSDR_K_ZOOM)) != SSDR_OK) return rcz;
        wf_src = c->d_zoom_out;
        wf_stride = n_out;
        c->zoom_run_samples = n_out;
    }
    const uint32_t n_lines = hop512 ? halves : halves / 2;
    const uint32_t total = c->wf_phase + n_lines;
    const uint32_t n_out = total / c->n_avg;
    const uint32_t n_groups = (total + c->n_avg - 1) / c->n_avg;
    if (c->wf_out_lines < n_out) {
        if (c->d_wf_out) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_wf_out)); c->d_wf_out = nullptr; c->wf_out_lines = 0; }
        HIP_TRY(hipMalloc(&c->d_wf_out, (size_t)n_out * c->n_ch * SSDR_NFFT * 2));
        c->wf_out_lines = n_out;
    }
    SsdrWfArgs a;
    a.iq = wf_src;
    a.ch_stride = wf_stride;
    a.n_ch = c->n_ch;
    a.n_lines = n_lines;
    a.tail = hop512 ? c->d_wf_tail : nullptr;
    a.n_avg = c->n_avg;
    a.phase = c->wf_phase;
    a.n_groups = n_groups;
    a.grp_run = 1;
    a.out = c->d_wf_out;
    a.acc_in = c->d_wf_acc[c->wf_acc_cur];
    a.acc_out = c->d_wf_acc[c->wf_acc_cur ^ 1];
    a.consts = c->d_consts;
    a.win = c->d_win;
    a.tw_stage = c->d_tw;
    a.lut = c->d_lut;
    uint64_t items = (uint64_t)((c->n_ch + 1) / 2) * n_groups;
    const uint32_t wf_grid = c->concurrent ? c->wf_grid_1 : c->wf_grid;
    if (hop512 && n_groups) {
        // a wave works through a run of consecutive groups of its channel pair, so the half-line two lines share is read
        // again by the wave that fetched it one line earlier; runs as long as still leave every resident wave ~8 items
        const uint64_t waves = (uint64_t)(wf_grid ? wf_grid : 1) * (SSDR_WF_BLOCK / 64);
        uint64_t run = items / (8 * waves);
        run = run < 1 ? 1 : (run > n_groups ? n_groups : run);
        a.grp_run = (uint32_t)run;
        items = (uint64_t)((c->n_ch + 1) / 2) * ((n_groups + run - 1) / run);
    }
    const uint64_t need = (items + SSDR_WF_BLOCK / 64 - 1) / (SSDR_WF_BLOCK / 64);
    const uint32_t grid = (uint32_t)(need < wf_grid ? need : wf_grid);
    int rc;
    if (c->fuse_next) {                      // the fused superframe kernel does this stage's work: ssdr_run_audio launches it
        c->fused_wf = a;
    } else {
        if ((rc = timed_begin(c)) != SSDR_OK) return rc;
        if (c->exact_bins) HIP_TRY(ssdr_launch_wf_exact(a, c->d_tw64, c->stream));
        else HIP_TRY(ssdr_launch_wf(a, grid ? grid : 1, c->stream));
        if ((rc = timed_end(c, SSDR_K_WF)) != SSDR_OK) return rc;
    }
    if (hop512 && !c->fuse_next) // the batch's last half-line is the next batch's first: [n_ch] rows of 2 KB out of the input
        HIP_TRY(hipMemcpy2DAsync(c->d_wf_tail, (SSDR_NFFT / 2) * 4, wf_src + (size_t)(halves - 1) * SSDR_FRAME,
                                 wf_stride * 4, (SSDR_NFFT / 2) * 4, c->n_ch, hipMemcpyDeviceToDevice, c->stream));
    c->wf_phase = total % c->n_avg;
    if (c->wf_phase) c->wf_acc_cur ^= 1;             // a partial group was written to acc_out
    c->wf_lines_ready = n_out;
    if (lines_ready) *lines_ready = n_out;
    if (wf_sum_out && n_out) {
        const size_t bytes = (size_t)n_out * c->n_ch * SSDR_NFFT * 2;
        HIP_TRY(hipMemcpyAsync(wf_sum_out, c->d_wf_out, bytes, out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
        if (!out_is_device) HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_run_audio(ssdr_ctx *c, int16_t *pcm_out, float *rssi_out, int out_is_device) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (!c->have_input) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->concurrent) { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }      // the previous frame's state before this one
    if (c->audio_frames < c->in_frames || c->flags_frames < c->in_frames) { int rcd = drain_audio(c); if (rcd != SSDR_OK) return rcd; }
    if (c->audio_frames < c->in_frames) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_pcm) { HIP_TRY(hipFree(c->d_pcm)); c->d_pcm = nullptr; }
        if (c->d_rssi) { HIP_TRY(hipFree(c->d_rssi)); c->d_rssi = nullptr; }
        c->audio_frames = 0;
        HIP_TRY(hipMalloc(&c->d_pcm, (size_t)c->n_ch * c->in_frames * SSDR_FRAME * 2));
        HIP_TRY(hipMalloc(&c->d_rssi, (size_t)c->n_ch * c->in_frames * sizeof(float)));
        c->audio_frames = c->in_frames;
    }
    if (c->flags_frames < c->in_frames) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_flags) { HIP_TRY(hipFree(c->d_flags)); c->d_flags = nullptr; }
        c->flags_frames = 0;
        HIP_TRY(hipMalloc(&c->d_flags, (size_t)c->n_ch * c->in_frames));
        c->flags_frames = c->in_frames;
    }
    SsdrAudioArgs a;
    a.iq = c->d_iq;
    a.ch_stride = (uint64_t)in_len(c, c->in_frames);
    a.n_ch = c->n_ch;
    a.n_frames = c->in_frames;
    a.consts = c->d_consts;
    a.taps = c->d_taps;
    a.state = c->d_state;
    a.hist = c->d_hist;
    a.pcm = c->d_pcm;
    a.rssi = c->d_rssi;
    a.flags = c->d_flags;
    a.iq_out = nullptr;
    c->iq_out_valid = false;
    {
        chan_summary(c);
        const bool any_iq = c->sum_any_iq;
        if (any_iq && c->feed.empty()) {          // (the pipelined feed hands out PCM rows only: an IQ channel's row carries I)
            if (c->iq_out_frames < c->in_frames) {
                { int rcd = drain_audio(c); if (rcd != SSDR_OK) return rcd; }
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (c->d_iq_out) { HIP_TRY(hipFree(c->d_iq_out)); c->d_iq_out = nullptr; }
                c->iq_out_frames = 0;
                HIP_TRY(hipMalloc(&c->d_iq_out, (size_t)c->n_ch * c->in_frames * SSDR_FRAME * 4));
                c->iq_out_frames = c->in_frames;
            }
            HIP_TRY(hipMemsetAsync(c->d_iq_out, 0, (size_t)c->n_ch * c->in_frames * SSDR_FRAME * 4, c->stream));   // rows of the other modes
            a.iq_out = c->d_iq_out;
            c->iq_out_valid = true;
        }
    }
    int rc;
    hipStream_t s = c->stream;
    if (c->concurrent) {
        // the audio kernel only depends on the input batch (and on its own previous launch): run it beside the
        // waterfall kernel on a second stream so that its waves fill the issue slots the waterfall leaves idle
        s = c->stream2;
        HIP_TRY(hipEventRecord(c->ev_in, c->stream));          // everything queued so far, incl. the input copy/synth
        HIP_TRY(hipStreamWaitEvent(s, c->ev_in, 0));
    }
    { int rcl = ensure_chan_list(c, s); if (rcl != SSDR_OK) return rcl; }
    c->audio_started = true;
    c->audio_run_frames = c->in_frames;
    if (c->fuse_next) {                      // waterfall + full-band AM audio in one kernel: one read of the input
        SsdrFusedArgs fa;
        fa.wf = c->fused_wf;
        fa.au = a;
        fa.au.chan_list = c->d_ws_list;        // (the wave-specialised kernel draws pairs from its own list of all channels)
        fa.au.list_n = c->n_ch;
        fa.ticket = c->d_ws_list + c->n_ch;
        fa.ticket_base = c->ws_ticket;
        const uint64_t pairs = (c->n_ch + 1) / 2;
        const uint64_t need = (pairs + SSDR_WF_BLOCK / 64 - 1) / (SSDR_WF_BLOCK / 64);
        const uint32_t grid = (uint32_t)(need < c->fused_grid ? need : c->fused_grid);
        if ((rc = timed_begin(c, s)) != SSDR_OK) return rc;
        if (c->fuse_ws_next) {
            const uint64_t need_g = ((uint64_t)c->n_ch + SSDR_WS_AUDIO_WAVES - 1) / SSDR_WS_AUDIO_WAVES;
            const uint32_t grid_g = (uint32_t)(need_g < c->ws_grid ? need_g : c->ws_grid);
            HIP_TRY(ssdr_launch_chain_ws(fa, grid_g ? grid_g : 1, s));
            c->ws_ticket += (uint32_t)pairs + (grid_g ? grid_g : 1) * (SSDR_WS_AUDIO_WAVES / 2);     // (wraps as the device word does)
        } else if (c->exact_bins) HIP_TRY(ssdr_launch_fused_exact_am(fa, c->d_tw64, s));
        else HIP_TRY(ssdr_launch_fused_am(fa, grid ? grid : 1, s));
        if ((rc = timed_end(c, SSDR_K_FUSED, s)) != SSDR_OK) return rc;
        if (fa.wf.tail)          // hop 512: only now may the carried half-line (the kernel's line 0 read it) become this batch's last one
            HIP_TRY(hipMemcpy2DAsync(c->d_wf_tail, (SSDR_NFFT / 2) * 4, fa.wf.iq + (size_t)(fa.wf.n_lines - 1) * SSDR_FRAME,
                                     fa.wf.ch_stride * 4, (SSDR_NFFT / 2) * 4, c->n_ch, hipMemcpyDeviceToDevice, s));
        return SSDR_OK;
    }
    // one kernel per non-empty path: the first on the stream itself, the others beside it on their own streams
    // (fork and join by events); the stage is timed between two events on `s`
    if (c->decim > 1 && (c->path_n[SSDR_PATH_DELAY4] || c->path_n[SSDR_PATH_AM_RAW]))
        return SSDR_ESTATE;                   // the decimating kernel is the general path: no channel may be compiled for a shift path
    if ((rc = timed_begin(c, s)) != SSDR_OK) return rc;
    if (c->decim > 1) {                       // ONE kernel over all channels (it takes no channel list)
        a.chan_list = c->d_chan_list;
        a.list_n = c->n_ch;
        HIP_TRY(ssdr_launch_audio_dec(a, c->decim, s));
    } else {
        int n_paths = 0, n_side = 0;
        for (int p = 0; p < SSDR_PATH_COUNT; p++) n_paths += c->path_n[p] != 0;
        const bool side = n_paths > 1 && !c->audio_serial;
        if (side) HIP_TRY(hipEventRecord(c->ev_fork, s));
        bool first = true;
        for (int p = 0; p < SSDR_PATH_COUNT; p++) {
            if (!c->path_n[p]) continue;
            a.chan_list = c->d_chan_list + c->path_off[p];
            a.list_n = c->path_n[p];
            if (first || !side) {
                HIP_TRY(ssdr_launch_audio(a, p, s));
            } else {
                hipStream_t ps = c->path_stream[n_side];
                HIP_TRY(hipStreamWaitEvent(ps, c->ev_fork, 0));
                HIP_TRY(ssdr_launch_audio(a, p, ps));
                HIP_TRY(hipEventRecord(c->ev_path[n_side], ps));
                n_side++;
            }
            first = false;
        }
        for (int i = 0; i < n_side; i++) HIP_TRY(hipStreamWaitEvent(s, c->ev_path[i], 0));
    }
    if ((rc = timed_end(c, SSDR_K_AUDIO, s)) != SSDR_OK) return rc;
    const hipMemcpyKind kind = out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (pcm_out) HIP_TRY(hipMemcpyAsync(pcm_out, c->d_pcm, (size_t)c->n_ch * c->in_frames * SSDR_FRAME * 2, kind, s));
    if (rssi_out) HIP_TRY(hipMemcpyAsync(rssi_out, c->d_rssi, (size_t)c->n_ch * c->in_frames * sizeof(float), kind, s));
    if ((pcm_out || rssi_out) && !out_is_device) HIP_TRY(hipStreamSynchronize(s));
    if (c->concurrent) {                                       // later work on the main stream (next input, playbuffer) follows the audio
        HIP_TRY(hipEventRecord(c->ev_a, s));
        c->audio_pending = true;
    }
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_run_chain(ssdr_ctx *c, uint32_t *lines_ready, int *fused) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (!c->have_input) return SSDR_ESTATE;
    chan_summary(c);
    const uint32_t n_am = c->sum_paths[SSDR_PATH_AM_RAW];
    // the fused kernel covers the metric's configuration: every channel on the full-band AM path, N = 1, 12 kHz IQ, either line
    // rate (hop 1024, or hop 512 = the reference's 23 lines/s); from eight frames per call on (a wave sets a channel pair's
    // carried state up once per call: measured ahead from there)
    const bool hop512 = c->hop == SSDR_NFFT / 2;            // (one line per frame: any frame count; hop 1024 needs whole lines)
    // (N > 1 and hop 512 are opt-in, ssdr_set_fused(ctx, 2): there the two stages side by side are faster)
    // (with float64 bins: the float64 counterpart, ssdr_fused_exact_am_kernel -- hop 1024 and N = 1 only)
    const bool eligible = n_am == c->n_ch && c->decim == 1 && (hop512 || !(c->in_frames & 1u)) &&
                          c->in_frames >= 8 &&
                          !c->concurrent && c->fused_grid != 0 && c->fused_enabled >= ((hop512 || c->n_avg > 1) ? 2 : 1) && c->zoom == 1 &&
                          (c->fused_enabled >= 2 || c->n_ch >= c->am_floor) &&
                          (!c->exact_bins || (!hop512 && c->n_avg == 1));
    // the wave-specialised kernel (ssdr_chain_ws.hip): any mix of audio paths, any filter and any N at hop 1024, fp32 bins -- both stages on one
    // read of the input.  By default where it is also the faster way (profiles/r06_ab_chain_ws.txt): when every channel runs the general path
    // (a filter to apply: the stages side by side are bound by the board's power cap there, and the second read of the input is energy);
    // for every batch it can take with ssdr_set_fused(ctx, 3) (full-band channels among them: 1 % slower than side by side, 39 % less HBM traffic)
    const bool ws_can = c->ws_grid != 0 && c->decim == 1 && !hop512 && !(c->in_frames & 1u) && !c->concurrent && c->zoom == 1 && !c->exact_bins;
    const bool eligible_ws = !eligible && ws_can &&
                             (c->fused_enabled >= 3 || (c->fused_enabled >= 1 && c->sum_paths[SSDR_PATH_GENERAL] == c->n_ch && c->in_frames >= 8 &&
                                                        c->n_ch >= c->ws_floor));
    if (fused) *fused = eligible ? 1 : (eligible_ws ? 2 : 0);
    c->fuse_next = eligible || eligible_ws;
    c->fuse_ws_next = eligible_ws;
    {   // both stages or neither: what ssdr_run_wf and ssdr_run_audio would refuse is refused before either is launched
        int rcv = validate_wf_batch(c);
        if (rcv == SSDR_OK && c->decim > 1) {
            chan_summary(c);
            if (c->sum_paths[SSDR_PATH_DELAY4] || c->sum_paths[SSDR_PATH_AM_RAW]) rcv = SSDR_ESTATE;
        }
        if (rcv == SSDR_OK && c->chan_list_dirty) {      // (the previous call's audio stage may still be reading the list)
            rcv = [&]() -> int { HIP_TRY(hipSetDevice(c->device)); return join_audio(c); }();
            if (rcv == SSDR_OK) rcv = ensure_chan_list(c, c->stream);
        }
        if (rcv != SSDR_OK) { c->fuse_next = false; c->fuse_ws_next = false; if (fused) *fused = 0; return rcv; }
    }
    // Everything else: the two stages side by side -- the audio stage on a second stream beside the waterfall kernel (one workgroup
    // per CU then), each filling the issue slots the other leaves: +2.7 % on configs[3], +9 % on the full chain at hop 512
    // (profiles/r04_ab_overlap.txt; there it beats the one-read kernel too, which is why that one is opt-in at hop 512)
    // (not with the float64 waterfall kernel: it fills the CUs' LDS by itself, and beside it the audio stage only gets in the way:
    //  3.61 ms one after the other, 3.75 ms side by side)
    const bool overlap = !c->fuse_next && c->overlap_enabled && !c->concurrent && !c->exact_bins;
    int rc;
    if (overlap) {
        // the audio stage first: its stream waits for what is queued so far (the input), not for the waterfall kernel that follows
        c->concurrent = true;
        rc = ssdr_run_audio(c, nullptr, nullptr, 0);
        if (rc == SSDR_OK) rc = ssdr_run_wf(c, nullptr, lines_ready, 0);
        c->concurrent = false;                       // (audio_pending stays set: whoever needs the results or the input joins first)
        if (rc == SSDR_OK && c->stream != c->own_stream) rc = join_audio(c);     // a caller's stream (ssdr_set_stream): what they order behind it covers both stages
    } else {
        // a one-read kernel: ssdr_run_wf only does the waterfall stage's bookkeeping and parks its arguments, ssdr_run_audio launches.  Should the
        // launch fail, the bookkeeping goes back to where it stood: both stages or neither
        const uint32_t phase0 = c->wf_phase, ready0 = c->wf_lines_ready;
        const int acc0 = c->wf_acc_cur;
        const bool one_read = c->fuse_next;
        rc = ssdr_run_wf(c, nullptr, lines_ready, 0);
        if (rc == SSDR_OK) {
            rc = ssdr_run_audio(c, nullptr, nullptr, 0);
            if (rc != SSDR_OK && one_read) {
                c->wf_phase = phase0; c->wf_lines_ready = ready0; c->wf_acc_cur = acc0;
                if (lines_ready) *lines_ready = 0;
            }
        }
    }
    c->fuse_next = false;
    c->fuse_ws_next = false;
    return rc;
} SSDR_UNGUARD

int ssdr_set_fused(ssdr_ctx *c, int on) SSDR_GUARD
{
    if (!c || on < 0 || on > 3) return SSDR_EINVAL;
    c->fused_enabled = on;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_chain_floors(ssdr_ctx *c, uint32_t fused_am_min_channels, uint32_t chain_ws_min_channels) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    c->am_floor = fused_am_min_channels;
    c->ws_floor = chain_ws_min_channels;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_get_chain_floors(ssdr_ctx *c, uint32_t *fused_am_min_channels, uint32_t *chain_ws_min_channels) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (fused_am_min_channels) *fused_am_min_channels = c->am_floor;
    if (chain_ws_min_channels) *chain_ws_min_channels = c->ws_floor;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_overlap(ssdr_ctx *c, int on) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    c->overlap_enabled = on != 0;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_audio_paths(ssdr_ctx *c, uint32_t counts[3]) SSDR_GUARD
{
    if (!c || !counts) return SSDR_EINVAL;
    chan_summary(c);
    for (int p = 0; p < SSDR_PATH_COUNT; p++) counts[p] = c->sum_paths[p];
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_audio_iq(ssdr_ctx *c, int16_t *iq_out, int out_is_device) SSDR_GUARD
{
    if (!c || !iq_out) return SSDR_EINVAL;
    if (!c->iq_out_valid || !c->d_iq_out || c->audio_run_frames == 0) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    HIP_TRY(hipMemcpyAsync(iq_out, c->d_iq_out, (size_t)c->n_ch * c->audio_run_frames * SSDR_FRAME * 4,
                           out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    if (!out_is_device) HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_audio_flags(ssdr_ctx *c, uint8_t *flags_out, int out_is_device) SSDR_GUARD
{
    if (!c || !flags_out) return SSDR_EINVAL;
    if (!c->d_flags || c->audio_run_frames == 0 || c->flags_frames < c->audio_run_frames) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    HIP_TRY(hipMemcpyAsync(flags_out, c->d_flags, (size_t)c->n_ch * c->audio_run_frames,
                           out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    if (!out_is_device) HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

// play_buffer's constant tables and carried history (utils_supersdr.py:999-1005), created at first use
static int ensure_play(ssdr_ctx *c)
{
    if (!c->d_play) {
        HIP_TRY(hipMalloc(&c->d_play, (size_t)c->n_ch * sizeof(ssdr_play_chan)));
        HIP_TRY(hipMalloc(&c->d_play_taps, 33 * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_play_hist, (size_t)c->n_ch * 8 * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_play_hist_alt, (size_t)c->n_ch * 8 * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_play_rs_taps, sizeof(SSDR_RS_TAPS)));
        HIP_TRY(hipMemsetAsync(c->d_play_hist, 0, (size_t)c->n_ch * 8 * sizeof(double), c->stream));   // old_buffer = zeros (:1005)
        if (c->pending_play_hist.size() == (size_t)c->n_ch * 8) {
            HIP_TRY(hipMemcpyAsync(c->d_play_hist, c->pending_play_hist.data(), (size_t)c->n_ch * 8 * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->pending_play_hist.clear();
        }
        double h[64];
        if (ssdr_design_lowpass(SSDR_RATE / 2.0, 48000.0, 63, h) != 33) return SSDR_EINVAL;             // filtering(KIWI_RATE/2, AUDIO_RATE)
        for (int j = 0; j < 33; j++) h[j] *= 4.0;       // "* self.SAMPLE_RATIO" (:1134) folded into the taps: a power of two commutes with every rounding of the sum
        HIP_TRY(hipMemcpyAsync(c->d_play_taps, h, 33 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->d_play_rs_taps, SSDR_RS_TAPS, sizeof(SSDR_RS_TAPS), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SSDR_OK;
}

static int wfdata_feed(ssdr_ctx *c, const float *color, uint32_t lines);

// ---- pipelined host feed ------------------------------------------------------------------------------------
// Three streams: host->device copy of batch k+1, the two kernels of batch k, device->host copy of batch k-1.
// The kernels stay on the ctx stream, in batch order, so the per-channel state and the waterfall's partial sums
// carry from batch to batch exactly as with ssdr_push_iq / ssdr_run_*.
int ssdr_feed_close(ssdr_ctx *c) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (c->feed.empty()) return SSDR_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->feed_s_in) (void)hipStreamSynchronize(c->feed_s_in);
    if (c->feed_s_out) (void)hipStreamSynchronize(c->feed_s_out);
    for (auto &s : c->feed) {
        void *hp[] = {s.h_in, s.h_wf, s.h_pcm, s.h_rssi, s.h_wire_rssi, s.h_flags, s.h_color, s.h_dbchan, s.h_playchan, s.h_play, s.h_mono};
        for (void *p : hp) if (p) (void)hipHostFree(p);
        void *dp[] = {s.d_in, s.d_wf, s.d_pcm, s.d_rssi, s.d_wire, s.d_wire_rssi, s.d_flags, s.d_color, s.d_dbchan, s.d_play, s.d_mono,
                      s.d_sel_wf, s.d_sel_pcm, s.d_sel_rssi, s.d_sel_wire_rssi, s.d_sel_flags};
        for (void *p : dp) if (p) (void)hipFree(p);
        hipEvent_t ev[] = {s.ev_in, s.ev_run, s.ev_out};
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    }
    c->feed.clear();
    if (c->feed_s_in) { (void)hipStreamDestroy(c->feed_s_in); c->feed_s_in = nullptr; }
    if (c->feed_s_out) { (void)hipStreamDestroy(c->feed_s_out); c->feed_s_out = nullptr; }
    c->feed_frames = c->feed_head = c->feed_tail = c->feed_inflight = 0;
    c->feed_taken = false;
    c->feed_post = false;
    c->feed_lazy = false;
    c->feed_last = -1;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_feed_open(ssdr_ctx *c, uint32_t n_frames, uint32_t depth, uint32_t flags) SSDR_GUARD
{
    if (!c || n_frames == 0 || (n_frames & 1u) || depth < 2 || depth > 16 || (flags & ~(uint32_t)(SSDR_FEED_WIRE | SSDR_FEED_POST | SSDR_FEED_LAZY_OUT))) return SSDR_EINVAL;
    if (!c->feed.empty() || c->concurrent || c->decim != 1 || c->zoom != 1) return SSDR_ESTATE;      // the feed's slots are sized for un-zoomed 12 kHz IQ
    HIP_TRY(hipSetDevice(c->device));
    const bool post = (flags & SSDR_FEED_POST) != 0;
    if (post) { int rcp = ensure_play(c); if (rcp != SSDR_OK) return rcp; }
    struct Undo { ssdr_ctx *c; bool armed; ~Undo() { if (armed) (void)ssdr_feed_close(c); } } undo{c, true};   // an error or an exception below: no half-open feed
    c->feed_post = post;
    if (post && c->feed_dbchan.size() != c->n_ch) {          // the reference's initial display state (utils_supersdr.py:599-603, 921, 945)
        ssdr_db2col_chan d;
        memset(&d, 0, sizeof d);
        d.auto_scale = 1; d.low_clip_db = -120.0f; d.high_clip_db = -60.0f; d.dynamic_range = 40.0f;
        c->feed_dbchan.assign(c->n_ch, d);
        ssdr_play_chan pc = {100.0, 0.0};
        c->feed_playchan.assign(c->n_ch, pc);
    }
    const size_t in_b = (size_t)c->n_ch * n_frames * SSDR_FRAME * 4;
    const size_t wire_b = (size_t)c->n_ch * n_frames * SSDR_WIRE_BODY;
    const bool wire = (flags & SSDR_FEED_WIRE) != 0;
    c->feed_wire = wire;
    const size_t wf_b = (size_t)(c->hop == SSDR_NFFT / 2 ? n_frames : n_frames / 2) * c->n_ch * SSDR_NFFT * 2;
    const size_t pcm_b = (size_t)c->n_ch * n_frames * SSDR_FRAME * 2;
    const size_t rssi_b = (size_t)c->n_ch * n_frames * sizeof(float);
    // SSDR_FEED_LAZY_OUT: what comes back to the host is sized for the listeners, not for the receivers
    const bool lazy = (flags & SSDR_FEED_LAZY_OUT) != 0;
    c->feed_lazy = lazy;
    c->feed_lazy_max = c->n_ch < SSDR_FEED_LAZY_MAX ? c->n_ch : SSDR_FEED_LAZY_MAX;
    const size_t out_ch = lazy ? c->feed_lazy_max : c->n_ch;
    const size_t h_wf_b = wf_b / c->n_ch * out_ch, h_pcm_b = pcm_b / c->n_ch * out_ch, h_rssi_b = rssi_b / c->n_ch * out_ch, h_flags_b = out_ch * n_frames;
    c->feed.resize(depth);
    bool ok = hipStreamCreateWithFlags(&c->feed_s_in, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->feed_s_out, hipStreamNonBlocking) == hipSuccess;
    for (auto &s : c->feed) {
        ok = ok && hipHostMalloc(&s.h_in, wire ? wire_b : in_b, hipHostMallocDefault) == hipSuccess;
        if (wire) {
            ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_wire_rssi), h_rssi_b, hipHostMallocDefault) == hipSuccess;
            ok = ok && hipMalloc(&s.d_wire, wire_b + 16) == hipSuccess /* the unpack kernel reads whole dwords */ && hipMalloc(&s.d_wire_rssi, rssi_b) == hipSuccess;
        }
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_wf), h_wf_b, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_pcm), h_pcm_b, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_rssi), h_rssi_b, hipHostMallocDefault) == hipSuccess;
        if (lazy) {
            ok = ok && hipMalloc(&s.d_sel_wf, h_wf_b) == hipSuccess && hipMalloc(&s.d_sel_pcm, h_pcm_b) == hipSuccess;
            ok = ok && hipMalloc(&s.d_sel_rssi, h_rssi_b) == hipSuccess && hipMalloc(&s.d_sel_flags, h_flags_b) == hipSuccess;
            if (wire) ok = ok && hipMalloc(&s.d_sel_wire_rssi, h_rssi_b) == hipSuccess;
        }
        ok = ok && hipMalloc(&s.d_in, in_b) == hipSuccess && hipMalloc(&s.d_wf, wf_b) == hipSuccess;
        ok = ok && hipMalloc(&s.d_pcm, pcm_b) == hipSuccess && hipMalloc(&s.d_rssi, rssi_b) == hipSuccess;
        ok = ok && hipMalloc(&s.d_flags, (size_t)c->n_ch * n_frames) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_flags), h_flags_b, hipHostMallocDefault) == hipSuccess;
        if (post) {
            const size_t color_b = wf_b * 2, db_b = (size_t)c->n_ch * sizeof(ssdr_db2col_chan);
            const size_t play_b = (size_t)c->n_ch * n_frames * 2048 * 2 * sizeof(int16_t);       // sized for the x4 form, either rate fits
            ok = ok && hipMalloc(&s.d_color, color_b) == hipSuccess && hipMalloc(&s.d_dbchan, db_b) == hipSuccess;
            ok = ok && hipMalloc(&s.d_play, play_b) == hipSuccess && hipMalloc(&s.d_mono, play_b / 2) == hipSuccess;
            ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_color), color_b, hipHostMallocDefault) == hipSuccess;
            ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_dbchan), db_b, hipHostMallocDefault) == hipSuccess;
            ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_playchan), (size_t)c->n_ch * sizeof(ssdr_play_chan), hipHostMallocDefault) == hipSuccess;
            ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_play), play_b, hipHostMallocDefault) == hipSuccess;
            ok = ok && hipHostMalloc(reinterpret_cast<void **>(&s.h_mono), play_b / 2, hipHostMallocDefault) == hipSuccess;
        }
        ok = ok && hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.ev_run, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming) == hipSuccess;
    }
    c->feed_frames = n_frames;
    if (!ok) { (void)hipGetLastError(); return SSDR_ENOMEM; }
    undo.armed = false;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_feed_slot(ssdr_ctx *c, void **host_iq) SSDR_GUARD
{
    if (!c || !host_iq) return SSDR_EINVAL;
    if (c->feed.empty() || c->feed_taken) return SSDR_ESTATE;
    if (c->feed_inflight == c->feed.size()) return SSDR_ESTATE;      // collect first: every slot is in flight
    *host_iq = c->feed[c->feed_head].h_in;
    c->feed_taken = true;
    return SSDR_OK;
} SSDR_UNGUARD

// host_in: where the batch lies (the slot's own pinned buffer, or the caller's -- ssdr_feed_submit_from)
static int feed_submit_impl(ssdr_ctx *c, const void *host_in)
{
    HIP_TRY(hipSetDevice(c->device));
    auto &s = c->feed[c->feed_head];
    const uint32_t nf = c->feed_frames;
    if (c->feed_lazy && (c->d_post_sel ? c->n_post : c->n_ch) > c->feed_lazy_max) return SSDR_ESTATE;     // more listeners than the compact rows hold
    if (c->feed_wire)
        HIP_TRY(hipMemcpyAsync(s.d_wire, host_in, (size_t)c->n_ch * nf * SSDR_WIRE_BODY, hipMemcpyHostToDevice, c->feed_s_in));
    else
        HIP_TRY(hipMemcpyAsync(s.d_in, host_in, (size_t)c->n_ch * nf * SSDR_FRAME * 4, hipMemcpyHostToDevice, c->feed_s_in));
    HIP_TRY(hipEventRecord(s.ev_in, c->feed_s_in));
    HIP_TRY(hipStreamWaitEvent(c->stream, s.ev_in, 0));
    if (c->feed_wire) {                                  // header strip + big-endian -> little-endian on the device
        SsdrWireArgs w;
        w.bodies = s.d_wire;
        w.n_ch = c->n_ch;
        w.n_frames = nf;
        w.iq = s.d_in;
        w.ch_stride = (uint64_t)nf * SSDR_FRAME;
        w.rssi = s.d_wire_rssi;
        w.gps = nullptr;
        int rcw;
        if ((rcw = timed_begin(c)) != SSDR_OK) return rcw;
        HIP_TRY(ssdr_launch_iqwire(w, c->stream));
        if ((rcw = timed_end(c, SSDR_K_WIRE)) != SSDR_OK) return rcw;
    }
    // run the two kernels on this slot's buffers: the ctx's own batch pointers are parked meanwhile
    const uint32_t *k_iq = c->d_iq; const uint32_t k_frames = c->in_frames; const bool k_have = c->have_input;
    int16_t *k_wf = c->d_wf_out; const size_t k_wf_lines = c->wf_out_lines; const uint32_t k_ready = c->wf_lines_ready;
    int16_t *k_pcm = c->d_pcm; float *k_rssi = c->d_rssi; const size_t k_af = c->audio_frames; const uint32_t k_arf = c->audio_run_frames;
    uint8_t *k_flags = c->d_flags; const size_t k_ff = c->flags_frames;
    c->d_iq = s.d_in; c->in_frames = nf; c->have_input = true;
    c->d_wf_out = s.d_wf; c->wf_out_lines = c->hop == SSDR_NFFT / 2 ? nf : nf / 2;
    c->d_pcm = s.d_pcm; c->d_rssi = s.d_rssi; c->audio_frames = nf;
    c->d_flags = s.d_flags; c->flags_frames = nf;
    uint32_t lines = 0;
    s.n_avg = c->n_avg;
    int rc = ssdr_run_chain(c, &lines, nullptr);           // the fused superframe kernel where the batch allows it
    if (rc == SSDR_OK) rc = join_audio(c);                  // (or the two stages side by side: what follows reads both results)
    if (rc == SSDR_OK && c->feed_post) rc = [&]() -> int {
        // spectrum_db2col of this batch's lines and play_buffer of its frames, on the slot's buffers, in batch order
        s.n_post = c->n_post;
        s.has_mono = false;
        if (c->n_post == 0) return SSDR_OK;                  // ssdr_set_post_channels with an empty list: nobody is looking
        if (lines) {
            memcpy(s.h_dbchan, c->feed_dbchan.data(), (size_t)c->n_post * sizeof(ssdr_db2col_chan));
            HIP_TRY(hipMemcpyAsync(s.d_dbchan, s.h_dbchan, (size_t)c->n_post * sizeof(ssdr_db2col_chan), hipMemcpyHostToDevice, c->stream));
            SsdrDb2colArgs d;
            d.wf = s.d_wf; d.n_ch = c->n_ch; d.n_lines = lines; d.n_avg = c->n_avg; d.chans = s.d_dbchan; d.color = s.d_color;
            d.sel = c->d_post_sel; d.n_sel = c->n_post;
            int r2;
            if ((r2 = timed_begin(c)) != SSDR_OK) return r2;
            HIP_TRY(ssdr_launch_db2col(d, c->stream));
            if ((r2 = timed_end(c, SSDR_K_DB2COL)) != SSDR_OK) return r2;
            if (c->d_wfdata) { r2 = wfdata_feed(c, s.d_color, lines); if (r2 != SSDR_OK) return r2; }
        }
        memcpy(s.h_playchan, c->feed_playchan.data(), (size_t)c->n_post * sizeof(ssdr_play_chan));
        HIP_TRY(hipMemcpyAsync(c->d_play, s.h_playchan, (size_t)c->n_post * sizeof(ssdr_play_chan), hipMemcpyHostToDevice, c->stream));
        SsdrPlayArgs pa;
        pa.pcm = s.d_pcm; pa.n_ch = c->n_ch; pa.n_frames = nf; pa.chans = c->d_play; pa.taps = c->d_play_taps; pa.hist = c->d_play_hist;
        pa.hist_out = c->d_play_hist_alt; pa.sel = c->d_post_sel; pa.n_sel = c->n_post;
        s.n_post = c->n_post;
        if (c->d_post_sel && c->kiwi_rate == SSDR_RATE)      // the channels outside the selection keep their history
            HIP_TRY(hipMemcpyAsync(c->d_play_hist_alt, c->d_play_hist, (size_t)c->n_ch * 8 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        pa.out = s.d_play; pa.rs_taps = c->d_play_rs_taps; pa.mono = c->recording ? s.d_mono : nullptr;
        s.has_mono = c->recording;
        int r3;
        if ((r3 = timed_begin(c)) != SSDR_OK) return r3;
        HIP_TRY(c->kiwi_rate != SSDR_RATE ? ssdr_launch_play_rs(pa, c->stream) : ssdr_launch_play(pa, c->stream));
        if (c->kiwi_rate == SSDR_RATE) std::swap(c->d_play_hist, c->d_play_hist_alt);
        if ((r3 = timed_end(c, SSDR_K_PLAY)) != SSDR_OK) return r3;
        return SSDR_OK;
    }();
    c->d_iq = k_iq; c->in_frames = k_frames; c->have_input = k_have;
    c->d_wf_out = k_wf; c->wf_out_lines = k_wf_lines; c->wf_lines_ready = k_ready;
    c->d_pcm = k_pcm; c->d_rssi = k_rssi; c->audio_frames = k_af; c->audio_run_frames = k_arf;
    c->d_flags = k_flags; c->flags_frames = k_ff;
    if (rc != SSDR_OK) return rc;
    s.lines = lines;
    // what goes back: every channel's rows, or (SSDR_FEED_LAZY_OUT) the selected channels' rows gathered into compact ones
    const int16_t *o_wf = s.d_wf, *o_pcm = s.d_pcm;
    const float *o_rssi = s.d_rssi, *o_wire_rssi = s.d_wire_rssi;
    const uint8_t *o_flags = s.d_flags;
    size_t o_ch = c->n_ch;
    s.n_sel = c->n_ch;
    if (c->feed_lazy) {
        SsdrGatherArgs g;
        g.wf = s.d_wf; g.pcm = s.d_pcm; g.rssi = s.d_rssi; g.flags = s.d_flags; g.wire_rssi = c->feed_wire ? s.d_wire_rssi : nullptr;
        g.wf_out = s.d_sel_wf; g.pcm_out = s.d_sel_pcm; g.rssi_out = s.d_sel_rssi; g.flags_out = s.d_sel_flags; g.wire_rssi_out = s.d_sel_wire_rssi;
        g.sel = c->d_post_sel; g.n_sel = c->d_post_sel ? c->n_post : c->n_ch; g.n_ch = c->n_ch; g.n_lines = lines; g.n_frames = nf;
        HIP_TRY(ssdr_launch_gather(g, c->stream));
        o_wf = s.d_sel_wf; o_pcm = s.d_sel_pcm; o_rssi = s.d_sel_rssi; o_wire_rssi = s.d_sel_wire_rssi; o_flags = s.d_sel_flags;
        o_ch = g.n_sel;
        s.n_sel = g.n_sel;
    }
    HIP_TRY(hipEventRecord(s.ev_run, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->feed_s_out, s.ev_run, 0));
    if (lines && o_ch)
        HIP_TRY(hipMemcpyAsync(s.h_wf, o_wf, (size_t)lines * o_ch * SSDR_NFFT * 2, hipMemcpyDeviceToHost, c->feed_s_out));
    if (o_ch) {
        HIP_TRY(hipMemcpyAsync(s.h_pcm, o_pcm, o_ch * nf * SSDR_FRAME * 2, hipMemcpyDeviceToHost, c->feed_s_out));
        HIP_TRY(hipMemcpyAsync(s.h_rssi, o_rssi, o_ch * nf * sizeof(float), hipMemcpyDeviceToHost, c->feed_s_out));
        if (c->feed_wire)
            HIP_TRY(hipMemcpyAsync(s.h_wire_rssi, o_wire_rssi, o_ch * nf * sizeof(float), hipMemcpyDeviceToHost, c->feed_s_out));
        HIP_TRY(hipMemcpyAsync(s.h_flags, o_flags, o_ch * nf, hipMemcpyDeviceToHost, c->feed_s_out));
    }
    if (c->feed_post) {
        const size_t per_frame = c->kiwi_rate != SSDR_RATE ? (size_t)SSDR_RS_OUT_PER_FRAME : 2048;
        if (lines && c->n_post) {
            HIP_TRY(hipMemcpyAsync(s.h_color, s.d_color, (size_t)lines * c->n_post * SSDR_NFFT * sizeof(float), hipMemcpyDeviceToHost, c->feed_s_out));
            HIP_TRY(hipMemcpyAsync(s.h_dbchan, s.d_dbchan, (size_t)c->n_post * sizeof(ssdr_db2col_chan), hipMemcpyDeviceToHost, c->feed_s_out));
        }
        if (c->n_post) {
            HIP_TRY(hipMemcpyAsync(s.h_play, s.d_play, (size_t)c->n_post * nf * per_frame * 2 * sizeof(int16_t), hipMemcpyDeviceToHost, c->feed_s_out));
            if (s.has_mono)
                HIP_TRY(hipMemcpyAsync(s.h_mono, s.d_mono, (size_t)c->n_post * nf * per_frame * sizeof(int16_t), hipMemcpyDeviceToHost, c->feed_s_out));
        }
    }
    HIP_TRY(hipEventRecord(s.ev_out, c->feed_s_out));
    c->feed_head = (c->feed_head + 1) % (uint32_t)c->feed.size();
    c->feed_inflight++;
    c->feed_taken = false;
    return SSDR_OK;
}

int ssdr_feed_submit(ssdr_ctx *c) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (c->feed.empty() || !c->feed_taken) return SSDR_ESTATE;
    return feed_submit_impl(c, c->feed[c->feed_head].h_in);
} SSDR_UNGUARD

int ssdr_feed_submit_from(ssdr_ctx *c, const void *host_in) SSDR_GUARD
{
    if (!c || !host_in) return SSDR_EINVAL;
    if (c->feed.empty() || c->feed_taken) return SSDR_ESTATE;
    if (c->feed_inflight == c->feed.size()) return SSDR_ESTATE;      // collect first: every slot is in flight
    return feed_submit_impl(c, host_in);
} SSDR_UNGUARD

int ssdr_host_alloc(ssdr_ctx *c, uint64_t bytes, void **out) SSDR_GUARD
{
    if (!c || !out || bytes == 0) return SSDR_EINVAL;
    *out = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_host_free(ssdr_ctx *c, void *ptr) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (!ptr) return SSDR_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostFree(ptr));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_feed_post(ssdr_ctx *c, const ssdr_db2col_chan *chans, const ssdr_play_chan *play) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (c->feed.empty() || !c->feed_post) return SSDR_ESTATE;
    if (chans) std::copy(chans, chans + c->n_post, c->feed_dbchan.begin());        // (n_post entries, in the order of the selection)
    if (play) std::copy(play, play + c->n_post, c->feed_playchan.begin());
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_feed_collect_post(ssdr_ctx *c, float **color, ssdr_db2col_chan **chans, int16_t **play, int16_t **mono) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (c->feed.empty() || !c->feed_post || c->feed_last < 0) return SSDR_ESTATE;
    auto &s = c->feed[c->feed_last];
    if (color) *color = s.lines ? s.h_color : nullptr;
    if (chans) *chans = s.lines ? s.h_dbchan : nullptr;
    if (play) *play = s.h_play;
    if (mono) *mono = s.has_mono ? s.h_mono : nullptr;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_feed_collect(ssdr_ctx *c, int16_t **wf_sum, uint32_t *lines, int16_t **pcm, float **rssi, float **wire_rssi,
                      uint8_t **flags, uint32_t *n_avg) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (c->feed.empty() || c->feed_inflight == 0) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    auto &s = c->feed[c->feed_tail];
    HIP_TRY(hipEventSynchronize(s.ev_out));
    if (wf_sum) *wf_sum = s.h_wf;
    if (lines) *lines = s.lines;
    if (pcm) *pcm = s.h_pcm;
    if (rssi) *rssi = s.h_rssi;
    if (wire_rssi) *wire_rssi = s.h_wire_rssi;            // NULL unless the feed was opened with SSDR_FEED_WIRE
    if (flags) *flags = s.h_flags;
    if (n_avg) *n_avg = s.n_avg;
    c->feed_last = (int)c->feed_tail;
    c->feed_tail = (c->feed_tail + 1) % (uint32_t)c->feed.size();
    c->feed_inflight--;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_feed_collect_lazy(ssdr_ctx *c, uint32_t *n_sel, int16_t **d_wf_sum, int16_t **d_pcm, float **d_rssi, uint8_t **d_flags) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    if (c->feed.empty() || c->feed_last < 0) return SSDR_ESTATE;
    auto &s = c->feed[c->feed_last];
    if (n_sel) *n_sel = s.n_sel;
    if (d_wf_sum) *d_wf_sum = s.d_wf;
    if (d_pcm) *d_pcm = s.d_pcm;
    if (d_rssi) *d_rssi = s.d_rssi;
    if (d_flags) *d_flags = s.d_flags;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_wf_device(ssdr_ctx *c, int16_t **ptr, uint32_t *lines) SSDR_GUARD
{
    if (!c || !ptr) return SSDR_EINVAL;
    *ptr = c->d_wf_out;
    if (lines) *lines = c->wf_lines_ready;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_copy_from_device(ssdr_ctx *c, void *host_dst, const void *device_src, uint64_t bytes) SSDR_GUARD
{
    if (!c || !host_dst || !device_src) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    HIP_TRY(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_audio_device(ssdr_ctx *c, int16_t **pcm, float **rssi) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }       // a consumer ordered behind the ctx stream sees the audio stage too
    if (pcm) *pcm = c->d_pcm;
    if (rssi) *rssi = c->d_rssi;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_get_consts(ssdr_ctx *c, uint32_t first, uint32_t count, ssdr_chan_consts *consts, float *taps) SSDR_GUARD
{
    if (!c || (uint64_t)first + count > c->n_ch) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    if (consts) HIP_TRY(hipMemcpyAsync(consts, c->d_consts + first, count * sizeof(ssdr_chan_consts), hipMemcpyDeviceToHost, c->stream));
    if (taps) HIP_TRY(hipMemcpyAsync(taps, c->d_taps + (size_t)first * SSDR_NTAP_MAX, (size_t)count * SSDR_NTAP_MAX * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_get_state(ssdr_ctx *c, uint32_t first, uint32_t count, ssdr_chan_state *state, int16_t *hist) SSDR_GUARD
{
    if (!c || (uint64_t)first + count > c->n_ch) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    if (state) HIP_TRY(hipMemcpyAsync(state, c->d_state + first, count * sizeof(ssdr_chan_state), hipMemcpyDeviceToHost, c->stream));
    if (hist) HIP_TRY(hipMemcpyAsync(hist, c->d_hist + (size_t)first * SSDR_HIST, (size_t)count * SSDR_HIST * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_state(ssdr_ctx *c, uint32_t first, uint32_t count, const ssdr_chan_state *state, const int16_t *hist) SSDR_GUARD
{
    if (!c || (uint64_t)first + count > c->n_ch) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    if (state) {
        HIP_TRY(hipMemcpyAsync(c->d_state + first, state, count * sizeof(ssdr_chan_state), hipMemcpyHostToDevice, c->stream));
        c->audio_started = true;             // a restored stream is live: ssdr_set_params must not re-seed its state
    }
    if (hist) HIP_TRY(hipMemcpyAsync(c->d_hist + (size_t)first * SSDR_HIST, hist, (size_t)count * SSDR_HIST * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

// ---- checkpoint: everything a stream carries from one call to the next, as one blob -------------------------
// header | consts | taps | state | hist | waterfall partial sums | play_buffer history (zeros if never used)
struct SsdrCkptHeader {
    uint32_t magic, version, n_ch, n_avg, wf_phase, audio_started, kiwi_rate, has_play;
    uint64_t synth_sample0;
    uint32_t hop, decim;
};
static const uint32_t kCkptMagic = 0x52445353u;          // "SSDR"
// 4: the channels' compiled constants and taps in the blob are informative only -- loading recompiles them from the saved
// ssdr_chan_params at the blob's decimation and rate, so a blob never carries a constants layout of another build into the
// kernels (version 3 blobs, whose `kfm` word meant padding, are refused)
static const uint32_t kCkptVersion = 4;

int ssdr_checkpoint_size(ssdr_ctx *c, uint64_t *bytes) SSDR_GUARD
{
    if (!c || !bytes) return SSDR_EINVAL;
    const uint64_t n = c->n_ch;
    *bytes = sizeof(SsdrCkptHeader) + n * (sizeof(ssdr_chan_consts) + SSDR_NTAP_MAX * sizeof(float) + sizeof(ssdr_chan_state) +
                                           SSDR_HIST * 4 + SSDR_NFFT * 2 + 8 * sizeof(double) + (SSDR_NFFT / 2) * 4 +
                                           sizeof(ssdr_chan_params));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_checkpoint_save(ssdr_ctx *c, void *blob) SSDR_GUARD
{
    if (!c || !blob) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    const size_t n = c->n_ch;
    if (c->zoom > 1) return SSDR_ESTATE;                     // the zoomed waterfall stream (phase, history, centres) is not part of the blob
    SsdrCkptHeader h = {kCkptMagic, kCkptVersion, c->n_ch, c->n_avg, c->wf_phase, c->audio_started ? 1u : 0u, c->kiwi_rate,
                        c->d_play_hist ? 1u : 0u, c->synth_sample0, c->hop, c->decim};
    char *p = static_cast<char *>(blob);
    memcpy(p, &h, sizeof h); p += sizeof h;
    const hipMemcpyKind d2h = hipMemcpyDeviceToHost;
    HIP_TRY(hipMemcpyAsync(p, c->d_consts, n * sizeof(ssdr_chan_consts), d2h, c->stream)); p += n * sizeof(ssdr_chan_consts);
    HIP_TRY(hipMemcpyAsync(p, c->d_taps, n * SSDR_NTAP_MAX * sizeof(float), d2h, c->stream)); p += n * SSDR_NTAP_MAX * sizeof(float);
    HIP_TRY(hipMemcpyAsync(p, c->d_state, n * sizeof(ssdr_chan_state), d2h, c->stream)); p += n * sizeof(ssdr_chan_state);
    HIP_TRY(hipMemcpyAsync(p, c->d_hist, n * SSDR_HIST * 4, d2h, c->stream)); p += n * SSDR_HIST * 4;
    HIP_TRY(hipMemcpyAsync(p, c->d_wf_acc[c->wf_acc_cur], n * SSDR_NFFT * 2, d2h, c->stream)); p += n * SSDR_NFFT * 2;
    if (c->d_play_hist) HIP_TRY(hipMemcpyAsync(p, c->d_play_hist, n * 8 * sizeof(double), d2h, c->stream));
    else memset(p, 0, n * 8 * sizeof(double));
    p += n * 8 * sizeof(double);
    if (c->hop == SSDR_NFFT / 2) HIP_TRY(hipMemcpyAsync(p, c->d_wf_tail, n * (SSDR_NFFT / 2) * 4, d2h, c->stream));
    else memset(p, 0, n * (SSDR_NFFT / 2) * 4);
    p += n * (SSDR_NFFT / 2) * 4;
    memcpy(p, c->h_params.data(), n * sizeof(ssdr_chan_params));          // what the constants were compiled from
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_checkpoint_load(ssdr_ctx *c, const void *blob, uint64_t bytes) SSDR_GUARD
{
    if (!c || !blob) return SSDR_EINVAL;
    uint64_t want = 0;
    (void)ssdr_checkpoint_size(c, &want);
    if (bytes != want) return SSDR_EINVAL;                    // a blob of another channel count, another version, or cut short
    SsdrCkptHeader h;
    memcpy(&h, blob, sizeof h);
    if (h.magic != kCkptMagic || h.version != kCkptVersion || h.n_ch != c->n_ch || h.n_avg < 1 || h.n_avg > 100 || h.wf_phase >= h.n_avg ||
        (h.hop != SSDR_NFFT && h.hop != SSDR_NFFT / 2) || (h.decim != 1 && h.decim != 2 && h.decim != 4) ||
        (h.kiwi_rate != SSDR_RATE && h.kiwi_rate != SSDR_RATE_WIDE))
        return SSDR_EINVAL;
    const size_t n = c->n_ch;
    const char *params_at = static_cast<const char *>(blob) + sizeof h + n * (sizeof(ssdr_chan_consts) + SSDR_NTAP_MAX * sizeof(float) +
                            sizeof(ssdr_chan_state) + SSDR_HIST * 4 + SSDR_NFFT * 2 + 8 * sizeof(double) + (SSDR_NFFT / 2) * 4);
    // what the kernels run on is compiled HERE from the saved parameters (at the blob's decimation and rate): a damaged or foreign
    // parameter set is refused by the compiler before anything is touched
    std::vector<ssdr_chan_params> prm(n);
    memcpy(prm.data(), params_at, n * sizeof(ssdr_chan_params));
    std::vector<ssdr_chan_consts> kc(n);
    std::vector<float> ktaps(n * SSDR_NTAP_MAX);
    for (size_t i = 0; i < n; i++)
        if (ssdr_compile_params_host(&prm[i], &kc[i], ktaps.data() + i * SSDR_NTAP_MAX, h.decim, h.kiwi_rate) != SSDR_OK) return SSDR_EINVAL;
    if (!c->feed.empty() || c->zoom > 1) return SSDR_ESTATE;
    std::vector<double> play_hist;                          // play_buffer state that arrives before its buffers exist: applied at first use
    if (h.has_play && !c->d_play_hist) {
        const double *q = reinterpret_cast<const double *>(static_cast<const char *>(blob) + sizeof h + n * (sizeof(ssdr_chan_consts) +
                          SSDR_NTAP_MAX * sizeof(float) + sizeof(ssdr_chan_state) + SSDR_HIST * 4 + SSDR_NFFT * 2));
        play_hist.assign(q, q + n * 8);
    }
    HIP_TRY(hipSetDevice(c->device));
    { int rch = ssdr_set_hop(c, h.hop); if (rch != SSDR_OK) return rch; }
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    const char *p = static_cast<const char *>(blob) + sizeof h;
    const hipMemcpyKind h2d = hipMemcpyHostToDevice;
    memcpy(c->h_consts.data(), kc.data(), n * sizeof(ssdr_chan_consts));
    HIP_TRY(hipMemcpyAsync(c->d_consts, kc.data(), n * sizeof(ssdr_chan_consts), h2d, c->stream)); p += n * sizeof(ssdr_chan_consts);
    HIP_TRY(hipMemcpyAsync(c->d_taps, ktaps.data(), n * SSDR_NTAP_MAX * sizeof(float), h2d, c->stream)); p += n * SSDR_NTAP_MAX * sizeof(float);
    HIP_TRY(hipMemcpyAsync(c->d_state, p, n * sizeof(ssdr_chan_state), h2d, c->stream)); p += n * sizeof(ssdr_chan_state);
    HIP_TRY(hipMemcpyAsync(c->d_hist, p, n * SSDR_HIST * 4, h2d, c->stream)); p += n * SSDR_HIST * 4;
    HIP_TRY(hipMemcpyAsync(c->d_wf_acc[c->wf_acc_cur], p, n * SSDR_NFFT * 2, h2d, c->stream)); p += n * SSDR_NFFT * 2;
    if (h.has_play && c->d_play_hist) HIP_TRY(hipMemcpyAsync(c->d_play_hist, p, n * 8 * sizeof(double), h2d, c->stream));
    if (h.hop == SSDR_NFFT / 2)
        HIP_TRY(hipMemcpyAsync(c->d_wf_tail, p + n * 8 * sizeof(double), n * (SSDR_NFFT / 2) * 4, h2d, c->stream));
    memcpy(c->h_params.data(), p + n * 8 * sizeof(double) + n * (SSDR_NFFT / 2) * 4, n * sizeof(ssdr_chan_params));
    c->decim = h.decim;
    c->have_input = false;                                  // a batch pushed before the load belongs to the old streams
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_avg = h.n_avg;
    c->wf_phase = h.wf_phase;
    c->audio_started = h.audio_started != 0;
    c->kiwi_rate = h.kiwi_rate;
    c->synth_sample0 = h.synth_sample0;
    c->summary_dirty = true;
    c->chan_list_dirty = true;
    c->pending_play_hist.swap(play_hist);                   // (every host allocation of the load was made before anything was touched)
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_selftest_quantiser(ssdr_ctx *c, uint64_t *mismatches) SSDR_GUARD
{
    if (!c || !mismatches) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->d_scratch, 0, 8, c->stream));
    HIP_TRY(ssdr_launch_quant_selftest(c->d_thr, c->d_lut, c->d_scratch, c->stream));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, c->d_scratch, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *mismatches = v;
    return SSDR_OK;
} SSDR_UNGUARD

// kiwi_waterfall.run's feeding of wf_data (utils_supersdr.py:893-897) for `lines` colour lines [lines][n_ch][1024] on
// the device: each line joins the 3-deep queue (a full queue drops its oldest entry first); from the 4th line on the
// oldest queued line becomes row 0 and the rows scroll down by one.
static int wfdata_feed(ssdr_ctx *c, const float *color, uint32_t lines)
{
    const size_t line = (size_t)c->n_post * SSDR_NFFT;       // the rows hold the selected channels' lines, in selection order
    for (uint32_t i = 0; i < lines; i++) {
        c->wfdata_seen++;                                                        // run_index += 1 (:889)
        if (c->wfpend_n == 3) { c->wfpend_first++; c->wfpend_n--; }              // deque(maxlen=3).appendleft on a full deque
        const uint64_t arrival = c->wfpend_first + c->wfpend_n;
        HIP_TRY(hipMemcpyAsync(c->d_wfpend + (arrival % 3) * line, color + (size_t)i * line, line * sizeof(float),
                               hipMemcpyDeviceToDevice, c->stream));
        c->wfpend_n++;
        if (c->wfdata_seen > 3) {                                                // run_index > wf_buffer_len
            c->wfdata_head = (c->wfdata_head + c->wfdata_rows - 1) % c->wfdata_rows;   // scroll one row down
            HIP_TRY(hipMemcpyAsync(c->d_wfdata + (size_t)c->wfdata_head * line, c->d_wfpend + (c->wfpend_first % 3) * line,
                                   line * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            c->wfpend_first++;
            c->wfpend_n--;
        }
    }
    return SSDR_OK;
}

int ssdr_set_post_channels(ssdr_ctx *c, const uint32_t *channels, uint32_t count) SSDR_GUARD
{
    if (!c || (count && !channels && count != c->n_ch) || count > c->n_ch) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    const bool all = channels == nullptr;                    // (NULL, 0) or (NULL, n_ch): every channel, the default
    if (!all)
        for (uint32_t i = 0; i < count; i++)
            if (channels[i] >= c->n_ch || (i && channels[i] <= channels[i - 1])) return SSDR_EINVAL;     // ascending, unique
    if (!all) {
        if (!c->d_post_sel) HIP_TRY(hipMalloc(&c->d_post_sel, (size_t)c->n_ch * sizeof(uint32_t)));
        // in stream order behind the batches already queued with the previous selection
        if (count) HIP_TRY(hipMemcpyAsync(c->d_post_sel, channels, (size_t)count * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));             // `channels` is the caller's
        c->n_post = count;
    } else {
        if (c->d_post_sel) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_post_sel)); c->d_post_sel = nullptr; }
        c->n_post = c->n_ch;
    }
    // the device copy of wf_data holds other channels' rows now: it starts over (kiwi_waterfall.__init__'s np.zeros, utils_supersdr.py:692)
    if (c->d_wfdata) {
        HIP_TRY(hipMemsetAsync(c->d_wfdata, 0, (size_t)c->wfdata_rows * c->n_ch * SSDR_NFFT * sizeof(float), c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->wfdata_head = 0; c->wfdata_seen = c->wfpend_first = 0; c->wfpend_n = 0;
    }
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_run_db2col(ssdr_ctx *c, ssdr_db2col_chan *chans, float *color_out, int out_is_device) SSDR_GUARD
{
    if (!c || !chans) return SSDR_EINVAL;
    if (!c->d_wf_out && c->wf_lines_ready) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t lines = c->wf_lines_ready;
    if (!c->d_db2col) HIP_TRY(hipMalloc(&c->d_db2col, (size_t)c->n_ch * sizeof(ssdr_db2col_chan)));
    if (lines == 0) return SSDR_OK;
    if (c->color_lines < lines) {
        if (c->d_color) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_color)); c->d_color = nullptr; c->color_lines = 0; }
        HIP_TRY(hipMalloc(&c->d_color, (size_t)lines * c->n_ch * SSDR_NFFT * sizeof(float)));
        c->color_lines = lines;
    }
    if (c->n_post == 0) return SSDR_OK;                       // an empty selection: nobody is looking
    HIP_TRY(hipMemcpyAsync(c->d_db2col, chans, (size_t)c->n_post * sizeof(ssdr_db2col_chan), hipMemcpyHostToDevice, c->stream));
    SsdrDb2colArgs a;
    a.wf = c->d_wf_out;
    a.n_ch = c->n_ch;
    a.n_lines = lines;
    a.n_avg = c->n_avg;
    a.chans = c->d_db2col;
    a.color = c->d_color;
    a.sel = c->d_post_sel;
    a.n_sel = c->n_post;
    int rc;
    if ((rc = timed_begin(c)) != SSDR_OK) return rc;
    HIP_TRY(ssdr_launch_db2col(a, c->stream));
    if ((rc = timed_end(c, SSDR_K_DB2COL)) != SSDR_OK) return rc;
    if (c->d_wfdata) { int rcw = wfdata_feed(c, c->d_color, lines); if (rcw != SSDR_OK) return rcw; }
    HIP_TRY(hipMemcpyAsync(chans, c->d_db2col, (size_t)c->n_post * sizeof(ssdr_db2col_chan), hipMemcpyDeviceToHost, c->stream));
    if (color_out)
        HIP_TRY(hipMemcpyAsync(color_out, c->d_color, (size_t)lines * c->n_post * SSDR_NFFT * sizeof(float),
                               out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_db2col_line(ssdr_ctx *c, const int16_t *wf_sum, uint32_t n_avg, ssdr_db2col_chan *chan, float *color_out) SSDR_GUARD
{
    if (!c || !wf_sum || !chan || !color_out || n_avg < 1 || n_avg > 100) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->d_line1) {
        HIP_TRY(hipMalloc(&c->d_line1, SSDR_NFFT * sizeof(int16_t)));
        HIP_TRY(hipMalloc(&c->d_dbchan1, sizeof(ssdr_db2col_chan)));
        HIP_TRY(hipMalloc(&c->d_color1, SSDR_NFFT * sizeof(float)));
    }
    HIP_TRY(hipMemcpyAsync(c->d_line1, wf_sum, SSDR_NFFT * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_dbchan1, chan, sizeof(ssdr_db2col_chan), hipMemcpyHostToDevice, c->stream));
    SsdrDb2colArgs a;
    a.wf = c->d_line1;
    a.n_ch = 1;
    a.n_lines = 1;
    a.n_avg = n_avg;
    a.chans = c->d_dbchan1;
    a.color = c->d_color1;
    a.sel = nullptr;
    a.n_sel = 1;
    int rc;
    if ((rc = timed_begin(c)) != SSDR_OK) return rc;
    HIP_TRY(ssdr_launch_db2col(a, c->stream));
    if ((rc = timed_end(c, SSDR_K_DB2COL)) != SSDR_OK) return rc;
    HIP_TRY(hipMemcpyAsync(chan, c->d_dbchan1, sizeof(ssdr_db2col_chan), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(color_out, c->d_color1, SSDR_NFFT * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_output_checksum(ssdr_ctx *c, uint64_t sums[3]) SSDR_GUARD
{
    if (!c || !sums) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    HIP_TRY(hipMemsetAsync(c->d_scratch, 0, 24, c->stream));
    if (c->d_wf_out && c->wf_lines_ready)
        HIP_TRY(ssdr_launch_checksum(c->d_wf_out, (uint64_t)c->wf_lines_ready * c->n_ch * (SSDR_NFFT / 2), c->d_scratch, c->stream));
    if (c->d_pcm && c->audio_run_frames) {
        HIP_TRY(ssdr_launch_checksum(c->d_pcm, (uint64_t)c->n_ch * c->audio_run_frames * (SSDR_FRAME / 2), c->d_scratch + 1, c->stream));
        HIP_TRY(ssdr_launch_checksum(c->d_rssi, (uint64_t)c->n_ch * c->audio_run_frames, c->d_scratch + 2, c->stream));
    }
    unsigned long long v[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(v, c->d_scratch, 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++) sums[i] = v[i];
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_wfdata_rows(ssdr_ctx *c, uint32_t rows) SSDR_GUARD
{
    if (!c || rows > 4096) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_wfdata) { HIP_TRY(hipFree(c->d_wfdata)); c->d_wfdata = nullptr; }
    if (c->d_wfpend) { HIP_TRY(hipFree(c->d_wfpend)); c->d_wfpend = nullptr; }
    c->wfdata_rows = rows;
    c->wfdata_head = 0;
    c->wfdata_seen = c->wfpend_first = 0;
    c->wfpend_n = 0;
    if (rows) {
        const size_t line = (size_t)c->n_ch * SSDR_NFFT * sizeof(float);
        if (hipMalloc(&c->d_wfdata, rows * line) != hipSuccess || hipMalloc(&c->d_wfpend, 3 * line) != hipSuccess) {
            if (c->d_wfdata) (void)hipFree(c->d_wfdata);
            c->d_wfdata = c->d_wfpend = nullptr;
            c->wfdata_rows = 0;
            return SSDR_ENOMEM;
        }
        HIP_TRY(hipMemsetAsync(c->d_wfdata, 0, rows * line, c->stream));         // wf_data = np.zeros (:692)
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_push_color_lines(ssdr_ctx *c, const float *color, uint32_t lines, int color_is_device) SSDR_GUARD
{
    if (!c || (!color && lines)) return SSDR_EINVAL;
    if (!c->d_wfdata) return SSDR_ESTATE;
    if (lines == 0) return SSDR_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)lines * c->n_post * SSDR_NFFT;
    const float *src = color;
    if (!color_is_device) {
        if (c->color_lines < lines) {
            if (c->d_color) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_color)); c->d_color = nullptr; c->color_lines = 0; }
            HIP_TRY(hipMalloc(&c->d_color, n * sizeof(float)));
            c->color_lines = lines;
        }
        HIP_TRY(hipMemcpyAsync(c->d_color, color, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        src = c->d_color;
    }
    int rc = wfdata_feed(c, src, lines);
    if (rc != SSDR_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_wfdata_white_flag(ssdr_ctx *c, uint32_t first, uint32_t count) SSDR_GUARD
{
    if (!c || first + count > c->n_post || first + count < first) return SSDR_EINVAL;      // (positions in the selection)
    if (!c->d_wfdata) return SSDR_ESTATE;
    if (count == 0) return SSDR_OK;
    HIP_TRY(hipSetDevice(c->device));
    const float white = 255.0f;                                                  // np.ones_like(wf_color) * 255 (:876)
    uint32_t bits;
    memcpy(&bits, &white, 4);
    float *row0 = c->d_wfdata + ((size_t)c->wfdata_head * c->n_post + first) * SSDR_NFFT;
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(row0), (int)bits, (size_t)count * SSDR_NFFT, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_run_trace(ssdr_ctx *c, uint32_t t_avg, uint32_t spectrum_height, double *trace_out, int32_t *y_out, int out_is_device) SSDR_GUARD
{
    if (!c || t_avg == 0) return SSDR_EINVAL;
    if (!c->d_wfdata) return SSDR_ESTATE;
    if (t_avg > c->wfdata_rows) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->n_post * SSDR_NFFT;
    if (!n) return SSDR_OK;
    if (!c->d_trace) {
        HIP_TRY(hipMalloc(&c->d_trace, (size_t)c->n_ch * SSDR_NFFT * sizeof(double)));
        HIP_TRY(hipMalloc(&c->d_trace_y, (size_t)c->n_ch * SSDR_NFFT * sizeof(int32_t)));
    }
    SsdrTraceArgs a;
    a.ring = c->d_wfdata;
    a.n_ch = c->n_post;
    a.rows = c->wfdata_rows;
    a.head = c->wfdata_head;
    a.t_avg = t_avg;
    a.spectrum_height = spectrum_height;
    a.trace = c->d_trace;
    a.y = c->d_trace_y;
    int rc;
    if ((rc = timed_begin(c)) != SSDR_OK) return rc;
    HIP_TRY(ssdr_launch_trace(a, c->stream));
    if ((rc = timed_end(c, SSDR_K_TRACE)) != SSDR_OK) return rc;
    const hipMemcpyKind kind = out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (trace_out) HIP_TRY(hipMemcpyAsync(trace_out, c->d_trace, n * sizeof(double), kind, c->stream));
    if (y_out) HIP_TRY(hipMemcpyAsync(y_out, c->d_trace_y, n * sizeof(int32_t), kind, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_run_smeter(ssdr_ctx *c, ssdr_smeter_chan *chans, const double *rssi_in, double fps) SSDR_GUARD
{
    if (!c || !chans || !(fps > 0.0)) return SSDR_EINVAL;
    if (!rssi_in && (!c->d_rssi || c->audio_frames == 0 || c->audio_run_frames == 0)) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    if (!c->d_smeter) {
        HIP_TRY(hipMalloc(&c->d_smeter, (size_t)c->n_ch * sizeof(ssdr_smeter_chan)));
        HIP_TRY(hipMalloc(&c->d_smeter_in, (size_t)c->n_ch * sizeof(double)));
    }
    HIP_TRY(hipMemcpyAsync(c->d_smeter, chans, (size_t)c->n_ch * sizeof(ssdr_smeter_chan), hipMemcpyHostToDevice, c->stream));
    if (rssi_in) HIP_TRY(hipMemcpyAsync(c->d_smeter_in, rssi_in, (size_t)c->n_ch * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SsdrSmeterArgs a;
    a.chans = c->d_smeter;
    a.rssi = c->d_rssi;
    a.rssi_in = rssi_in ? c->d_smeter_in : nullptr;
    a.n_ch = c->n_ch;
    a.n_frames = c->audio_run_frames;
    a.fps = fps;
    int rc;
    if ((rc = timed_begin(c)) != SSDR_OK) return rc;
    HIP_TRY(ssdr_launch_smeter(a, c->stream));
    if ((rc = timed_end(c, SSDR_K_SMETER)) != SSDR_OK) return rc;
    HIP_TRY(hipMemcpyAsync(chans, c->d_smeter, (size_t)c->n_ch * sizeof(ssdr_smeter_chan), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_kiwi_rate(ssdr_ctx *c, uint32_t kiwi_rate) SSDR_GUARD
{
    if (!c || (kiwi_rate != SSDR_RATE && kiwi_rate != SSDR_RATE_WIDE)) return SSDR_EINVAL;
    if (kiwi_rate == c->kiwi_rate) return SSDR_OK;
    if (!c->feed.empty()) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    // the rate of the play-back stage AND of the IQ the channels receive: every channel's constants (NCO steps, filter,
    // AGC time constants, NBFM scale) are compiled for it, and streams of the old rate mean nothing at the new one
    std::vector<ssdr_chan_params> all = c->h_params;
    const uint32_t keep = c->kiwi_rate;
    c->kiwi_rate = kiwi_rate;
    int rc = ssdr_set_params(c, 0, c->n_ch, all.data());
    if (rc == SSDR_OK) {
        c->have_input = false;
        rc = ssdr_reset_state(c, 0, c->n_ch);
        if (rc == SSDR_OK) rc = zoom_restart(c, 0, c->n_ch);
    }
    if (rc != SSDR_OK) {                                          // all or nothing, as ssdr_set_decimation
        c->kiwi_rate = keep;
        (void)ssdr_set_params(c, 0, c->n_ch, all.data());
        (void)ssdr_reset_state(c, 0, c->n_ch);
    }
    return rc;
} SSDR_UNGUARD

int ssdr_playbuffer_frame_len(ssdr_ctx *c, uint32_t *samples_per_frame) SSDR_GUARD
{
    if (!c || !samples_per_frame) return SSDR_EINVAL;
    *samples_per_frame = (c->kiwi_rate == SSDR_RATE) ? 2048u : (uint32_t)SSDR_RS_OUT_PER_FRAME;   // int(512 * SAMPLE_RATIO) (:1211)
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_run_playbuffer(ssdr_ctx *c, const ssdr_play_chan *chans, int16_t *out, int out_is_device) SSDR_GUARD
{
    if (!c || !chans) return SSDR_EINVAL;
    if (!c->d_pcm || c->audio_run_frames == 0 || c->audio_frames < c->audio_run_frames) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    { int rcj = join_audio(c); if (rcj != SSDR_OK) return rcj; }
    const uint32_t nf = c->audio_run_frames;
    const bool wide = c->kiwi_rate != SSDR_RATE;                  // SAMPLE_RATIO % 1 != 0 (:1125)
    const size_t per_frame = wide ? (size_t)SSDR_RS_OUT_PER_FRAME : 2048;
    { int rcp = ensure_play(c); if (rcp != SSDR_OK) return rcp; }
    if (c->play_frames < nf) {          // sized for the longer (x4) form, either path fits
        if (c->d_play_out) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_play_out)); c->d_play_out = nullptr; c->play_frames = 0; }
        HIP_TRY(hipMalloc(&c->d_play_out, (size_t)c->n_ch * nf * 2048 * 2 * sizeof(int16_t)));
        c->play_frames = nf;
    }
    if (c->recording && c->play_mono_frames < nf) {
        if (c->d_play_mono) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_play_mono)); c->d_play_mono = nullptr; c->play_mono_frames = 0; }
        HIP_TRY(hipMalloc(&c->d_play_mono, (size_t)c->n_ch * nf * 2048 * sizeof(int16_t)));
        c->play_mono_frames = nf;
    }
    if (c->n_post == 0) { c->play_run_frames = 0; return SSDR_OK; }
    HIP_TRY(hipMemcpyAsync(c->d_play, chans, (size_t)c->n_post * sizeof(ssdr_play_chan), hipMemcpyHostToDevice, c->stream));
    if (c->d_post_sel && !wide)          // the channels outside the selection keep their history
        HIP_TRY(hipMemcpyAsync(c->d_play_hist_alt, c->d_play_hist, (size_t)c->n_ch * 8 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    SsdrPlayArgs a;
    a.pcm = c->d_pcm;
    a.n_ch = c->n_ch;
    a.n_frames = nf;
    a.chans = c->d_play;
    a.taps = c->d_play_taps;
    a.hist = c->d_play_hist;
    a.hist_out = c->d_play_hist_alt;
    a.sel = c->d_post_sel;
    a.n_sel = c->n_post;
    a.out = c->d_play_out;
    a.rs_taps = c->d_play_rs_taps;
    a.mono = c->recording ? c->d_play_mono : nullptr;
    c->play_run_frames = c->recording ? nf : 0;
    c->play_run_len = (uint32_t)per_frame;
    int rc;
    if ((rc = timed_begin(c)) != SSDR_OK) return rc;
    HIP_TRY(wide ? ssdr_launch_play_rs(a, c->stream) : ssdr_launch_play(a, c->stream));
    if (!wide) std::swap(c->d_play_hist, c->d_play_hist_alt);            // (the 64/27 branch carries no history)
    if ((rc = timed_end(c, SSDR_K_PLAY)) != SSDR_OK) return rc;
    if (out)
        HIP_TRY(hipMemcpyAsync(out, c->d_play_out, (size_t)c->n_post * nf * per_frame * 2 * sizeof(int16_t),
                               out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_recording(ssdr_ctx *c, int on) SSDR_GUARD
{
    if (!c) return SSDR_EINVAL;
    c->recording = on != 0;
    if (!c->recording) c->play_run_frames = 0;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_playbuffer_mono(ssdr_ctx *c, int16_t *mono_out, int out_is_device) SSDR_GUARD
{
    if (!c || !mono_out) return SSDR_EINVAL;
    if (!c->d_play_mono || c->play_run_frames == 0) return SSDR_ESTATE;       // the last ssdr_run_playbuffer did not record
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(mono_out, c->d_play_mono, (size_t)c->n_post * c->play_run_frames * c->play_run_len * sizeof(int16_t),
                           out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_push_iq_wire(ssdr_ctx *c, const uint8_t *bodies, uint32_t n_frames, float *rssi_out) SSDR_GUARD
{
    if (!c || !bodies || n_frames == 0) return SSDR_EINVAL;
    if (c->decim != 1) return SSDR_ESTATE;                       // SND bodies carry 512 IQ samples at 12 kHz
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_input(c, n_frames);
    if (rc != SSDR_OK) return rc;
    if (c->wire_frames < n_frames) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_wire) { HIP_TRY(hipFree(c->d_wire)); c->d_wire = nullptr; }
        if (c->d_wire_rssi) { HIP_TRY(hipFree(c->d_wire_rssi)); c->d_wire_rssi = nullptr; }
        if (c->d_wire_gps) { HIP_TRY(hipFree(c->d_wire_gps)); c->d_wire_gps = nullptr; }
        c->wire_frames = 0;
        HIP_TRY(hipMalloc(&c->d_wire, (size_t)c->n_ch * n_frames * SSDR_WIRE_BODY + 16));   // (the unpack kernel reads whole dwords)
        HIP_TRY(hipMalloc(&c->d_wire_rssi, (size_t)c->n_ch * n_frames * sizeof(float)));
        HIP_TRY(hipMalloc(&c->d_wire_gps, (size_t)c->n_ch * n_frames * 4 * sizeof(uint32_t)));
        c->wire_frames = n_frames;
    }
    HIP_TRY(hipMemcpyAsync(c->d_wire, bodies, (size_t)c->n_ch * n_frames * SSDR_WIRE_BODY, hipMemcpyHostToDevice, c->stream));
    SsdrWireArgs a;
    a.bodies = c->d_wire;
    a.n_ch = c->n_ch;
    a.n_frames = n_frames;
    a.iq = c->d_iq_own;
    a.ch_stride = (uint64_t)n_frames * SSDR_FRAME;
    a.rssi = c->d_wire_rssi;
    a.gps = c->d_wire_gps;
    c->wire_run_frames = n_frames;
    if ((rc = timed_begin(c)) != SSDR_OK) return rc;
    HIP_TRY(ssdr_launch_iqwire(a, c->stream));
    if ((rc = timed_end(c, SSDR_K_WIRE)) != SSDR_OK) return rc;
    if (rssi_out)
        HIP_TRY(hipMemcpyAsync(rssi_out, c->d_wire_rssi, (size_t)c->n_ch * n_frames * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->d_iq = c->d_iq_own;
    c->in_frames = n_frames;
    c->have_input = true;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_wire_gps(ssdr_ctx *c, uint32_t *gps_out) SSDR_GUARD
{
    if (!c || !gps_out) return SSDR_EINVAL;
    if (!c->d_wire_gps || c->wire_run_frames == 0) return SSDR_ESTATE;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(gps_out, c->d_wire_gps, (size_t)c->n_ch * c->wire_run_frames * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_adpcm_decode(ssdr_ctx *c, const uint8_t *data, uint32_t n_streams, uint32_t n_bytes, int32_t *state, int16_t *out) SSDR_GUARD
{
    if (!c || !data || !state || !out || n_streams == 0 || n_bytes == 0) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    uint8_t *d_in = nullptr;
    int32_t *d_st = nullptr;
    int16_t *d_out = nullptr;
    const size_t nin = (size_t)n_streams * n_bytes;
    int rc = [&]() -> int {
        HIP_TRY(hipMalloc(&d_in, nin));
        HIP_TRY(hipMalloc(&d_st, (size_t)n_streams * 8));
        HIP_TRY(hipMalloc(&d_out, nin * 4));
        HIP_TRY(hipMemcpyAsync(d_in, data, nin, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_st, state, (size_t)n_streams * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(ssdr_launch_adpcm(d_in, n_streams, n_bytes, d_st, d_out, c->stream));
        HIP_TRY(hipMemcpyAsync(out, d_out, nin * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(state, d_st, (size_t)n_streams * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return SSDR_OK;
    }();
    if (d_in) (void)hipFree(d_in);
    if (d_st) (void)hipFree(d_st);
    if (d_out) (void)hipFree(d_out);
    return rc;
} SSDR_UNGUARD

int ssdr_set_wf_lines(ssdr_ctx *c, const int16_t *wf_sum, uint32_t lines) SSDR_GUARD
{
    if (!c || !wf_sum || lines == 0) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    if (c->wf_out_lines < lines) {
        if (c->d_wf_out) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_wf_out)); c->d_wf_out = nullptr; c->wf_out_lines = 0; }
        HIP_TRY(hipMalloc(&c->d_wf_out, (size_t)lines * c->n_ch * SSDR_NFFT * 2));
        c->wf_out_lines = lines;
    }
    HIP_TRY(hipMemcpyAsync(c->d_wf_out, wf_sum, (size_t)lines * c->n_ch * SSDR_NFFT * 2, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->wf_lines_ready = lines;
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_set_pcm(ssdr_ctx *c, const int16_t *pcm, uint32_t n_frames) SSDR_GUARD
{
    if (!c || !pcm || n_frames == 0) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    { int rcd = drain_audio(c); if (rcd != SSDR_OK) return rcd; }
    if (c->audio_frames < n_frames) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_pcm) { HIP_TRY(hipFree(c->d_pcm)); c->d_pcm = nullptr; }
        if (c->d_rssi) { HIP_TRY(hipFree(c->d_rssi)); c->d_rssi = nullptr; }
        c->audio_frames = 0;
        HIP_TRY(hipMalloc(&c->d_pcm, (size_t)c->n_ch * n_frames * SSDR_FRAME * 2));
        HIP_TRY(hipMalloc(&c->d_rssi, (size_t)c->n_ch * n_frames * sizeof(float)));
        c->audio_frames = n_frames;
    }
    HIP_TRY(hipMemcpyAsync(c->d_pcm, pcm, (size_t)c->n_ch * n_frames * SSDR_FRAME * 2, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->audio_run_frames = n_frames;          // the input batch (d_iq / in_frames / have_input) is not touched
    return SSDR_OK;
} SSDR_UNGUARD

int ssdr_selftest_sqrt_values(ssdr_ctx *c, const float *in, float *out_scaled, float *out_int, uint32_t n) SSDR_GUARD
{
    if (!c || !in || !out_scaled || !out_int || n == 0) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    float *d = nullptr;
    HIP_TRY(hipMalloc(&d, (size_t)n * 3 * sizeof(float)));
    int rc = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(d, in, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(ssdr_launch_sqrt_values(d, d + n, d + 2 * (size_t)n, n, c->stream));
        HIP_TRY(hipMemcpyAsync(out_scaled, d + n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out_int, d + 2 * (size_t)n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return SSDR_OK;
    }();
    (void)hipFree(d);
    return rc;
} SSDR_UNGUARD

int ssdr_selftest_sqrt(ssdr_ctx *c, uint64_t *mismatches) SSDR_GUARD
{
    if (!c || !mismatches) return SSDR_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->d_scratch, 0, 8, c->stream));
    HIP_TRY(ssdr_launch_sqrt_selftest(c->d_scratch, c->stream));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, c->d_scratch, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *mismatches = v;
    return SSDR_OK;
} SSDR_UNGUARD

} // extern "C"
