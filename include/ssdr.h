/*
 * ssdr.h -- C ABI of libssdr.so: the MI355X (gfx950) implementation of the SuperSDR
 * DSP hot path (batched 1024-pt FFT waterfall + 12 kHz IQ audio chain).
 *
 * The reference (mcogoni/supersdr) is pure Python and has no FFI for this path; its
 * boundary is two worker classes and one callback hierarchy.
 * Each entry point below names the reference interface it sits behind.  The Python
 * host (supersdr_amd/) binds these with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 (SSDR_OK) or
 * a negative SSDR_E* code and never throws or aborts; the caller owns every host
 * buffer; a ctx owns its device memory, stream and per-channel state; one ctx per
 * GPU, single-owner (not thread-safe), used from the thread that created it.
 *
 * Data layouts (all little-endian, channel-major):
 *   IQ in   : int16 [n_ch][n_frames*512][2]  interleaved I,Q -- the sample type of
 *             KiwiSDRStream._process_aud's IQ branch (kiwi/client.py:443-454) before
 *             its float conversion (host byte-swaps the wire's big-endian once).
 *   WF out  : int16 [n_lines_out][n_ch][1024] -- SUM of `averaging` consecutive byte
 *             lines, ascending frequency; float32(sum)/float32(N) is bit-identical to
 *             the reference's np.mean time binning (utils_supersdr.py:881-888) and the
 *             bytes carry the reference's wire semantics dBm = byte - 255
 *             (utils_supersdr.py:780-791).
 *   PCM out : int16 [n_ch][n_frames*512] -- what kiwi_sound.process_audio_stream
 *             returns per SND frame (utils_supersdr.py:1044-1076).
 *   RSSI out: float [n_ch][n_frames] dBm -- replaces the SND header's smeter field
 *             (rssi = 0.1*smeter - 127, utils_supersdr.py:1068-1069).
 */
#ifndef SSDR_H
#define SSDR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDR_NFFT 1024          /* kiwi_waterfall.WF_BINS        utils_supersdr.py:596 */
#define SSDR_FRAME 512          /* KIWI_SAMPLES_PER_FRAME        utils_supersdr.py:909 */
#define SSDR_RATE 12000         /* KIWI_RATE                     utils_supersdr.py:906 */
#define SSDR_NTAP_MAX 128
#define SSDR_HIST 128

enum {
    SSDR_OK = 0,
    SSDR_EINVAL = -1,           /* bad argument                                         */
    SSDR_ENOMEM = -2,           /* host or device allocation failed                     */
    SSDR_EHIP = -3,             /* HIP runtime error (ssdr_last_hip_error has the text) */
    SSDR_ENODEV = -4,           /* no such GPU                                          */
    SSDR_ESTATE = -5            /* call out of order (e.g. run before push)             */
};

/* demodulator selection: "SET mod=%s" (utils_supersdr.py:1028; kiwi/client.py:217-249) */
enum { SSDR_MODE_AM = 0, SSDR_MODE_LSB = 1, SSDR_MODE_USB = 2, SSDR_MODE_CW = 3, SSDR_MODE_NBFM = 4,
       /* "SET mod=iq" (kiwi/client.py:217-249, default passband +-5 kHz): no demodulator -- the channel's tuned, filtered and
        * gain-controlled complex baseband itself.  The PCM row of such a channel carries I; I,Q pairs: ssdr_audio_iq */
       SSDR_MODE_IQ = 5 };

/* Per-channel parameters: the reference's a13 parameter surface:
 *   "SET mod=%s low_cut=%d high_cut=%d freq=%.3f"              utils_supersdr.py:1028
 *   "SET agc=%d hang=%d thresh=%d slope=%d decay=%d manGain=%d" utils_supersdr.py:1023 */
typedef struct ssdr_chan_params {
    int32_t mode;               /* SSDR_MODE_*                                          */
    int32_t agc_on;             /* kiwi_sound.on      (utils_supersdr.py:937)           */
    int32_t agc_hang;           /* kiwi_sound.hang    (:938)                            */
    int32_t reserved;
    double f_shift_hz;          /* tuning offset inside the 12 kHz IQ band: |f| <= 6000, else SSDR_EINVAL */
    double low_cut, high_cut;   /* passband Hz (kiwi_sound.lc/hc, :932; change_passband)*/
    double agc_thresh;          /* dBm  (:939, UI range -135..-20 supersdr.py:551-564)  */
    double agc_slope;           /* dB   (:940)                                          */
    double agc_decay;           /* ms   (:941-944)                                      */
    double agc_man_gain;        /* dB   (:942)                                          */
    double wf_cal_db;           /* additive waterfall calibration, within +-200 dB      */
    double smeter_cal_db;       /* dBFS->dBm offset (Kiwi default -13, :790)            */
} ssdr_chan_params;

/* Kernel-side constants derived from ssdr_chan_params (read back for tests). 64 B. */
typedef struct ssdr_chan_consts {
    uint32_t mode, ntap8, dphi1, dphi2;
    float wf_cal_lin, smeter_cal_db;
    float agc_c0, agc_c1, agc_knee, agc_delta8;
    uint32_t hang_frames, ntap;
    uint32_t tap_groups;            /* bit g set: taps 4g..4g+3 are not all zero (the FIR skips the others) */
    uint32_t fir_flags;             /* SSDR_FIR_*                                                            */
    uint32_t decim;                 /* D: the IQ arrives at D * 12 kHz (ssdr_set_decimation); taps are then stream-major, ntap8 per stream */
    float kfm;                      /* NBFM: output per radian = 16384 * rate / (2 pi 5000) -- 5 kHz deviation = half scale at the channel's rate */
} ssdr_chan_consts;
/* the channel filter is exactly a 4-sample delay (one unit tap at index 4: the full-band passband +-6 kHz at 12 kHz,
 * the reference's AM default, utils_supersdr.py:46).  The kernel then shifts samples across lanes instead of filtering,
 * and in AM -- where |x e^{j phi}| = |x| -- skips the NCO as well. */
#define SSDR_FIR_DELAY4 1u

/* Per-channel carried state (read back / restored for tests and checkpointing). 64 B. */
typedef struct ssdr_chan_state {
    uint32_t phi1, phi2;
    float dc, agc_d;
    float agc_m[8];
    float prev_re, prev_im;
    uint32_t pad[2];
} ssdr_chan_state;

typedef struct ssdr_ctx ssdr_ctx;

/* -- lifetime.  Stands where kiwi_waterfall.__init__/kiwi_sound.__init__ open their
 *    server connections (utils_supersdr.py:606-745, :911-1007): one ctx serves
 *    n_channels receivers.  nfft must be 1024 and frame 512. */
int ssdr_create(int device_id, uint32_t n_channels, uint32_t nfft, uint32_t frame, ssdr_ctx **out);
void ssdr_destroy(ssdr_ctx *ctx);

/* -- control plane: the SET commands listed above.  Changing parameters keeps the
 *    carried state (NCO phase, FIR history, AGC envelope), as a Kiwi server does.  Until the
 *    first ssdr_run_audio after ssdr_create / a full ssdr_reset_state there is no stream yet:
 *    ssdr_set_params then also puts the channel's AGC envelope at its new knee (the initial
 *    state of ssdr_reset_state: full gain, no pop). */
int ssdr_set_params(ssdr_ctx *ctx, uint32_t first, uint32_t count, const ssdr_chan_params *p);
int ssdr_default_params(int mode, ssdr_chan_params *out);      /* reference defaults, utils:42-50,936-944 */
int ssdr_reset_state(ssdr_ctx *ctx, uint32_t first, uint32_t count);
/* kiwi_waterfall.averaging_n (utils_supersdr.py:616, 881-886; supersdr.py:376-385), 1..100 */
int ssdr_set_averaging(ssdr_ctx *ctx, uint32_t n);

/* Input rate.  D = 1 (default): the IQ arrives at 12 kHz.  D = 2 or 4: it arrives at D * 12 kHz and the audio chain's
 * channel filter is a decimating FIR (the reference's tap formula, utils_supersdr.py:334-344, evaluated at the input
 * rate; output y[m] = sum_k h[k] z[D m - k]) down to the 12 kHz the demodulators, the AGC and the PCM frames run at.
 * A frame is still what yields 512 PCM samples: ssdr_push_iq then takes int16 [n_ch][n_frames*512*D][2], f_shift_hz may
 * reach +-D*6000, and the waterfall draws its lines (1024 or 512 input samples apart) from the wide stream.  Changing D
 * recompiles every channel's parameters and resets the streams.  Not available on the pipelined feed / wire input. */
int ssdr_set_decimation(ssdr_ctx *ctx, uint32_t decim);
int ssdr_compile_params_decim(const ssdr_chan_params *p, uint32_t decim, ssdr_chan_consts *consts, float *taps /*[128]*/);

/* Waterfall line rate.  hop = 1024 (default): one line per 1024 samples, 11.72 lines/s.  hop = 512: lines overlap by half,
 * 23.44 lines/s -- the rate the reference's waterfall runs at (kiwi_waterfall.MAX_FPS = 23, "SET wf_speed=4",
 * utils_supersdr.py:597, 742).  Line k then covers the 512 samples before frame k and frame k itself; ssdr_run_wf
 * accepts any frame count and delivers one line per frame; the half-line before the first frame is carried from the
 * previous batch (silence after ssdr_create / ssdr_set_hop).  A change restarts the averaging group. */
int ssdr_set_hop(ssdr_ctx *ctx, uint32_t hop);

/* Waterfall zoom: "SET zoom=%d start=%d" (utils_supersdr.py:741, 753-758, 839) -- a span narrower than the IQ band.  The
 * server zooms with a DDC in front of its FFT; here a zoom stage sits in front of the waterfall kernel: per channel the
 * input stream is mixed down by its zoom centre (offset_hz from the IQ band's centre, 32-bit phase accumulator), low-pass
 * filtered with the reference's tap formula (utils_supersdr.py:334-344: cut-off = the new Nyquist rate/(2 Z), 32 Z - 1 taps)
 * and decimated by Z to int16 I,Q (round-half-even, saturating) -- a stream at 1/Z of the input rate whose 1024-sample lines
 * span 1/Z of the band around offset_hz.  zoom Z in {1, 2, 4, 8} is ctx-wide (every channel's lines keep the same cadence:
 * one per 1024 Z input samples, or per 512 Z at hop 512), the centre is per channel.  A batch must then hold a whole
 * number of lines (SSDR_EINVAL otherwise).  Changing Z or a centre restarts that stream (phase, filter history, averaging
 * group).  The audio chain is not affected.  ssdr_read_zoom returns the zoomed I,Q of the last ssdr_run_wf (tests). */
int ssdr_set_wf_zoom(ssdr_ctx *ctx, uint32_t zoom);
int ssdr_set_wf_center(ssdr_ctx *ctx, uint32_t first, uint32_t count, const double *offset_hz);
int ssdr_read_zoom(ssdr_ctx *ctx, uint32_t first, uint32_t count, int16_t *iq_out /*[count][n_in/Z][2]*/, uint32_t *samples_per_channel);

/* Exact bins.  The waterfall kernel computes in float32; ~3e-4 of its bins, those whose |X| lies within the fp32 FFT's rounding
 * error of a 1-dB threshold, land one step away from where the float64 definition (oracle/ssdr_oracle.py: NumPy float64
 * FFT) puts them.  on = 1: ssdr_run_wf evaluates the stage in float64 instead (same window table, same thresholds) and
 * its int16 sums equal the float64 definition's bit for bit.  About 1.9x the default kernel's time (csrc/ssdr_wf_exact.hip); off by default.
 * ssdr_run_chain on the metric's configuration (every channel full-band AM, N = 1, hop 1024) then takes the float64 counterpart of
 * its fused kernel (ssdr_fused_exact_am_kernel: one read of the input, same bits as the two kernels); every other batch runs the
 * float64 waterfall kernel and the audio stage one after the other. */
int ssdr_set_exact_bins(ssdr_ctx *ctx, int on);

/* -- data plane.  ssdr_push_iq is the IQ ingest hook: KiwiSDRStream._process_iq_samples
 *    (kiwi/client.py:493-494).  iq may be a host pointer (copied) or a device pointer
 *    (referenced, must stay valid until the run_* calls on it have completed). */
int ssdr_push_iq(ssdr_ctx *ctx, const int16_t *iq, uint32_t n_frames, int is_device);
/* Fills what kiwi_waterfall.receive_spectrum leaves in self.spectrum (utils:780-785),
 * for every channel and every completed averaging group of the pushed batch
 * (n_frames must be even: one line per 1024 samples).  wf_sum_out may be NULL (results
 * stay in the ctx's device buffer, see ssdr_wf_device). */
int ssdr_run_wf(ssdr_ctx *ctx, int16_t *wf_sum_out, uint32_t *lines_ready, int out_is_device);
/* Fills what kiwi_sound.process_audio_stream returns (utils:1044-1076): int16 PCM and
 * rssi per 512-sample frame.  Either output may be NULL. */
int ssdr_run_audio(ssdr_ctx *ctx, int16_t *pcm_out, float *rssi_out, int out_is_device);
/* SND header flags, bit 1 "ADC overflow" (kiwi_sound.adc_overflow_flag, utils_supersdr.py:1066-1067) for every frame of the
 * last ssdr_run_audio: flags_out uint8 [n_ch][n_frames], 1 where a sample of that frame has |I| or |Q| >= 32767. */
int ssdr_audio_flags(ssdr_ctx *ctx, uint8_t *flags_out, int out_is_device);
/* The IQ-mode channels' output of the last ssdr_run_audio, as a KiwiSDR sends it in mod=iq SND frames (kiwi/client.py:443-454
 * before the byte order): iq_out int16 [n_ch][n_frames*512][2] interleaved I,Q = saturate(rint(y g)) of the filtered
 * baseband y and the AGC gain g.  Rows of channels in other modes are zero.  SSDR_ESTATE if no channel is in IQ mode. */
int ssdr_audio_iq(ssdr_ctx *ctx, int16_t *iq_out, int out_is_device);
/* Both stages on the current batch, results kept on the device (ssdr_wf_device / ssdr_audio_device / ssdr_audio_flags): what
 * ssdr_run_wf followed by ssdr_run_audio do, with results that are theirs bit for bit.  *fused (may be NULL) tells which way it went:
 *   1  every channel is on the reference's full-band AM passband, N = 1, hop 1024, 12 kHz IQ, no zoom, an even number of at least 8
 *      frames, and the ctx has at least ssdr_get_chain_floors' first number of channels (the configuration of the metric): ONE kernel does both, each 4 KB line is read once for its FFT and its two audio
 *      frames (ssdr_fused_am_kernel);
 *   2  every channel has a channel filter to apply (SSB, CW, IQ, a narrowed AM / NBFM passband: the general audio path), hop 1024,
 *      12 kHz IQ, no zoom, fp32 bins, any N, an even number of at least 8 frames, at least the second floor's channels: one kernel again, in which audio waves hand every
 *      raw frame to an FFT wave of their workgroup through the LDS (ssdr_chain_ws_kernel, round 6: one read of the input,
 *      1-2 % faster than the stages side by side);
 *   0  every other batch: the two stages side by side on two streams (ssdr_set_overlap).
 * ssdr_set_fused(ctx, 0) keeps the two kernels in every case.  The pipelined feed (ssdr_feed_*) runs its batches through this call. */
int ssdr_run_chain(ssdr_ctx *ctx, uint32_t *lines_ready, int *fused);
/* 0: never a one-read kernel; 1 (default): as listed at ssdr_run_chain; 2: ssdr_fused_am_kernel at hop 512, with N > 1 and below its channel
 * floor as well (there the two stages side by side are as fast or faster: ssdr_set_overlap); 3: ssdr_chain_ws_kernel for EVERY batch it can take --
 * any mix of audio paths (full-band channels among them), any filter, any N, an even number of frames at hop 1024: on BASELINE's
 * configs[3] 1 % slower than the stages side by side, with 39 % less HBM traffic (profiles/r06_ab_chain_ws.txt). */
int ssdr_set_fused(ssdr_ctx *ctx, int on);
/* The batch sizes from which ssdr_run_chain's default (level 1) takes a one-read kernel.  Below them the two stages side by side are the faster way
 * (a one-read kernel walks all lines of a channel pair in ONE wave; the waterfall kernel spreads them over the chip): at the reference's own scale,
 * tens of receivers, 2-3 x.  ssdr_create sets them from the device (MI355X: 8192 channels for ssdr_fused_am_kernel = one pair per resident wave,
 * 32768 for ssdr_chain_ws_kernel); 0 = no floor.  ssdr_set_fused(ctx, 2) / (ctx, 3) ignore the respective floor. */
int ssdr_set_chain_floors(ssdr_ctx *ctx, uint32_t fused_am_min_channels, uint32_t chain_ws_min_channels);
int ssdr_get_chain_floors(ssdr_ctx *ctx, uint32_t *fused_am_min_channels, uint32_t *chain_ws_min_channels);
/* Batches ssdr_run_chain does not fuse (mixed modes, N > 1, hop 512, float64 bins, ...) run their two stages SIDE BY SIDE: the audio
 * stage on a second HIP stream beside the waterfall kernel, both reading the same input batch (default on; results are those of
 * one after the other, bit for bit).  Every later call that needs the audio stage's results, its state or the input buffer
 * waits for it first.  0: one after the other (per-stage timings that do not overlap). */
int ssdr_set_overlap(ssdr_ctx *ctx, int on);
int ssdr_sync(ssdr_ctx *ctx);

/* -- the reference's own post-processing of the two streams, on the GPU.
 *    Per-channel display state of kiwi_waterfall.spectrum_db2col (utils_supersdr.py:787-813). 48 B. */
typedef struct ssdr_db2col_chan {
    int32_t zoom;                   /* kiwi_waterfall.zoom                                           */
    int32_t auto_scale;             /* wf_auto_scaling                                               */
    int32_t delta_low_db, delta_high_db;
    float low_clip_db, high_clip_db, dynamic_range;     /* in (when !auto_scale) / out               */
    float wf_min_db, wf_max_db;                         /* out (:807-808)                            */
    uint32_t pad[3];
} ssdr_db2col_chan;
/* The reference's post-processing is per viewer (one kiwi_waterfall / kiwi_sound per receiver somebody looks at or listens to); a ctx
 * of 10^5 channels has a handful.  ssdr_set_post_channels names the channels the post-processing entry points work on from
 * now on -- `channels` ascending and unique, `count` of them; (NULL, 0): every channel again, the default.  With a selection
 * set, every per-channel array of ssdr_run_db2col, ssdr_run_playbuffer, ssdr_playbuffer_mono, ssdr_feed_post,
 * ssdr_feed_collect_post, ssdr_run_trace, ssdr_push_color_lines and ssdr_wfdata_white_flag has `count` entries in the
 * order of the list (display state in, colours / 48 kHz blocks / traces out); cost and transfers scale with the listeners,
 * not with the ctx.  play_buffer's carried history stays per channel (a channel that leaves and re-enters the selection
 * continues where it was).  The device copy of wf_data (ssdr_set_wfdata_rows) starts over.  count = 0 with a non-NULL list:
 * nobody is looking -- the post-processing calls return at once.  Batches already submitted to the pipelined feed keep the
 * selection they were submitted with (the caller remembers it for ssdr_feed_collect_post). */
int ssdr_set_post_channels(ssdr_ctx *ctx, const uint32_t *channels, uint32_t count);
/* spectrum_db2col for every channel and every line produced by the last ssdr_run_wf:
 * color_out float32 [lines][n_ch][1024] in 0..254 (wf_color), chans[] updated in place (host memory). */
int ssdr_run_db2col(ssdr_ctx *ctx, ssdr_db2col_chan *chans, float *color_out, int out_is_device);
/* spectrum_db2col of ONE line of one receiver that did its own time binning (a client whose N differs from the hub's):
 * wf_sum int16 [1024] = sum of n_avg byte lines, chan in/out, color_out float32 [1024] (all host memory).  Touches nothing
 * the other channels see: not the lines of the last ssdr_run_wf, not the device copy of wf_data. */
int ssdr_db2col_line(ssdr_ctx *ctx, const int16_t *wf_sum, uint32_t n_avg, ssdr_db2col_chan *chan, float *color_out);

/* kiwi_sound.play_buffer (utils_supersdr.py:1106-1148) for every channel and every frame of the last
 * ssdr_run_audio: volume, x4 interpolation with filtering(KIWI_RATE/2, AUDIO_RATE) (:999), pan^2,
 * truncating int16 stereo pack.  out int16 [n_ch][n_frames*2048][2].  The (n_tap-1)-sample history
 * (old_buffer, :1005,1133) is carried per channel in the ctx. */
typedef struct ssdr_play_chan {
    double volume;                  /* kiwi_sound.volume, percent (:921, supersdr.py:397-406)        */
    double balance;                 /* kiwi_sound.audio_balance in [-1, 1] (:945)                    */
} ssdr_play_chan;
int ssdr_run_playbuffer(ssdr_ctx *ctx, const ssdr_play_chan *chans, int16_t *out, int out_is_device);

/* kiwi_sound.KIWI_RATE as announced by the server ("audio_init ... audio_rate=", utils_supersdr.py:988-994):
 * SSDR_RATE (12000, default) or SSDR_RATE_WIDE (20250, three-channel KiwiSDRs).  With 20250 SAMPLE_RATIO =
 * 48000/20250 is fractional and play_buffer takes its resample_poly(popped, 64, 27, padtype="line")[:-1] branch
 * (:1000-1001, 1125-1126): ssdr_run_playbuffer then writes int16 [n_ch][n_frames*1213][2], each frame resampled
 * on its own.  ssdr_playbuffer_frame_len returns the stereo samples per frame of the selected path
 * (2048 or 1213 = int(512 * SAMPLE_RATIO), the OutputStream blocksize of :1211). */
#define SSDR_RATE_WIDE 20250
int ssdr_set_kiwi_rate(ssdr_ctx *ctx, uint32_t kiwi_rate);
/* The same rate is the rate of the IQ the channels receive (a three-channel KiwiSDR delivers 20.25 kHz IQ and SND frames,
 * utils_supersdr.py:988-994): NCO steps, channel-filter design, AGC time constants and the NBFM scale are compiled for it
 * (ssdr_compile_params_rate), a frame stays 512 samples, the waterfall's 1024 bins then span 20.25 kHz and f_shift_hz may
 * reach +-10125.  A change recompiles every channel's parameters and resets the streams, like ssdr_set_decimation. */
int ssdr_compile_params_rate(const ssdr_chan_params *p, uint32_t decim, uint32_t rate, ssdr_chan_consts *consts, float *taps /*[128]*/);
/* audio_rec.recording_flag (utils_supersdr.py:149-157, 1139-1140): while set, ssdr_run_playbuffer also keeps what play_buffer
 * appends to audio_rec.audio_buffer -- the interpolated block before the pan, pyaudio_buffer.astype(np.int16) -- and
 * ssdr_playbuffer_mono returns it for the last run: int16 [n_ch][n_frames*L], L = ssdr_playbuffer_frame_len(). */
int ssdr_set_recording(ssdr_ctx *ctx, int on);
int ssdr_playbuffer_mono(ssdr_ctx *ctx, int16_t *mono_out, int out_is_device);
int ssdr_playbuffer_frame_len(ssdr_ctx *ctx, uint32_t *samples_per_frame);

/* -- display reductions on device-resident state
 *
 * ssdr_set_wfdata_rows(rows > 0): every colour line ssdr_run_db2col produces is also kept on the device as the newest
 * rows of kiwi_waterfall.wf_data (utils_supersdr.py:692-693, 893-897: float64 [WF_HEIGHT][1024] fed through a 3-deep
 * deque, newest row on top; only the `rows` newest rows are kept).  rows = 0 (default) turns it off and frees it.
 * ssdr_push_color_lines feeds colour lines float32 [lines][n_ch][1024] that were not produced by ssdr_run_db2col
 * through the same queue; ssdr_wfdata_white_flag is kiwi_waterfall.set_white_flag (:875-877) for channels
 * [first, first + count): row 0 becomes 255.
 *
 * ssdr_run_trace: display_stuff.plot_spectrum's reduction (utils_supersdr.py:1678-1679) for every channel:
 *   trace_out double [n_ch][1024] = np.nanmean(wf_data.T[:, :t_avg], axis=1)      (t_avg <= rows; reference: 15)
 *   y_out     int32  [n_ch][1024] = SPECTRUM_HEIGHT-1-int(v/255 * SPECTRUM_HEIGHT)  (may be NULL)
 *
 * ssdr_run_smeter: one display frame of the main loop's S-meter smoothing (supersdr.py:164-168, 190-191, 936-947)
 * for every channel; chans[] (host memory) is updated in place.  rssi_in double [n_ch] (host) is the frame's
 * kiwi_snd.rssi reading; NULL takes the last frame's RSSI of the last ssdr_run_audio. */
typedef struct ssdr_smeter_chan {
    double rssi_smooth, rssi_smooth_slow;   /* supersdr.py:166-167                                            */
    double hist[10];                        /* rssi_hist = deque(maxlen=rssi_maxlen=10) (:164-165), ring      */
    uint32_t hist_pos, run_index;           /* next ring slot; run_index of the main loop (:169, % 20 at :945) */
    double decay_ms;                        /* kiwi_snd.decay (:941)                                          */
} ssdr_smeter_chan;                         /* 112 B                                                          */
int ssdr_set_wfdata_rows(ssdr_ctx *ctx, uint32_t rows);
int ssdr_push_color_lines(ssdr_ctx *ctx, const float *color, uint32_t lines, int color_is_device);
int ssdr_wfdata_white_flag(ssdr_ctx *ctx, uint32_t first, uint32_t count);
int ssdr_run_trace(ssdr_ctx *ctx, uint32_t t_avg, uint32_t spectrum_height, double *trace_out, int32_t *y_out, int out_is_device);
int ssdr_run_smeter(ssdr_ctx *ctx, ssdr_smeter_chan *chans, const double *rssi_in, double fps);

/* KiwiSDRStream._process_aud, IQ branch (kiwi/client.py:384-389, 443-454): n_frames SND bodies per channel
 * (each 7 B flags/seq/smeter + 10 B GPS + 512 big-endian I,Q pairs = 2065 B, layout [n_ch][n_frames][2065],
 * host memory) become the current input batch, as ssdr_push_iq would; rssi_out (may be NULL) receives
 * 0.1*smeter - 127 per frame. */
int ssdr_push_iq_wire(ssdr_ctx *ctx, const uint8_t *bodies, uint32_t n_frames, float *rssi_out);
/* The GNSS stamps of those frames -- the `gps` dict _process_aud hands to _process_iq_samples (kiwi/client.py:444-445, 454):
 * gps_out uint32 [n_ch][n_frames][4] = last_gps_solution, dummy, gpssec, gpsnsec of each frame of the last
 * ssdr_push_iq_wire (host memory). */
int ssdr_wire_gps(ssdr_ctx *ctx, uint32_t *gps_out);

/* IMA ADPCM decoder of compressed SND / W-F payloads (kiwi/client.py:33-87, 461-464, 476-479): n_streams
 * independent streams of n_bytes each (host memory, [n_streams][n_bytes]); state int32 [n_streams][2] =
 * {index, prev} in/out (zero it per W/F line, keep it across SND frames); out int16 [n_streams][2*n_bytes]. */
int ssdr_adpcm_decode(ssdr_ctx *ctx, const uint8_t *data, uint32_t n_streams, uint32_t n_bytes, int32_t *state, int16_t *out);

/* -- pipelined host feed: the path a live ingest takes (KiwiSDRStream._process_iq_samples -> batches, kiwi/client.py:493)
 *
 * ssdr_push_iq + ssdr_run_* from pageable host memory serialise copy-in, kernels and copy-out.  The feed keeps `depth`
 * slots of pinned host memory with their own device buffers and runs three HIP streams: while the kernels work on
 * batch k, batch k+1 is copied in and the results of batch k-1 are copied out.  Results are those of ssdr_push_iq /
 * ssdr_run_wf / ssdr_run_audio on the same batches in the same order (state and partial waterfall sums carry over).
 *
 *   ssdr_feed_open(ctx, n_frames, depth, flags)
 *                                          n_frames (even) 512-sample frames per channel and batch, 2 <= depth <= 16;
 *                                          flags = SSDR_FEED_WIRE: the slots take the SND bodies as they come off the
 *                                          socket (uint8 [n_ch][n_frames][2065], layout of ssdr_push_iq_wire) and the
 *                                          header strip / byte swap runs on the device
 *   ssdr_feed_slot(ctx, &in)               pinned int16 [n_ch][n_frames*512][2] (or bodies) to fill; SSDR_ESTATE if all slots are in flight
 *   ssdr_feed_submit(ctx)                  queue the slot: copy-in, both kernels, copy-out; returns at once
 *   ssdr_feed_collect(ctx, &wf, &lines, &pcm, &rssi, &wire_rssi, &flags, &n_avg)
 *                                          wait for the OLDEST submitted batch; pinned int16 [lines][n_ch][1024], int16
 *                                          [n_ch][n_frames*512], float [n_ch][n_frames]; wire_rssi float [n_ch][n_frames] =
 *                                          0.1*smeter - 127 of the SND headers (NULL without SSDR_FEED_WIRE); flags uint8
 *                                          [n_ch][n_frames] ADC overflow per frame (ssdr_audio_flags); n_avg = the averaging N
 *                                          that was in force when the batch was submitted; valid until that slot is handed
 *                                          out again.  Any pointer may be NULL.
 *   ssdr_feed_close(ctx)
 * flags = SSDR_FEED_POST: every batch also goes through the reference's post-processing on the device, in the same slot
 * pipeline -- spectrum_db2col of its waterfall lines and play_buffer of its PCM frames (what ssdr_run_db2col /
 * ssdr_run_playbuffer do for an un-pipelined batch, state carried from batch to batch the same way):
 *   ssdr_feed_post(ctx, chans, play)       display state for the batches submitted from now on: ssdr_db2col_chan [n_ch]
 *                                          and ssdr_play_chan [n_ch] (host, copied; NULL keeps what was set before)
 *   ssdr_feed_collect_post(ctx, &color, &chans, &play, &mono)
 *                                          of the batch ssdr_feed_collect returned last: float32 [lines][n_ch][1024]
 *                                          wf_color, ssdr_db2col_chan [n_ch] as spectrum_db2col left them, int16
 *                                          [n_ch][n_frames*L][2] (L = ssdr_playbuffer_frame_len), int16 [n_ch][n_frames*L]
 *                                          mono block (NULL unless ssdr_set_recording)
 * Without SSDR_FEED_POST the post-processing entry points keep referring to the last ssdr_run_* batch, not to fed ones. */
#define SSDR_FEED_WIRE 1u
#define SSDR_FEED_POST 2u
/* flags = SSDR_FEED_LAZY_OUT (round 5): a hub of 10^5 receivers has a handful of listeners (README.md:8 "dozens of instances"; one
 * kiwi_waterfall / kiwi_sound pair each, utils_supersdr.py:780-785, 1044-1076), and copying every channel's line, PCM, RSSI and flags
 * back -- 4 KB per channel-superframe of PCIe and host DRAM writes nobody reads -- is what the feed spent its return path on.  With this
 * flag only the channels of ssdr_set_post_channels (at most SSDR_FEED_LAZY_MAX; no selection: all channels, if they are that few) come
 * back: ssdr_feed_collect's arrays are then COMPACT -- wf [lines][n_sel][1024], pcm [n_sel][n_frames*512], rssi / wire_rssi / flags
 * [n_sel][n_frames], rows in the order of the selection in force at the batch's submit (ssdr_feed_collect_lazy tells n_sel) -- and the
 * whole-batch results stay on the device, in the slot's buffers, for device-side consumers (ssdr_feed_collect_lazy: valid until
 * depth - 1 further batches have been submitted). */
#define SSDR_FEED_LAZY_OUT 4u
#define SSDR_FEED_LAZY_MAX 4096u
int ssdr_feed_collect_lazy(ssdr_ctx *ctx, uint32_t *n_sel, int16_t **d_wf_sum, int16_t **d_pcm, float **d_rssi, uint8_t **d_flags);
int ssdr_feed_open(ssdr_ctx *ctx, uint32_t n_frames, uint32_t depth, uint32_t flags);
int ssdr_feed_slot(ssdr_ctx *ctx, void **host_in);
int ssdr_feed_submit(ssdr_ctx *ctx);
/* The same, but the batch is taken from the caller's own host buffer instead of the slot ssdr_feed_slot hands out (layout and
 * size of that slot).  For an ingest that assembles its batches in place (supersdr_amd/workers.py:IQHub, whose ring of
 * superframe slots is the thing KiwiSDRStream._process_iq_samples fills, kiwi/client.py:493-494): no copy into the slot.
 * The buffer must stay untouched until ssdr_feed_collect has returned that batch; pinned memory (ssdr_host_alloc) keeps
 * the copy asynchronous.  SSDR_ESTATE while a slot from ssdr_feed_slot is outstanding or every slot is in flight. */
int ssdr_feed_submit_from(ssdr_ctx *ctx, const void *host_in);
/* Pinned host memory for such buffers (hipHostMalloc); freed by ssdr_host_free, not by ssdr_destroy. */
int ssdr_host_alloc(ssdr_ctx *ctx, uint64_t bytes, void **out);
int ssdr_host_free(ssdr_ctx *ctx, void *ptr);
int ssdr_feed_collect(ssdr_ctx *ctx, int16_t **wf_sum, uint32_t *lines, int16_t **pcm, float **rssi, float **wire_rssi,
                      uint8_t **flags, uint32_t *n_avg);
int ssdr_feed_post(ssdr_ctx *ctx, const ssdr_db2col_chan *chans, const ssdr_play_chan *play);
int ssdr_feed_collect_post(ssdr_ctx *ctx, float **color, ssdr_db2col_chan **chans, int16_t **play, int16_t **mono);
int ssdr_feed_close(ssdr_ctx *ctx);

/* -- device-resident results of the last run_* (for zero-copy consumers and bench) */
int ssdr_wf_device(ssdr_ctx *ctx, int16_t **ptr, uint32_t *lines);
int ssdr_copy_from_device(ssdr_ctx *ctx, void *host_dst, const void *device_src, uint64_t bytes);   /* ordered behind the ctx's work */
int ssdr_audio_device(ssdr_ctx *ctx, int16_t **pcm, float **rssi);

/* Position-weighted 64-bit checksums of the device-resident results of the last ssdr_run_wf / ssdr_run_audio (or
 * ssdr_run_chain): sums[0] waterfall sums, sums[1] PCM, sums[2] RSSI bit patterns.  Integer arithmetic only, so equal
 * results give equal checksums on any GPU and in any launch shape -- the per-rank parity hash of the multi-GPU bench
 * (SURVEY.md 8e): a rank's channel block must hash to what a one-rank run of the same block hashes to. */
int ssdr_output_checksum(ssdr_ctx *ctx, uint64_t sums[3]);

/* -- measurement */
int ssdr_set_stream(ssdr_ctx *ctx, void *hip_stream);           /* NULL = ctx's own stream.  On a caller's stream ssdr_run_chain joins its
                                                                  * side-by-side audio stage before it returns: work ordered behind that stream sees both stages */
int ssdr_set_profiling(ssdr_ctx *ctx, int on);                  /* HIP-event pair around every launch */
/* bit 0: run the audio stage on a second stream beside the waterfall kernel (which then takes one workgroup per CU);
 * bit 1: run the audio stage's per-path kernels one after the other instead of side by side (measurement only) */
int ssdr_set_concurrent(ssdr_ctx *ctx, int on);
enum { SSDR_K_WF = 0, SSDR_K_AUDIO = 1, SSDR_K_SYNTH = 2, SSDR_K_DB2COL = 3, SSDR_K_PLAY = 4, SSDR_K_WIRE = 5, SSDR_K_TRACE = 6, SSDR_K_SMETER = 7, SSDR_K_FUSED = 8, SSDR_K_ZOOM = 9, SSDR_K_COUNT = 10 };
int ssdr_kernel_stats(ssdr_ctx *ctx, int which, float *total_ms, uint32_t *launches, int reset);
/* channels per frame path of the audio stage (one kernel each, timed together as SSDR_K_AUDIO): counts[0] general
 * (NCO -> FIR), counts[1] full-band lane shift, counts[2] full-band AM (no NCO, no FIR) */
int ssdr_audio_paths(ssdr_ctx *ctx, uint32_t counts[3]);
int ssdr_elapsed_ms(ssdr_ctx *ctx, float *ms);                  /* last run_* call, device time */

/* -- synthetic input generated on the device (bench): makes a batch of
 *    n_frames frames for all channels the current input, as ssdr_push_iq would. */
int ssdr_synth_iq(ssdr_ctx *ctx, uint32_t n_frames, uint32_t seed, uint32_t first_channel_id);
int ssdr_read_input(ssdr_ctx *ctx, uint32_t first, uint32_t count, int16_t *iq_out);

/* -- introspection for tests */
enum { SSDR_T_WINDOW = 0, SSDR_T_TWIDDLE_RE = 1, SSDR_T_TWIDDLE_IM = 2, SSDR_T_DB_THRESH = 3 };
int ssdr_table(int which, float *out, uint32_t n);             /* 1024 / 512 / 512 / 256 floats */
int ssdr_compile_params(const ssdr_chan_params *p, ssdr_chan_consts *consts, float *taps /*[128]*/);
int ssdr_get_consts(ssdr_ctx *ctx, uint32_t first, uint32_t count, ssdr_chan_consts *consts, float *taps);
int ssdr_get_state(ssdr_ctx *ctx, uint32_t first, uint32_t count, ssdr_chan_state *state, int16_t *hist);
int ssdr_set_state(ssdr_ctx *ctx, uint32_t first, uint32_t count, const ssdr_chan_state *state, const int16_t *hist);
/* Checkpoint: everything a ctx carries from one call to the next -- compiled channel constants and taps, NCO phases, FIR
 * history, DC / AGC / discriminator state, the waterfall's partial sums with their phase and N, the play_buffer
 * history -- as one blob of ssdr_checkpoint_size bytes (host memory).  Loading it into a ctx of the same channel count
 * (fresh or not) continues the streams bit for bit; the restored stream counts as live, so a later ssdr_set_params
 * keeps its state. */
/* Not in the blob: the zoomed waterfall stream (ssdr_set_wf_zoom > 1: save and load return SSDR_ESTATE while a zoom is set) and
 * mode switches that carry no stream state (ssdr_set_exact_bins, ssdr_set_fused).  The kernels' per-channel constants are
 * recompiled from the saved ssdr_chan_params on load (blob version 4; older blobs are refused with SSDR_EINVAL). */
int ssdr_checkpoint_size(ssdr_ctx *ctx, uint64_t *bytes);
int ssdr_checkpoint_save(ssdr_ctx *ctx, void *blob);
/* bytes must equal ssdr_checkpoint_size; the header and every channel's compiled constants are validated before anything
 * is touched (SSDR_EINVAL for a short, foreign or damaged blob).  The blob carries its own hop, decimation and N:
 * read them back with ssdr_get_config. */
int ssdr_checkpoint_load(ssdr_ctx *ctx, const void *blob, uint64_t bytes);
/* what the ctx currently runs with (any pointer may be NULL): waterfall hop (ssdr_set_hop), input decimation
 * (ssdr_set_decimation), averaging N (ssdr_set_averaging), play-back rate (ssdr_set_kiwi_rate) */
int ssdr_get_config(ssdr_ctx *ctx, uint32_t *hop, uint32_t *decim, uint32_t *averaging, uint32_t *kiwi_rate);
/* inject results as if ssdr_run_wf / ssdr_run_audio had produced them (golden-vector tests of the post-processing) */
int ssdr_set_wf_lines(ssdr_ctx *ctx, const int16_t *wf_sum /*[lines][n_ch][1024]*/, uint32_t lines);
int ssdr_set_pcm(ssdr_ctx *ctx, const int16_t *pcm /*[n_ch][n_frames*512]*/, uint32_t n_frames);
int ssdr_selftest_quantiser(ssdr_ctx *ctx, uint64_t *mismatches);   /* all positive floats vs binary search */
int ssdr_selftest_sqrt(ssdr_ctx *ctx, uint64_t *mismatches);        /* AM envelope sqrt vs the device's IEEE sqrtf, exhaustively */
/* the same two square roots (scaled form; integer-power form, valid for 0 and [1, 2^33)) on n caller-chosen arguments, for
 * a check against an IEEE sqrt that is not the device's own */
int ssdr_selftest_sqrt_values(ssdr_ctx *ctx, const float *in, float *out_scaled, float *out_int, uint32_t n);

const char *ssdr_strerror(int code);
const char *ssdr_last_hip_error(void);
const char *ssdr_version(void);

#ifdef __cplusplus
}
#endif
#endif
