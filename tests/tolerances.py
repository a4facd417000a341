"""The north_star audio tolerance (PCM within 1e-5 RMS of full scale vs the float64 oracle) and the condition it is
asserted under -- one statement, used by the CPU sweep (twin vs oracle) and the GPU sweeps (kernels vs oracle).

An fp32 chain cannot be within 1e-5 of full scale of the float64 chain on EVERY input: the channel filter takes an
out-of-band carrier 60 dB down and what is left is only known to 2^-24 of the INPUT; the FM discriminator takes the
phase of samples whose magnitude passes through zero; 99 dB of manual gain behind an AM detector amplifies the
cancellation of envelope and DC estimate.  Instead of excusing such channels by hand (rounds 1-3 did: a 40 dB rule, an
NBFM exception, a trimmed RMS -- and a 120-seed soak then found 3 % of sweeps outside it), the rule is the numerical
analyst's: the kernels must be a BACKWARD-STABLE evaluation of the float64 chain.

  * The oracle propagates, per output sample, what a relative error EPS in the channel filter's sums does to ITS OWN
    output, to first order (oracle/ssdr_oracle.py: AudioChannel.error_sensitivity -- the dot-product bound
    sum |h||z| through envelope / product detector / discriminator incl. its branch cut / AGC gain law / int16 clip):
    bound[ch, n], in LSB.
  * EPS = 2^-20 = 16 units of fp32 roundoff (the classical bound for a 127-tap sum alone is 127 u).
  * EVERY sample of EVERY channel:   |pcm - pcm_oracle| <= 1 LSB + bound             (1 LSB: the two roundings to int16)
  * EVERY channel:                   RMS(pcm - pcm_oracle) <= 1e-5 + RMS(bound)             of full scale (Minkowski)
  * a channel is WELL CONDITIONED when RMS(bound) <= 1e-5 of full scale (1/3 LSB: the float64 chain itself moves by
    less than the tolerance under that perturbation); every well-conditioned channel must meet the plain north_star
    figure, 1e-5 RMS UNTRIMMED -- no per-mode exception.  So that this says something, a sweep must hold its share of
    well-conditioned channels: the draw of tests/random_params.py gives 84 %; a sweep of n channels must have at least
    0.8 n - 4 sqrt(0.16 n) of them (four standard deviations under 80 %: 61 of 96, 11 of 24).

Calibration (tools/soak_parity.py, profiles/r04_soak_parity.txt): 600 sweeps x 96 channels x 3072 samples of
tests/random_params.py: the largest EPS any sample needed was 2^-21.3; 83 % (AM/SSB/CW) and 90 % (NBFM) of the channels
are well conditioned, their largest untrimmed RMS deviation is 5.2e-6 (GPU soak, 54 048 channels: 5.6e-6).
The rule's bound is the oracle's statement about itself; two things keep it honest (round 5): an ABSOLUTE ceiling on every channel
(ABS_RMS_CEILING, over the samples off the discriminator's branch cut) that does not use the bound at all, and a test that the bound
is tight -- the float64 chain run with its filter sums actually perturbed by EPS times their dot-product bound moves, at its worst sample, by most of
what error_sensitivity predicts (0.9-1.0 in the test's channels), never by more (tests/test_oracle_known_answers.py::test_error_sensitivity_is_a_tight_bound).
The figures of each sweep are kept in REPORT; how many channels fall outside the plain 1e-5 is GATED since round 6 (PLAIN_MISS_SHARE: at most 1 % of all the
channels compared in a session) and written to gpurun_out/tolerance_report_{cpu,gpu}.json by tests/conftest.py.
"""
import numpy as np

from oracle import ssdr_oracle as O

PCM_RMS_TOL = 1e-5
ABS_RMS_CEILING = 3e-3          # EVERY channel, however badly conditioned, over all its samples but those at the discriminator's branch cut:
                                # an absolute figure that owes nothing to the oracle's own bound (advisor, round 4).  Calibration: 400 CPU sweeps /
                                # 9600 channels of tests/random_params.py: largest value 2.6e-4; GPU soak, 394 sweeps / 27 024 channels incl. the
                                # decimating front ends: largest 1.07e-3 (one D = 4 channel behind its 125 taps; profiles/r05_soak_parity.txt).
                                # 0.4-0.5 % of the channels miss the plain 1e-5
PLAIN_MISS_SHARE = 0.01         # GATE (round 6; it was a printout): over all the channels compared so far in this process (REPORT), at most 1 % may
                                # miss north_star's PLAIN 1e-5 RMS -- plus three standard deviations of that share while the count is small
                                # (96 channels: 3; 384: 9; 10 000: 129 = 1.3 %; the soaks measure 0.4-0.5 %).  assert_share_outside_plain()
EPS = 2.0 ** -20
WELL_FRACTION = 0.8            # of a random sweep, expected (measured: 0.84); min_well_for(n) is four sigma under it
REPORT = []                     # one dict per sweep


def min_well_for(n_ch):
    return max(int(np.floor(WELL_FRACTION * n_ch - 4.0 * np.sqrt(WELL_FRACTION * (1.0 - WELL_FRACTION) * n_ch))), 1)


def rssi_well_conditioned(iq, rssi_o, smeter_cal_db):
    """The S-meter's own condition (it reads the filtered POWER, whatever the AGC then does with the audio): the channel
    filter leaves, in every frame, a power within 40 dB of the input's -> the fp32 RSSI is good to 1e-3 dB.
    iq int16 [n_ch, n, 2]; rssi_o [n_ch, n_frames] of the oracle; smeter_cal_db [n_ch] -> bool [n_ch]"""
    in_db = 10 * np.log10(np.maximum((iq.astype(np.float64) ** 2).sum(axis=2).mean(axis=1), 1e-20) / 32768.0 ** 2)
    in_db = in_db + np.asarray(smeter_cal_db, np.float64)
    return (rssi_o > in_db[:, None] - 40).all(axis=1)


def oracle_with_bound(iq, params, decim=1, rate=O.RATE):
    """(pcm_o, rssi_o, bound): the float64 chain and, per sample, how far a backward-stable fp32 evaluation may lie from it"""
    return O.audio_chain_with_bound(iq, params, EPS, decim, rate)


def share_outside_plain(report=None):
    """-> (channels, how many of them miss the plain 1e-5 RMS, the number the gate allows) over the sweeps recorded so far"""
    rep = REPORT if report is None else report
    n = sum(r["channels"] for r in rep)
    out = sum(r["outside_plain_tolerance"] for r in rep)
    allowed = int(np.floor(PLAIN_MISS_SHARE * n + 3.0 * np.sqrt(PLAIN_MISS_SHARE * (1.0 - PLAIN_MISS_SHARE) * n)))
    return n, out, allowed


def assert_share_outside_plain(report=None, strict_above=5000):
    """the gate: the share of channels that miss north_star's plain figure stays at or under 1 % (from `strict_above` channels on without
    the small-sample allowance)"""
    n, out, allowed = share_outside_plain(report)
    if n >= strict_above:
        allowed = int(np.floor(PLAIN_MISS_SHARE * n))
    assert out <= allowed, ("too many channels outside the plain 1e-5 RMS", out, "of", n, "allowed", allowed)
    return n, out, allowed


def write_report(path, report=None):
    """REPORT + its totals as JSON (tests/conftest.py writes gpurun_out/tolerance_report_*.json at the end of a session)"""
    import json
    import os
    rep = REPORT if report is None else report
    n, out, allowed = share_outside_plain(rep)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        json.dump({"north_star_plain_tolerance_rms": PCM_RMS_TOL, "channels": n, "outside_plain_tolerance": out,
                   "share_outside_plain_tolerance": (out / n) if n else None, "gate": "<= %g of the channels (+ 3 sigma while n < 5000)" % PLAIN_MISS_SHARE,
                   "allowed_at_this_count": allowed, "well_conditioned": sum(r["well_conditioned"] for r in rep),
                   "largest_rms_of_any_channel_off_the_branch_cut": max((r["rms_all_channels_off_the_branch_cut_max"] for r in rep), default=None),
                   "largest_untrimmed_rms_of_a_well_conditioned_channel": max((r["untrimmed_max_well"] for r in rep), default=None),
                   "absolute_ceiling": ABS_RMS_CEILING, "eps": EPS, "sweeps": rep}, f, indent=1)


def assert_pcm_within_tolerance(pcm, pcm_o, bound, min_well=None, what=""):
    """pcm, pcm_o [n_ch, n]; bound [n_ch, n] of oracle_with_bound.  min_well: channels that must be well conditioned
    (default: min_well_for(n_ch)).  Returns (well [n_ch] bool, rms [n_ch])."""
    err = np.abs(np.asarray(pcm, np.float64) - np.asarray(pcm_o, np.float64))
    n_ch = err.shape[0]
    rms = np.sqrt((err ** 2).mean(axis=1)) / 32768.0
    brms = np.sqrt((bound ** 2).mean(axis=1)) / 32768.0
    well = brms <= PCM_RMS_TOL
    need = min_well_for(n_ch) if min_well is None else int(min_well)
    with np.errstate(divide="ignore", invalid="ignore"):       # the share of its bound a sample uses beyond the rounding LSB
        worst = np.where(err > 1.0, (err - 1.0) / bound, 0.0)
    turn = bound >= 30000.0                                     # the discriminator at its branch cut: +pi or -pi, a full turn apart
    # an absolute figure that owes nothing to the oracle's own bound (advisor, round 4): the RMS of every channel, well conditioned or not,
    # over all its samples but those at the discriminator's branch cut
    rms_abs = np.sqrt((np.where(turn, 0.0, err) ** 2).mean(axis=1)) / 32768.0
    REPORT.append({"what": what, "channels": n_ch, "well_conditioned": int(well.sum()), "outside_plain_tolerance": int((rms >= PCM_RMS_TOL).sum()),
                   "rms_all_channels_off_the_branch_cut_max": float(rms_abs.max()),
                   "untrimmed_max_well": float(rms[well].max(initial=0.0)), "untrimmed_max_all": float(rms.max()),
                   "worst_sample_over_its_bound": float(worst.max()), "worst_sample_not_at_the_branch_cut": float(worst[~turn].max(initial=0.0)),
                   "samples_at_the_branch_cut": int(turn.sum())})
    print("pcm vs float64 oracle %s: %d/%d well conditioned, their untrimmed RMS max %.2e; all channels %.2e; beyond the rounding LSB the worst sample uses %.2f of its bound (%d samples at the FM branch cut set aside)"
          % (what, well.sum(), n_ch, rms[well].max(initial=0.0), rms.max(), worst[~turn].max(initial=0.0), turn.sum()))
    assert well.sum() >= need, (int(well.sum()), need, n_ch)
    c, n = np.unravel_index(int(np.argmax(worst)), worst.shape)
    assert worst.max() <= 1.0, ("sample beyond its bound", int(c), int(n), float(err[c, n]), float(bound[c, n]))
    over = rms - (PCM_RMS_TOL + brms)
    assert over.max() < 0.0, ("channel RMS beyond its bound", int(np.argmax(over)), float(rms[np.argmax(over)]), float(brms[np.argmax(over)]))
    assert rms[well].max(initial=0.0) < PCM_RMS_TOL, (int(np.argmax(rms * well)), float((rms * well).max()))
    assert rms_abs.max() < ABS_RMS_CEILING, ("channel beyond the absolute ceiling", int(np.argmax(rms_abs)), float(rms_abs.max()))
    assert_share_outside_plain()                                # cumulative over this process's sweeps: a gate, not a printout
    return well, rms
