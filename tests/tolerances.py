"""The north_star audio tolerance (PCM within 1e-5 RMS of full scale vs the float64 oracle) and the conditioning rule it
is asserted under -- one statement, used by the CPU sweep (twin vs oracle) and the GPU sweeps (kernels vs oracle).

fp32 conditioning: the chain's roundings sit ~140 dB under the INPUT level, so what is left of a signal that the
channel filter takes 40+ dB down (out-of-band carrier, narrow passband) is only known to ~1e-4 relative, and the FM
discriminator turns that straight into phase.  Rule:
  * a channel is WELL CONDITIONED when its filtered power (the oracle's RSSI) stays within 40 dB of its input power in
    every frame;
  * every well-conditioned channel -- all of them, no best-of selection -- must meet 1e-5 RMS of full scale after its
    0.5 % largest sample deviations are set aside (the instants where a filtered transient crosses zero);
  * every channel, conditioned or not, must stay within 1e-3 RMS.
"""
import numpy as np

PCM_RMS_TOL = 1e-5
LOOSE_TOL = 1e-3


def well_conditioned(iq, rssi_o, smeter_cal_db):
    """iq int16 [n_ch, n, 2]; rssi_o [n_ch, n_frames] of the oracle; smeter_cal_db [n_ch] -> bool [n_ch]"""
    in_db = 10 * np.log10(np.maximum((iq.astype(np.float64) ** 2).sum(axis=2).mean(axis=1), 1e-20) / 32768.0 ** 2)
    in_db = in_db + np.asarray(smeter_cal_db, np.float64)
    return (rssi_o > in_db[:, None] - 40).all(axis=1)


def assert_pcm_within_tolerance(pcm, pcm_o, iq, rssi_o, smeter_cal_db, min_well=None):
    pcm = np.asarray(pcm, np.float64)
    well = well_conditioned(iq, rssi_o, smeter_cal_db)
    if min_well is not None:
        assert well.sum() >= min_well, (int(well.sum()), min_well)
    rms = np.sqrt(((pcm - pcm_o) ** 2).mean(axis=1)) / 32768.0
    dev = np.sort(np.abs(pcm - pcm_o), axis=1)[:, : int(pcm_o.shape[1] * 0.995)]
    rms_trim = np.sqrt((dev ** 2).mean(axis=1)) / 32768.0
    assert rms_trim[well].max(initial=0.0) < PCM_RMS_TOL, (int(np.argmax(rms_trim * well)), float((rms_trim * well).max()))
    assert rms.max() < LOOSE_TOL, (int(np.argmax(rms)), float(rms.max()))
    return well, rms_trim
