"""The NumPy float64 oracle over many channels at once: a process pool over the cores this process may use (spawn: the caller
holds a HIP context).  Round 5 widened the float64 checks at the timed launch shapes from 6-8 channels to a strided 256; one core
does ~300 channel-superframes of the full chain per second, so the pool keeps that to seconds.

*** TEST INFRASTRUCTURE *** (like oracle/ itself): imported by tests only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.join(os.path.dirname(HERE), "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def workers():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:                                            # noqa: BLE001
        pass
    return max(1, min(n, 16))


def _wf_job(args):
    import ssdr_oracle as O
    iq, n_avg, cal_db = args
    lines = iq.reshape(-1, 1024, 2)
    g = O.wf_allowed_diff(lines)
    L = g.shape[0] // n_avg
    return O.wf_sum_lines(lines, n_avg, cal_db), g[: L * n_avg].reshape(L, n_avg, 1024).sum(axis=1)


def _audio_job(args):
    import ssdr_oracle as O
    iq, kw, eps = args
    p = O.ChanParams(**kw)
    if eps is None:
        pcm, rssi = O.audio_chain(iq[None], [p])
        return pcm[0], rssi[0], None
    pcm, rssi, bound = O.audio_chain_with_bound(iq[None], [p], eps)
    return pcm[0], rssi[0], bound[0]


def _pool():
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    return ProcessPoolExecutor(workers(), mp_context=mp.get_context("spawn"))


def wf(iq, n_avg, cal_db=0.0):
    """iq int16 [n_ch, n, 2] -> (wf_sum int32 [lines, n_ch, 1024], allowed steps per summed bin [lines, n_ch, 1024]): per channel
    ssdr_oracle.wf_sum_lines and the guard band of ssdr_oracle.wf_allowed_diff summed over each group"""
    with _pool() as ex:
        out = list(ex.map(_wf_job, [(iq[c], n_avg, cal_db) for c in range(iq.shape[0])], chunksize=max(1, iq.shape[0] // (4 * workers()))))
    return np.stack([o[0] for o in out], axis=1), np.stack([o[1] for o in out], axis=1)


def params_kw(p):
    """an oracle ChanParams (dataclass / namespace) as the dict that rebuilds it in a worker"""
    return dict(vars(p)) if not hasattr(p, "__dataclass_fields__") else {k: getattr(p, k) for k in p.__dataclass_fields__}


def audio(iq, params, eps=None):
    """iq int16 [n_ch, n, 2], params: oracle ChanParams per channel -> (pcm [n_ch, n], rssi [n_ch, frames], bound [n_ch, n] or None)"""
    with _pool() as ex:
        out = list(ex.map(_audio_job, [(iq[c], params_kw(params[c]), eps) for c in range(iq.shape[0])], chunksize=max(1, iq.shape[0] // (4 * workers()))))
    return np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), (None if eps is None else np.stack([o[2] for o in out]))
