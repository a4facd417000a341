#!/usr/bin/env python3
"""Differential fuzz of the ingest hub on the GPU box: the SAME random traffic into a synchronous IQHub and into a pipelined one
(pinned slots, ssdr_feed_submit_from, three streams, post-processing inside the slot pipeline; optionally several superframes per
GPU run) -- ragged per-channel feeds, blocks of channels, in-place reserve / commit, receivers that fall behind until the stall rule
runs without them, parameter and averaging changes in mid-stream, display-state changes of the attached workers.  After a flush the
attached channels' queues must hold the same lines, frames, colours and 48 kHz blocks, item for item, and the hubs' stall / drop
counters must agree.

    python tools/fuzz_hub.py [--first 1] [--count 40] [--steps 150] [--out gpurun_out/fuzz_hub.txt]
"""
import argparse
import os
import queue
import sys
import time
import traceback
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def drain(q):
    out = []
    while True:
        try:
            out.append(q.get_nowait())
        except queue.Empty:
            return out


def one_sequence(S, IQHub, seed, steps):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 3, 7, 20, 48]))
    post = bool(rng.random() < 0.6)
    lazy = bool(rng.random() < 0.5)
    K = int(rng.choice([1, 1, 2]))
    kw = dict(gpu_post=post, lazy=lazy, max_queue=1 << 14, backlog_superframes=8 * K, stall_superframes=3 * K, batch_superframes=K)
    # (round 5: on a lazy hub the pipelined side copies back the listeners' rows only, every other time -- SSDR_FEED_LAZY_OUT)
    hubs = [IQHub(n, **kw), IQHub(n, pipeline=True, depth=int(rng.choice([2, 3, 4])), lazy_out=bool(lazy and rng.random() < 0.5), **kw)]
    listeners = sorted(set(int(c) for c in rng.integers(0, n, max(1, n // 3))))
    compared = 0
    try:
        workers = {}
        for c in listeners:                                        # display state per listener, shared by both hubs (read at every run)
            w = SimpleNamespace(zoom=int(rng.integers(0, 15)), wf_auto_scaling=bool(rng.random() < 0.5), delta_low_db=int(rng.integers(-10, 11)),
                                delta_high_db=int(rng.integers(-10, 11)), low_clip_db=-120.0, high_clip_db=-60.0, dynamic_range=40.0)
            s = SimpleNamespace(volume=int(rng.choice([50, 100, 150])), audio_balance=float(rng.choice([-0.5, 0.0, 0.5])), decay=4000,
                                audio_rec=SimpleNamespace(recording_flag=False))
            workers[c] = (w, s)
            for h in hubs:
                h.attach(c, wf=True, snd=True)
                if post:
                    h.wf_clients[c], h.snd_clients[c] = w, s
        for step in range(steps):
            op = int(rng.integers(0, 12))
            m = int(rng.choice([64, 512, 700, 1024, 1500, 2048]))
            if op <= 3:                                            # one receiver; the last one is often silent for long
                c = int(rng.integers(0, n if step % 50 < 12 else max(n - 1, 1)))
                iq = rng.integers(-6000, 6000, (m, 2)).astype(np.int16)
                for h in hubs:
                    h.feed(c, iq)
            elif op <= 7:                                          # a block of receivers, by copy or in place
                f = int(rng.integers(0, n))
                k = int(rng.integers(1, n - f + 1))
                iq = rng.integers(-6000, 6000, (k, m, 2)).astype(np.int16)
                in_place = rng.random() < 0.4
                for h in hubs:
                    v = h.reserve(f, k) if in_place else None
                    if v is not None and v.shape[1] >= m:
                        v[:, :m] = iq
                        assert h.commit(f, k, m)
                    else:
                        h.feed_block(f, iq)
            elif op == 8:
                c = int(rng.integers(0, n))
                mode = ["am", "usb", "lsb", "cw", "nbfm"][int(rng.integers(0, 5))]
                p = S.default_params(mode, f_shift_hz=float(rng.integers(-3000, 3000)))
                for h in hubs:
                    h.set_params(c, p)
            elif op == 9:
                nn = int(rng.choice([1, 1, 2, 5]))
                for h in hubs:
                    h.set_averaging(nn)
            elif op == 10 and post:
                c = listeners[int(rng.integers(0, len(listeners)))]
                workers[c][0].zoom = int(rng.integers(0, 15))
                workers[c][1].volume = int(rng.choice([50, 100, 150]))
            # op 11: nothing (a pause in the traffic)
        for h in hubs:
            h.flush()
        assert list(hubs[0].stalled) == list(hubs[1].stalled) and list(hubs[0].dropped) == list(hubs[1].dropped)
        assert hubs[0].superframes == hubs[1].superframes
        for c in listeners:
            wa, wb = drain(hubs[0].wf_queue[c]), drain(hubs[1].wf_queue[c])
            assert len(wa) == len(wb), ("lines", c, len(wa), len(wb))
            for (la, na, pa), (lb, nb, pb) in zip(wa, wb):
                assert na == nb and np.array_equal(la, lb), ("line", c)
                assert (pa is None) == (pb is None), ("post", c)
                if pa is not None:
                    assert np.array_equal(pa[0], pb[0], equal_nan=True) and np.array_equal(np.array(pa[1:], np.float64), np.array(pb[1:], np.float64), equal_nan=True), ("colour", c)
            sa, sb = drain(hubs[0].snd_queue[c]), drain(hubs[1].snd_queue[c])
            assert len(sa) == len(sb), ("frames", c, len(sa), len(sb))
            for fa, fb in zip(sa, sb):
                assert np.array_equal(np.asarray(fa), np.asarray(fb)) and fa.rssi == fb.rssi and fa.adc_overflow == fb.adc_overflow, ("frame", c)
                assert (fa.play_block is None) == (fb.play_block is None)
                if fa.play_block is not None:
                    assert np.array_equal(fa.play_block, fb.play_block), ("play", c)
            compared += len(wa) + len(sa)
        return compared, int(hubs[0].stalled.sum())
    finally:
        for h in hubs:
            h.close()


def wire_bodies(rng, iq):
    """iq int16 [k, f * 512, 2] -> SND bodies uint8 [k, f, 2065] as they come off the socket (kiwi/client.py:443-454): 7 header bytes,
    a 10-byte GNSS stamp, 512 big-endian I,Q pairs"""
    k, f = iq.shape[0], iq.shape[1] // 512
    b = rng.integers(0, 256, (k, f, 2065)).astype(np.uint8)            # header and stamp: anything
    b[:, :, 17:] = iq.astype(">i2").view(np.uint8).reshape(k, f, 2048)
    return b


def one_wire_sequence(S, IQHub, seed, steps):
    """three hubs, one traffic: SND bodies into a synchronous wire hub and a pipelined wire hub (header strip and byte swap on the
    device), the decoded samples into a synchronous sample hub"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 5, 12, 33]))
    K = int(rng.choice([1, 2]))
    kw = dict(gpu_post=False, lazy=False, max_queue=1 << 14, backlog_superframes=8 * K, stall_superframes=3 * K, batch_superframes=K)
    hubs = [IQHub(n, wire=True, **kw), IQHub(n, wire=True, pipeline=True, depth=int(rng.choice([2, 3])), **kw), IQHub(n, **kw)]
    compared = 0
    try:
        for step in range(steps):
            op = int(rng.integers(0, 10))
            f = int(rng.integers(1, 5))
            if op <= 6:
                first = int(rng.integers(0, n)) if op > 2 else int(rng.integers(0, n if step % 40 < 10 else max(n - 1, 1)))
                k = 1 if op <= 2 else int(rng.integers(1, n - first + 1))
                iq = rng.integers(-9000, 9000, (k, f * 512, 2)).astype(np.int16)
                bodies = wire_bodies(rng, iq)
                in_place = rng.random() < 0.3
                for h in hubs[:2]:
                    v = h.reserve(first, k) if in_place else None
                    if v is not None and v.shape[1] >= f:
                        v[:, :f] = bodies
                        assert h.commit(first, k, f)
                    else:
                        h.feed_wire_block(first, bodies)
                hubs[2].feed_block(first, iq)
            elif op == 7:
                c = int(rng.integers(0, n))
                p = S.default_params(["am", "usb", "lsb", "cw", "nbfm"][int(rng.integers(0, 5))], f_shift_hz=float(rng.integers(-3000, 3000)))
                for h in hubs:
                    h.set_params(c, p)
            elif op == 8:
                nn = int(rng.choice([1, 1, 3]))
                for h in hubs:
                    h.set_averaging(nn)
        for h in hubs:
            h.flush()
        for h in hubs[1:]:
            assert list(h.stalled) == list(hubs[0].stalled) and h.superframes == hubs[0].superframes
        assert hubs[0]._U * 512 == hubs[2]._U                             # a wire hub counts frames, a sample hub samples
        assert list(hubs[0].dropped) == list(hubs[1].dropped) and list(hubs[0].dropped * 512) == list(hubs[2].dropped)
        for c in range(n):
            w = [drain(h.wf_queue[c]) for h in hubs]
            s = [drain(h.snd_queue[c]) for h in hubs]
            for other in (1, 2):
                assert len(w[0]) == len(w[other]) and len(s[0]) == len(s[other]), ("counts", c, other)
                for (la, na, _), (lb, nb, _) in zip(w[0], w[other]):
                    assert na == nb and np.array_equal(la, lb), ("line", c, other)
                for fa, fb in zip(s[0], s[other]):
                    assert np.array_equal(np.asarray(fa), np.asarray(fb)) and fa.rssi == fb.rssi and fa.adc_overflow == fb.adc_overflow, ("frame", c, other)
            compared += len(w[0]) + len(s[0])
        return compared, int(hubs[0].stalled.sum())
    finally:
        for h in hubs:
            h.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=1)
    ap.add_argument("--count", type=int, default=40)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import supersdr_amd as S
    import supersdr_amd.engine as _E
    _E.DEFAULT_CHAIN_FLOORS = (0, 0)                    # small batches through the one-read kernels too (ssdr_set_chain_floors)
    from supersdr_amd.workers import IQHub
    lines, bad, items, stalls, t0 = [], 0, 0, 0, time.time()
    for seed in range(a.first, a.first + a.count):
        try:
            c, s = (one_wire_sequence if seed % 4 == 3 else one_sequence)(S, IQHub, seed, a.steps)     # every fourth: SND bodies (wire hubs)
            items += c
            stalls += s
        except Exception as e:                                     # noqa: BLE001 -- reported per seed, the run goes on
            bad += 1
            tb = traceback.format_exc().strip().splitlines()
            lines.append("  seed %d: %s: %s | %s" % (seed, type(e).__name__, e, " / ".join(x.strip() for x in tb[-4:-1])[:300]))
            print(lines[-1], flush=True)
    lines.append("differential hub fuzz (synchronous vs pipelined IQHub; every fourth sequence: SND bodies into wire hubs vs samples): seeds %d..%d, %d steps each: %d sequences differed; %d queued lines / frames "
                 "compared, %d stalled superframes on the way (%.0f s)" % (a.first, a.first + a.count - 1, a.steps, bad, items, stalls, time.time() - t0))
    text = "\n".join(lines) + "\n"
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(text)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
