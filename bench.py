#!/usr/bin/env python3
"""bench.py -- SuperSDR hot path on MI355X: real-time IQ channels sustained (WF + demod).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full|wf|mixed]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic IQ that is already
resident in HBM: `channels` receivers x `superframes` x 1024 samples (one waterfall
line + two 512-sample audio frames per superframe).  Channels shard block-wise across
ranks with NO data-path collective (weak scaling: per-GPU work is fixed); torch.distributed
is used only for the barrier and the max-over-ranks of the wall time.

Workloads (BASELINE.json configs):
  full  (default) configs[2]: 65536 ch/GPU, WF (N=1) + AM demod + AGC   <- the metric's config
  wf              configs[1]: 4096 ch/GPU, waterfall only, 256 lines per launch
  mixed           configs[3]: 65536 ch/GPU, AM/USB/LSB/NBFM by c mod 4, 10x time binning
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RT_SUPERFRAMES_PER_S = 12000.0 / 1024.0          # 11.71875
HBM_PEAK_GBPS = 8000.0                           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    #          channels  superframes  n_avg  modes                  wf    audio
    "full":  (65536,     16,          1,     ("am",),               True, True),      # >= 16 superframes
    "wf":    (4096,      256,         1,     ("am",),               True, False),
    "mixed": (65536,     10,          10,    ("am", "usb", "lsb", "nbfm"), True, True),
    # configs[4]: 2^20 channels in total, 2^20 / N per GPU (strong scaling; 16 GiB of input on one GPU at N = 1)
    "million": (1 << 20, 4,           1,     ("am",),               True, True),
}


def cpu_baseline(workload, budget_s=12.0):
    """The oracle's fp32 C twin (a port of the same algorithm) timed on the host cores on a
    bounded sample of the same workload.  Reported, never the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import twinlib
    import ssdr_oracle as O
    from concurrent.futures import ThreadPoolExecutor
    channels, sframes, n_avg, modes, do_wf, do_audio = WORKLOADS[workload]
    twin = twinlib.load()
    cores = os.cpu_count() or 1
    sf = min(sframes, 4)

    def make(nch):
        iq = O.synth_iq(nch, sf * 1024, seed=7)
        consts = np.zeros(nch, twinlib.CONSTS_DTYPE)
        taps = np.zeros((nch, 128), np.float32)
        for c in range(nch):
            m = modes[c % len(modes)]
            lc, hc = {"am": (-6000, 6000), "usb": (30, 3000), "lsb": (-3000, -30), "nbfm": (-6000, 6000)}[m]
            k = O.compile_params(O.ChanParams(mode=m, f_shift_hz=((c * 37) % 97 - 48) * 100.0, low_cut=lc, high_cut=hc))
            for f in ("mode", "ntap", "dphi1", "dphi2", "wf_cal_lin", "smeter_cal_db", "agc_c0", "agc_c1",
                      "agc_knee", "agc_delta8", "hang_frames", "fir_flags"):
                consts[f][c] = k[f]
            consts["ntap8"][c] = (k["ntap"] + 7) // 8 * 8
            taps[c] = k["taps"]
        return iq, consts, taps

    def work(args):
        iq, consts, taps = args
        if do_wf:
            twin.wf(iq, n_avg if sf % n_avg == 0 else 1, consts["wf_cal_lin"])
        if do_audio:
            st, hist = twinlib.fresh_state(consts)
            twin.audio(iq, consts, taps, st, hist)

    # calibrate on a small block, then size the sample for ~budget_s of CPU work per core
    blk = make(8)
    t0 = time.perf_counter()
    work(blk)
    per_ch = (time.perf_counter() - t0) / 8
    nch_core = int(max(8, min(2048, budget_s / max(per_ch, 1e-9))))
    blocks = [make(nch_core) if i == 0 else None for i in range(cores)]
    blocks = [blocks[0]] * cores                      # same bytes per worker; ctypes releases the GIL
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, blocks))
    wall = time.perf_counter() - t0
    units = nch_core * cores * sf
    out = {"value": units / wall / RT_SUPERFRAMES_PER_S, "unit": "rt_channels", "cores": cores, "kind": "port",
           "sample": "%d ch x %d superframes per core on %d threads, oracle/ssdr_twin.c (fp32 C port), %.1f s"
                     % (nch_core, sf, cores, wall)}
    # north_star also asks for the NumPy path: the float64 oracle (vectorised NumPy, one process) on a few channels
    nn = 16
    iq = O.synth_iq(nn, sf * 1024, seed=7)
    prm = [O.ChanParams(mode=modes[c % len(modes)], f_shift_hz=((c * 37) % 97 - 48) * 100.0) for c in range(nn)]
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 3.0:
        if do_wf:
            for c in range(nn):
                O.wf_sum_lines(iq[c].reshape(-1, 1024, 2), 1, 0.0)
        if do_audio:
            O.audio_chain(iq, prm)
        reps += 1
    wall_np = time.perf_counter() - t0
    out["numpy_oracle"] = {"value": nn * sf * reps / wall_np / RT_SUPERFRAMES_PER_S, "unit": "rt_channels", "cores": 1,
                           "sample": "%d ch x %d superframes x %d passes, oracle/ssdr_oracle.py (NumPy float64), %.1f s"
                                     % (nn, sf, reps, wall_np)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spinup", type=float, default=1.5,
                    help="seconds of untimed steps before the W warmup steps: an idle MI355X sits at ~100 MHz and takes "
                         "~0.5 s of load to settle at its sustained clock (profiles/README.md, clock ramp)")
    ap.add_argument("--workload", default="full", choices=sorted(WORKLOADS))
    ap.add_argument("--channels", type=int, default=0, help="override channels per GPU")
    ap.add_argument("--superframes", type=int, default=0, help="override superframes per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-feed", type=int, default=0,
                    help="1: inputs come from (pinned) host memory (2: as SND wire bodies, unpacked on the device) and results go back to it through the pipelined feed "
                         "(ssdr_feed_*): the PCIe-inclusive rate of DESIGN.md, never the headline value")
    ap.add_argument("--concurrent", type=int, default=0, help="bit 0: audio stage on a second stream beside the waterfall kernel; bit 1: the audio stage's per-path kernels one after the other")
    args = ap.parse_args()

    import torch
    from supersdr_amd.dist import Rendezvous, env_rank
    rank, local_rank, world = env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    if "SSDR_BENCH_DEVICE" in os.environ:          # test hook: several ranks on one GPU (exercises the N>1 control flow on a 1-GPU box)
        local_rank = int(os.environ["SSDR_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    rdv = Rendezvous("nccl", torch.device("cuda", local_rank))     # barrier + max-over-ranks only

    import supersdr_amd as S
    from supersdr_amd import _lib as L

    channels, sframes, n_avg, modes, do_wf, do_audio = WORKLOADS[args.workload]
    if args.workload == "million":
        channels //= world
    channels = args.channels or channels
    sframes = args.superframes or sframes
    n_frames = 2 * sframes

    eng = S.SsdrEngine(channels, device=local_rank)
    params = [S.default_params(modes[c % len(modes)], f_shift_hz=(((rank * channels + c) * 37) % 97 - 48) * 100.0)
              for c in range(min(channels, 388))]          # the parameter pattern repeats every 4*97 channels
    for first in range(0, channels, len(params)):
        eng.set_params(first, params[: min(len(params), channels - first)])
    eng.reset_state()
    eng.set_averaging(n_avg)
    eng.set_concurrent(args.concurrent)
    eng.synth_iq(n_frames, seed=0x5D5D, first_channel_id=rank * channels)    # resident in HBM from here on
    eng.sync()

    def step():
        if do_wf:
            eng.run_wf(fetch=False)
        if do_audio:
            eng.run_audio(fetch=False)

    if args.host_feed:
        depth = 3
        host_batch = eng.read_input()                     # one synthetic batch, replayed from pinned host memory
        eng.feed_open(n_frames, depth, wire=(args.host_feed == 2))
        if args.host_feed == 2:                           # SND bodies as they come off the socket: 17-byte header + big-endian I,Q
            bodies = np.zeros((channels, n_frames, 2065), np.uint8)
            bodies[:, :, 17:] = host_batch.reshape(channels, n_frames, 512, 2).astype(">i2").view(np.uint8).reshape(channels, n_frames, 2048)
            host_batch = bodies
        else:
            host_batch = host_batch.reshape(channels, -1, 2)
        for _ in range(depth):                            # fill every slot once: the timed loop measures transport + kernels
            eng.feed_slot()[:] = host_batch
            eng.feed_submit()
        for _ in range(depth):
            eng.feed_collect()
        inflight = [0]

        def step():                                       # noqa: F811  (steady state: one submit, one collect)
            eng.feed_slot()
            eng.feed_submit()
            inflight[0] += 1
            if inflight[0] == depth:
                eng.feed_collect()
                inflight[0] -= 1

    barrier = rdv.barrier

    t_spin = time.perf_counter()            # clock spin-up from the idle state, then the W warmup steps proper
    while time.perf_counter() - t_spin < args.spinup:
        for _ in range(8):
            step()
        eng.sync()
    for _ in range(args.warmup):
        step()
    eng.sync()
    eng.set_profiling(True)                 # HIP-event pair around every launch, on the launch stream, no host sync
    for k in (L.K_WF, L.K_AUDIO):
        eng.kernel_stats(k, reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if args.host_feed:
        while inflight[0]:
            eng.feed_collect()
            inflight[0] -= 1
    eng.sync()
    torch.cuda.synchronize()
    barrier()
    wall = rdv.max_over_ranks(time.perf_counter() - t0)

    wf_ms, wf_n = eng.kernel_stats(L.K_WF)
    au_ms, au_n = eng.kernel_stats(L.K_AUDIO)
    units = channels * sframes * args.steps * world            # channel-superframes, whole job
    value = units / wall / RT_SUPERFRAMES_PER_S

    # algorithmic bytes per launch: WF 4096 B in + 2048/N B out per line;
    # audio 2048 B in + 1024 B out per 512-sample frame
    wf_bytes = channels * sframes * (4096.0 + 2048.0 / n_avg)
    au_bytes = channels * n_frames * 3072.0
    stages = {}
    if wf_n:
        avg = wf_ms / wf_n
        stages["ssdr_wf_kernel"] = {"avg_ms": avg, "launches": wf_n, "bytes": wf_bytes, "GBps": wf_bytes / avg / 1e6}
    if au_n:
        avg = au_ms / au_n
        stages["ssdr_audio_kernel"] = {"avg_ms": avg, "launches": au_n, "bytes": au_bytes, "GBps": au_bytes / avg / 1e6}
    dom = max(stages, key=lambda k: stages[k]["avg_ms"])

    # HBM bytes per launch from the PMC passes committed under profiles/ (collected with rocprofv3 in separate
    # runs, corrected as MI355X_MICROARCH.md prescribes; tools/profile_round.sh + tools/traffic_json.py).
    # Only used when it was measured on exactly this workload shape; otherwise null.
    measured = {}
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json"))):
            t = json.load(open(path))
            if (t.get("workload"), t.get("channels_per_gpu"), t.get("superframes_per_step")) == (args.workload, channels, sframes):
                measured = {k: v["hbm_bytes_per_launch"] for k, v in t["kernels"].items()}
    except Exception:
        measured = {}

    def roof(name):
        s = stages[name]
        return {"kernel": name, "bound": "hbm", "achieved": s["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": s["GBps"] / HBM_PEAK_GBPS, "traffic": measured.get(name), "avg_kernel_ms": s["avg_ms"],
                "algorithmic_bytes_per_launch": s["bytes"]}

    out = {
        "metric": "real-time IQ channels sustained (WF+demod)", "value": value, "unit": "rt_channels",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if args.workload == "million" else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": {"full": "65536 channels full chain (WF + AM demod + AGC), BASELINE configs[2]",
                                "wf": "4096 channels batched 1024-pt FFT + log-mag waterfall only, BASELINE configs[1]",
                                "mixed": "65536 channels mixed AM/USB/LSB/NBFM + 10x time binning, BASELINE configs[3]",
                                "million": "2^20 channels full chain in total, channel-sharded, BASELINE configs[4]"}[args.workload],
                   "channels_per_gpu": channels, "superframes_per_step": sframes, "averaging_n": n_avg,
                   "clock_spinup_s": args.spinup,
                   "input": "pinned host memory, pipelined H2D / kernels / D2H (PCIe-inclusive)" if args.host_feed else "resident in HBM",
                   "sharding": "channel blocks per GPU, no collectives", "rendezvous": rdv.backend if world > 1 else "none"},
        "roofline": roof(dom),
    }
    if "ssdr_wf_kernel" in stages:
        out["roofline_fft"] = roof("ssdr_wf_kernel")
    if "ssdr_audio_kernel" in stages and dom != "ssdr_audio_kernel":
        out["roofline_audio"] = roof("ssdr_audio_kernel")
    eng.close()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    rdv.close()


if __name__ == "__main__":
    main()
