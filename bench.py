#!/usr/bin/env python3
"""bench.py -- SuperSDR hot path on MI355X: real-time IQ channels sustained (WF + demod).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full|wf|mixed|million]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic IQ that is already
resident in HBM: `channels` receivers x `superframes` x 1024 samples (one waterfall
line + two 512-sample audio frames per superframe).  Channels shard block-wise across
ranks with NO data-path collective; torch.distributed is used only for the barrier and
the max-over-ranks of the wall time.

`--gpus N` is the number of ranks.  Under torch.distributed.run the ranks exist already
(WORLD_SIZE must equal N, anything else is an error); without it `--gpus N` with N > 1
starts the N ranks itself, one process per GPU, and relays rank 0's JSON line.

Workloads (BASELINE.json configs):
  full  (default) configs[2]: 65536 ch/GPU, WF (N=1) + AM demod + AGC   <- the metric's config
  wf              configs[1]: 4096 ch/GPU, waterfall only, 256 lines per launch
  mixed           configs[3]: 65536 ch/GPU, AM/USB/LSB/NBFM by c mod 4, 10x time binning
  million         configs[4]: 2^20 channels in total, 2^20/N per GPU (strong scaling)
The default run also times `wf` and `mixed` briefly after the main measurement and reports
them, each with its own rooflines, under "extra" in the same JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
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
WORKLOAD_TEXT = {"full": "65536 channels full chain (WF + AM demod + AGC), BASELINE configs[2]",
                 "wf": "4096 channels batched 1024-pt FFT + log-mag waterfall only, BASELINE configs[1]",
                 "mixed": "65536 channels mixed AM/USB/LSB/NBFM + 10x time binning, BASELINE configs[3]",
                 "million": "2^20 channels full chain in total, channel-sharded, BASELINE configs[4]"}
PATH_NAMES = ("FIR", "shift", "AM-shift")        # ssdr_audio_kernel<0|1|2>


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle timed on the host cores on a bounded sample.  Reported, never the target.
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(workload, budget_s=5.0):
    """the NumPy float64 oracle (the code that defines parity) on all host cores
    (`multiprocessing`, os.cpu_count() workers, channels block-sharded) and on one core; next to it the oracle's
    fp32 C twin on all host threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import multiprocessing as mp
    import twinlib
    import ssdr_oracle as O
    import cpu_bench as CB                       # oracle/cpu_bench.py (test infrastructure, like the oracle itself)
    from concurrent.futures import ThreadPoolExecutor
    channels, sframes, n_avg, modes, do_wf, do_audio = WORKLOADS[workload]
    cores = os.cpu_count() or 1
    sf_np = min(sframes, 16)
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # one thread per worker process
        os.environ[v] = "1"

    # ---- NumPy oracle: one process first (calibration == the per-core figure), then all cores
    CB.run_block((0, 1, 1, 1, modes, do_wf, do_audio))           # imports and first-call costs out of the way
    t0 = time.perf_counter()
    n1 = CB.run_block((0, 4, sf_np, n_avg, modes, do_wf, do_audio))
    t1 = time.perf_counter() - t0
    one = {"value": n1 * sf_np / t1 / RT_SUPERFRAMES_PER_S, "cores": 1, "sample": "4 ch x %d superframes, one process" % sf_np}
    per = int(min(4096 // cores if cores <= 4096 else 1, max(1, budget_s / max(t1 / 4, 1e-9))))   # 4096 ch x 16 superframes
    per = max(per, 1)
    jobs = [(w * per, per, sf_np, n_avg, modes, do_wf, do_audio) for w in range(cores)]
    ctx = mp.get_context("spawn")                # never fork a process that holds a HIP context
    with ctx.Pool(cores) as pool:
        # every worker up, imported and through one small block before the clock starts
        pool.map(CB.run_block, [(0, 1, 1, 1, modes, do_wf, do_audio)] * (4 * cores), chunksize=1)
        t0 = time.perf_counter()
        done = sum(pool.map(CB.run_block, jobs, chunksize=1))
        wall_np = time.perf_counter() - t0
    box = done * sf_np / wall_np / RT_SUPERFRAMES_PER_S
    out = {"value": box, "unit": "rt_channels", "cores": cores, "kind": "port", "per_core": box / cores,
           "sample": "%d ch x %d superframes, block-sharded over %d processes, oracle/ssdr_oracle.py (NumPy float64), %.1f s"
                     % (done, sf_np, cores, wall_np),
           "one_process": one}

    # ---- the fp32 C twin on all host threads (ctypes releases the GIL)
    twin = twinlib.load()
    sf = min(sframes, 4)

    def make(nch):
        iq = O.synth_iq(nch, sf * 1024, seed=7)
        consts = np.zeros(nch, twinlib.CONSTS_DTYPE)
        taps = np.zeros((nch, 128), np.float32)
        for c in range(nch):
            k = O.compile_params(CB.chan_params(c, modes))
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

    blk = make(8)
    t0 = time.perf_counter()
    work(blk)
    per_ch = (time.perf_counter() - t0) / 8
    nch_core = int(max(8, min(1024, 0.5 * budget_s / max(per_ch, 1e-9))))       # threads share cores: half the single-thread estimate
    block = make(nch_core)                        # same bytes for every worker
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, [block] * cores))
    wall = time.perf_counter() - t0
    out["c_twin"] = {"value": nch_core * cores * sf / wall / RT_SUPERFRAMES_PER_S, "unit": "rt_channels", "cores": cores,
                     "kind": "port",
                     "sample": "%d ch x %d superframes per thread on %d threads, oracle/ssdr_twin.c (fp32 C port), %.1f s"
                               % (nch_core, sf, cores, wall)}
    return out


# ---------------------------------------------------------------------------------------------------------------
# launching the ranks
# ---------------------------------------------------------------------------------------------------------------
def check_world(gpus):
    """--gpus against the environment: (rank, local_rank, world) or SystemExit.  world == 0 means: start the ranks."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU "
                             "(python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d), "
                             "or run `python bench.py --gpus %d` alone and let it start the ranks" % (gpus, world, gpus, gpus, gpus))
        return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world
    return (0, 0, 1) if gpus == 1 else (0, 0, 0)


def spawn_ranks(gpus, argv):
    """One process per GPU; rank 0's stdout (the JSON line) is ours.  No data moves between them."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %r" % rcs)


# ---------------------------------------------------------------------------------------------------------------
# one measurement
# ---------------------------------------------------------------------------------------------------------------
def measure(S, L, torch, rdv, rank, world, local_rank, workload, channels, sframes, steps, warmup, spinup,
            concurrent=0, host_feed=0, hop=1024, fused=0):
    """Spin the clocks up, W warm-up steps, then exactly `steps` timed steps between barrier + synchronize pairs.
    -> dict(value, ms_per_step, stages)."""
    _, _, n_avg, modes, do_wf, do_audio = WORKLOADS[workload]
    n_frames = 2 * sframes
    eng = S.SsdrEngine(channels, device=local_rank)
    params = [S.default_params(modes[c % len(modes)], f_shift_hz=(((rank * channels + c) * 37) % 97 - 48) * 100.0)
              for c in range(min(channels, 388))]          # the parameter pattern repeats every 4*97 channels
    for first in range(0, channels, len(params)):
        eng.set_params(first, params[: min(len(params), channels - first)])
    eng.reset_state()
    eng.set_hop(hop)
    eng.set_fused(fused)
    eng.set_averaging(n_avg)
    eng.set_concurrent(concurrent)
    eng.synth_iq(n_frames, seed=0x5D5D, first_channel_id=rank * channels)    # resident in HBM from here on
    eng.sync()

    def step():
        if do_wf:
            eng.run_wf(fetch=False)
        if do_audio:
            eng.run_audio(fetch=False)

    if fused and do_wf and do_audio:
        def step():                                       # noqa: F811  (one kernel when the configuration allows it)
            eng.run_chain()

    inflight = [0]
    if host_feed:
        depth = 3
        host_batch = eng.read_input()                     # one synthetic batch, replayed from pinned host memory
        eng.feed_open(n_frames, depth, wire=(host_feed == 2))
        if host_feed == 2:                                # SND bodies as they come off the socket: 17-byte header + big-endian I,Q
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

        def step():                                       # noqa: F811  (steady state: one submit, one collect)
            eng.feed_slot()
            eng.feed_submit()
            inflight[0] += 1
            if inflight[0] == depth:
                eng.feed_collect()
                inflight[0] -= 1

    t_spin = time.perf_counter()            # clock spin-up from the idle state, then the W warmup steps proper
    while time.perf_counter() - t_spin < spinup:
        for _ in range(8):
            step()
        eng.sync()
    for _ in range(warmup):
        step()
    eng.sync()
    eng.set_profiling(True)                 # HIP-event pair around every launch, on the launch stream, no host sync
    for k in (L.K_WF, L.K_AUDIO, L.K_FUSED):
        eng.kernel_stats(k, reset=True)
    rdv.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    while inflight[0]:
        eng.feed_collect()
        inflight[0] -= 1
    eng.sync()
    torch.cuda.synchronize()
    rdv.barrier()
    wall = rdv.max_over_ranks(time.perf_counter() - t0)

    wf_ms, wf_n = eng.kernel_stats(L.K_WF)
    au_ms, au_n = eng.kernel_stats(L.K_AUDIO)
    fu_ms, fu_n = eng.kernel_stats(L.K_FUSED)
    paths = eng.audio_paths()
    eng.close()
    units = channels * sframes * steps * world                 # channel-superframes, whole job
    # algorithmic bytes per launch: WF 4096 B in + 2048/N B out per line;
    # audio 2048 B in + 1024 B out per 512-sample frame
    stages = {}
    if wf_n:
        avg = wf_ms / wf_n
        # per line: hop 1024 reads 4096 B, hop 512 reads 2048 new bytes (the other half-line was the previous line's); 2048/N out
        lines = channels * sframes * (2 if hop == 512 else 1)
        b = lines * ((2048.0 if hop == 512 else 4096.0) + 2048.0 / n_avg)
        stages["wf"] = {"kernel": "ssdr_wf_kernel<%s, %s>" % ("true" if n_avg > 1 else "false", "true" if hop == 512 else "false"),
                        "avg_ms": avg, "launches": wf_n, "bytes": b, "GBps": b / avg / 1e6, "lines_per_launch": lines}
    if au_n:
        avg = au_ms / au_n
        b = channels * n_frames * 3072.0
        live = [p for p in range(3) if paths[p]]
        name = ("ssdr_audio_kernel<%d>" % live[0]) if len(live) == 1 else \
               "audio stage: " + " + ".join("ssdr_audio_kernel<%d> (%s, %d ch)" % (p, PATH_NAMES[p], paths[p]) for p in live) + \
               (" one after the other" if concurrent & 2 else " side by side")
        stages["audio"] = {"kernel": name, "avg_ms": avg, "launches": au_n, "bytes": b, "GBps": b / avg / 1e6}
    if fu_n:
        avg = fu_ms / fu_n
        b = channels * sframes * 8192.0                 # SURVEY.md 8d, fused budget at N = 1: 4096 in + 2048 + 2048 out
        stages["fused"] = {"kernel": "ssdr_fused_am_kernel", "avg_ms": avg, "launches": fu_n, "bytes": b, "GBps": b / avg / 1e6}
    return {"value": units / wall / RT_SUPERFRAMES_PER_S, "ms_per_step": wall / steps * 1e3, "stages": stages,
            "n_avg": n_avg}


def pmc_traffic(workload, channels, sframes, hop=1024):
    """HBM bytes per launch from the PMC passes committed under profiles/ (collected with rocprofv3 in separate
    runs, corrected as MI355X_MICROARCH.md prescribes; tools/profile_round.sh + tools/traffic_json.py).
    Only used when it was measured on exactly this workload shape; the newest round wins."""
    import glob
    found, src = {}, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json"))):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if (t.get("workload"), t.get("channels_per_gpu"), t.get("superframes_per_step"), t.get("wf_hop", 1024)) == (workload, channels, sframes, hop):
            found, src = {k: v["hbm_bytes_per_launch"] for k, v in t["kernels"].items()}, os.path.basename(path)
    return found, src


def stage_traffic(traffic, stage):
    """PMC bytes of a stage: its kernel's, or the sum over the kernels a multi-kernel audio stage names"""
    import re
    names = re.findall(r"ssdr_\w+_kernel<[^>]*>", stage["kernel"])
    vals = [traffic.get(n) for n in names]
    return sum(vals) if vals and all(v is not None for v in vals) else None


def roofline(stage, traffic=None, src=None):
    r = {"kernel": stage["kernel"], "bound": "hbm", "achieved": stage["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": stage["GBps"] / HBM_PEAK_GBPS, "traffic": traffic, "avg_kernel_ms": stage["avg_ms"],
         "algorithmic_bytes_per_launch": stage["bytes"]}
    if traffic is not None:
        r["traffic_source"] = "profiles/" + src
    return r


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
    ap.add_argument("--no-extra", action="store_true", help="skip the short wf / mixed measurements after the main one")
    ap.add_argument("--host-feed", type=int, default=0,
                    help="1: inputs come from (pinned) host memory (2: as SND wire bodies, unpacked on the device) and results go back to it through the pipelined feed "
                         "(ssdr_feed_*): the PCIe-inclusive rate of DESIGN.md, never the headline value")
    ap.add_argument("--concurrent", type=int, default=0, help="bit 0: audio stage on a second stream beside the waterfall kernel; bit 1: the audio stage's per-path kernels one after the other")
    ap.add_argument("--fused", type=int, default=0,
                    help="1: ssdr_run_chain with the fused superframe kernel where the configuration allows it (full-band AM, N = 1)")
    ap.add_argument("--hop", type=int, default=1024, choices=[512, 1024],
                    help="samples between waterfall lines: 512 = 23.4 lines/s, the reference's waterfall rate (utils_supersdr.py:597)")
    ap.add_argument("--dry-run", action="store_true",
                    help="control flow only (ranks, rendezvous over gloo, channel blocks, JSON line), no GPU work: the CPU test of --gpus")
    args = ap.parse_args()

    rank, local_rank, world = check_world(args.gpus)
    if world == 0:
        return spawn_ranks(args.gpus, sys.argv[1:])

    import torch            # before libssdr.so: the library then binds to the HIP runtime torch has already loaded (one runtime per process)
    from supersdr_amd.dist import Rendezvous, channel_block
    channels, sframes, n_avg, modes, do_wf, do_audio = WORKLOADS[args.workload]
    if args.workload == "million":
        channels = channel_block(rank, world, channels)[1]
    channels = args.channels or channels
    sframes = args.superframes or sframes

    if args.dry_run:
        rdv = Rendezvous("gloo", None)
        rdv.barrier()
        total = rdv.sum_over_ranks(channels)
        wall = rdv.max_over_ranks(1e-3 * (rank + 1))
        if rank == 0:
            print(json.dumps({"metric": "real-time IQ channels sustained (WF+demod)", "value": None, "unit": "rt_channels",
                              "n_gpus": world, "dry_run": True, "channels_total": int(total), "max_wall": wall,
                              "config": {"workload": WORKLOAD_TEXT[args.workload], "channels_per_gpu": channels,
                                         "rendezvous": rdv.backend if world > 1 else "none"}}), flush=True)
        rdv.close()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    if "SSDR_BENCH_DEVICE" in os.environ:          # test hook: several ranks on one GPU (exercises the N>1 control flow on a 1-GPU box)
        local_rank = int(os.environ["SSDR_BENCH_DEVICE"])
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible); --gpus must not exceed the GPUs of the node"
                         % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    rdv = Rendezvous("nccl", torch.device("cuda", local_rank))     # barrier + max-over-ranks only

    import supersdr_amd as S
    from supersdr_amd import _lib as L

    m = measure(S, L, torch, rdv, rank, world, local_rank, args.workload, channels, sframes, args.steps, args.warmup,
                args.spinup, args.concurrent, args.host_feed, args.hop, args.fused)
    stages = m["stages"]
    dom = max(stages, key=lambda k: stages[k]["avg_ms"])
    traffic, src = pmc_traffic(args.workload, channels, sframes, args.hop)

    out = {
        "metric": "real-time IQ channels sustained (WF+demod)", "value": m["value"], "unit": "rt_channels",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
        "higher_is_better": True, "scaling": "strong" if args.workload == "million" else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOAD_TEXT[args.workload],
                   "channels_per_gpu": channels, "superframes_per_step": sframes, "averaging_n": n_avg, "wf_hop": args.hop,
                   "clock_spinup_s": args.spinup,
                   "input": "pinned host memory, pipelined H2D / kernels / D2H (PCIe-inclusive)" if args.host_feed else "resident in HBM",
                   "sharding": "channel blocks per GPU, no collectives", "rendezvous": rdv.backend if world > 1 else "none"},
        "roofline": roofline(stages[dom], stage_traffic(traffic, stages[dom]), src),
    }
    for k, label in (("wf", "roofline_fft"), ("audio", "roofline_audio"), ("fused", "roofline_fused")):
        if k in stages and k != dom:
            out[label] = roofline(stages[k], stage_traffic(traffic, stages[k]), src)
    if "wf" in stages and "audio" in stages:
        # the chain as one unit (fused budget): the input counted once, 4096 + 2048/N + 2048 B per channel-superframe
        b = channels * sframes * (4096.0 + (2 if args.hop == 512 else 1) * 2048.0 / n_avg + 2048.0)
        ms = stages["wf"]["avg_ms"] + stages["audio"]["avg_ms"]
        out["roofline_chain"] = {"kernel": "waterfall + audio stage", "bound": "hbm", "achieved": b / ms / 1e6, "peak": HBM_PEAK_GBPS,
                                 "unit": "GB/s", "frac": b / ms / 1e6 / HBM_PEAK_GBPS, "traffic": None, "avg_kernel_ms": ms,
                                 "algorithmic_bytes_per_launch": b}

    if rank == 0 and world == 1 and args.workload == "full" and not args.no_extra and not args.host_feed:
        # configs[1] and configs[3] in the same driver-timed line: shorter runs, each with its own rooflines
        extra = {}
        for wl in ("wf", "mixed"):
            ch, sf = WORKLOADS[wl][0], WORKLOADS[wl][1]
            e = measure(S, L, torch, rdv, rank, world, local_rank, wl, ch, sf, max(20, args.steps // 2), 2, 0.5, args.concurrent)
            tr, tsrc = pmc_traffic(wl, ch, sf)
            extra[wl] = {"workload": WORKLOAD_TEXT[wl], "value": e["value"], "unit": "rt_channels", "ms_per_step": e["ms_per_step"],
                         "steps": max(20, args.steps // 2), "channels_per_gpu": ch, "superframes_per_step": sf, "averaging_n": e["n_avg"],
                         "rooflines": [roofline(s, stage_traffic(tr, s), tsrc) for s in e["stages"].values()]}
        # variants of the default workload that the design discusses, timed by the same run:
        # both stages side by side on two streams; the fused superframe kernel; the waterfall at the reference's line rate
        nst = max(20, args.steps // 3)
        ch, sf = WORKLOADS["full"][0], WORKLOADS["full"][1]
        for key, kw in (("full_concurrent", dict(concurrent=1)), ("full_fused", dict(fused=1))):
            e = measure(S, L, torch, rdv, rank, world, local_rank, "full", ch, sf, nst, 2, 0.5, **kw)
            b = ch * sf * 8192.0
            extra[key] = {"workload": WORKLOAD_TEXT["full"] + (", waterfall and audio stage side by side (--concurrent 1)" if "concurrent" in kw
                                                             else ", one fused kernel (--fused 1)"),
                          "value": e["value"], "unit": "rt_channels", "ms_per_step": e["ms_per_step"], "steps": nst,
                          "chain_GBps": b / e["ms_per_step"] / 1e6, "chain_frac": b / e["ms_per_step"] / 1e6 / HBM_PEAK_GBPS}
        ch, sf = WORKLOADS["wf"][0], WORKLOADS["wf"][1]
        e = measure(S, L, torch, rdv, rank, world, local_rank, "wf", ch, sf, nst, 2, 0.5, hop=512)
        tr, tsrc = pmc_traffic("wf", ch, sf, 512)
        extra["wf_hop512"] = {"workload": WORKLOAD_TEXT["wf"] + ", hop 512 (23.4 lines/s)", "value": e["value"], "unit": "rt_channels",
                              "ms_per_step": e["ms_per_step"], "steps": nst, "lines_per_s": e["stages"]["wf"]["lines_per_launch"] / e["ms_per_step"] * 1e3,
                              "rooflines": [roofline(s, stage_traffic(tr, s), tsrc) for s in e["stages"].values()]}
        out["extra"] = extra

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    rdv.close()


if __name__ == "__main__":
    main()
