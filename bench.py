#!/usr/bin/env python3
"""bench.py -- SuperSDR hot path on MI355X: real-time IQ channels sustained (WF + demod).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full|wf|mixed|million|decim4]
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
  decim4          the decimating front end: 16384 ch/GPU, IQ at 48 kHz, 125-tap channel filters
The default run also times `wf`, `mixed`, `million`, `decim4` and the variants DESIGN.md discusses briefly after the
main measurement and reports them, each with its own rooflines, under "extra" in the same JSON line.
Every run ends with the parity hash of SURVEY.md 8e: each rank's checksums of a probe of its channel block, cross-checked
by its neighbour rank ("parity" in the JSON; a mismatch is an error).
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
    # configs[4]: 2^20 channels in total, 2^20 / N per GPU (strong scaling; 16 superframes per call on every rank: 64 GiB of input on
    # one GPU at N = 1, 8 GiB per GPU at N = 8 -- a wave sets a channel pair's state up once per call, SURVEY.md 8d ">= 16 superframes")
    "million": (1 << 20, 16,          1,     ("am",),               True, True),
    # the decimating front end (ssdr_set_decimation(4)): IQ at 48 kHz, 125-tap channel filters, waterfall lines from the wide stream
    "decim4": (16384,    8,           1,     ("usb", "lsb"),        True, True),
    # configs[2]'s shape with the AM passband narrowed to +-4 kHz (change_passband, utils_supersdr.py:1078-1092): the channel filter is a
    # real 25-tap FIR then and the NCO mixes -- every channel on the general path, which the default workload's full-band AM never times
    "am_narrow": (65536, 16,          1,     ("am",),               True, True),
    # A/B shapes of round 6's wave-specialised chain kernel (tools/ab_ws.sh): every channel a side-band listener (33-tap filters), and
    # three general-path channels in four
    "ssb":       (65536, 16,          1,     ("usb", "lsb"),        True, True),
    "mixed75":   (65536, 16,          1,     ("usb", "lsb", "am", "usb"), True, True),
}
WORKLOAD_DECIM = {"decim4": 4}
WORKLOAD_PARAMS = {"am_narrow": {"low_cut": -4000.0, "high_cut": 4000.0}}
WORKLOAD_TEXT = {"full": "65536 channels full chain (WF + AM demod + AGC), BASELINE configs[2]",
                 "wf": "4096 channels batched 1024-pt FFT + log-mag waterfall only, BASELINE configs[1]",
                 "mixed": "65536 channels mixed AM/USB/LSB/NBFM + 10x time binning, BASELINE configs[3]",
                 "million": "2^20 channels full chain in total, channel-sharded, BASELINE configs[4]",
                 "decim4": "16384 channels, IQ at 48 kHz (ssdr_set_decimation(4)): USB / LSB behind 125-tap decimating channel filters + waterfall",
                 "am_narrow": "65536 channels full chain, every channel AM with the passband narrowed to +-4 kHz (NCO + 25-tap channel FIR: the general audio path)",
                 "ssb": "65536 channels full chain, USB / LSB by channel (33-tap channel FIR: the general audio path)",
                 "mixed75": "65536 channels full chain, USB / LSB / full-band AM / USB by channel (three of four on the general audio path)"}
PATH_NAMES = ("FIR", "shift", "AM-shift")        # ssdr_audio_kernel<0|1|2>
PATH_TEXT = ("general: NCO -> FIR -> demodulator", "full-band lane shift: NCO, no FIR", "full-band AM: no NCO, no FIR (|x e^{j phi}| = |x|)")
# what the headline's audio stage does NOT time when every channel sits on the reference's default AM passband: extra.full_am_narrow does
PATH_SHORT = ("general (NCO -> channel FIR -> demodulator)", "full-band lane shift (NCO, no FIR)", "full-band AM (no NCO, no FIR)")
F32_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md: vector f32 peak (an FMA = 2 flop)
RIDGE_FLOP_PER_BYTE = F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBPS * 1e9)          # 19.7
# Executed vector work per unit: VALU wave-instructions from PMC SQ_INSTS_VALU / wave-units.  The per-launch counts are READ from the
# newest committed profiles/rNN_<workload>_pmc_summary.txt (tools/profile_round.sh) and divided by the units that workload's launch
# processes (VALU_SOURCES); the constants below are only the fall-back for a kernel no committed summary names (round 4's figures).
# FMA share: from the opcode mix of the loops (profiles/r04_isa_histograms.txt; the filters' multiply-adds -- 512 per frame for
# 33 taps, 2000 for 125 at D = 4 -- are dynamic).  lane-ops = 64 lanes x (instructions + FMA instructions): issue-slot work, see roofline().
KERNEL_VALU_FALLBACK = {
    "ssdr_wf_kernel<false, false>": ("line", 1173.5 / 2, 0.65),
    "ssdr_wf_kernel<true, false>": ("line", 1137.5 / 2, 0.60),
    "ssdr_wf_kernel<false, true>": ("line", 1163.9 / 2, 0.65),
    "ssdr_wf_kernel<true, true>": ("line", 1137.5 / 2, 0.60),
    "ssdr_audio_kernel<0>": ("frame", 832.6, 0.75),
    "ssdr_audio_kernel<1>": ("frame", 579.2, 0.42),
    "ssdr_audio_kernel<2>": ("frame", 241, 0.22),
    "ssdr_fused_am_kernel<false, false>": ("channel-superframe", 1074.0, 0.45),
    "ssdr_fused_am_kernel<true, false>": ("channel-superframe", 1658.0, 0.50),
    "ssdr_audio_dec_kernel<4>": ("frame", 2682, 0.85),
    "ssdr_chain_ws_kernel<false>": ("channel-superframe", 1796.0, 0.55),        # am_narrow, profiles/r06_am_narrow_pmc_summary.txt
    "ssdr_chain_ws_kernel<true>": ("channel-superframe", 1886.0, 0.59),         # configs[3] with --fused 3
}
# kernel -> (summary of which profiled workload, units of that kernel in one launch of it)
VALU_SOURCES = {
    "ssdr_wf_kernel<false, false>": ("wf", 4096 * 256),
    "ssdr_wf_kernel<true, false>": ("mixed_serial", 65536 * 10),
    "ssdr_wf_kernel<false, true>": ("wf_hop512", 4096 * 512),
    "ssdr_audio_kernel<0>": ("mixed_serial", 32768 * 20),
    "ssdr_audio_kernel<1>": ("mixed_serial", 16384 * 20),
    "ssdr_audio_kernel<2>": ("mixed_serial", 16384 * 20),
    "ssdr_fused_am_kernel<false, false>": ("full", 65536 * 16),
    "ssdr_fused_am_kernel<true, false>": ("full_hop512_fused", 65536 * 16),
    "ssdr_audio_dec_kernel<4>": ("decim4", 16384 * 16),
    "ssdr_chain_ws_kernel<false>": ("am_narrow", 65536 * 16),
    "ssdr_chain_ws_kernel<true>": ("mixed_chain_ws", 65536 * 10),
}


def load_kernel_valu():
    """-> (KERNEL_VALU, {kernel: the profiles/ file its instruction count was read from})"""
    import glob, re
    table, sources = dict(KERNEL_VALU_FALLBACK), {}
    for kernel, (wl, units) in VALU_SOURCES.items():
        best = None
        for path in glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc_summary.txt" % wl)):
            m = re.match(r"r(\d+)_", os.path.basename(path))
            if m and (best is None or int(m.group(1)) > best[0]):
                best = (int(m.group(1)), path)
        if best is None:
            continue
        stem = kernel[:kernel.index("<")]
        targs = kernel[kernel.index("<"):]
        for line in open(best[1]):
            # names are cut at 60 characters ("...ssdr_fused_am_kernel<false, fals"): the template arguments match as a prefix
            m = re.search(r"(ssdr_\w+_kernel)(<[^(>]*>?)?.*\bSQ_INSTS_VALU\s+n=\d+\s+mean=([0-9.e+]+)", line)
            if m and m.group(1) == stem and m.group(2) and targs.startswith(m.group(2).rstrip()):
                unit, _, fma = KERNEL_VALU_FALLBACK[kernel]
                table[kernel] = (unit, float(m.group(3)) / units, fma)
                sources[kernel] = os.path.basename(best[1])
                break
    table["ssdr_wf_kernel<true, true>"] = table["ssdr_wf_kernel<true, false>"]
    return table, sources


KERNEL_VALU, KERNEL_VALU_SOURCE = load_kernel_valu()

# What a kernel that does nothing but move bytes sustains on this chip for the read : write mix of each kernel (tools/ubench/hbm_stream.hip, bare
# stream kernels, streaming loads / stores, best grid; two boxes: profiles/r04_ubench_hbm_stream.txt) -- the denominator of
# roofline.frac_of_measured_stream.  The roofline fractions proper stay against the 8 TB/s of MI355X_MICROARCH.md.
STREAM_SOURCE = "profiles/r04_ubench_hbm_stream.txt (tools/ubench/hbm_stream.hip)"
MEASURED_STREAM_GBPS = {                      # kernel stem -> (mix, lowest, highest GB/s of the two boxes)
    "ssdr_fused_am_kernel": ("copy 1:1", 5290.0, 5670.0),            # 4096 B in, 2048 + 2048 B out per channel-superframe
    "ssdr_fused_exact_am_kernel": ("copy 1:1", 5290.0, 5670.0),
    "ssdr_chain_ws_kernel": ("copy 1:1", 5290.0, 5670.0),            # 4096 B in, 2048 / N + 2048 B out per channel-superframe
    "ssdr_wf_kernel": ("2 read : 1 write", 5340.0, 5510.0),          # 4096 in, 2048 out
    "ssdr_wf_exact_kernel": ("2 read : 1 write", 5340.0, 5510.0),
    "ssdr_audio_kernel": ("2 read : 1 write", 5340.0, 5510.0),       # 2048 in, 1024 out
    "ssdr_audio_dec_kernel": ("read only", 6940.0, 7030.0),          # 8192 in, 1024 out at D = 4: nearest probe
    "ssdr_db2col_kernel": ("1 read : 2 write", 5070.0, 5550.0),
    "ssdr_play_kernel": ("1 read : 8 write", 4450.0, 5280.0),
    "ssdr_play_rs_kernel": ("1 read : 8 write", 4450.0, 5280.0),
    "ssdr_iqwire_kernel": ("copy 1:1", 5290.0, 5670.0),
}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle timed on the host cores on a bounded sample.  Reported, never the target.
# ---------------------------------------------------------------------------------------------------------------
def host_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container that sees 256 logical CPUs
    behind `cpu.max = 1600000 100000` gets 16 CPUs' worth of time: 256 busy workers are throttled to that).
    -> (workers, dict with every number it was derived from)"""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota, src = None, None
    try:                                                             # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        src = "/sys/fs/cgroup/cpu.max = %s %s" % (q, per)
        if q != "max":
            quota = float(q) / float(per)
    except Exception:                                                # noqa: BLE001
        try:                                                         # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            src = "cpu.cfs_quota_us / cpu.cfs_period_us = %d / %d" % (q, per)
            if q > 0:
                quota = q / per
        except Exception:                                            # noqa: BLE001
            pass
    workers = aff if quota is None else max(1, min(aff, int(quota)))
    return workers, {"os_cpu_count": os.cpu_count(), "sched_affinity": aff, "cgroup_cpu_quota": quota, "cgroup_source": src, "workers": workers}


def steady_oracle(CB, workers, job, seconds):
    """`workers` processes, each on its own block of channels (oracle/cpu_bench.py:steady_worker): imports, IQ generation and a first
    pass happen before a barrier; behind it every worker runs whole blocks back to back for >= `seconds`.
    -> (channel-superframes per second over the slowest worker's elapsed time, wall of the slowest, per-worker rates)"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")                # never fork a process that holds a HIP context
    barrier, results = ctx.Barrier(workers), ctx.Queue()
    procs = [ctx.Process(target=CB.steady_worker, args=(w,) + job + (seconds, barrier, results), daemon=True) for w in range(workers)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(workers):
            got.append(results.get(timeout=600))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    errs = [g[3] for g in got if g[3]]
    if errs:
        raise RuntimeError("cpu_baseline worker failed: " + errs[0])
    wall = max(g[2] for g in got)
    return sum(g[1] for g in got) / wall, wall, sorted(g[1] / g[2] for g in got)


def cpu_baseline(workload, seconds=3.0):
    """the NumPy float64 oracle (the code that defines parity) on all the host cores this process may use
    and on one; next to it the oracle's fp32 C twin on as many threads.  Round 5: the worker count is the affinity mask capped by
    the cgroup quota (both in the line), the IQ is generated outside the timed region, every worker does >= `seconds` of back-to-back
    blocks behind a barrier, and `scaling_efficiency` = all-core rate / (workers x one-process rate) is reported -- under 0.5 the
    line says so, loudly."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import twinlib
    import ssdr_oracle as O
    import cpu_bench as CB                       # oracle/cpu_bench.py (test infrastructure, like the oracle itself)
    from concurrent.futures import ThreadPoolExecutor
    channels, sframes, n_avg, modes, do_wf, do_audio = WORKLOADS[workload]
    cores, host = host_cores()
    sf_np = min(sframes, 16)
    n_blk = 8                                    # channels per block: 8 x 16 superframes is ~40 ms of one core
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # one thread per worker process
        os.environ[v] = "1"
    job = (n_blk, sf_np, n_avg, modes, do_wf, do_audio)
    r1, w1, _ = steady_oracle(CB, 1, job, seconds)
    rN, wN, per = steady_oracle(CB, cores, job, seconds)
    one = {"value": r1 / RT_SUPERFRAMES_PER_S, "cores": 1, "sample": "blocks of %d ch x %d superframes back to back for %.1f s, one process" % (n_blk, sf_np, w1)}
    box = rN / RT_SUPERFRAMES_PER_S
    eff = box / (cores * one["value"])
    out = {"value": box, "unit": "rt_channels", "cores": cores, "kind": "port", "per_core": box / cores,
           "sample": "%d processes (one per usable core), each blocks of %d ch x %d superframes of its own channels back to back for %.1f s behind a "
                     "barrier, IQ generated before it; oracle/ssdr_oracle.py (NumPy float64)" % (cores, n_blk, sf_np, wN),
           "one_process": one, "scaling_efficiency": eff, "host": host,
           "worker_rates_rt_channels": {"min": per[0] / RT_SUPERFRAMES_PER_S, "median": per[len(per) // 2] / RT_SUPERFRAMES_PER_S, "max": per[-1] / RT_SUPERFRAMES_PER_S}}
    if eff < 0.5:
        out["scaling_note"] = ("LOW: %d workers give %.1fx one process (efficiency %.2f): shared memory bandwidth / SMT siblings / a quota "
                               "tighter than cpu.max says" % (cores, box / one["value"], eff))
        print("bench.py: cpu_baseline scaling efficiency %.2f < 0.5 on %d workers (%r)" % (eff, cores, host), file=sys.stderr, flush=True)
    budget_s = seconds

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
                      "agc_knee", "agc_delta8", "hang_frames", "fir_flags", "kfm"):
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
    nch_core = int(max(8, min(4096, budget_s / max(per_ch, 1e-9))))             # >= budget_s of work per thread, one thread per usable core
    block = make(nch_core)                        # same bytes for every worker
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, [block] * cores))
    wall = time.perf_counter() - t0
    # ---- SURVEY.md 8f on this host: the oracle's restatements of spectrum_db2col and play_buffer (NumPy / SciPy, what the reference runs),
    #      one core, next to extra.post
    rng = np.random.default_rng(3)
    lines = rng.integers(60, 200, (200, 1024)).astype(np.float32)
    t0 = time.perf_counter()
    for ln in lines:
        O.spectrum_db2col(ln, 0)
    db_us = (time.perf_counter() - t0) / len(lines) * 1e6
    frames = rng.integers(-20000, 20000, (400, 512)).astype(np.int16)
    pb = O.PlayBuffer()
    t0 = time.perf_counter()
    for fr in frames:
        pb(fr, volume=100, balance=0.0)
    pb_us = (time.perf_counter() - t0) / len(frames) * 1e6
    out["post"] = {"spectrum_db2col_us_per_line": round(db_us, 1), "play_buffer_us_per_frame": round(pb_us, 1), "cores": 1, "kind": "port",
                   "sample": "200 lines / 400 frames, oracle/ssdr_oracle.py restatements of utils_supersdr.py:787-813, 1106-1148"}
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
        if "SSDR_BENCH_DEVICE" not in os.environ:      # one host process per GPU, confined to it; the rank then
            mask = [d for d in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if d.strip()]      # sees exactly one device, index 0
            if mask and len(mask) < gpus:
                raise SystemExit("bench.py: --gpus %d but HIP_VISIBLE_DEVICES=%s permits %d" % (gpus, os.environ["HIP_VISIBLE_DEVICES"], len(mask)))
            env["HIP_VISIBLE_DEVICES"] = mask[r].strip() if mask else str(r)       # rank r -> the r-th PERMITTED device
            env["SSDR_BENCH_DEVICE"] = "0"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %r" % rcs)


# ---------------------------------------------------------------------------------------------------------------
# one measurement
# ---------------------------------------------------------------------------------------------------------------
class SetupFailed(RuntimeError):
    """raised on EVERY rank alike when one of them could not set a measurement up (measure())"""


# What the driver's single `--gpus N` command (N > 1) measures after the main `full` run, on every rank, by the same barrier / max-wall
# rule: BASELINE configs[4] (2^20 channels in total, 2^20 / N per GPU: the strong-scaling curve) and configs[3] (weak).
#                     key        workload   steps  scaling
MULTI_RANK_EXTRAS = (("million", "million", 5,     "strong"),
                     ("mixed",   "mixed",   60,    "weak"))


def multi_rank_extras(rdv, rank, world, run_one, probe=None):
    """Collective: every rank calls it.  run_one(workload, channels, sframes, first_channel_id, steps) -> measure()'s dict;
    probe(workload, first_channel_id, fused) -> the three checksums of a fresh probe of that block (None: no parity ring).
    -> {key: {value (all ranks' channel-superframes over the slowest rank's wall), per_rank, channels_per_gpu, scaling, parity ...}}"""
    from supersdr_amd.dist import channel_block
    out = {}
    for key, wl, nsteps, scaling in MULTI_RANK_EXTRAS:
        total, sf, n_avg = WORKLOADS[wl][0], WORKLOADS[wl][1], WORKLOADS[wl][2]
        first, ch = channel_block(rank, world, total) if scaling == "strong" else (rank * total, total)
        try:
            e = run_one(wl, ch, sf, first, nsteps)
        except SetupFailed as ex:
            out[key] = {"error": str(ex)[:300]}
            print("bench.py: extra.%s skipped on all ranks: %s" % (key, ex), file=sys.stderr, flush=True)
            continue
        per_rank = [v[0] for v in rdv.gather_floats([e["own_value"]])]
        chans = [int(v[0]) for v in rdv.gather_ints([ch])]
        firsts = [int(v[0]) for v in rdv.gather_ints([first])]
        b = ch * sf * (4096.0 + 2048.0 / n_avg + 2048.0)           # SURVEY.md 8d fused budget of THIS rank's block, over the step time
        out[key] = {"workload": WORKLOAD_TEXT[wl], "value": e["value"], "unit": "rt_channels", "ms_per_step": e["ms_per_step"], "steps": nsteps,
                    "scaling": scaling, "n_gpus": world, "channels_per_gpu": chans, "channels_total": sum(chans), "first_channel_ids": firsts,
                    "superframes_per_step": sf, "averaging_n": n_avg,
                    "chain_frac": b / e["ms_per_step"] / 1e6 / HBM_PEAK_GBPS,
                    "per_rank": {"values": per_rank, "value_min": min(per_rank), "value_max": max(per_rank),
                                 "note": "each rank's own channel-superframes / its own wall time; `value` uses the max wall over ranks"}}
        if "kernels" in e:
            out[key]["kernels"] = e["kernels"]
        if e.get("power"):                                           # (rank 0's GPU)
            out[key]["avg_watts"], out[key]["joules_per_step"] = e["power"]["avg_watts"], e["power"]["joules_per_step"]
        if probe is not None:
            own, cross = probe(wl, first, 1), probe(wl, firsts[(rank + 1) % world], 0)
            par = parity_report(rdv, world, firsts, own, cross, "as the main parity ring, on this workload's channel blocks")
            out[key]["parity"] = {"ranks_agree": par["ranks_agree"], "mismatching_ranks": par["mismatching_ranks"]}
    return out


def configure(S, eng, workload, channels, first_channel_id, hop=1024, fused=1, concurrent=0, exact=0, overlap=1):
    """channel parameters of `workload` for the block of channels that starts at global id `first_channel_id`
    (mode by channel id mod len(modes), tuning by the generator's carrier formula)"""
    _, _, n_avg, modes, _, _ = WORKLOADS[workload]
    decim = WORKLOAD_DECIM.get(workload, 1)
    if decim != 1:
        eng.set_decimation(decim)
    period = 97 * len(modes)                               # the parameter pattern repeats every len(modes) * 97 channels
    params = [S.default_params(modes[(first_channel_id + c) % len(modes)], f_shift_hz=(((first_channel_id + c) * 37) % 97 - 48) * 100.0,
                               **WORKLOAD_PARAMS.get(workload, {}))
              for c in range(min(channels, period))]
    for first in range(0, channels, len(params)):
        eng.set_params(first, params[: min(len(params), channels - first)])
    eng.reset_state()
    eng.set_hop(hop)
    eng.set_fused(fused)
    eng.set_overlap(overlap)
    eng.set_averaging(n_avg)
    eng.set_concurrent(concurrent)
    eng.set_exact_bins(exact)
    return n_avg, decim


def parity_probe(S, local_rank, workload, first_channel_id, channels=256, sframes=4, steps=2, fused=1):
    """SURVEY.md 8e "parity hash": a fresh ctx runs `steps` steps of `workload` on the `channels` channels that start at
    global id `first_channel_id` and returns the checksums of what it produced (ssdr_output_checksum: waterfall sums,
    PCM, RSSI).  Integer arithmetic over the result bytes: the same channel block gives the same three numbers on any
    rank, any GPU, any launch shape -- or the ranks do not compute the same thing."""
    _, _, n_avg, _, do_wf, do_audio = WORKLOADS[workload]
    sframes = max(sframes, n_avg)                         # (N-line binning: at least one complete group per step)
    with S.SsdrEngine(channels, device=local_rank) as eng:
        configure(S, eng, workload, channels, first_channel_id, fused=fused)
        eng.synth_iq(2 * sframes, seed=0x5D5D, first_channel_id=first_channel_id)
        for _ in range(steps):
            if do_wf and do_audio:
                eng.run_chain()
            elif do_wf:
                eng.run_wf(fetch=False)
            elif do_audio:
                eng.run_audio(fetch=False)
        return eng.output_checksum()


class EnergyProbe:
    """The board's energy accumulator (rocm_smi: rsmi_dev_energy_count_get, 15.3 uJ steps) around a timed region -> joules and average socket
    power of the GPU this rank runs on.  DESIGN.md section 5: every chain kernel sits at the board's power cap, so a step costs its joules;
    the line carries the evidence (roofline.power, extra.*.avg_watts).  None wherever the library or the counter is missing."""
    _lib = None

    def __init__(self, device_index):
        import ctypes as C
        self.C, self.dev, self.ok, self.cap = C, int(device_index), False, None
        try:
            if EnergyProbe._lib is None:
                lib = C.CDLL("librocm_smi64.so")
                if lib.rsmi_init(C.c_uint64(0)) != 0:
                    return
                EnergyProbe._lib = lib
            self.ok = self._read() is not None
            cap = C.c_uint64(0)
            if self.ok and EnergyProbe._lib.rsmi_dev_power_cap_get(C.c_uint32(self.dev), C.c_uint32(0), C.byref(cap)) == 0:
                self.cap = cap.value / 1e6                      # microwatts
        except (OSError, AttributeError):
            self.ok = False

    def _read(self):
        C = self.C
        e, res, ts = C.c_uint64(0), C.c_float(0.0), C.c_uint64(0)
        if EnergyProbe._lib.rsmi_dev_energy_count_get(C.c_uint32(self.dev), C.byref(e), C.byref(res), C.byref(ts)) != 0:
            return None
        return e.value * float(res.value) * 1e-6               # joules

    def start(self):
        self.e0 = self._read() if self.ok else None

    def stop(self, seconds, steps):
        e1 = self._read() if self.ok and self.e0 is not None else None
        if e1 is None or seconds <= 0 or e1 < self.e0:
            return None
        j = e1 - self.e0
        return {"avg_watts": round(j / seconds, 1), "joules_per_step": round(j / steps, 4), "cap_watts": self.cap,
                "source": "rocm_smi energy accumulator of this rank's GPU over the timed region"}


def measure(S, L, torch, rdv, rank, world, local_rank, workload, channels, sframes, steps, warmup, spinup,
            concurrent=0, host_feed=0, hop=1024, fused=1, exact=0, first_channel_id=None, overlap=1):
    """Spin the clocks up, W warm-up steps, then exactly `steps` timed steps between barrier + synchronize pairs.
    -> dict(value, ms_per_step, stages, ...)."""
    _, _, _, modes, do_wf, do_audio = WORKLOADS[workload]
    n_frames = 2 * sframes
    if first_channel_id is None:
        first_channel_id = rank * channels
    # set-up (context, buffers, synthetic input): a rank that cannot do it (memory) says so and ALL ranks leave together -- nobody waits at
    # the barrier below for a rank that never reaches it
    eng, err = None, None
    try:
        eng = S.SsdrEngine(channels, device=local_rank)
        n_avg, decim = configure(S, eng, workload, channels, first_channel_id, hop, fused, concurrent, exact, overlap)
        eng.synth_iq(n_frames, seed=0x5D5D, first_channel_id=first_channel_id)    # resident in HBM from here on
        eng.sync()
    except Exception as ex:                                # noqa: BLE001
        err = ex
    if not rdv.all_ok(err is None):
        if eng is not None:
            eng.close()
        raise SetupFailed("%s on %d channels x %d superframes: set-up failed on %s" % (
            workload, channels, sframes, ("this rank (%d): %s: %s" % (rank, type(err).__name__, str(err)[:160])) if err else "another rank"))

    def step():
        if do_wf:
            eng.run_wf(fetch=False)
        if do_audio:
            eng.run_audio(fetch=False)

    chain_kind = [0]                                      # which way ssdr_run_chain went: 0 two stages, 1 ssdr_fused_am_kernel, 2 ssdr_chain_ws_kernel
    if do_wf and do_audio:
        def step():                                       # noqa: F811  (ssdr_run_chain: a one-read kernel where the
            chain_kind[0] = eng.run_chain()[1]            #  configuration allows it and --fused is not 0, else the two kernels)

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
    probe = EnergyProbe(local_rank)
    rdv.barrier()
    torch.cuda.synchronize()
    probe.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    while inflight[0]:
        eng.feed_collect()
        inflight[0] -= 1
    eng.sync()
    torch.cuda.synchronize()
    own_wall = time.perf_counter() - t0                        # this rank's own time (reported per rank, never the value)
    power = probe.stop(own_wall, steps)
    rdv.barrier()
    wall = rdv.max_over_ranks(time.perf_counter() - t0)

    wf_ms, wf_n = eng.kernel_stats(L.K_WF)
    au_ms, au_n = eng.kernel_stats(L.K_AUDIO)
    fu_ms, fu_n = eng.kernel_stats(L.K_FUSED)
    paths = eng.audio_paths()
    eng.close()
    units = rdv.sum_over_ranks(channels) * sframes * steps     # channel-superframes, whole job (blocks may differ by one channel)
    # algorithmic bytes per launch: WF 4096 B in + 2048/N B out per line;
    # audio 2048 B in + 1024 B out per 512-sample frame
    stages = {}
    if wf_n:
        avg = wf_ms / wf_n
        # per line: hop 1024 reads 4096 B, hop 512 reads 2048 new bytes (the other half-line was the previous line's); 2048/N out
        lines = channels * sframes * decim * (2 if hop == 512 else 1)      # at D > 1 the lines come from the wide stream
        b = lines * ((2048.0 if hop == 512 else 4096.0) + 2048.0 / n_avg)
        stages["wf"] = {"kernel": "ssdr_wf_exact_kernel (float64)" if exact else
                                  "ssdr_wf_kernel<%s, %s>" % ("true" if n_avg > 1 else "false", "true" if hop == 512 else "false"),
                        "avg_ms": avg, "launches": wf_n, "bytes": b, "GBps": b / avg / 1e6, "lines_per_launch": lines, "units": lines}
    if au_n:
        avg = au_ms / au_n
        b = channels * n_frames * (2048.0 * decim + 1024.0)
        live = [p for p in range(3) if paths[p]]
        if decim > 1:
            name = "ssdr_audio_dec_kernel<%d>" % decim
        else:
            name = ("ssdr_audio_kernel<%d>" % live[0]) if len(live) == 1 else \
                   "audio stage: " + " + ".join("ssdr_audio_kernel<%d> (%s, %d ch)" % (p, PATH_NAMES[p], paths[p]) for p in live) + \
                   (" one after the other" if concurrent & 2 else " side by side")
        stages["audio"] = {"kernel": name, "avg_ms": avg, "launches": au_n, "bytes": b, "GBps": b / avg / 1e6,
                           "units_by_kernel": {"ssdr_audio_kernel<%d>" % p: paths[p] * n_frames for p in live} if decim == 1
                                              else {name: channels * n_frames}}
    if fu_n:
        avg = fu_ms / fu_n
        # SURVEY.md 8d, fused budget: 4096 in + 2048 / N per line + 2048 PCM out
        b = channels * sframes * (4096.0 + (2 if hop == 512 else 1) * 2048.0 / n_avg + 2048.0)
        stages["fused"] = {"kernel": "ssdr_fused_exact_am_kernel (float64 waterfall)" if exact else
                                     ("ssdr_chain_ws_kernel<%s>" % ("true" if n_avg > 1 else "false")) if chain_kind[0] == 2 else
                                     "ssdr_fused_am_kernel<%s, %s>" % ("true" if hop == 512 else "false", "true" if n_avg > 1 else "false"),
                           "avg_ms": avg, "launches": fu_n,
                           "bytes": b, "GBps": b / avg / 1e6, "units": channels * sframes}
    side = bool(do_wf and do_audio and "fused" not in stages and ((overlap and not exact) or concurrent & 1))
    for st in stages.values():
        st["side_by_side"] = side                       # the two stages ran beside each other: their durations overlap
    return {"value": units / wall / RT_SUPERFRAMES_PER_S, "ms_per_step": wall / steps * 1e3, "stages": stages,
            "n_avg": n_avg, "paths": paths, "decim": decim, "side_by_side": side, "chain_kind": chain_kind[0], "power": power,
            "own_value": channels * sframes * steps / own_wall / RT_SUPERFRAMES_PER_S}


# reference timings of the two functions the post kernels stand for (the reference's own code, survey
# container, one core, 2000 iterations) -- the one like-for-like comparison this path has
REFERENCE_US = {"ssdr_db2col_kernel": (141.0, "line", "kiwi_waterfall.spectrum_db2col, utils_supersdr.py:787-813"),
                "ssdr_play_kernel": (43.8, "frame", "kiwi_sound.play_buffer, utils_supersdr.py:1106-1148")}


def measure_post(S, L, local_rank, channels=65536, sframes=16, steps=10, spinup=0.3):
    """SURVEY.md 8f on the metric's shape: spectrum_db2col of the 16 lines and play_buffer (x4 branch, then the 64/27 branch of
    20.25 kHz KiwiSDRs) of the 32 frames a 65 536-channel chain run leaves on the device, and the IQ wire unpack of 8 frames
    per channel.  Kernel times from HIP events around each launch (ssdr_set_profiling); results stay on the device.
    Algorithmic bytes: db2col 2048 B in + 4096 B out per line; play_buffer 1024 B in + 8192 B out per frame (x4),
    1024 + 4852 (64/27); wire 2065 B in + 2048 B out per frame."""
    from supersdr_amd._lib import Db2colChan, PlayChan
    from supersdr_amd.workers import _fill_struct_array
    n_frames = 2 * sframes
    db, play = (Db2colChan * channels)(), (PlayChan * channels)()
    _fill_struct_array(db, Db2colChan(auto_scale=1, low_clip_db=-120.0, high_clip_db=-60.0, dynamic_range=40.0))
    _fill_struct_array(play, PlayChan(100.0, 0.0))
    stages = {}

    def timed(eng, which, call, name, units, nbytes):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < spinup:
            call()
        eng.kernel_stats(which, reset=True)
        for _ in range(steps):
            call()
        ms, n = eng.kernel_stats(which)
        avg = ms / max(n, 1)
        st = {"kernel": name, "avg_ms": avg, "launches": n, "bytes": float(nbytes), "GBps": nbytes / avg / 1e6, "units": units,
              "us_per_unit": avg * 1e3 / units}
        if name in REFERENCE_US:
            us, unit, what = REFERENCE_US[name]
            st["reference_cpu"] = {"us_per_" + unit: us, "what": what, "where": "BASELINE.md section 2 (survey container, 1 core)",
                                   "units_per_s_one_core": 1e6 / us, "units_per_s_gpu": units / avg * 1e3}
        return st

    for rate in (12000, 20250):
        eng = S.SsdrEngine(channels, device=local_rank)
        eng.set_kiwi_rate(rate)
        configure(S, eng, "full", channels, 0)
        eng.synth_iq(n_frames)
        eng.run_chain()
        eng.sync()
        eng.set_profiling(True)
        if rate == 12000:
            stages["db2col"] = timed(eng, L.K_DB2COL, lambda: eng.run_db2col(db, sframes, fetch=False), "ssdr_db2col_kernel",
                                     channels * sframes, channels * sframes * 6144.0)
            stages["play"] = timed(eng, L.K_PLAY, lambda: eng.run_playbuffer(play, fetch=False), "ssdr_play_kernel",
                                   channels * n_frames, channels * n_frames * (1024.0 + 8192.0))
            wire_frames = 8
            bodies = np.zeros((channels, wire_frames, 2065), np.uint8)
            bodies[:, :, 17::7] = 3
            stages["wire"] = timed(eng, L.K_WIRE, lambda: eng.push_iq_wire(bodies), "ssdr_iqwire_kernel",
                                   channels * wire_frames, channels * wire_frames * (2065.0 + 2048.0))
            del bodies
        else:
            stages["play_rs"] = timed(eng, L.K_PLAY, lambda: eng.run_playbuffer(play, fetch=False), "ssdr_play_rs_kernel",
                                      channels * n_frames, channels * n_frames * (1024.0 + 1213 * 4.0))
        eng.close()
    return stages


def gpu_numa_node(torch, local_rank):
    """NUMA node of the GPU this rank drives (sysfs, through its PCI address), or None"""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % addr).read())
        return node if node >= 0 else None
    except Exception:                                            # noqa: BLE001
        return None


def parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def bind_to_numa_node(node):
    """this process (and the threads it starts, and the pages it touches first: the hub's pinned ring) onto the CPUs of `node`.
    -> dict for the JSON line"""
    info = {"gpu_numa_node": node, "bound": False}
    if node is None or not hasattr(os, "sched_setaffinity"):
        return info
    try:
        cpus = parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()) & os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            info.update(bound=True, cpus=len(cpus))
    except Exception as ex:                                      # noqa: BLE001
        info["error"] = "%s: %s" % (type(ex).__name__, ex)
    return info


def measure_hub(S, L, torch, local_rank, channels, sframes, steps, in_place, batch_superframes=4, copy_threads=0, lazy_out=False, rdv=None):
    """The pipelined feed driven through the product's own ingest API (supersdr_amd/workers.py:IQHub, the thing
    KiwiSDRStream._process_iq_samples fills, kiwi/client.py:493-494): `channels` receivers, one block call per superframe.
    in_place=False: IQHub.feed_block (one copy of the block into the hub's pinned slot -- the hub's whole host cost);
    in_place=True: reserve / commit (the producer writes into the slot itself; here the slots keep the data they were
    filled with once, so what is timed is bookkeeping + H2D + kernels + D2H).  Results come back as whole-batch arrays;
    one channel has a listener attached.  PCIe-inclusive: never the headline value."""
    from supersdr_amd.workers import IQHub
    eng = S.SsdrEngine(channels, device=local_rank)
    configure(S, eng, "full", channels, 0)
    eng.synth_iq(2 * batch_superframes)
    block = eng.read_input()                                     # [channels, K * 1024, 2]: K superframes of synthetic IQ, replayed
    hub = IQHub(channels, engine=eng, gpu_post=False, pipeline=True, depth=3, lazy=True, batch_superframes=batch_superframes,
                backlog_superframes=2 * batch_superframes, stall_superframes=batch_superframes, copy_threads=copy_threads, lazy_out=lazy_out)
    hub.attach(channels // 2, wf=True, snd=True)
    seen, rows = [0], [0]
    hub.subscribe(lambda r: seen.__setitem__(0, seen[0] + r.pcm.shape[1] // 1024))
    hub.subscribe(lambda r: rows.__setitem__(0, r.pcm.shape[0]))
    U = 1024 * batch_superframes

    def one_batch():
        if in_place:
            v = hub.reserve(0, channels)
            hub.commit(0, channels, v.shape[1])
        else:
            hub.feed_block(0, block)

    for _ in range(len(hub._slots) + 2):                         # every slot filled once, the pipeline primed
        hub.feed_block(0, block)
    hub.flush()
    n_batches = max(1, steps * sframes // batch_superframes)
    seen[0] = 0
    if rdv is not None:
        rdv.barrier()                                            # every rank's feed primed: the timed region starts together
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_batches):
        one_batch()
    hub.flush()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if rdv is not None:
        rdv.barrier()
    wall_all = rdv.max_over_ranks(time.perf_counter() - t0) if rdv is not None else wall
    assert seen[0] == n_batches * batch_superframes, (seen, n_batches)
    q = hub.snd_queue[channels // 2].qsize()
    hub.close()
    units = channels * n_batches * batch_superframes
    # bytes per channel-superframe that cross PCIe: 4096 in; back either every channel's line + PCM + RSSI + flags, or the listeners' rows only
    d2h = (2048.0 + 2048.0 + 8.0 + 2.0) * (rows[0] / channels)
    return {"value": units / wall / RT_SUPERFRAMES_PER_S, "unit": "rt_channels", "ms_per_superframe": wall / (n_batches * batch_superframes) * 1e3,
            "channels": channels, "superframes_per_gpu_run": batch_superframes, "superframes": n_batches * batch_superframes,
            "ingest": "IQHub.reserve/commit (in place)" if in_place else "IQHub.feed_block (one host copy%s)" % (", %d threads" % copy_threads if copy_threads > 1 else ""),
            "results": "rows of the %d attached channel(s) only (SSDR_FEED_LAZY_OUT), the rest stays on the device" % rows[0] if lazy_out else "every channel's rows copied back",
            "h2d_bytes_per_channel_superframe": 4096.0, "d2h_bytes_per_channel_superframe": round(d2h, 3),
            "host_GBps": units * (4096.0 + d2h) / wall / 1e9, "frames_queued_for_the_one_listener": q,
            "units": units, "wall_s": wall, "wall_all_ranks_s": wall_all}


def csrc_sha256():
    """content hash of what libssdr.so is built from (supersdr_amd/csrc + include/ssdr.h): ties a PMC pass to the kernels it was taken on
    (the GPU box has no .git; tools/profile_round.sh stores this hash, and the commit it was told, in the traffic JSON)"""
    import glob, hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "supersdr_amd", "csrc", "*"))) + [os.path.join(ROOT, "include", "ssdr.h")]
    for f in files:
        if os.path.isfile(f) and not f.endswith((".o", ".so")):
            h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def measure_ui_hub(S, local_rank, receivers=2, superframes=60):
    """The reference's own scale (README.md:8 "dozens of instances"; one kiwi_waterfall + kiwi_sound pair per receiver): a synchronous IQHub of a
    few receivers with spectrum_db2col / play_buffer on the GPU and a bound worker pair on each -- what ONE superframe (85.3 ms of signal) costs
    from the ingest call to the workers' queues, i.e. the latency this path adds in front of the UI.  Not a throughput figure."""
    from supersdr_amd.workers import IQHub, bind_headless
    mod = bind_headless()

    class Disp:
        DISPLAY_WIDTH, WF_HEIGHT = 1024, 8

    hub = IQHub(receivers, device=local_rank)
    wfs = [mod.kiwi_waterfall("gpu", 0, "", 8, 7100.0, None, Disp(), hub=hub, channel=c, timeout=1.0) for c in range(receivers)]
    snds = [mod.kiwi_sound(7100.0, "USB", 30, 3000, "", w, 8) for w in wfs]
    rng = np.random.default_rng(1)
    iq = rng.integers(-8000, 8000, (receivers, 1024, 2)).astype(np.int16)
    for _ in range(5):
        hub.feed_block(0, iq)
    ts = []
    for _ in range(superframes):
        t0 = time.perf_counter()
        hub.feed_block(0, iq)                                    # synchronous hub: returns when the results are in the workers' queues
        for w, s_ in zip(wfs, snds):
            w.step()                                             # kiwi_waterfall.run's body for one line: receive_spectrum + spectrum_db2col + scroll
            s_.process_audio_stream(); s_.process_audio_stream()
        ts.append((time.perf_counter() - t0) * 1e3)
    hub.close()
    ts.sort()
    return {"receivers": receivers, "superframes": superframes, "ms_per_superframe_median": ts[len(ts) // 2], "ms_per_superframe_max": ts[-1],
            "real_time_ms": 1024 / 12.0, "what": "IQHub.feed_block -> waterfall + db2col + audio + play_buffer kernels -> bound kiwi_waterfall / kiwi_sound "
            "workers consume the line and both frames (synchronous hub, %d receivers)" % receivers}


def pmc_traffic(workload, channels, sframes, hop=1024):
    """HBM bytes per launch from the PMC passes committed under profiles/ (collected with rocprofv3 in separate
    runs, corrected as MI355X_MICROARCH.md prescribes; tools/profile_round.sh + tools/traffic_json.py).
    Only used when it was measured on exactly this workload shape; the newest round wins."""
    import glob, re
    hits = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json"))):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if (t.get("workload"), t.get("channels_per_gpu"), t.get("superframes_per_step"), t.get("wf_hop", 1024)) == (workload, channels, sframes, hop):
            m = re.match(r"r(\d+)_", os.path.basename(path))
            hits.append((int(m.group(1)) if m else 0, os.path.basename(path), t))
    found, src = {}, None
    newest = max((h[0] for h in hits), default=None)
    for rnd, name, t in hits:                       # the newest round's files (--fused 0 / 1 of one shape are two of them)
        if rnd == newest:
            found.update({k: v["hbm_bytes_per_launch"] for k, v in t["kernels"].items()})
            src = name if src is None else src + ", " + name
            TRAFFIC_BUILD[name] = (t.get("git_commit"), t.get("csrc_sha256"))
    return found, src


TRAFFIC_BUILD = {}          # traffic file -> (git commit, csrc hash) the PMC pass was taken at
_STALE_WARNED = set()


def stage_traffic(traffic, stage):
    """PMC bytes of a stage: its kernel's, or the sum over the kernels a multi-kernel audio stage names"""
    import re
    names = re.findall(r"ssdr_\w+_kernel(?:<[^>]*>)?", stage["kernel"])
    bare = {}                                      # a kernel named without its template arguments: the one instance that was profiled
    for k, v in traffic.items():
        bare.setdefault(k.split("<")[0], []).append(v)
    vals = [traffic.get(n, bare[n.split("<")[0]][0] if len(bare.get(n.split("<")[0], [])) == 1 else None) for n in names]
    return sum(vals) if vals and all(v is not None for v in vals) else None


def executed_flops(stage):
    """vector flops one launch executes (KERNEL_VALU), or None for a kernel that has no PMC record"""
    per = stage.get("units_by_kernel") or ({stage["kernel"]: stage["units"]} if "units" in stage else {})
    total = 0.0
    for name, units in per.items():
        if name not in KERNEL_VALU:
            return None
        _, instr, fma = KERNEL_VALU[name]
        total += units * instr * 64.0 * (1.0 + fma)
    return total or None


def roofline(stage, traffic=None, src=None):
    """The roof is chosen by the kernel's executed arithmetic intensity against the machine balance (157.3 Tflop/s over
    8 TB/s = 19.7 flop/B): below it the HBM roof applies, above it the vector-f32 roof.  `frac` is against that roof;
    `frac_hbm` (algorithmic bytes / time / 8 TB/s) is always given -- it is the figure north_star's target is stated in."""
    hbm = stage["GBps"] / HBM_PEAK_GBPS
    r = {"kernel": stage["kernel"], "bound": "hbm", "achieved": stage["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": hbm, "frac_hbm": hbm, "traffic": traffic, "avg_kernel_ms": stage["avg_ms"],
         "algorithmic_bytes_per_launch": stage["bytes"]}
    import re
    stems = set(re.findall(r"(ssdr_\w+_kernel)", stage["kernel"]))
    mixes = {MEASURED_STREAM_GBPS[k] for k in stems if k in MEASURED_STREAM_GBPS}
    if len(mixes) == 1 and len(stems & set(MEASURED_STREAM_GBPS)) == len(stems):
        mix, lo, hi = next(iter(mixes))
        # against the HIGHER of the two boxes' figures: 0.53-of-spec and 0.7x-of-a-copy side by side (review r5, item 7)
        r.update({"frac_of_measured_stream": stage["GBps"] / hi, "measured_stream_GBps": [lo, hi], "measured_stream_mix": mix,
                  "measured_stream_source": STREAM_SOURCE})
    fl = executed_flops(stage)
    if fl is not None:
        tf = fl / stage["avg_ms"] / 1e9
        # ISSUE-SLOT work, not floating-point flops: every executed VALU lane-instruction counts once (moves, converts and integer
        # ops included), an FMA twice -- the measure of how busy the vector ALU is against its f32 FMA peak
        r["issue"] = {"lane_ops_per_launch": fl, "lane_ops_per_algorithmic_byte": fl / stage["bytes"], "ridge_per_byte": RIDGE_FLOP_PER_BYTE,
                      "achieved_tera_ops": tf, "peak_tflops": F32_PEAK_TFLOPS, "frac_f32": tf / F32_PEAK_TFLOPS,
                      "source": "PMC SQ_INSTS_VALU per wave-unit x 64 lanes read from %s; FMA share from the opcode mix "
                                "(profiles/r04_isa_histograms.txt); not an observation of this run"
                                % (", ".join(sorted({"profiles/" + KERNEL_VALU_SOURCE[k] for k in (stage.get("units_by_kernel") or {stage["kernel"]: 0})
                                                     if k in KERNEL_VALU_SOURCE})) or "bench.py:KERNEL_VALU_FALLBACK (round 4)")}
        if fl / stage["bytes"] > RIDGE_FLOP_PER_BYTE:       # right of the ridge: the vector ALU's roof is the lower one
            r.update({"bound": "valu", "achieved": tf, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F32_PEAK_TFLOPS})
    if traffic is not None:
        # `traffic` is a constant of the committed PMC pass, NOT an observation of this run: say where it comes from, which build it
        # was taken on, and whether that build is this one
        r["traffic_source"] = "profiles/" + src
        builds = [TRAFFIC_BUILD.get(n.strip(), (None, None)) for n in src.split(",")]
        here = csrc_sha256()
        r["traffic_commit"] = builds[0][0]
        r["traffic_csrc_sha256"] = builds[0][1]
        r["traffic_matches_this_build"] = all(b[1] == here for b in builds)
        if not r["traffic_matches_this_build"] and src not in _STALE_WARNED:
            _STALE_WARNED.add(src)
            print("bench.py: WARNING: profiles/%s was taken on csrc %s (commit %s), this build is %s: roofline.traffic may be stale -- "
                  "re-run tools/profile_round.sh" % (src, builds[0][1], builds[0][0], here), file=sys.stderr, flush=True)
    return r


def parity_report(rdv, world, firsts, own, cross, what):
    """gather every rank's (own block, next rank's block) checksums; rank r's view of block r+1 must equal rank r+1's own"""
    allv = rdv.gather_ints(list(own) + list(cross))
    owns = [v[:3] for v in allv]
    crosses = [v[3:] for v in allv]
    bad = [r for r in range(world) if crosses[r] != owns[(r + 1) % world]]
    return {"ranks_agree": not bad, "mismatching_ranks": bad, "probe": what,
            "check": "rank r recomputes the probe of rank (r+1) mod N's channel block: must equal that rank's own checksums"
                     + (" (N = 1: a second fresh ctx must reproduce the first)" if world == 1 else ""),
            "first_channel_ids": firsts, "checksums": [["%016x" % x for x in o] for o in owns]}


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
                         "(ssdr_feed_*): the PCIe-inclusive rate of DESIGN.md, never the headline value; 3: the same through the product's ingest API, "
                         "IQHub.feed_block (one host copy per block); 4: IQHub.reserve / commit (in place)")
    ap.add_argument("--hub-lazy-out", type=int, default=0,
                    help="--host-feed 3 / 4: 1 = only the attached channels' results are copied back (IQHub(lazy_out=True), SSDR_FEED_LAZY_OUT)")
    ap.add_argument("--hub-copy-threads", type=int, default=0, help="--host-feed 3: threads of IQHub.feed_block's copy")
    ap.add_argument("--numa-bind", type=int, default=1,
                    help="--host-feed 3 / 4: bind each rank (and its pinned ring, by first touch) to the CPUs of its GPU's NUMA node (default on)")
    ap.add_argument("--concurrent", type=int, default=0, help="bit 0: audio stage on a second stream beside the waterfall kernel; bit 1: the audio stage's per-path kernels one after the other")
    ap.add_argument("--fused", type=int, default=1,
                    help="1 (the library's default): ssdr_run_chain uses the fused superframe kernel where the configuration allows it "
                         "(every channel full-band AM, N = 1, hop 1024); 0: always the two per-stage kernels")
    ap.add_argument("--overlap", type=int, default=1,
                    help="1 (the library's default): batches ssdr_run_chain does not fuse run the audio stage beside the waterfall kernel on a "
                         "second stream; 0: one after the other (per-stage durations that do not overlap)")
    ap.add_argument("--hop", type=int, default=1024, choices=[512, 1024],
                    help="samples between waterfall lines: 512 = 23.4 lines/s, the reference's waterfall rate (utils_supersdr.py:597)")
    ap.add_argument("--exact", type=int, default=0, help="1: ssdr_set_exact_bins -- the waterfall stage in float64 (bins equal the float64 oracle bit for bit)")
    ap.add_argument("--no-parity-probe", action="store_true",
                    help="profiling runs only: skip the probe launches of the parity hash (they would enter the per-kernel PMC means)")
    ap.add_argument("--rendezvous", default="gloo", choices=["gloo", "nccl"],
                    help="process group for the barrier around the timed region, the max-over-ranks of the wall time and the parity-hash "
                         "gather (no data moves between ranks): gloo (default), or nccl = RCCL (falls back to gloo, loudly, if it cannot come up)")
    ap.add_argument("--record", default="", help="also write the long form of the result (every roofline object, notes, sources) to this file")
    ap.add_argument("--verbose-line", action="store_true", help="print the long form as the JSON line (rounds 1-3's format)")
    ap.add_argument("--host-feed-extra", type=int, default=1, help="0: skip extra.hub_feed (the pipelined feed through IQHub; takes 8 GiB of pinned host memory)")
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

    first_id = channel_block(rank, world, WORKLOADS["million"][0])[0] if args.workload == "million" and not args.channels else rank * channels
    if args.dry_run:
        # control flow of the N-rank run without a GPU: rendezvous, channel blocks, and the parity-hash gather with stand-in
        # checksums (a pure function of the block's first channel id, as the real ones are of the block's results)
        rdv = Rendezvous("gloo", None)
        rdv.barrier()
        total = rdv.sum_over_ranks(channels)
        wall = rdv.max_over_ranks(1e-3 * (rank + 1))
        fake = lambda fid: [(fid * 2654435761 + k * 40503 + 12345) & 0xFFFFFFFFFFFFFFFF for k in range(3)]      # noqa: E731
        firsts = [int(v[0]) for v in rdv.gather_ints([first_id])]
        nxt = firsts[(rank + 1) % world]
        own, cross = fake(first_id), fake(nxt if os.environ.get("SSDR_DRYRUN_BREAK_RANK") != str(rank) else nxt + 1)
        parity = parity_report(rdv, world, firsts, own, cross, "dry run: stand-in checksums")
        if args.host_feed in (3, 4):                              # the multi-rank hub line's control flow: barrier, sum over ranks, max wall, per-rank values
            units = float(channels * sframes * args.steps)
            own_wall = 1e-3 * (rank + 1)
            line = hub_line(rdv, rank, world, args, {"value": units / own_wall / RT_SUPERFRAMES_PER_S, "units": units, "wall_s": own_wall,
                                                     "wall_all_ranks_s": rdv.max_over_ranks(own_wall), "host_GBps": 0.0, "channels": channels},
                            {"gpu_numa_node": None, "bound": False})
            if rank == 0:
                print(json.dumps(dict(line, dry_run=True)), flush=True)
            rdv.close()
            return
        dry_extra = None
        if world > 1 and args.workload == "full" and not args.no_extra:
            def fake_run(wl, ch, sf, first, nsteps):               # the arithmetic of measure(): all ranks' units over the slowest rank's wall
                own_wall = 1e-3 * (rank + 1)
                if os.environ.get("SSDR_DRYRUN_FAIL_SETUP") == "%s:%d" % (wl, rank):
                    ok = rdv.all_ok(False)
                else:
                    ok = rdv.all_ok(True)
                if not ok:
                    raise SetupFailed("%s: set-up failed on a rank (dry run)" % wl)
                units = rdv.sum_over_ranks(ch) * sf * nsteps
                w = rdv.max_over_ranks(own_wall)
                return {"value": units / w / RT_SUPERFRAMES_PER_S, "ms_per_step": w / nsteps * 1e3, "own_value": ch * sf * nsteps / own_wall / RT_SUPERFRAMES_PER_S}
            dry_extra = multi_rank_extras(rdv, rank, world, fake_run, lambda wl, fid, fused: fake(fid))
        if rank == 0:
            print(json.dumps({"metric": "real-time IQ channels sustained (WF+demod)", "value": None, "unit": "rt_channels",
                              **({"extra": dry_extra} if dry_extra is not None else {}),
                              "n_gpus": world, "dry_run": True, "channels_total": int(total), "max_wall": wall, "parity": parity,
                              "config": {"workload": WORKLOAD_TEXT[args.workload], "channels_per_gpu": channels,
                                         "first_channel_ids": firsts, "rendezvous": rdv.backend if world > 1 else "none"}}), flush=True)
        rdv.close()
        if not parity["ranks_agree"]:
            raise SystemExit(3)
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    if "SSDR_BENCH_DEVICE" in os.environ:          # test hook: several ranks on one GPU (exercises the N>1 control flow on a 1-GPU box)
        local_rank = int(os.environ["SSDR_BENCH_DEVICE"])
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible); --gpus must not exceed the GPUs of the node"
                         % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    # barrier + max-over-ranks + the parity-hash gather: a few scalars, no data path.  SURVEY.md 8e has the parent gather them;
    # under torch.distributed.run the ranks gather them themselves, over gloo (TCP on the loopback) unless --rendezvous nccl
    # asks for RCCL -- which this path has no use for and which has never had more than one device to come up on here
    rdv = Rendezvous(args.rendezvous, torch.device("cuda", local_rank) if args.rendezvous == "nccl" else None)

    import supersdr_amd as S
    from supersdr_amd import _lib as L

    if args.host_feed in (3, 4):         # the pipelined feed behind the product's own ingest API (IQHub); PCIe-inclusive, its own short line
        # every rank: its process, its copy threads and (by first touch) its pinned ring on the NUMA node of ITS GPU; a barrier on both sides of
        # the timed region; value = the channel-superframes of ALL ranks over the slowest rank's time
        numa = bind_to_numa_node(gpu_numa_node(torch, local_rank)) if args.numa_bind else {"gpu_numa_node": gpu_numa_node(torch, local_rank), "bound": False}
        h = measure_hub(S, L, torch, local_rank, channels, sframes, args.steps, in_place=(args.host_feed == 4),
                        copy_threads=args.hub_copy_threads, lazy_out=bool(args.hub_lazy_out), rdv=rdv)
        line = hub_line(rdv, rank, world, args, h, numa)
        if rank == 0:
            print(json.dumps(line), flush=True)
        rdv.close()
        return
    m = measure(S, L, torch, rdv, rank, world, local_rank, args.workload, channels, sframes, args.steps, args.warmup,
                args.spinup, args.concurrent, args.host_feed, args.hop, args.fused, args.exact, first_id, args.overlap)
    # SURVEY.md 8e parity hash (untimed): every rank hashes a probe of its own channel block and of the NEXT rank's block;
    # rank r's view of block r+1 must equal rank r+1's own (at N = 1: a second fresh ctx must reproduce the first)
    firsts = [int(v[0]) for v in rdv.gather_ints([first_id])]
    if args.no_parity_probe:
        parity = {"ranks_agree": True, "skipped": "--no-parity-probe"}
    else:
        own = parity_probe(S, local_rank, args.workload, first_id, fused=args.fused)
        cross = parity_probe(S, local_rank, args.workload, firsts[(rank + 1) % world], fused=0)
        parity = parity_report(rdv, world, firsts, own, cross,
                               "fresh ctx, 256 channels from the block's first id x 4 superframes x 2 steps, ssdr_output_checksum (wf, pcm, rssi); "
                               "own block through ssdr_run_chain as the timed steps run it (the fused kernel where it applies), "
                               "the neighbour's block through the two per-stage kernels")
    per_rank = [v[0] for v in rdv.gather_floats([m["own_value"]])]
    stages = m["stages"]
    dom = max(stages, key=lambda k: stages[k]["avg_ms"])
    traffic, src = pmc_traffic(args.workload, channels, sframes, args.hop)

    full = {
        "metric": "real-time IQ channels sustained (WF+demod)", "value": m["value"], "unit": "rt_channels",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
        "higher_is_better": True, "scaling": "strong" if args.workload == "million" else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOAD_TEXT[args.workload], "workload_key": args.workload,
                   "channels_per_gpu": channels, "superframes_per_step": sframes, "averaging_n": n_avg, "wf_hop": args.hop,
                   "audio_paths": {PATH_TEXT[p]: m["paths"][p] for p in range(3) if m["paths"][p]} if do_audio else {},
                   "audio_path": (" + ".join(PATH_SHORT[p] for p in range(3) if m["paths"][p]) if do_audio else "none"),
                   "input_decimation": m["decim"], "wf_exact_bins": bool(args.exact),
                   "chain": ("ssdr_run_chain: one wave-specialised kernel for both stages (audio waves hand raw frames to FFT waves through the LDS: "
                             "one read of the input; bit-identical to the two per-stage kernels)" if m.get("chain_kind") == 2 else
                             "ssdr_run_chain: one fused kernel for both stages (one read of the input; bit-identical to the two per-stage "
                             "kernels, which extra.full_two_kernels times)" if "fused" in m["stages"] else
                             "the per-stage kernels" + (" side by side on two streams" if m["side_by_side"] else " one after the other")),
                   "clock_spinup_s": args.spinup,
                   "input": {0: "resident in HBM", 1: "pinned host memory, pipelined H2D / kernels / D2H (PCIe-inclusive)",
                             2: "SND wire bodies in pinned host memory, pipelined, unpacked on the device (PCIe-inclusive)"}[args.host_feed],
                   "sharding": "channel blocks per GPU, no collectives",
                   "rendezvous": ("none" if world == 1 else rdv.backend if not rdv.fallback_reason else
                                  "gloo (RCCL FAILED: %s)" % rdv.fallback_reason)},
        "parity": parity,
        "build": {"csrc_sha256": csrc_sha256(), "git_commit": open(os.path.join(ROOT, ".ssdr_head")).read().strip()
                  if os.path.exists(os.path.join(ROOT, ".ssdr_head")) else None},
        "per_rank": {"value_min": min(per_rank), "value_max": max(per_rank), "values": per_rank,
                     "note": "each rank's own channel-superframes / its own wall time; `value` uses the max wall over ranks"},
        "roofline": roofline(stages[dom], stage_traffic(traffic, stages[dom]), src),
    }
    if m.get("power"):
        # what the chain is bound by: the board's power cap.  Socket power of rank 0's GPU over the timed region
        full["roofline"]["power"] = m["power"]
    # every measured kernel on every configuration, one short entry each: what the driver's record keeps of the line
    summary = {}

    def note(label, stage, tr=None, tsrc=None):
        r = roofline(stage, stage_traffic(tr, stage) if tr else None, tsrc)
        e = {"kernel": stage["kernel"], "ms": round(stage["avg_ms"], 4), "frac_hbm": round(r["frac_hbm"], 4)}
        if stage.get("side_by_side"):
            e["side_by_side"] = True            # ran beside the other stage: ms is its (longer) duration in company, frac_hbm is per that duration
        if "issue" in r:
            e["frac_f32_issue"] = round(r["issue"]["frac_f32"], 3)
        if r.get("traffic"):
            e["traffic_over_algorithmic"] = round(r["traffic"] / stage["bytes"], 3)
        if "reference_cpu" in stage:
            e["gpu_over_one_reference_core"] = round(stage["reference_cpu"]["units_per_s_gpu"] / stage["reference_cpu"]["units_per_s_one_core"])
        summary[label] = e
        return r

    for k, label in (("wf", "roofline_fft"), ("audio", "roofline_audio"), ("fused", "roofline_fused")):
        if k in stages:
            r = note("%s.%s" % (args.workload, k), stages[k], traffic, src)
            if k != dom:
                full[label] = r
    if "wf" in stages and "audio" in stages:
        # the chain as one unit (fused budget): the input counted once, 4096 + 2048/N + 2048 B per channel-superframe
        b = channels * sframes * (4096.0 + (2 if args.hop == 512 else 1) * 2048.0 / n_avg + 2048.0)
        ms = stages["wf"]["avg_ms"] + stages["audio"]["avg_ms"]
        full["roofline_chain"] = {"kernel": "waterfall + audio stage", "bound": "hbm", "achieved": b / ms / 1e6, "peak": HBM_PEAK_GBPS,
                                  "unit": "GB/s", "frac": b / ms / 1e6 / HBM_PEAK_GBPS, "traffic": None, "avg_kernel_ms": ms,
                                  "algorithmic_bytes_per_launch": b}

    if world > 1 and args.workload == "full" and not args.no_extra and not args.host_feed:
        # the driver's single N > 1 command: after the weak-scaling `full` figure (which stays `value`, so N = 1 agrees with BENCH), every rank
        # times configs[4] (its 2^20 / N block: the strong-scaling curve) and configs[3], same barrier / max-wall rule (review r5, item 1)
        def run_one(wl, ch, sf, first, nsteps):
            e = measure(S, L, torch, rdv, rank, world, local_rank, wl, ch, sf, nsteps, 2, 0.5, first_channel_id=first)
            e["kernels"] = {k: {"kernel": v["kernel"], "ms": round(v["avg_ms"], 4), "frac_hbm": round(v["GBps"] / HBM_PEAK_GBPS, 4)} for k, v in e["stages"].items()}
            return e
        full["extra"] = multi_rank_extras(rdv, rank, world, run_one,
                                          None if args.no_parity_probe else (lambda wl, fid, fused: parity_probe(S, local_rank, wl, fid, fused=fused)))
        for k, v in full["extra"].items():
            if "parity" in v and not v["parity"]["ranks_agree"]:
                parity = dict(parity, ranks_agree=False, extra_mismatch=k)
                full["parity"] = parity

    if rank == 0 and world == 1 and args.workload == "full" and not args.no_extra and not args.host_feed:
        extra = {}

        def chain_frac(e, ch, sf, navg, hop=1024):
            """the chain as one unit against the HBM roof: SURVEY.md 8d's fused budget over the step time"""
            b = ch * sf * (4096.0 + (2 if hop == 512 else 1) * 2048.0 / navg + 2048.0)
            return {"chain_GBps": b / e["ms_per_step"] / 1e6, "chain_frac": b / e["ms_per_step"] / 1e6 / HBM_PEAK_GBPS}

        def guarded(key, fn):
            """an extra that fails (memory, a box without 8 GiB to pin, ...) is reported, not fatal: the main measurement stands"""
            try:
                return fn()
            except Exception as ex:                                  # noqa: BLE001
                extra[key] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
                print("bench.py: extra.%s failed: %r" % (key, ex), file=sys.stderr, flush=True)
                return None

        def run_extra(key, wl, nsteps, text="", warm=2, spin=0.5, **kw):
            return guarded(key, lambda: run_extra_(key, wl, nsteps, text, warm, spin, **kw))

        def run_extra_(key, wl, nsteps, text="", warm=2, spin=0.5, **kw):
            ch, sf = WORKLOADS[wl][0], WORKLOADS[wl][1]
            e = measure(S, L, torch, rdv, rank, world, local_rank, wl, ch, sf, nsteps, warm, spin, **kw)
            tr, tsrc = pmc_traffic(wl, ch, sf, kw.get("hop", 1024))
            extra[key] = {"workload": WORKLOAD_TEXT[wl] + text, "value": e["value"], "unit": "rt_channels", "ms_per_step": e["ms_per_step"],
                          "steps": nsteps, "channels_per_gpu": ch, "superframes_per_step": sf, "averaging_n": e["n_avg"],
                          "input_decimation": e["decim"],
                          "audio_paths": {PATH_TEXT[p]: e["paths"][p] for p in range(3) if e["paths"][p]} if WORKLOADS[wl][5] else {},
                          "rooflines": [note("%s.%s" % (key, sk), sv, tr, tsrc) for sk, sv in e["stages"].items()]}
            if e.get("power"):
                extra[key]["avg_watts"] = e["power"]["avg_watts"]
                extra[key]["joules_per_step"] = e["power"]["joules_per_step"]
            if WORKLOADS[wl][4] and WORKLOADS[wl][5]:
                extra[key].update(chain_frac(e, ch, sf, e["n_avg"], kw.get("hop", 1024)))
            return e

        nst = max(20, args.steps // 3)
        # configs[1] and configs[3] in the same driver-timed line: shorter runs, each with its own rooflines
        run_extra("wf", "wf", max(20, args.steps // 2))
        run_extra("mixed", "mixed", max(60, args.steps // 2), ", the two stages side by side (ssdr_run_chain's default for what it does not fuse)", spin=1.0)
        run_extra("mixed_serial", "mixed", max(60, args.steps // 2), ", the two stages one after the other (--overlap 0): per-kernel durations and rooflines", spin=1.0, overlap=0)
        # variants of the default workload that the design discusses, timed by the same run
        run_extra("full_two_kernels", "full", nst, ", the two per-stage kernels one after the other (--fused 0 --overlap 0)", fused=0, overlap=0)
        run_extra("full_exact", "full", nst, ", float64 waterfall stage (--exact 1: bins equal the NumPy float64 path bit for bit), then the audio stage", exact=1)
        run_extra("wf_exact_bins", "wf", nst, ", float64 waterfall stage (--exact 1)", exact=1)
        e = run_extra("wf_hop512", "wf", nst, ", hop 512 (23.4 lines/s)", hop=512)
        if e is not None:
            extra["wf_hop512"]["lines_per_s"] = e["stages"]["wf"]["lines_per_launch"] / e["ms_per_step"] * 1e3
        run_extra("full_am_narrow", "am_narrow", nst, ": what an AM listener who narrows the passband gets -- every channel on the general audio path: "
                  "ssdr_run_chain's wave-specialised kernel (one read of the input)")
        run_extra("full_am_narrow_two_kernels", "am_narrow", nst, ", the general audio path beside the waterfall kernel (--fused 0): what round 5 ran by default", fused=0)
        run_extra("mixed_chain_ws", "mixed", max(60, args.steps // 2), ", the wave-specialised kernel (--fused 3: opt-in where full-band channels are among them)", spin=1.0, fused=3)
        run_extra("full_hop512", "full", nst, ", waterfall at hop 512 (23.4 lines/s, the reference's line rate), the two stages side by side", hop=512)
        # configs[4] at N = 1 (2^20 channels on this one GPU, 16 superframes per call) and the decimating front end
        run_extra("million", "million", 5, warm=1, spin=0.3)
        run_extra("decim4", "decim4", nst, warm=1, spin=0.3)
        # the reference's own post-processing on the GPU, next to the reference's own timings
        def post_extra():
            post = measure_post(S, L, local_rank)
            extra["post"] = {"workload": "spectrum_db2col / play_buffer (x4 and 64/27) / IQ wire unpack on the results of a 65536-channel x 16-superframe chain run",
                             "rooflines": [note("post." + k, v) for k, v in post.items()],
                             "reference_cpu": {v["kernel"]: v["reference_cpu"] for v in post.values() if "reference_cpu" in v}}
        guarded("post", post_extra)
        # the reference's own scale: a two-receiver UI hub, latency per superframe
        guarded("ui_hub", lambda: extra.__setitem__("ui_hub", measure_ui_hub(S, local_rank)))
        # the product's own ingest API in front of the pipelined feed (PCIe-inclusive, never `value`)
        if args.host_feed_extra:
            def hub_extra():
                # (6 "steps" = 24 GPU runs of 4 superframes each: the 8 runs of rounds 3-4 were mostly the copy pool's first touches)
                extra["hub_feed"] = {k: measure_hub(S, L, torch, local_rank, 65536, 16, 6, in_place=ip, copy_threads=ct, lazy_out=lo)
                                     for k, ip, ct, lo in (("feed_block", False, 0, False), ("feed_block_8_threads", False, 8, False), ("in_place", True, 0, False),
                                                           ("feed_block_8_threads_lazy_out", False, 8, True), ("in_place_lazy_out", True, 0, True))}
            guarded("hub_feed", hub_extra)
        full["extra"] = extra
    full["roofline"]["stages"] = summary

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                full["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as ex:                                      # noqa: BLE001  (a box that cannot start 256 workers still reports its GPU line)
                full["cpu_baseline"] = {"value": None, "unit": "rt_channels", "cores": os.cpu_count() or 1, "kind": "port",
                                        "sample": "FAILED: %s: %s" % (type(ex).__name__, str(ex)[:200])}
        if args.record:
            with open(args.record, "w") as f:
                json.dump(full, f, indent=1)
        print(json.dumps(full if args.verbose_line else compact_line(full)), flush=True)
    rdv.close()
    if not parity["ranks_agree"]:
        raise SystemExit("bench.py: PARITY HASH MISMATCH between ranks: %r" % (parity,))


def hub_line(rdv, rank, world, args, h, numa):
    """the --host-feed 3 / 4 line over all ranks: sum of the ranks' channel-superframes over the slowest rank's wall (barrier on both sides
    of the timed region), every rank's own rate and NUMA placement, the host-memory traffic of the whole job"""
    units_all = rdv.sum_over_ranks(h["units"])
    per_rank = [v[0] for v in rdv.gather_floats([h["value"]])]
    gbps = rdv.sum_over_ranks(h["host_GBps"])
    nodes = [int(v[0]) - 1 for v in rdv.gather_ints([(numa.get("gpu_numa_node") if numa.get("gpu_numa_node") is not None else -1) + 1])]
    bound = [bool(v[0]) for v in rdv.gather_ints([1 if numa.get("bound") else 0])]
    out = {k: v for k, v in h.items() if k not in ("units", "wall_s", "wall_all_ranks_s", "value", "host_GBps")}
    out.update(metric="real-time IQ channels sustained (WF+demod), inputs and results in host memory, through IQHub",
               value=units_all / h["wall_all_ranks_s"] / RT_SUPERFRAMES_PER_S, unit="rt_channels", n_gpus=world, steps=args.steps,
               per_rank={"values": per_rank, "value_min": min(per_rank), "value_max": max(per_rank),
                         "gpu_numa_node": [n if n >= 0 else None for n in nodes], "numa_bound": bound},
               host_GBps_all_ranks=gbps, wall_s=h["wall_all_ranks_s"], scaling="weak",
               note="PCIe- and host-copy-inclusive: never the headline value; value = all ranks' channel-superframes / the slowest rank's time")
    return out


def compact_line(full):
    """the ONE JSON line: the contract's fields, the dominant kernel's roofline with a one-entry-per-kernel summary of every
    configuration measured (roofline.stages), the CPU baseline, and the value / step time of every extra workload.  The long
    form (every roofline object, sources, notes) goes to --record FILE."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full[k] for k in keep}
    c = full["config"]
    out["config"] = {k: c[k] for k in ("workload", "workload_key", "channels_per_gpu", "superframes_per_step", "averaging_n", "wf_hop", "wf_exact_bins",
                                        "audio_path", "input", "sharding", "rendezvous")}
    out["config"]["chain"] = ("wave-specialised kernel (ssdr_run_chain)" if c["chain"].startswith("ssdr_run_chain: one wave-specialised") else
                              "fused kernel (ssdr_run_chain)" if c["chain"].startswith("ssdr_run_chain") else c["chain"])
    p = full["parity"]
    out["parity"] = {k: p[k] for k in ("ranks_agree", "checksums", "skipped") if k in p}
    out["per_rank"] = {k: full["per_rank"][k] for k in ("value_min", "value_max")}
    r = full["roofline"]
    out["roofline"] = {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_hbm", "frac_of_measured_stream", "measured_stream_GBps",
                                          "measured_stream_mix", "measured_stream_source", "traffic", "traffic_source",
                                          "traffic_commit", "traffic_csrc_sha256", "traffic_matches_this_build", "avg_kernel_ms",
                                          "algorithmic_bytes_per_launch", "stages", "power") if k in r}
    out["build"] = full.get("build")
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "scaling_efficiency", "scaling_note", "host") if k in cb}
        for k in ("c_twin", "one_process"):
            if k in cb:
                out["cpu_baseline"][k + "_value"] = cb[k]["value"]
        if "post" in cb:
            out["cpu_baseline"]["post"] = cb["post"]
    if "extra" in full and "value" in full["extra"].get("full_exact", {}):
        # the default's bins are the fp32 FFT's (one step from the float64 oracle in ~2e-5 of them: the guard band); the mode whose bins equal the
        # float64 definition bit for bit runs at this rate -- the two figures belong together (review r4, weak #2)
        out["value_with_float64_bins"] = full["extra"]["full_exact"]["value"]
        out["config"]["bins"] = "fp32 FFT (guard band vs the float64 oracle); bit-exact float64 bins: value_with_float64_bins"
    if "extra" in full:
        out["extra"] = {}
        for k, v in full["extra"].items():
            if "error" in v:
                out["extra"][k] = {"error": v["error"]}
            elif "value" in v:
                out["extra"][k] = {kk: (round(v[kk], 4) if isinstance(v[kk], float) else v[kk])
                                   for kk in ("value", "ms_per_step", "chain_frac", "avg_watts", "joules_per_step", "scaling", "n_gpus", "channels_total", "channels_per_gpu", "kernels", "parity") if kk in v}
                if "per_rank" in v:
                    out["extra"][k]["per_rank"] = {kk: v["per_rank"][kk] for kk in ("value_min", "value_max", "values")}
            elif k == "ui_hub":
                out["extra"][k] = {kk: (round(v[kk], 3) if isinstance(v[kk], float) else v[kk]) for kk in ("receivers", "ms_per_superframe_median", "ms_per_superframe_max", "real_time_ms")}
            elif k == "hub_feed":
                out["extra"][k] = {kk: {"value": vv["value"], "ms_per_superframe": round(vv["ms_per_superframe"], 3),
                                        "d2h_bytes_per_channel_superframe": vv.get("d2h_bytes_per_channel_superframe")} for kk, vv in v.items()}
    return out


if __name__ == "__main__":
    main()
