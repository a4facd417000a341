"""bench.py's rank handling, on the CPU: --gpus N starts N ranks (or insists that the launcher did), the ranks meet over
gloo, shard the channels without overlap, and rank 0 prints one JSON line.  (--dry-run: control flow only, no GPU work.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def last_json(p):
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout + p.stderr
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_that_meet_and_shard():
    p = run(["--gpus", "2", "--dry-run"])
    assert p.returncode == 0, p.stderr
    d = last_json(p)
    assert d["n_gpus"] == 2 and d["config"]["rendezvous"] == "gloo"
    assert d["channels_total"] == 2 * 65536                       # weak scaling: per-GPU work fixed
    assert abs(d["max_wall"] - 2e-3) < 1e-9                       # MAX over ranks, not rank 0's own


def check_multi_rank_extras(d, world):
    """round 6: the driver's ONE `--gpus N` command also returns BASELINE configs[4] (2^20 channels over the ranks: the strong-scaling
    curve) and configs[3], timed on every rank by the barrier / max-wall rule; `value` stays the weak-scaling `full` figure"""
    ex = d["extra"]
    assert set(ex) == {"million", "mixed"}
    m, x = ex["million"], ex["mixed"]
    assert m["scaling"] == "strong" and m["n_gpus"] == world and m["channels_total"] == 1 << 20
    assert m["channels_per_gpu"] == [(1 << 20) // world] * world and m["first_channel_ids"] == [r * ((1 << 20) // world) for r in range(world)]
    assert x["scaling"] == "weak" and x["channels_total"] == world * 65536 and x["channels_per_gpu"] == [65536] * world
    for e, units in ((m, (1 << 20) * 16 * 5), (x, world * 65536 * 10 * 60)):
        # all ranks' channel-superframes over the SLOWEST rank's wall (the dry run's rank r takes (r + 1) ms)
        assert abs(e["value"] - units / (world * 1e-3) / 11.71875) < 1e-6 * e["value"]
        assert len(e["per_rank"]["values"]) == world and e["per_rank"]["value_max"] == e["per_rank"]["values"][0]
        assert abs(e["per_rank"]["value_min"] * world - e["value"]) < 1e-6 * e["value"]       # the slowest rank's own rate x N = the job's
        assert e["parity"]["ranks_agree"] and e["parity"]["mismatching_ranks"] == []


def test_two_ranks_also_time_the_million_channel_and_mixed_configurations():
    p = run(["--gpus", "2", "--dry-run"])
    assert p.returncode == 0, p.stderr
    d = last_json(p)
    check_multi_rank_extras(d, 2)
    assert d["channels_total"] == 2 * 65536                       # the main figure is still the weak-scaling `full` one
    # --no-extra and the other workloads: the main measurement only
    assert "extra" not in last_json(run(["--gpus", "2", "--dry-run", "--no-extra"]))
    assert "extra" not in last_json(run(["--gpus", "2", "--dry-run", "--workload", "mixed"]))
    assert "extra" not in last_json(run(["--gpus", "1", "--dry-run"]))


def test_a_rank_that_cannot_set_an_extra_up_makes_all_ranks_skip_it_together():
    """one rank out of memory for configs[4] must not leave the others at a barrier: the set-up result is agreed (MIN all-reduce),
    the extra is reported as an error, the next one runs, the line is printed and the exit code is 0"""
    p = run(["--gpus", "3", "--dry-run"], {"SSDR_DRYRUN_FAIL_SETUP": "million:1"})
    assert p.returncode == 0, p.stderr
    d = last_json(p)
    assert "error" in d["extra"]["million"] and "set-up failed" in d["extra"]["million"]["error"]
    assert d["extra"]["mixed"]["channels_total"] == 3 * 65536 and d["extra"]["mixed"]["parity"]["ranks_agree"]


def test_million_workload_is_strong_scaling_over_the_ranks():
    p = run(["--gpus", "3", "--dry-run", "--workload", "million"])
    assert p.returncode == 0, p.stderr
    d = last_json(p)
    assert d["n_gpus"] == 3 and d["channels_total"] == 1 << 20    # blocks differ by at most one channel, none lost
    assert d["config"]["channels_per_gpu"] == (1 << 20) // 3 + 1


def test_gpus_must_match_the_launcher():
    p = run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
    p = run(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0"}, drop=())
    assert p.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in p.stderr


def test_without_a_gpu_every_rank_fails_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this box has a GPU")
    p = run(["--gpus", "2", "--no-cpu-baseline"])
    assert p.returncode != 0
    assert "needs a GPU" in p.stderr and "rank exit codes" in p.stderr


def test_parity_hash_gather_and_mismatch_is_an_error():
    """every rank's probe checksums are gathered, rank r's view of rank r+1's block is compared with that
    rank's own, and a disagreement fails the run (exit code != 0) after the JSON line was printed."""
    p = run(["--gpus", "3", "--dry-run"])
    assert p.returncode == 0, p.stderr
    d = last_json(p)
    par = d["parity"]
    assert par["ranks_agree"] and par["mismatching_ranks"] == [] and par["first_channel_ids"] == [0, 65536, 131072]
    assert len(par["checksums"]) == 3 and all(len(c) == 3 and all(len(x) == 16 for x in c) for c in par["checksums"])
    assert len({tuple(c) for c in par["checksums"]}) == 3          # different blocks hash differently
    p = run(["--gpus", "3", "--dry-run"], {"SSDR_DRYRUN_BREAK_RANK": "1"})
    assert p.returncode != 0
    d = last_json(p)
    assert not d["parity"]["ranks_agree"] and d["parity"]["mismatching_ranks"] == [1]
    p = run(["--gpus", "2", "--dry-run", "--workload", "million"])
    assert last_json(p)["parity"]["first_channel_ids"] == [0, 1 << 19]


def test_spawned_ranks_are_confined_to_their_gpu():
    """bench.py --gpus N without a launcher: rank r gets HIP_VISIBLE_DEVICES=r and addresses device 0"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    seen = []

    class P:
        def __init__(self, argv, env=None, stdout=None):
            seen.append(env)

        def wait(self):
            return 0

    real_popen, env0 = subprocess.Popen, dict(os.environ)
    subprocess.Popen = P                                   # (bench.py uses the module's attribute)
    os.environ.pop("SSDR_BENCH_DEVICE", None)
    try:
        b.spawn_ranks(3, ["--gpus", "3"])
    finally:
        subprocess.Popen = real_popen
        os.environ.clear()
        os.environ.update(env0)
    assert [e["HIP_VISIBLE_DEVICES"] for e in seen] == ["0", "1", "2"] and all(e["SSDR_BENCH_DEVICE"] == "0" for e in seen)
    assert [e["RANK"] for e in seen] == ["0", "1", "2"] and all(e["WORLD_SIZE"] == "3" for e in seen)
    # a mask the user set already is honoured: rank r takes the r-th PERMITTED device
    del seen[:]
    subprocess.Popen = P
    os.environ["HIP_VISIBLE_DEVICES"] = "4,5, 6,7"
    try:
        b.spawn_ranks(3, ["--gpus", "3"])
        import pytest
        with pytest.raises(SystemExit):
            b.spawn_ranks(5, ["--gpus", "5"])
    finally:
        subprocess.Popen = real_popen
        os.environ.clear()
        os.environ.update(env0)
    assert [e["HIP_VISIBLE_DEVICES"] for e in seen[:3]] == ["4", "5", "6"]


def test_eight_ranks_meet_over_gloo_shard_a_million_channels_and_close_the_parity_ring():
    """the driver's N = 8 shape on the CPU: eight ranks (started by the launcher the driver uses), the default rendezvous
    (gloo: no RCCL anywhere on this path), 2^20 channels in eight blocks of 131 072, the parity ring r -> r + 1 -> ... -> 0"""
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "8", "--dry-run", "--workload", "million"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = last_json(p)
    assert d["n_gpus"] == 8 and d["channels_total"] == 1 << 20 and d["config"]["channels_per_gpu"] == 131072
    assert d["config"]["rendezvous"] == "gloo" and d["parity"]["ranks_agree"]
    assert d["parity"]["first_channel_ids"] == [r * 131072 for r in range(8)] and len({tuple(c) for c in d["parity"]["checksums"]}) == 8
    assert abs(d["max_wall"] - 8e-3) < 1e-9


def test_eight_ranks_under_the_drivers_command_return_both_scaling_curves():
    """exactly the driver's N = 8 command line (default workload): the weak-scaling `full` figure plus extra.million / extra.mixed"""
    port = 31500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "8", "--steps", "20", "--warmup", "3", "--dry-run"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = last_json(p)
    assert d["n_gpus"] == 8 and d["channels_total"] == 8 * 65536 and d["parity"]["ranks_agree"]
    check_multi_rank_extras(d, 8)


def test_hub_feed_line_sums_over_the_ranks():
    """Round 5: bench.py --gpus N --host-feed 3|4 (the PCIe-inclusive rate through IQHub) is a multi-rank measurement: a barrier on both
    sides of the timed region, the ranks' channel-superframes summed, over the slowest rank's time -- not rank 0's own line."""
    p = run(["--gpus", "3", "--dry-run", "--host-feed", "4", "--channels", "1000", "--superframes", "4", "--steps", "5"])
    assert p.returncode == 0, p.stderr
    d = last_json(p)
    assert d["n_gpus"] == 3 and d["dry_run"] and d["scaling"] == "weak"
    units = 3 * 1000 * 4 * 5
    assert abs(d["value"] - units / 3e-3 / 11.71875) < 1e-6 * d["value"]          # all ranks' units / the slowest rank's wall (3 ms)
    assert len(d["per_rank"]["values"]) == 3 and d["per_rank"]["value_max"] == d["per_rank"]["values"][0]
    assert abs(d["per_rank"]["values"][2] - 1000 * 4 * 5 / 3e-3 / 11.71875) < 1e-3
    assert d["per_rank"]["numa_bound"] == [False] * 3 and d["per_rank"]["gpu_numa_node"] == [None] * 3


def test_numa_cpulist_parser():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and b.parse_cpulist("") == set()
    assert b.bind_to_numa_node(None) == {"gpu_numa_node": None, "bound": False}


def test_energy_probe_is_silent_where_there_is_no_counter():
    """bench.py's roofline.power comes from the board's energy accumulator (rocm_smi); on a host without a GPU, without the library or for a
    device index that does not exist the probe yields None -- it never raises and never holds the line up"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    for dev in (0, 63):
        p = bench.EnergyProbe(dev)
        p.start()
        r = p.stop(0.5, 10)
        assert r is None or (r["avg_watts"] >= 0 and r["joules_per_step"] >= 0)
    q = bench.EnergyProbe(0)
    q.start()
    assert q.stop(0.0, 10) is None                        # no time, no figure
