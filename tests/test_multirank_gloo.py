"""The N>1 path on CPU: two processes, gloo backend.  The data path has no collective
(channels are independent), so what N>1 adds is: rendezvous, disjoint channel blocks with
globally unique channel ids, a barrier around the timed region and the max-over-ranks of
the wall time -- supersdr_amd/dist.py, exactly what bench.py uses on the GPU box with nccl."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from supersdr_amd.dist import Rendezvous, channel_block
    import ssdr_oracle as O
    rdv = Rendezvous("gloo")
    first, count = channel_block(rdv.rank, rdv.world, total)
    # each rank generates ITS channels with global ids and processes them with no communication
    iq = O.synth_iq(count, 1024, first_ch=first)
    lines = O.wf_line(iq.reshape(count, 1, 1024, 2)).astype(np.int64)
    checksum = int(lines.sum())
    rdv.barrier()
    wall = 0.010 * (rank + 1)                     # pretend rank 1 was slower
    slowest = rdv.max_over_ranks(wall)
    total_units = rdv.sum_over_ranks(count)
    total_checksum = rdv.sum_over_ranks(checksum)
    rdv.barrier()
    q.put((rank, first, count, checksum, slowest, total_units, total_checksum))
    rdv.close()


def test_two_ranks_gloo_shard_and_timing():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ssdr_oracle as O
    total, world, port = 6, 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, f0, n0, c0, s0, u0, t0), (r1, f1, n1, c1, s1, u1, t1) = res
    assert (f0, n0, f1, n1) == (0, 3, 3, 3)
    assert s0 == s1 == pytest.approx(0.020)       # max over ranks, seen by both
    assert u0 == u1 == total
    # sharded result == single-process result on all channels (no exchange needed)
    iq = O.synth_iq(total, 1024)
    want = int(O.wf_line(iq.reshape(total, 1, 1024, 2)).astype(np.int64).sum())
    assert c0 + c1 == want and t0 == t1 == want
