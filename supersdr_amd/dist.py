"""Multi-GPU harness: channel-block sharding and the timing rendezvous.

The path has no exchange step (channels are independent), so there is NO
data-path collective: torch.distributed is used only for the barrier around the timed region,
the max-over-ranks of the wall time and the parity-hash gather -- a few scalars.  The default
backend is "gloo" (TCP on the loopback, what SURVEY.md 8e's "the parent gathers over pipes" amounts
to under torch.distributed.run); "nccl" (= RCCL) is opt-in (bench.py --rendezvous nccl) and falls
back to gloo, loudly, if it cannot come up.  One process per GPU, launched by torch.distributed.run.
"""
import os


def env_rank():
    """(rank, local_rank, world) from the torchrun environment (single process if absent)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def channel_block(rank, world, total_channels):
    """Contiguous block [first, first+count) of `total_channels` owned by `rank`; blocks differ
    by at most one channel and cover every channel exactly once."""
    base, extra = divmod(int(total_channels), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


import contextlib


@contextlib.contextmanager
def _stdout_to_stderr():
    """file descriptor 1 -> 2 for the duration (native libraries that print to stdout)"""
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class Rendezvous:
    """barrier() and max_over_ranks() for the bench; a no-op world of one needs no process group."""

    def __init__(self, backend="gloo", device=None):
        self.rank, self.local_rank, self.world = env_rank()
        self.backend = backend
        self.device = device
        self._dist = None
        self.fallback_reason = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts
            try:
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                with _stdout_to_stderr():             # (gloo announces its peers on stdout: the bench's stdout is ONE JSON line)
                    dist.init_process_group(backend=backend, **kw)
                    dist.barrier()                    # forces communicator / connection creation now, not inside the timed region
            except Exception as e:
                # The rendezvous carries no data (a barrier and a few scalars): if RCCL cannot come up, gloo does the
                # same job over TCP and the measurement is unaffected -- but it is said, loudly, and it is in the JSON.
                if backend != "nccl":
                    raise
                import sys
                self.fallback_reason = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
                print("supersdr_amd.dist: RCCL rendezvous FAILED on rank %d (%s); falling back to gloo over TCP for the "
                      "barrier / max-over-ranks (no data path uses it)" % (self.rank, self.fallback_reason), file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                self.backend = backend = "gloo"
                with _stdout_to_stderr():
                    dist.init_process_group(backend="gloo")
                    dist.barrier()
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max_over_ranks(self, value):
        if self._dist is None:
            return float(value)
        import torch
        dev = self.device if (self.backend == "nccl" and self.device is not None) else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok):
        """True when `ok` is true on EVERY rank (a MIN all-reduce: also a rendezvous).  What lets the ranks skip an optional
        measurement together when one of them could not set it up, instead of the others waiting at a barrier it never reaches."""
        if self._dist is None:
            return bool(ok)
        import torch
        dev = self.device if (self.backend == "nccl" and self.device is not None) else "cpu"
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def sum_over_ranks(self, value):
        if self._dist is None:
            return float(value)
        import torch
        dev = self.device if (self.backend == "nccl" and self.device is not None) else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def gather_ints(self, values):
        """every rank's list of (up to 64-bit, unsigned) integers -> list over ranks of lists (on every rank)"""
        values = [int(v) for v in values]
        if self._dist is None:
            return [values]
        import torch
        dev = self.device if (self.backend == "nccl" and self.device is not None) else "cpu"
        # as two 32-bit halves in int64: no signed overflow on any backend
        t = torch.tensor([[v >> 32, v & 0xFFFFFFFF] for v in values], dtype=torch.int64, device=dev).reshape(-1)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(outs, t)
        return [[(int(o[2 * i]) << 32) | int(o[2 * i + 1]) for i in range(len(values))] for o in outs]

    def gather_floats(self, values):
        if self._dist is None:
            return [[float(v) for v in values]]
        import torch
        dev = self.device if (self.backend == "nccl" and self.device is not None) else "cpu"
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(outs, t)
        return [[float(x) for x in o] for o in outs]

    def close(self):
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None
