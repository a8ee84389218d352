"""Work partitioning across the GPUs of one node: one process per GPU, independent
multiplications (or CRT primes) dealt to ranks, no data-path collective.  The only
communication is the barrier / max-reduction that brackets a timed region."""
from __future__ import annotations

import time


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of `total` independent units owned by `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def prime_owner(prime_index: int, world: int) -> int:
    """CRT modulus k lives on rank k mod world (the reference gives each modulus a thread, main_pthread.cc:413-422)."""
    return prime_index % world


def timed_region(fn, steps: int, sync, dist=None, device=None):
    """barrier + sync, run fn() `steps` times, sync + barrier; returns the MAX elapsed seconds over ranks."""
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt
