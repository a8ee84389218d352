"""world_size-2 run of the multi-GPU host logic on CPU (gloo): batch sharding with no
data-path collective, and the barrier + max-over-ranks timing that bench.py reports.
The per-rank compute is stood in for by the oracle (no GPU here)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ntt_amd.sharding import shard_range, timed_region
    from oracle import pyoracle as po
    O = po.Oracle()
    p, n, total = 469762049, 64, 10
    lo, hi = shard_range(total, rank, world)
    res = {}

    def step():
        for i in range(lo, hi):
            a, b = O.fill_random(n, 100 + i, p), O.fill_random(n, 200 + i, p)
            res[i] = O.poly_multiply_crt(a, b, p)

    dt = timed_region(step, 2, sync=lambda: None, dist=dist)
    # every rank must see the same (max) time
    t = torch.tensor([dt], dtype=torch.float64)
    lst = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(lst, t)
    assert all(float(x) == dt for x in lst)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), **{str(k): v for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_batch_sharding_two_ranks(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = {}
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        for k in z.files:
            assert int(k) not in got
            got[int(k)] = z[k]
    assert sorted(got) == list(range(10))
    p, n = 469762049, 64
    for i in range(10):
        a, b = oracle.fill_random(n, 100 + i, p), oracle.fill_random(n, 200 + i, p)
        assert np.array_equal(got[i], oracle.poly_multiply_crt(a, b, p))
