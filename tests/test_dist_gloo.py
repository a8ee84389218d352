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


# ---- exchange-step workloads: primes dealt to ranks, and one product split by columns ----------
def _crt_case(oracle, batch, n, p, seed):
    a = np.stack([oracle.fill_random(n, seed + i, p) for i in range(batch)]).astype(np.uint64)
    b = np.stack([oracle.fill_random(n, seed + 500 + i, p) for i in range(batch)]).astype(np.uint64)
    want = np.stack([oracle.poly_multiply_crt(a[i], b[i], p) for i in range(batch)])
    return a, b, want


def _run_crt(comm, O, a, b, n, p, gather):
    from ntt_amd.distributed import CrtSharded
    from tests.cpu_engine import OracleEngine
    sh = CrtSharded(comm, OracleEngine(O))
    lo, hi = sh.batch_range(a.shape[0])
    ta = torch.from_numpy(a[lo:hi].view(np.int64).copy())
    tb = torch.from_numpy(b[lo:hi].view(np.int64).copy())
    out = sh.multiply(ta, tb, n, p, gather=gather)
    return lo, hi, sh.pr, sh.gp, out.numpy().view(np.uint64).copy()


def _fourstep_case(oracle, log_len, p, linear):
    L = 1 << log_len
    a, b = oracle.fill_random(L, 11, p).astype(np.uint32), oracle.fill_random(L, 12, p).astype(np.uint32)
    if linear:
        a[L // 2:] = 0
        b[L // 2:] = 0
    return a, b, oracle.ntt_compute(a, b, p)


def _run_fourstep(comm, O, a, b, log_len, log_t, p, linear):
    from ntt_amd.distributed import FourStepSharded
    from tests.cpu_engine import OracleEngine
    fs = FourStepSharded(comm, OracleEngine(O), log_len, p, log_t=log_t)
    xa = fs.slab_of(torch.from_numpy(a.view(np.int32).copy()))
    xb = fs.slab_of(torch.from_numpy(b.view(np.int32).copy()))
    xc = fs.multiply(xa, xb, zero_hi=linear)
    return fs.col0, fs.cols, xc.numpy().view(np.uint32).copy()


def _exchange_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ntt_amd.distributed import TorchComm
    from oracle import pyoracle as po
    O = po.Oracle()
    comm = TorchComm()
    p64 = 1337006139375617
    a, b, _ = _crt_case(O, 3, 50, p64, 40)
    lo, hi, pr, gp, out = _run_crt(comm, O, a, b, 50, p64, gather=True)
    lo2, hi2, pr2, gp2, part = _run_crt(comm, O, a, b, 50, p64, gather=False)
    fa, fb, _ = _fourstep_case(O, 12, 469762049, True)
    col0, cols, slab = _run_fourstep(comm, O, fa, fb, 12, 6, 469762049, True)
    np.savez(os.path.join(outdir, f"x{rank}.npz"), lo=lo, hi=hi, pr=pr, gp=gp, out=out, part=part, col0=col0, cols=cols, slab=slab)
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_steps_two_ranks_gloo(tmp_path, oracle):
    world = 2
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    p64, n = 1337006139375617, 50
    a, b, want = _crt_case(oracle, 3, n, p64, 40)
    fa, fb, fwant = _fourstep_case(oracle, 12, 469762049, True)
    full = np.zeros(1 << 12, dtype=np.uint32)
    for r in range(world):
        z = np.load(tmp_path / f"x{r}.npz")
        assert int(z["gp"]) == 2 and int(z["lo"]) == 0 and int(z["hi"]) == 3      # 2 primes per rank, whole batch
        assert np.array_equal(z["out"], want)
        sl = 128 // 2
        q = int(z["pr"])
        padded = np.zeros((3, 128), dtype=np.uint64)
        padded[:, : 2 * n - 1] = want
        assert np.array_equal(z["part"][:, : min(sl, 2 * n - 1 - q * sl)], padded[:, q * sl: (q + 1) * sl][:, : min(sl, 2 * n - 1 - q * sl)])
        full.reshape(64, 64)[:, int(z["col0"]): int(z["col0"]) + int(z["cols"])] = z["slab"]
    assert np.array_equal(full, fwant)


def test_exchange_steps_in_process_ranks(oracle):
    """4 and 8 ranks as threads: primes x batch layout of the CRT path, 4-way column split."""
    from ntt_amd.distributed import ThreadComm, crt_layout
    assert crt_layout(1) == (1, 1) and crt_layout(2) == (2, 1) and crt_layout(4) == (4, 1)
    assert crt_layout(8) == (4, 2) and crt_layout(3) == (1, 3) and crt_layout(6) == (2, 3)
    p64, n = 1337006139375617, 33
    a, b, want = _crt_case(oracle, 5, n, p64, 70)
    for world in (1, 3, 4, 8):
        outs = ThreadComm.run(world, lambda comm: _run_crt(comm, oracle, a, b, n, p64, True))
        seen = np.zeros(5, dtype=int)
        for lo, hi, pr, gp, out in outs:
            assert np.array_equal(out, want[lo:hi])
            if pr == 0:
                seen[lo:hi] += 1
        assert (seen == 1).all()
    # beyond the exactness bound n*(p-1)^2 < M the sharded product refuses like the single-GPU entry point (NTTX_ERANGE)
    import pytest
    from ntt_amd import NttxError
    big = (1 << 61) - 1
    with pytest.raises(NttxError) as err:
        ThreadComm.run(1, lambda comm: _run_crt(comm, oracle, a % np.uint64(big), b % np.uint64(big), n, big, True))
    assert err.value.code == -2
    for p, log_len, log_t, world, linear in ((998244353, 12, 6, 4, False), (1224736769, 13, 7, 2, True), (469762049, 12, 6, 1, True),
                                             (469762049, 11, 6, 2, False)):     # the last one: rows too short for the pieces-layout product
        fa, fb, fwant = _fourstep_case(oracle, log_len, p, linear)
        outs = ThreadComm.run(world, lambda comm: _run_fourstep(comm, oracle, fa, fb, log_len, log_t, p, linear))
        full = np.zeros(1 << log_len, dtype=np.uint32)
        for col0, cols, slab in outs:
            full.reshape(1 << log_t, -1)[:, col0: col0 + cols] = slab
        assert np.array_equal(full, fwant)
