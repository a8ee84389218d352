#!/usr/bin/env python3
"""Randomised parity run: random entry point, modulus, length and batch against the CPU oracle for a given number of
seconds (GPU box).  Every case is bit-exact or the run stops with the case's parameters.
usage: python tools/fuzz_gpu.py [seconds] [seed] [big]      (big: lengths drawn towards the long end)"""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ntt_amd  # noqa: E402
from oracle import pyoracle  # noqa: E402

P32 = [469762049, 998244353, 1004535809, 1224736769, 104857601, 7340033, 167772161, 754974721, 2013265921, 595591169,
       1811939329, 113246209, 377487361, 2130706433]
P64 = [1337006139375617, 263882790666241, 2251756864012289, 1125899437080577, 8588886017, 4611685941117976577,
       2305843009146585089, 1350000109944833, 1349999656960001, 1099469684737]
PCRT = [1000000007, 65539, 97, (1 << 36) - 5, 281474976710597, 2147483647, 1337006139375617, 2, 6, 1 << 20, (1 << 40) + 15]
M4 = 469762049 * 998244353 * 1004535809 * 1224736769


def root_of(p):
    return next(g for g in (3, 2, 5, 6, 7, 10, 11, 13, 14, 15, 17, 19) if pow(g, (p - 1) // 2, p) == p - 1)


def v2(x):
    k = 0
    while x % 2 == 0:
        x //= 2
        k += 1
    return k


def dev(x, dt):
    return torch.from_numpy(np.ascontiguousarray(x).view(dt)).cuda()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = random.Random(seed)
    big = len(sys.argv) > 3
    draw = (lambda lo, hi: max(rng.randint(lo, hi), rng.randint(lo, hi), rng.randint(lo, hi))) if big else rng.randint
    O = pyoracle.Oracle()
    ntt_amd.init(0)
    st = torch.cuda.current_stream().cuda_stream
    t_end = time.time() + seconds
    counts = {}
    case = None
    t_print = time.time() + 20
    try:
        while time.time() < t_end:
            kind = rng.choice(["linear32", "cyclic32", "nega32", "host32", "crt", "crt_host", "direct64", "nega64", "hostu64"])
            s1, s2 = rng.randrange(1 << 30), rng.randrange(1 << 30)
            if kind in ("linear32", "cyclic32", "nega32", "host32"):
                p = rng.choice(P32)
                kmax = min(v2(p - 1) - (1 if kind == "nega32" else 0), 20)
                k = draw(1, kmax)
                L = 1 << k
                batch = rng.randint(1, max(1, min(70, (1 << 21) // L)))
                if kind == "linear32" or kind == "host32":
                    n = rng.randint(L // 4 + 1, L // 2) if L >= 4 else 1
                else:
                    n = L
                case = (kind, p, k, n, batch, s1, s2)
                a = np.zeros((batch, L), np.uint32)
                b = np.zeros((batch, L), np.uint32)
                a[:, :n] = O.fill_random(n * batch, s1, p).reshape(batch, n)
                b[:, :n] = O.fill_random(n * batch, s2, p).reshape(batch, n)
                r = rng.randrange(batch)
                a[r, :n] = p - 1
                b[r, : max(1, n // 2)] = p - 1
                if kind == "host32":
                    got = ntt_amd.poly_multiply_batch(a[:, :n].copy(), b[:, :n].copy(), p)
                    for i in range(batch):
                        assert np.array_equal(got[i], O.ntt_compute(a[i], b[i], p, root_of(p))[: 2 * n - 1]), case
                else:
                    da, db = dev(a, np.int32), dev(b, np.int32)
                    dc = torch.full_like(da, 5)
                    if kind == "linear32":
                        ntt_amd.dev_poly_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, st)
                    elif kind == "cyclic32":
                        ntt_amd.dev_cyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, st)
                    else:
                        ntt_amd.dev_negacyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, st)
                    torch.cuda.synchronize()
                    got = dc.cpu().numpy().view(np.uint32)
                    for i in range(batch):
                        if kind == "nega32":
                            pa, pb = np.zeros(2 * L, np.uint32), np.zeros(2 * L, np.uint32)
                            pa[:L], pb[:L] = a[i], b[i]
                            lin = O.ntt_compute(pa, pb, p, root_of(p)).astype(np.int64)
                            assert np.array_equal(got[i].astype(np.int64), (lin[:L] - lin[L:]) % p), case
                        else:
                            assert np.array_equal(got[i], O.ntt_compute(a[i], b[i], p, root_of(p))), case
            elif kind in ("crt", "crt_host"):
                p = rng.choice(PCRT)
                n = rng.randint(1, rng.choice([50, 3000, 40000, 200000]) if not big else rng.choice([40000, 200000, 524288]))
                while n * (p - 1) ** 2 >= M4:
                    n //= 2
                n = max(n, 1)
                batch = 1 if kind == "crt_host" else rng.randint(1, max(1, min(40, (1 << 19) // n)))
                top = p if rng.random() < 0.7 else (1 << rng.randint(40, 63))
                case = (kind, p, n, batch, top, s1, s2)
                a = np.stack([O.fill_random(n, s1 + i, top) for i in range(batch)])
                b = np.stack([O.fill_random(n, s2 + i, top) for i in range(batch)])
                if kind == "crt_host":
                    got = ntt_amd.poly_multiply_crt(a[0], b[0], n, p)[None, :]
                else:
                    da, db = dev(a, np.int64), dev(b, np.int64)
                    dc = torch.empty((batch, 2 * n - 1), dtype=torch.int64, device="cuda")
                    ntt_amd.dev_crt_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, st)
                    torch.cuda.synchronize()
                    got = dc.cpu().numpy().view(np.uint64)
                for i in range(batch):
                    ra, rb = a[i] % np.uint64(p), b[i] % np.uint64(p)
                    assert np.array_equal(got[i], O.poly_multiply_crt(ra, rb, p)), case
            else:
                p = rng.choice(P64)
                kmax = min(v2(p - 1) - (1 if kind == "nega64" else 0), 18)
                k = draw(1, kmax)
                L = 1 << k
                n = L if kind == "nega64" else (rng.randint(L // 4 + 1, L // 2) if L >= 4 else 1)
                batch = 1 if kind == "hostu64" else rng.randint(1, max(1, min(16, (1 << 18) // L)))
                exact = n * (p - 1) ** 2 < M4
                if not exact and n > 3000:
                    continue                                  # (no cheap exact CPU product there)
                case = (kind, p, k, n, batch, s1, s2)
                a = np.zeros((batch, L), np.uint64)
                b = np.zeros((batch, L), np.uint64)
                for i in range(batch):
                    a[i, :n], b[i, :n] = O.fill_random(n, s1 + i, p), O.fill_random(n, s2 + i, p)
                want = [O.poly_multiply_crt(a[i, :n], b[i, :n], p) if exact else O.poly_multiply_naive_u64(a[i, :n], b[i, :n], p)
                        for i in range(batch)]
                if kind == "hostu64":
                    assert np.array_equal(ntt_amd.poly_multiply(a[0, :n], b[0, :n], n, p, dtype=np.uint64), want[0]), case
                else:
                    da, db = dev(a, np.int64), dev(b, np.int64)
                    dc = torch.empty_like(da)
                    if kind == "direct64":
                        ntt_amd.dev_direct_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, st)
                    else:
                        ntt_amd.dev_negacyclic_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, st)
                    torch.cuda.synchronize()
                    got = dc.cpu().numpy().view(np.uint64)
                    for i in range(batch):
                        if kind == "direct64":
                            assert np.array_equal(got[i, : 2 * n - 1], want[i]), case
                        else:
                            lin = np.concatenate([want[i], np.zeros(1, np.uint64)])
                            lo, hi = lin[:L], lin[L:]
                            assert np.array_equal(got[i], np.where(lo >= hi, lo - hi, lo + np.uint64(p) - hi)), case
            counts[kind] = counts.get(kind, 0) + 1
            if time.time() > t_print:
                print(f"  ... {sum(counts.values())} cases", flush=True)
                t_print = time.time() + 20
        print(f"fuzz ok: seed {seed}, {sum(counts.values())} cases in {seconds:.0f} s: {counts}", flush=True)
    except BaseException:
        print(f"FUZZ FAILURE at case {case} (seed {seed}) after {counts}", flush=True)
        raise
    finally:
        ntt_amd.shutdown()


if __name__ == "__main__":
    main()
