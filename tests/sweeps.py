"""Parity sweeps of the HIP path against the CPU oracle, grouped by the kernel family they drive.

Used two ways: in-process by tests/test_gpu_dispatch.py (the default dispatch), and as a program
(`python tests/sweeps.py <suite> [<suite> ...]`) in a fresh child process per measurement switch, because the
library reads its NTTX_* switches once per process.  Every case is bit-exact against the oracle (the restatement
of main.cc:185-209 / main.cc:286-338, pinned to the compiled reference).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P9, P4, P2 = 469762049, 998244353, 1224736769      # nine / four / two multiples of p below 2^32
PW = 104857601                                      # below 2^27: sixteen-multiple signed rows
P51 = 1337006139375617                              # id 4's modulus: 19 * 2^46 + 1


def _dev(x, dt):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x).view(dt)).cuda()


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def cyclic(nttx, O, p, log_len, batch, seed=1, check=None):
    """[batch][2^log_len] cyclic products against ntt_compute_omp (main.cc:185-209)."""
    import torch
    L = 1 << log_len
    a = O.fill_random(L * batch, seed, p).astype(np.uint32).reshape(batch, L)
    b = O.fill_random(L * batch, seed + 1000, p).astype(np.uint32).reshape(batch, L)
    a[0, :] = p - 1                                  # drives every lazy sum to its bound
    b[0, :] = p - 1
    da, db = _dev(a, np.int32), _dev(b, np.int32)
    dc = torch.empty_like(da)
    nttx.dev_cyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, _stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint32)
    for i in (range(batch) if check is None else check):
        assert np.array_equal(got[i], O.ntt_compute(a[i], b[i], p)), ("cyclic", p, log_len, batch, i)
    return a, b, got


def linear(nttx, O, p, n, batch, seed=1, check=None):
    """[batch] linear products of n coefficients on zero-padded device arrays (the padding is never read) against
    the oracle's cyclic product of the padded vectors."""
    import torch
    L = 1
    while L < 2 * n:
        L *= 2
    a = np.zeros((batch, L), np.uint32)
    b = np.zeros((batch, L), np.uint32)
    a[:, :n] = O.fill_random(n * batch, seed, p).reshape(batch, n)
    b[:, :n] = O.fill_random(n * batch, seed + 1000, p).reshape(batch, n)
    a[0, :n] = p - 1
    b[0, :n] = p - 1
    da, db = _dev(a, np.int32), _dev(b, np.int32)
    dc = torch.full_like(da, 7)
    nttx.dev_poly_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, _stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint32)
    for i in (range(batch) if check is None else check):
        assert np.array_equal(got[i], O.ntt_compute(a[i], b[i], p)), ("linear", p, n, batch, i)
    return a, b, got


def sums_hold(a, b, c, p):
    """c(1) = a(1) b(1) and c(-1) = a(-1) b(-1) modulo p for every product of the batch (cyclic and linear alike: the
    lengths are even), and canonical outputs."""
    assert int(c.max()) < p
    sg = np.where(np.arange(a.shape[1]) % 2 == 0, 1, -1).astype(np.int64)
    for s in (np.ones(a.shape[1], np.int64), sg):
        ea, eb, ec = ((x.astype(np.int64) * s).sum(1) % p for x in (a, b, c))
        want = np.array([int(x) * int(y) % p for x, y in zip(ea, eb)], dtype=np.int64)
        assert np.array_equal(ec, want)


def negacyclic(nttx, O, p, log_len, batch, seed=1):
    import torch
    L = 1 << log_len
    a = O.fill_random(L * batch, seed, p).astype(np.uint32).reshape(batch, L)
    b = O.fill_random(L * batch, seed + 1000, p).astype(np.uint32).reshape(batch, L)
    da, db = _dev(a, np.int32), _dev(b, np.int32)
    dc = torch.empty_like(da)
    nttx.dev_negacyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, _stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint32)
    for i in range(batch):
        pa, pb = np.zeros(2 * L, np.uint32), np.zeros(2 * L, np.uint32)
        pa[:L], pb[:L] = a[i], b[i]
        lin = O.ntt_compute(pa, pb, p).astype(np.int64)
        assert np.array_equal(got[i].astype(np.int64), (lin[:L] - lin[L:]) % p), ("negacyclic", p, log_len, i)


def crt(nttx, O, p, n, batch, seed=1, hi_bits=0):
    """The four-prime route on device arrays (main.cc:286-338); hi_bits > 0: coefficients up to 2^hi_bits, far above p."""
    import torch
    top = (1 << hi_bits) if hi_bits else p
    a = np.stack([O.fill_random(n, seed + i, top) for i in range(batch)])
    b = np.stack([O.fill_random(n, seed + 500 + i, top) for i in range(batch)])
    if not hi_bits:
        a[0, :] = p - 1
        b[0, :] = p - 1
    da, db = _dev(a, np.int64), _dev(b, np.int64)
    dc = torch.empty((batch, 2 * n - 1), dtype=torch.int64, device="cuda")
    nttx.dev_crt_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, _stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint64)
    for i in range(batch):
        assert np.array_equal(got[i], O.poly_multiply_crt(a[i], b[i], p)), ("crt", p, n, batch, i)


def direct64(nttx, O, p, n, batch, seed=1):
    """The 64-bit-lane route on device arrays against the four-prime oracle (n (p-1)^2 below its bound)."""
    import torch
    L = 1
    while L < 2 * n:
        L *= 2
    a = np.zeros((batch, L), np.uint64)
    b = np.zeros((batch, L), np.uint64)
    for i in range(batch):
        a[i, :n], b[i, :n] = O.fill_random(n, seed + i, p), O.fill_random(n, seed + 500 + i, p)
    da, db = _dev(a, np.int64), _dev(b, np.int64)
    dc = torch.empty_like(da)
    nttx.dev_direct_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, _stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint64)
    for i in range(batch):
        assert np.array_equal(got[i, :2 * n - 1], O.poly_multiply_crt(a[i, :n], b[i, :n], p)), ("direct64", p, n, i)


def negacyclic64(nttx, O, p, log_len, batch, seed=1):
    import torch
    L = 1 << log_len
    a = np.stack([O.fill_random(L, seed + i, p) for i in range(batch)])
    b = np.stack([O.fill_random(L, seed + 500 + i, p) for i in range(batch)])
    da, db = _dev(a, np.int64), _dev(b, np.int64)
    dc = torch.empty_like(da)
    nttx.dev_negacyclic_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, _stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint64)
    for i in range(batch):
        lin = np.zeros(2 * L, np.uint64)
        lin[:2 * L - 1] = O.poly_multiply_crt(a[i], b[i], p)
        want = (lin[:L].astype(np.int64) - lin[L:].astype(np.int64)) % p        # (both moduli are below 2^51)
        assert np.array_equal(got[i].astype(np.int64), want), ("negacyclic64", p, log_len, i)


# ---- suites -------------------------------------------------------------------------------------------------------

def suite_col(nttx, O):
    """Two-pass lengths 2^15 .. 2^18: every column-tile shape (narrow tiles at one and three products per call, full
    lines and the 64-column tiles of 2^6-point passes from 64 products on), forward with and without the zero half,
    ragged and exact n, the three unsigned classes and the small-modulus class behind them."""
    for p in (P9, P4, P2, PW):
        for k in (15, 16, 17, 18):
            for batch in (1, 3):
                cyclic(nttx, O, p, k, batch, seed=10 * k + batch)
                linear(nttx, O, p, (1 << (k - 1)) - (0 if batch == 1 else 777), batch, seed=20 * k + batch)
    for p in (P9, P4, P2, PW):
        for batch in (64, 100):
            cyclic(nttx, O, p, 15, batch, seed=batch)
            linear(nttx, O, p, 16384, batch, seed=batch + 1)       # exactly the lower half holds data
            linear(nttx, O, p, 12345, batch, seed=batch + 2)       # ragged: words from n on read as zero
            linear(nttx, O, p, 8193, batch, seed=batch + 3)
    for p in (P9, P4, P2):
        for k in (16, 17):
            cyclic(nttx, O, p, k, 64, seed=k)
            linear(nttx, O, p, (1 << (k - 1)) - 5, 64, seed=k + 1)


def suite_col_lite(nttx, O):
    """The shapes of suite_col on fewer moduli and with the oracle on a sample of the larger batches (the measurement
    switches get this one: each of them is a child process of its own)."""
    for p in (P9, P2, PW):
        for k in (15, 16, 17, 18):
            cyclic(nttx, O, p, k, 1, seed=10 * k)
            linear(nttx, O, p, (1 << (k - 1)) - 777, 3, seed=20 * k)
    sample = (0, 1, 31, 32, 62, 63)
    for p in (P9, P4, P2):
        cyclic(nttx, O, p, 15, 64, seed=64, check=sample)
        linear(nttx, O, p, 12345, 64, seed=66, check=sample)
    cyclic(nttx, O, P9, 16, 64, seed=16, check=sample)
    linear(nttx, O, P2, (1 << 16) - 5, 64, seed=17, check=sample)


def suite_col_big(nttx, O):
    """4096 products of 2^15 points per call: the oracle on a sample, the evaluations at 1 and -1 on every product."""
    sample = (0, 1, 63, 64, 1023, 1024, 2047, 3000, 4094, 4095)
    for maker, arg in ((cyclic, 15), (linear, 16000)):
        a, b, c = maker(nttx, O, P9, arg, 4096, seed=77, check=sample)
        sums_hold(a, b, c, P9)


def suite_rows(nttx, O):
    """Row kernels: tiled rows (2 .. 2^9 points, 2^10 / 2^11 single), the wave-per-row 512-point kernel alone and
    behind a column pass, long rows alone (2^12 .. 2^14) and behind a column pass (2^19 .. 2^23), negacyclic variants."""
    for p in (P9, P4, P2, PW, 7340033):
        for k, batch in ((1, 3), (2, 3), (3, 5), (5, 3), (8, 9), (9, 1), (9, 37), (10, 3), (11, 2), (12, 5), (13, 3), (14, 2)):
            cyclic(nttx, O, p, k, batch, seed=k)
        for k in (18, 19, 20):
            if (p - 1) % (1 << k) == 0:
                cyclic(nttx, O, p, k, 2, seed=k)
        for k in (4, 9, 11, 12, 13, 14, 16):
            if (p - 1) % (1 << (k + 1)) == 0:
                negacyclic(nttx, O, p, k, 2, seed=k)
    for p, k in ((P9, 21), (P9, 22), (P4, 22), (P2, 23), (PW, 21)):
        cyclic(nttx, O, p, k, 1, seed=k)
    linear(nttx, O, P9, (1 << 20) - 3, 1, seed=3)


def suite_crt(nttx, O):
    """The four-prime route: one product (one stream) and batches (primes side by side), two-pass and one-pass lengths,
    a modulus that needs all four primes, ones that need three, two or one, inputs far above p."""
    for n, batch in ((1000, 1), (5000, 3), (1 << 14, 1), (1 << 14, 8), (1 << 15, 7), (1 << 17, 1), (1 << 17, 6)):
        crt(nttx, O, P51, n, batch, seed=n + batch)
    for p in (1000000007, 65539, 97, (1 << 36) - 5, 281474976710597):
        crt(nttx, O, p, 1 << 14, 2, seed=p % 1000)
        crt(nttx, O, p, 70001, 6, seed=p % 1000 + 1)
    crt(nttx, O, 1000000007, 3000, 4, seed=5, hi_bits=45)


def suite_d64(nttx, O):
    """The 64-bit-lane route modulo id 4's prime (floating-point class by default) and a 48-bit prime: one-pass rows
    including the one-workgroup-per-row lengths, two-pass plans, negacyclic products."""
    for p in (P51, 263882790666241):
        for n, batch in ((100, 3), (1 << 9, 2), (1 << 10, 2), (1 << 11, 3), (1 << 12, 2), (5000, 2), (1 << 14, 2),
                         (1 << 16, 2), (1 << 17, 1), (1 << 17, 3), (1 << 19, 1)):
            if n * (p - 1) ** 2 < 469762049 * 998244353 * 1004535809 * 1224736769:     # the oracle's own exactness bound
                direct64(nttx, O, p, n, batch, seed=n % 997 + batch)
        for k in (6, 11, 12, 13, 15):
            negacyclic64(nttx, O, p, k, 2, seed=k)


SUITES = {"col": suite_col, "col_lite": suite_col_lite, "col_big": suite_col_big, "rows": suite_rows, "crt": suite_crt, "d64": suite_d64}


def main(argv):
    import ntt_amd
    from oracle import pyoracle
    O = pyoracle.Oracle()
    ntt_amd.init(0)
    try:
        for name in argv:
            SUITES[name](ntt_amd, O)
            print(f"sweep {name} ok", flush=True)
    finally:
        ntt_amd.shutdown()


if __name__ == "__main__":
    main(sys.argv[1:])
