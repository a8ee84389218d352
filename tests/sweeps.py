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


def _root(p):
    """The reference's root 3 (main.cc:241) where it is a quadratic non-residue modulo p, else the smallest non-residue:
    the result of a product does not depend on which generator of the roots of unity the transforms use."""
    return next(g for g in (3, 2, 5, 6, 7, 10, 11, 13, 14, 15, 17, 19) if pow(g, (p - 1) // 2, p) == p - 1)


def _worst(x, p):
    """Row 0 drives every lazy sum to its bound: all p - 1 where the batch has other rows, else p - 1 on the first third
    (a one-row batch keeps random words, so that its transform is not that of a constant)."""
    if x.shape[0] > 1:
        x[0, :] = p - 1
    else:
        x[0, : max(1, x.shape[1] // 3)] = p - 1


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
    _worst(a, p)
    _worst(b, p)
    da, db = _dev(a, np.int32), _dev(b, np.int32)
    dc = torch.empty_like(da)
    nttx.dev_cyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, _stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint32)
    for i in (range(batch) if check is None else check):
        assert np.array_equal(got[i], O.ntt_compute(a[i], b[i], p, _root(p))), ("cyclic", p, log_len, batch, i)
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
        assert np.array_equal(got[i], O.ntt_compute(a[i], b[i], p, _root(p))), ("linear", p, n, batch, i)
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
        lin = O.ntt_compute(pa, pb, p, _root(p)).astype(np.int64)
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
        cyclic(nttx, O, p, 15, 16, seed=3)                          # between the narrow tiles and the 64-column ones
        linear(nttx, O, p, (1 << 17) - 9, 8, seed=4)                # full-line 2^9-point tiles
        cyclic(nttx, O, p, 18, 8, seed=5, check=(0, 3, 7))


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
        for k in (4, 9, 10, 11, 12, 13, 14, 16):
            if (p - 1) % (1 << (k + 1)) == 0:
                negacyclic(nttx, O, p, k, 2, seed=k)
    for p, k in ((P9, 21), (P9, 22), (P4, 22), (P2, 23), (PW, 21)):
        cyclic(nttx, O, p, k, 1, seed=k)
    linear(nttx, O, P9, (1 << 20) - 3, 1, seed=3)


def suite_crt(nttx, O):
    """The four-prime route: one product (one stream) and batches (primes side by side), two-pass and one-pass lengths,
    a modulus that needs all four primes, ones that need three, two or one, inputs far above p."""
    for n, batch in ((1000, 1), (5000, 3), (1 << 14, 1), (1 << 14, 8), (1 << 15, 1), (1 << 15, 7), (1 << 16, 2), (1 << 16, 6),
                     (1 << 17, 1), (1 << 17, 6), (12345, 64), (1 << 14, 64), (200000, 1), (1 << 19, 1)):
        crt(nttx, O, P51 if n <= (1 << 17) else 263882790666241, n, batch, seed=n + batch)
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


# ---- every column-pass shape through the slab entry point ------------------------------------------------------------

def fourstep_one_rank(nttx, O, p, log_len, log_t, linear, seed=1):
    """One product of 2^log_len points as nttx_dev_colpass_u32 on a and b (2^log_t-point columns over ALL 2^(log_len-log_t)
    columns as one slab), row products, inverse column pass: the single-rank form of the column-split product, against
    ntt_compute_omp (main.cc:185-209).  The slab width picks the column kernel: 16 columns the half-line tiles, 8192 and
    more the split-table variants; linear = the zero-half variants."""
    import torch
    from ntt_amd.distributed import FourStepSharded, HipEngine, ThreadComm
    L = 1 << log_len
    a = O.fill_random(L, seed, p).astype(np.uint32)
    b = O.fill_random(L, seed + 1, p).astype(np.uint32)
    a[:7] = p - 1
    b[:7] = p - 1
    if linear:
        a[L // 2:] = 0
        b[L // 2:] = 0
    comm = ThreadComm.run(1, lambda c: c)[0]
    fs = FourStepSharded(comm, HipEngine(), log_len, p, log_t=log_t)
    xa, xb = fs.slab_of(_dev(a, np.int32)).clone(), fs.slab_of(_dev(b, np.int32)).clone()
    xc = fs.multiply(xa, xb, zero_hi=linear)
    torch.cuda.synchronize()
    got = xc.cpu().numpy().view(np.uint32).reshape(L)
    assert np.array_equal(got, O.ntt_compute(a, b, p, _root(p))), ("fourstep", p, log_len, log_t, linear)


P4B, P2B = 754974721, 2013265921      # 45 * 2^24 + 1 (four multiples below 2^32), 15 * 2^27 + 1 (two): the longest of their classes


def suite_colpass(nttx, O):
    """Slabs of 16 columns (half-line tiles), 32 and 512 (full-line tiles, narrow tiles where the launch is small) and
    8192 (split factor tables), 2^6 .. 2^9-point columns, with and without the zero half, forward and inverse, the
    three unsigned classes."""
    for p in (P9, P4, P2):
        for log_t in (6, 7, 8, 9):
            for log_s in (4, 5, 9, 13):
                if log_s == 13 and log_t == 6:
                    continue                              # (no split-table kernel for 2^6-point columns: the plain table serves)
                for linear in (False, True):
                    fourstep_one_rank(nttx, O, p, log_t + log_s, log_t, linear, seed=log_t * 16 + log_s)
        fourstep_one_rank(nttx, O, p, 6 + 13, 6, True, seed=5)


def pieces(nttx, O, p, log_len, npieces, rows, seed=1):
    """Row products on [pieces][rows][len / pieces] arrays (nttx_dev_cyclic_multiply_pieces_u32) against ntt_compute_omp."""
    import torch
    L = 1 << log_len
    a = O.fill_random(L * rows, seed, p).astype(np.uint32).reshape(rows, L)
    b = O.fill_random(L * rows, seed + 9, p).astype(np.uint32).reshape(rows, L)
    part = L // npieces
    lay = lambda x: np.ascontiguousarray(x.reshape(rows, npieces, part).transpose(1, 0, 2))      # noqa: E731
    da, db = _dev(lay(a), np.int32), _dev(lay(b), np.int32)
    dc = torch.empty_like(da)
    nttx.dev_cyclic_multiply_pieces_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, npieces, rows, p, _stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint32).reshape(npieces, rows, part).transpose(1, 0, 2).reshape(rows, L)
    for i in range(rows):
        assert np.array_equal(got[i], O.ntt_compute(a[i], b[i], p, _root(p))), ("pieces", p, log_len, npieces, i)


def sliced(nttx, O, n, batch, slices, seed=1):
    """Residues modulo all four CRT primes in the [slice][prime][product][len / slices] send-buffer layout
    (nttx_dev_crt_residues_sliced_u64) against the oracle's per-prime products (main.cc:292-311)."""
    import torch
    L = 1
    while L < 2 * n:
        L *= 2
    a = np.stack([O.fill_random(n, seed + i, 1 << 62) for i in range(batch)])
    b = np.stack([O.fill_random(n, seed + 50 + i, 1 << 62) for i in range(batch)])
    da, db = _dev(a, np.int64), _dev(b, np.int64)
    res = torch.empty((slices, 4, batch, L // slices), dtype=torch.int32, device="cuda")
    nttx.dev_crt_residues_sliced_u64(da.data_ptr(), db.data_ptr(), res.data_ptr(), n, 15, batch, slices, _stream())
    torch.cuda.synchronize()
    got = res.cpu().numpy().view(np.uint32)
    for k in range(4):
        m = nttx.crt_prime(k)
        for i in range(batch):
            pa, pb = np.zeros(L, np.uint32), np.zeros(L, np.uint32)
            pa[:n], pb[:n] = (a[i] % np.uint64(m)).astype(np.uint32), (b[i] % np.uint64(m)).astype(np.uint32)
            want = O.ntt_compute(pa, pb, m, _root(m))
            assert np.array_equal(got[:, k, i, :].reshape(L), want), ("sliced", n, slices, k, i)


def suite_layouts(nttx, O):
    """The exchange layouts on every modulus class: pieces-layout row products (2^15 .. 2^17), sliced residues
    (2^15 .. 2^20), negacyclic products on two-pass lengths (twisted column passes of 2^6 .. 2^9 points and the
    twisted row kernels below them), three-pass products with and without the zero half."""
    for p in (P9, P4, P2, PW):
        for k, np_, rows in ((15, 1, 2), (15, 8, 3), (16, 4, 2), (17, 2, 2), (17, 8, 1)):
            pieces(nttx, O, p, k, np_, rows, seed=k + np_)
    for n, batch, slices in ((1 << 14, 2, 4), (12345, 3, 2), (1 << 15, 2, 8), (1 << 16, 1, 4), (1 << 17, 2, 4), (1 << 17, 1, 1),
                             (1 << 18, 1, 2), (1 << 19, 1, 8)):
        sliced(nttx, O, n, batch, slices, seed=n % 1000)
    for p in (P9, P4, P2, PW, 7340033):
        for k in (1, 2, 3, 5, 6, 7, 8, 10, 15, 17, 18, 19):
            if (p - 1) % (1 << (k + 1)) == 0:
                negacyclic(nttx, O, p, k, 2 if k < 17 else 1, seed=k)
    for p, k in ((P9, 24), (P2, 24), (P4B, 24), (P2B, 24)):
        cyclic(nttx, O, p, k, 1, seed=k)
        linear(nttx, O, p, (1 << (k - 1)) - 3, 1, seed=k + 1)
    cyclic(nttx, O, P9, 24, 2, seed=99)      # two three-pass products per call (the workspace holds one padded copy per vector)


def suite_three_pass_long(nttx, O):
    """2^25 .. 2^27-point products (split-table column passes of 2^8 and 2^9 points on every class that has such a
    modulus): too long for the CPU oracle, so c(1) = a(1) b(1), c(-1) = a(-1) b(-1), canonical outputs, and equality of
    the linear product with the cyclic one of the same zero-padded vectors (two different first passes)."""
    import torch
    for p, k in ((P9, 25), (P9, 26), (P2B, 26), (P2B, 27)):
        L, n = 1 << k, 1 << (k - 1)
        g = torch.Generator(device="cuda")
        g.manual_seed(k)
        a = torch.zeros((1, L), dtype=torch.int32, device="cuda")
        b = torch.zeros_like(a)
        a[0, :n] = torch.randint(0, p, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
        b[0, :n] = torch.randint(0, p, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
        c1, c2 = torch.empty_like(a), torch.empty_like(a)
        nttx.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c1.data_ptr(), n, p, 1, _stream())
        nttx.dev_cyclic_multiply_u32(a.data_ptr(), b.data_ptr(), c2.data_ptr(), L, p, 1, _stream())
        torch.cuda.synchronize()
        assert torch.equal(c1, c2), ("linear vs cyclic", p, k)
        u = lambda x: x[0].to(torch.int64) & 0xFFFFFFFF                                   # noqa: E731
        assert int(u(c1).max().item()) < p
        sg = torch.ones(L, dtype=torch.int64, device="cuda")
        sg[1::2] = -1
        for s_ in (torch.ones_like(sg), sg):
            ea, eb, ec = (int((u(x) * s_).sum().item()) % p for x in (a, b, c1))
            assert ea * eb % p == ec, ("evaluation", p, k)
        del a, b, c1, c2


# ---- 64-bit lanes: every arithmetic class on every pass shape ---------------------------------------------------------

D64 = {"fp": P51,                               # floating-point class (moduli below 1.35e15)
       "fp_50": 1125899437080577,               # 67108836 * 2^24 + 1, below 2^50: thirteen lazy stages fit when the integer classes are forced
       "fp_lo1_48": 263882790666241,            # 15 * 2^44 + 1: the same with low word 1
       "int": 1350000109944833,                 # 80466372 * 2^24 + 1: lazy integer class, generic reduction
       "int_lo1": 2251756864012289,             # just below 2^51, low word 1: lazy class, word-serial reduction
       "safe": 2305843009146585089,             # 2^26 | p - 1: Harvey class, generic reduction
       "safe_lo1": 4611685941117976577}         # just below 2^62, low word 1


def _horner(coeffs, x, p):
    acc = 0
    for c in coeffs[::-1]:
        acc = (acc * x + int(c)) % p
    return acc


def product64(nttx, O, p, n, batch, seed=1):
    """[batch] linear products on 64-bit lanes: schoolbook oracle up to 4096 coefficients, the four-prime oracle where
    n (p-1)^2 stays below its bound, evaluation at random points otherwise."""
    import torch
    L = 1
    while L < 2 * n:
        L *= 2
    a = np.zeros((batch, L), np.uint64)
    b = np.zeros((batch, L), np.uint64)
    for i in range(batch):
        a[i, :n], b[i, :n] = O.fill_random(n, seed + i, p), O.fill_random(n, seed + 500 + i, p)
    a[0, :n] = p - 1
    da, db = _dev(a, np.int64), _dev(b, np.int64)
    dc = torch.empty_like(da)
    nttx.dev_direct_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, _stream())
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint64)
    exact_crt = n * (p - 1) ** 2 < 469762049 * 998244353 * 1004535809 * 1224736769
    for i in range(batch):
        if n <= 4096:
            want = O.poly_multiply_naive_u64(a[i, :n], b[i, :n], p)
        elif exact_crt:
            want = O.poly_multiply_crt(a[i, :n], b[i, :n], p)
        else:
            assert int(got[i].max()) < p
            for x in (2, 0x123456789ABCDEF % p):
                assert _horner(a[i, :n], x, p) * _horner(b[i, :n], x, p) % p == _horner(got[i, :2 * n - 1], x, p), ("eval64", p, n, i)
            continue
        assert np.array_equal(got[i, :2 * n - 1], want), ("product64", p, n, i)
    return a, b, got


def negacyclic64_any(nttx, O, p, log_len, batch, seed=1):
    """Negacyclic products on any class: lin[i] - lin[i + len] of the schoolbook product (up to 2^12 points) or of the
    library's own linear product on the same lanes (pinned by product64)."""
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
        lin = O.poly_multiply_naive_u64(a[i], b[i], p) if L <= 4096 else nttx.poly_multiply_direct64(a[i], b[i], L, p)
        lin = np.concatenate([lin, np.zeros(1, np.uint64)])
        lo, hi = lin[:L], lin[L:]
        assert np.array_equal(got[i], np.where(lo >= hi, lo - hi, lo + np.uint64(p) - hi)), ("negacyclic64", p, log_len, i)


def transform64(nttx, O, p, log_len, inverse, seed=1):
    """nttx_dev_ntt_u64: X[bitrev(k)] = sum_j x[j] w^(jk) with w = 3^((p-1)/len) (main.cc:156) or its inverse
    (main.cc:157), checked by Horner evaluation at a handful of indices k."""
    import torch
    T = 1 << log_len
    x = O.fill_random(T, seed, p)
    dx = _dev(x.copy(), np.int64)
    nttx.dev_ntt_u64(dx.data_ptr(), T, p, inverse, 1, _stream())
    torch.cuda.synchronize()
    X = dx.cpu().numpy().view(np.uint64)
    g = next(c for c in (3, 2, 5, 6, 7, 10, 11, 13) if pow(c, (p - 1) // 2, p) == p - 1)
    w = pow(g, (p - 1) // T, p)
    if inverse:
        w = pow(w, p - 2, p)
    for k in sorted({0, 1, T - 1, (T // 3) | 1, T // 2}):
        if k >= T:
            continue
        r = int(format(k, f"0{log_len}b")[::-1], 2) if log_len else 0
        assert int(X[r]) == _horner(x, pow(w, k, p), p), ("transform64", p, log_len, inverse, k)


def sparse64(nttx, p, log_len):
    """Linear product of a dense random a (n = 2^(log_len-1) coefficients) with a six-term b of unit coefficients: the
    expected coefficients are sums of shifted copies of a, computed on the device in 64-bit integers (lengths beyond
    what a CPU product finishes in seconds; three-pass plans)."""
    import torch
    L, n = 1 << log_len, 1 << (log_len - 1)
    g = torch.Generator(device="cuda")
    g.manual_seed(77 + log_len)
    a = torch.zeros(L, dtype=torch.int64, device="cuda")
    a[:n] = torch.randint(0, p, (n,), generator=g, device="cuda", dtype=torch.int64)
    b = torch.zeros(L, dtype=torch.int64, device="cuda")
    want = torch.zeros(L, dtype=torch.int64, device="cuda")
    for k in (0, 1, 12345, n // 2 + 5, n - 1, (n // 3) | 1):
        b[k] = 1
        want[k:k + n] = (want[k:k + n] + a[:n]) % p
    c = torch.empty_like(a)
    for x, y in ((a, b), (b, a)):
        nttx.dev_direct_multiply_u64(x.data_ptr(), y.data_ptr(), c.data_ptr(), n, p, 1, _stream())
        torch.cuda.synchronize()
        assert torch.equal(c[: 2 * n - 1], want[: 2 * n - 1]), ("sparse64", p, log_len)


def suite_d64_classes(nttx, O):
    """Five arithmetic classes x every pass shape of the 64-bit lanes: one-pass rows of 2 .. 2^13 points (products,
    negacyclic products, transforms), two-pass plans over 2^6 .. 2^9-point column passes, a three-pass plan."""
    for name, p in D64.items():
        for k in range(1, 14):
            product64(nttx, O, p, (1 << (k - 1)) - (k % 3 if k > 3 else 0), 2, seed=k)
            if (p - 1) % (2 << k) == 0:
                negacyclic64_any(nttx, O, p, k, 2, seed=k)
            transform64(nttx, O, p, k, bool(k & 1), seed=k)
            transform64(nttx, O, p, k, not (k & 1), seed=k + 1)
        for k in (14, 15, 16, 17, 18):
            product64(nttx, O, p, (1 << (k - 1)) - 11, 1, seed=k)
            negacyclic64_any(nttx, O, p, k, 1, seed=k)
            transform64(nttx, O, p, k, bool(k & 1), seed=k)
        for k in (20, 22):
            if len(nttx.plan_info_u64(1 << k, p)) == 2:
                negacyclic64_any(nttx, O, p, k, 1, seed=k)
        for k in ((23, 24, 27) if name == "fp" else () if name.startswith("fp_") else (23,)):      # three passes: inner column passes of 2^7, 2^8, 2^9 points
            assert len(nttx.plan_info_u64(1 << k, p)) == 3
            sparse64(nttx, p, k)


SUITES = {"col": suite_col, "colpass": suite_colpass, "layouts": suite_layouts, "three_pass_long": suite_three_pass_long,
          "d64_classes": suite_d64_classes, "col_lite": suite_col_lite, "col_big": suite_col_big, "rows": suite_rows, "crt": suite_crt, "d64": suite_d64}


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
