"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer work: the bar is bit-exact equality."""
import hashlib
import json
import os
from math import isqrt

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P_ID = {0: 7340033, 1: 7340033, 2: 104857601, 3: 469762049, 4: 1337006139375617}  # test.o:3-11
NTT_PRIMES = (7340033, 104857601, 469762049, 998244353, 1004535809, 1224736769)


def rand(oracle, n, seed, p):
    return oracle.fill_random(n, seed, p)


def test_id0_reference_golden(nttx):
    """files/0.out of the reference, input recovered by factoring the output polynomial."""
    want = np.loadtxt(os.path.join(GOLD, "0.out"), dtype=np.uint64)
    for dt in (np.int32, np.uint64):
        got = nttx.poly_multiply([1, 5, 5, 4], [4, 1, 5, 2], 4, 7340033, dtype=dt)
        assert np.array_equal(got.astype(np.uint64), want)
    assert np.array_equal(nttx.poly_multiply_crt([1, 5, 5, 4], [4, 1, 5, 2], 4, 7340033), want)


@pytest.mark.parametrize("p", NTT_PRIMES)
def test_direct_small_and_ragged(nttx, oracle, p):
    for n in (1, 2, 3, 4, 5, 8, 31, 32, 33, 64, 255, 1000, 1024, 1025, 2048, 4096, 5000):
        a, b = rand(oracle, n, 7 * n + 1, p), rand(oracle, n, 7 * n + 2, p)
        want = oracle.poly_multiply_crt(a, b, p)
        assert nttx.path_for(p, n) == nttx.PATH_DIRECT32
        got = nttx.poly_multiply(a, b, n, p, dtype=np.uint64)
        assert np.array_equal(got, want), (p, n)
        got32 = nttx.poly_multiply(a.astype(np.int32), b.astype(np.int32), n, p, dtype=np.int32)
        assert np.array_equal(got32.astype(np.uint64), want), (p, n)


@pytest.mark.parametrize("p", NTT_PRIMES)
def test_transform_matches_ntt_iter_barrett(nttx, oracle, p):
    for logn in range(0, 15):
        n = 1 << logn
        x = rand(oracle, n, 100 + logn, p).astype(np.uint32)
        for inv in (False, True):
            want = oracle.ntt_iter_barrett(x, p, inv)
            got = nttx.ntt(x, p, inv)
            assert np.array_equal(got, want), (p, logn, inv)


@pytest.mark.parametrize("p", (469762049, 998244353, 1004535809, 1224736769))
def test_cyclic_multiply_matches_ntt_compute(nttx, oracle, p):
    for logn in (1, 2, 3, 6, 9, 11, 12, 13, 16):
        n = 1 << logn
        a, b = rand(oracle, n, 5 + logn, p).astype(np.uint32), rand(oracle, n, 55 + logn, p).astype(np.uint32)
        assert np.array_equal(nttx.cyclic_multiply(a, b, p), oracle.ntt_compute(a, b, p)), (p, logn)


@pytest.mark.parametrize("p", (1337006139375617, 263882790666241, 7340033, 2147483647, 1125899906842597))
def test_crt_combine(nttx, oracle, p):
    n = 2000
    # residues of genuine integers below M keep the reference's sum%M exact
    res = [rand(oracle, 4096, 900 + k, m).astype(np.uint32) for k, m in enumerate(nttx.CRT_PRIMES)]
    want = oracle.crt_combine(res, 2 * n - 1, p)
    got = nttx.crt_combine(res, 2 * n - 1, p)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("p", (1337006139375617, 263882790666241, 281474976710597))
def test_crt_path_small(nttx, oracle, p):
    """The reference's route (four primes + CRT, main.cc:286-338), forced; 2^48 - 59 has no roots of
    its own, so the automatic choice takes this route too."""
    for n in (1, 2, 5, 64, 1000, 4096):
        a, b = rand(oracle, n, n + 11, p), rand(oracle, n, n + 12, p)
        want = oracle.poly_multiply_crt(a, b, p)
        assert np.array_equal(nttx.poly_multiply_crt(a, b, n, p), want), (p, n)
        if p == 281474976710597 and n >= 5:      # p - 1 = 4 * odd: transforms longer than 4 points do not exist
            assert nttx.path_for(p, n) == nttx.PATH_CRT4
            assert np.array_equal(nttx.poly_multiply(a, b, n, p, dtype=np.uint64), want), (p, n)
        if n <= 64:
            assert np.array_equal(want, oracle.poly_multiply_naive_u64(a, b, p))


def test_crt_fewer_primes_bound(nttx, oracle):
    """The CRT route uses only three, two or one of the four primes (the smallest where they suffice, else
    the largest) where n*(p-1)^2 stays below their product.  Worst-case inputs (every coefficient p-1, so the middle coefficient of the
    integer product reaches n*(p-1)^2) on both sides of each bound, plus random inputs, against the
    four-prime oracle."""
    primes = (469762049, 998244353, 1004535809, 1224736769)
    for n in (1000, 4096, 65536):
        moduli = [2147483647, (1 << 36) - 5, 65537, 97, 12345678]
        for ns in (1, 2, 3):
            for subset in (primes[:ns], primes[4 - ns:]):     # the cheapest and the largest ns primes
                prod = 1
                for m in subset:
                    prod *= m
                top = isqrt((prod - 1) // n) + 1          # the largest modulus with n*(p-1)^2 below the product
                moduli += [top, top + 1]
        for p in moduli:
            if nttx.path_for(p, n) != nttx.PATH_CRT4:
                p += 1                      # an NTT-friendly prime would take a direct route: step off it
            assert nttx.path_for(p, n) == nttx.PATH_CRT4, (p, n)
            worst = np.full(n, p - 1, dtype=np.uint64)
            assert np.array_equal(nttx.poly_multiply(worst, worst, n, p, dtype=np.uint64), oracle.poly_multiply_crt(worst, worst, p)), (n, p)
            a, b = rand(oracle, n, 11 + n, p), rand(oracle, n, 12 + n, p)
            assert np.array_equal(nttx.poly_multiply(a, b, n, p, dtype=np.uint64), oracle.poly_multiply_crt(a, b, p)), (n, p)


@pytest.mark.parametrize("p", (2, 3, 4, 6, 12, 17, 65536, 65537, 1 << 31, 3 << 40, 1 << 62, 1000000007, (1 << 61) - 1))
def test_any_modulus_like_main_cc(nttx, oracle, p):
    """main.cc reduces the exact integer product modulo whatever p it is given (main.cc:334-336):
    small, even and composite moduli included, as long as n (p-1)^2 stays below its CRT bound."""
    for n in (1, 4, 33, 500):
        if n * (p - 1) ** 2 >= 469762049 * 998244353 * 1004535809 * 1224736769 and nttx.path_for(p, n) == nttx.PATH_CRT4:
            with pytest.raises(nttx.NttxError):
                nttx.poly_multiply(rand(oracle, n, 1, p), rand(oracle, n, 2, p), n, p, dtype=np.uint64)
            continue
        a, b = rand(oracle, n, 3 * n + 1, p), rand(oracle, n, 3 * n + 2, p)
        want = oracle.poly_multiply_naive_u64(a, b, p)
        if n * (p - 1) ** 2 < 469762049 * 998244353 * 1004535809 * 1224736769:
            assert np.array_equal(oracle.poly_multiply_crt(a, b, p), want)
        assert np.array_equal(nttx.poly_multiply(a, b, n, p, dtype=np.uint64), want), (p, n)
        if p < (1 << 31):
            got = nttx.poly_multiply(a.astype(np.int32), b.astype(np.int32), n, p, dtype=np.int32)
            assert np.array_equal(got.astype(np.uint64), want), (p, n)


@pytest.mark.parametrize("tid", (1, 2, 3, 4))
def test_harness_sizes_full(nttx, oracle, tid):
    """n = 131072 (product length 2^18) on the four harness moduli, seeded inputs."""
    p, n = P_ID[tid], 131072
    a, b = rand(oracle, n, 1000 + tid, p), rand(oracle, n, 2000 + tid, p)
    want = oracle.poly_multiply_crt(a, b, p)
    got = nttx.poly_multiply(a, b, n, p, dtype=np.uint64)
    assert np.array_equal(got, want)
    # the forced CRT path (what main.cc runs for every id) gives the same coefficients
    assert np.array_equal(nttx.poly_multiply_crt(a, b, n, p), want)
    c = json.load(open(os.path.join(GOLD, "harness.json")))["cases"][f"{tid}_17"]
    assert hashlib.sha256(got.astype("<u8").tobytes()).hexdigest() == c["sha256"]


def test_batch(nttx, oracle):
    p = 469762049
    for n, batch in ((1000, 7), (4096, 5), (32768, 3)):
        a = rand(oracle, n * batch, 31, p).astype(np.uint32).reshape(batch, n)
        b = rand(oracle, n * batch, 32, p).astype(np.uint32).reshape(batch, n)
        got = nttx.poly_multiply_batch(a, b, p)
        for i in range(batch):
            want = oracle.poly_multiply_crt(a[i].astype(np.uint64), b[i].astype(np.uint64), p)
            assert np.array_equal(got[i].astype(np.uint64), want), (n, i)


def _horner(v, x, p):
    r = 0
    for c in v[::-1]:
        r = (r * x + int(c)) % p
    return r


def test_three_pass_length(nttx, oracle):
    """2^24-point cyclic product (three passes): a sparse factor makes the expected product a handful of
    shifted copies of the dense one, so no CPU transform of that size is needed."""
    p, L = 469762049, 1 << 24
    assert len(nttx.plan_info(L)) == 3
    a = rand(oracle, L, 41, p).astype(np.uint32)
    b = np.zeros(L, np.uint32)
    terms = {0: 3, 1: p - 1, 12345: 77, L // 2 + 5: 123456789, L - 1: 2}
    for k, v in terms.items():
        b[k] = v
    want = np.zeros(L, np.uint64)
    for k, v in terms.items():
        want = (want + np.roll(a, k).astype(np.uint64) * np.uint64(v)) % np.uint64(p)
    assert np.array_equal(nttx.cyclic_multiply(a, b, p), want.astype(np.uint32))
    assert np.array_equal(nttx.cyclic_multiply(b, a, p), want.astype(np.uint32))


_THREE_PASS_22 = """
import sys, numpy as np
sys.path.insert(0, {root!r})
import ntt_amd
from oracle import pyoracle
O = pyoracle.Oracle()
ntt_amd.init(0)
p, n = 469762049, 1 << 21
assert len(ntt_amd.plan_info(1 << 22)) == 3, ntt_amd.plan_info(1 << 22)
a, b = O.fill_random(n, 41, p), O.fill_random(n, 42, p)
assert np.array_equal(ntt_amd.poly_multiply(a, b, n, p, dtype=np.uint64), O.poly_multiply_crt(a, b, p))
ntt_amd.shutdown()
print("three-pass ok")
"""


def test_three_pass_plan_at_2_22(oracle):
    """The three-pass plan on a length the default plan covers in two passes (long rows switched off in a
    child process), against the oracle's product."""
    import subprocess
    import sys
    root = os.path.dirname(GOLD[:-len("/golden")])
    env = dict(os.environ, NTTX_LONGROW="0")
    r = subprocess.run([sys.executable, "-c", _THREE_PASS_22.format(root=root)], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and "three-pass ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("p", (469762049, 998244353, 1224736769))
def test_long_row_single_pass(nttx, oracle, p):
    """Lengths 2^12 .. 2^14 run as ONE pass (a whole workgroup per row); all three modulus classes and
    the signed-lane class, batches that do and do not fill the grid, against ntt_compute_omp."""
    for logn, batch in ((12, 5), (13, 3), (14, 2)):
        n = 1 << logn
        assert nttx.plan_info(n) == [logn]
        a = rand(oracle, n * batch, 300 + logn, p).astype(np.uint32).reshape(batch, n)
        b = rand(oracle, n * batch, 400 + logn, p).astype(np.uint32).reshape(batch, n)
        got = _dev_cyclic(nttx, a, b, p)
        for i in range(batch):
            assert np.array_equal(got[i], oracle.ntt_compute(a[i], b[i], p)), (p, logn, i)


def _dev_cyclic(nttx, a, b, p):
    import torch
    da, db = torch.from_numpy(a.view(np.int32)).cuda(), torch.from_numpy(b.view(np.int32)).cuda()
    dc = torch.empty_like(da)
    nttx.dev_cyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), a.shape[1], p, a.shape[0],
                                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dc.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("p", (132120577, 104857601, 7340033))
def test_sixteen_multiple_signed_rows(nttx, oracle, p):
    """Moduli below 2^27 run their fused row kernels on signed lanes with sixteen multiples of headroom: the 512-point
    kernel (single pass and behind a column pass), the long-row kernels with radix-8 and radix-16 rounds, and a three-pass
    length, on random operands and on the operands that drive every sum to its bound (all p - 1, and p - 1 against 1,
    p - 1 alternating).  132120577 = 63 * 2^21 + 1 is the largest such modulus below 2^27 with these lengths; 3 is a
    quadratic residue there, so the reference's transform (root 3, main.cc:161) does not exist for it and the expected
    product is the folded linear product of the reference's four-prime CRT route instead."""
    def expected(x, y):
        if p != 132120577:
            return oracle.ntt_compute(x, y, p)
        lin = np.zeros(2 * len(x), dtype=np.uint64)
        lin[: 2 * len(x) - 1] = oracle.poly_multiply_crt(x.astype(np.uint64), y.astype(np.uint64), p)
        return ((lin[: len(x)] + lin[len(x):]) % np.uint64(p)).astype(np.uint32)

    for logn, batch in ((9, 3), (11, 2), (12, 2), (13, 2), (14, 1), (18, 1), (20, 1), (21, 1)):
        n = 1 << logn
        if (p - 1) % n or (p == 132120577 and logn > 20):
            continue
        a = rand(oracle, n * batch, 700 + logn, p).astype(np.uint32).reshape(batch, n)
        b = rand(oracle, n * batch, 800 + logn, p).astype(np.uint32).reshape(batch, n)
        a[0, :] = p - 1
        b[0, :] = p - 1
        if batch > 1:
            a[1, :] = p - 1
            b[1, 0::2] = 1
            b[1, 1::2] = p - 1
        got = _dev_cyclic(nttx, a, b, p)
        for i in range(batch):
            assert np.array_equal(got[i], expected(a[i], b[i])), (p, logn, i)


@pytest.mark.parametrize("p,logn", [(469762049, 21), (469762049, 22), (469762049, 23), (998244353, 22), (1224736769, 23)])
def test_long_row_two_pass(nttx, oracle, p, logn):
    """One column pass around long rows (2^12 .. 2^14 points): cyclic products of length 2^21 .. 2^23
    against ntt_compute_omp."""
    n = 1 << logn
    assert len(nttx.plan_info(n)) == 2
    a, b = rand(oracle, n, 500 + logn, p).astype(np.uint32), rand(oracle, n, 600 + logn, p).astype(np.uint32)
    assert np.array_equal(nttx.cyclic_multiply(a, b, p), oracle.ntt_compute(a, b, p))


def test_inputs_not_reduced(nttx, oracle):
    """Coefficients >= p are legal for the (uint64_t) flavour: main.cc reduces them per prime."""
    p, n = 7340033, 100
    a = rand(oracle, n, 1, 1 << 40)
    b = rand(oracle, n, 2, 1 << 40)
    want = oracle.poly_multiply_crt(a, b, p)
    assert np.array_equal(nttx.poly_multiply(a, b, n, p, dtype=np.uint64), want)


def test_crt_route_inputs_above_modulus(nttx, oracle):
    """The CRT route with coefficients far above p (legal for the uint64_t flavour, main.cc:300-301 reduces per
    prime): the route picks fewer than four primes from n*(p-1)^2, so words at or above p must be reduced modulo p
    first.  Host and device-resident entry points against the reference's four-prime product, which is exact
    here (n * 2^45 * 2^45 stays below its M)."""
    import torch
    n, batch = 100, 3
    for p in (1000000007, 65539, 97, 281474976710597, 12345678, 3 << 20):
        assert nttx.path_for(p, n) == nttx.PATH_CRT4       # none of these has the roots a direct route needs
        a = np.stack([rand(oracle, n, 40 + i, 1 << 45) for i in range(batch)])
        b = np.stack([rand(oracle, n, 50 + i, 1 << 45) for i in range(batch)])
        a[0, 0], b[0, 0], a[1, 5] = p, p - 1, (1 << 45) - 1
        want = np.stack([oracle.poly_multiply_crt(a[i], b[i], p) for i in range(batch)])
        assert np.array_equal(want[0], oracle.poly_multiply_naive_u64(a[0] % p, b[0] % p, p))
        for i in range(batch):
            assert np.array_equal(nttx.poly_multiply_crt(a[i], b[i], n, p), want[i]), (p, i)
            assert np.array_equal(nttx.poly_multiply(a[i], b[i], n, p, dtype=np.uint64), want[i]), (p, i)
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        dc = torch.empty((batch, 2 * n - 1), dtype=torch.int64, device="cuda")
        nttx.dev_crt_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(dc.cpu().numpy().view(np.uint64), want), p


def test_crt_exact_bound_edge(nttx, oracle):
    """n*(p-1)^2 < M decides exactly: the largest admissible modulus multiplies worst-case inputs correctly with all
    four primes, the next one is refused."""
    big_m = 469762049 * 998244353 * 1004535809 * 1224736769
    for n in (1000, 4096):
        top = isqrt((big_m - 1) // n) + 1
        while nttx.path_for(top, n) != nttx.PATH_CRT4:
            top -= 1
        assert n * (top - 1) ** 2 < big_m
        worst = np.full(n, top - 1, dtype=np.uint64)
        assert np.array_equal(nttx.poly_multiply_crt(worst, worst, n, top), oracle.poly_multiply_crt(worst, worst, top)), (n, top)
        over = isqrt((big_m - 1) // n) + 2
        with pytest.raises(nttx.NttxError) as e:
            nttx.poly_multiply_crt(worst, worst, n, over)
        assert e.value.code == -2


def test_batch_beyond_grid_dimension(nttx, oracle):
    """70000 products in one call: the element-wise kernels (pad, unpad, split, CRT) stride over the batch instead of
    putting it into a grid dimension, which holds at most 65535."""
    import torch
    batch, n = 70000, 32
    p = 469762049
    a = rand(oracle, n * batch, 5, p).reshape(batch, n).astype(np.uint32)
    b = rand(oracle, n * batch, 6, p).reshape(batch, n).astype(np.uint32)
    got = nttx.poly_multiply_batch(a, b, p)
    probe = [0, 1, 65534, 65535, 65536, 65537, batch - 1]
    for i in probe:
        assert np.array_equal(got[i].astype(np.uint64), oracle.poly_multiply_crt(a[i].astype(np.uint64), b[i].astype(np.uint64), p)), i
    # every product: c(1) = a(1) b(1) and c(-1) = a(-1) b(-1) modulo p
    sign = np.where(np.arange(2 * n - 1) % 2 == 0, 1, -1).astype(np.int64)
    a64, b64, c64 = a.astype(np.int64), b.astype(np.int64), got.astype(np.int64)
    for sg in (np.ones(2 * n - 1, np.int64), sign):
        ea, eb, ec = (a64 * sg[:n]).sum(1) % p, (b64 * sg[:n]).sum(1) % p, (c64 * sg).sum(1) % p
        assert np.array_equal(ec, np.array([int(x) * int(y) % p for x, y in zip(ea, eb)], dtype=np.int64))
    # the CRT route on device arrays at the same batch
    q = 1000000007
    a8, b8 = (a.astype(np.uint64) % q), (b.astype(np.uint64) % q)
    da, db = torch.from_numpy(a8.view(np.int64)).cuda(), torch.from_numpy(b8.view(np.int64)).cuda()
    dc = torch.empty((batch, 2 * n - 1), dtype=torch.int64, device="cuda")
    nttx.dev_crt_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, q, batch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got8 = dc.cpu().numpy().view(np.uint64)
    for i in probe:
        assert np.array_equal(got8[i], oracle.poly_multiply_crt(a8[i], b8[i], q)), i
    c8 = got8.astype(np.int64)
    ea, eb, ec = a8.astype(np.int64).sum(1) % q, b8.astype(np.int64).sum(1) % q, c8.sum(1) % q
    assert np.array_equal(ec, np.array([int(x) * int(y) % q for x, y in zip(ea, eb)], dtype=np.int64))


def test_calls_on_two_streams_share_the_workspace_safely(nttx, oracle):
    """Two device-resident calls on different streams, the second enqueued while the first still runs: the library
    orders the second behind the first on the device (one workspace per context), so both results are right."""
    import torch
    p, L, batch = 469762049, 1 << 18, 48
    n = L // 2
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    def inputs():
        x = torch.zeros((batch, L), dtype=torch.int32, device="cuda")
        x[:, :n] = torch.randint(0, p, (batch, n), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
        return x
    a1, b1, a2, b2 = inputs(), inputs(), inputs(), inputs()
    ref1, ref2 = torch.empty_like(a1), torch.empty_like(a1)
    st0 = torch.cuda.current_stream().cuda_stream
    nttx.dev_poly_multiply_u32(a1.data_ptr(), b1.data_ptr(), ref1.data_ptr(), n, p, batch, st0)
    torch.cuda.synchronize()
    nttx.dev_poly_multiply_u32(a2.data_ptr(), b2.data_ptr(), ref2.data_ptr(), n, p, batch, st0)
    torch.cuda.synchronize()
    want = oracle.ntt_compute(a2[3].cpu().numpy().view(np.uint32), b2[3].cpu().numpy().view(np.uint32), p)
    assert np.array_equal(ref2[3].cpu().numpy().view(np.uint32), want)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        c1, c2 = torch.zeros_like(a1), torch.zeros_like(a1)
        torch.cuda.synchronize()
        nttx.dev_poly_multiply_u32(a1.data_ptr(), b1.data_ptr(), c1.data_ptr(), n, p, batch, s1.cuda_stream)
        nttx.dev_poly_multiply_u32(a2.data_ptr(), b2.data_ptr(), c2.data_ptr(), n, p, batch, s2.cuda_stream)
        # a host-buffer call in between runs on the null stream and uses the same workspace
        small = nttx.poly_multiply([1, 5, 5, 4], [4, 1, 5, 2], 4, 7340033, dtype=np.uint64)
        nttx.dev_poly_multiply_u32(a1.data_ptr(), b1.data_ptr(), c1.data_ptr(), n, p, batch, s1.cuda_stream)
        torch.cuda.synchronize()
        assert list(small) == [4, 21, 30, 48, 39, 30, 8]
        assert torch.equal(c1, ref1) and torch.equal(c2, ref2), rep


def test_page_locked_host_buffers(nttx, oracle):
    """nttx_host_alloc: page-locked arrays for the harness's long-lived a, b, ab; same coefficients as pageable ones."""
    import ctypes as C
    u64p = C.POINTER(C.c_uint64)
    for p, n in ((469762049, 131072), (1337006139375617, 70001), (1000000007, 5000)):
        a, b, ab = nttx.host_alloc(n, np.uint64), nttx.host_alloc(n, np.uint64), nttx.host_alloc(2 * n - 1, np.uint64)
        try:
            a[:], b[:] = rand(oracle, n, 3, p), rand(oracle, n, 4, p)
            for _ in range(2):
                ab[:] = 0
                rc = nttx.lib().nttx_poly_multiply_u64(a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), ab.ctypes.data_as(u64p), n, p)
                assert rc == 0
                assert np.array_equal(ab, oracle.poly_multiply_crt(a, b, p)), (p, n)
        finally:
            for x in (a, b, ab):
                nttx.host_free(x)


def test_errors(nttx):
    with pytest.raises(nttx.NttxError):
        nttx.poly_multiply([1], [1], 0, 7340033)
    with pytest.raises(nttx.NttxError):   # n*(p-1)^2 >= M: the reference would silently wrap
        nttx.poly_multiply_crt(np.ones(1 << 17, np.uint64), np.ones(1 << 17, np.uint64), 1 << 17, (1 << 61) - 1)
    with pytest.raises(nttx.NttxError):   # 7340033 - 1 = 7 * 2^20: no 2^21-point transform
        nttx.ntt(np.zeros(1 << 21, np.uint32), 7340033)


def test_device_resident_api(nttx, oracle):
    import torch
    p, L, batch = 469762049, 1 << 18, 4
    n = L // 2
    a = np.zeros((batch, L), np.uint32)
    b = np.zeros((batch, L), np.uint32)
    a[:, :n] = rand(oracle, n * batch, 71, p).reshape(batch, n)
    b[:, :n] = rand(oracle, n * batch, 72, p).reshape(batch, n)
    da = torch.from_numpy(a.view(np.int32)).cuda()
    db = torch.from_numpy(b.view(np.int32)).cuda()
    dc = torch.empty_like(da)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):  # second call: inputs must have survived the first
        nttx.dev_cyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, st)
    torch.cuda.synchronize()
    c = dc.cpu().numpy().view(np.uint32)
    assert np.array_equal(da.cpu().numpy().view(np.uint32), a)
    for i in range(batch):
        assert np.array_equal(c[i], oracle.ntt_compute(a[i], b[i], p)), i


def _seeds(n, p):
    return (n * 1000003 + p) % (1 << 62), (n * 999983 + 7 * p + 1) % (1 << 62)


def test_golden_small_products(nttx, oracle):
    """Outputs of the real reference programs (tests/golden/small_products.npz)."""
    z = np.load(os.path.join(GOLD, "small_products.npz"))
    for key in z.files:
        kind, p, n = key.split("_")
        p, n = int(p), int(n)
        sa, sb = _seeds(n, p)
        a, b = oracle.fill_random(n, sa, p), oracle.fill_random(n, sb, p)
        if kind == "main":
            assert np.array_equal(nttx.poly_multiply(a, b, n, p, dtype=np.uint64), z[key]), key
            assert np.array_equal(nttx.poly_multiply_crt(a, b, n, p), z[key]), key
        else:
            got = nttx.poly_multiply(a.astype(np.int32), b.astype(np.int32), n, p, dtype=np.int32)
            assert np.array_equal(got.astype(np.uint32), z[key]), key


def test_golden_transforms(nttx, oracle):
    z = np.load(os.path.join(GOLD, "transforms.npz"))
    for key in z.files:
        d, p, logn = key.split("_")
        p, logn = int(p), int(logn)
        x = oracle.fill_random(1 << logn, 424242 + logn + p % 1000, p).astype(np.uint32)
        assert np.array_equal(nttx.ntt(x, p, d == "inv"), z[key]), key


@pytest.mark.parametrize("logn", (15, 16))
def test_golden_harness_hashes(nttx, oracle, logn):
    h = json.load(open(os.path.join(GOLD, "harness.json")))
    for tid in (1, 2, 3, 4):
        c = h["cases"][f"{tid}_{logn}"]
        a = oracle.fill_random(c["n"], 1000 + tid + 10 * logn, c["p"])
        b = oracle.fill_random(c["n"], 2000 + tid + 10 * logn, c["p"])
        got = nttx.poly_multiply(a, b, c["n"], c["p"], dtype=np.uint64)
        assert hashlib.sha256(got.astype("<u8").tobytes()).hexdigest() == c["sha256"]


def test_linearity_and_roundtrip_large(nttx, oracle):
    """Size-independent properties at 2^20 points: forward then inverse transform is n*x, and
    the product is bilinear."""
    p, L = 998244353, 1 << 20
    x = oracle.fill_random(L, 5, p).astype(np.uint32)
    y = nttx.ntt(nttx.ntt(x, p, False), p, True)
    assert np.array_equal(y.astype(np.uint64), x.astype(np.uint64) * (L % p) % p)
    n = L // 2
    a, b, c = (oracle.fill_random(n, s, p) for s in (6, 7, 8))
    ab = nttx.poly_multiply(a, b, n, p, dtype=np.uint64)
    ac = nttx.poly_multiply(a, c, n, p, dtype=np.uint64)
    a_bc = nttx.poly_multiply(a, (b + c) % p, n, p, dtype=np.uint64)
    assert np.array_equal(a_bc, (ab + ac) % p)


def test_harness_flow(nttx, oracle, tmp_path):
    """fRead -> poly_multiply -> fCheck -> fWrite over ids 0..4 with the reference's file formats."""
    from oracle import pyoracle as po
    from ntt_amd import harness
    d = tmp_path / "nttdata"
    d.mkdir()
    cases = [([1, 5, 5, 4], [4, 1, 5, 2], 7340033)]
    for tid in (1, 2, 3, 4):
        p = P_ID[tid]
        cases.append((rand(oracle, 4096, 1 + tid, p), rand(oracle, 4096, 9 + tid, p), p))
    for i, (a, b, p) in enumerate(cases):
        po.write_in_file(str(d / f"{i}.in"), a, b, p)
        po.write_out_file(str(d / f"{i}.out"), oracle.poly_multiply_crt(a, b, p))
    assert harness.run(0, 4, str(d), str(tmp_path / "files"))
    assert open(tmp_path / "files" / "0.out").read() == open(os.path.join(GOLD, "0.out")).read()


def test_cpp_harness_binary(oracle, tmp_path):
    """nttx_harness (C++ driver over the C ABI) on ids 0..4: stdout lines and dumps as the reference's."""
    import subprocess
    from oracle import pyoracle as po
    exe = os.path.join(os.path.dirname(GOLD), "..", "ntt_amd", "csrc", "nttx_harness")
    d = tmp_path / "nttdata"
    d.mkdir()
    (tmp_path / "files").mkdir()
    cases = [([1, 5, 5, 4], [4, 1, 5, 2], 7340033)]
    for tid in (1, 2, 3, 4):
        p = P_ID[tid]
        cases.append((rand(oracle, 131072, 1000 + tid, p), rand(oracle, 131072, 2000 + tid, p), p))
    wants = []
    for i, (a, b, p) in enumerate(cases):
        po.write_in_file(str(d / f"{i}.in"), a, b, p)
        wants.append(oracle.poly_multiply_crt(a, b, p))
        po.write_out_file(str(d / f"{i}.out"), wants[-1])
    env = dict(os.environ, NTT_DATA_DIR=str(d), NTT_OUT_DIR=str(tmp_path / "files"))
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("多项式乘法结果正确") == 5
    assert r.stdout.count("average latency for n = 131072") == 4
    for i in range(5):
        got = np.loadtxt(tmp_path / "files" / f"{i}.out", dtype=np.uint64, ndmin=1)
        assert np.array_equal(got, wants[i])


def test_cpp_harness_binary_int_flavour(oracle, tmp_path):
    """nttx_harness_int: the (int, int) course harness of monttry2.cpp:344, 354-400 (poly_multiply(int*, int*, int*,
    int, int) between fRead and fCheck) on ids 0..3 at the harness sizes; id 4's 51-bit modulus does not fit an int
    and is reported as an error (the reference reads garbage there)."""
    import subprocess
    from oracle import pyoracle as po
    exe = os.path.join(os.path.dirname(GOLD), "..", "ntt_amd", "csrc", "nttx_harness_int")
    d = tmp_path / "nttdata"
    d.mkdir()
    (tmp_path / "files").mkdir()
    cases = [([1, 5, 5, 4], [4, 1, 5, 2], 7340033)]
    for tid in (1, 2, 3, 4):
        p = P_ID[tid]
        cases.append((rand(oracle, 131072, 1000 + tid, p), rand(oracle, 131072, 2000 + tid, p), p))
    wants = []
    for i, (a, b, p) in enumerate(cases):
        po.write_in_file(str(d / f"{i}.in"), a, b, p)
        wants.append(oracle.poly_multiply_crt(a, b, p))
        po.write_out_file(str(d / f"{i}.out"), wants[-1])
    env = dict(os.environ, NTT_DATA_DIR=str(d), NTT_OUT_DIR=str(tmp_path / "files"))
    r = subprocess.run([exe, "0", "3"], env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("多项式乘法结果正确") == 4
    assert r.stdout.count("average latency for n = 131072") == 3 and r.stdout.count("(us)") == 4
    for i in range(4):
        got = np.loadtxt(tmp_path / "files" / f"{i}.out", dtype=np.uint64, ndmin=1)
        assert np.array_equal(got, wants[i])
    assert open(tmp_path / "files" / "0.out").read() == open(os.path.join(GOLD, "0.out")).read()
    r = subprocess.run([exe, "4", "4"], env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 1 and "does not fit" in r.stderr


@pytest.mark.parametrize("n", (131072, 100003, 70000, 65537))
def test_device_linear_product_skips_padding(nttx, oracle, n):
    """nttx_dev_poly_multiply_u32: zero-padded device arrays, padding never read; ragged n."""
    import torch
    p, batch = 469762049, 3
    L = 1
    while L < 2 * n:
        L *= 2
    a = np.zeros((batch, L), np.uint32)
    b = np.zeros((batch, L), np.uint32)
    a[:, :n] = rand(oracle, n * batch, 81, p).reshape(batch, n)
    b[:, :n] = rand(oracle, n * batch, 82, p).reshape(batch, n)
    da = torch.from_numpy(a.view(np.int32)).cuda()
    db = torch.from_numpy(b.view(np.int32)).cuda()
    dc = torch.full_like(da, 7)
    nttx.dev_poly_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = dc.cpu().numpy().view(np.uint32)
    for i in range(batch):
        want = oracle.poly_multiply_crt(a[i, :n].astype(np.uint64), b[i, :n].astype(np.uint64), p)
        assert np.array_equal(c[i, :2 * n - 1].astype(np.uint64), want), (n, i)
        assert not c[i, 2 * n - 1:].any()


_NEGA = ([(469762049, k, 3) for k in range(1, 17)] + [(998244353, k, 2) for k in (4, 8, 9, 10, 12, 14, 15, 17)]
         + [(1224736769, k, 2) for k in (6, 9, 11, 13, 16)] + [(469762049, 18, 2), (469762049, 20, 1), (998244353, 21, 1),
                                                               (469762049, 24, 1)]
         + [(104857601, k, 2) for k in (9, 11, 13, 14, 16)] + [(7340033, 9, 2), (7340033, 12, 1)])   # sixteen-multiple signed rows


@pytest.mark.parametrize("p,log_len,batch", _NEGA)
def test_negacyclic_product(nttx, oracle, p, log_len, batch):
    """c = a b mod (x^len + 1, p): equals lin[i] - lin[i + len] of the linear product, which the
    oracle gives as a cyclic product of the zero-padded inputs (ntt_compute, main.cc:185-209).  Every
    kernel family that applies the psi factors itself (tiled rows, 512-point rows, long rows, two-pass
    plans) on all modulus classes, and a three-pass length (separate element-wise passes)."""
    import torch
    L = 1 << log_len
    a = np.stack([oracle.fill_random(L, 300 + i, p) for i in range(batch)]).astype(np.uint32)
    b = np.stack([oracle.fill_random(L, 400 + i, p) for i in range(batch)]).astype(np.uint32)
    da, db = torch.from_numpy(a.view(np.int32)).cuda(), torch.from_numpy(b.view(np.int32)).cuda()
    dc = torch.empty_like(da)
    nttx.dev_negacyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint32)
    for i in range(batch):
        pa, pb = np.zeros(2 * L, dtype=np.uint32), np.zeros(2 * L, dtype=np.uint32)
        pa[:L], pb[:L] = a[i], b[i]
        lin = oracle.ntt_compute(pa, pb, p).astype(np.int64)
        want = (lin[:L] - lin[L:]) % p
        assert np.array_equal(got[i].astype(np.int64), want), (p, log_len, i)
    assert torch.equal(da.cpu(), torch.from_numpy(a.view(np.int32)))     # inputs untouched
    with pytest.raises(nttx.NttxError):                                  # 7340033 - 1 = 7 * 2^20: no 2^21-th root
        nttx.dev_negacyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), 1 << 20, 7340033, 1, 0)


def test_fuzz_small_products(nttx, oracle):
    """Seeded sweep over ragged lengths and every route: 32-bit-lane, 64-bit-lane and CRT moduli
    (prime, composite, even), int and uint64 flavours, against the oracle's schoolbook product."""
    rng = np.random.default_rng(20240607)
    moduli = [7340033, 104857601, 469762049, 998244353, 1004535809, 1224736769, 2147483647, 1000000007,
              1337006139375617, 263882790666241, 8588886017, 4611685941117976577, 2305843009146585089,
              281474976710597, 97, 2, 65536, 12345678, (1 << 61) - 1, 3 << 40]
    big_m = 469762049 * 998244353 * 1004535809 * 1224736769
    done = 0
    for _ in range(260):
        p = moduli[int(rng.integers(len(moduli)))]
        n = int(rng.integers(1, 700))
        if nttx.path_for(p, n) == nttx.PATH_CRT4 and n * (p - 1) ** 2 >= big_m:
            continue
        a, b = rand(oracle, n, int(rng.integers(1 << 30)), p), rand(oracle, n, int(rng.integers(1 << 30)), p)
        want = oracle.poly_multiply_naive_u64(a, b, p)
        got = nttx.poly_multiply(a, b, n, p, dtype=np.uint64)
        assert np.array_equal(got, want), (p, n)
        if p < (1 << 31):
            g32 = nttx.poly_multiply(a.astype(np.int32), b.astype(np.int32), n, p, dtype=np.int32)
            assert np.array_equal(g32.astype(np.uint64), want), (p, n, "i32")
        done += 1
    assert done > 200


def test_fuzz_multi_pass_lengths(nttx, oracle):
    """Seeded ragged lengths across the one-, two- and three-pass plans (product lengths 2^11 .. 2^22)
    on the three routes, against the oracle's product (the restatement of main.cc).  Product length
    2^22 is beyond the reference's prime set (2^21 does not divide 1004535809 - 1 twice over), but with
    p = 469762049, itself the first CRT prime, its final "mod p" only sees that prime's residue."""
    rng = np.random.default_rng(77)
    cases = [(469762049, 11, 22), (1004535809, 11, 21), (263882790666241, 11, 21), (281474976710597, 11, 21)]
    for p, lo, hi in cases:
        for k in sorted(set(int(x) for x in rng.integers(lo, hi + 1, size=5)) | {hi}):
            n = int(rng.integers((1 << (k - 2)) + 1, (1 << (k - 1)) + 1))      # product length exactly 2^k
            a, b = rand(oracle, n, 900 + k, p), rand(oracle, n, 950 + k, p)
            want = oracle.poly_multiply_crt(a, b, p)
            got = nttx.poly_multiply(a, b, n, p, dtype=np.uint64)
            assert np.array_equal(got, want), (p, n, k)


def test_device_call_is_graph_capturable(nttx, oracle):
    """After nttx_reserve and one warm-up call (plans and workspace exist), the device-resident product makes no
    allocation and no synchronising call, so a caller can capture it in a HIP graph; replays give the eager result."""
    import torch
    p, L, batch = 469762049, 1 << 16, 4
    n = L // 2
    a = torch.zeros((batch, L), dtype=torch.int32, device="cuda")
    b = torch.zeros_like(a)
    a[:, :n] = torch.from_numpy(rand(oracle, n * batch, 81, p).astype(np.int64).reshape(batch, n)).cuda().to(torch.int32)
    b[:, :n] = torch.from_numpy(rand(oracle, n * batch, 82, p).astype(np.int64).reshape(batch, n)).cuda().to(torch.int32)
    eager, replayed = torch.empty_like(a), torch.zeros_like(a)
    nttx.reserve(L, batch, 1)
    nttx.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), eager.data_ptr(), n, p, batch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        nttx.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), replayed.data_ptr(), n, p, batch, side.cuda_stream)
        torch.cuda.synchronize()
        replayed.zero_()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for i in range(batch):     # one launch group per product, as a latency-bound caller would issue them
                nttx.dev_poly_multiply_u32(a[i].data_ptr(), b[i].data_ptr(), replayed[i].data_ptr(), n, p, 1,
                                           torch.cuda.current_stream().cuda_stream)
    assert int(replayed.abs().max()) == 0          # capture records, it does not run
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(replayed, eager)
    want = oracle.ntt_compute(a[1].cpu().numpy().view(np.uint32), b[1].cpu().numpy().view(np.uint32), p)
    assert np.array_equal(replayed[1].cpu().numpy().view(np.uint32), want)


def test_stream_copy(nttx):
    """The measurement aid nttx_dev_stream_copy (the library's own 16-bytes-per-lane copy kernel) copies exactly."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    for words in (4, 1000 * 4, (1 << 22) + 12):
        src = torch.randint(0, 1 << 31, (words,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
        dst = torch.zeros_like(src)
        nttx.dev_stream_copy(dst.data_ptr(), src.data_ptr(), 4 * words, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(src, dst)
    with pytest.raises(nttx.NttxError):
        nttx.dev_stream_copy(dst.data_ptr(), src.data_ptr(), 12, 0)
