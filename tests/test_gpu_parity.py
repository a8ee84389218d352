"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer work: the bar is bit-exact equality."""
import hashlib
import os

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


@pytest.mark.parametrize("p", (1337006139375617, 263882790666241))
def test_crt_path_small(nttx, oracle, p):
    for n in (1, 2, 5, 64, 1000, 4096):
        a, b = rand(oracle, n, n + 11, p), rand(oracle, n, n + 12, p)
        want = oracle.poly_multiply_crt(a, b, p)
        assert nttx.path_for(p, n) == nttx.PATH_CRT4
        assert np.array_equal(nttx.poly_multiply(a, b, n, p, dtype=np.uint64), want), (p, n)
        if n <= 64:
            assert np.array_equal(want, oracle.poly_multiply_naive_u64(a, b, p))


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
    gold = os.path.join(GOLD, "harness_sha256.txt")
    if os.path.exists(gold):
        lines = dict(l.split() for l in open(gold).read().splitlines() if l.strip())
        assert hashlib.sha256(got.astype("<u8").tobytes()).hexdigest() == lines[str(tid)]


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
    """2^22-point product (three passes) against the oracle, plus a point-evaluation check."""
    p, n = 469762049, 1 << 21
    assert len(nttx.plan_info(1 << 22)) == 3
    a, b = rand(oracle, n, 41, p), rand(oracle, n, 42, p)
    got = nttx.poly_multiply(a, b, n, p, dtype=np.uint64)
    want = oracle.poly_multiply_crt(a, b, p)
    assert np.array_equal(got, want)


def test_inputs_not_reduced(nttx, oracle):
    """Coefficients >= p are legal for the (uint64_t) flavour: main.cc reduces them per prime."""
    p, n = 7340033, 100
    a = rand(oracle, n, 1, 1 << 40)
    b = rand(oracle, n, 2, 1 << 40)
    want = oracle.poly_multiply_crt(a, b, p)
    assert np.array_equal(nttx.poly_multiply(a, b, n, p, dtype=np.uint64), want)


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
