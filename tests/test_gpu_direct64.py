"""GPU parity of the 64-bit-lane direct path (one transform modulo p itself for NTT-friendly
moduli between 2^31 and 2^62) through the C ABI.  Checkers: the oracle's CRT product (the
restatement of main.cc, exact while n*(p-1)^2 stays below the product of its four primes), its
schoolbook product for the moduli beyond that bound, the forced CRT route of the library on
the same inputs, and evaluation of both sides at random points for full-size cases."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P_ID4 = 1337006139375617          # 19 * 2^46 + 1: harness id 4 (lazy class, low word 1)
PRIMES = {
    "lazy_lo1": P_ID4,
    "lazy_lo1_b": 263882790666241,          # 15 * 2^44 + 1
    "lazy_lo1_top": 2251756864012289,       # just below 2^51
    "lazy": 1125899437080577,               # 67108836 * 2^24 + 1
    "lazy_small": 8588886017,               # 8191 * 2^20 + 1, just above 2^31
    "safe_lo1": 4611685941117976577,        # just below 2^62, 2^32 | p-1
    "safe": 2305843009146585089,            # 2^26 | p-1
}


def rand(oracle, n, seed, p):
    return oracle.fill_random(n, seed, p).astype(np.uint64)


def horner(coeffs, x, p):
    acc = 0
    for c in coeffs[::-1]:
        acc = (acc * x + int(c)) % p
    return acc


@pytest.mark.parametrize("name", sorted(PRIMES))
def test_paths_and_small_products(nttx, oracle, name):
    p = PRIMES[name]
    assert nttx.path_for(p, 1000) == nttx.PATH_DIRECT64
    for n in (1, 2, 3, 5, 8, 31, 33, 64, 257, 1000, 1024, 1025, 2048):
        a, b = rand(oracle, n, 7 * n + 1, p), rand(oracle, n, 7 * n + 2, p)
        want = oracle.poly_multiply_naive_u64(a, b, p)
        got = nttx.poly_multiply_direct64(a, b, n, p)
        assert np.array_equal(got, want), (name, n)
        assert np.array_equal(nttx.poly_multiply(a, b, n, p, dtype=np.uint64), want)


@pytest.mark.parametrize("name,n", [("lazy_lo1", 4097), ("lazy", 20000), ("lazy_lo1_b", 65536), ("lazy_small", 70001),
                                    ("lazy_lo1_top", 3000), ("lazy", 3000), ("lazy_small", 4000), ("lazy_lo1", 2500)])
def test_two_pass_lengths_vs_crt_oracle(nttx, oracle, name, n):
    p = PRIMES[name]
    a, b = rand(oracle, n, 11, p), rand(oracle, n, 12, p)
    want = oracle.poly_multiply_crt(a, b, p)
    assert np.array_equal(nttx.poly_multiply_direct64(a, b, n, p), want)


@pytest.mark.parametrize("name,n", [("safe_lo1", 5000), ("safe", 4100), ("safe_lo1", 3000), ("safe", 4096)])
def test_two_pass_lengths_safe_class(nttx, oracle, name, n):
    """Moduli beyond the CRT bound: schoolbook oracle."""
    p = PRIMES[name]
    a, b = rand(oracle, n, 21, p), rand(oracle, n, 22, p)
    assert np.array_equal(nttx.poly_multiply_direct64(a, b, n, p), oracle.poly_multiply_naive_u64(a, b, p))


def test_id4_full_size(nttx, oracle):
    """n = 131072 on the modulus of harness id 4: equal to the oracle, to the library's CRT route and
    to the sha256 of the compiled reference's own output for these inputs."""
    p, n = P_ID4, 131072
    a, b = rand(oracle, n, 1004, p), rand(oracle, n, 2004, p)
    got = nttx.poly_multiply_direct64(a, b, n, p)
    assert np.array_equal(got, oracle.poly_multiply_crt(a, b, p))
    assert np.array_equal(got, nttx.poly_multiply_crt(a, b, n, p))
    c = json.load(open(os.path.join(GOLD, "harness.json")))["cases"]["4_17"]
    assert hashlib.sha256(got.astype("<u8").tobytes()).hexdigest() == c["sha256"]


@pytest.mark.parametrize("name", ("safe_lo1", "safe", "lazy"))
def test_full_size_by_evaluation(nttx, oracle, name):
    """n = 131072 where no exact CPU product is cheap: a(x) b(x) = ab(x) at random points mod p."""
    p, n = PRIMES[name], 131072
    a, b = rand(oracle, n, 31, p), rand(oracle, n, 32, p)
    got = nttx.poly_multiply_direct64(a, b, n, p)
    assert got.max() < p
    for x in (2, 0x123456789ABCDEF % p, p - 5):
        assert horner(a, x, p) * horner(b, x, p) % p == horner(got, x, p)


def test_inputs_not_reduced(nttx, oracle):
    p, n = P_ID4, 700
    a, b = rand(oracle, n, 41, p), rand(oracle, n, 42, p)
    big_a = a + np.uint64(p) * np.uint64(5000)          # still below 2^64
    big_b = b.copy()
    big_b[::3] += np.uint64(p)
    assert np.array_equal(nttx.poly_multiply_direct64(big_a, big_b, n, p), oracle.poly_multiply_crt(a, b, p))


def _sparse_check(nttx, p, log_len, unit_terms):
    """Linear product of a dense random a (n = 2^(log_len-1) coefficients) with a sparse b: the expected
    coefficients are a few shifted, scaled copies of a, computed on the device in 64-bit integers."""
    L, n = 1 << log_len, 1 << (log_len - 1)
    g = torch.Generator(device="cuda")
    g.manual_seed(77 + log_len)
    a = torch.zeros(L, dtype=torch.int64, device="cuda")
    a[:n] = torch.randint(0, p, (n,), generator=g, device="cuda", dtype=torch.int64)
    terms = {0: 3, 1: 1, 12345: 77, n // 2 + 5: 1000, n - 1: 2, (n // 3) | 1: 1}
    if unit_terms:                       # moduli near 2^62: keep every intermediate below 2^63
        terms = {k: 1 for k in terms}
    b = torch.zeros(L, dtype=torch.int64, device="cuda")
    want = torch.zeros(L, dtype=torch.int64, device="cuda")
    for k, v in terms.items():
        b[k] = v
        want[k:k + n] = (want[k:k + n] + a[:n] * v % p) % p
    c = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    for x, y in ((a, b), (b, a)):
        nttx.dev_direct_multiply_u64(x.data_ptr(), y.data_ptr(), c.data_ptr(), n, p, 1, st)
        torch.cuda.synchronize()
        assert torch.equal(c[: 2 * n - 1], want[: 2 * n - 1])


@pytest.mark.parametrize("name,log_len,passes", [("lazy_lo1", 20, 2), ("lazy_lo1", 21, 2), ("lazy_lo1", 22, 2), ("lazy", 22, 2),
                                                 ("lazy_lo1_top", 21, 2), ("lazy_lo1_top", 22, 3), ("safe_lo1", 21, 2),
                                                 ("safe", 22, 2), ("lazy_lo1", 23, 3), ("safe_lo1", 24, 3)])
def test_long_lengths_sparse_factor(nttx, name, log_len, passes):
    """Two-pass plans around rows of 2^11 .. 2^13 points (2^20 .. 2^22) and three-pass plans beyond, every
    arithmetic class (floating point below 1.35e15, lazy integer up to 2^51, Harvey beyond).  A 51-bit modulus of the
    lazy integer class leaves room for 12 unreduced forward stages, not 13, so its 2^22 plan keeps three passes."""
    p = PRIMES[name]
    assert len(nttx.plan_info_u64(1 << log_len, p)) == passes
    _sparse_check(nttx, p, log_len, unit_terms=name.startswith("safe"))


def test_three_pass_length(nttx, oracle):
    """Product length 2^21 against the library's CRT route, the
    longest length that route supports, on a 40-bit modulus (n p^2 must stay below the CRT bound);
    the id 4 modulus at the same length by evaluation."""
    p, n = 1099469684737, 1 << 20              # 262134 * 2^22 + 1
    assert nttx.path_for(p, n) == nttx.PATH_DIRECT64
    a, b = rand(oracle, n, 51, p), rand(oracle, n, 52, p)
    got = nttx.poly_multiply_direct64(a, b, n, p)
    assert np.array_equal(got, nttx.poly_multiply_crt(a, b, n, p))
    p = P_ID4
    a, b = rand(oracle, n, 53, p), rand(oracle, n, 54, p)
    got = nttx.poly_multiply_direct64(a, b, n, p)
    assert got.max() < p
    for x in (3, p - 2):
        assert horner(a, x, p) * horner(b, x, p) % p == horner(got, x, p)


def test_device_resident_batch_and_transform(nttx, oracle):
    p, n, batch = P_ID4, 3000, 5
    L = 8192
    a = np.zeros((batch, L), dtype=np.uint64)
    b = np.zeros((batch, L), dtype=np.uint64)
    for i in range(batch):
        a[i, :n], b[i, :n] = rand(oracle, n, 60 + i, p), rand(oracle, n, 70 + i, p)
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    dc = torch.empty_like(da)
    nttx.reserve_u64(L, batch)
    nttx.dev_direct_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), n, p, batch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = dc.cpu().numpy().view(np.uint64)
    for i in range(batch):
        assert np.array_equal(c[i, : 2 * n - 1], oracle.poly_multiply_crt(a[i, :n], b[i, :n], p))
        assert not c[i, 2 * n - 1:].any()
    assert torch.equal(da.cpu(), torch.from_numpy(a.view(np.int64)))       # inputs untouched
    # transform level: X[bitrev(k)] = sum_j x[j] w^(jk), w = 3^((p-1)/len) (main.cc:156), on 64 points
    T = 64
    x = rand(oracle, T, 80, p)
    dx = torch.from_numpy(x.view(np.int64).copy()).cuda()
    nttx.dev_ntt_u64(dx.data_ptr(), T, p, False, 1, torch.cuda.current_stream().cuda_stream)
    X = dx.cpu().numpy().view(np.uint64)
    w = pow(3, (p - 1) // T, p)
    for k in range(T):
        want = sum(int(x[j]) * pow(w, j * k, p) for j in range(T)) % p
        assert int(X[int(format(k, "06b")[::-1], 2)]) == want


def test_errors(nttx):
    a = np.ones(8, dtype=np.uint64)
    with pytest.raises(nttx.NttxError):
        nttx.poly_multiply_direct64(a, a, 8, 1337006139375617 + 2)      # not prime / no root
    with pytest.raises(nttx.NttxError):
        nttx.poly_multiply_direct64(a, a, 8, 469762049)                  # 32-bit modulus: other path
    assert nttx.path_for((1 << 61) - 1, 8) == nttx.PATH_CRT4               # prime, but 2^4 does not divide p-1


@pytest.mark.parametrize("name,log_len,batch", [("lazy_lo1", 1, 3), ("lazy_lo1", 5, 3), ("lazy_lo1", 10, 3), ("lazy_lo1", 12, 2),
                                                ("lazy_lo1_top", 13, 2), ("lazy", 14, 2), ("lazy_small", 11, 2), ("safe_lo1", 12, 2),
                                                ("safe", 13, 2), ("lazy_lo1", 16, 1), ("safe_lo1", 17, 1), ("lazy_lo1", 21, 1),
                                                ("lazy_lo1", 23, 1)])
def test_negacyclic_product_u64(nttx, oracle, name, log_len, batch):
    """c = a b mod (x^len + 1, p) on 64-bit lanes: lin[i] - lin[i + len] of the linear product, which the library's
    own direct route gives (itself pinned to the oracle above); for small lengths the schoolbook product.  One-pass
    and two-pass plans apply the psi factors inside their kernels, the three-pass length (2^23) in separate passes."""
    p = PRIMES[name]
    L = 1 << log_len
    a = np.stack([rand(oracle, L, 500 + i, p) for i in range(batch)])
    b = np.stack([rand(oracle, L, 600 + i, p) for i in range(batch)])
    a[0, 0] += np.uint64(p)                                  # inputs need not be reduced
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    dc = torch.empty_like(da)
    nttx.dev_negacyclic_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint64)
    a[0, 0] -= np.uint64(p)
    for i in range(batch):
        lin = (oracle.poly_multiply_naive_u64(a[i], b[i], p) if L <= 1024 else nttx.poly_multiply_direct64(a[i], b[i], L, p))
        lin = np.concatenate([lin, np.zeros(1, np.uint64)])
        lo, hi = lin[:L], lin[L:]
        want = np.where(lo >= hi, lo - hi, lo + np.uint64(p) - hi)
        assert np.array_equal(got[i], want), (name, log_len, i)
    assert torch.equal(da.cpu(), torch.from_numpy((a + np.array([[p] + [0] * (L - 1)] + [[0] * L] * (batch - 1), dtype=np.uint64)).view(np.int64)))
    with pytest.raises(nttx.NttxError):                      # 8191 * 2^20 + 1: no 2^21-th root
        nttx.dev_negacyclic_multiply_u64(da.data_ptr(), db.data_ptr(), dc.data_ptr(), 1 << 20, PRIMES["lazy_small"], 1, 0)


FP_EDGE = {"fp_top": 1349999656960001,       # 80466345 * 2^24 + 1: the largest such prime of the floating-point class
           "int_bottom": 1350000109944833}    # 80466372 * 2^24 + 1: the first above it (lazy integer class)


@pytest.mark.parametrize("name", ["fp_top", "int_bottom", "id4"])
@pytest.mark.parametrize("log_len", [10, 12, 13, 16, 18, 21])
def test_floating_point_class_worst_case(nttx, oracle, name, log_len):
    """Every coefficient p-1 (the largest operands every butterfly can see; (p-1)^2 = 1 mod p makes the product
    min(k+1, 2n-1-k)), then random inputs against the CRT oracle, on the largest modulus of the floating-point class,
    the first modulus beyond it and id 4's; product lengths covering one- and two-pass plans; linear and negacyclic."""
    p = FP_EDGE.get(name, P_ID4)
    n = 1 << (log_len - 1)
    a = np.full(n, p - 1, dtype=np.uint64)
    k = np.arange(2 * n - 1, dtype=np.uint64)
    want = np.minimum(k + np.uint64(1), np.uint64(2 * n - 1) - k) % np.uint64(p)
    assert np.array_equal(nttx.poly_multiply_direct64(a, a, n, p), want)
    if log_len <= 18:
        x, y = rand(oracle, n, 900 + log_len, p), rand(oracle, n, 901 + log_len, p)
        assert np.array_equal(nttx.poly_multiply_direct64(x, y, n, p), oracle.poly_multiply_crt(x, y, p))
    if log_len <= 16:                                    # negacyclic: (p-1)^2 x^(i+j) summed with the wrap-around sign
        N = n
        da = torch.from_numpy(np.full((1, N), p - 1, dtype=np.uint64).view(np.int64)).cuda()
        dc = torch.empty_like(da)
        nttx.dev_negacyclic_multiply_u64(da.data_ptr(), da.data_ptr(), dc.data_ptr(), N, p, 1, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        i = np.arange(N, dtype=np.int64)
        neg = ((i + 1) - (N - 1 - i)) % p                # (i+1) terms without wrap, (N-1-i) with sign -1
        assert np.array_equal(dc.cpu().numpy().view(np.uint64)[0], neg.astype(np.uint64))
