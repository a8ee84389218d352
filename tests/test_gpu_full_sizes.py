"""GPU parity at the benchmark configurations' full sizes, where the CPU oracle would take too long for
every element: size-independent properties over ALL outputs plus the oracle on sampled products."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 469762049


def _rand_dev(torch, shape, seed, p=P):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randint(0, p, shape, generator=g, device="cuda", dtype=torch.int64)


def _eval_mod(torch, x, point, p=P):
    """sum_i x[..., i] * point^i mod p for point in {1, -1}: row sums stay below 2^63 for rows of up to 2^34 / p words."""
    if point == 1:
        return x.sum(dim=-1) % p
    return (x[..., 0::2].sum(dim=-1) - x[..., 1::2].sum(dim=-1)) % p


@pytest.mark.parametrize("log_len,batch", [(16, 4096), (17, 2048), (18, 1024)])
def test_batched_products_full_size(nttx, oracle, log_len, batch):
    """4096 independent products of length 2^16 (and the 2^17 / 2^18 variants) in one call: for EVERY product
    c(1) = a(1) b(1) and c(-1) = a(-1) b(-1) mod p (checksums of all outputs), every output is canonical, the
    padding of the inputs is untouched, and sampled products equal the oracle's."""
    import torch
    L, n = 1 << log_len, 1 << (log_len - 1)
    a = torch.zeros((batch, L), dtype=torch.int64, device="cuda")
    b = torch.zeros((batch, L), dtype=torch.int64, device="cuda")
    a[:, :n] = _rand_dev(torch, (batch, n), 11)
    b[:, :n] = _rand_dev(torch, (batch, n), 12)
    a32, b32 = a.to(torch.int32), b.to(torch.int32)
    c32 = torch.empty_like(a32)
    nttx.dev_poly_multiply_u32(a32.data_ptr(), b32.data_ptr(), c32.data_ptr(), n, P, batch, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = c32.to(torch.int64)
    assert int(c.min()) >= 0 and int(c.max()) < P
    for point in (1, -1):
        want = (_eval_mod(torch, a, point) * _eval_mod(torch, b, point)) % P
        assert torch.equal(_eval_mod(torch, c, point), want), point
    assert torch.equal(a32, a.to(torch.int32)) and int(a32[:, n:].abs().max()) == 0
    for i in (0, 1, batch // 2, batch - 1):
        ah, bh = a32[i].cpu().numpy().view(np.uint32), b32[i].cpu().numpy().view(np.uint32)
        assert np.array_equal(c32[i].cpu().numpy().view(np.uint32), oracle.ntt_compute(ah, bh, P)), i


@pytest.mark.parametrize("log_len", [24, 25, 26])
def test_long_products_full_size(nttx, log_len):
    """One cyclic product of length 2^24 .. 2^26 (three passes): a dense random vector times a sparse one, so the
    expected product is a handful of rotated, scaled copies of the dense vector, computed on the device with
    64-bit integer arithmetic; both operand orders."""
    import torch
    L = 1 << log_len
    a = _rand_dev(torch, (L,), 21)
    terms = {0: 3, 1: P - 1, 12345: 77, L // 2 + 5: 123456789, L - 1: 2, (L // 3) | 1: P // 2}
    b = torch.zeros(L, dtype=torch.int64, device="cuda")
    want = torch.zeros(L, dtype=torch.int64, device="cuda")
    for k, v in terms.items():
        b[k] = v
        want = (want + torch.roll(a, k) * v) % P
    a32, b32 = a.to(torch.int32), b.to(torch.int32)
    c32 = torch.empty_like(a32)
    st = torch.cuda.current_stream().cuda_stream
    assert len(nttx.plan_info(L)) == 3
    for x, y in ((a32, b32), (b32, a32)):
        nttx.dev_cyclic_multiply_u32(x.data_ptr(), y.data_ptr(), c32.data_ptr(), L, P, 1, st)
        torch.cuda.synchronize()
        assert torch.equal(c32.to(torch.int64), want)


def test_big_modulus_products_full_size(nttx, oracle):
    """64 products of n = 2^17 coefficients modulo id 4's 51-bit prime: the four-prime CRT route and the
    64-bit-lane direct route (independent kernels and arithmetic) agree on every coefficient, and sampled
    products equal the oracle's (main.cc's path)."""
    import torch
    p, n, batch = 1337006139375617, 1 << 17, 64
    L = 2 * n
    a, b = _rand_dev(torch, (batch, n), 31, p), _rand_dev(torch, (batch, n), 32, p)
    st = torch.cuda.current_stream().cuda_stream
    c_crt = torch.empty((batch, 2 * n - 1), dtype=torch.int64, device="cuda")
    nttx.dev_crt_multiply_u64(a.data_ptr(), b.data_ptr(), c_crt.data_ptr(), n, p, batch, st)
    pa = torch.zeros((batch, L), dtype=torch.int64, device="cuda")
    pb = torch.zeros((batch, L), dtype=torch.int64, device="cuda")
    pa[:, :n], pb[:, :n] = a, b
    pc = torch.empty_like(pa)
    nttx.dev_direct_multiply_u64(pa.data_ptr(), pb.data_ptr(), pc.data_ptr(), n, p, batch, st)
    torch.cuda.synchronize()
    assert torch.equal(pc[:, : 2 * n - 1], c_crt)
    assert int(c_crt.min()) >= 0 and int(c_crt.max()) < p
    for i in (0, batch - 1):
        want = oracle.poly_multiply_crt(a[i].cpu().numpy().view(np.uint64), b[i].cpu().numpy().view(np.uint64), p)
        assert np.array_equal(c_crt[i].cpu().numpy().view(np.uint64), want), i


@pytest.mark.parametrize("p,log_len", [(469762049, 21), (469762049, 22), (998244353, 23), (469762049, 24)])
def test_long_transform_roundtrip_and_oracle(nttx, oracle, p, log_len):
    """Single transforms at two-pass (long rows) and three-pass lengths through the transform API: forward then
    inverse is len * x, the forward transform of a sparse vector matches its defining sum at sampled outputs, and
    at 2^21 the whole output equals ntt_iter_barrett."""
    L = 1 << log_len
    x = oracle.fill_random(L, 90 + log_len, p).astype(np.uint32)
    fx = nttx.ntt(x, p, False)
    back = nttx.ntt(fx, p, True)
    assert np.array_equal(back.astype(np.uint64), x.astype(np.uint64) * np.uint64(L % p) % np.uint64(p))
    if log_len == 21:
        assert np.array_equal(fx, oracle.ntt_iter_barrett(x, p, False))
    sp = np.zeros(L, np.uint32)
    terms = {0: 5, 3: 7, L // 2 + 1: 11, L - 1: 13}
    for k, v in terms.items():
        sp[k] = v
    fs = nttx.ntt(sp, p, False)
    w = pow(3, (p - 1) // L, p)                      # main.cc:156
    for out in (0, 1, 2, L // 3, L - 1):
        want = sum(v * pow(w, (k * out) % L, p) for k, v in terms.items()) % p
        assert int(fs[out]) == want, (log_len, out)
