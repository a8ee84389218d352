"""GPU parity of the multi-process building blocks (slab column pass, per-prime residues,
slice recombination) against the oracle-backed CPU engine, and of the two exchange-step
workloads run with several in-process ranks on the one GPU of the test box.  All calls go
through the C ABI (ntt_amd.distributed.HipEngine -> ntt_amd.api -> libnttx.so)."""
import numpy as np
import pytest
import torch

from tests.cpu_engine import CRT_PRIMES, OracleEngine

pytestmark = pytest.mark.gpu

P64 = 1337006139375617     # 19 * 2^46 + 1, the modulus of harness id 4


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("p,log_len,log_t,col0,cols,zero_hi", [
    (469762049, 12, 6, 0, 64, False),       # the whole matrix as one slab
    (469762049, 13, 6, 32, 32, False),
    (998244353, 14, 7, 64, 16, True),
    (1224736769, 15, 8, 96, 32, True),
    (1004535809, 16, 9, 0, 64, False),
    (469762049, 17, 9, 112, 16, True),
])
def test_colpass_matches_cpu_engine(nttx, oracle, p, log_len, log_t, col0, cols, zero_hi):
    from ntt_amd.distributed import HipEngine
    T = 1 << log_t
    x = oracle.fill_random(T * cols, 5, p).astype(np.uint32).reshape(T, cols)
    if zero_hi:
        x[T // 2:] = 0
    want = torch.from_numpy(x.view(np.int32).copy())
    OracleEngine(oracle).colpass(want, log_len, log_t, col0, p, False, zero_hi)
    got = _dev(x.view(np.int32))
    HipEngine().colpass(got, log_len, log_t, col0, p, False, zero_hi)
    assert torch.equal(got.cpu(), want)
    # inverse pass on the forward output gives the input back (the slab factors and 1/T cancel)
    back = got.clone()
    HipEngine().colpass(back, log_len, log_t, col0, p, True)
    assert np.array_equal(back.cpu().numpy().view(np.uint32), x)
    want_inv = want.clone()
    OracleEngine(oracle).colpass(want_inv, log_len, log_t, col0, p, True)
    assert np.array_equal(want_inv.numpy().view(np.uint32), x)


def test_colpass_rejects_bad_slabs(nttx):
    from ntt_amd.distributed import HipEngine
    e = HipEngine()
    x = torch.zeros((64, 24), dtype=torch.int32, device="cuda")
    with pytest.raises(nttx.NttxError):
        e.colpass(x, 12, 6, 0, 469762049, False)          # 24 columns: not a power of two
    x = torch.zeros((64, 32), dtype=torch.int32, device="cuda")
    with pytest.raises(nttx.NttxError):
        e.colpass(x, 12, 6, 48, 469762049, False)         # not aligned to its own width
    with pytest.raises(nttx.NttxError):
        e.colpass(x, 12, 6, 64, 469762049, False)         # beyond the S = 64 columns
    with pytest.raises(nttx.NttxError):
        e.colpass(x, 12, 6, 0, 7340033 * 3, False)        # not a prime with the needed root


@pytest.mark.parametrize("mask", (1, 2, 4, 8, 5, 10, 15))
def test_crt_residues_and_recombine(nttx, oracle, mask):
    from ntt_amd.distributed import HipEngine
    n, batch = 300, 3
    a = np.stack([oracle.fill_random(n, 20 + i, P64) for i in range(batch)]).astype(np.uint64)
    b = np.stack([oracle.fill_random(n, 30 + i, P64) for i in range(batch)]).astype(np.uint64)
    ta, tb = torch.from_numpy(a.view(np.int64)), torch.from_numpy(b.view(np.int64))
    want = OracleEngine(oracle).crt_residues(ta, tb, n, mask)
    got = HipEngine().crt_residues(ta.cuda(), tb.cuda(), n, mask)
    assert torch.equal(got.cpu(), want)
    assert [nttx.crt_prime(k) for k in range(4)] == list(CRT_PRIMES)
    if mask == 15:
        out = HipEngine().crt_recombine([got[k].contiguous() for k in range(4)], P64)
        full = out.cpu().numpy().view(np.uint64)
        for i in range(batch):
            assert np.array_equal(full[i, : 2 * n - 1], oracle.poly_multiply_crt(a[i], b[i], P64))
        # a slice of the indices recombines to the same values
        sl = [got[k][:, 128:384].contiguous() for k in range(4)]
        part = HipEngine().crt_recombine(sl, P64).cpu().numpy().view(np.uint64)
        assert np.array_equal(part, full[:, 128:384])


@pytest.mark.parametrize("world", (1, 2, 4, 8))
def test_crt_sharded_in_process_ranks(nttx, oracle, world):
    from ntt_amd.distributed import CrtSharded, HipEngine, ThreadComm
    n, batch = 1000, 4
    a = np.stack([oracle.fill_random(n, 120 + i, P64) for i in range(batch)]).astype(np.uint64)
    b = np.stack([oracle.fill_random(n, 130 + i, P64) for i in range(batch)]).astype(np.uint64)
    want = np.stack([oracle.poly_multiply_crt(a[i], b[i], P64) for i in range(batch)])

    def body(comm):
        torch.cuda.set_device(0)
        sh = CrtSharded(comm, HipEngine())
        lo, hi = sh.batch_range(batch)
        out = sh.multiply(_dev(a[lo:hi].view(np.int64)), _dev(b[lo:hi].view(np.int64)), n, P64)
        return lo, hi, out.cpu().numpy().view(np.uint64)

    for lo, hi, out in ThreadComm.run(world, body):
        assert np.array_equal(out, want[lo:hi])


@pytest.mark.parametrize("p,log_len,log_t,world,linear", [
    (469762049, 14, 6, 2, True),
    (998244353, 16, 8, 4, False),
    (1224736769, 17, 9, 8, True),
    (469762049, 16, 7, 1, True),
])
def test_fourstep_sharded_vs_oracle(nttx, oracle, p, log_len, log_t, world, linear):
    from ntt_amd.distributed import FourStepSharded, HipEngine, ThreadComm
    L = 1 << log_len
    a, b = oracle.fill_random(L, 41, p).astype(np.uint32), oracle.fill_random(L, 42, p).astype(np.uint32)
    if linear:
        a[L // 2:] = 0
        b[L // 2:] = 0
    want = oracle.ntt_compute(a, b, p)

    def body(comm):
        torch.cuda.set_device(0)
        fs = FourStepSharded(comm, HipEngine(), log_len, p, log_t=log_t)
        xa, xb = fs.slab_of(_dev(a.view(np.int32))), fs.slab_of(_dev(b.view(np.int32)))
        return fs.col0, fs.cols, fs.multiply(xa, xb, zero_hi=linear).cpu().numpy().view(np.uint32)

    full = np.zeros(L, dtype=np.uint32)
    for col0, cols, slab in ThreadComm.run(world, body):
        full.reshape(1 << log_t, -1)[:, col0: col0 + cols] = slab
    assert np.array_equal(full, want)


@pytest.mark.parametrize("log_len,world", [(22, 4), (24, 2), (25, 4)])
def test_fourstep_sharded_large_vs_single_gpu(nttx, oracle, log_len, world):
    """Long products over in-process ranks equal the single-GPU three-pass product bit for bit
    (2^24 / 2 ranks and 2^25 / 4 ranks have slabs of 2^14 columns: the split-table column pass)."""
    from ntt_amd.distributed import FourStepSharded, HipEngine, ThreadComm
    p = 469762049
    L = 1 << log_len
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    a = torch.randint(0, p, (L,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    b = torch.randint(0, p, (L,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    a[L // 2:] = 0
    b[L // 2:] = 0
    ref = torch.empty_like(a)
    nttx.dev_cyclic_multiply_u32(a.data_ptr(), b.data_ptr(), ref.data_ptr(), L, p, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    full = torch.zeros_like(a)

    def body(comm):
        torch.cuda.set_device(0)
        fs = FourStepSharded(comm, HipEngine(), log_len, p)
        xc = fs.multiply(fs.slab_of(a), fs.slab_of(b), zero_hi=True)
        fs.place_slab(full, xc)
        torch.cuda.synchronize()
        return None

    ThreadComm.run(world, body)
    assert torch.equal(full, ref)
    # spot-check against the definition at a few indices (exact integer arithmetic on the host)
    ah, bh = a.cpu().numpy().view(np.uint32).astype(np.uint64), b.cpu().numpy().view(np.uint32).astype(np.uint64)
    fh = full.cpu().numpy().view(np.uint32)
    for i in (0, 1, 12345, L // 2 - 1, L // 2, L - 2):
        lo, hi = max(0, i - (L // 2 - 1)), min(i, L // 2 - 1)
        terms = ah[lo: hi + 1] * bh[i - hi: i - lo + 1][::-1] % np.uint64(p)     # each product < 2^58
        assert int(terms.sum(dtype=np.uint64)) % p == int(fh[i])                # at most 2^21 terms < 2^29


@pytest.mark.parametrize("p,log_len,pieces,rows", [(469762049, 15, 2, 3), (469762049, 15, 64, 2), (998244353, 16, 8, 3),
                                                   (1224736769, 17, 4, 2), (469762049, 17, 256, 1), (1004535809, 15, 1, 2)])
def test_cyclic_multiply_in_pieces_layout(nttx, oracle, p, log_len, pieces, rows):
    """nttx_dev_cyclic_multiply_pieces_u32: rows held as [pieces][rows][len / pieces] (what an all-to-all between
    processes leaves), products returned in the same layout; against the oracle's ntt_compute on the assembled rows."""
    import torch
    L = 1 << log_len
    part = L // pieces
    a = np.stack([oracle.fill_random(L, 700 + r, p) for r in range(rows)]).astype(np.uint32)
    b = np.stack([oracle.fill_random(L, 800 + r, p) for r in range(rows)]).astype(np.uint32)
    to_pieces = lambda x: np.ascontiguousarray(x.reshape(rows, pieces, part).transpose(1, 0, 2))
    da = torch.from_numpy(to_pieces(a).view(np.int32)).cuda()
    db = torch.from_numpy(to_pieces(b).view(np.int32)).cuda()
    dc = torch.full_like(da, 3)
    nttx.dev_cyclic_multiply_pieces_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, pieces, rows, p, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dc.cpu().numpy().view(np.uint32).transpose(1, 0, 2).reshape(rows, L)
    for r in range(rows):
        assert np.array_equal(got[r], oracle.ntt_compute(a[r], b[r], p)), (p, log_len, pieces, r)
    assert np.array_equal(da.cpu().numpy().view(np.uint32), to_pieces(a))          # inputs untouched
    with pytest.raises(nttx.NttxError):                                            # single-pass length: no such variant
        nttx.dev_cyclic_multiply_pieces_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), 1 << 12, 2, 1, p, 0)


@pytest.mark.parametrize("n,mask,slices,batch", [(131072, 0b0101, 2, 2), (131072, 0b1000, 4, 3), (20000, 0b1111, 4, 2), (300000, 0b0011, 2, 1)])
def test_crt_residues_sliced_layout(nttx, oracle, n, mask, slices, batch):
    """nttx_dev_crt_residues_sliced_u64 writes [slice][prime][batch][len / slices], the send buffer of the exchange
    between the processes that share the primes: the same words as the ordinary layout, sliced."""
    from ntt_amd.distributed import HipEngine
    e = HipEngine()
    a = np.stack([oracle.fill_random(n, 910 + r, P64) for r in range(batch)])
    b = np.stack([oracle.fill_random(n, 920 + r, P64) for r in range(batch)])
    da, db = _dev(a.view(np.int64)), _dev(b.view(np.int64))
    plain = e.crt_residues(da, db, n, mask)                    # [primes][batch][len], checked against the oracle elsewhere
    sliced = e.crt_residues_sliced(da, db, n, mask, slices)
    torch.cuda.synchronize()
    nsel, _, length = plain.shape
    want = plain.view(nsel, batch, slices, length // slices).permute(2, 0, 1, 3)
    assert torch.equal(sliced, want.contiguous())
    k = [i for i in range(4) if mask >> i & 1][0]
    m = CRT_PRIMES[k]
    pa, pb = np.zeros(length, np.uint32), np.zeros(length, np.uint32)
    pa[:n], pb[:n] = (a[0] % np.uint64(m)).astype(np.uint32), (b[0] % np.uint64(m)).astype(np.uint32)
    got = sliced[:, 0, 0].reshape(-1).cpu().numpy().view(np.uint32)
    assert np.array_equal(got, oracle.ntt_compute(pa, pb, m))
