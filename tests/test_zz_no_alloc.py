"""nttx_dev_poly_multiply_u32_static: the device-resident linear product for callers that cannot allocate device memory.
Its transformed copies live in a fixed array inside the library instead of hipMalloc'ed workspace; results must equal
those of nttx_dev_poly_multiply_u32 bit for bit.  Collected last so that a failure here hides no other test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 469762049


def _inputs(torch, oracle, batch, n, length, seed):
    a = np.zeros((batch, length), np.uint32)
    b = np.zeros((batch, length), np.uint32)
    a[:, :n] = oracle.fill_random(n * batch, seed, P).reshape(batch, n)
    b[:, :n] = oracle.fill_random(n * batch, seed + 1, P).reshape(batch, n)
    return a, b, torch.from_numpy(a.view(np.int32)).cuda(), torch.from_numpy(b.view(np.int32)).cuda()


def test_static_scratch_entry_point_matches_allocating_one(nttx, oracle):
    import torch
    st = torch.cuda.current_stream().cuda_stream
    # two calls in a row, the second smaller (fewer products, shorter and ragged): the scratch keeps the first call's
    # contents, and nothing of them may leak into the second result
    cases = [(4, 131072, 1 << 18, 11), (3, 30000, 1 << 16, 21)]
    prepared = []
    for batch, n, length, seed in cases:
        a, b, da, db = _inputs(torch, oracle, batch, n, length, seed)
        want = torch.empty_like(da)
        nttx.dev_poly_multiply_u32(da.data_ptr(), db.data_ptr(), want.data_ptr(), n, P, batch, st)   # also builds the tables
        torch.cuda.synchronize()
        ref = oracle.poly_multiply_crt(a[batch - 1, :n].astype(np.uint64), b[batch - 1, :n].astype(np.uint64), P)
        assert np.array_equal(want[batch - 1].cpu().numpy().view(np.uint32)[: 2 * n - 1].astype(np.uint64), ref)
        prepared.append((batch, n, da, db, want, torch.full_like(da, 5)))
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for batch, n, da, db, want, got in prepared:
        nttx.dev_poly_multiply_u32_static(da.data_ptr(), db.data_ptr(), got.data_ptr(), n, P, batch, st)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before            # no device memory was taken
    for batch, n, da, db, want, got in prepared:
        assert torch.equal(got, want), (batch, n)
    # again in the other order, and on a side stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for batch, n, da, db, want, got in reversed(prepared):
            got.fill_(9)
            nttx.dev_poly_multiply_u32_static(da.data_ptr(), db.data_ptr(), got.data_ptr(), n, P, batch, side.cuda_stream)
    torch.cuda.synchronize()
    for batch, n, da, db, want, got in prepared:
        assert torch.equal(got, want), (batch, n, "side stream")


def test_static_scratch_entry_point_limits(nttx, oracle):
    import torch
    st = torch.cuda.current_stream().cuda_stream
    # a single-pass length uses no scratch at all and works the same way
    a, b, da, db = _inputs(torch, oracle, 5, 1000, 2048, 31)
    want, got = torch.empty_like(da), torch.empty_like(da)
    nttx.dev_poly_multiply_u32(da.data_ptr(), db.data_ptr(), want.data_ptr(), 1000, P, 5, st)
    nttx.dev_poly_multiply_u32_static(da.data_ptr(), db.data_ptr(), got.data_ptr(), 1000, P, 5, st)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # beyond the scratch: 17 products of length 2^18 are 17 * 2^18 > 2^22 words
    big = torch.zeros((17, 1 << 18), dtype=torch.int32, device="cuda")
    with pytest.raises(nttx.NttxError) as e:
        nttx.dev_poly_multiply_u32_static(big.data_ptr(), big.data_ptr(), big.data_ptr(), 131072, P, 17, st)
    assert e.value.code == -2 and "static scratch" in str(e.value)
    # tables that were never built (13631489 = 13 * 2^20 + 1 is a modulus no other test uses): the call may not build them
    with pytest.raises(nttx.NttxError) as e:
        nttx.dev_poly_multiply_u32_static(big.data_ptr(), big.data_ptr(), big.data_ptr(), 200000, 13631489, 1, st)
    assert e.value.code == -5 and "set-up" in str(e.value)
    # the largest admissible call: 16 products of length 2^18 fill the scratch exactly
    a, b, da, db = _inputs(torch, oracle, 16, 131072, 1 << 18, 41)
    want, got = torch.empty_like(da), torch.empty_like(da)
    nttx.dev_poly_multiply_u32(da.data_ptr(), db.data_ptr(), want.data_ptr(), 131072, P, 16, st)
    nttx.dev_poly_multiply_u32_static(da.data_ptr(), db.data_ptr(), got.data_ptr(), 131072, P, 16, st)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_static_scratch_entry_point_under_graph_capture(nttx, oracle):
    """The callers this entry point is for: a HIP graph captures it (nothing in it allocates or synchronises) and
    replays give the eager result."""
    import torch
    n, length, batch = 131072, 1 << 18, 2
    a, b, da, db = _inputs(torch, oracle, batch, n, length, 51)
    eager, replayed = torch.empty_like(da), torch.zeros_like(da)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        # set-up on the stream that will be captured: builds the tables (allocating entry point) and makes that stream
        # the one the library saw last, so that the captured call records no cross-stream wait
        nttx.dev_poly_multiply_u32(da.data_ptr(), db.data_ptr(), eager.data_ptr(), n, P, batch, side.cuda_stream)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            nttx.dev_poly_multiply_u32_static(da.data_ptr(), db.data_ptr(), replayed.data_ptr(), n, P, batch,
                                              torch.cuda.current_stream().cuda_stream)
    assert int(replayed.abs().max()) == 0          # capture records, it does not run
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(replayed, eager)
    want = oracle.poly_multiply_crt(a[1, :n].astype(np.uint64), b[1, :n].astype(np.uint64), P)
    assert np.array_equal(replayed[1].cpu().numpy().view(np.uint32)[: 2 * n - 1].astype(np.uint64), want)
