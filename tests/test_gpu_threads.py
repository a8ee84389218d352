"""Several host threads inside the C ABI at once (ctypes releases the interpreter lock for the duration of a call): host-buffer
products of every route and device-resident products on a stream per thread, all against the CPU oracle, bit-exact.  The
reference's call is synchronous and single-threaded (main.cc:277-341); the library serialises calls on a context with a
mutex and orders the shared workspace between streams with an event, which is what this exercises."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P32 = (469762049, 998244353, 1224736769, 7340033)
P64 = 1337006139375617            # 19 * 2^46 + 1: the direct 64-bit route
PCRT = (1000000007, 281474976710597)


def _cases(oracle):
    """(kind, inputs, expected) triples, expected values from the oracle on this thread."""
    out = []
    for i, n in enumerate((1, 5, 64, 1000, 4097, 20000, 70000)):
        p = P32[i % len(P32)]
        a, b = oracle.fill_random(n, 11 * i + 1, p), oracle.fill_random(n, 11 * i + 2, p)
        out.append(("host32", (a, b, n, p), oracle.poly_multiply_crt(a, b, p)))
    for i, n in enumerate((3, 700, 9000, 40000)):
        p = PCRT[i % 2]
        a, b = oracle.fill_random(n, 31 * i + 1, p), oracle.fill_random(n, 31 * i + 2, p)
        out.append(("crt", (a, b, n, p), oracle.poly_multiply_crt(a, b, p)))
    for i, n in enumerate((2, 513, 30000)):
        a, b = oracle.fill_random(n, 41 * i + 1, P64), oracle.fill_random(n, 41 * i + 2, P64)
        out.append(("host64", (a, b, n, P64), oracle.poly_multiply_crt(a, b, P64)))
    for i, (k, batch) in enumerate(((8, 9), (12, 5), (15, 3), (17, 2))):
        p = P32[i % 3]
        L = 1 << k
        a = oracle.fill_random(L * batch, 51 * i + 1, p).astype(np.uint32).reshape(batch, L)
        b = oracle.fill_random(L * batch, 51 * i + 2, p).astype(np.uint32).reshape(batch, L)
        want = np.stack([oracle.ntt_compute(a[r], b[r], p) for r in range(batch)])
        out.append(("dev32", (a, b, L, p, batch), want))
    return out


def test_concurrent_calls(nttx, oracle):
    import torch
    cases = _cases(oracle)
    nthreads, rounds = 4, 6
    # device-resident operands are staged once, on the default stream, before any thread starts
    staged = {}
    for idx, (kind, args, _) in enumerate(cases):
        if kind == "dev32":
            a, b, L, p, batch = args
            for t in range(nthreads):
                da = torch.from_numpy(a.view(np.int32)).cuda()
                db = torch.from_numpy(b.view(np.int32)).cuda()
                staged[(idx, t)] = (da, db, torch.empty_like(da))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    errors = []
    start = threading.Barrier(nthreads)

    def work(t):
        try:
            rng = np.random.default_rng(1000 + t)
            start.wait()
            for _ in range(rounds):
                for idx in rng.permutation(len(cases)):
                    kind, args, want = cases[idx]
                    if kind == "host32":
                        a, b, n, p = args
                        got = nttx.poly_multiply(a, b, n, p, dtype=np.uint64)
                    elif kind == "crt":
                        a, b, n, p = args
                        got = nttx.poly_multiply_crt(a, b, n, p)
                    elif kind == "host64":
                        a, b, n, p = args
                        got = nttx.poly_multiply(a, b, n, p, dtype=np.uint64)
                    else:
                        a, b, L, p, batch = args
                        da, db, dc = staged[(idx, t)]
                        dc.fill_(7)                                   # (default stream of this thread's calls: ordered below)
                        torch.cuda.current_stream().synchronize()
                        nttx.dev_cyclic_multiply_u32(da.data_ptr(), db.data_ptr(), dc.data_ptr(), L, p, batch, streams[t].cuda_stream)
                        streams[t].synchronize()
                        got = dc.cpu().numpy().view(np.uint32)
                    if not np.array_equal(got, want):
                        errors.append((t, kind, args[2:], "mismatch"))
                        return
        except BaseException as e:                                    # noqa: BLE001 (reported by the main thread)
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=150)
    assert not any(th.is_alive() for th in threads), "a thread did not finish"
    assert not errors, errors
