"""One process, a LIST of devices (nttx_init_devices): run by tests/test_gpu_multidev.py in a fresh process.
argv[1] = how many contexts to open on GPU 0 (more than one needs NTTX_REHEARSE_DEVICES=1 in the environment: the
several-device path, host threads and dealing included, on the one GPU of the test box).  Results against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P = 469762049


def main(nctx):
    import ntt_amd
    from oracle import pyoracle
    O = pyoracle.Oracle()
    assert ntt_amd.device_count() == 0
    ntt_amd.init_devices([0] * nctx)
    try:
        assert ntt_amd.device_count() == nctx
        ntt_amd.init_devices([0] * nctx)                  # the same list again: no-op
        if nctx == 1:
            ntt_amd.init(0)                               # ... and the one-device form of the same list
        for bad in ([0, 0, 0, 0, 0] if nctx != 5 else [0], [7]):
            try:
                ntt_amd.init_devices(bad)
                raise AssertionError("a different device list was accepted without a shutdown")
            except ntt_amd.NttxError:
                pass
        ntt_amd.set_device(0)
        try:
            ntt_amd.set_device(3)
            raise AssertionError("a device outside the list was accepted")
        except ntt_amd.NttxError:
            pass
        for n, batch in ((4, 1), (1000, 7), (5000, 2), (40000, 5), (131072, 3)):
            a = O.fill_random(n * batch, n + 1, P).astype(np.uint32).reshape(batch, n)
            b = O.fill_random(n * batch, n + 2, P).astype(np.uint32).reshape(batch, n)
            got = ntt_amd.poly_multiply_batch(a, b, P)
            L = 1
            while L < 2 * n:
                L *= 2
            for i in range(batch):
                pa, pb = np.zeros(L, np.uint32), np.zeros(L, np.uint32)
                pa[:n], pb[:n] = a[i], b[i]
                assert np.array_equal(got[i], O.ntt_compute(pa, pb, P)[: 2 * n - 1]), (n, batch, i)
        # the other entry points act on the current device's context
        x, y = O.fill_random(3000, 5, P), O.fill_random(3000, 6, P)
        assert np.array_equal(ntt_amd.poly_multiply(x, y, 3000, P, dtype=np.uint64), O.poly_multiply_crt(x, y, P))
        print(f"multidev ok {nctx}", flush=True)
    finally:
        ntt_amd.shutdown()
    assert ntt_amd.device_count() == 0


if __name__ == "__main__":
    main(int(sys.argv[1]))
