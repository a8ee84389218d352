"""PCIe-inclusive latency of the host-buffer entry points (never the benchmark's `value`)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ntt_amd
from oracle import pyoracle as po
O = po.Oracle()
ntt_amd.init(0)
for p in (469762049, 1337006139375617):
    n = 131072
    a, b = O.fill_random(n, 1, p), O.fill_random(n, 2, p)
    ntt_amd.poly_multiply(a, b, n, p, dtype=np.uint64)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); ntt_amd.poly_multiply(a, b, n, p, dtype=np.uint64); ts.append(time.perf_counter() - t0)
    print(f"poly_multiply_u64 n={n} p={p} path={ntt_amd.path_for(p, n)}: median {sorted(ts)[10]*1e3:.3f} ms, min {min(ts)*1e3:.3f} ms")
ntt_amd.shutdown()
