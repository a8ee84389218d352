"""CRT route on resident u64 inputs without per-kernel events: us per product for the batch sizes in argv.
Run once per setting of NTTX_CRT_STREAMS (0: one stream, 1: primes side by side, unset: automatic)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ntt_amd
from oracle import pyoracle as po
P, n = int(os.environ.get("P", 1337006139375617)), 1 << 17
ntt_amd.init(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
st = torch.cuda.current_stream().cuda_stream
O = po.Oracle()
out = {"NTTX_CRT_STREAMS": os.environ.get("NTTX_CRT_STREAMS", "auto"), "p": P, "us_per_mult": {}}
for B in (int(x) for x in sys.argv[1:]):
    a = torch.randint(0, P, (B, n), generator=g, device=dev, dtype=torch.int64)
    b = torch.randint(0, P, (B, n), generator=g, device=dev, dtype=torch.int64)
    c = torch.empty((B, 2 * n - 1), dtype=torch.int64, device=dev)
    step = lambda: ntt_amd.dev_crt_multiply_u64(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P, B, st)
    for _ in range(30): step()
    torch.cuda.synchronize()
    for i in (0, B - 1):
        want = O.poly_multiply_crt(a[i].cpu().numpy().view(np.uint64), b[i].cpu().numpy().view(np.uint64), P)
        assert np.array_equal(c[i].cpu().numpy().view(np.uint64), want), "CRT path differs from the oracle"
    K = max(20, 2000 // B)
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out["us_per_mult"][B] = round(dt / (B * K) * 1e6, 3)
    del a, b, c
print(json.dumps(out))
ntt_amd.shutdown()
