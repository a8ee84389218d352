"""Negacyclic products of N = 2^k coefficients on 64-bit lanes (resident [batch][N] u64 arrays, p = 19*2^46+1) next to
linear products of product length N (n = N/2 coefficients, same arrays and transform length): k:batch ..."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ntt_amd
P = 1337006139375617
ntt_amd.init(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
st = torch.cuda.current_stream().cuda_stream
for arg in sys.argv[1:]:
    k, B = (int(x) for x in arg.split(":"))
    N = 1 << k
    a = torch.randint(0, P, (B, N), generator=g, device=dev, dtype=torch.int64)
    b = torch.randint(0, P, (B, N), generator=g, device=dev, dtype=torch.int64)
    c = torch.empty_like(a)
    res = {"workload": "negacyclic, 64-bit lanes", "p": P, "log2_len": k, "batch": B}
    for name, fn in (("negacyclic_us", lambda: ntt_amd.dev_negacyclic_multiply_u64(a.data_ptr(), b.data_ptr(), c.data_ptr(), N, P, B, st)),
                     ("linear_same_length_us", lambda: ntt_amd.dev_direct_multiply_u64(a.data_ptr(), b.data_ptr(), c.data_ptr(), N // 2, P, B, st))):   # n = N/2: arrays are [B][2n]
        for _ in range(30): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 40 / B * 1e6
    print(json.dumps(res))
    del a, b, c
ntt_amd.shutdown()
