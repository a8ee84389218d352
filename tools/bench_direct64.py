"""64-bit-lane direct path on resident inputs: batches of linear products of n = 2^(k-1) coefficients
(product length 2^k) modulo 19*2^46+1, one JSON line per length.  usage: bench_direct64.py k:batch ..."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ntt_amd
P = int(os.environ.get("P", 1337006139375617))
ntt_amd.init(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
st = torch.cuda.current_stream().cuda_stream
for arg in sys.argv[1:]:
    k, B = (int(x) for x in arg.split(":"))
    L, n = 1 << k, 1 << (k - 1)
    a = torch.zeros((B, L), dtype=torch.int64, device=dev); b = torch.zeros((B, L), dtype=torch.int64, device=dev)
    a[:, :n] = torch.randint(0, P, (B, n), generator=g, device=dev, dtype=torch.int64)
    b[:, :n] = torch.randint(0, P, (B, n), generator=g, device=dev, dtype=torch.int64)
    c = torch.empty_like(a)
    ntt_amd.reserve_u64(L, B)
    step = lambda: ntt_amd.dev_direct_multiply_u64(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P, B, st)
    for _ in range(30): step()
    torch.cuda.synchronize()
    ntt_amd.profile_reset(); ntt_amd.profile_enable(True)
    K = int(os.environ.get("STEPS", 40))     # (STEPS: long runs for power traces)
    t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ntt_amd.profile_enable(False)
    prof = {kk: round(v[0] / K, 4) for kk, v in ntt_amd.profile_read().items() if v[1]}
    print(json.dumps({"workload": "direct 64-bit lanes", "p": P, "log2_len": k, "batch": B, "us_per_mult": dt / (B * K) * 1e6,
                      "ps_per_point": dt / (B * K) / L * 1e12, "kernels_ms_per_step": prof}))
    del a, b, c
ntt_amd.shutdown()
