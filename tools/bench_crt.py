"""Throughput of the CRT path on resident u64 inputs: n = 2^17 coefficients, p = 1337006139375617
(harness id 4; four primes) or the modulus in $P (moduli below 2^36 take three primes), batch of
independent products; for the default modulus also the direct 64-bit-lane path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ntt_amd
from oracle import pyoracle as po
P, n, B = int(os.environ.get("P", 1337006139375617)), 1 << 17, int(os.environ.get("B", 32))
ntt_amd.init(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
a = torch.randint(0, P, (B, n), generator=g, device=dev, dtype=torch.int64)
b = torch.randint(0, P, (B, n), generator=g, device=dev, dtype=torch.int64)
c = torch.empty((B, 2 * n - 1), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
step = lambda: ntt_amd.dev_crt_multiply_u64(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P, B, st)
for _ in range(3): step()
torch.cuda.synchronize()
O = po.Oracle()
want = O.poly_multiply_crt(a[0].cpu().numpy().view(np.uint64), b[0].cpu().numpy().view(np.uint64), P)
assert np.array_equal(c[0].cpu().numpy().view(np.uint64), want), "CRT path differs from the oracle"
ntt_amd.profile_reset(); ntt_amd.profile_enable(True)
K = int(os.environ.get("K", 50))
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ntt_amd.profile_enable(False)
prof = {k: round(v[0] / K, 4) for k, v in ntt_amd.profile_read().items() if v[1]}
L = 2 * n
print(json.dumps({"workload": "CRT route, n=2^17, p=%d" % P, "batch": B, "poly_mults_per_s": B * K / dt,
                  "us_per_mult": dt / (B * K) * 1e6, "ns_per_butterfly": dt / (B * K) / (4 * 3 * (L // 2) * 18) * 1e9,
                  "kernels_ms_per_step": prof, "model_bytes_280L": 280 * L, "model_280L_bytes_x_rate_GBs_NOT_traffic": 280 * L * B * K / dt / 1e9}))
if ntt_amd.path_for(P, n) != ntt_amd.PATH_DIRECT64:
    ntt_amd.shutdown()
    sys.exit(0)

# the same products on the 64-bit-lane direct path (zero-padded [B][2n] arrays, as the 32-bit device call takes)
pa = torch.zeros((B, L), dtype=torch.int64, device=dev); pa[:, :n] = a
pb = torch.zeros((B, L), dtype=torch.int64, device=dev); pb[:, :n] = b
pc = torch.empty_like(pa)
ntt_amd.reserve_u64(L, B)
step64 = lambda: ntt_amd.dev_direct_multiply_u64(pa.data_ptr(), pb.data_ptr(), pc.data_ptr(), n, P, B, st)
K = int(os.environ.get("K", 50))
for _ in range(20): step64()
torch.cuda.synchronize()
assert np.array_equal(pc[0, : 2 * n - 1].cpu().numpy().view(np.uint64), want), "direct 64-bit path differs from the oracle"
ntt_amd.profile_reset(); ntt_amd.profile_enable(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): step64()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ntt_amd.profile_enable(False)
prof = {k: round(v[0] / K, 4) for k, v in ntt_amd.profile_read().items() if v[1]}
print(json.dumps({"workload": "direct 64-bit lanes, n=2^17, p=1337006139375617", "batch": B, "poly_mults_per_s": B * K / dt,
                  "us_per_mult": dt / (B * K) * 1e6, "ns_per_butterfly": dt / (B * K) / (3 * (L // 2) * 18) * 1e9,
                  "kernels_ms_per_step": prof, "moved_bytes_64L": 64 * L, "moved_GBs": 64 * L * B * K / dt / 1e9}))
ntt_amd.shutdown()
