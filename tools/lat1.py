#!/usr/bin/env python3
"""One resident product at a time (batch 1): wall time per product over back-to-back calls.
usage: tools/lat1.py [log2_len] [calls]   (run under `rocprofv3 --kernel-trace` to see the kernels and the gaps)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ntt_amd  # noqa: E402

logL = int(sys.argv[1]) if len(sys.argv) > 1 else 18
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
P = 469762049
L, n = 1 << logL, 1 << (logL - 1)
ntt_amd.init(0)
a = torch.zeros((1, L), dtype=torch.int32, device="cuda")
b = torch.zeros_like(a)
a[:, :n] = torch.randint(0, P, (1, n), device="cuda", dtype=torch.int64).to(torch.int32)
b[:, :n] = torch.randint(0, P, (1, n), device="cuda", dtype=torch.int64).to(torch.int32)
c = torch.empty_like(a)
st = torch.cuda.current_stream().cuda_stream
for _ in range(100):
    ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P, 1, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P, 1, st)
torch.cuda.synchronize()
print(f"log2_len {logL}: {(time.perf_counter() - t0) / calls * 1e6:.2f} us per product (batch 1, {calls} calls back to back)")
ntt_amd.shutdown()
