#!/usr/bin/env python3
"""Single-rank RCCL contact test for ntt_amd.distributed.TorchComm: process group on backend nccl,
all_to_all / all_gather / barrier / split on device tensors, then the two exchange-step workloads
through TorchComm (with one rank they issue no exchange; the group machinery still runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")

import torch
import torch.distributed as dist
import ntt_amd
from ntt_amd.distributed import CrtSharded, FourStepSharded, HipEngine, TorchComm

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
comm = TorchComm()
x = torch.arange(64, dtype=torch.int32, device=dev).view(1, 8, 8)
assert torch.equal(comm.all_to_all(x), x)
assert torch.equal(comm.all_gather(x[0])[0], x[0])
comm.barrier()
sub = comm.split(lambda r: 0)
assert sub.world == 1 and torch.equal(sub.all_to_all(x), x)
ntt_amd.init(0)
p64 = 1337006139375617
a = torch.randint(0, p64, (2, 1000), device=dev, dtype=torch.int64)
b = torch.randint(0, p64, (2, 1000), device=dev, dtype=torch.int64)
out = CrtSharded(comm, HipEngine()).multiply(a, b, 1000, p64)
ref = torch.empty((2, 1999), dtype=torch.int64, device=dev)
ntt_amd.dev_crt_multiply_u64(a.data_ptr(), b.data_ptr(), ref.data_ptr(), 1000, p64, 2, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert torch.equal(out, ref)
fs = FourStepSharded(comm, HipEngine(), 20, 469762049)
va = torch.randint(0, 469762049, (1 << 20,), device=dev, dtype=torch.int64).to(torch.int32)
vb = torch.randint(0, 469762049, (1 << 20,), device=dev, dtype=torch.int64).to(torch.int32)
xc = fs.multiply(fs.slab_of(va).clone(), fs.slab_of(vb).clone())   # slabs are overwritten; with one rank the slab is the vector itself
rc = torch.empty_like(va)
ntt_amd.dev_cyclic_multiply_u32(va.data_ptr(), vb.data_ptr(), rc.data_ptr(), 1 << 20, 469762049, 1, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert torch.equal(xc.reshape(-1), rc)
dist.barrier()
dist.destroy_process_group()
ntt_amd.shutdown()
print("rccl smoke ok")
