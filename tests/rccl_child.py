"""One rank of an RCCL ("nccl") process group on the single GPU of the test box, driving the exchange code of
ntt_amd.distributed with force_exchange=True: the sliced-residue send buffer, all_to_all_single (blocking and
async_op=True + wait), the pieces-layout row products and the way back all run over RCCL on tensors libnttx.so
wrote on the caller's stream.  Results against the oracle.  Started by tests/test_gpu_rccl.py in a fresh process
(WORLD_SIZE = 1, rendezvous on 127.0.0.1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P32, P51 = 469762049, 1337006139375617


def main():
    import torch
    import torch.distributed as dist
    import ntt_amd
    from ntt_amd.distributed import CrtSharded, FourStepSharded, HipEngine, TorchComm
    from oracle import pyoracle
    O = pyoracle.Oracle()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    probe = torch.ones(1, device=dev)
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    assert int(probe.item()) == dist.get_world_size() == 1
    ntt_amd.init(0)
    try:
        comm, eng = TorchComm(), HipEngine()
        # -- four-prime CRT products: sliced residues -> all-to-all -> recombination on the slice -> all-gather
        sh = CrtSharded(comm, eng, force_exchange=True)
        for n, nb in ((1 << 17, 3), (40000, 2), (3000, 2), (1 << 20, 1)):     # sliced send buffer / permuting-copy fallback
            if n * (P51 - 1) ** 2 >= 469762049 * 998244353 * 1004535809 * 1224736769:
                p = 263882790666241
            else:
                p = P51
            a = np.stack([O.fill_random(n, 10 + i, p) for i in range(nb)])
            b = np.stack([O.fill_random(n, 20 + i, p) for i in range(nb)])
            da, db = torch.from_numpy(a.view(np.int64)).to(dev), torch.from_numpy(b.view(np.int64)).to(dev)
            for rep in range(2):     # twice in a row: the second call's kernels queue behind the first call's exchange
                out = sh.multiply(da, db, n, p, gather=True)
            got = out.cpu().numpy().view(np.uint64)
            for i in range(nb):
                assert np.array_equal(got[i], O.poly_multiply_crt(a[i], b[i], p)), ("crt", n, i)
            part = sh.multiply(da, db, n, p, gather=False)       # the un-gathered slice (here: every index)
            assert np.array_equal(part.cpu().numpy().view(np.uint64)[:, : 2 * n - 1], got)
        print("rccl crt ok", flush=True)
        # -- one long product split by columns: async exchange of a during b's column pass, pieces-layout rows
        for log_len, log_t, linear in ((22, 6, True), (24, 9, True), (23, 7, False), (20, 9, True), (26, 9, True)):
            L = 1 << log_len
            a, b = O.fill_random(L, 41, P32).astype(np.uint32), O.fill_random(L, 42, P32).astype(np.uint32)
            if linear:
                a[L // 2:] = 0
                b[L // 2:] = 0
            fs = FourStepSharded(comm, eng, log_len, P32, log_t=log_t, force_exchange=True)
            da, db = torch.from_numpy(a.view(np.int32)).to(dev), torch.from_numpy(b.view(np.int32)).to(dev)
            xa, xb = fs.slab_of(da).clone(), fs.slab_of(db).clone()     # (with one rank slab_of returns a view; multiply overwrites its slabs)
            xc = fs.multiply(xa, xb, zero_hi=linear)
            got = xc.cpu().numpy().view(np.uint32).reshape(L)
            if log_len <= 24:
                assert np.array_equal(got, O.ntt_compute(a, b, P32)), ("fourstep", log_len, log_t)
            else:       # beyond what the CPU oracle finishes in seconds: the single-GPU plan (itself pinned to the oracle)
                ref = torch.empty_like(da)
                ntt_amd.dev_cyclic_multiply_u32(da.data_ptr(), db.data_ptr(), ref.data_ptr(), L, P32, 1,
                                                torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                assert np.array_equal(got, ref.cpu().numpy().view(np.uint32)), ("fourstep", log_len, log_t)
        print("rccl fourstep ok", flush=True)
    finally:
        ntt_amd.shutdown()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
