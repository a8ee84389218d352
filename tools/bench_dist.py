#!/usr/bin/env python3
"""Secondary workloads (the headline one is bench.py): run alone for one GPU or under
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_dist.py ...`.

  --workload crt       4-prime CRT products of n = 2^17 u64 coefficients, primes dealt to ranks,
                       one all-to-all of residue slices per step (ntt_amd.distributed.CrtSharded)
  --workload batch     4096 independent products of length 2^16 / 2^17, batch dealt to ranks, no collective
  --workload nega      batch of negacyclic products (mod x^N + 1) of N = 2^log-len coefficients, batch dealt to ranks
  --workload fourstep  one product of length 2^24..2^26 split by columns, two all-to-alls
                       (ntt_amd.distributed.FourStepSharded); with one rank it is also compared
                       with the single-GPU three-pass plan

Prints one JSON line per workload on rank 0.  Timing: barrier + synchronize on both sides, max over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

P32 = 469762049
P64 = 1337006139375617


def timed(fn, steps, warmup, torch, dist, dev):
    for _ in range(warmup):
        fn()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="crt", choices=("crt", "batch", "fourstep", "nega"))
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--log-len", type=int, default=0)
    ap.add_argument("--p", type=int, default=0, help="batch / nega workloads: another 32-bit NTT prime than 469762049")
    args = ap.parse_args()

    import torch
    import ntt_amd
    from ntt_amd.distributed import CrtSharded, FourStepSharded, HipEngine, ThreadComm, TorchComm
    from ntt_amd.sharding import shard_range

    global P32
    if args.p:
        P32 = args.p
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        comm = TorchComm()
    else:
        comm = ThreadComm.run(1, lambda c: c)[0]
    ntt_amd.init(local_rank)
    eng = HipEngine()
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    out = {"workload": args.workload, "n_gpus": world, "steps": args.steps}

    if args.workload == "crt":
        n, B = 1 << 17, args.batch or 32 * world
        sh = CrtSharded(comm, eng)
        lo, hi = sh.batch_range(B)
        gi = torch.Generator(device=dev)
        gi.manual_seed(1000 + sh.br)          # the ranks of a batch group hold the same inputs
        a = torch.randint(0, P64, (hi - lo, n), generator=gi, device=dev, dtype=torch.int64)
        b = torch.randint(0, P64, (hi - lo, n), generator=gi, device=dev, dtype=torch.int64)
        dt = timed(lambda: sh.multiply(a, b, n, P64, gather=False), args.steps, args.warmup, torch, dist, dev)
        out.update(batch=B, layout={"prime_groups": sh.gp, "batch_groups": sh.gb}, poly_mults_per_s=B * args.steps / dt,
                   ms_per_step=dt / args.steps * 1e3, p=P64, n=n,
                   exchange_bytes_per_rank_per_step=int((hi - lo) * (1 << 18) * 4 * (4 // sh.gp) * (sh.gp - 1) / sh.gp))
    elif args.workload == "batch":
        logL = args.log_len or 16
        L, n, B = 1 << logL, 1 << (logL - 1), args.batch or 4096
        lo, hi = shard_range(B, rank, world)
        nb = hi - lo
        a = torch.zeros((nb, L), dtype=torch.int32, device=dev)
        b = torch.zeros((nb, L), dtype=torch.int32, device=dev)
        a[:, :n] = torch.randint(0, P32, (nb, n), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        b[:, :n] = torch.randint(0, P32, (nb, n), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        c = torch.empty_like(a)
        ntt_amd.reserve(L, nb, 1)
        st = torch.cuda.current_stream().cuda_stream
        dt = timed(lambda: ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P32, nb, st),
                   args.steps, args.warmup, torch, dist, dev)
        ntt_amd.profile_reset()
        ntt_amd.profile_enable(True)                # per-kernel HIP-event times of a few more steps
        for _ in range(10):
            ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P32, nb, st)
        torch.cuda.synchronize()
        ntt_amd.profile_enable(False)
        out["kernels_ms_per_step"] = {k: round(v[0] / 10, 4) for k, v in ntt_amd.profile_read().items() if v[1] > 0}
        out.update(batch=B, p=P32, log2_len=logL, poly_mults_per_s=B * args.steps / dt, ms_per_step=dt / args.steps * 1e3,
                   us_per_mult_per_gpu=dt / args.steps / nb * 1e6, passes=ntt_amd.plan_info(L),
                   model_52L_bytes_x_rate_GBs_NOT_traffic=52 * L * nb * args.steps / dt / 1e9)
    elif args.workload == "nega":
        logN = args.log_len or 12
        N, B = 1 << logN, args.batch or 4096
        lo, hi = shard_range(B, rank, world)
        nb = hi - lo
        a = torch.randint(0, P32, (nb, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        b = torch.randint(0, P32, (nb, N), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        c = torch.empty_like(a)
        st = torch.cuda.current_stream().cuda_stream
        dt = timed(lambda: ntt_amd.dev_negacyclic_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), N, P32, nb, st),
                   args.steps, args.warmup, torch, dist, dev)
        dtc = timed(lambda: ntt_amd.dev_cyclic_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), N, P32, nb, st),
                    args.steps, args.warmup, torch, dist, dev)
        out.update(batch=B, log2_len=logN, products_per_s=B * args.steps / dt, us_per_product_per_gpu=dt / args.steps / nb * 1e6,
                   cyclic_us_per_product_per_gpu=dtc / args.steps / nb * 1e6, passes=ntt_amd.plan_info(N),
                   moved_GBs_12N=12 * N * nb * args.steps / dt / 1e9)
    else:
        logL = args.log_len or 24
        L = 1 << logL
        fs = FourStepSharded(comm, eng, logL, P32)
        xa = torch.randint(0, P32, (fs.T, fs.cols), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        xb = torch.randint(0, P32, (fs.T, fs.cols), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        xa[fs.T // 2:] = 0
        xb[fs.T // 2:] = 0
        wa, wb = xa.clone(), xb.clone()

        def step():
            wa.copy_(xa)
            wb.copy_(xb)
            fs.multiply(wa, wb, zero_hi=True)

        dt = timed(step, args.steps, args.warmup, torch, dist, dev)
        out.update(log2_len=logL, log2_t=fs.log_t, ms_per_product=dt / args.steps * 1e3,
                   ns_per_butterfly=dt / args.steps / (3 * (L // 2) * logL) * 1e9,
                   exchange_bytes_per_rank_per_product=int(3 * fs.T * fs.cols * 4 * (world - 1) / world))
        if world == 1:
            a = torch.zeros(L, dtype=torch.int32, device=dev)
            b = torch.zeros(L, dtype=torch.int32, device=dev)
            a.view(fs.T, fs.S)[:] = xa
            b.view(fs.T, fs.S)[:] = xb
            c = torch.empty_like(a)
            st = torch.cuda.current_stream().cuda_stream
            dt1 = timed(lambda: ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), L // 2, P32, 1, st),
                        args.steps, args.warmup, torch, dist, dev)
            same = bool(torch.equal(c.view(fs.T, fs.S), fs.multiply(xa.clone(), xb.clone(), zero_hi=True)))
            out.update(single_plan_ms_per_product=dt1 / args.steps * 1e3, single_plan_passes=ntt_amd.plan_info(L),
                       single_plan_model_52L_bytes_x_rate_GBs_NOT_traffic=52 * L * args.steps / dt1 / 1e9, matches_single_plan=same)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ntt_amd.shutdown()


if __name__ == "__main__":
    main()
