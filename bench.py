#!/usr/bin/env python3
"""Headline benchmark: NTT polynomial multiplication at product length 2^18 on MI355X.

Workload (the single-prime configuration the headline metric is quoted on): prime 469762049, n = 2^17 coefficients
per input (product length L = 2^18), a batch of independent multiplications resident in
HBM.  A "step" is one pass of the hot path over the whole batch (batch multiplications).
With N GPUs every rank runs its own batch (weak scaling, no data-path collective).

Prints ONE JSON line (rank 0).  `value` = poly-mults/s over all ranks.  `roofline` is for
the dominant kernel, timed live with HIP events on the launch stream; `cpu_baseline` is the
compiled reference program (oracle/_ref/ref_main = main.cc) timed on this host.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P = 469762049
LOG_L = 18
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak
# algorithmic HBM bytes per multiplication, per kernel (u32 words of length L moved once):
#   col_fwd reads the n = L/2 coefficients of a and b (their zero padding is never read) and
#           writes both transformed copies                  -> 1 + 2 = 3 L words
#   row_mul reads both copies and writes the product rows   -> 3 L words
#   col_inv reads and writes the product                    -> 2 L words
KERNEL_WORDS = {"col_fwd": 3, "row_mul": 3, "col_inv": 2}
RAMP_STEPS = 50             # untimed steps always run before the timed region (clock ramp)
EVENT_EVERY = 4             # steps of the timed region whose kernels are bracketed by HIP events: every 4th
MODEL_BYTES_PER_L = 52       # two-passes-per-transform model: 13 L words per multiplication


def cpu_baseline(n, p, reps=80, reps_extra=5):
    """Time the reference CPU program itself on this host (its own chrono line): 4 * reps multiplications,
    about ten seconds of CPU work; the other CPU programs (reported extras) run reps_extra times each."""
    import numpy as np
    from oracle import pyoracle as po
    O = po.Oracle()
    cores = O.max_threads()
    cases = [([1, 5, 5, 4], [4, 1, 5, 2], 7340033)]
    for k in range(4):
        cases.append((O.fill_random(n, 1001 + k, p), O.fill_random(n, 2001 + k, p), p))
    if po.have_ref():
        _, lat, _ = po.run_ref_main(cases, which="ref_main", reps=reps)
        ms = [x for k in (1, 2, 3, 4) for x in lat.get(k, [])]
        kind = "reference"
        what = (f"oracle/_ref/ref_main (main.cc: Barrett, 4-prime CRT, OpenMP): {len(ms)} multiplications, "
                f"n={n}, p={p}, its own timed block (alloc+split+4 NTT products+CRT), median")
    else:
        ms = []
        for k in range(1, 5):
            t0 = time.perf_counter()
            O.poly_multiply_crt(cases[k][0], cases[k][1], p)
            ms.append((time.perf_counter() - t0) * 1e3)
        kind = "port"
        what = f"oracle/ntt_oracle.c restatement of main.cc, {len(ms)} multiplications, n={n}, p={p}, median"
    med = statistics.median(ms)
    out = {"value": 1000.0 / med, "unit": "poly-mults/s", "cores": cores, "kind": kind, "sample": what,
           "ms_per_mult": med, "note": "the reference's NTT stage uses at most 4 threads (one per CRT prime, main.cc:306)"}
    if po.have_ref():     # the pthread twin of the same four-prime program (main_pthread.cc: 4 NTT threads + CRT threads)
        try:
            _, lat2, _ = po.run_ref_main(cases, which="ref_main_pthread", reps=4 * reps_extra)
            ms2 = [x for k in (1, 2, 3, 4) for x in lat2.get(k, [])]
            if ms2:
                out["main_pthread_ms_per_mult"] = statistics.median(ms2)
        except Exception:                                  # noqa: BLE001  (a reported extra, never fatal)
            pass
    # the reference's single-prime programs on the same workload (the CPU analogue of the GPU path):
    # get_result_pthread (pthread_normal.cpp:361-413, 8 threads, Montgomery R = 2^30, conversions to and
    # from Montgomery form included) and one ntt_compute_omp call (main.cc:185-209, Barrett, one thread)
    if po.have_ref():
        single = {}
        a32 = np.asarray(cases[1][0], dtype=np.int32)
        b32 = np.asarray(cases[1][1], dtype=np.int32)
        for name, fn in (("pthread_normal.cpp get_result_pthread (8 threads)",
                          lambda: po.RefMont("pthread_normal").poly_multiply(a32, b32, p)),
                         ("monttry2.cpp get_result (8 threads per stage)",
                          lambda: po.RefMont("monttry2").poly_multiply(a32, b32, p))):
            ts = []
            for _ in range(reps_extra + 1):
                t0 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t0) * 1e3)
            single[name] = statistics.median(ts[1:])
        L = 1 << (2 * n - 1).bit_length()
        au = np.zeros(L, dtype=np.uint32)
        bu = np.zeros(L, dtype=np.uint32)
        au[:n] = a32
        bu[:n] = b32
        R = po.RefCRT("main")
        ts = []
        for _ in range(reps_extra):
            t0 = time.perf_counter()
            R.ntt_compute(au, bu, p)
            ts.append((time.perf_counter() - t0) * 1e3)
        single["main.cc ntt_compute_omp, one prime (1 thread)"] = statistics.median(ts)
        out["single_prime_ms_per_mult"] = single
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256, help="multiplications per GPU per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-check", action="store_true", help="ablation builds only: skip the oracle spot-check")
    ap.add_argument("--no-events", action="store_true", help="experiment: no per-kernel HIP events in the timed region")
    ap.add_argument("--no-context", action="store_true", help="profiling runs: skip the device-copy and batch-1 measurements "
                    "after the timed region, so a kernel trace holds only the benchmark's own launches")
    args = ap.parse_args()

    import numpy as np
    import torch
    import ntt_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # NTTX_BENCH_ONE_DEVICE=1 (rehearsal on a one-GPU box): every rank uses GPU 0 and the control-plane
    # collectives (barrier, max of the elapsed time) run over gloo, since RCCL wants one GPU per rank
    rehearsal = os.environ.get("NTTX_BENCH_ONE_DEVICE") == "1"
    device_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            # RCCL carries only the barrier and the max of the elapsed time here (the data path has no collective);
            # if it cannot come up on this node, the same two calls run over gloo on host memory
            try:
                dist.init_process_group("nccl", device_id=dev)
                probe = torch.zeros(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
            except Exception as exc:                       # noqa: BLE001
                sys.stderr.write(f"rank {rank}: RCCL unavailable ({exc}); control-plane collectives over gloo\n")
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("gloo")
                rehearsal = True                           # reductions on host tensors from here on
    ntt_amd.init(device_index)

    L, n, B = 1 << LOG_L, 1 << (LOG_L - 1), args.batch
    # synthetic inputs, generated on the device: uniform in [0, p), upper half zero (padding)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    a = torch.zeros((B, L), dtype=torch.int32, device=dev)
    b = torch.zeros((B, L), dtype=torch.int32, device=dev)
    a[:, :n] = torch.randint(0, P, (B, n), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    b[:, :n] = torch.randint(0, P, (B, n), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    c = torch.empty_like(a)
    ntt_amd.reserve(L, B, 1)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P, B, stream)

    step()
    torch.cuda.synchronize()

    # spot-check one product of the batch against the oracle before timing anything
    if rank == 0 and not args.no_check:
        from oracle import pyoracle as po
        O = po.Oracle()
        ah = a[0].cpu().numpy().view(np.uint32)
        bh = b[0].cpu().numpy().view(np.uint32)
        want = O.ntt_compute(ah, bh, P)
        got = c[0].cpu().numpy().view(np.uint32)
        if not np.array_equal(got, want):
            print(json.dumps({"error": "GPU result differs from the oracle"}))
            sys.exit(1)

    # warm-up directly before the timed region: the host-side check above leaves the GPU idle long
    # enough for its clocks to drop
    # (MI355X takes some 20-30 ms of load to reach its steady clocks), topped up to RAMP_STEPS steps
    ramp = max(0, RAMP_STEPS - args.warmup)
    for _ in range(args.warmup + ramp):
        step()
    torch.cuda.synchronize()
    ntt_amd.profile_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # per-kernel HIP events bracket every EVENT_EVERY-th step of the timed region: an event record keeps the
    # next kernel from starting while the previous one drains (2.5 % on the whole step when every step carries them)
    for i in range(args.steps):
        if not args.no_events and i % EVENT_EVERY == 0:
            ntt_amd.profile_enable(True)
            step()
            ntt_amd.profile_enable(False)
        else:
            step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ntt_amd.profile_enable(False)
    prof = ntt_amd.profile_read()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # two context figures, outside the timed region: what a plain device-to-device copy of the same
    # buffers reaches on this GPU (read L*B words + write L*B words), and the latency of ONE resident product
    copy_gbs = lat1_us = None
    if rank == 0 and not args.no_events and not args.no_context:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            c.copy_(a)
        e0.record()
        for _ in range(20):
            c.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2 * 4 * L * B * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        for _ in range(50):
            ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P, 1, stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(200):
            ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P, 1, stream)
        torch.cuda.synchronize()
        lat1_us = (time.perf_counter() - t1) / 200 * 1e6

    if rank == 0:
        mults = world * B * args.steps
        value = mults / elapsed
        butterflies = 3 * (L // 2) * LOG_L
        kern = {k: v for k, v in prof.items() if v[1] > 0}
        if not kern:
            print(json.dumps({"us_per_mult_per_gpu": elapsed / (B * args.steps) * 1e6, "ms_per_step": elapsed / args.steps * 1e3, "events": False}))
            ntt_amd.shutdown()
            return
        dom = max(kern, key=lambda k: kern[k][0])
        dom_ms, dom_launches = kern[dom]
        dom_avg_s = dom_ms / dom_launches * 1e-3
        dom_bytes = KERNEL_WORDS.get(dom, 0) * 4 * L * B
        achieved = dom_bytes / dom_avg_s / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(dom)
            except Exception:
                traffic = None
        model_gbs = MODEL_BYTES_PER_L * L * (value / world) / 1e9
        out = {
            "metric": "poly-mults/sec at product length 2^18 (single 32-bit prime)",
            "value": value,
            "unit": "poly-mults/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_ramp_steps": ramp,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: single prime 469762049, product length 2^18 (n=2^17 per input), "
                                   "Montgomery LDS-fused NTT, batch resident in HBM",
                       "p": P, "log2_len": LOG_L, "batch_per_gpu": B, "passes": ntt_amd.plan_info(L),
                       "parallelism": f"batch-sharded x{world}, no collective"},
            "ns_per_butterfly": elapsed / (mults / world) / butterflies * 1e9,
            "us_per_mult_per_gpu": elapsed / (B * args.steps) * 1e6,
            "us_per_mult_batch_1": lat1_us,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms / dom_launches,
                         "launches_timed": dom_launches,
                         "device_copy_GBs": copy_gbs, "frac_of_device_copy": achieved / copy_gbs if copy_gbs else None},
            "kernels_ms_per_step": {k: v[0] / v[1] for k, v in kern.items()},
            "pipeline_model": {"bytes_per_mult_52L": MODEL_BYTES_PER_L * L, "achieved_GBs_per_gpu": model_gbs,
                               "frac_of_peak": model_gbs / HBM_PEAK_GBS,
                               "bytes_per_mult_moved": 32 * L},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(n, P)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ntt_amd.shutdown()


if __name__ == "__main__":
    main()
