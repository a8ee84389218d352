#!/usr/bin/env python3
"""Benchmarks of the NTT polynomial-multiplication path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload single|crt|batch|fourstep]

Workloads (a "step" is one pass of the hot path over one batch of synthetic input):

  single    (default, the headline metric) prime 469762049, n = 2^17 coefficients per input (product length
            L = 2^18), a batch of independent products resident in HBM; every rank runs its own batch
            (weak scaling, no data-path collective)
  crt       51-bit modulus 1337006139375617 through the reference's four primes + CRT (main.cc:286-338),
            n = 2^17: the primes are dealt to min(N, 4) ranks, ONE all-to-all hands every rank the four
            residues of its slice of the indices, recombination on slices (ntt_amd.distributed.CrtSharded);
            32 products per rank and step (weak scaling)
  batch     4096 independent products of length 2^16 dealt to the ranks, no collective (strong scaling)
  fourstep  ONE product of length 2^24 (--log-len up to 26): one rank runs the single-GPU three-pass plan,
            N > 1 ranks split the T x S matrix by columns, one all-to-all per vector (a, b, product)
            (ntt_amd.distributed.FourStepSharded; strong scaling)

Ranks: under `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process is
one rank.  Started plainly with --gpus N > 1 it spawns N child ranks itself, BEFORE any GPU call is made in this
process, and relays rank 0's JSON line.  One process per GPU, RCCL ("nccl") between them.
NTTX_BENCH_ONE_DEVICE=1 rehearses several ranks on ONE GPU (collectives staged through host memory over gloo,
flagged in the output); --dry-run exercises only the launch / rendezvous / timing plumbing (no GPU needed).

Prints ONE JSON line (rank 0).  `value` = whole-job throughput over all ranks with inputs resident in HBM.
`roofline` is for the dominant kernel, timed live with HIP events on the launch stream, against the HBM bytes it
moves; `cpu_baseline` (rank 0, N = 1) is the reference's CPU program for the workload timed on this host.
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P32 = 469762049
P64 = 1337006139375617       # harness id 4's modulus (test.o:11)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak
CLOCK_GHZ = 2.4              # MI355X peak engine clock
SIMDS = 1024                 # 256 CUs x 4 SIMDs
RAMP_MS = 30.0               # untimed back-to-back load run before every timed region (clock ramp)
EVENT_EVERY = 4              # steps of the timed region whose kernels are bracketed by HIP events: every 4th
MODEL_BYTES_PER_L = 52       # two-passes-per-transform model: 13 L words per single-prime product
KERNEL_SOURCES = ("nttx_kernels.hpp", "nttx_arith.hpp", "nttx.hip", "nttx_col.hip", "nttx_launch.hpp")


def kernel_source_tag():
    """Hash of the kernel sources: counter figures recorded under profiles/ are only quoted while it matches."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "ntt_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def recorded_counters(name, workload, kernel=None):
    """profiles/<name>.json = {"kernel_source_tag": ..., "workloads": {workload: {"kernels": {kind: value}, ...}}}.
    Returns (value of one kind, or the whole workload record when kernel is None; state); None when absent or stale."""
    path = os.path.join(ROOT, "profiles", name + ".json")
    try:
        rec = json.load(open(path))
    except Exception:                                      # noqa: BLE001
        return None, "absent"
    if rec.get("kernel_source_tag") != kernel_source_tag():
        return None, "stale (kernel sources changed since it was measured)"
    w = rec.get("workloads", {}).get(workload)
    if w is None:
        return None, "not recorded for this workload"
    return (w if kernel is None else w.get("kernels", {}).get(kernel)), "current"


VALU_CYCLES_PER_INSTRUCTION = 4.6     # butterfly code measured alone on gfx950: one wave64 VALU instruction per ~4.6 cycles per SIMD


# ------------------------------------------------------------------------------------------------
# CPU baselines: the reference's own programs (oracle/_ref) timed on this host, bounded samples
# ------------------------------------------------------------------------------------------------
def cpu_baseline_main_cc(n, p, reps=80, reps_extra=5, extras=True):
    """main.cc itself (its own chrono line): 4 * reps products, about ten seconds of CPU work."""
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
    if not (extras and po.have_ref()):
        return out
    try:   # the pthread twin of the same four-prime program (main_pthread.cc: 4 NTT threads + CRT threads)
        _, lat2, _ = po.run_ref_main(cases, which="ref_main_pthread", reps=4 * reps_extra)
        ms2 = [x for k in (1, 2, 3, 4) for x in lat2.get(k, [])]
        if ms2:
            out["main_pthread_ms_per_mult"] = statistics.median(ms2)
    except Exception:                                      # noqa: BLE001  (a reported extra, never fatal)
        pass
    if p < (1 << 30):
        # the reference's single-prime programs on the same workload (the CPU analogue of the GPU path):
        # get_result_pthread (pthread_normal.cpp:361-413, 8 threads, Montgomery R = 2^30, conversions to and
        # from Montgomery form included) and one ntt_compute_omp call (main.cc:185-209, Barrett, one thread)
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
        # the same stage-parallel program with its thread handling tuned (oracle/cpu_pool.c: one standing pool and a
        # barrier per stage instead of 18 create/join rounds per transform; the butterflies are the reference's)
        for threads in sorted({8, min(cores, 32)}):
            ts = []
            for _ in range(reps_extra + 1):
                t0 = time.perf_counter()
                O.pool_poly_multiply(a32, b32, p, threads)
                ts.append((time.perf_counter() - t0) * 1e3)
            single[f"pooled variant of monttry2.cpp / pthread_normal.cpp ({threads} threads, port)"] = statistics.median(ts[1:])
        out["single_prime_ms_per_mult"] = single
    return out


def cpu_baseline_single_prime(n, p, budget_s=12.0):
    """pthread_normal.cpp's get_result_pthread (the reference's single-prime program, 8 threads, conversions to and
    from Montgomery form included) on products of n coefficients, for about budget_s seconds."""
    import numpy as np
    from oracle import pyoracle as po
    O = po.Oracle()
    a = np.asarray(O.fill_random(n, 1001, p), dtype=np.int32)
    b = np.asarray(O.fill_random(n, 2001, p), dtype=np.int32)
    if po.have_ref():
        R = po.RefMont("pthread_normal")
        fn = lambda: R.poly_multiply(a, b, p)              # noqa: E731
        kind, cores = "reference", 8
        what = "oracle/_ref/libref_pthread_normal.so (pthread_normal.cpp get_result_pthread, NUM_THREADS = 8)"
    else:
        fn = lambda: O.poly_multiply_mont(a, b, p)         # noqa: E731
        kind, cores = "port", 1
        what = "oracle/ntt_oracle.c restatement of monttry2.cpp's single-prime product (one thread)"
    fn()
    ts, t_end = [], time.perf_counter() + budget_s
    while time.perf_counter() < t_end or len(ts) < 5:
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    med = statistics.median(ts)
    out = {"value": 1000.0 / med, "unit": "poly-mults/s", "cores": cores, "kind": kind, "ms_per_mult": med,
           "sample": f"{what}: {len(ts)} products of n={n} coefficients, p={p}, one after the other, median"}
    tp = []
    for _ in range(6):     # the tuned-thread-handling port of the same program (oracle/cpu_pool.c), a reported extra
        t0 = time.perf_counter()
        O.pool_poly_multiply(a, b, p, 8)
        tp.append((time.perf_counter() - t0) * 1e3)
    out["pooled_variant_8_threads_ms_per_mult"] = statistics.median(tp[1:])
    return out


def cpu_baseline_long(log_len, p):
    """The reference's arrays stop at 300000 words (main.cc:228), so its programs cannot run these lengths; the
    CPU figure is the oracle's restatement of ntt_compute_omp (main.cc:185-209) on heap vectors, one product."""
    import numpy as np
    from oracle import pyoracle as po
    O = po.Oracle()
    L = 1 << log_len
    a = np.zeros(L, dtype=np.uint32)
    b = np.zeros(L, dtype=np.uint32)
    a[: L // 2] = O.fill_random(L // 2, 7, p)
    b[: L // 2] = O.fill_random(L // 2, 8, p)
    t0 = time.perf_counter()
    O.ntt_compute(a, b, p)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "poly-mults/s", "cores": 1, "kind": "port", "ms_per_mult": dt * 1e3,
            "sample": f"oracle/ntt_oracle.c restatement of ntt_compute_omp (main.cc:185-209), ONE product of length 2^{log_len}, "
                      f"p={p} (the reference's own arrays stop at 300000 words, main.cc:228)"}


# ------------------------------------------------------------------------------------------------
# launching ranks
# ------------------------------------------------------------------------------------------------
def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="single", choices=("single", "crt", "batch", "fourstep"))
    ap.add_argument("--batch", type=int, default=0, help="single / crt: products per GPU per step; batch: products in all")
    ap.add_argument("--log-len", type=int, default=0, help="batch: product length (16 or 17); fourstep: 24 .. 26")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-check", action="store_true", help="ablation builds only: skip the oracle spot-check")
    ap.add_argument("--no-events", action="store_true", help="experiment: no per-kernel HIP events in the timed region")
    ap.add_argument("--no-context", action="store_true", help="profiling runs: skip the device-copy and batch-1 measurements "
                    "after the timed region, so a kernel trace holds only the benchmark's own launches")
    ap.add_argument("--dry-run", action="store_true", help="launch / rendezvous / timing plumbing only: the step is empty, "
                    "no GPU is touched, the line carries dry_run: true and no value")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"single": 200, "crt": 20, "batch": 20, "fourstep": 20}[args.workload]
    if args.warmup is None:
        args.warmup = {"single": 50, "crt": 5, "batch": 5, "fourstep": 5}[args.workload]
    return args


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """Parent of a plain `python bench.py --gpus N` run: starts N fresh child ranks and waits for them.  Nothing here
    touches the GPU (device_count() does not initialise it), so the children are ordinary child processes of a process
    without a HIP context.  Rank 0 writes the JSON line to this process's stdout; the other ranks' stdout goes to stderr."""
    one_device = os.environ.get("NTTX_BENCH_ONE_DEVICE") == "1"
    if "--dry-run" not in sys.argv and not one_device:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print(json.dumps({"error": f"--gpus {n} but this node shows {have} GPU(s)"}))
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def relay():     # rank 0's JSON line goes to stdout, anything else a library printed there goes to stderr
        for line in procs[0].stdout:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()

    import threading
    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write(f"bench.py: rank {r} exited with {code}; stopping the other ranks\n")
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    pump.join(timeout=10)
    return rc


class HostStagedComm:
    """Rehearsal only (several ranks on ONE GPU, where RCCL cannot run): the collectives of a TorchComm over gloo,
    with device tensors staged through host memory."""

    def __init__(self, inner, device):
        self._c, self._dev = inner, device
        self.world, self.rank = inner.world, inner.rank

    def all_to_all(self, send):
        return self._c.all_to_all(send.cpu()).to(self._dev)

    def all_to_all_start(self, send):
        from ntt_amd.distributed import _Pending
        return _Pending(None, self.all_to_all(send))

    def all_gather(self, x):
        return self._c.all_gather(x.cpu()).to(self._dev)

    def barrier(self):
        self._c.barrier()

    def split(self, color_of_rank):
        return HostStagedComm(self._c.split(color_of_rank), self._dev)


class Rank:
    """This process as one rank: device, process group, and the barrier-bracketed timer."""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = None
        self.rccl_ranks = None
        self.rehearsal = os.environ.get("NTTX_BENCH_ONE_DEVICE") == "1"
        self.dry = args.dry_run
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if self.dry:
            self.torch = __import__("torch")
            self.dev = None
            if self.world > 1:
                import torch.distributed as dist
                dist.init_process_group("gloo")
                self.dist, self.backend = dist, "gloo"
            return
        import torch
        self.torch = torch
        index = 0 if self.rehearsal else self.local_rank
        torch.cuda.set_device(index)
        self.dev = torch.device("cuda", index)
        self.device_index = index
        if self.world > 1:
            import torch.distributed as dist
            if self.rehearsal:
                dist.init_process_group("gloo")
                self.backend = "gloo (rehearsal: every rank on GPU 0, collectives staged through host memory)"
            else:
                dist.init_process_group("nccl", device_id=self.dev)
                probe = torch.ones(1, device=self.dev)
                dist.all_reduce(probe)                      # brings RCCL up; the sum is the rank count it reaches
                torch.cuda.synchronize()
                self.rccl_ranks = int(probe.item())
                self.backend = "nccl (RCCL)"
            self.dist = dist

    def comm(self):
        from ntt_amd.distributed import ThreadComm, TorchComm
        if self.world == 1:
            return ThreadComm.run(1, lambda c: c)[0]
        c = TorchComm()
        return HostStagedComm(c, self.dev) if self.rehearsal else c

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        cpu = self.dry or self.rehearsal
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cpu" if cpu else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, steps, warmup, events=False):
        """warmup untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides;
        returns the MAX elapsed seconds over ranks.  events: every EVENT_EVERY-th step has its kernels bracketed by
        HIP events on the launch stream (an event record keeps the next kernel from starting while the previous one
        drains: 2.5 % on the whole step when every step carries them)."""
        import ntt_amd
        for _ in range(warmup):
            step()
        self.sync()
        # MI355X needs some 20-30 ms of load to reach the clocks it then holds; a host-side check or a short warm-up leaves
        # it below them, so the warm-up is topped up with untimed steps until RAMP_MS of back-to-back load have run
        # (the count comes from four measured steps, maximum over the ranks, so every rank runs the same number of steps:
        # steps may hold collectives)
        self.ramp_steps = 0
        if not self.dry:
            t_ramp = time.perf_counter()
            for _ in range(4):
                step()
            self.sync()
            per_step = self.max_over_ranks((time.perf_counter() - t_ramp) / 4)
            self.ramp_steps = max(4, min(4000, int(RAMP_MS * 1e-3 / max(per_step, 1e-6)) + 1))
            for _ in range(self.ramp_steps - 4):
                step()
            self.sync()
        if events:
            ntt_amd.profile_reset()
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            if events and i % EVENT_EVERY == 0:
                ntt_amd.profile_enable(True)
                step()
                ntt_amd.profile_enable(False)
            else:
                step()
        self.sync()
        self.barrier()
        elapsed = time.perf_counter() - t0
        return self.max_over_ranks(elapsed)

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def roofline_of(prof, bytes_per_step, launches_per_step, workload=None):
    """Dominant kernel kind (largest share of the event-timed device time) against the HBM bytes it moves, plus the
    whole step, the counter records of this workload (HBM traffic, VALU instructions) and the bound they name.
    prof: {kind: (total ms, launches)}; bytes_per_step / launches_per_step: {kind: ...} of this rank."""
    kern = {k: v for k, v in prof.items() if v[1] > 0}
    if not kern:
        return None, {}
    dom = max(kern, key=lambda k: kern[k][0])
    ms, launches = kern[dom]
    avg_s = ms / launches * 1e-3
    per_launch = bytes_per_step.get(dom, 0) / max(1, launches_per_step.get(dom, 1))
    achieved = per_launch / avg_s / 1e9
    traffic, traffic_state = recorded_counters("traffic", workload, dom)
    roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_record": traffic_state,
            "bytes": "HBM bytes the kernel moves per launch (its reads + writes, each word once)",
            "algorithmic_bytes_per_launch": per_launch, "avg_launch_ms": ms / launches, "launches_timed": launches,
            "timing": "HIP events on the launch stream around every kernel of every 4th timed step; a step whose kernels are "
                      "bracketed by events runs 1-2 % slower than ms_per_step (an event record keeps the next kernel from "
                      "starting while the previous one drains), so these fractions are slightly conservative"}
    # whole step: every kernel's moved bytes over the sum of the kernels' event-timed durations
    tot_b = sum(bytes_per_step.get(k, 0) for k in kern)
    per_kind_s = {k: kern[k][0] / kern[k][1] * launches_per_step.get(k, 1) * 1e-3 for k in kern}
    tot_s = sum(per_kind_s.values())
    roof["whole_step"] = {"moved_bytes_per_step": tot_b, "kernel_ms_per_step": tot_s * 1e3, "achieved": tot_b / tot_s / 1e9,
                          "frac": tot_b / tot_s / 1e9 / HBM_PEAK_GBS}
    wrec, wstate = recorded_counters("traffic", workload)
    if wrec:
        t = sum(wrec["kernels"].get(k, 0) * launches_per_step.get(k, 1) for k in kern)
        roof["whole_step"]["traffic_bytes_per_step"] = t
        roof["whole_step"]["traffic_over_moved"] = t / tot_b if tot_b else None
    vrec, vstate = recorded_counters("valu", workload)
    if vrec:
        insts_k, clk_k = vrec.get("kernels", {}), vrec.get("clock_GHz", {})
        insts, clk = insts_k.get(dom), clk_k.get(dom)
        valu = {"record": vstate, "model_cycles_per_instruction": VALU_CYCLES_PER_INSTRUCTION,
                "note": "SQ_INSTS_VALU (wave-level vector instructions) per launch from the tagged counter record of this "
                        "workload over 1024 SIMDs x the clock held in the profile x the live launch time; butterfly code "
                        "measured alone issues one instruction per about 4.6 cycles per SIMD (64-bit multiply-adds 4.5, "
                        "32-bit multiplies 4.2, adds 2.5), so issue_busy_frac near 1 means the kernel is bound by integer-VALU "
                        "issue, not by HBM"}
        if insts and clk:
            valu.update({"wave_instructions_per_launch": insts, "clock_GHz_in_profile": clk,
                         "simd_cycles_per_instruction_at_2.4GHz": SIMDS * CLOCK_GHZ * 1e9 * avg_s / insts,
                         "simd_cycles_per_instruction_at_profiled_clock": SIMDS * clk * 1e9 * avg_s / insts,
                         "issue_busy_frac": insts * VALU_CYCLES_PER_INSTRUCTION / (SIMDS * clk * 1e9 * avg_s)})
        # the whole step on the same model: every kind's instructions at the clock it ran at in the profile
        busy_s = sum(insts_k[k] * launches_per_step.get(k, 1) * VALU_CYCLES_PER_INSTRUCTION / (SIMDS * clk_k[k] * 1e9)
                     for k in kern if insts_k.get(k) and clk_k.get(k))
        if busy_s:
            valu["whole_step"] = {"wave_instructions_per_step": sum(insts_k[k] * launches_per_step.get(k, 1) for k in kern if insts_k.get(k)),
                                  "issue_busy_ms_per_step_by_model": busy_s * 1e3, "kernel_ms_per_step": tot_s * 1e3,
                                  "issue_busy_frac": busy_s / tot_s}
        roof["valu"] = valu
        # the bound the evidence names: the larger of the two fractions (HBM bytes moved over the spec peak, VALU issue by
        # the model above), for the dominant kernel
        if valu.get("issue_busy_frac", 0) > roof["frac"]:
            roof["bound"] = "valu"
            roof["bound_note"] = ("integer-VALU issue: the dominant kernel's vector instructions fill %.0f %% of its SIMD cycles by "
                                  "the issue model while its HBM bytes are %.0f %% of the 8 TB/s peak; achieved / peak / frac stay "
                                  "the HBM figures (no MFMA on this path)" % (100 * valu["issue_busy_frac"], 100 * roof["frac"]))
    else:
        roof["valu"] = {"record": vstate}
    return roof, {k: v[0] / v[1] for k, v in kern.items()}


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def power_sample(step, torch, settle_s=1.5):
    """Socket power and shader clocks while `step` runs back to back (a context figure, outside the timed region): the
    step is enqueued for settle_s seconds, then `amd-smi metric --power --clock` runs as a child process while more
    steps keep the GPU busy.  None where the tool is absent or its output has another shape."""
    import re
    import shutil
    import subprocess
    if not shutil.which("amd-smi"):
        return None
    try:
        t_end = time.perf_counter() + settle_s
        while time.perf_counter() < t_end:
            for _ in range(50):
                step()
            torch.cuda.synchronize()
        proc = subprocess.Popen(["amd-smi", "metric", "-g", "0", "--power", "--clock"], stdout=subprocess.PIPE,
                                stderr=subprocess.DEVNULL, text=True)
        t_end = time.perf_counter() + 30.0
        while proc.poll() is None and time.perf_counter() < t_end:
            for _ in range(50):
                step()
            torch.cuda.synchronize()
        if proc.poll() is None:
            proc.kill()
            return None
        text = proc.stdout.read()
        watts = [int(x) for x in re.findall(r"SOCKET_POWER: (\d+) W", text)]
        gfx = text.split("GFX_0:")[1].split("MEM_0:")[0] if "GFX_0:" in text and "MEM_0:" in text else ""
        clocks = [int(x) for x in re.findall(r"^\s+CLK: (\d+) MHz", gfx, re.M)]
        cap = None
        if shutil.which("rocm-smi"):
            m = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)",
                          subprocess.run(["rocm-smi", "-d", "0", "--showmaxpower"], capture_output=True, text=True, timeout=30).stdout)
            cap = float(m.group(1)) if m else None
        if not watts or not clocks:
            return None
        return {"socket_W": watts[0], "cap_W": cap, "shader_MHz_per_xcd": clocks, "shader_MHz_max": 2400,
                "what": f"one amd-smi sample of GPU 0 taken while the step ran back to back (after {settle_s} s of load), outside "
                        "the timed region: at the cap the part lowers its clock, and the VALU fractions refer to that clock"}
    except Exception:                                      # noqa: BLE001  (a reported extra, never fatal)
        return None


def rand_padded(torch, gen, dev, rows, n, length, p):
    x = torch.zeros((rows, length), dtype=torch.int32, device=dev)
    x[:, :n] = torch.randint(0, p, (rows, n), generator=gen, device=dev, dtype=torch.int64).to(torch.int32)
    return x


def run_single(R, args):
    import numpy as np
    import ntt_amd
    torch = R.torch
    LOG_L = 18
    L, n, B = 1 << LOG_L, 1 << (LOG_L - 1), args.batch or 256
    g = torch.Generator(device=R.dev)
    g.manual_seed(1234 + R.rank)
    a, b = rand_padded(torch, g, R.dev, B, n, L, P32), rand_padded(torch, g, R.dev, B, n, L, P32)
    c = torch.empty_like(a)
    ntt_amd.reserve(L, B, 1)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P32, B, stream)

    step()
    torch.cuda.synchronize()
    if R.rank == 0 and not args.no_check:   # spot-check one product of the batch against the oracle before timing anything
        from oracle import pyoracle as po
        want = po.Oracle().ntt_compute(a[0].cpu().numpy().view(np.uint32), b[0].cpu().numpy().view(np.uint32), P32)
        if not np.array_equal(c[0].cpu().numpy().view(np.uint32), want):
            raise SystemExit(json.dumps({"error": "GPU result differs from the oracle"}))
    elapsed = R.timed(step, args.steps, args.warmup, events=not args.no_events)
    ramp = R.ramp_steps
    prof = ntt_amd.profile_read() if not args.no_events else {}

    copy_gbs = lat1_us = power = None
    if R.rank == 0 and not args.no_context:
        # two context figures, outside the timed region: what the library's own 16-bytes-per-lane streaming copy kernel
        # reaches on the same buffers on this GPU (read L*B words + write L*B words: the achievable ceiling of a
        # read-once / write-once pass), and the latency of ONE resident product
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            ntt_amd.dev_stream_copy(c.data_ptr(), a.data_ptr(), 4 * L * B, stream)
        e0.record()
        for _ in range(20):
            ntt_amd.dev_stream_copy(c.data_ptr(), a.data_ptr(), 4 * L * B, stream)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2 * 4 * L * B * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        for _ in range(50):
            ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P32, 1, stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(200):
            ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P32, 1, stream)
        torch.cuda.synchronize()
        lat1_us = (time.perf_counter() - t1) / 200 * 1e6
        if R.world == 1:
            power = power_sample(step, torch)
    # HBM words each kernel moves per product (u32 words of length L moved once): col_fwd reads the n = L/2
    # coefficients of a and b (their zero padding is never read) and writes both transformed copies (1 + 2 = 3 L words),
    # row_mul reads both copies and writes the product rows (3 L), col_inv reads and writes the product (2 L)
    words = {"col_fwd": 3, "row_mul": 3, "col_inv": 2}
    mults = R.world * B * args.steps
    res = {
        "metric": "poly-mults/sec at product length 2^18 (single 32-bit prime)", "unit": "poly-mults/s", "dtype": "u32",
        "units": mults, "elapsed": elapsed, "scaling": "weak", "clock_ramp_steps": ramp,
        "config": {"workload": "configs[1]: single prime 469762049, product length 2^18 (n=2^17 per input), "
                               "Montgomery LDS-fused NTT, batch resident in HBM",
                   "p": P32, "log2_len": LOG_L, "batch_per_gpu": B, "passes": ntt_amd.plan_info(L),
                   "parallelism": f"batch-sharded x{R.world}, no collective"},
        "butterflies_per_unit": 3 * (L // 2) * LOG_L,
        "bytes_per_step": {k: w * 4 * L * B for k, w in words.items()}, "launches_per_step": {k: 1 for k in words},
        "prof": prof, "roofline_kind": f"single_b{B}",
        "extra": {"us_per_mult_per_gpu": elapsed / (B * args.steps) * 1e6, "us_per_mult_batch_1": lat1_us, "power": power,
                  "pipeline_model": {"what": "a MODEL, not traffic: a two-HBM-passes-per-transform byte count "
                                             "(52 L per product) times the measured rate; the kernels move 32 L",
                                     "model_bytes_per_mult_52L": MODEL_BYTES_PER_L * L, "bytes_per_mult_moved": 32 * L,
                                     "model_GBs_per_gpu": MODEL_BYTES_PER_L * L * (mults / R.world / elapsed) / 1e9,
                                     "model_frac_of_peak": MODEL_BYTES_PER_L * L * (mults / R.world / elapsed) / 1e9 / HBM_PEAK_GBS}},
        "achievable_copy_GBs": copy_gbs,
        "cpu": (lambda: cpu_baseline_main_cc(n, P32)),
    }
    return res


def run_batch(R, args):
    import numpy as np
    import ntt_amd
    from ntt_amd.sharding import shard_range
    torch = R.torch
    logL = args.log_len or 16
    L, n, total = 1 << logL, 1 << (logL - 1), args.batch or 4096
    lo, hi = shard_range(total, R.rank, R.world)
    nb = hi - lo
    g = torch.Generator(device=R.dev)
    g.manual_seed(4321 + R.rank)
    a, b = rand_padded(torch, g, R.dev, nb, n, L, P32), rand_padded(torch, g, R.dev, nb, n, L, P32)
    c = torch.empty_like(a)
    ntt_amd.reserve(L, nb, 1)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, P32, nb, stream)

    step()
    torch.cuda.synchronize()
    if R.rank == 0 and not args.no_check:
        from oracle import pyoracle as po
        want = po.Oracle().ntt_compute(a[nb - 1].cpu().numpy().view(np.uint32), b[nb - 1].cpu().numpy().view(np.uint32), P32)
        if not np.array_equal(c[nb - 1].cpu().numpy().view(np.uint32), want):
            raise SystemExit(json.dumps({"error": "GPU result differs from the oracle"}))
    elapsed = R.timed(step, args.steps, args.warmup, events=not args.no_events)
    prof = ntt_amd.profile_read() if not args.no_events else {}
    words = {"col_fwd": 3, "row_mul": 3, "col_inv": 2}
    return {
        "metric": f"poly-mults/sec, {total} independent products of length 2^{logL}", "unit": "poly-mults/s", "dtype": "u32",
        "units": total * args.steps, "elapsed": elapsed, "scaling": "strong",
        "config": {"workload": f"configs[3]: {total} independent products of length 2^{logL} (n=2^{logL - 1} per input), prime 469762049, "
                               "dealt to the ranks, no inter-GPU communication",
                   "p": P32, "log2_len": logL, "batch_total": total, "batch_this_rank": nb, "passes": ntt_amd.plan_info(L),
                   "parallelism": f"batch-sharded x{R.world}, no collective"},
        "butterflies_per_unit": 3 * (L // 2) * logL,
        "bytes_per_step": {k: w * 4 * L * nb for k, w in words.items()}, "launches_per_step": {k: 1 for k in words},
        "prof": prof, "roofline_kind": f"batch_2p{logL}_b{nb}",
        "extra": {"us_per_mult_per_gpu": elapsed / (nb * args.steps) * 1e6},
        "cpu": (lambda: cpu_baseline_single_prime(n, P32)),
    }


def run_crt(R, args):
    import numpy as np
    import ntt_amd
    from ntt_amd.distributed import CrtSharded, HipEngine
    torch = R.torch
    n, L = 1 << 17, 1 << 18
    per_rank = args.batch or 32
    total = per_rank * R.world
    sh = CrtSharded(R.comm(), HipEngine())
    lo, hi = sh.batch_range(total)
    nb = hi - lo
    gi = torch.Generator(device=R.dev)
    gi.manual_seed(1000 + sh.br)                # the ranks of a batch group hold the same inputs
    a = torch.randint(0, P64, (nb, n), generator=gi, device=R.dev, dtype=torch.int64)
    b = torch.randint(0, P64, (nb, n), generator=gi, device=R.dev, dtype=torch.int64)
    ntt_amd.reserve(L, nb, len(sh.own))
    out = sh.multiply(a, b, n, P64, gather=True)
    torch.cuda.synchronize()
    if R.rank == 0 and not args.no_check:
        from oracle import pyoracle as po
        want = po.Oracle().poly_multiply_crt(a[0].cpu().numpy().view(np.uint64), b[0].cpu().numpy().view(np.uint64), P64)
        if not np.array_equal(out[0].cpu().numpy().view(np.uint64), want):
            raise SystemExit(json.dumps({"error": "GPU result differs from the oracle"}))
    del out
    elapsed = R.timed(lambda: sh.multiply(a, b, n, P64, gather=False), args.steps, args.warmup, events=not args.no_events)
    prof = ntt_amd.profile_read() if not args.no_events else {}
    k = len(sh.own)                               # primes this rank transforms for
    sl = L // sh.gp                               # indices this rank recombines per product
    # the residue split happens inside the first column pass (which then reads the u64 coefficients: 16 n + 8 L bytes per
    # product and prime) unless the library was told to run it as a pass of its own (a split kernel shows in the timings)
    fused = not (prof.get("split") and prof["split"][1] > 0) if prof else True
    bytes_per_step = {"col_fwd": (16 if fused else 12) * L * nb * k, "row_mul": 12 * L * nb * k, "col_inv": 8 * L * nb * k,
                      "crt": (16 + 8) * sl * nb}
    launches = {"col_fwd": k, "row_mul": k, "col_inv": k, "crt": 1}
    if not fused:
        bytes_per_step["split"] = (16 + 8 * k) * n * nb
        launches["split"] = 1
    exch = int(nb * L * 4 * k * (sh.gp - 1) / sh.gp)
    return {
        "metric": "poly-mults/sec at n=2^17 coefficients modulo a 51-bit prime (four 32-bit NTT primes + CRT)", "unit": "poly-mults/s",
        "dtype": "u32 (transforms), u64 (split and recombination)",
        "units": total * args.steps, "elapsed": elapsed, "scaling": "weak",
        "config": {"workload": "configs[2]: modulus 1337006139375617 (harness id 4) via the four NTT primes of main.cc:245-248 + CRT, "
                               "n=2^17 per input, primes dealt to ranks, one all-to-all of residue slices per step",
                   "p": P64, "n": n, "batch_total": total, "batch_this_rank": nb, "primes_this_rank": k,
                   "layout": {"prime_groups": sh.gp, "batch_groups": sh.gb},
                   "parallelism": f"{sh.gp} prime group(s) x {sh.gb} batch group(s)"},
        "butterflies_per_unit": 4 * 3 * (L // 2) * 18,
        "bytes_per_step": bytes_per_step, "launches_per_step": launches, "prof": prof, "roofline_kind": f"crt_b{nb}_k{k}",
        "kernels_overlap": k > 1,
        "extra": {"us_per_mult_whole_job": elapsed / (total * args.steps) * 1e6, "exchange_bytes_per_rank": exch},
        "cpu": (lambda: cpu_baseline_main_cc(n, P64, extras=False)),
    }


def run_fourstep(R, args):
    import numpy as np
    import ntt_amd
    from ntt_amd.distributed import FourStepSharded, HipEngine
    torch = R.torch
    logL = args.log_len or 24
    L = 1 << logL
    g = torch.Generator(device=R.dev)
    g.manual_seed(77)                               # every rank draws the same whole vectors and keeps its slab
    stream = torch.cuda.current_stream().cuda_stream
    if R.world == 1:
        a, b = rand_padded(torch, g, R.dev, 1, L // 2, L, P32), rand_padded(torch, g, R.dev, 1, L // 2, L, P32)
        c = torch.empty_like(a)
        ntt_amd.reserve(L, 1, 1)

        def step():
            ntt_amd.dev_poly_multiply_u32(a.data_ptr(), b.data_ptr(), c.data_ptr(), L // 2, P32, 1, stream)

        step()
        torch.cuda.synchronize()
        if not args.no_check:
            # a(1) b(1) = c(1) and a(-1) b(-1) = c(-1) modulo p: every coefficient of the product takes part
            sg = torch.ones(L, dtype=torch.int64, device=R.dev)
            sg[1::2] = -1
            for s in (torch.ones_like(sg), sg):
                ea, eb, ec = (int((x[0].to(torch.int64) * s).sum().item()) % P32 for x in (a, b, c))
                if ea * eb % P32 != ec:
                    raise SystemExit(json.dumps({"error": "GPU result fails the evaluation check"}))
        plan = ntt_amd.plan_info(L)
        # three passes: outer column pass (zero half never read: 1 + 2 = 3 L words), inner column pass on both copies
        # (2 + 2 = 4 L), rows (3 L), inner inverse (2 L), outer inverse (2 L)
        bytes_per_step = {"col_fwd": (3 + 4) * 4 * L, "row_mul": 3 * 4 * L, "col_inv": (2 + 2) * 4 * L}
        launches = {"col_fwd": 2, "row_mul": 1, "col_inv": 2}
        par, exch, cfg_extra = "one GPU, three-pass plan, no exchange", 0, {"passes": plan}
    else:
        fs = FourStepSharded(R.comm(), HipEngine(), logL, P32)
        va = torch.randint(0, P32, (L // 2,), generator=g, device=R.dev, dtype=torch.int64).to(torch.int32)
        vb = torch.randint(0, P32, (L // 2,), generator=g, device=R.dev, dtype=torch.int64).to(torch.int32)
        half = fs.T // 2
        xa = torch.zeros((fs.T, fs.cols), dtype=torch.int32, device=R.dev)
        xb = torch.zeros_like(xa)
        xa[:half] = va.view(half, fs.S)[:, fs.col0: fs.col0 + fs.cols]
        xb[:half] = vb.view(half, fs.S)[:, fs.col0: fs.col0 + fs.cols]
        wa, wb = xa.clone(), xb.clone()

        def step():
            wa.copy_(xa)
            wb.copy_(xb)
            return fs.multiply(wa, wb, zero_hi=True)

        xc = step()
        torch.cuda.synchronize()
        if not args.no_check:
            # evaluation at x = 1: partial sums of every rank's slab, combined over the ranks
            t = torch.tensor([int(xa.to(torch.int64).sum().item()) % P32, int(xb.to(torch.int64).sum().item()) % P32,
                              int(xc.to(torch.int64).sum().item()) % P32], dtype=torch.int64)
            tt = t.clone() if R.rehearsal else t.to(R.dev)
            R.dist.all_reduce(tt)
            ea, eb, ec = (int(v) % P32 for v in tt.cpu().tolist())
            if ea * eb % P32 != ec:
                raise SystemExit(json.dumps({"error": "GPU result fails the evaluation check"}))
        rows_len = fs.S
        rplan = ntt_amd.plan_info(rows_len)
        slab = fs.T * fs.cols                       # words of one slab = L / G
        if len(rplan) == 1:
            bytes_per_step = {"col_fwd": 2 * 6 * slab, "row_mul": 12 * slab, "col_inv": 8 * slab}
            launches = {"col_fwd": 2, "row_mul": 1, "col_inv": 1}
        else:
            bytes_per_step = {"col_fwd": (2 * 6 + 16) * slab, "row_mul": 12 * slab, "col_inv": (8 + 8) * slab}
            launches = {"col_fwd": 3, "row_mul": 1, "col_inv": 2}
        par = f"columns split over {R.world} ranks, one all-to-all per vector (a, b, product)"
        exch = int(3 * slab * 4 * (R.world - 1) / R.world)
        cfg_extra = {"log2_t": fs.log_t, "row_passes": rplan}
    elapsed = R.timed(step, args.steps, args.warmup, events=not args.no_events)
    prof = ntt_amd.profile_read() if not args.no_events else {}
    return {
        "metric": f"poly-mults/sec, one product of length 2^{logL} (single 32-bit prime)", "unit": "poly-mults/s", "dtype": "u32",
        "units": args.steps, "elapsed": elapsed, "scaling": "strong",
        "config": dict({"workload": f"configs[4]: ONE product of length 2^{logL} (n=2^{logL - 1} per input), prime 469762049, four-step NTT",
                        "p": P32, "log2_len": logL, "parallelism": par}, **cfg_extra),
        "butterflies_per_unit": 3 * (L // 2) * logL,
        "bytes_per_step": bytes_per_step, "launches_per_step": launches, "prof": prof, "roofline_kind": f"fourstep_2p{logL}_g{R.world}",
        "extra": {"ms_per_product": elapsed / args.steps * 1e3, "exchange_bytes_per_rank": exch},
        "cpu": (lambda: cpu_baseline_long(logL, P32)),
    }


def run_dry(R, args):
    elapsed = R.timed(lambda: None, args.steps, args.warmup)
    return {"metric": "none (dry run of the launch and timing plumbing)", "unit": "steps/s", "dtype": "none", "units": 0,
            "elapsed": elapsed, "scaling": "weak", "config": {"workload": f"dry run of --workload {args.workload}"},
            "butterflies_per_unit": 0, "bytes_per_step": {}, "launches_per_step": {}, "prof": {}, "roofline_kind": None,
            "extra": {"dry_run": True}, "cpu": None}


def run_rank(args):
    R = Rank(args)
    import ntt_amd
    if not R.dry:
        ntt_amd.init(R.device_index)
    try:
        fn = run_dry if R.dry else {"single": run_single, "crt": run_crt, "batch": run_batch, "fourstep": run_fourstep}[args.workload]
        res = fn(R, args)
        if R.rank == 0:
            units, elapsed = res["units"], res["elapsed"]
            value = units / elapsed if units else None
            out = {"metric": res["metric"], "value": value, "unit": res["unit"], "n_gpus": R.world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                   "scaling": res["scaling"], "vs_baseline": None, "dtype": res["dtype"], "data": "synthetic",
                   "config": res["config"], "workload": args.workload, "backend": R.backend, "rccl_ranks": R.rccl_ranks}
            out["clock_ramp_steps"] = getattr(R, "ramp_steps", 0)
            if R.rehearsal and R.world > 1:
                out["rehearsal"] = "NTTX_BENCH_ONE_DEVICE=1: every rank ran on GPU 0; not a multi-GPU measurement"
            if units and res["butterflies_per_unit"]:
                out["ns_per_butterfly"] = elapsed / (units * res["butterflies_per_unit"]) * 1e9
            out.update(res["extra"])
            out["counter_record_key"] = res["roofline_kind"]
            roof, per_kernel = roofline_of(res["prof"], res["bytes_per_step"], res["launches_per_step"], res["roofline_kind"])
            if roof is not None and res.get("kernels_overlap"):
                # the single-prime products run side by side on forked streams: event times of kernels that share the GPU
                # are not separable, so the figure is every kernel's moved bytes over the step's wall time
                tot_b = sum(res["bytes_per_step"].values())
                step_s = elapsed / args.steps
                wrec, wstate = recorded_counters("traffic", res["roofline_kind"])
                wtraffic = (sum(wrec["kernels"].get(k, 0) * res["launches_per_step"].get(k, 1) for k in res["launches_per_step"])
                            if wrec else None)
                vrec, vstate = recorded_counters("valu", res["roofline_kind"])
                roof = {"bound": "hbm", "kernel": "whole step (the primes' kernels overlap on side streams; per-kernel event times are not separable)",
                        "achieved": tot_b / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tot_b / step_s / 1e9 / HBM_PEAK_GBS,
                        "traffic": wtraffic, "traffic_record": wstate,
                        "bytes": "HBM bytes all kernels of a step move (reads + writes, each word once)",
                        "algorithmic_bytes_per_launch": tot_b, "avg_launch_ms": step_s * 1e3, "launches_timed": args.steps}
                if vrec:
                    ik, ck = vrec.get("kernels", {}), vrec.get("clock_GHz", {})
                    busy_s = sum(ik[k] * res["launches_per_step"].get(k, 1) * VALU_CYCLES_PER_INSTRUCTION / (SIMDS * ck[k] * 1e9)
                                 for k in res["launches_per_step"] if ik.get(k) and ck.get(k))
                    roof["valu"] = {"record": vstate, "whole_step": {
                        "wave_instructions_per_step": sum(ik[k] * res["launches_per_step"].get(k, 1) for k in res["launches_per_step"] if ik.get(k)),
                        "issue_busy_ms_per_step_by_model": busy_s * 1e3, "step_ms": step_s * 1e3, "issue_busy_frac": busy_s / step_s}}
                    if busy_s / step_s > roof["frac"]:
                        roof["bound"] = "valu"
                per_kernel = {k + " (overlapped)": v for k, v in per_kernel.items()}
            if roof is not None:
                if res.get("achievable_copy_GBs"):
                    roof["achievable_copy_GBs"] = res["achievable_copy_GBs"]
                    roof["frac_of_achievable_copy"] = roof["achieved"] / res["achievable_copy_GBs"]
                    roof["whole_step"]["frac_of_achievable_copy"] = roof["whole_step"]["achieved"] / res["achievable_copy_GBs"]
                out["roofline"] = roof
                out["kernels_ms_per_launch"] = per_kernel
            elif not R.dry:
                out["roofline"] = None
            if R.world == 1 and not args.no_cpu and res["cpu"] is not None:
                out["cpu_baseline"] = res["cpu"]()
                if value:
                    out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            print(json.dumps(out), flush=True)
    finally:
        R.finish()
        if not R.dry:
            ntt_amd.shutdown()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    run_rank(args)


if __name__ == "__main__":
    main()
