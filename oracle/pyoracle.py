"""ctypes front-end of the CPU oracle (liboracle.so) and of the compiled reference
programs (oracle/_ref/*.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package ntt_amd never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_longlong)

CRT_PRIMES = (469762049, 998244353, 1004535809, 1224736769)  # main.cc:245-248


def _ptr(a, t):
    return a.ctypes.data_as(t)


class _MontT(C.Structure):
    _fields_ = [("N", C.c_uint64), ("R", C.c_uint64), ("logR", C.c_int), ("N_inv_neg", C.c_uint64), ("R2", C.c_uint64)]


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(os.path.join(HERE, "liboracle.so")) or os.path.exists("/root/reference/main.cc"):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)


class Oracle:
    """The C restatement (oracle/ntt_oracle.c)."""

    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        L.oracle_barrett_reduce.restype = C.c_uint64
        L.oracle_barrett_reduce.argtypes = [C.c_uint64, C.c_uint64]
        L.oracle_barrett_mul_mod.restype = C.c_uint64
        L.oracle_barrett_mul_mod.argtypes = [C.c_uint64] * 3
        L.oracle_ntt_iter_barrett.argtypes = [_u32p, C.c_int, C.c_uint64, C.c_int, C.c_int]
        L.oracle_ntt_compute.argtypes = [_u32p, _u32p, _u32p, C.c_int, C.c_uint64, C.c_int]
        L.oracle_crt_combine.argtypes = [C.POINTER(_u32p), _u64p, C.c_int, C.c_int, C.c_int64]
        L.oracle_poly_multiply_crt.argtypes = [_u64p, _u64p, _u64p, C.c_int, C.c_int64]
        L.oracle_mont_init.argtypes = [C.POINTER(_MontT), C.c_uint64, C.c_uint64]
        for f in (L.oracle_mont_to, L.oracle_mont_from):
            f.restype = C.c_uint64
            f.argtypes = [C.POINTER(_MontT), C.c_uint64]
        L.oracle_mont_mul.restype = C.c_uint64
        L.oracle_mont_mul.argtypes = [C.POINTER(_MontT), C.c_uint64, C.c_uint64]
        L.oracle_mont_modmul.argtypes = [C.POINTER(_MontT), C.c_uint64, C.c_uint64, _u64p]
        L.oracle_ntt_iter_mont.argtypes = [_i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_MontT)]
        L.oracle_poly_multiply_mont.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int]
        L.oracle_poly_multiply_naive.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int]
        L.oracle_poly_multiply_naive_u64.argtypes = [_u64p, _u64p, _u64p, C.c_int, C.c_uint64]
        L.oracle_fill_random.argtypes = [_u64p, C.c_longlong, C.c_uint64, C.c_uint64]
        L.oracle_max_threads.restype = C.c_int
        L.pool_poly_multiply_mont.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int]

    # --- Barrett / transform level (main.cc) ---
    def barrett_reduce(self, mod, a):
        return int(self.lib.oracle_barrett_reduce(mod, a))

    def barrett_mul_mod(self, mod, a, b):
        return int(self.lib.oracle_barrett_mul_mod(mod, a, b))

    def ntt_iter_barrett(self, a, p, invert, root=3):
        v = np.ascontiguousarray(a, dtype=np.uint32).copy()
        self.lib.oracle_ntt_iter_barrett(_ptr(v, _u32p), len(v), p, root, int(bool(invert)))
        return v

    def ntt_compute(self, a, b, p, root=3):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        r = np.empty_like(a)
        rc = self.lib.oracle_ntt_compute(_ptr(a, _u32p), _ptr(b, _u32p), _ptr(r, _u32p), len(a), p, root)
        assert rc == 0
        return r

    def crt_combine(self, residues, nout, p):
        vs = [np.ascontiguousarray(r, dtype=np.uint32) for r in residues]
        arr = (_u32p * 4)(*[_ptr(v, _u32p) for v in vs])
        out = np.zeros(nout, dtype=np.uint64)
        self.lib.oracle_crt_combine(arr, _ptr(out, _u64p), len(vs[0]), nout, p)
        return out

    def poly_multiply_crt(self, a, b, p):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        n = len(a)
        out = np.zeros(2 * n - 1, dtype=np.uint64)
        rc = self.lib.oracle_poly_multiply_crt(_ptr(a, _u64p), _ptr(b, _u64p), _ptr(out, _u64p), n, p)
        assert rc == 0
        return out

    # --- Montgomery level (monttry2.cpp) ---
    def mont(self, R, N):
        m = _MontT()
        rc = self.lib.oracle_mont_init(C.byref(m), R, N)
        return rc, m

    def mont_to(self, m, a):
        return int(self.lib.oracle_mont_to(C.byref(m), a))

    def mont_from(self, m, a):
        return int(self.lib.oracle_mont_from(C.byref(m), a))

    def mont_mul(self, m, a, b):
        return int(self.lib.oracle_mont_mul(C.byref(m), a, b))

    def mont_modmul(self, m, a, b):
        out = C.c_uint64(0)
        rc = self.lib.oracle_mont_modmul(C.byref(m), a, b, C.byref(out))
        return rc, int(out.value)

    def poly_multiply_mont(self, a, b, p):
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        n = len(a)
        out = np.zeros(2 * n - 1, dtype=np.int32)
        rc = self.lib.oracle_poly_multiply_mont(_ptr(a, _i32p), _ptr(b, _i32p), _ptr(out, _i32p), n, p)
        return rc, out

    def poly_multiply_naive(self, a, b, p):
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        n = len(a)
        out = np.zeros(2 * n - 1, dtype=np.int32)
        self.lib.oracle_poly_multiply_naive(_ptr(a, _i32p), _ptr(b, _i32p), _ptr(out, _i32p), n, p)
        return out

    def poly_multiply_naive_u64(self, a, b, p):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        n = len(a)
        out = np.zeros(2 * n - 1, dtype=np.uint64)
        self.lib.oracle_poly_multiply_naive_u64(_ptr(a, _u64p), _ptr(b, _u64p), _ptr(out, _u64p), n, p)
        return out

    def fill_random(self, count, seed, p):
        """splitmix64 stream reduced mod p (the seeded synthetic inputs of tests and bench)."""
        out = np.empty(count, dtype=np.uint64)
        self.lib.oracle_fill_random(_ptr(out, _u64p), count, seed, p)
        return out

    def pool_poly_multiply(self, a, b, p, threads=8):
        """cpu_pool.c: monttry2.cpp / pthread_normal.cpp's stage-parallel product on a standing thread pool."""
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        n = len(a)
        out = np.zeros(2 * n - 1, dtype=np.int32)
        rc = self.lib.pool_poly_multiply_mont(_ptr(a, _i32p), _ptr(b, _i32p), _ptr(out, _i32p), n, p, threads)
        return rc, out

    def max_threads(self):
        return int(self.lib.oracle_max_threads())


def have_ref() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libref_main.so"))


class RefCRT:
    """main.cc / main_pthread.cc compiled in place (oracle/_ref/libref_main*.so)."""

    def __init__(self, which="main"):
        self.which = which
        L = self.lib = C.CDLL(os.path.join(HERE, "_ref", f"libref_{which}.so"))
        L.ref_ntt_iter_barrett.argtypes = [_u32p, C.c_int, C.c_uint64, C.c_int, C.c_int]
        L.ref_barrett_reduce.restype = C.c_uint64
        L.ref_barrett_reduce.argtypes = [C.c_uint64, C.c_uint64]
        L.ref_barrett_mul_mod.restype = C.c_uint64
        L.ref_barrett_mul_mod.argtypes = [C.c_uint64] * 3
        if which == "main":
            L.ref_ntt_compute.argtypes = [_u32p, _u32p, _u32p, C.c_int, C.c_uint64, C.c_int]

    def ntt_iter_barrett(self, a, p, invert, root=3):
        v = np.ascontiguousarray(a, dtype=np.uint32).copy()
        self.lib.ref_ntt_iter_barrett(_ptr(v, _u32p), len(v), p, root, int(bool(invert)))
        return v

    def ntt_compute(self, a, b, p, root=3):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        r = np.empty_like(a)
        self.lib.ref_ntt_compute(_ptr(a, _u32p), _ptr(b, _u32p), _ptr(r, _u32p), len(a), p, root)
        return r


class RefMont:
    """monttry2.cpp / pthread_normal.cpp compiled in place."""

    def __init__(self, which="monttry2"):
        L = self.lib = C.CDLL(os.path.join(HERE, "_ref", f"libref_{which}.so"))
        L.ref_mont_probe.argtypes = [C.c_uint64, C.c_uint64]
        for f in (L.ref_mont_to, L.ref_mont_from):
            f.restype = C.c_uint64
            f.argtypes = [C.c_uint64] * 3
        L.ref_mont_mul.restype = C.c_uint64
        L.ref_mont_mul.argtypes = [C.c_uint64] * 4
        L.ref_mont_modmul.argtypes = [C.c_uint64] * 4 + [_u64p]
        L.ref_poly_multiply_mont.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int]
        L.ref_poly_multiply_naive.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int]

    def poly_multiply(self, a, b, p):
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        n = len(a)
        out = np.zeros(2 * n - 1, dtype=np.int32)
        self.lib.ref_poly_multiply_mont(_ptr(a, _i32p), _ptr(b, _i32p), _ptr(out, _i32p), n, p)
        return out

    def poly_multiply_naive(self, a, b, p):
        a = np.ascontiguousarray(a, dtype=np.int32).copy()
        b = np.ascontiguousarray(b, dtype=np.int32).copy()
        n = len(a)
        out = np.zeros(2 * n - 1, dtype=np.int32)
        self.lib.ref_poly_multiply_naive(_ptr(a, _i32p), _ptr(b, _i32p), _ptr(out, _i32p), n, p)
        return out


def write_in_file(path, a, b, p):
    """Harness input format (fRead, main.cc:85-93): n p, then a, then b, whitespace separated."""
    n = len(a)
    with open(path, "w") as f:
        f.write(f"{n} {p}\n")
        f.write(" ".join(str(int(x)) for x in a) + "\n")
        f.write(" ".join(str(int(x)) for x in b) + "\n")


def write_out_file(path, ab):
    """fWrite format (main.cc:104-107): one coefficient per line."""
    with open(path, "w") as f:
        f.write("".join(f"{int(x)}\n" for x in ab))


def run_ref_main(cases, which="ref_main", threads=None, reps=1):
    """Run the compiled reference driver (its whole main loop over ids 0..4) on up to five
    (a, b, p) cases.  Returns (outputs per id, list of latency-ms per id per rep, stdout).
    Ids without a case get a copy of case 0 so the loop has something to read."""
    exe = os.path.join(HERE, "_ref", which)
    outs, lat, text = {}, {}, ""
    with tempfile.TemporaryDirectory(prefix="nttref_") as d:
        data = os.path.join(d, "nttdata")
        os.makedirs(data)
        os.makedirs(os.path.join(d, "files"))
        full = list(cases) + [cases[0]] * (5 - len(cases))
        for i, (a, b, p) in enumerate(full):
            write_in_file(os.path.join(data, f"{i}.in"), a, b, p)
            open(os.path.join(data, f"{i}.out"), "w").write("0\n")
        env = dict(os.environ, NTT_REF_DATA_DIR=data)
        if threads:
            env["OMP_NUM_THREADS"] = str(threads)
        for _ in range(reps):
            r = subprocess.run([exe], cwd=d, env=env, capture_output=True, text=True, check=True)
            text = r.stdout
            k = 0
            for line in r.stdout.splitlines():
                if line.startswith("average latency"):
                    ms = float(line.split(":")[1].split("(")[0])
                    lat.setdefault(k, []).append(ms)
                    k += 1
        for i in range(len(cases)):
            outs[i] = np.loadtxt(os.path.join(d, "files", f"{i}.out"), dtype=np.uint64, ndmin=1)
    return outs, lat, text
