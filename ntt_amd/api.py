"""ctypes binding of libnttx.so (include/nttx.h).  Mirrors the reference's call shapes:
poly_multiply(a, b, n, p) for both harness flavours ((int,int) of monttry2.cpp:344 and
(uint64_t,int64_t) of main.cc:78-128), ntt() for ntt_iter_barrett (main.cc:143)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NTTX_LIB") or os.path.join(_HERE, "csrc", "libnttx.so")  # NTTX_LIB: experiment builds

CRT_PRIMES = (469762049, 998244353, 1004535809, 1224736769)  # main.cc:245-248
PATH_DIRECT32, PATH_CRT4, PATH_DIRECT64 = 0, 1, 2

_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)

# every symbol include/nttx.h declares: (name, restype, argtypes)
_SIGS = [
    ("nttx_init", C.c_int, [C.c_int]),
    ("nttx_shutdown", None, []),
    ("nttx_last_error", C.c_char_p, []),
    ("nttx_host_alloc", C.c_void_p, [C.c_size_t]),
    ("nttx_host_free", None, [C.c_void_p]),
    ("nttx_path_for", C.c_int, [C.c_uint64, C.c_int]),
    ("nttx_poly_multiply_i32", C.c_int, [_i32p, _i32p, _i32p, C.c_int, C.c_int32]),
    ("nttx_poly_multiply_u64", C.c_int, [_u64p, _u64p, _u64p, C.c_int, C.c_uint64]),
    ("nttx_poly_multiply_direct_u64", C.c_int, [_u64p, _u64p, _u64p, C.c_int, C.c_uint64]),
    ("nttx_poly_multiply_crt_u64", C.c_int, [_u64p, _u64p, _u64p, C.c_int, C.c_uint64]),
    ("nttx_poly_multiply_batch_u32", C.c_int, [_u32p, _u32p, _u32p, C.c_int, C.c_uint32, C.c_int]),
    ("nttx_ntt_u32", C.c_int, [_u32p, C.c_int, C.c_uint32, C.c_int]),
    ("nttx_cyclic_multiply_u32", C.c_int, [_u32p, _u32p, _u32p, C.c_int, C.c_uint32]),
    ("nttx_crt_combine_u64", C.c_int, [C.POINTER(_u32p), _u64p, C.c_int, C.c_int, C.c_uint64]),
    ("nttx_dev_cyclic_multiply_u32", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p]),
    ("nttx_dev_cyclic_multiply_pieces_u32", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p]),
    ("nttx_dev_negacyclic_multiply_u32", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p]),
    ("nttx_dev_poly_multiply_u32", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p]),
    ("nttx_dev_poly_multiply_u32_static", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p]),
    ("nttx_dev_crt_multiply_u64", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p]),
    ("nttx_dev_ntt_u32", C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("nttx_crt_prime", C.c_int, [C.c_int, _u32p]),
    ("nttx_crt_primes_for", C.c_int, [C.c_uint64, C.c_int, _u32p]),
    ("nttx_dev_crt_residues_u64", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("nttx_dev_crt_residues_sliced_u64", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("nttx_dev_crt_recombine_u64", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_uint64, C.c_void_p]),
    ("nttx_dev_colpass_u32", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p]),
    ("nttx_dev_direct_multiply_u64", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p]),
    ("nttx_dev_ntt_u64", C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_void_p]),
    ("nttx_reserve", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("nttx_reserve_u64", C.c_int, [C.c_int, C.c_int]),
    ("nttx_profile_enable", C.c_int, [C.c_int]),
    ("nttx_profile_reset", C.c_int, []),
    ("nttx_profile_read", C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    ("nttx_plan_info", C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    ("nttx_dev_negacyclic_multiply_u64", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p]),
    ("nttx_plan_info_u64", C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_int)]),
    ("nttx_dev_stream_copy", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("nttx_init_devices", C.c_int, [C.POINTER(C.c_int), C.c_int]),
    ("nttx_device_count", C.c_int, []),
    ("nttx_set_device", C.c_int, [C.c_int]),
    ("nttx_deal_batch", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
]


class NttxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nttx error {code}: {msg}")
        self.code = code


_lib = None


def exported_symbols():
    return [s[0] for s in _SIGS]


def lib():
    """Load libnttx.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if os.environ.get("NTTX_NO_TORCH") != "1":
            # One HIP runtime per process: PyTorch's wheel carries its own libamdhip64.so.7.
            # Loading it first makes libnttx.so bind to that same runtime instead of a second copy.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(LIB_PATH):
            raise NttxError(-100, f"{LIB_PATH} is missing: run `make -C ntt_amd/csrc` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        for name, res, args in _SIGS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise NttxError(rc, lib().nttx_last_error().decode())
    return rc


def init(device: int = 0):
    _check(lib().nttx_init(device))


def init_devices(devices):
    """One process driving several GPUs: a context per listed device, the first one current."""
    arr = (C.c_int * len(devices))(*devices)
    _check(lib().nttx_init_devices(arr, len(devices)))


def device_count() -> int:
    return _check(lib().nttx_device_count())


def set_device(device: int):
    _check(lib().nttx_set_device(device))


def deal_batch(batch: int, ndev: int, k: int):
    """The share [lo, hi) of a batch that device k of ndev gets from poly_multiply_batch (needs no device)."""
    lo, hi = C.c_int(0), C.c_int(0)
    _check(lib().nttx_deal_batch(batch, ndev, k, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def shutdown():
    if _lib is not None:
        _lib.nttx_shutdown()


def host_alloc(count: int, dtype):
    """A page-locked numpy array of `count` elements (host-buffer calls move it by DMA); release with host_free."""
    dt = np.dtype(dtype)
    ptr = lib().nttx_host_alloc(max(1, count) * dt.itemsize)
    if not ptr:
        raise NttxError(-3, lib().nttx_last_error().decode())
    buf = (C.c_char * (count * dt.itemsize)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dt, count=count)
    arr.flags.writeable = True
    return arr


def host_free(arr):
    lib().nttx_host_free(arr.ctypes.data)


def path_for(p: int, n: int) -> int:
    return _check(lib().nttx_path_for(p, n))


def plan_info(length: int):
    lt = (C.c_int * 3)()
    k = _check(lib().nttx_plan_info(length, lt))
    return list(lt[:k])


def plan_info_u64(length: int, p: int):
    lt = (C.c_int * 3)()
    k = _check(lib().nttx_plan_info_u64(length, p, lt))
    return list(lt[:k])


def reserve(length: int, batch: int, nprimes: int = 1):
    _check(lib().nttx_reserve(length, batch, nprimes))


def _arr(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


def poly_multiply(a, b, n: int, p: int, dtype=None):
    """ab[0..2n-2] = a*b mod p.  dtype int32 -> the (int,int) harness flavour, uint64 -> the
    (uint64_t,int64_t) flavour; default: int32 if p < 2^31 else uint64."""
    if dtype is None:
        dtype = np.int32 if p < (1 << 31) else np.uint64
    dtype = np.dtype(dtype)
    L = lib()
    if dtype == np.int32:
        a, b = _arr(a, np.int32)[:n], _arr(b, np.int32)[:n]
        ab = np.zeros(max(2 * n - 1, 0), dtype=np.int32)
        _check(L.nttx_poly_multiply_i32(a.ctypes.data_as(_i32p), b.ctypes.data_as(_i32p), ab.ctypes.data_as(_i32p), n, p))
    elif dtype == np.uint64:
        a, b = _arr(a, np.uint64)[:n], _arr(b, np.uint64)[:n]
        ab = np.zeros(max(2 * n - 1, 0), dtype=np.uint64)
        _check(L.nttx_poly_multiply_u64(a.ctypes.data_as(_u64p), b.ctypes.data_as(_u64p), ab.ctypes.data_as(_u64p), n, p))
    else:
        raise TypeError("dtype must be int32 or uint64")
    return ab


def poly_multiply_crt(a, b, n: int, p: int):
    """Force the 4-prime CRT path (main.cc:286-338)."""
    a, b = _arr(a, np.uint64)[:n], _arr(b, np.uint64)[:n]
    ab = np.zeros(max(2 * n - 1, 0), dtype=np.uint64)
    _check(lib().nttx_poly_multiply_crt_u64(a.ctypes.data_as(_u64p), b.ctypes.data_as(_u64p), ab.ctypes.data_as(_u64p), n, p))
    return ab


def poly_multiply_direct64(a, b, n: int, p: int):
    """One 64-bit-lane transform modulo p itself (NTT-friendly p between 2^31 and 2^62)."""
    a, b = _arr(a, np.uint64)[:n], _arr(b, np.uint64)[:n]
    ab = np.zeros(max(2 * n - 1, 0), dtype=np.uint64)
    _check(lib().nttx_poly_multiply_direct_u64(a.ctypes.data_as(_u64p), b.ctypes.data_as(_u64p), ab.ctypes.data_as(_u64p), n, p))
    return ab


def poly_multiply_batch(a, b, p: int):
    """a, b: [batch][n] uint32 -> [batch][2n-1]."""
    a, b = _arr(a, np.uint32), _arr(b, np.uint32)
    batch, n = a.shape
    ab = np.zeros((batch, 2 * n - 1), dtype=np.uint32)
    _check(lib().nttx_poly_multiply_batch_u32(a.ctypes.data_as(_u32p), b.ctypes.data_as(_u32p), ab.ctypes.data_as(_u32p), n, p, batch))
    return ab


def ntt(x, p: int, invert: bool = False):
    """Natural-order transform of ntt_iter_barrett (main.cc:143-182), root 3, no 1/n."""
    v = _arr(x, np.uint32).copy()
    _check(lib().nttx_ntt_u32(v.ctypes.data_as(_u32p), len(v), p, int(bool(invert))))
    return v


def cyclic_multiply(a, b, p: int):
    """ntt_compute_omp (main.cc:185-209): cyclic product of two length-2^k residue vectors."""
    a, b = _arr(a, np.uint32), _arr(b, np.uint32)
    c = np.empty_like(a)
    _check(lib().nttx_cyclic_multiply_u32(a.ctypes.data_as(_u32p), b.ctypes.data_as(_u32p), c.ctypes.data_as(_u32p), len(a), p))
    return c


def crt_combine(residues, nout: int, p: int):
    """CRT recombination of the four residue vectors (main.cc:313-338)."""
    vs = [_arr(r, np.uint32) for r in residues]
    arr = (_u32p * 4)(*[v.ctypes.data_as(_u32p) for v in vs])
    out = np.zeros(nout, dtype=np.uint64)
    _check(lib().nttx_crt_combine_u64(arr, out.ctypes.data_as(_u64p), len(vs[0]), nout, p))
    return out


# ---- device-resident calls: arguments are raw device pointers (e.g. torch tensor.data_ptr()) ----
def dev_cyclic_multiply_u32(d_a: int, d_b: int, d_c: int, length: int, p: int, batch: int, stream: int = 0):
    _check(lib().nttx_dev_cyclic_multiply_u32(d_a, d_b, d_c, length, p, batch, stream))


def dev_cyclic_multiply_pieces_u32(d_a: int, d_b: int, d_c: int, length: int, pieces: int, rows: int, p: int, stream: int = 0):
    """Cyclic products of [pieces][rows][length / pieces] arrays (the layout an all-to-all leaves rows in)."""
    _check(lib().nttx_dev_cyclic_multiply_pieces_u32(d_a, d_b, d_c, length, pieces, rows, p, stream))


def dev_negacyclic_multiply_u32(d_a: int, d_b: int, d_c: int, length: int, p: int, batch: int, stream: int = 0):
    """c = a*b mod (x^length + 1, p) on [batch][length] u32 device arrays."""
    _check(lib().nttx_dev_negacyclic_multiply_u32(d_a, d_b, d_c, length, p, batch, stream))


def dev_poly_multiply_u32(d_a: int, d_b: int, d_c: int, n: int, p: int, batch: int, stream: int = 0):
    """Linear product of zero-padded [batch][len] device arrays (len = pow2 >= 2n)."""
    _check(lib().nttx_dev_poly_multiply_u32(d_a, d_b, d_c, n, p, batch, stream))


def dev_poly_multiply_u32_static(d_a: int, d_b: int, d_c: int, n: int, p: int, batch: int, stream: int = 0):
    """dev_poly_multiply_u32 without any device allocation (fixed scratch inside the library; batch*len <= 2^22)."""
    _check(lib().nttx_dev_poly_multiply_u32_static(d_a, d_b, d_c, n, p, batch, stream))


def dev_crt_multiply_u64(d_a: int, d_b: int, d_ab: int, n: int, p: int, batch: int, stream: int = 0):
    _check(lib().nttx_dev_crt_multiply_u64(d_a, d_b, d_ab, n, p, batch, stream))


def dev_direct_multiply_u64(d_a: int, d_b: int, d_c: int, n: int, p: int, batch: int, stream: int = 0):
    """Linear product of zero-padded [batch][len] u64 device arrays on the 64-bit-lane path."""
    _check(lib().nttx_dev_direct_multiply_u64(d_a, d_b, d_c, n, p, batch, stream))


def dev_ntt_u64(d_x: int, length: int, p: int, invert: bool, batch: int, stream: int = 0):
    _check(lib().nttx_dev_ntt_u64(d_x, length, p, int(bool(invert)), batch, stream))


def reserve_u64(length: int, batch: int):
    _check(lib().nttx_reserve_u64(length, batch))


def dev_ntt_u32(d_x: int, length: int, p: int, invert: bool, bit_reversed_out: bool, batch: int, stream: int = 0):
    _check(lib().nttx_dev_ntt_u32(d_x, length, p, int(bool(invert)), int(bool(bit_reversed_out)), batch, stream))


def crt_prime(k: int) -> int:
    v = C.c_uint32(0)
    _check(lib().nttx_crt_prime(k, C.byref(v)))
    return v.value


def dev_negacyclic_multiply_u64(d_a: int, d_b: int, d_c: int, length: int, p: int, batch: int, stream: int = 0):
    """c = a*b mod (x^length + 1, p) on [batch][length] u64 device arrays (64-bit lanes, 2^31 <= p < 2^62)."""
    _check(lib().nttx_dev_negacyclic_multiply_u64(d_a, d_b, d_c, length, p, batch, stream))


def crt_primes_for(p: int, n: int):
    """The CRT moduli the engine uses for n coefficients modulo p (as few of main.cc:245-248's four as keep the
    product exact); needs no device."""
    v = (C.c_uint32 * 4)()
    k = _check(lib().nttx_crt_primes_for(p, n, v))
    return [int(v[i]) for i in range(k)]


def dev_crt_residues_u64(d_a: int, d_b: int, d_res: int, n: int, prime_mask: int, batch: int, stream: int = 0):
    """Residues of the products modulo the CRT primes in prime_mask -> [primes][batch][len] u32."""
    _check(lib().nttx_dev_crt_residues_u64(d_a, d_b, d_res, n, prime_mask, batch, stream))


def dev_crt_residues_sliced_u64(d_a: int, d_b: int, d_res: int, n: int, prime_mask: int, batch: int, slices: int, stream: int = 0):
    """Residues laid out [slice][primes][batch][len / slices]: the send buffer of the exchange between `slices` processes."""
    _check(lib().nttx_dev_crt_residues_sliced_u64(d_a, d_b, d_res, n, prime_mask, batch, slices, stream))


def dev_crt_recombine_u64(d_r, d_out: int, count: int, p: int, stream: int = 0):
    """CRT recombination of a slice: d_r = four device pointers to `count` u32 residues each."""
    _check(lib().nttx_dev_crt_recombine_u64(d_r[0], d_r[1], d_r[2], d_r[3], d_out, count, p, stream))


def dev_colpass_u32(d_x: int, log_len: int, log_t: int, col0: int, cols: int, p: int, invert: bool, zero_hi: bool = False,
                    stream: int = 0):
    """One column pass over this process's [2^log_t][cols] slab of a length-2^log_len transform."""
    _check(lib().nttx_dev_colpass_u32(d_x, log_len, log_t, col0, cols, p, int(bool(invert)), int(bool(zero_hi)), stream))


KERNEL_KINDS = ("col_fwd", "row_mul", "col_inv", "row_fwd", "row_inv", "split", "crt", "misc")


def dev_stream_copy(d_dst: int, d_src: int, nbytes: int, stream: int = 0):
    """Plain streaming copy by the library's own 16-bytes-per-lane kernel (the achievable-bandwidth yardstick)."""
    _check(lib().nttx_dev_stream_copy(d_dst, d_src, nbytes, stream))


def profile_enable(on: bool = True):
    _check(lib().nttx_profile_enable(int(bool(on))))


def profile_reset():
    _check(lib().nttx_profile_reset())


def profile_read():
    """{kind: (total_ms, launches)} of the kernels timed since the last reset."""
    out = {}
    for k, name in enumerate(KERNEL_KINDS):
        ms, cnt = C.c_double(0), C.c_longlong(0)
        _check(lib().nttx_profile_read(k, C.byref(ms), C.byref(cnt)))
        out[name] = (ms.value, cnt.value)
    return out
