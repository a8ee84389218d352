"""ntt_amd — MI355X (gfx950) NTT polynomial multiplication behind the reference
harness's ``poly_multiply(a, b, ab, n, p)`` entry point.

The compute path is the HIP library ``ntt_amd/csrc/libnttx.so`` (C ABI in
``include/nttx.h``); this package is a thin ctypes binding plus a mirror of the harness
flow (``fRead`` / ``poly_multiply`` / ``fCheck`` / ``fWrite``).  There is no CPU
fallback: importing works without a GPU, but every compute call needs one and raises
if the library or device is missing.
"""
from .api import (  # noqa: F401
    NttxError,
    CRT_PRIMES,
    PATH_CRT4,
    PATH_DIRECT32,
    PATH_DIRECT64,
    crt_combine,
    cyclic_multiply,
    crt_prime,
    crt_primes_for,
    dev_colpass_u32,
    dev_stream_copy,
    dev_crt_multiply_u64,
    dev_direct_multiply_u64,
    dev_ntt_u64,
    poly_multiply_direct64,
    reserve_u64,
    dev_crt_recombine_u64,
    dev_crt_residues_u64,
    dev_crt_residues_sliced_u64,
    dev_cyclic_multiply_u32,
    dev_cyclic_multiply_pieces_u32,
    dev_poly_multiply_u32,
    host_alloc,
    host_free,
    dev_poly_multiply_u32_static,
    dev_negacyclic_multiply_u32,
    dev_negacyclic_multiply_u64,
    dev_ntt_u32,
    exported_symbols,
    init,
    init_devices,
    device_count,
    set_device,
    deal_batch,
    lib,
    ntt,
    path_for,
    plan_info,
    plan_info_u64,
    poly_multiply,
    poly_multiply_batch,
    poly_multiply_crt,
    profile_enable,
    profile_read,
    profile_reset,
    KERNEL_KINDS,
    reserve,
    shutdown,
)
from . import harness  # noqa: F401

__all__ = [
    "NttxError", "CRT_PRIMES", "PATH_CRT4", "PATH_DIRECT32", "PATH_DIRECT64", "dev_direct_multiply_u64", "dev_ntt_u64", "poly_multiply_direct64", "reserve_u64", "crt_combine", "cyclic_multiply",
    "crt_prime", "crt_primes_for", "dev_colpass_u32", "dev_stream_copy", "dev_negacyclic_multiply_u32", "dev_negacyclic_multiply_u64", "dev_crt_recombine_u64", "dev_crt_residues_u64", "dev_crt_residues_sliced_u64",
    "dev_crt_multiply_u64", "dev_cyclic_multiply_u32", "dev_cyclic_multiply_pieces_u32", "dev_poly_multiply_u32", "dev_poly_multiply_u32_static", "host_alloc", "host_free", "dev_ntt_u32", "exported_symbols", "init", "init_devices", "device_count", "set_device", "deal_batch", "lib", "ntt",
    "path_for", "plan_info", "plan_info_u64", "poly_multiply", "poly_multiply_batch", "poly_multiply_crt", "profile_enable", "profile_read", "profile_reset", "KERNEL_KINDS",
    "reserve", "shutdown",
    "harness",
]
