"""Mirror of the reference harness flow (main.cc:272-346): for each test id, fRead ->
poly_multiply -> fCheck -> print latency -> fWrite, with the same file formats and stdout
lines.  The data directory is configurable (the reference hard-codes /nttdata)."""
from __future__ import annotations

import os
import time

import numpy as np

from . import api

OK_LINE = "多项式乘法结果正确"      # main.cc:126
BAD_LINE = "多项式乘法结果错误"     # main.cc:122


def data_dir():
    return os.environ.get("NTT_DATA_DIR", "/nttdata")


def fRead(input_id: int, directory: str | None = None):
    """n p a[0..n) b[0..n), whitespace separated (main.cc:78-95).  Returns (a, b, n, p)."""
    path = os.path.join(directory or data_dir(), f"{input_id}.in")
    with open(path) as f:
        tok = f.read().split()
    n, p = int(tok[0]), int(tok[1])
    vals = np.array(tok[2:2 + 2 * n], dtype=np.uint64)
    return vals[:n].copy(), vals[n:2 * n].copy(), n, p


def fWrite(ab, n: int, input_id: int, directory: str = "files"):
    """2n-1 lines, one coefficient each (main.cc:97-109)."""
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, f"{input_id}.out"), "w") as f:
        f.write("".join(f"{int(x)}\n" for x in ab[: 2 * n - 1]))


def fCheck(ab, n: int, input_id: int, directory: str | None = None, quiet: bool = False) -> bool:
    """Compare with <id>.out (main.cc:111-128).  Unlike the reference this also returns the verdict."""
    path = os.path.join(directory or data_dir(), f"{input_id}.out")
    with open(path) as f:
        want = np.array(f.read().split()[: 2 * n - 1], dtype=np.uint64)
    ok = len(want) == 2 * n - 1 and bool(np.array_equal(want, np.asarray(ab[: 2 * n - 1], dtype=np.uint64)))
    if not quiet:
        print(OK_LINE if ok else BAD_LINE)
    return ok


def run(test_begin: int = 0, test_end: int = 4, directory: str | None = None, out_dir: str = "files") -> bool:
    """The driver loop of main.cc:272-346.  Returns True when every id checked out."""
    all_ok = True
    for i in range(test_begin, test_end + 1):
        a, b, n, p = fRead(i, directory)
        t0 = time.perf_counter()
        ab = api.poly_multiply(a, b, n, p, dtype=np.uint64)
        ms = (time.perf_counter() - t0) * 1e3
        all_ok &= fCheck(ab, n, i, directory)
        print(f"average latency for n = {n} p = {p} : {ms:g} (ms)")
        fWrite(ab, n, i, out_dir)
    return all_ok
