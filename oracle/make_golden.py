#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference programs compiled in place
(oracle/_ref, see oracle/Makefile).  Run in the build container (needs /root/reference):

    python oracle/make_golden.py

Fixtures are data only: seeds, sizes and expected outputs.  Inputs are regenerated from the
seed with the splitmix64 stream of oracle_fill_random (coefficient = next() mod p).

  0.in / 0.out            the reference's own files/0.out and the input recovered from it
  small_products.npz      full outputs of main.cc's driver (ref_main) and of monttry2.cpp for
                          tiny and ragged n on every modulus
  transforms.npz          ntt_iter_barrett (main.cc:143) forward/inverse outputs
  harness.json            sha256 + first/last 8 coefficients of the n = 2^15..2^17 products on
                          the four harness moduli (main.cc driver), and the sha256 of the
                          reference's files/1..3.out whose inputs are not available offline
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SMALL_N = (1, 2, 3, 4, 5, 8, 31, 32, 33, 64, 1000, 1024, 2048)
PRIMES32 = (7340033, 104857601, 469762049, 998244353, 1004535809, 1224736769)
BIG = (1337006139375617, 263882790666241)   # test.o:11 and the comment at monttry2.cpp:357
HARNESS_P = {1: 7340033, 2: 104857601, 3: 469762049, 4: 1337006139375617}


def seeds(n, p):
    return (n * 1000003 + p) % (1 << 62), (n * 999983 + 7 * p + 1) % (1 << 62)


def main():
    po.build()
    assert po.have_ref(), "oracle/_ref is missing: /root/reference not available?"
    os.makedirs(GOLD, exist_ok=True)
    O = po.Oracle()
    M = po.RefMont("monttry2")

    # (1) the reference's own golden
    shutil.copy("/root/reference/files/0.out", os.path.join(GOLD, "0.out"))
    po.write_in_file(os.path.join(GOLD, "0.in"), [1, 5, 5, 4], [4, 1, 5, 2], 7340033)
    outs, _, _ = po.run_ref_main([([1, 5, 5, 4], [4, 1, 5, 2], 7340033)])
    assert np.array_equal(outs[0], np.loadtxt(os.path.join(GOLD, "0.out"), dtype=np.uint64))

    # (2) small products through main.cc's whole driver, five cases per process
    small = {}
    todo = [(n, p) for p in PRIMES32 + BIG for n in SMALL_N]
    for i in range(0, len(todo), 5):
        chunk = todo[i:i + 5]
        cases = []
        for n, p in chunk:
            sa, sb = seeds(n, p)
            cases.append((O.fill_random(n, sa, p), O.fill_random(n, sb, p), p))
        outs, _, _ = po.run_ref_main(cases)
        for k, (n, p) in enumerate(chunk):
            small[f"main_{p}_{n}"] = outs[k]
            if p < (1 << 30):   # monttry2.cpp uses R = 2^30 and int I/O
                r = M.poly_multiply(cases[k][0].astype(np.int32), cases[k][1].astype(np.int32), p)
                assert np.array_equal(r.astype(np.uint64), outs[k]), (n, p)
                small[f"mont_{p}_{n}"] = r.astype(np.uint32)
    np.savez_compressed(os.path.join(GOLD, "small_products.npz"), **small)

    # (3) transform level
    R = po.RefCRT("main")
    tr = {}
    for p in PRIMES32:
        for logn in (3, 6, 10, 13):
            n = 1 << logn
            x = O.fill_random(n, 424242 + logn + p % 1000, p).astype(np.uint32)
            tr[f"fwd_{p}_{logn}"] = R.ntt_iter_barrett(x, p, False)
            tr[f"inv_{p}_{logn}"] = R.ntt_iter_barrett(x, p, True)
    np.savez_compressed(os.path.join(GOLD, "transforms.npz"), **tr)

    # (4) harness sizes: hashes only
    h = {"seed_rule": "a: seed 1000+id, b: seed 2000+id (+ 10*log2 n for n < 2^17)", "cases": {}, "reference_files": {}}
    for logn in (15, 16, 17):
        n = 1 << logn
        cases = [([1, 5, 5, 4], [4, 1, 5, 2], 7340033)]
        for tid in (1, 2, 3, 4):
            off = 0 if logn == 17 else 10 * logn
            p = HARNESS_P[tid]
            cases.append((O.fill_random(n, 1000 + tid + off, p), O.fill_random(n, 2000 + tid + off, p), p))
        outs, _, _ = po.run_ref_main(cases)
        for tid in (1, 2, 3, 4):
            v = outs[tid].astype("<u8")
            h["cases"][f"{tid}_{logn}"] = {"p": HARNESS_P[tid], "n": n, "sha256": hashlib.sha256(v.tobytes()).hexdigest(),
                                          "head": [int(x) for x in v[:8]], "tail": [int(x) for x in v[-8:]]}
    for k in (1, 2, 3):
        h["reference_files"][f"files/{k}.out"] = hashlib.sha256(open(f"/root/reference/files/{k}.out", "rb").read()).hexdigest()
    json.dump(h, open(os.path.join(GOLD, "harness.json"), "w"), indent=1)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
