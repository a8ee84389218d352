"""The oracle against the real reference programs compiled in place (oracle/_ref).  The
prebuilt files travel with the tree; where they are absent these tests are skipped and
parity rests on the committed fixtures (test_oracle_golden.py)."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("which", ("main", "main_pthread"))
def test_transform_level(oracle, which):
    R = po.RefCRT(which)
    for p in po.CRT_PRIMES + (7340033, 104857601):
        for n in (2, 8, 256, 4096):
            x = oracle.fill_random(n, n + p % 97, p).astype(np.uint32)
            for inv in (False, True):
                assert np.array_equal(oracle.ntt_iter_barrett(x, p, inv), R.ntt_iter_barrett(x, p, inv))


def test_ntt_compute(oracle):
    R = po.RefCRT("main")
    for p in po.CRT_PRIMES:
        for n in (2, 64, 2048):
            a, b = oracle.fill_random(n, 1, p).astype(np.uint32), oracle.fill_random(n, 2, p).astype(np.uint32)
            assert np.array_equal(oracle.ntt_compute(a, b, p), R.ntt_compute(a, b, p))


@pytest.mark.parametrize("which", ("monttry2", "pthread_normal"))
def test_montgomery_programs(oracle, which):
    M = po.RefMont(which)
    for p in (7340033, 104857601, 469762049):
        for n in (1, 3, 17, 500, 8192):
            a, b = oracle.fill_random(n, 3, p).astype(np.int32), oracle.fill_random(n, 4, p).astype(np.int32)
            rc, r = oracle.poly_multiply_mont(a, b, p)
            assert rc == 0 and np.array_equal(r, M.poly_multiply(a, b, p))
    a, b = np.arange(1, 40, dtype=np.int32), np.arange(5, 44, dtype=np.int32)
    assert np.array_equal(oracle.poly_multiply_naive(a, b, 7340033), M.poly_multiply_naive(a, b, 7340033))
    assert M.lib.ref_mont_probe(3 << 20, 7340033) == 1 and M.lib.ref_mont_probe(1 << 30, 7340032) == 2


@pytest.mark.parametrize("which", ("ref_main", "ref_main_pthread"))
def test_whole_driver(oracle, which):
    """The reference's main loop, CRT included, on five seeded cases (one of them the 51-bit modulus)."""
    cases = [([1, 5, 5, 4], [4, 1, 5, 2], 7340033)]
    for k, (n, p) in enumerate(((1000, 7340033), (4097, 104857601), (30000, 469762049), (65536, 1337006139375617))):
        cases.append((oracle.fill_random(n, 10 + k, p), oracle.fill_random(n, 20 + k, p), p))
    outs, lat, _ = po.run_ref_main(cases, which=which)
    assert len(lat) == 5
    for i, (a, b, p) in enumerate(cases):
        assert np.array_equal(oracle.poly_multiply_crt(a, b, p), outs[i]), i
