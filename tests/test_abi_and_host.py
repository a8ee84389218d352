"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the
header declares, host-side dispatch logic, the harness file formats, batch sharding."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nttx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nttx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ntt_amd
    L = ntt_amd.lib()
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), n
    assert sorted(ntt_amd.exported_symbols()) == names


def test_header_is_plain_c_and_links(tmp_path):
    """include/nttx.h compiles as C99 (no C++ or torch types in the boundary) and a C program linked
    against libnttx.so resolves the device-free queries."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "nttx.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "  int lt[3]; uint32_t pr[4];\n"
        "  int np = nttx_plan_info(1 << 18, lt);\n"
        "  int nc = nttx_crt_primes_for(1337006139375617ull, 131072, pr);\n"
        "  int path = nttx_path_for(469762049ull, 131072);\n"
        '  printf("%d %d %d %d %d %u\\n", np, lt[0], lt[1], nc, path, pr[0]);\n'
        "  return nttx_poly_multiply_i32(0, 0, 0, 0, 7340033) == 0;   /* must fail cleanly: no device call made */\n"
        "}\n")
    exe = tmp_path / "abi"
    lib = os.path.join(root, "ntt_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", lib, "-lnttx", f"-Wl,-rpath,{lib}"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["2", "9", "9", "4", str(0), "469762049"], r.stdout


def test_no_cpu_fallback_without_gpu():
    """Without a device the product must fail loudly, never compute on the host."""
    import torch
    import ntt_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ntt_amd.NttxError) as e:
        ntt_amd.init(0)
    assert e.value.code == -3
    with pytest.raises(ntt_amd.NttxError) as e:
        ntt_amd.poly_multiply([1, 2], [3, 4], 2, 7340033)
    assert e.value.code == -4


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ntt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower(), f


def test_path_selection():
    import ntt_amd
    D, C = ntt_amd.PATH_DIRECT32, ntt_amd.PATH_CRT4
    assert ntt_amd.path_for(7340033, 4) == D
    assert ntt_amd.path_for(7340033, 131072) == D           # 2^18 | 7 * 2^20
    assert ntt_amd.path_for(7340033, 1 << 20) == C          # would need 2^21
    assert ntt_amd.path_for(104857601, 131072) == D
    assert ntt_amd.path_for(469762049, 131072) == D
    assert ntt_amd.path_for(1224736769, 131072) == D        # > 2^30: canonical-lane kernels
    D64 = ntt_amd.PATH_DIRECT64
    assert ntt_amd.path_for(1337006139375617, 131072) == D64  # 51-bit modulus (test.o:11) = 19*2^46+1
    assert ntt_amd.path_for(1337006139375617, 1 << 28) == D64 # 2^29 transform, three passes
    assert ntt_amd.path_for(1337006139375617, 1 << 29) == C   # beyond the longest 64-bit-lane plan
    assert ntt_amd.path_for((1 << 61) - 1, 8) == C            # prime beyond 32 bits without the roots
    assert ntt_amd.path_for(4611685941117976577, 1000) == D64 # just below 2^62
    assert ntt_amd.path_for(1000000007, 8) == C             # prime but 2^4 does not divide p-1
    assert ntt_amd.path_for(7340033 * 3, 4) == C            # composite


def test_crt_prime_selection():
    """As few CRT primes as keep n*(p-1)^2 below their product; the smallest ones where they suffice."""
    import ntt_amd
    primes = [469762049, 998244353, 1004535809, 1224736769]
    n = 1 << 17
    assert ntt_amd.crt_primes_for(1337006139375617, n) == primes           # id 4's modulus: all four
    assert ntt_amd.crt_primes_for(2147483647, n) == primes[:3]
    assert ntt_amd.crt_primes_for((1 << 36) - 5, n) == primes[1:]          # needs the three largest
    assert ntt_amd.crt_primes_for(65537, n) == primes[:2]
    assert ntt_amd.crt_primes_for(97, 4) == primes[:1]
    # the bound is exact: the largest p with n*(p-1)^2 < product is p = isqrt((product-1)//n) + 1
    from math import isqrt
    for ns in (1, 2, 3, 4):
        for subset in (primes[:ns], primes[4 - ns:]):
            prod = 1
            for m in subset:
                prod *= m
            top = isqrt((prod - 1) // n) + 1
            assert n * (top - 1) ** 2 < prod <= n * top ** 2
            assert len(ntt_amd.crt_primes_for(top, n)) <= ns
            if ns == 4:
                with pytest.raises(ntt_amd.NttxError) as e:
                    ntt_amd.crt_primes_for(top + 1, n)
                assert e.value.code == -2
            else:
                assert len(ntt_amd.crt_primes_for(top + 1, n)) >= (ns if subset == primes[:ns] else ns + 1)
    with pytest.raises(ntt_amd.NttxError):
        ntt_amd.crt_primes_for((1 << 61) - 1, n)


def test_plan_decomposition():
    import ntt_amd
    assert ntt_amd.plan_info(2) == [1]
    assert ntt_amd.plan_info(1 << 11) == [11]
    assert ntt_amd.plan_info(1 << 12) == [12]               # one workgroup per row up to 2^14 points
    assert ntt_amd.plan_info(1 << 14) == [14]
    assert ntt_amd.plan_info(1 << 15) == [6, 9] and ntt_amd.plan_info(1 << 16) == [7, 9]
    assert ntt_amd.plan_info(1 << 18) == [9, 9]
    assert ntt_amd.plan_info(1 << 20) == [9, 11]
    assert ntt_amd.plan_info(1 << 21) == [9, 12]
    assert ntt_amd.plan_info(1 << 23) == [9, 14]
    for k in range(24, 31):
        lt = ntt_amd.plan_info(1 << k)
        assert len(lt) == 3 and sum(lt) == k and all(6 <= x <= 10 for x in lt[:2]) and lt[2] <= 11
    with pytest.raises(ntt_amd.NttxError):
        ntt_amd.plan_info(12)
    # 64-bit lanes: moduli below 1.35e15 run their products on the floating-point class (no growth: rows up to 2^13);
    # above that the lazy integer class needs 2^(k+1) p <= 2^64 for a row of k stages; the Harvey class has no limit
    id4, top51, p62 = 1337006139375617, 2251756864012289, 4611685941117976577
    assert ntt_amd.plan_info_u64(1 << 12, id4) == [12] and ntt_amd.plan_info_u64(1 << 13, id4) == [13]
    assert ntt_amd.plan_info_u64(1 << 12, top51) == [12] and ntt_amd.plan_info_u64(1 << 13, top51) == [6, 7]
    assert ntt_amd.plan_info_u64(1 << 13, p62) == [13]
    assert ntt_amd.plan_info_u64(1 << 18, id4) == [9, 9]
    assert ntt_amd.plan_info_u64(1 << 22, id4) == [9, 13] and len(ntt_amd.plan_info_u64(1 << 23, id4)) == 3
    assert ntt_amd.plan_info_u64(1 << 21, top51) == [9, 12] and len(ntt_amd.plan_info_u64(1 << 22, top51)) == 3


def test_harness_file_formats(tmp_path, oracle):
    """fRead / fWrite / fCheck keep the reference's formats (main.cc:78-128)."""
    from ntt_amd import harness
    from oracle import pyoracle as po
    p, n = 104857601, 37
    a, b = oracle.fill_random(n, 5, p), oracle.fill_random(n, 6, p)
    d = tmp_path / "nttdata"
    d.mkdir()
    po.write_in_file(str(d / "7.in"), a, b, p)
    ra, rb, rn, rp = harness.fRead(7, str(d))
    assert rn == n and rp == p and np.array_equal(ra, a) and np.array_equal(rb, b)
    ab = oracle.poly_multiply_crt(a, b, p)
    harness.fWrite(ab, n, 7, str(d / "files"))
    text = open(d / "files" / "7.out").read()
    assert text == "".join(f"{int(x)}\n" for x in ab)
    os.rename(d / "files" / "7.out", d / "7.out")
    assert harness.fCheck(ab, n, 7, str(d), quiet=True)
    bad = ab.copy()
    bad[3] ^= 1
    assert not harness.fCheck(bad, n, 7, str(d), quiet=True)
    # the golden id0 pair round-trips through the same reader
    gold = os.path.join(ROOT, "tests", "golden")
    a0, b0, n0, p0 = harness.fRead(0, gold)
    assert (list(a0), list(b0), n0, p0) == ([1, 5, 5, 4], [4, 1, 5, 2], 4, 7340033)
    assert harness.fCheck(oracle.poly_multiply_crt(a0, b0, p0), n0, 0, gold, quiet=True)


def test_shard_range():
    from ntt_amd.sharding import prime_owner, shard_range
    for total in (0, 1, 7, 256, 4096):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(total, r, world)
                cover += list(range(lo, hi))
                assert 0 <= hi - lo <= total // world + 1
            assert cover == list(range(total))
    assert [prime_owner(k, 4) for k in range(4)] == [0, 1, 2, 3]
    assert [prime_owner(k, 2) for k in range(4)] == [0, 1, 0, 1]


def test_batch_dealing_matches_the_rank_sharding():
    """nttx_deal_batch (how nttx_poly_multiply_batch_u32 deals products to the devices of nttx_init_devices) gives the
    shares ntt_amd.sharding.shard_range gives ranks: contiguous, covering, sizes differing by at most one."""
    import ntt_amd
    from ntt_amd.sharding import shard_range
    for batch in (0, 1, 2, 7, 8, 9, 100, 4096, 70001):
        for ndev in (1, 2, 3, 4, 8):
            shares = [ntt_amd.deal_batch(batch, ndev, k) for k in range(ndev)]
            assert shares == [tuple(shard_range(batch, k, ndev)) for k in range(ndev)]
            assert shares[0][0] == 0 and shares[-1][1] == batch
            assert all(shares[k][1] == shares[k + 1][0] for k in range(ndev - 1))
            sizes = [hi - lo for lo, hi in shares]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ntt_amd.NttxError):
        ntt_amd.deal_batch(10, 2, 2)
    assert ntt_amd.device_count() == 0 or ntt_amd.device_count() >= 1      # callable without a device
