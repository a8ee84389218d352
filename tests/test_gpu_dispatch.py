"""Every kernel the dispatch can reach is compared with the oracle: the default dispatch in this process, and each
measurement switch (NTTX_* environment variables, read once per process) in a child process of its own that runs
the sweep of the kernel family the switch changes.  tools/kernel_coverage.sh traces the same runs and lists the
kernels of libnttx.so that no oracle comparison launched."""
import os
import subprocess
import sys

import pytest

from tests import sweeps

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("suite", ["col", "col_big", "rows", "crt", "d64", "colpass", "layouts", "three_pass_long", "d64_classes"])
def test_default_dispatch(nttx, oracle, suite):
    sweeps.SUITES[suite](nttx, oracle)


# (environment, suites): one child process each
SWITCHES = [
    ({"NTTX_COL_SEG256": "0"}, ["col_lite"]),                       # 2^6-point passes on 128-byte tiles at every batch
    ({"NTTX_COL_NARROW": "0"}, ["col_lite"]),                       # full-line tiles for small launches too
    ({"NTTX_COL_NARROW": "2"}, ["col_lite"]),                       # 4-column tiles forced
    ({"NTTX_COL_NARROW": "3"}, ["col_lite"]),                       # 8-column tiles forced
    ({"NTTX_GRID_COL": "5", "NTTX_GRID_COLI": "3"}, ["col_lite"]),  # few workgroups: each strides over many tiles
    ({"NTTX_UNSIGNED": "1"}, ["rows", "col_lite"]),                 # unsigned lanes in the fused row kernels
    ({"NTTX_WIDE": "0"}, ["rows"]),                            # four-multiple signed rows for moduli below 2^27 too
    ({"NTTX_LONGROW": "0"}, ["rows"]),                         # plans without one-workgroup rows
    ({"NTTX_LONGROW": "1"}, ["rows"]),
    ({"NTTX_LONGROW": "2"}, ["rows"]),
    ({"NTTX_GRID_ROW": "3", "NTTX_GRID_ROWG": "2", "NTTX_GRID_LROW": "2"}, ["rows"]),
    ({"NTTX_TWIST_UNFUSED": "1"}, ["rows", "d64", "d64_classes"]),            # negacyclic products through separate twist passes
    ({"NTTX_CRT_STREAMS": "0"}, ["crt"]),                      # CRT primes on one stream
    ({"NTTX_CRT_STREAMS": "1"}, ["crt"]),                      # ... side by side at every batch
    ({"NTTX_CRT_FOUR": "1"}, ["crt"]),                         # always four primes
    ({"NTTX_CRT_FUSED_SPLIT": "0"}, ["crt"]),                  # residue split as a pass of its own
    ({"NTTX_FP64": "0"}, ["d64", "d64_classes"]),                             # integer classes on the 64-bit lanes
    ({"NTTX_LONGROW64": "0"}, ["d64", "d64_classes"]),                        # 64-bit plans without rows beyond 2^11 points
    ({"NTTX_GRID_ROW64": "3", "NTTX_GRID_COL64": "3"}, ["d64"]),
]


@pytest.mark.parametrize("env,suites", SWITCHES, ids=[",".join(f"{k}={v}" for k, v in e.items()) for e, _ in SWITCHES])
def test_switch_in_child_process(env, suites):
    r = subprocess.run([sys.executable, os.path.join(HERE, "sweeps.py")] + suites, env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for s in suites:
        assert f"sweep {s} ok" in r.stdout
