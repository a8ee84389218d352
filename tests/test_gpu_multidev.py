"""The device-list form of the C ABI (nttx_init_devices, nttx_set_device, the batch entry point dealing its products over
the listed devices) in fresh child processes: one device, bit-equal to the oracle like the one-device form, and the
several-device code path rehearsed with several contexts on the one GPU of the test box."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("nctx", [1, 2, 3])
def test_device_list(nctx):
    env = dict(os.environ)
    if nctx > 1:
        env["NTTX_REHEARSE_DEVICES"] = "1"
    r = subprocess.run([sys.executable, os.path.join(HERE, "multidev_child.py"), str(nctx)], env=env, capture_output=True,
                       text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"multidev ok {nctx}" in r.stdout
