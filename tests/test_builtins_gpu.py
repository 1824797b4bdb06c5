"""The OpenCL built-ins the reference's kernels rely on (dot, fma, round, convert_*_sat*: v210.ts:68-77,
148-155,176-183) - AMD's own device-library bitcode executed NATIVELY on the MI355X against the product's
hand-spelled primitives (phaneron_amd/csrc/ph_device.h, ph_ldslut.h), and against the x86-64 retarget of
the same bitcode that oracle/_ref links (oracle/refbuild/devlib_builtins.py).

  tests/native/builtins_check.hip : all 2^32 floats for every convert, 2^28 operand sets for dot3 / dot4 / fma
"""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "native", "builtins_check")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    if not os.path.exists(EXE):
        pytest.fail("tests/native/builtins_check is not built (run __graft_entry__.build())")
    dump = str(tmp_path_factory.mktemp("builtins") / "devlib_gfx950.bin")
    p = subprocess.run([EXE, "28", dump], capture_output=True, text=True, timeout=600)
    assert p.returncode in (0, 1), p.stderr
    return json.loads(p.stdout.strip().splitlines()[-1]), dump


def test_product_primitives_equal_the_device_library_on_gfx950(run):
    report, _ = run
    assert report["device"].startswith("gfx950")
    checks = [c for c in report["checks"] if not c["control"]]
    controls = [c for c in report["checks"] if c["control"]]
    assert len(checks) == 9 and len(controls) == 2
    for c in checks:
        assert c["tested"] >= (1 << 28), c
        assert c["bad"] == 0, "%s: %d of %d disagree, e.g. operand %s" % (c["check"], c["bad"], c["tested"], c["example"])
    converts = [c for c in checks if "convert" in c["check"]]
    assert len(converts) == 6 and all(c["tested"] == 1 << 32 for c in converts)
    # the checker can see a wrong spelling: a truncating convert and an unfused dot both disagree somewhere
    for c in controls:
        assert c["bad"] > 0, c


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref (x86 retarget of the device library) not present")
def test_x86_retarget_of_the_device_library_equals_its_gfx950_execution(run):
    """Closes the loop for the goldens: the bitcode bodies that oracle/_ref runs on the host give, on the
    same operands, the bits the GPU computed with them."""
    _, dump = run
    raw = np.fromfile(dump, np.uint8)
    n = int(raw[:4].view(np.uint32)[0])
    off = 4
    d4 = raw[off:off + n * 36].view(np.float32).reshape(n, 9)
    off += n * 36
    d3 = raw[off:off + n * 4].view(np.float32)
    off += n * 4
    bits = raw[off:off + n * 4].view(np.uint32)
    off += n * 4
    conv = raw[off:off + n * 10].view(np.uint16).reshape(5, n)
    r = orc.ref()
    a, b = np.ascontiguousarray(d4[:, 0:4]), np.ascontiguousarray(d4[:, 4:8])

    def same(x, y):
        nan = np.isnan(x)
        return np.array_equal(nan, np.isnan(y)) and np.array_equal(x.view(np.uint32)[~nan], y.view(np.uint32)[~nan])

    assert same(orc.ref_builtin_dot(r, a, b), np.ascontiguousarray(d4[:, 8]))
    assert same(orc.ref_builtin_dot(r, np.ascontiguousarray(a[:, :3]), np.ascontiguousarray(b[:, :3])), d3)
    for which in range(5):
        assert np.array_equal(orc.ref_builtin_convert_list(r, which, bits), conv[which]), which
