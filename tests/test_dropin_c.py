"""Builds tests/c/test_dropin.c against include/cpecan_dropin.h + libcpecan_hip.so and runs it: the reference-named C
entry points (stList / StateMachine / Hmm based) as a C caller would use them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "cpecan_amd")


def _build(tmp_path):
    exe = str(tmp_path / "test_dropin")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "test_dropin.c"), "-o", exe,
                           "-L", LIBDIR, "-lcpecan_hip", "-lm", "-Wl,-rpath," + LIBDIR])
    return exe


def _run(exe, mode, cwd):
    r = subprocess.run([exe, mode], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "0 failure(s)" in r.stdout, r.stdout


def test_dropin_host_side(tmp_path):
    """bands, split points, Hmm I/O, symbols, model priors, parameter defaults: no GPU needed."""
    if not os.path.exists(os.path.join(LIBDIR, "libcpecan_hip.so")):
        pytest.skip("libcpecan_hip.so not built")
    _run(_build(tmp_path), "cpu", str(tmp_path))


@pytest.mark.gpu
def test_dropin_known_answers_on_gpu(tmp_path):
    _run(_build(tmp_path), "gpu", str(tmp_path))
