"""the bit-identity claim under test.  `make -C cpecan_amd/csrc exact` (part of build()) makes
cpecan_amd/libcpecan_hip_exact.so, the same kernels with the reference's logAdd operation for operation
(impl/pairwiseAligner.c:287-307); tests/exact_check.py, run here as ONE child process with that library, asserts
equality -- tolerance zero -- of the sweeps' debug buffers, the emitted triples and the forward probabilities with the
oracle.  (The shipped library's logAdd is fused: within 1e-9 of these values, tests/parity.py.)"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXACT = os.path.join(ROOT, "cpecan_amd", "libcpecan_hip_exact.so")


@pytest.mark.gpu
def test_exact_build_is_bit_identical_to_the_oracle():
    assert os.path.exists(EXACT), "%s is missing: __graft_entry__.build() makes it (`make -C cpecan_amd/csrc exact`)" % EXACT
    env = dict(os.environ, CPECAN_LIB=EXACT)
    r = subprocess.run([sys.executable, os.path.join(HERE, "exact_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "forward probabilities equal" in r.stdout


def test_exact_library_exports_the_same_abi():
    """CPU: the diagnostic library is the same ABI (loads, exports every declared symbol)."""
    import ctypes
    from cpecan_amd import api
    if not os.path.exists(EXACT):
        pytest.skip("exact library not built")
    L = ctypes.CDLL(EXACT)
    for name in api.EXPORTS:
        assert hasattr(L, name), name
