"""cpecan_amd -- MI355X-native banded pair-HMM forward/backward/posterior path behind cPecan's API.

Only what the hot path needs: ``csrc/`` (HIP kernels + C host code + the C ABI in include/cpecan_hip.h),
``api`` (ctypes binding mirroring the reference's operator names), ``dist`` (pair sharding + the EM count
all-reduce) and ``workload`` (seeded synthetic inputs).
"""
from . import api, dist, workload  # noqa: F401

__all__ = ["api", "dist", "workload"]
