"""mpc_bulletproof_amd -- MI355X-native Bulletproofs hot path (Stark curve) behind a C ABI.

This package is a thin ctypes view of ``libbpgpu.so`` (include/bpgpu.h) for tests and benchmarks.
There is no CPU implementation here: importing it raises if the HIP library has not been built
(``python -c 'import __graft_entry__ as g; g.build()'``), and every call fails with
BPGPU_E_DEVICE when no GPU is present.
"""
from . import lib  # noqa: F401
from .lib import BpGpu, BpGpuError  # noqa: F401
