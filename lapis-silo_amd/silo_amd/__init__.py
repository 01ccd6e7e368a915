"""silo_amd — MI355X-native engine for SILO's filter -> Aggregated / Mutations hot path.

Python here is plumbing (ctypes binding of the C ABI, synthetic-data parameters, bench drivers); the
product is lib/libsilo_gpu.so (HIP kernels) and lib/libsilo_engine.so (C++ QueryEngine mirror).
"""
from . import alphabet  # noqa: F401
