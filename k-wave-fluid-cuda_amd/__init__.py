"""MI355X-native k-space pseudospectral acoustic solver: per-step hot path of klepo/k-Wave-Fluid-CUDA.

Layout: csrc/ (HIP kernels + C-ABI, include/kwave_hip.h), host/ (C++ mirror of the reference's
Parameters / MatrixContainer / KSpaceFirstOrderSolver), capi.py (ctypes binding of the C-ABI),
synthetic.py (input generator).  Import as `kwave_amd` through the repo-root shim kwave_amd.py.
"""
__all__ = ["synthetic"]
