"""zkdl_amd -- MI355X-native zkDL proving hot path (HIP kernels behind a C ABI).

    from zkdl_amd.capi import Zk, FC, ReLU, Transcript

The package holds only what the path needs: csrc/ (HIP kernels + C ABI), host/ (C++ mirror of the
reference's FrTensor / G1Tensor / Commitment / zkFC / zkReLU classes and the demo driver) and the
ctypes binding used by the tests and bench.py.
"""
__all__ = ["capi"]
