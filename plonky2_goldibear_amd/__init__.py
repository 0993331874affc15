"""plonky2_goldibear_amd - MI355X (gfx950) implementation of the plonky2_goldibear commitment hot path.

csrc/      hand-written HIP kernels + the C ABI declared in include/goldibear_gpu.h
native     ctypes binding of that ABI (no fallback: raises if the library is missing)
polynomial_batch  host-side mirror of PolynomialBatch / MerkleTree (fri/oracle.rs, hash/merkle_tree.rs)
"""
from .native import GB_BABYBEAR, GB_GOLDILOCKS, GoldibearError, ShapeError  # noqa: F401
from .native import PermArgZeroError, TooManyPermArgFailuresError, VerifyError  # noqa: F401
from .polynomial_batch import GpuContext, MerkleTree, PolynomialBatch  # noqa: F401
from .prover import CircuitData, VerifierCircuitData  # noqa: F401
