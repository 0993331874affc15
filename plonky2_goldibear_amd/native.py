"""ctypes binding of libgoldibear_gpu.so - exactly the symbols include/goldibear_gpu.h declares.

There is no fallback: if the HIP library is missing or fails to load this raises, so a GPU test
can never pass on a silent CPU path.
"""
import ctypes as C
import os

from . import build as _build

# Several contexts proving at once (one stream and one host thread each - recursion-sized proofs cannot fill the GPU alone) need
# a hardware queue each to overlap: the HIP runtime maps streams onto GPU_MAX_HW_QUEUES queues, 4 unless the environment says
# otherwise, and reads the variable when it initialises - so it is set here, before anything touches the GPU, unless the host has
# chosen a value (tools/bench_recursion_shape.py --inflight 6: 576 proofs/s on 4 queues, 831 on 8).  A C host sets it itself
# (INTEGRATION.md).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

GB_OK, GB_ERR_INVALID, GB_ERR_HIP, GB_ERR_OOM, GB_ERR_UNSUPPORTED = 0, 1, 2, 3, 4
GB_ERR_PERM_ARG_ZERO, GB_ERR_OPENING_IN_SUBGROUP, GB_ERR_BUFFER_TOO_SMALL, GB_ERR_VERIFY = 16, 17, 18, 19
GB_GOLDILOCKS, GB_BABYBEAR = 0, 1
GB_INPUT_HOST, GB_INPUT_DEVICE = 0, 1
GB_INPUT_P3_REPR = 2   # host elements are the reference's field types as they lie in memory (p3 words), not canonical values
GB_SALT_SIZE = 4

_vp, _u32, _u64, _sz, _i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t, C.c_int32
_pvp = C.POINTER(C.c_void_p)
_cols = C.POINTER(C.c_void_p)   # const void* const*: one pointer per separately allocated column

# name -> (restype, argtypes); kept in step with include/goldibear_gpu.h (tests/test_abi.py checks it)
SIGNATURES = {
    "gb_ctx_create": (_i32, [C.c_int, _pvp]),
    "gb_ctx_destroy": (_i32, [_vp]),
    "gb_last_error": (C.c_char_p, [_vp]),
    "gb_ctx_synchronize": (_i32, [_vp]),
    "gb_ctx_trim": (_i32, [_vp]),
    "gb_ctx_stream": (_i32, [_vp, _pvp]),
    "gb_ctx_set_profiling": (_i32, [_vp, _i32]),
    "gb_ctx_scope_ms": (_i32, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    "gb_ctx_scope_reset": (_i32, [_vp]),
    "gb_ctx_set_option": (_i32, [_vp, C.c_char_p, C.c_int64]),
    "gb_host_alloc": (_i32, [_vp, _sz, _pvp]),
    "gb_host_free": (_i32, [_vp, _vp]),
    "gb_host_register": (_i32, [_vp, _vp, _sz]),
    "gb_host_unregister": (_i32, [_vp, _vp]),
    "gb_commit_values_cols": (_i32, [_vp, _u32, _cols, _sz, _u32, _u32, _u32, _vp, _u32, _pvp]),
    "gb_commit_coeffs_cols": (_i32, [_vp, _u32, _cols, _sz, _u32, _u32, _u32, _vp, _u32, _pvp]),
    "gb_commit_values": (_i32, [_vp, _u32, _vp, _sz, _u32, _u32, _u32, _vp, _u32, _pvp]),
    "gb_commit_coeffs": (_i32, [_vp, _u32, _vp, _sz, _u32, _u32, _u32, _vp, _u32, _pvp]),
    "gb_batch_free": (_i32, [_vp]),
    "gb_batch_info": (_i32, [_vp, C.POINTER(_u32), C.POINTER(_sz), C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "gb_batch_cap": (_i32, [_vp, _vp]),
    "gb_batch_coeffs": (_i32, [_vp, _sz, _vp]),
    "gb_batch_lde_values": (_i32, [_vp, _u64, _u64, _vp]),
    "gb_batch_leaf": (_i32, [_vp, _u64, _vp, _vp, C.POINTER(_u32)]),
    "gb_batch_digests": (_i32, [_vp, _vp]),
    "gb_batch_leaves": (_i32, [_vp, _vp]),
    "gb_batch_device_ptrs": (_i32, [_vp, _pvp, _pvp, _pvp]),
    "gb_batch_eval_ext": (_i32, [_vp, _vp, _vp]),
    "gb_verify": (_i32, [_vp, _vp, _sz]),
    "gb_proof_compress": (_i32, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "gb_proof_decompress": (_i32, [_vp, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "gb_verify_compressed": (_i32, [_vp, _vp, _sz]),
    "gb_verifier_create": (_i32, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _pvp]),
    "gb_pow_grind": (_i32, [_vp, C.c_uint32, _vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    "gb_permute": (_i32, [_vp, _u32, _vp, _vp, _u64]),
    "gb_circuit_create": (_i32, [_vp, _vp, _vp, _vp, _u32, _pvp]),
    "gb_circuit_create_gates": (_i32, [_vp, _vp, _vp, _u32, _vp, _vp, _u32, _pvp]),
    "gb_circuit_create_cols": (_i32, [_vp, _vp, _cols, _vp, _u32, _pvp]),
    "gb_circuit_create_gates_cols": (_i32, [_vp, _vp, _vp, _u32, _cols, _vp, _u32, _pvp]),
    "gb_circuit_free": (_i32, [_vp]),
    "gb_circuit_verifier_data": (_i32, [_vp, _vp, _vp]),
    "gb_circuit_constants_sigmas_commitment": (_i32, [_vp, _pvp]),
    "gb_circuit_set_fri_reduction_arity_bits": (_i32, [_vp, C.POINTER(_u32), _u32]),
    "gb_circuit_fri_reduction_arity_bits": (_i32, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "gb_zs_partial_products": (_i32, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "gb_quotient_polys": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    "gb_prove_openings": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "gb_prove": (_i32, [_vp, _vp, _u32, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "gb_prove_retry": (_i32, [_vp, _vp, _u32, _u32, _u64, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "gb_circuit_drop_retry": (_i32, [_vp]),
    "gb_prove_salted": (_i32, [_vp, _vp, _u32, _vp, _sz, _vp, _vp, _sz, C.POINTER(_sz)]),
    "gb_prove_cols": (_i32, [_vp, _cols, _u32, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "gb_prove_retry_cols": (_i32, [_vp, _cols, _u32, _u32, _u64, _vp, _sz, _vp, _sz, C.POINTER(_sz)]),
    "gb_prove_salted_cols": (_i32, [_vp, _cols, _u32, _vp, _sz, _vp, _vp, _sz, C.POINTER(_sz)]),
    "gb_zs_partial_products_cols": (_i32, [_vp, _cols, _u32, _vp, _vp, _vp]),
}
# include/goldibear_gpu_test_hooks.h: exports for the test suite, outside the product ABI
TEST_HOOK_SIGNATURES = {
    "gb_test_arm_perm_arg_failure": (_i32, [_vp]),
}

_lib = None


def library_path():
    return _build.LIB


def load():
    """Load the in-tree HIP library (raises if it has not been built - no CPU fallback)."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the gfx950 HIP library is the only implementation; there is no CPU fallback)" % path)
        # PyTorch bundles its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one
        # process cannot both open the GPU.  Importing torch first makes the dynamic loader bind
        # this library's NEEDED libamdhip64.so.7 to the copy torch already loaded.
        import torch  # noqa: F401
        lib = C.CDLL(path)
        for name, (res, args) in list(SIGNATURES.items()) + list(TEST_HOOK_SIGNATURES.items()):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class GoldibearError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("gb_status=%d: %s" % (status, message))
        self.status = status


class PermArgZeroError(GoldibearError):
    """GB_ERR_PERM_ARG_ZERO: ProverError::InvZeroPermArg (plonk/prover.rs:512-514) - re-randomise and retry."""


class TooManyPermArgFailuresError(GoldibearError):
    """ProverError::TooManyPermArgFailures (plonk/prover.rs:188-192, 226): MAX_PERM_ARG_RETRIES attempts failed, or
    the circuit has no random wire to re-randomise."""


class VerifyError(GoldibearError):
    """GB_ERR_VERIFY: the proof does not verify (the message names the failed check)."""


class ShapeError(GoldibearError, ValueError):
    """GB_ERR_INVALID: where the reference would assert!/panic! on a shape violation."""


def check(status, ctx_handle=None):
    if status == GB_OK:
        return
    msg = load().gb_last_error(ctx_handle)
    msg = msg.decode() if msg else ""
    cls = {GB_ERR_INVALID: ShapeError, GB_ERR_PERM_ARG_ZERO: PermArgZeroError, GB_ERR_VERIFY: VerifyError}.get(status, GoldibearError)
    raise cls(status, msg)
