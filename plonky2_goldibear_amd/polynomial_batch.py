"""Host-side mirror of the reference's commitment interface, over the C ABI.

Same names and argument meaning as plonky2/src/fri/oracle.rs (PolynomialBatch::from_values /
from_coeffs / get_lde_values, fields polynomials / merkle_tree / degree_log / rate_bits /
blinding) and plonky2/src/hash/merkle_tree.rs (MerkleTree::get / prove, fields cap / digests /
leaves).  Shape violations raise ShapeError (a ValueError) where the reference panics.
Everything large stays on the GPU; accessors copy back only what is asked for.
"""
import atexit
import ctypes as C
import sys
import weakref

import numpy as np

from . import native as N

GL_P = 0xFFFFFFFF00000001

# Handles must be released while the HIP runtime is still alive: at interpreter exit free every
# live batch, then every context, before C++ static destructors run.
_live_batches = weakref.WeakSet()
_live_contexts = weakref.WeakSet()


@atexit.register
def _shutdown():
    from . import prover as _prover
    for c in list(_prover._live_circuits):
        c.free()
    for b in list(_live_batches):
        b.free()
    for c in list(_live_contexts):
        c.close()


class GpuContext:
    """One per HIP device (gb_ctx).  Independent contexts are how proofs shard one-per-GPU."""

    def __init__(self, device=0):
        self._lib = N.load()
        h = C.c_void_p()
        N.check(self._lib.gb_ctx_create(device, C.byref(h)))
        self.handle, self.device = h, device
        _live_contexts.add(self)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.gb_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        if not sys.is_finalizing():
            self.close()

    def synchronize(self):
        N.check(self._lib.gb_ctx_synchronize(self.handle), self.handle)

    def trim(self):
        """Return pooled (freed-batch) device memory to HIP."""
        N.check(self._lib.gb_ctx_trim(self.handle), self.handle)

    @property
    def stream(self):
        s = C.c_void_p()
        N.check(self._lib.gb_ctx_stream(self.handle, C.byref(s)), self.handle)
        return s.value

    # timed!() scopes of the reference (util/proving_process_info.rs:196-212)
    def set_profiling(self, on=True):
        N.check(self._lib.gb_ctx_set_profiling(self.handle, int(on)), self.handle)

    def scope_ms(self, scope):
        ms, cnt = C.c_double(), C.c_uint64()
        N.check(self._lib.gb_ctx_scope_ms(self.handle, scope.encode(), C.byref(ms), C.byref(cnt)), self.handle)
        return ms.value, cnt.value

    def scope_reset(self):
        N.check(self._lib.gb_ctx_scope_reset(self.handle), self.handle)

    def set_option(self, key, value):
        """gb_ctx_set_option: "copy_threads", "retry_verify", and the A/B switches of DESIGN.md section 4"""
        N.check(self._lib.gb_ctx_set_option(self.handle, key.encode(), int(value)), self.handle)

    # page-locked host memory (include/goldibear_gpu.h, "host memory"): what a host uses to place its columns where the copy
    # engine reads them directly
    def host_alloc(self, shape, dtype):
        """gb_host_alloc -> a numpy array over page-locked memory; give it back with host_free(array)"""
        count = int(np.prod(shape))
        p = C.c_void_p()
        N.check(self._lib.gb_host_alloc(self.handle, count * np.dtype(dtype).itemsize, C.byref(p)), self.handle)
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._host_blocks = getattr(self, "_host_blocks", {})
        self._host_blocks[a.ctypes.data] = p
        return a

    def host_free(self, array):
        p = getattr(self, "_host_blocks", {}).pop(array.ctypes.data)
        N.check(self._lib.gb_host_free(self.handle, p), self.handle)

    def host_register(self, array):
        """gb_host_register: page-lock an existing (contiguous) array in place"""
        N.check(self._lib.gb_host_register(self.handle, array.ctypes.data, array.nbytes), self.handle)

    def host_unregister(self, array):
        N.check(self._lib.gb_host_unregister(self.handle, array.ctypes.data), self.handle)

    def pow_grind(self, sponge_state, witness_pos, min_leading_zeros, field=N.GB_GOLDILOCKS):
        """fri_proof_of_work (fri/prover.rs:136-188): minimum nonce for the given duplex state."""
        st = np.ascontiguousarray(sponge_state, dtype=_dtype(field))
        if st.shape != ((12,) if field == N.GB_GOLDILOCKS else (16,)):
            raise N.ShapeError(N.GB_ERR_INVALID, "sponge state must have the permutation's width")
        out = C.c_uint64()
        N.check(self._lib.gb_pow_grind(self.handle, field, st.ctypes.data, witness_pos, min_leading_zeros, C.byref(out)), self.handle)
        return out.value

    def permute(self, states, field=N.GB_GOLDILOCKS):
        """PoseidonGoldilocks::poseidon on each row of `states` [count][12]."""
        x = np.ascontiguousarray(states, dtype=_dtype(field))
        out = np.empty_like(x)
        N.check(self._lib.gb_permute(self.handle, field, x.ctypes.data, out.ctypes.data, x.shape[0]), self.handle)
        return out


def _dtype(field):
    return np.uint64 if field == N.GB_GOLDILOCKS else np.uint32


def _as_input(x, field=N.GB_GOLDILOCKS):
    """numpy array (host) or torch CUDA tensor (device, same-width integer bit pattern) -> (ptr, shape, flags, keepalive)"""
    dt = _dtype(field)
    if isinstance(x, np.ndarray) or not hasattr(x, "data_ptr"):
        a = np.ascontiguousarray(x, dtype=dt)
        return a.ctypes.data, a.shape, N.GB_INPUT_HOST, a
    assert x.is_cuda and x.is_contiguous() and x.element_size() == np.dtype(dt).itemsize, \
        "device input must be a contiguous CUDA tensor of the field's word size"
    return x.data_ptr(), tuple(x.shape), N.GB_INPUT_DEVICE, x


def is_column_list(x):
    """a list / tuple of separately allocated 1-D columns: the reference's Vec<PolynomialValues<F>> / Vec<Vec<F>>"""
    return isinstance(x, (list, tuple)) and len(x) > 0 and all(hasattr(c, "shape") and len(c.shape) == 1 for c in x)


def _as_columns(cols, field=N.GB_GOLDILOCKS):
    """list of 1-D numpy arrays (host; pageable or page-locked) or 1-D torch CUDA tensors -> (const void* const* array,
    (ncols, n), flags, keepalive): the pointer table of the *_cols entry points, nothing is copied or flattened"""
    dt = _dtype(field)
    n = int(cols[0].shape[0])
    ptrs, keep = (C.c_void_p * len(cols))(), []
    dev = hasattr(cols[0], "data_ptr")
    for i, c in enumerate(cols):
        if int(c.shape[0]) != n:
            raise N.ShapeError(N.GB_ERR_INVALID, "columns differ in length")
        if dev:
            assert c.is_cuda and c.is_contiguous() and c.element_size() == np.dtype(dt).itemsize
            ptrs[i] = c.data_ptr()
        else:
            if c.dtype != dt or not c.flags.c_contiguous:
                raise N.ShapeError(N.GB_ERR_INVALID, "columns must be contiguous arrays of the field's word type")
            ptrs[i] = c.ctypes.data
        keep.append(c)
    return ptrs, (len(cols), n), (N.GB_INPUT_DEVICE if dev else N.GB_INPUT_HOST), keep


class MerkleTree:
    """View of a batch's tree: hash/merkle_tree.rs:46-62,183-222."""

    def __init__(self, batch):
        self._b = weakref.proxy(batch)  # no cycle: the batch frees its device memory when dropped

    @property
    def cap(self):
        b = self._b
        out = np.empty((1 << b.cap_height, b._hout), dtype=b._dt)
        N.check(b._lib.gb_batch_cap(b.handle, out.ctypes.data), b.ctx.handle)
        return out

    @property
    def digests(self):
        """The reference's interleaved digest vector (merkle_tree.rs:50-58), copied to the host."""
        b = self._b
        n = 2 * ((1 << (b.degree_log + b.rate_bits)) - (1 << b.cap_height))
        out = np.empty((n, b._hout), dtype=b._dt)
        N.check(b._lib.gb_batch_digests(b.handle, out.ctypes.data), b.ctx.handle)
        return out

    @property
    def leaves(self):
        b = self._b
        out = np.empty((1 << (b.degree_log + b.rate_bits), b.width), dtype=b._dt)
        N.check(b._lib.gb_batch_leaves(b.handle, out.ctypes.data), b.ctx.handle)
        return out

    def get(self, i):
        return self._b._leaf(i)[0]

    def prove(self, leaf_index):
        """MerkleTree::prove -> siblings [layers][H]"""
        return self._b._leaf(leaf_index)[1]


class PolynomialBatch:
    """fri/oracle.rs:29-158 with the data resident on the GPU."""

    def __init__(self, ctx, handle, borrowed=False):
        self.ctx, self.handle, self._lib = ctx, handle, ctx._lib
        self._borrowed = borrowed   # owned by a circuit object (its constants_sigmas_commitment): never freed here
        f, nc, dl, rb, ch, bl = C.c_uint32(), C.c_size_t(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        N.check(self._lib.gb_batch_info(handle, C.byref(f), C.byref(nc), C.byref(dl), C.byref(rb), C.byref(ch), C.byref(bl)))
        self.field, self.num_polys, self.degree_log = f.value, nc.value, dl.value
        self.rate_bits, self.cap_height, self.blinding = rb.value, ch.value, bool(bl.value)
        self._hout = 4 if self.field == N.GB_GOLDILOCKS else 8
        self._dt = _dtype(self.field)
        self.width = self.num_polys + (N.GB_SALT_SIZE if self.blinding else 0)
        self.merkle_tree = MerkleTree(self)
        _live_batches.add(self)

    @classmethod
    def _commit(cls, fn_name, ctx, cols, rate_bits, cap_height, salts, field, p3_repr=False):
        if is_column_list(cols):   # Vec<PolynomialValues<F>>: separately allocated columns, handed over as pointers
            ptr, shape, flags, keep = _as_columns(cols, field)
            fn_name += "_cols"
        else:
            ptr, shape, flags, keep = _as_input(cols, field)
        if len(shape) != 2:
            raise N.ShapeError(N.GB_ERR_INVALID, "expected a [num_polys][n] matrix")
        ncols, n = shape
        log_n = int(n).bit_length() - 1
        if n == 0 or (1 << log_n) != n:
            raise N.ShapeError(N.GB_ERR_INVALID, "polynomial length must be a power of two (util log2_strict)")
        sptr, skeep = None, None
        if salts is not None:
            sptr, sshape, sflags, skeep = _as_input(salts, field)
            if tuple(sshape) != (N.GB_SALT_SIZE, n << rate_bits) or sflags != flags:
                raise N.ShapeError(N.GB_ERR_INVALID, "salts must be [4][n << rate_bits] in the same memory space as the columns")
        h = C.c_void_p()
        if p3_repr:
            flags |= N.GB_INPUT_P3_REPR
        st = getattr(ctx._lib, fn_name)(ctx.handle, field, ptr, ncols, log_n, rate_bits, cap_height, sptr, flags, C.byref(h))
        N.check(st, ctx.handle)
        del keep, skeep
        return cls(ctx, h)

    @classmethod
    def from_values(cls, ctx, values, rate_bits, cap_height, salts=None, field=N.GB_GOLDILOCKS, p3_repr=False):
        """PolynomialBatch::from_values (oracle.rs:68-90). blinding == (salts is not None).  `values`: a [num_polys][n] matrix, or
        a list of separately allocated columns (the reference's Vec<PolynomialValues<F>>; gb_commit_values_cols).  p3_repr: host
        elements are the reference's in-memory words (GB_INPUT_P3_REPR) instead of canonical values."""
        return cls._commit("gb_commit_values", ctx, values, rate_bits, cap_height, salts, field, p3_repr)

    @classmethod
    def from_coeffs(cls, ctx, coeffs, rate_bits, cap_height, salts=None, field=N.GB_GOLDILOCKS, p3_repr=False):
        """PolynomialBatch::from_coeffs (oracle.rs:93-123)."""
        return cls._commit("gb_commit_coeffs", ctx, coeffs, rate_bits, cap_height, salts, field, p3_repr)

    def free(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None) and not self._borrowed:
            self._lib.gb_batch_free(self.handle)
        self.handle = None

    def __del__(self):
        if not sys.is_finalizing():
            self.free()

    def polynomial(self, col):
        """.polynomials[col].coeffs"""
        out = np.empty(1 << self.degree_log, dtype=self._dt)
        N.check(self._lib.gb_batch_coeffs(self.handle, col, out.ctypes.data), self.ctx.handle)
        return out

    @property
    def polynomials(self):
        return np.stack([self.polynomial(c) for c in range(self.num_polys)])

    def get_lde_values(self, index, step):
        """oracle.rs:153-158"""
        out = np.empty(self.num_polys, dtype=self._dt)
        N.check(self._lib.gb_batch_lde_values(self.handle, index, step, out.ctypes.data), self.ctx.handle)
        return out

    def eval_ext(self, z):
        """plonk/proof.rs:359-363 eval_commitment: every polynomial at the extension point z (D canonical words)
        -> [num_polys][D]"""
        d = 2 if self.field == N.GB_GOLDILOCKS else 4
        zz = np.ascontiguousarray(z, dtype=self._dt)
        if zz.shape != (d,):
            raise N.ShapeError(N.GB_ERR_INVALID, "z must have %d coordinates" % d)
        out = np.empty((self.num_polys, d), dtype=self._dt)
        N.check(self._lib.gb_batch_eval_ext(self.handle, zz.ctypes.data, out.ctypes.data), self.ctx.handle)
        return out

    def _leaf(self, i):
        row = np.empty(self.width, dtype=self._dt)
        layers = self.degree_log + self.rate_bits - self.cap_height
        sib = np.empty((max(layers, 1), self._hout), dtype=self._dt)
        n = C.c_uint32()
        N.check(self._lib.gb_batch_leaf(self.handle, i, row.ctypes.data, sib.ctypes.data, C.byref(n)), self.ctx.handle)
        return row, sib[: n.value]

    def device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        N.check(self._lib.gb_batch_device_ptrs(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value
