//! Safe wrappers over the C ABI of `libgoldibear_gpu.so` (`include/goldibear_gpu.h`).
//!
//! This is the shim the reference crate would depend on under a `gpu-mi355x` feature.  Field elements
//! cross the boundary either as canonical little-endian words (`u64` Goldilocks, `u32` BabyBear - what
//! `as_canonical_u64()` / `as_canonical_u32()` give; `Repr::Canonical`) or exactly as the reference's field types lie
//! in memory (`Repr::P3InMemory` = `GB_INPUT_P3_REPR`: p3-goldilocks' possibly non-canonical `u64`, p3-baby-bear's
//! Montgomery `u32`), in which case a `Vec<F>` is handed over without any conversion pass.  Matrices are taken where the
//! reference has them: the `*_columns` methods bind the `*_cols` entry points, which read `Vec<Vec<F>>` /
//! `Vec<PolynomialValues<F>>` column by column from pageable memory - no flattening copy on the host.
//! Written against the header; not compiled in this repository (the build image has no Rust toolchain).
use std::ffi::{c_char, c_void, CStr};
use std::ptr;

#[repr(C)]
pub struct gb_ctx { _p: [u8; 0] }
#[repr(C)]
pub struct gb_batch { _p: [u8; 0] }
#[repr(C)]
pub struct gb_circuit { _p: [u8; 0] }

pub const GB_OK: i32 = 0;
pub const GB_ERR_INVALID: i32 = 1;
pub const GB_ERR_PERM_ARG_ZERO: i32 = 16;
pub const GB_ERR_VERIFY: i32 = 19;
pub const GB_GOLDILOCKS: u32 = 0;
pub const GB_BABYBEAR: u32 = 1;
pub const GB_INPUT_HOST: u32 = 0;
pub const GB_INPUT_P3_REPR: u32 = 2;
pub const GB_MAX_FRI_LAYERS: usize = 32;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct gb_circuit_config {
    pub field: u32,
    pub degree_bits: u32,
    pub num_wires: u32,
    pub num_routed_wires: u32,
    pub num_constants: u32,
    pub num_challenges: u32,
    pub max_quotient_degree_factor: u32,
    pub rate_bits: u32,
    pub cap_height: u32,
    pub proof_of_work_bits: u32,
    pub num_query_rounds: u32,
    pub arity_bits: u32,
    pub final_poly_bits: u32,
    pub num_selectors: u32,
    pub gate_constant: u32,
    pub gate_pi: u32,
    /// CircuitConfig.zero_knowledge (= FriParams.hiding): salted wires / Zs / quotient leaves
    pub zero_knowledge: u32,
    /// CommonCircuitData.num_public_inputs: proofs carrying any other count are rejected (plonk/validate_shape.rs:22-25)
    pub num_public_inputs: u32,
}

/// One entry of CommonCircuitData.gates with selectors_info flattened in (include/goldibear_gpu.h: gb_gate, GB_GATE_*).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct gb_gate {
    pub kind: u32,
    pub param: u32,
    pub selector_index: u32,
    pub group_start: u32,
    pub group_end: u32,
    pub param2: u32,
    pub param3: u32,
}

/// Challenger<F, H> by value (iop/challenger.rs:18-31) for `gb_prove_openings`: canonical values as u64 for either field;
/// challenges pop from the END of `output_buffer`.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct gb_challenger_state {
    pub sponge_state: [u64; 16],
    pub input_buffer: [u64; 8],
    pub output_buffer: [u64; 8],
    pub input_len: u32,
    pub output_len: u32,
}

/// How the words of a host matrix encode field elements (`flags` of the entry points that take matrices).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Repr {
    /// canonical values (`as_canonical_u64()` / `as_canonical_u32()`)
    Canonical,
    /// the reference's field types as they lie in memory (`GB_INPUT_P3_REPR`): `transmute`-compatible with `Vec<F>`
    P3InMemory,
}
impl Repr {
    fn flags(self) -> u32 {
        match self {
            Repr::Canonical => GB_INPUT_HOST,
            Repr::P3InMemory => GB_INPUT_HOST | GB_INPUT_P3_REPR,
        }
    }
}

extern "C" {
    fn gb_ctx_create(device: i32, out: *mut *mut gb_ctx) -> i32;
    fn gb_ctx_destroy(ctx: *mut gb_ctx) -> i32;
    fn gb_last_error(ctx: *const gb_ctx) -> *const c_char;
    fn gb_ctx_set_option(ctx: *mut gb_ctx, key: *const c_char, value: i64) -> i32;
    fn gb_host_alloc(ctx: *mut gb_ctx, bytes: usize, out: *mut *mut c_void) -> i32;
    fn gb_host_free(ctx: *mut gb_ctx, p: *mut c_void) -> i32;
    fn gb_host_register(ctx: *mut gb_ctx, p: *mut c_void, bytes: usize) -> i32;
    fn gb_host_unregister(ctx: *mut gb_ctx, p: *mut c_void) -> i32;
    fn gb_commit_values_cols(ctx: *mut gb_ctx, field: u32, cols: *const *const c_void, ncols: usize, log_n: u32, rate_bits: u32,
                             cap_height: u32, salts: *const c_void, flags: u32, out: *mut *mut gb_batch) -> i32;
    fn gb_commit_coeffs_cols(ctx: *mut gb_ctx, field: u32, cols: *const *const c_void, ncols: usize, log_n: u32, rate_bits: u32,
                             cap_height: u32, salts: *const c_void, flags: u32, out: *mut *mut gb_batch) -> i32;
    fn gb_prove_cols(c: *mut gb_circuit, wire_cols: *const *const c_void, flags: u32, public_inputs: *const u64,
                     num_public_inputs: usize, proof_out: *mut c_void, proof_cap: usize, proof_len: *mut usize) -> i32;
    fn gb_prove_retry_cols(c: *mut gb_circuit, wire_cols: *const *const c_void, flags: u32, wire: u32, row: u64,
                           public_inputs: *const u64, num_public_inputs: usize, proof_out: *mut c_void, proof_cap: usize,
                           proof_len: *mut usize) -> i32;
    fn gb_prove_salted_cols(c: *mut gb_circuit, wire_cols: *const *const c_void, flags: u32, public_inputs: *const u64,
                            num_public_inputs: usize, salts: *const c_void, proof_out: *mut c_void, proof_cap: usize,
                            proof_len: *mut usize) -> i32;
    fn gb_zs_partial_products_cols(c: *mut gb_circuit, wire_cols: *const *const c_void, flags: u32, betas: *const c_void,
                                   gammas: *const c_void, values_out: *mut c_void) -> i32;
    fn gb_commit_values(ctx: *mut gb_ctx, field: u32, cols: *const c_void, ncols: usize, log_n: u32, rate_bits: u32,
                        cap_height: u32, salts: *const c_void, flags: u32, out: *mut *mut gb_batch) -> i32;
    fn gb_commit_coeffs(ctx: *mut gb_ctx, field: u32, cols: *const c_void, ncols: usize, log_n: u32, rate_bits: u32,
                        cap_height: u32, salts: *const c_void, flags: u32, out: *mut *mut gb_batch) -> i32;
    fn gb_batch_free(b: *mut gb_batch) -> i32;
    fn gb_batch_cap(b: *mut gb_batch, out: *mut c_void) -> i32;
    fn gb_batch_coeffs(b: *mut gb_batch, col: usize, out: *mut c_void) -> i32;
    fn gb_batch_lde_values(b: *mut gb_batch, index: u64, step: u64, out: *mut c_void) -> i32;
    fn gb_batch_leaf(b: *mut gb_batch, leaf_index: u64, row: *mut c_void, siblings: *mut c_void, nsib: *mut u32) -> i32;
    fn gb_batch_eval_ext(b: *mut gb_batch, z: *const c_void, out: *mut c_void) -> i32;
    fn gb_circuit_create(ctx: *mut gb_ctx, cfg: *const gb_circuit_config, constants_sigmas: *const c_void, k_is: *const c_void,
                         flags: u32, out: *mut *mut gb_circuit) -> i32;
    fn gb_circuit_create_cols(ctx: *mut gb_ctx, cfg: *const gb_circuit_config, constants_sigmas_cols: *const *const c_void,
                              k_is: *const c_void, flags: u32, out: *mut *mut gb_circuit) -> i32;
    fn gb_circuit_create_gates_cols(ctx: *mut gb_ctx, cfg: *const gb_circuit_config, gates: *const gb_gate, num_gates: u32,
                                    constants_sigmas_cols: *const *const c_void, k_is: *const c_void, flags: u32,
                                    out: *mut *mut gb_circuit) -> i32;
    fn gb_circuit_free(c: *mut gb_circuit) -> i32;
    fn gb_circuit_verifier_data(c: *mut gb_circuit, cap_out: *mut c_void, digest_out: *mut c_void) -> i32;
    fn gb_circuit_set_fri_reduction_arity_bits(c: *mut gb_circuit, arity_bits: *const u32, num_layers: u32) -> i32;
    fn gb_circuit_fri_reduction_arity_bits(c: *mut gb_circuit, arity_bits_out: *mut u32, num_layers_out: *mut u32) -> i32;
    fn gb_prove(c: *mut gb_circuit, witness: *const c_void, flags: u32, public_inputs: *const u64, num_public_inputs: usize,
                proof_out: *mut c_void, proof_cap: usize, proof_len: *mut usize) -> i32;
    fn gb_verify(c: *mut gb_circuit, proof: *const c_void, proof_len: usize) -> i32;
    fn gb_circuit_constants_sigmas_commitment(c: *mut gb_circuit, out: *mut *mut gb_batch) -> i32;
    fn gb_zs_partial_products(c: *mut gb_circuit, witness: *const c_void, flags: u32, betas: *const c_void, gammas: *const c_void,
                              values_out: *mut c_void) -> i32;
    fn gb_quotient_polys(c: *mut gb_circuit, wires: *mut gb_batch, zs_partial_products: *mut gb_batch,
                         public_inputs_hash: *const c_void, betas: *const c_void, gammas: *const c_void, alphas: *const c_void,
                         flags: u32, chunks_out: *mut c_void) -> i32;
    fn gb_prove_openings(c: *mut gb_circuit, wires: *mut gb_batch, zs_partial_products: *mut gb_batch, quotient: *mut gb_batch,
                         zeta: *const c_void, challenger: *mut gb_challenger_state, fri_proof_out: *mut c_void,
                         fri_proof_cap: usize, fri_proof_len: *mut usize) -> i32;
    fn gb_circuit_create_gates(ctx: *mut gb_ctx, cfg: *const gb_circuit_config, gates: *const gb_gate, num_gates: u32,
                               constants_sigmas: *const c_void, k_is: *const c_void, flags: u32, out: *mut *mut gb_circuit) -> i32;
    fn gb_prove_retry(c: *mut gb_circuit, witness: *const c_void, flags: u32, wire: u32, row: u64, public_inputs: *const u64,
                      num_public_inputs: usize, proof_out: *mut c_void, proof_cap: usize, proof_len: *mut usize) -> i32;
    fn gb_circuit_drop_retry(c: *mut gb_circuit) -> i32;
    fn gb_prove_salted(c: *mut gb_circuit, witness: *const c_void, flags: u32, public_inputs: *const u64, num_public_inputs: usize,
                       salts: *const c_void, proof_out: *mut c_void, proof_cap: usize, proof_len: *mut usize) -> i32;
    fn gb_verifier_create(ctx: *mut gb_ctx, cfg: *const gb_circuit_config, gates: *const gb_gate, num_gates: u32, k_is: *const c_void,
                          constants_sigmas_cap: *const c_void, circuit_digest: *const c_void, out: *mut *mut gb_circuit) -> i32;
    fn gb_verify_compressed(c: *mut gb_circuit, compressed: *const c_void, len: usize) -> i32;
    fn gb_proof_compress(c: *mut gb_circuit, proof: *const c_void, len: usize, out: *mut c_void, cap: usize, out_len: *mut usize) -> i32;
    fn gb_proof_decompress(c: *mut gb_circuit, compressed: *const c_void, len: usize, out: *mut c_void, cap: usize,
                           out_len: *mut usize) -> i32;
}

/// Status code + the library's message.
#[derive(Debug)]
pub struct GpuError {
    pub status: i32,
    pub message: String,
}

/// The library reads fixed element counts through the raw pointers it is given: a safe wrapper checks the slice first.
fn need(what: &str, have: usize, want: usize) -> Result<(), GpuError> {
    if have == want {
        return Ok(());
    }
    Err(GpuError { status: GB_ERR_INVALID, message: format!("{what}: {have} elements, the circuit configuration needs {want}") })
}

fn check(ctx: *const gb_ctx, status: i32) -> Result<(), GpuError> {
    if status == GB_OK {
        return Ok(());
    }
    let p = unsafe { gb_last_error(ctx) };
    let message = if p.is_null() { String::new() } else { unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned() };
    Err(GpuError { status, message })
}

/// One per HIP device; calls on a context are serialised by the caller (the prover thread).
pub struct GpuContext(*mut gb_ctx);
unsafe impl Send for GpuContext {}

impl GpuContext {
    pub fn new(device: i32) -> Result<Self, GpuError> {
        let mut h = ptr::null_mut();
        check(ptr::null(), unsafe { gb_ctx_create(device, &mut h) })?;
        Ok(GpuContext(h))
    }
    /// Tuning / debugging switches (`gb_ctx_set_option`): "copy_threads", "retry_verify", ...; none changes a result.
    pub fn set_option(&self, key: &str, value: i64) -> Result<(), GpuError> {
        let k = std::ffi::CString::new(key).map_err(|_| GpuError { status: GB_ERR_INVALID, message: "key holds a NUL".into() })?;
        check(self.0, unsafe { gb_ctx_set_option(self.0, k.as_ptr(), value) })
    }
    /// Page-locked memory for `len` words (hipHostMalloc): columns built here go to the copy engine without the library's staging
    /// copy.  The natural use is an allocator for the witness columns (`MatrixWitness.wire_values`).
    pub fn host_alloc<W: Copy + Default>(&self, len: usize) -> Result<PinnedBuf<'_, W>, GpuError> {
        let mut p = ptr::null_mut();
        check(self.0, unsafe { gb_host_alloc(self.0, len * std::mem::size_of::<W>(), &mut p) })?;
        let buf = PinnedBuf { ctx: self, ptr: p as *mut W, len };
        unsafe { std::slice::from_raw_parts_mut(buf.ptr, len) }.fill(W::default());
        Ok(buf)
    }
    /// Page-lock an existing buffer in place (hipHostRegister) for as long as the guard lives; pays for buffers that are reused.
    pub fn host_register<'a, W>(&'a self, buf: &'a mut [W]) -> Result<Registered<'a, W>, GpuError> {
        check(self.0, unsafe { gb_host_register(self.0, buf.as_mut_ptr() as *mut c_void, std::mem::size_of_val(buf)) })?;
        Ok(Registered { ctx: self, buf })
    }
}
/// `len` words of page-locked host memory owned by the library's allocator (`gb_host_alloc` / `gb_host_free`).
pub struct PinnedBuf<'c, W> {
    ctx: &'c GpuContext,
    ptr: *mut W,
    len: usize,
}
impl<'c, W> std::ops::Deref for PinnedBuf<'c, W> {
    type Target = [W];
    fn deref(&self) -> &[W] {
        unsafe { std::slice::from_raw_parts(self.ptr, self.len) }
    }
}
impl<'c, W> std::ops::DerefMut for PinnedBuf<'c, W> {
    fn deref_mut(&mut self) -> &mut [W] {
        unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}
impl<'c, W> Drop for PinnedBuf<'c, W> {
    fn drop(&mut self) {
        unsafe { gb_host_free(self.ctx.0, self.ptr as *mut c_void) };
    }
}
/// A caller's buffer page-locked in place (`gb_host_register`); unregistered on drop.
pub struct Registered<'a, W> {
    ctx: &'a GpuContext,
    buf: &'a mut [W],
}
impl<'a, W> std::ops::Deref for Registered<'a, W> {
    type Target = [W];
    fn deref(&self) -> &[W] {
        self.buf
    }
}
impl<'a, W> std::ops::DerefMut for Registered<'a, W> {
    fn deref_mut(&mut self) -> &mut [W] {
        self.buf
    }
}
impl<'a, W> Drop for Registered<'a, W> {
    fn drop(&mut self) {
        unsafe { gb_host_unregister(self.ctx.0, self.buf.as_mut_ptr() as *mut c_void) };
    }
}

/// The pointer table of the `*_cols` entry points: one pointer per separately allocated column, every column `n` words long.
/// Nothing is copied: `columns[i]` is `values[i].values.as_slice()` / `witness.wire_values[i].as_slice()`.
fn column_table<W, C: AsRef<[W]>>(what: &str, columns: &[C], n: usize) -> Result<Vec<*const c_void>, GpuError> {
    let mut table = Vec::with_capacity(columns.len());
    for (i, c) in columns.iter().enumerate() {
        need(&format!("{what}[{i}]"), c.as_ref().len(), n)?;
        table.push(c.as_ref().as_ptr() as *const c_void);
    }
    Ok(table)
}
impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { gb_ctx_destroy(self.0) };
    }
}

/// `PolynomialBatch` with coefficients, LDE (leaf order) and Merkle digests resident on the GPU (fri/oracle.rs:29-40).
/// `W` is the canonical word type: `u64` for Goldilocks, `u32` for BabyBear.
pub struct GpuPolynomialBatch<'c, W> {
    ctx: &'c GpuContext,
    handle: *mut gb_batch,
    pub num_polys: usize,
    pub degree_log: u32,
    pub rate_bits: u32,
    pub cap_height: u32,
    pub blinding: bool,
    _w: std::marker::PhantomData<W>,
}

fn field_tag<W>() -> u32 {
    if std::mem::size_of::<W>() == 8 { GB_GOLDILOCKS } else { GB_BABYBEAR }
}

impl<'c, W: Copy + Default> GpuPolynomialBatch<'c, W> {
    /// `PolynomialBatch::from_values` (oracle.rs:68-90): `values` = the columns laid end to end ([ncols][n]);
    /// `salts` = [4][n << rate_bits] when blinding.
    pub fn from_values(ctx: &'c GpuContext, values: &[W], num_polys: usize, rate_bits: u32, cap_height: u32,
                       salts: Option<&[W]>) -> Result<Self, GpuError> {
        Self::commit(ctx, values, num_polys, rate_bits, cap_height, salts, false)
    }
    /// `PolynomialBatch::from_coeffs` (oracle.rs:93-123)
    pub fn from_coeffs(ctx: &'c GpuContext, coeffs: &[W], num_polys: usize, rate_bits: u32, cap_height: u32,
                       salts: Option<&[W]>) -> Result<Self, GpuError> {
        Self::commit(ctx, coeffs, num_polys, rate_bits, cap_height, salts, true)
    }
    fn commit(ctx: &'c GpuContext, cols: &[W], num_polys: usize, rate_bits: u32, cap_height: u32, salts: Option<&[W]>,
              coeffs: bool) -> Result<Self, GpuError> {
        assert!(num_polys > 0 && cols.len() % num_polys == 0);
        let n = cols.len() / num_polys;
        assert!(n.is_power_of_two());
        let degree_log = n.trailing_zeros();
        let mut h = ptr::null_mut();
        let sp = salts.map_or(ptr::null(), |s| s.as_ptr() as *const c_void);
        let f = if coeffs { gb_commit_coeffs } else { gb_commit_values };
        check(ctx.0, unsafe {
            f(ctx.0, field_tag::<W>(), cols.as_ptr() as *const c_void, num_polys, degree_log, rate_bits, cap_height, sp, GB_INPUT_HOST, &mut h)
        })?;
        Ok(Self { ctx, handle: h, num_polys, degree_log, rate_bits, cap_height, blinding: salts.is_some(), _w: std::marker::PhantomData })
    }
    /// `PolynomialBatch::from_values` (oracle.rs:68-90) over `values: Vec<PolynomialValues<F>>` AS IT IS: `columns[i]` =
    /// `values[i].values` - separately allocated, pageable - handed over as a pointer table (`gb_commit_values_cols`).
    pub fn from_value_columns<C: AsRef<[W]>>(ctx: &'c GpuContext, columns: &[C], rate_bits: u32, cap_height: u32, salts: Option<&[W]>,
                                             repr: Repr) -> Result<Self, GpuError> {
        Self::commit_columns(ctx, columns, rate_bits, cap_height, salts, repr, false)
    }
    /// `PolynomialBatch::from_coeffs` (oracle.rs:93-123) over `polynomials: Vec<PolynomialCoeffs<F>>` (`gb_commit_coeffs_cols`)
    pub fn from_coeff_columns<C: AsRef<[W]>>(ctx: &'c GpuContext, columns: &[C], rate_bits: u32, cap_height: u32, salts: Option<&[W]>,
                                             repr: Repr) -> Result<Self, GpuError> {
        Self::commit_columns(ctx, columns, rate_bits, cap_height, salts, repr, true)
    }
    fn commit_columns<C: AsRef<[W]>>(ctx: &'c GpuContext, columns: &[C], rate_bits: u32, cap_height: u32, salts: Option<&[W]>,
                                     repr: Repr, coeffs: bool) -> Result<Self, GpuError> {
        assert!(!columns.is_empty());   // oracle.rs:101
        let n = columns[0].as_ref().len();
        assert!(n.is_power_of_two());
        let degree_log = n.trailing_zeros();
        let table = column_table("columns", columns, n)?;
        if let Some(s) = salts {
            need("salts", s.len(), 4 * (n << rate_bits))?;
        }
        let sp = salts.map_or(ptr::null(), |s| s.as_ptr() as *const c_void);
        let f = if coeffs { gb_commit_coeffs_cols } else { gb_commit_values_cols };
        let mut h = ptr::null_mut();
        // the columns (and salts) are the caller's again when the call returns: the library has staged or uploaded them
        check(ctx.0, unsafe {
            f(ctx.0, field_tag::<W>(), table.as_ptr(), table.len(), degree_log, rate_bits, cap_height, sp, repr.flags(), &mut h)
        })?;
        Ok(Self { ctx, handle: h, num_polys: columns.len(), degree_log, rate_bits, cap_height, blinding: salts.is_some(),
                  _w: std::marker::PhantomData })
    }
    fn hash_len() -> usize {
        if std::mem::size_of::<W>() == 8 { 4 } else { 8 }
    }
    /// `merkle_tree.cap`: 2^cap_height hashes of NUM_HASH_OUT_ELTS words
    pub fn cap(&self) -> Result<Vec<W>, GpuError> {
        let mut out = vec![W::default(); Self::hash_len() << self.cap_height];
        check(self.ctx.0, unsafe { gb_batch_cap(self.handle, out.as_mut_ptr() as *mut c_void) })?;
        Ok(out)
    }
    /// `.polynomials[col]`
    pub fn polynomial(&self, col: usize) -> Result<Vec<W>, GpuError> {
        let mut out = vec![W::default(); 1usize << self.degree_log];
        check(self.ctx.0, unsafe { gb_batch_coeffs(self.handle, col, out.as_mut_ptr() as *mut c_void) })?;
        Ok(out)
    }
    /// `get_lde_values(index, step)` (oracle.rs:153-158)
    pub fn get_lde_values(&self, index: usize, step: usize) -> Result<Vec<W>, GpuError> {
        let mut out = vec![W::default(); self.num_polys];
        check(self.ctx.0, unsafe { gb_batch_lde_values(self.handle, index as u64, step as u64, out.as_mut_ptr() as *mut c_void) })?;
        Ok(out)
    }
    /// `merkle_tree.get(i)` and `merkle_tree.prove(i)` (merkle_tree.rs:183-222): (leaf row incl. salts, siblings)
    pub fn leaf_and_proof(&self, leaf_index: usize) -> Result<(Vec<W>, Vec<W>), GpuError> {
        let width = self.num_polys + if self.blinding { 4 } else { 0 };
        let layers = (self.degree_log + self.rate_bits - self.cap_height) as usize;
        let mut row = vec![W::default(); width];
        let mut sib = vec![W::default(); layers.max(1) * Self::hash_len()];
        let mut nsib = 0u32;
        check(self.ctx.0, unsafe {
            gb_batch_leaf(self.handle, leaf_index as u64, row.as_mut_ptr() as *mut c_void, sib.as_mut_ptr() as *mut c_void, &mut nsib)
        })?;
        sib.truncate(nsib as usize * Self::hash_len());
        Ok((row, sib))
    }
    /// every polynomial at the extension point `z` (D canonical words): plonk/proof.rs:359-363
    pub fn eval_ext(&self, z: &[W]) -> Result<Vec<W>, GpuError> {
        let d = z.len();
        let mut out = vec![W::default(); self.num_polys * d];
        check(self.ctx.0, unsafe { gb_batch_eval_ext(self.handle, z.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void) })?;
        Ok(out)
    }
}
impl<'c, W> Drop for GpuPolynomialBatch<'c, W> {
    fn drop(&mut self) {
        unsafe { gb_batch_free(self.handle) };
    }
}

/// What `CircuitBuilder::build()` leaves for the prover, resident on the GPU (circuit_builder.rs:1214-1312).
pub struct GpuCircuit<'c, W> {
    ctx: &'c GpuContext,
    handle: *mut gb_circuit,
    pub config: gb_circuit_config,
    _w: std::marker::PhantomData<W>,
}

pub enum ProveOutcome {
    Proof(Vec<u8>),
    /// `ProverError::InvZeroPermArg`: re-randomise the random wire and call again (plonk/prover.rs:183-226)
    PermArgZero,
}

impl<'c, W: Copy + Default> GpuCircuit<'c, W> {
    pub fn new(ctx: &'c GpuContext, mut config: gb_circuit_config, constants_sigmas: &[W], k_is: &[W]) -> Result<Self, GpuError> {
        config.field = field_tag::<W>();
        let mut h = ptr::null_mut();
        check(ctx.0, unsafe {
            gb_circuit_create(ctx.0, &config, constants_sigmas.as_ptr() as *const c_void, k_is.as_ptr() as *const c_void, GB_INPUT_HOST, &mut h)
        })?;
        Ok(Self { ctx, handle: h, config, _w: std::marker::PhantomData })
    }
    /// The same for any gate set the library evaluates (`gates` = CommonCircuitData.gates with selectors_info, sorted as build()
    /// leaves them); GB_ERR_UNSUPPORTED (status 4) tells the caller to keep the CPU prover for this circuit.
    pub fn with_gates(ctx: &'c GpuContext, mut config: gb_circuit_config, gates: &[gb_gate], constants_sigmas: &[W], k_is: &[W])
                      -> Result<Self, GpuError> {
        config.field = field_tag::<W>();
        let mut h = ptr::null_mut();
        check(ctx.0, unsafe {
            gb_circuit_create_gates(ctx.0, &config, gates.as_ptr(), gates.len() as u32, constants_sigmas.as_ptr() as *const c_void,
                                    k_is.as_ptr() as *const c_void, GB_INPUT_HOST, &mut h)
        })?;
        Ok(Self { ctx, handle: h, config, _w: std::marker::PhantomData })
    }
    /// `with_gates` over `constants_sigmas_vecs` as `build()` holds them (circuit_builder.rs:1198-1229): one column per `Vec`, canonical
    /// words, handed over as a pointer table (`gb_circuit_create_gates_cols`)
    pub fn with_gates_columns<C: AsRef<[W]>>(ctx: &'c GpuContext, mut config: gb_circuit_config, gates: &[gb_gate], constants_sigmas: &[C],
                                             k_is: &[W]) -> Result<Self, GpuError> {
        config.field = field_tag::<W>();
        need("constants_sigmas", constants_sigmas.len(), (config.num_selectors + config.num_constants + config.num_routed_wires) as usize)?;
        need("k_is", k_is.len(), config.num_routed_wires as usize)?;
        let table = column_table("constants_sigmas", constants_sigmas, 1usize << config.degree_bits)?;
        let mut h = ptr::null_mut();
        check(ctx.0, unsafe {
            gb_circuit_create_gates_cols(ctx.0, &config, gates.as_ptr(), gates.len() as u32, table.as_ptr(), k_is.as_ptr() as *const c_void,
                                         GB_INPUT_HOST, &mut h)
        })?;
        Ok(Self { ctx, handle: h, config, _w: std::marker::PhantomData })
    }
    /// Zero-knowledge circuits: `salts` = [3][4][N] canonical words, the F::rand_vec columns of the wires / Zs / quotient
    /// commitments (fri/oracle.rs:144-148)
    pub fn prove_salted(&self, witness: &[W], public_inputs: &[u64], salts: &[W]) -> Result<ProveOutcome, GpuError> {
        let mut buf = vec![0u8; 8 << 20];
        let mut len = 0usize;
        let st = unsafe {
            gb_prove_salted(self.handle, witness.as_ptr() as *const c_void, GB_INPUT_HOST, public_inputs.as_ptr(), public_inputs.len(),
                            salts.as_ptr() as *const c_void, buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len)
        };
        if st == GB_ERR_PERM_ARG_ZERO {
            return Ok(ProveOutcome::PermArgZero);
        }
        check(self.ctx.0, st)?;
        buf.truncate(len);
        Ok(ProveOutcome::Proof(buf))
    }
    /// `ProofWithPublicInputs::compress` on serialized bytes (plonk/proof.rs:111-122)
    pub fn compress(&self, proof: &[u8]) -> Result<Vec<u8>, GpuError> {
        let mut buf = vec![0u8; proof.len().max(1 << 16)];
        let mut len = 0usize;
        check(self.ctx.0, unsafe { gb_proof_compress(self.handle, proof.as_ptr() as *const c_void, proof.len(),
                                                     buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len) })?;
        buf.truncate(len);
        Ok(buf)
    }
    /// `CompressedProofWithPublicInputs::decompress` (plonk/proof.rs:221-236)
    pub fn decompress(&self, compressed: &[u8]) -> Result<Vec<u8>, GpuError> {
        let mut buf = vec![0u8; (2 * compressed.len()).max(1 << 16)];
        let mut len = 0usize;
        check(self.ctx.0, unsafe { gb_proof_decompress(self.handle, compressed.as_ptr() as *const c_void, compressed.len(),
                                                       buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len) })?;
        buf.truncate(len);
        Ok(buf)
    }
    /// `CompressedProofWithPublicInputs::verify` (plonk/proof.rs:238-265)
    pub fn verify_compressed(&self, compressed: &[u8]) -> Result<bool, GpuError> {
        let st = unsafe { gb_verify_compressed(self.handle, compressed.as_ptr() as *const c_void, compressed.len()) };
        if st == GB_ERR_VERIFY {
            return Ok(false);
        }
        check(self.ctx.0, st)?;
        Ok(true)
    }
    /// `FriParams.reduction_arity_bits` of a circuit whose `FriReductionStrategy` is `Fixed(..)` or `MinSize(..)`
    /// (fri/reduction_strategies.rs:29-56): hand over `common.fri_params.reduction_arity_bits` before the first proof.
    pub fn set_fri_reduction_arity_bits(&self, arity_bits: &[usize]) -> Result<(), GpuError> {
        let bits: Vec<u32> = arity_bits.iter().map(|&b| b as u32).collect();
        check(self.ctx.0, unsafe { gb_circuit_set_fri_reduction_arity_bits(self.handle, bits.as_ptr(), bits.len() as u32) })
    }
    /// the list in force (derived from ConstantArityBits unless set)
    pub fn fri_reduction_arity_bits(&self) -> Result<Vec<usize>, GpuError> {
        let mut bits = [0u32; GB_MAX_FRI_LAYERS];
        let mut n = 0u32;
        check(self.ctx.0, unsafe { gb_circuit_fri_reduction_arity_bits(self.handle, bits.as_mut_ptr(), &mut n) })?;
        Ok(bits[..n as usize].iter().map(|&b| b as usize).collect())
    }
    /// (constants_sigmas_cap, circuit_digest)
    pub fn verifier_data(&self) -> Result<(Vec<W>, Vec<W>), GpuError> {
        let hl = if std::mem::size_of::<W>() == 8 { 4 } else { 8 };
        let mut cap = vec![W::default(); hl << self.config.cap_height];
        let mut digest = vec![W::default(); hl];
        check(self.ctx.0, unsafe { gb_circuit_verifier_data(self.handle, cap.as_mut_ptr() as *mut c_void, digest.as_mut_ptr() as *mut c_void) })?;
        Ok((cap, digest))
    }
    /// `internal_prove_with_partition_witness` (plonk/prover.rs:228-447): `witness` = wire_values [num_wires][n]
    pub fn prove(&self, witness: &[W], public_inputs: &[u64]) -> Result<ProveOutcome, GpuError> {
        need("witness", witness.len(), (self.config.num_wires as usize) << self.config.degree_bits)?;
        need("public_inputs", public_inputs.len(), self.config.num_public_inputs as usize)?;
        let mut buf = vec![0u8; 8 << 20];
        let mut len = 0usize;
        let st = unsafe {
            gb_prove(self.handle, witness.as_ptr() as *const c_void, GB_INPUT_HOST, public_inputs.as_ptr(), public_inputs.len(),
                     buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len)
        };
        if st == GB_ERR_PERM_ARG_ZERO {
            return Ok(ProveOutcome::PermArgZero);
        }
        check(self.ctx.0, st)?;
        buf.truncate(len);
        Ok(ProveOutcome::Proof(buf))
    }
    /// The retry of `prove_with_partition_witness` (plonk/prover.rs:183-226): `witness` is the matrix of the `prove` call that has
    /// just returned `PermArgZero`, with `wire_values[wire][row]` (the circuit's `random_wire`) re-drawn.  Same result as `prove`.
    pub fn prove_retry(&self, witness: &[W], wire: usize, row: usize, public_inputs: &[u64]) -> Result<ProveOutcome, GpuError> {
        need("witness", witness.len(), (self.config.num_wires as usize) << self.config.degree_bits)?;
        need("public_inputs", public_inputs.len(), self.config.num_public_inputs as usize)?;
        let mut buf = vec![0u8; 8 << 20];
        let mut len = 0usize;
        let st = unsafe {
            gb_prove_retry(self.handle, witness.as_ptr() as *const c_void, GB_INPUT_HOST, wire as u32, row as u64, public_inputs.as_ptr(),
                           public_inputs.len(), buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len)
        };
        if st == GB_ERR_PERM_ARG_ZERO {
            return Ok(ProveOutcome::PermArgZero);
        }
        check(self.ctx.0, st)?;
        buf.truncate(len);
        Ok(ProveOutcome::Proof(buf))
    }
    fn finish_proof(&self, st: i32, mut buf: Vec<u8>, len: usize) -> Result<ProveOutcome, GpuError> {
        if st == GB_ERR_PERM_ARG_ZERO {
            return Ok(ProveOutcome::PermArgZero);
        }
        check(self.ctx.0, st)?;
        buf.truncate(len);
        Ok(ProveOutcome::Proof(buf))
    }
    /// `internal_prove_with_partition_witness` (plonk/prover.rs:228-447) from `MatrixWitness.wire_values` AS IT IS
    /// (iop/witness.rs:277-279: `Vec<Vec<F>>`): `wire_values[w]` = the n values of wire w, every column its own pageable
    /// allocation, handed over as a pointer table (`gb_prove_cols`) - no flattened copy of the 1 GiB witness.
    pub fn prove_columns<C: AsRef<[W]>>(&self, wire_values: &[C], public_inputs: &[u64], repr: Repr) -> Result<ProveOutcome, GpuError> {
        need("wire_values", wire_values.len(), self.config.num_wires as usize)?;
        need("public_inputs", public_inputs.len(), self.config.num_public_inputs as usize)?;
        let table = column_table("wire_values", wire_values, 1usize << self.config.degree_bits)?;
        let mut buf = vec![0u8; 8 << 20];
        let mut len = 0usize;
        let st = unsafe {
            gb_prove_cols(self.handle, table.as_ptr(), repr.flags(), public_inputs.as_ptr(), public_inputs.len(),
                          buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len)
        };
        self.finish_proof(st, buf, len)
    }
    /// `prove_retry` over the same columns with `wire_values[wire][row]` (the circuit's `random_wire`) re-drawn
    pub fn prove_retry_columns<C: AsRef<[W]>>(&self, wire_values: &[C], wire: usize, row: usize, public_inputs: &[u64], repr: Repr)
                                              -> Result<ProveOutcome, GpuError> {
        need("wire_values", wire_values.len(), self.config.num_wires as usize)?;
        need("public_inputs", public_inputs.len(), self.config.num_public_inputs as usize)?;
        let table = column_table("wire_values", wire_values, 1usize << self.config.degree_bits)?;
        let mut buf = vec![0u8; 8 << 20];
        let mut len = 0usize;
        let st = unsafe {
            gb_prove_retry_cols(self.handle, table.as_ptr(), repr.flags(), wire as u32, row as u64, public_inputs.as_ptr(),
                                public_inputs.len(), buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len)
        };
        self.finish_proof(st, buf, len)
    }
    /// `prove_salted` over separately allocated wire columns; `salts` = [3][4][N] words in the same representation
    pub fn prove_salted_columns<C: AsRef<[W]>>(&self, wire_values: &[C], public_inputs: &[u64], salts: &[W], repr: Repr)
                                               -> Result<ProveOutcome, GpuError> {
        need("wire_values", wire_values.len(), self.config.num_wires as usize)?;
        need("public_inputs", public_inputs.len(), self.config.num_public_inputs as usize)?;
        need("salts", salts.len(), 12usize << (self.config.degree_bits + self.config.rate_bits))?;
        let table = column_table("wire_values", wire_values, 1usize << self.config.degree_bits)?;
        let mut buf = vec![0u8; 8 << 20];
        let mut len = 0usize;
        let st = unsafe {
            gb_prove_salted_cols(self.handle, table.as_ptr(), repr.flags(), public_inputs.as_ptr(), public_inputs.len(),
                                 salts.as_ptr() as *const c_void, buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len)
        };
        self.finish_proof(st, buf, len)
    }
    /// `zs_partial_products` from the witness columns (only the routed ones are read); the result is canonical
    pub fn zs_partial_products_columns<C: AsRef<[W]>>(&self, wire_values: &[C], betas: &[W], gammas: &[W], repr: Repr)
                                                      -> Result<Option<Vec<W>>, GpuError> {
        let c = &self.config;
        need("wire_values", wire_values.len(), c.num_wires as usize)?;
        need("betas", betas.len(), c.num_challenges as usize)?;
        need("gammas", gammas.len(), c.num_challenges as usize)?;
        let table = column_table("wire_values", wire_values, 1usize << c.degree_bits)?;
        let chunks = (c.num_routed_wires + c.max_quotient_degree_factor - 1) / c.max_quotient_degree_factor;
        let mut out = vec![W::default(); (c.num_challenges * chunks) as usize << c.degree_bits];
        let st = unsafe {
            gb_zs_partial_products_cols(self.handle, table.as_ptr(), repr.flags(), betas.as_ptr() as *const c_void,
                                        gammas.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void)
        };
        if st == GB_ERR_PERM_ARG_ZERO {
            return Ok(None);
        }
        check(self.ctx.0, st)?;
        Ok(Some(out))
    }
    /// Give up after `PermArgZero` (`ProverError::TooManyPermArgFailures`, prover.rs:221-225): releases what the failed attempt
    /// left on the device for `prove_retry` (~12 GB at 2^20 Goldilocks rows).  No-op when nothing is held.
    pub fn drop_retry(&self) -> Result<(), GpuError> {
        check(self.ctx.0, unsafe { gb_circuit_drop_retry(self.handle) })
    }
    /// `wires_permutation_partial_products_and_zs` for every challenge (plonk/prover.rs:305-329, 449-546): the values handed to
    /// `PolynomialBatch::from_values`, Zs first.  `Ok(None)` = `ProverError::InvZeroPermArg`.
    pub fn zs_partial_products(&self, witness: &[W], betas: &[W], gammas: &[W]) -> Result<Option<Vec<W>>, GpuError> {
        let c = &self.config;
        need("witness", witness.len(), (c.num_wires as usize) << c.degree_bits)?;
        need("betas", betas.len(), c.num_challenges as usize)?;
        need("gammas", gammas.len(), c.num_challenges as usize)?;
        let chunks = (c.num_routed_wires + c.max_quotient_degree_factor - 1) / c.max_quotient_degree_factor;
        let mut out = vec![W::default(); (c.num_challenges * chunks) as usize << c.degree_bits];
        let st = unsafe {
            gb_zs_partial_products(self.handle, witness.as_ptr() as *const c_void, GB_INPUT_HOST, betas.as_ptr() as *const c_void,
                                   gammas.as_ptr() as *const c_void, out.as_mut_ptr() as *mut c_void)
        };
        if st == GB_ERR_PERM_ARG_ZERO {
            return Ok(None);
        }
        check(self.ctx.0, st)?;
        Ok(Some(out))
    }
    /// `compute_quotient_polys` + the split into chunks (plonk/prover.rs:345-376): coefficients for `from_coeffs`
    pub fn quotient_polys(&self, wires: &GpuPolynomialBatch<'c, W>, zs_partial_products: &GpuPolynomialBatch<'c, W>,
                          public_inputs_hash: &[W], betas: &[W], gammas: &[W], alphas: &[W]) -> Result<Vec<W>, GpuError> {
        let c = &self.config;
        need("public_inputs_hash", public_inputs_hash.len(), if std::mem::size_of::<W>() == 8 { 4 } else { 8 })?;
        need("betas", betas.len(), c.num_challenges as usize)?;
        need("gammas", gammas.len(), c.num_challenges as usize)?;
        need("alphas", alphas.len(), c.num_challenges as usize)?;
        let mut out = vec![W::default(); (c.num_challenges * c.max_quotient_degree_factor) as usize << c.degree_bits];
        check(self.ctx.0, unsafe {
            gb_quotient_polys(self.handle, wires.handle, zs_partial_products.handle, public_inputs_hash.as_ptr() as *const c_void,
                              betas.as_ptr() as *const c_void, gammas.as_ptr() as *const c_void, alphas.as_ptr() as *const c_void,
                              GB_INPUT_HOST, out.as_mut_ptr() as *mut c_void)
        })?;
        Ok(out)
    }
    /// `PolynomialBatch::prove_openings` on this circuit's FRI instance (fri/oracle.rs:187-246): FriProof bytes; `challenger`
    /// (the transcript after observe_openings) is advanced as the reference's is.
    pub fn prove_openings(&self, wires: &GpuPolynomialBatch<'c, W>, zs_partial_products: &GpuPolynomialBatch<'c, W>,
                          quotient: &GpuPolynomialBatch<'c, W>, zeta: &[W], challenger: &mut gb_challenger_state)
                          -> Result<Vec<u8>, GpuError> {
        need("zeta", zeta.len(), if std::mem::size_of::<W>() == 8 { 2 } else { 4 })?;   // extension degree D
        let mut buf = vec![0u8; 8 << 20];
        let mut len = 0usize;
        check(self.ctx.0, unsafe {
            gb_prove_openings(self.handle, wires.handle, zs_partial_products.handle, quotient.handle, zeta.as_ptr() as *const c_void,
                              challenger, buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len)
        })?;
        buf.truncate(len);
        Ok(buf)
    }
    /// `prover_data.constants_sigmas_commitment`: borrowed from the circuit, for `eval_ext` / `get_lde_values`
    pub fn constants_sigmas_commitment_handle(&self) -> Result<*mut gb_batch, GpuError> {
        let mut h = ptr::null_mut();
        check(self.ctx.0, unsafe { gb_circuit_constants_sigmas_commitment(self.handle, &mut h) })?;
        Ok(h)
    }
    /// `CircuitData::verify` for the gate sets the library evaluates, on the host: Ok(true), Ok(false) when a check fails
    pub fn verify(&self, proof: &[u8]) -> Result<bool, GpuError> {
        let st = unsafe { gb_verify(self.handle, proof.as_ptr() as *const c_void, proof.len()) };
        if st == GB_ERR_VERIFY {
            return Ok(false);
        }
        check(self.ctx.0, st)?;
        Ok(true)
    }
}
impl<'c, W> Drop for GpuCircuit<'c, W> {
    fn drop(&mut self) {
        unsafe { gb_circuit_free(self.handle) };
    }
}

/// `VerifierCircuitData` (plonk/circuit_data.rs:358-380): common data + verifier-only data; touches no device.
pub struct Verifier<W> {
    handle: *mut gb_circuit,
    _w: std::marker::PhantomData<W>,
}
impl<W: Copy + Default> Verifier<W> {
    pub fn new(mut config: gb_circuit_config, gates: &[gb_gate], k_is: &[W], constants_sigmas_cap: &[W], circuit_digest: &[W])
               -> Result<Self, GpuError> {
        config.field = field_tag::<W>();
        let mut h = ptr::null_mut();
        check(ptr::null(), unsafe {
            gb_verifier_create(ptr::null_mut(), &config, gates.as_ptr(), gates.len() as u32, k_is.as_ptr() as *const c_void,
                               constants_sigmas_cap.as_ptr() as *const c_void, circuit_digest.as_ptr() as *const c_void, &mut h)
        })?;
        Ok(Self { handle: h, _w: std::marker::PhantomData })
    }
    pub fn verify(&self, proof: &[u8]) -> Result<bool, GpuError> {
        let st = unsafe { gb_verify(self.handle, proof.as_ptr() as *const c_void, proof.len()) };
        if st == GB_ERR_VERIFY {
            return Ok(false);
        }
        check(ptr::null(), st)?;
        Ok(true)
    }
}
impl<W> Drop for Verifier<W> {
    fn drop(&mut self) {
        unsafe { gb_circuit_free(self.handle) };
    }
}
