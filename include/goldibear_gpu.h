/* goldibear_gpu.h - C ABI of libgoldibear_gpu.so, the MI355X (gfx950) implementation of the
 * plonky2_goldibear commitment hot path.
 *
 * The reference has no FFI: the seam is the Rust struct `PolynomialBatch` with public fields
 * (plonky2/src/fri/oracle.rs:29-40), built by `from_values` / `from_coeffs` (:68-123) and read by
 * the prover.  Each entry point below names the reference item it replaces.  INTEGRATION.md shows
 * the Rust `extern "C"` block and the `GpuPolynomialBatch` newtype that binds them.
 *
 * Conventions
 *  - every function returns a gb_status (0 = ok) and never unwinds; gb_last_error() gives text;
 *    GB_ERR_INVALID is returned where the reference would assert!/panic! on a shape violation
 *    (oracle.rs:139, merkle_tree.rs:154-157, fft.rs:174-180);
 *  - one gb_ctx per HIP device; calls on one ctx are serialised by the caller (the reference
 *    prover is single-threaded above Rayon, plonk/prover.rs:228-447); contexts are independent,
 *    which is how independent proofs shard one-per-GPU;
 *  - field elements are canonical little-endian u64 (Goldilocks) / u32 (BabyBear);
 *  - matrices are COLUMN-MAJOR [ncols][n]: the layout of Vec<PolynomialValues<F>>
 *    (iop/witness.rs:277-284) with the per-column Vecs laid end to end; the *_cols entry points take the columns where the
 *    reference has them - ncols separately allocated arrays (`const void* const* cols`, cols[i] = n elements), pageable or
 *    page-locked, no flattening copy on the host side;
 *  - gb_batch is an opaque device-resident handle; nothing large is copied back unless asked;
 *  - host input buffers (`cols`, `salts`, `witness`, `constants_sigmas`, ...) belong to the caller again as soon as the call
 *    returns (the reference moves its Vecs in): gb_commit_* wait for their uploads - not for the kernels behind them - before
 *    returning, gb_prove / gb_circuit_create end on a synchronised read-back.  Device inputs (GB_INPUT_DEVICE) are read by
 *    kernels enqueued on the context's stream and must stay valid until gb_ctx_synchronize or any read-back on that context.
 */
#ifndef GOLDIBEAR_GPU_H
#define GOLDIBEAR_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gb_ctx gb_ctx;
typedef struct gb_batch gb_batch;

typedef int32_t gb_status;
enum {
    GB_OK = 0,
    GB_ERR_INVALID = 1,     /* shape / argument violation (reference: assert!/panic!) */
    GB_ERR_HIP = 2,         /* a HIP runtime call failed */
    GB_ERR_OOM = 3,         /* device allocation failed */
    GB_ERR_UNSUPPORTED = 4  /* valid in the reference, not implemented here (e.g. log_n > 24, the lookup gates) */
};

enum { GB_GOLDILOCKS = 0, GB_BABYBEAR = 1 }; /* field tag: F = Goldilocks (Poseidon-12) | BabyBear (Poseidon2-16) */

/* `flags` of the entry points that take field-element matrices; any other bit is GB_ERR_INVALID */
enum {
    GB_INPUT_HOST = 0,   /* `cols` / `salts` / `witness` are host pointers (the drop-in case) */
    GB_INPUT_DEVICE = 1, /* they are device pointers on ctx's device (chained calls, benchmarks) */
    /* host inputs only: the elements are the reference's field types AS THEY LIE IN MEMORY (Cargo.toml:17-24 - p3-goldilocks:
     * any u64 representative of the value, not necessarily below p; p3-baby-bear / p3-monty-31: the Montgomery word
     * x * 2^32 mod p), so a Vec<F> is handed over without an as_canonical_*() pass.  Without it elements are canonical.
     * Applies to `cols` / `witness` / `salts`; results (proof bytes, accessors) are canonical either way. */
    GB_INPUT_P3_REPR = 2
};

#define GB_SALT_SIZE 4 /* fri/oracle.rs:25 */

/* ---- context ------------------------------------------------------------------------------- */
/* One context = one HIP stream (+ a copy stream) and one proving thread.  Several contexts of one process prove concurrently -
 * the way to use the GPU with recursion-sized circuits, which cannot fill it alone.  They overlap only as far as the HIP runtime
 * gives their streams hardware queues of their own: it creates GPU_MAX_HW_QUEUES of them (environment, read when the runtime
 * initialises; 4 by default), so a host that keeps more than four contexts busy exports GPU_MAX_HW_QUEUES=8 before its first
 * HIP call (six 2^12-row proofs in flight: 576 proofs/s on 4 queues, 831 on 8).  The library never changes the environment. */
gb_status gb_ctx_create(int device, gb_ctx** out);
gb_status gb_ctx_destroy(gb_ctx* ctx);
const char* gb_last_error(const gb_ctx* ctx); /* valid until the next call on ctx; ctx may be NULL */
gb_status gb_ctx_synchronize(gb_ctx* ctx);
/* Freed batches keep their device blocks in a per-context pool for reuse by later commits of the
 * same shape (the reference allocates fresh Vecs per PolynomialBatch); this returns them to HIP - together with what failed
 * attempts keep for gb_prove_retry, the work buffers of transforms above 2^22 rows and the page-locked staging ring of pageable
 * inputs (four slots of max(64 MiB, one column): 256 MiB of host memory up to 2^23 Goldilocks rows, 512 MiB at 2^24; its copy
 * threads). */
gb_status gb_ctx_trim(gb_ctx* ctx);
/* hipStream_t all work of this ctx is enqueued on (for callers that record their own events) */
gb_status gb_ctx_stream(gb_ctx* ctx, void** stream_out);
/* Tuning and debugging switches of THIS context; none changes a result.  Unknown key / value out of range: GB_ERR_INVALID.
 *   "copy_threads"   threads of the context that stage PAGEABLE host columns into the library's page-locked ring (below):
 *                    default 4; 0 = the calling thread copies; -1 = no ring, hipMemcpyAsync straight from pageable memory
 *   "retry_verify"   1: gb_prove_retry first compares the caller's whole matrix with the copy the failed attempt kept and
 *                    returns GB_ERR_INVALID if they differ in more than witness[wire][row] (one read-back of the witness) */
gb_status gb_ctx_set_option(gb_ctx* ctx, const char* key, int64_t value);

/* ---- host memory ---------------------------------------------------------------------------
 * The reference's inputs are pageable Vecs (MatrixWitness.wire_values: Vec<Vec<F>>, iop/witness.rs:277-279;
 * Vec<PolynomialValues<F>>, fri/oracle.rs:68-75).  Every host input may be pageable: big batches are staged through a
 * page-locked ring the context owns (four slots of max(64 MiB, one column), allocated at the first such call and grown when a
 * longer column arrives) by its copy threads, column by column, while
 * the columns before are transformed and hashed.  A host that can place its columns itself skips that copy: memory from
 * gb_host_alloc (hipHostMalloc) or registered with gb_host_register (hipHostRegister; unregister before freeing it) is read by
 * the copy engine directly - e.g. an allocator for the witness columns.  Registration costs about as much as one copy: it pays
 * for buffers that are reused. */
gb_status gb_host_alloc(gb_ctx* ctx, size_t bytes, void** out);
gb_status gb_host_free(gb_ctx* ctx, void* p);
gb_status gb_host_register(gb_ctx* ctx, void* p, size_t bytes);
gb_status gb_host_unregister(gb_ctx* ctx, void* p);

/* ---- timing: the reference's timed!() scopes (util/proving_process_info.rs:196-212) ---------
 * With profiling on, each commit records HIP events on ctx's stream around the scopes
 * "IFFT", "FFT + blinding", "build Merkle tree" (fri/oracle.rs:76-114; "transpose LDEs" does not
 * exist here); gb_prove adds the scopes of prove() ("compute wires commitment", ..., "fri query rounds") and two of the library's
 * own for the transforms that belong to no commitment: "quotient IFFT" (the per-coset inverse transforms of compute_quotient_polys'
 * coset_ifft) and "FRI LDE" (the layers' coset_fft, fri/prover.rs:122-125).  gb_ctx_scope_ms returns the accumulated milliseconds of
 * a scope since the last reset (*count_out: how many times it was entered) and synchronises the stream. */
gb_status gb_ctx_set_profiling(gb_ctx* ctx, int32_t on);
gb_status gb_ctx_scope_ms(gb_ctx* ctx, const char* scope, double* ms_out, uint64_t* count_out);
gb_status gb_ctx_scope_reset(gb_ctx* ctx);

/* ---- PolynomialBatch ------------------------------------------------------------------------ */
/* PolynomialBatch::from_values (fri/oracle.rs:68-90): `cols` holds ncols columns of n = 2^log_n
 * evaluations on the subgroup H_n.  Computes the coefficients, the rate-2^rate_bits LDE on the
 * coset 7*H_N, and the Merkle tree with 2^cap_height cap entries.  `salts`: NULL (blinding =
 * false) or GB_SALT_SIZE columns of N = n << rate_bits elements - the F::rand_vec columns of
 * oracle.rs:144-148, supplied by the host so that results are reproducible.
 * Errors: cap_height > log_n + rate_bits -> GB_ERR_INVALID (merkle_tree.rs:154-157). */
gb_status gb_commit_values(gb_ctx* ctx, uint32_t field, const void* cols, size_t ncols, uint32_t log_n,
                           uint32_t rate_bits, uint32_t cap_height, const void* salts, uint32_t flags,
                           gb_batch** out);
/* PolynomialBatch::from_coeffs (fri/oracle.rs:93-123): same, input already coefficients. */
gb_status gb_commit_coeffs(gb_ctx* ctx, uint32_t field, const void* cols, size_t ncols, uint32_t log_n,
                           uint32_t rate_bits, uint32_t cap_height, const void* salts, uint32_t flags,
                           gb_batch** out);
/* The same two constructors over the reference's own layout: cols[i] points to column i (n elements), each column its own
 * allocation - `values: Vec<PolynomialValues<F>>` / `polynomials: Vec<PolynomialCoeffs<F>>` (fri/oracle.rs:68-75, :93-100)
 * with cols[i] = values[i].values.as_ptr().  Host columns may be pageable (staged by the library, see "host memory") or
 * page-locked; with GB_INPUT_DEVICE they are device pointers.  `salts` stays one block of GB_SALT_SIZE columns. */
gb_status gb_commit_values_cols(gb_ctx* ctx, uint32_t field, const void* const* cols, size_t ncols, uint32_t log_n,
                                uint32_t rate_bits, uint32_t cap_height, const void* salts, uint32_t flags, gb_batch** out);
gb_status gb_commit_coeffs_cols(gb_ctx* ctx, uint32_t field, const void* const* cols, size_t ncols, uint32_t log_n,
                                uint32_t rate_bits, uint32_t cap_height, const void* salts, uint32_t flags, gb_batch** out);
gb_status gb_batch_free(gb_batch* b);

/* shape queries: .polynomials.len(), .degree_log, .rate_bits, .blinding (oracle.rs:35-39) */
gb_status gb_batch_info(const gb_batch* b, uint32_t* field, size_t* ncols, uint32_t* degree_log,
                        uint32_t* rate_bits, uint32_t* cap_height, uint32_t* blinding);

/* .merkle_tree.cap (hash/merkle_tree.rs:60-61): out[2^cap_height][H], H = 4 (GL) / 8 (BB) */
gb_status gb_batch_cap(gb_batch* b, void* out);
/* .polynomials[col].coeffs (oracle.rs:35): out[n] */
gb_status gb_batch_coeffs(gb_batch* b, size_t col, void* out);
/* get_lde_values(index, step) (oracle.rs:153-158): out[ncols] (salt columns dropped) */
gb_status gb_batch_lde_values(gb_batch* b, uint64_t index, uint64_t step, void* out);
/* MerkleTree::get(leaf_index) + MerkleTree::prove(leaf_index) (merkle_tree.rs:183-222):
 * row[ncols + salt], siblings[(log_n + rate_bits - cap_height)][H], *nsib = that count */
gb_status gb_batch_leaf(gb_batch* b, uint64_t leaf_index, void* row, void* siblings, uint32_t* nsib);
/* .merkle_tree.digests in the reference's interleaved layout (merkle_tree.rs:50-58):
 * out[2 * (N - 2^cap_height)][H].  Parity/debug aid; the device keeps level-major digests. */
gb_status gb_batch_digests(gb_batch* b, void* out);
/* .merkle_tree.leaves: out[N][ncols + salt] row-major (oracle.rs:108-109 order). Debug aid. */
gb_status gb_batch_leaves(gb_batch* b, void* out);
/* every polynomial of the batch evaluated at one extension-field point: what OpeningSet::new's eval_commitment
 * (plonk/proof.rs:359-363) computes with `p.to_extension().eval(z)`.  z: [D] canonical elements (D = 2 Goldilocks,
 * 4 BabyBear), out: [ncols][D] canonical. */
gb_status gb_batch_eval_ext(gb_batch* b, const void* z, void* out);
/* device pointers for zero-copy chaining: coeffs [ncols][n], lde [ncols+salt][N] in leaf order */
gb_status gb_batch_device_ptrs(gb_batch* b, void** coeffs, void** lde, void** digest_levels);

/* ---- circuit + prove() ------------------------------------------------------------------------
 * gb_circuit holds what CircuitBuilder::build() leaves in ProverOnlyCircuitData / CommonCircuitData
 * for the prover (plonk/circuit_builder.rs:1214-1312): the constants||sigmas commitment (committed
 * here, :1230-1239), the sigma values, k_is, and circuit_digest (:1300-1312, empty domain separator).
 * gb_circuit_create takes the gate set of the reference's dummy circuit: gates sorted by
 * (degree, id) = [NoopGate, ConstantGate{num_constants}, PublicInputGate<H>], one selector column
 * (gates/selectors.rs:142-159), evaluated inside the quotient kernel.  gb_circuit_create_gates (below) takes any gate set
 * over the eighteen gate kinds GB_GATE_* with any selector grouping - every gate DefaultGateSerializer knows except
 * LookupGate / LookupTableGate, which are GB_ERR_UNSUPPORTED - and evaluates it on the GPU as well.
 * Both of the reference's configurations are served (plonk/config.rs:119-150): GB_GOLDILOCKS = D 2, H 4,
 * Poseidon-12, 8-byte elements; GB_BABYBEAR = D 4 (x^4 - 11), H 8, Poseidon2-16, 4-byte elements.
 * Configuration range of the prover: degree_bits from 2 up to the field's two-adicity less rate_bits (32 / 27: the reference has no
 * other cap, plonk/prover.rs:228-447; 2^21 and 2^22 rows run the library's own transform passes, larger ones an outer radix step
 * around them, and what does not fit the device is GB_ERR_OOM - a 2^23-row Goldilocks proof holds ~170 GB of commitments);
 * max_quotient_degree_factor 8 (Goldilocks also 16); rate_bits from
 * log2 of that factor up to 8 - above it the quotient is computed on every 2^(rate_bits - log2 factor)-th point of the LDE, as
 * plonk/prover.rs:735-749 does (the reference's size-optimised recursion proofs use rate_bits 7 and 8,
 * recursion/recursive_verifier.rs:573-611); num_challenges 1 .. 16 (BabyBear from 4: circuit_builder.rs:1190-1192 demands
 * (31 - degree_bits) * c >= 100 - CircuitConfig.security_bits is fixed at the 100 of every configuration the reference defines,
 * plonk/circuit_data.rs:102-159, and is not a field of gb_circuit_config - and a count that fails that assert is GB_ERR_INVALID for
 * either field); FRI arity_bits 1 .. 8;
 * num_constants <= 4.  The quotient and gate kernels keep their per-challenge sums in registers and are compiled for 1 .. 4
 * challenges (Goldilocks) and 4 .. 10 (BabyBear) - the stock configurations; other counts run as slices of those widths. */
typedef struct gb_circuit gb_circuit;
typedef struct gb_circuit_config {
    uint32_t field;                 /* GB_GOLDILOCKS | GB_BABYBEAR */
    uint32_t degree_bits;
    uint32_t num_wires, num_routed_wires, num_constants; /* CircuitConfig (plonk/circuit_data.rs:63-93) */
    uint32_t num_challenges, max_quotient_degree_factor;
    uint32_t rate_bits, cap_height, proof_of_work_bits, num_query_rounds; /* FriConfig (fri/mod.rs:25-45) */
    uint32_t arity_bits, final_poly_bits;  /* FriReductionStrategy::ConstantArityBits.  A circuit whose strategy is Fixed(..) or
                                              MinSize(..) passes arity_bits = 0 (or any pair ConstantArityBits would panic on,
                                              fri/reduction_strategies.rs:45) and hands its list over after the create call
                                              (gb_circuit_set_fri_reduction_arity_bits); until then gb_prove* / gb_prove_openings /
                                              gb_verify* / gb_proof_* on the object answer GB_ERR_INVALID */
    uint32_t num_selectors;         /* 1 */
    uint32_t gate_constant, gate_pi;/* selector values of ConstantGate / PublicInputGate (NoopGate is the third) */
    uint32_t zero_knowledge;        /* CircuitConfig.zero_knowledge (= FriParams.hiding): the wires / Zs / quotient leaves carry
                                       SALT_SIZE salt elements (fri/oracle.rs:133-148): gb_prove_salted, gb_verify */
    uint32_t num_public_inputs;     /* CommonCircuitData.num_public_inputs (plonk/circuit_data.rs:590): gb_prove takes exactly
                                       this many; gb_verify / gb_verify_compressed / gb_proof_decompress reject a proof carrying
                                       any other count with GB_ERR_INVALID, as validate_proof_with_pis_shape does
                                       (plonk/validate_shape.rs:22-25) - hash_no_pad does not pad, so [a,b,c] and [a,b,c,0]
                                       have the same public-inputs hash and only this check tells them apart */
} gb_circuit_config;

/* constants_sigmas: [num_selectors + num_constants + num_routed_wires][2^degree_bits] VALUES on H_n
 * (selector, constants, sigma columns - circuit_builder.rs:1198-1229); k_is: [num_routed_wires].  Canonical words, or with
 * GB_INPUT_P3_REPR (host input) the field types' in-memory words - constants_sigmas and k_is alike. */
gb_status gb_circuit_create(gb_ctx* ctx, const gb_circuit_config* cfg, const void* constants_sigmas, const void* k_is,
                            uint32_t flags, gb_circuit** out);
/* the same with constants_sigmas as build() holds it (circuit_builder.rs:1198-1229: a Vec<PolynomialValues<F>>, one allocation per
 * column): constants_sigmas_cols[i] points to the 2^degree_bits canonical values of column i */
gb_status gb_circuit_create_cols(gb_ctx* ctx, const gb_circuit_config* cfg, const void* const* constants_sigmas_cols, const void* k_is,
                                 uint32_t flags, gb_circuit** out);
gb_status gb_circuit_free(gb_circuit* c);
/* The same for a general gate set: `gates` is CommonCircuitData.gates - sorted by (degree, id) as
 * circuit_builder.rs:1194-1196 leaves them - with selectors_info flattened into each entry (gates/selectors.rs:16-26:
 * selector_indices[i], groups[selector_indices[i]]); entry i is the gate whose selector value is i.  The constraint evaluators
 * (csrc/gates.hpp) cover NoopGate, ConstantGate{param = num_consts} (gates/constant.rs), PublicInputGate<H>
 * (gates/public_input.rs), ArithmeticGate{param = num_ops} (gates/arithmetic_base.rs) and the in-circuit hash of the field's
 * configuration - PoseidonGate for Goldilocks (gates/poseidon_goldilocks.rs), Poseidon2BabyBearGate{param = num_ops} for
 * BabyBear (gates/poseidon2_babybear.rs) - which build() needs for any circuit with public inputs (circuit_builder.rs:1126-1137);
 * and the remaining gates of the recursion circuits listed below; any other kind (LookupGate, LookupTableGate) is
 * GB_ERR_UNSUPPORTED.  cfg->num_selectors = selectors_info.groups.len(),
 * cfg->num_constants = the constant columns after the selectors (max over the gates' num_constants()); cfg->gate_constant and
 * cfg->gate_pi are ignored.  constants_sigmas: [num_selectors + num_constants + num_routed_wires][2^degree_bits]. */
#define GB_GATE_NOOP 0
#define GB_GATE_CONSTANT 1
#define GB_GATE_PUBLIC_INPUT 2
#define GB_GATE_ARITHMETIC 3
#define GB_GATE_POSEIDON 4
#define GB_GATE_POSEIDON2_BABYBEAR 5
/* the rest of the recursion circuits' gate set (the gates of the reference's RECURSIVE_VERIFIER_GL fixture and
 * ExponentiationGate); D = the extension degree of the field's configuration */
#define GB_GATE_ARITHMETIC_EXTENSION 6 /* gates/arithmetic_extension.rs   param = num_ops */
#define GB_GATE_MUL_EXTENSION 7        /* gates/multiplication_extension.rs param = num_ops */
#define GB_GATE_BASE_SUM 8             /* gates/base_sum.rs               param = num_limbs, param2 = B (0 reads as 2) */
#define GB_GATE_REDUCING 9             /* gates/reducing.rs               param = num_coeffs */
#define GB_GATE_REDUCING_EXTENSION 10  /* gates/reducing_extension.rs     param = num_coeffs */
#define GB_GATE_RANDOM_ACCESS 11       /* gates/random_access.rs          param = bits, param2 = num_copies, param3 = num_extra_constants */
#define GB_GATE_POSEIDON_MDS 12        /* gates/poseidon_goldilocks_mds.rs (Goldilocks) */
#define GB_GATE_COSET_INTERPOLATION 13 /* gates/coset_interpolation.rs    param = subgroup_bits (<= 4), param2 = degree; the
                                          barycentric weights are x_i / 2^subgroup_bits and are not passed */
#define GB_GATE_EXPONENTIATION 14      /* gates/exponentiation.rs         param = num_power_bits */
#define GB_GATE_ADD_MANY 15            /* gates/add_many.rs               param = num_addends, param2 = num_ops */
#define GB_GATE_APPLY_MAT4 16          /* gates/apply_mat4.rs             param = num_ops */
#define GB_GATE_POSEIDON2_INTERNAL_PERMUTATION 17 /* gates/poseidon2_internal_permutation.rs (BabyBear) */
typedef struct gb_gate {
    uint32_t kind;            /* GB_GATE_* */
    uint32_t param;           /* ConstantGate num_consts / ArithmeticGate, Poseidon2BabyBearGate num_ops; see above; 0 otherwise */
    uint32_t selector_index;  /* which selector column carries this gate */
    uint32_t group_start, group_end; /* the gate indices sharing that column */
    uint32_t param2, param3;  /* see the gate list above; 0 otherwise */
} gb_gate;
gb_status gb_circuit_create_gates(gb_ctx* ctx, const gb_circuit_config* cfg, const gb_gate* gates, uint32_t num_gates,
                                  const void* constants_sigmas, const void* k_is, uint32_t flags, gb_circuit** out);
gb_status gb_circuit_create_gates_cols(gb_ctx* ctx, const gb_circuit_config* cfg, const gb_gate* gates, uint32_t num_gates,
                                       const void* const* constants_sigmas_cols, const void* k_is, uint32_t flags, gb_circuit** out);
/* ProverOnlyCircuitData.constants_sigmas_commitment (plonk/circuit_data.rs:532-534), the PolynomialBatch built by
 * gb_circuit_create*: a BORROWED handle - owned by the circuit, valid until gb_circuit_free, never passed to gb_batch_free.  The
 * reference's prover reads it for the opening set (plonk/proof.rs:359-377) and the query rounds. */
gb_status gb_circuit_constants_sigmas_commitment(gb_circuit* c, gb_batch** out);
/* VerifierOnlyCircuitData: constants_sigmas_cap [2^cap_height][H] and circuit_digest [H], field elements */
gb_status gb_circuit_verifier_data(gb_circuit* c, void* cap_out, void* digest_out);
/* FriParams.reduction_arity_bits (fri/mod.rs, filled by FriConfig::fri_params from FriReductionStrategy::reduction_arity_bits,
 * fri/reduction_strategies.rs:29-56).  gb_circuit_create* / gb_verifier_create derive the list of the stock strategy
 * ConstantArityBits(cfg.arity_bits, cfg.final_poly_bits); a circuit configured with Fixed(..) or MinSize(..) hands the list its
 * CommonCircuitData holds over with the setter before the first proof (prover, verifier and proof compression all read it).
 * Rejected with GB_ERR_INVALID: more than GB_MAX_FRI_LAYERS layers, an arity_bits outside [1, 8], arities that sum past degree_bits, or a layer whose tree would be lower than cap_height
 * (MerkleTree::new's assert, hash/merkle_tree.rs:154-157).  The getter writes at most GB_MAX_FRI_LAYERS entries. */
#define GB_MAX_FRI_LAYERS 32
gb_status gb_circuit_set_fri_reduction_arity_bits(gb_circuit* c, const uint32_t* arity_bits, uint32_t num_layers);
gb_status gb_circuit_fri_reduction_arity_bits(gb_circuit* c, uint32_t* arity_bits_out, uint32_t* num_layers_out);
/* prove_with_partition_witness -> internal_prove_with_partition_witness (plonk/prover.rs:160-447):
 * witness = MatrixWitness.wire_values [num_wires][n] (iop/witness.rs:277-284), already generated.
 * Writes ProofWithPublicInputs bytes (util/serialization/mod.rs:2134-2151) to proof_out; *proof_len
 * is the size needed.  The PoW witness is the MINIMUM valid nonce (the reference's find_any with one
 * thread).  GB_ERR_PERM_ARG_ZERO mirrors ProverError::InvZeroPermArg (prover.rs:512-514): the caller
 * re-randomises the random wire and retries, as prover.rs:186-226 does (in a 31-bit field about one
 * 2^20-row proof in five needs it).  public_inputs are canonical values as u64 for either field; elements of
 * `witness` and of the proof are 8 (Goldilocks) / 4 (BabyBear) bytes (hash/hash_types.rs:39-41, :73-75). */
#define GB_ERR_PERM_ARG_ZERO 16
#define GB_ERR_OPENING_IN_SUBGROUP 17
#define GB_ERR_BUFFER_TOO_SMALL 18
gb_status gb_prove(gb_circuit* c, const void* witness, uint32_t flags, const uint64_t* public_inputs,
                   size_t num_public_inputs, void* proof_out, size_t proof_cap, size_t* proof_len);
/* The retry of prove_with_partition_witness (plonk/prover.rs:183-226).  When gb_prove returns GB_ERR_PERM_ARG_ZERO the reference
 * overwrites ONE wire - prover_data.random_wire (circuit_builder.rs:1073-1075: the last wire of the PublicInputGate row) - with a
 * fresh F::rand() and proves again.  gb_prove_retry(witness, wire, row) is that second call: `witness` is the matrix of the failed
 * gb_prove on this circuit with wire_values[wire][row] re-drawn and nothing else changed.  It returns exactly what
 * gb_prove(witness) returns - same bytes, same errors (another GB_ERR_PERM_ARG_ZERO included: call it again) - but where the failed
 * attempt has left its wires commitment behind (host or device witness, >= 2^19 leaves, more than 32 wires, `wire` among the columns of the
 * last leaf-sponge segment - true for the random wire of every stock configuration) only that wire's column is transformed again
 * and only the last absorption of every leaf sponge and the tree above are re-hashed: ~5 ms instead of ~40 at 2^20 BabyBear rows,
 * where one proof in five needs it.  Anything else - no failed attempt before it, another gb_prove* in between, a device attempt
 * retried with a host matrix, a zero-knowledge circuit - is simply the full computation.  The state of a failed attempt (the wires
 * commitment, the device copy of a host witness, the leaf sponges' state: ~12 GB at 2^20 Goldilocks rows) is held until the next
 * gb_prove* / gb_circuit_free on the circuit, gb_circuit_drop_retry, gb_ctx_trim - or until an allocation on the context would
 * otherwise fail.  A host retry trusts the kept device copy for every element but witness[wire][row]: a caller that changed anything
 * else must call gb_prove.  (gb_ctx_set_option(ctx, "retry_verify", 1) makes the library compare the whole matrix with the kept
 * copy first - a debugging aid, one read-back of the witness - and return GB_ERR_INVALID when they differ elsewhere.) */
gb_status gb_prove_retry(gb_circuit* c, const void* witness, uint32_t flags, uint32_t wire, uint64_t row, const uint64_t* public_inputs,
                         size_t num_public_inputs, void* proof_out, size_t proof_cap, size_t* proof_len);
/* gb_prove / gb_prove_retry / gb_prove_salted over MatrixWitness.wire_values as the reference holds it (iop/witness.rs:277-279:
 * Vec<Vec<F>>): wire_cols[w] points to the n = 2^degree_bits values of wire w, every column its own (pageable or page-locked)
 * allocation - wire_cols[w] = witness.wire_values[w].as_ptr(), no flattening copy.  Same results, same errors. */
gb_status gb_prove_cols(gb_circuit* c, const void* const* wire_cols, uint32_t flags, const uint64_t* public_inputs,
                        size_t num_public_inputs, void* proof_out, size_t proof_cap, size_t* proof_len);
gb_status gb_prove_retry_cols(gb_circuit* c, const void* const* wire_cols, uint32_t flags, uint32_t wire, uint64_t row,
                              const uint64_t* public_inputs, size_t num_public_inputs, void* proof_out, size_t proof_cap,
                              size_t* proof_len);
/* Release what a failed attempt left behind without proving again - a caller that gives up after GB_ERR_PERM_ARG_ZERO
 * (ProverError::TooManyPermArgFailures, prover.rs:221-225).  No-op when nothing is held. */
gb_status gb_circuit_drop_retry(gb_circuit* c);
/* The same for a circuit built with cfg.zero_knowledge (prover.rs:267,334,382: `blinding` = true for the wires, Zs / partial
 * products and quotient commitments).  salts: [3][GB_SALT_SIZE][N = 2^(degree_bits + rate_bits)] canonical elements in
 * LDE-point order - the columns the reference draws with F::rand_vec (fri/oracle.rs:144-148) - for those three commitments,
 * in the same memory space as `witness` (flags); a host input for the same reason the PublicInputGate's random wires are
 * (determinism contract: same inputs, same proof bytes).  The circuit's blinding rows (circuit_builder.rs:935-975
 * blind_and_pad) are the builder's business and part of `witness`. */
gb_status gb_prove_salted(gb_circuit* c, const void* witness, uint32_t flags, const uint64_t* public_inputs,
                          size_t num_public_inputs, const void* salts, void* proof_out, size_t proof_cap, size_t* proof_len);
gb_status gb_prove_salted_cols(gb_circuit* c, const void* const* wire_cols, uint32_t flags, const uint64_t* public_inputs,
                               size_t num_public_inputs, const void* salts, void* proof_out, size_t proof_cap, size_t* proof_len);

/* verify() of a proof produced for this circuit (plonk/verifier.rs:17-128, fri/verifier.rs:67-250,
 * plonk/get_challenges.rs:26-101), restated for the dummy gate set and run on the HOST like the reference's verifier: the
 * Fiat-Shamir replay, vanishing(zeta) == Z_H(zeta) * quotient(zeta), the proof of work, every Merkle path and the FRI
 * folding checks.  GB_OK = the proof verifies; GB_ERR_VERIFY names the failed check in gb_last_error; GB_ERR_INVALID =
 * malformed bytes. */
#define GB_ERR_VERIFY 19
gb_status gb_verify(gb_circuit* c, const void* proof, size_t proof_len);
/* Compressed proofs, on the host like the reference's (hash/path_compression.rs, fri/proof.rs:137-384, plonk/proof.rs:96-260):
 * gb_proof_compress = ProofWithPublicInputs::compress -> CompressedProofWithPublicInputs bytes (util/serialization/mod.rs:2168-2256:
 * Merkle paths of one tree share their siblings, a repeated query index is stored once, the FRI evaluation the verifier can infer
 * is dropped); gb_proof_decompress = CompressedProofWithPublicInputs::decompress (replays the transcript, recomputes the inferred
 * evaluations, re-inflates the paths) -> the original bytes; gb_verify_compressed = CompressedProofWithPublicInputs::verify.
 * *out_len receives the size written (or needed, with GB_ERR_BUFFER_TOO_SMALL).  They need the circuit's common data only, so a
 * gb_verifier_create object serves as well. */
gb_status gb_proof_compress(gb_circuit* c, const void* proof, size_t proof_len, void* out, size_t out_cap, size_t* out_len);
gb_status gb_proof_decompress(gb_circuit* c, const void* compressed, size_t compressed_len, void* out, size_t out_cap, size_t* out_len);
gb_status gb_verify_compressed(gb_circuit* c, const void* compressed, size_t compressed_len);
/* A circuit object for gb_verify alone, from what a verifier holds: CommonCircuitData (cfg + gates, as for
 * gb_circuit_create_gates; k_is [num_routed_wires]) and VerifierOnlyCircuitData (constants_sigmas_cap [2^cap_height][H],
 * circuit_digest [H]), all host pointers to canonical elements.  Touches no device: ctx may be NULL (gb_last_error(NULL) then
 * reports).  With it gb_verify checks the reference's own serialized proofs - tests/test_abi_verify_fixture.py runs the
 * RECURSIVE_VERIFIER_GL regression proof (recursion/regression_test_data.rs) through it.  Free with gb_circuit_free. */
gb_status gb_verifier_create(gb_ctx* ctx, const gb_circuit_config* cfg, const gb_gate* gates, uint32_t num_gates, const void* k_is,
                             const void* constants_sigmas_cap, const void* circuit_digest, gb_circuit** out);

/* ---- the stages of prove(), one at a time ---------------------------------------------------------------------------------
 * For a host that keeps the reference's prover loop (plonk/prover.rs:228-447) and its Challenger and swaps in the heavy calls
 * one by one: wires / Zs / quotient commitments through gb_commit_values / gb_commit_coeffs, the opening set through
 * gb_batch_eval_ext, and the three functions below.  gb_prove is exactly this sequence with the transcript kept inside the
 * library (tests/test_gpu_stage_abi.py drives the stages from Python with the oracle's Challenger and gets gb_prove's bytes).
 * Challenges are canonical field elements, host pointers.  Results are canonical; `flags` (GB_INPUT_HOST / GB_INPUT_DEVICE) names
 * the memory space of the large operand AND of the result buffer. */

/* wires_permutation_partial_products_and_zs + all_wires_permutation_partial_products (plonk/prover.rs:449-546): witness =
 * wire_values [num_wires][n] (only the routed columns are read); betas, gammas: [num_challenges].  values_out:
 * [num_challenges * (1 + num_partial_products)][n] VALUES on H_n in the order prover.rs:318-329 hands them to from_values - every
 * challenge's Z first, then each challenge's partial products.  GB_ERR_PERM_ARG_ZERO = ProverError::InvZeroPermArg (:512-514). */
gb_status gb_zs_partial_products(gb_circuit* c, const void* witness, uint32_t flags, const void* betas, const void* gammas,
                                 void* values_out);
/* the same with the witness as separately allocated columns (only wire_cols[0 .. num_routed_wires) are read) */
gb_status gb_zs_partial_products_cols(gb_circuit* c, const void* const* wire_cols, uint32_t flags, const void* betas,
                                      const void* gammas, void* values_out);
/* compute_quotient_polys (plonk/prover.rs:712-926) followed by the split into degree-n chunks (:345-376).  wires and
 * zs_partial_products are the commitments of the same proof (gb_commit_values of the witness / of gb_zs_partial_products' output),
 * made on this circuit's context; public_inputs_hash: [H]; betas, gammas, alphas: [num_challenges].  chunks_out:
 * [num_challenges * quotient_degree_factor][n] COEFFICIENTS, ready for gb_commit_coeffs (:376-387). */
gb_status gb_quotient_polys(gb_circuit* c, gb_batch* wires, gb_batch* zs_partial_products, const void* public_inputs_hash,
                            const void* betas, const void* gammas, const void* alphas, uint32_t flags, void* chunks_out);
/* Challenger<F, H> (iop/challenger.rs:18-31) by value: sponge_state (SPONGE_WIDTH = 12 / 16 words used), input_buffer and
 * output_buffer (at most SPONGE_RATE = 8 each; challenges pop from the END of output_buffer, :84-94).  Canonical values as u64 for
 * either field. */
typedef struct gb_challenger_state {
    uint64_t sponge_state[16];
    uint64_t input_buffer[8];
    uint64_t output_buffer[8];
    uint32_t input_len, output_len;
} gb_challenger_state;
/* PolynomialBatch::prove_openings (fri/oracle.rs:187-246) on the PLONK instance of this circuit (get_fri_instance(zeta),
 * plonk/circuit_data.rs:438-520: constants/sigmas, wires, Zs / partial products, quotient - the Zs also at g * zeta) followed by
 * fri_proof (fri/prover.rs:29-81): commit phase, proof of work (the MINIMUM nonce), query rounds.  zeta: [D]; `challenger` is the
 * transcript after observe_openings (plonk/prover.rs:418) and is left as the reference leaves it.  fri_proof_out receives the
 * FriProof bytes (util/serialization/mod.rs:1679-1695): commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness -
 * the part of ProofWithPublicInputs between the opening set and the public inputs; *fri_proof_len is the size needed.
 * On ANY error - GB_ERR_BUFFER_TOO_SMALL of a size query included - *challenger is left exactly as it was passed in, so the
 * call can be repeated with a buffer of *fri_proof_len bytes and returns the bytes gb_prove would (the query itself runs the
 * whole stage: ask once, keep the size - it depends on the configuration only). */
gb_status gb_prove_openings(gb_circuit* c, gb_batch* wires, gb_batch* zs_partial_products, gb_batch* quotient, const void* zeta,
                            gb_challenger_state* challenger, void* fri_proof_out, size_t fri_proof_cap, size_t* fri_proof_len);

/* fri_proof_of_work (fri/prover.rs:136-188) on its own, for a host that keeps the Challenger: sponge_state is the
 * duplex state with the pending input buffer already written over lanes 0..witness_pos-1 (`duplex_intermediate_state`,
 * :165-167; [12] u64 / [16] u32 canonical), witness_pos = input_buffer.len().  Returns the MINIMUM candidate whose
 * response `permute(state with candidate at witness_pos)[7]` has >= min_leading_zeros leading zeros as a u64
 * (min_leading_zeros = proof_of_work_bits + 64 - F::order().bits(), :147). */
gb_status gb_pow_grind(gb_ctx* ctx, uint32_t field, const void* sponge_state, uint32_t witness_pos, uint32_t min_leading_zeros,
                       uint64_t* nonce);

/* ---- bare kernels (parity tests and microbenchmarks) ---------------------------------------- */
/* `count` Poseidon-12 (GL) / Poseidon2-16 (BB) permutations: in/out [count][width], host memory.
 * PoseidonGoldilocks::poseidon (hash/poseidon_goldilocks.rs:912-922). */
gb_status gb_permute(gb_ctx* ctx, uint32_t field, const void* in, void* out, uint64_t count);

#ifdef __cplusplus
}
#endif
#endif /* GOLDIBEAR_GPU_H */
