// Internal launch interface between the C-ABI layer (api.hip) and the gfx950 kernels.
// Every launcher enqueues on `stream` and returns immediately; no allocation, no sync.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "field_traits.hpp"
#include "gate_set.hpp"
#include "challenge_slices.hpp"

namespace gbk {

// ---------------------------------------------------------------- Goldilocks NTT / LDE (kernels_ntt.hip)

// Device-resident twiddle tables for one transform size n = 2^log_n (built once per ctx and size).
struct GlNttTables {
    u32 log_n;
    const u64* tw4096_fwd;   // [4096] w_4096^j
    const u64* tw4096_inv;   // [4096] w_4096^-j
    const u64* tw_lo_fwd;    // [1024] w_n^e            (e < 1024)
    const u64* tw_hi_fwd;    // [max(1, n/1024)] w_n^(1024 e)
    const u64* tw_lo_inv;    // same for w_n^-1
    const u64* tw_hi_inv;
    u64 n_inv;               // n^-1 mod p
    // the same tables times R = 2^64 mod p (Montgomery form) for the register-radix kernels of kernels_ntt16.hip, whose
    // multiplications by table values then end in the 8-instruction Montgomery fold (gl::mul_mont) instead of fold128's 11;
    // tw4096_fwd_m is [8192]: the powers, then the same powers in k_gl_lde_pb16's stage-1 order [slot][tid]
    const u64 *tw4096_fwd_m, *tw4096_inv_m, *tw_lo_fwd_m, *tw_hi_fwd_m, *tw_lo_inv_m, *tw_hi_inv_m;
    u64 n_inv_m;
    // 2^21 and 2^22 rows run the same three + two passes as 2^20 (round 6): the inverse transform's middle pass is a radix-32 / radix-64
    // DFT with twiddles w_{2^14}^-j (tw16k_inv_m, [2^14], Montgomery form)
    const u64* tw16k_inv_m;
    // log_n > 22: one outer radix-2^outer_bits step (ntt_outer.hpp) around transforms of log_n - outer_bits rows (`sub`); else null / 0
    const GlNttTables* sub;
    u32 outer_bits;
    const u64* tw_top_fwd;   // log_n > 22: [256] w_n^(brev8(j) 2^(log_m - 8)), log_m = log_n - outer_bits - the factor of the outer
                             // step's twiddle that varies over the 256 threads of a workgroup (ntt_outer.hpp k_lde_combine), contiguous
};
static constexpr u32 NTT_NATIVE_LOG = 22;   // largest transform the passes run without an outer step
static constexpr u32 NTT_OUTER_MAX_BITS = 4;
inline u32 ntt_outer_bits(u32 log_n) { return log_n <= NTT_NATIVE_LOG ? 0 : (log_n - NTT_NATIVE_LOG < NTT_OUTER_MAX_BITS ? log_n - NTT_NATIVE_LOG : NTT_OUTER_MAX_BITS); }

// Coset tables for the LDE of rate 2^rate_bits: coset c (leaf block c) has shift
// s_c = 7 * w_N^bitrev_r(c);  pow_lo[c][l] = s_c^l (l < 4096 or n), pow_hi[c][h] = s_c^(4096 h).
struct GlCosetTables {
    u32 rate_bits;
    const u64* pow_lo;  // [2^r][min(n,4096)]
    const u64* pow_hi;  // [2^r][max(1, n/4096)]
    const u64 *pow_lo_m, *pow_hi_m;  // the same times R (Montgomery form), for kernels_ntt16.hip
    // log_n > 22 (ntt_outer.hpp): the coset tables of the sub-transforms (shift^R) and the context's work buffer of this level - the
    // ADDRESS of the context's pointer / size, read at the call: the block moves when it grows
    const GlCosetTables* sub;
    void* const* work;
    const size_t* work_bytes;
};

// Columns per group of the inverse transform's passes from 2^18 rows up (BabyBear: twice as many): the scratch block of one group is
// reused by the next and never leaves the Infinity Cache - measured -8 % on the three memory-bound passes (HISTORY.md, round 3); the
// LDE passes are bound by VALU issue and run all columns per launch.
static constexpr size_t INTT_GROUP = 16;

// values on H_n (natural order) -> coefficients (natural order), in `coeffs` [ncols][n].
// `scratch` must hold ncols*n elements. src may equal coeffs.
void gl_intt_columns(const u64* src, u64* coeffs, u64* scratch, size_t ncols, const GlNttTables& t, hipStream_t stream);

// any u64 representative -> the canonical one, in place (GB_INPUT_P3_REPR: p3-goldilocks' in-memory words)
void gl_canonicalize(u64* p, size_t count, hipStream_t stream);

// coefficients [ncols][n] -> LDE [ncols][N] in LEAF order: lde[c][j] = P_c(7 * w_N^bitrev_logN(j))
// (fri/oracle.rs:108-109 order, no transpose / bit-reverse pass needed afterwards).
void gl_lde_columns(const u64* coeffs, u64* lde, size_t ncols, const GlNttTables& t, const GlCosetTables& ct,
                    hipStream_t stream);

// ---------------------------------------------------------------- Poseidon-12 Merkle (kernels_merkle.hip)

// leaf digests: out[j] = hash_or_noop(row j), row j = { cols[c*col_stride + j] : c < width }
void gl_merkle_leaves(const u64* cols, size_t col_stride, u32 width, u64 num_leaves, u64* out, hipStream_t stream);
// a column segment [c_begin, c_end) of every leaf's sponge, the state parked in `state` ([4 + 8 - keep_from][num_leaves], rows as used) between segments
void gl_merkle_leaves_segment(const u64* cols, size_t col_stride, u32 c_begin, u32 c_end, u64 num_leaves, u64* state, bool last,
                              u32 next_cols, u64* out, hipStream_t stream);
// one level: out[i] = two_to_one(in[2i], in[2i+1]), i < num_out
void gl_merkle_level(const u64* in, u64* out, u64 num_out, hipStream_t stream);
// one state per 16-lane row for small trees; false = not applicable (caller uses the lane-per-leaf kernel)
bool gl_fri_leaves_coop(const u64* vals, size_t len, u32 arity_bits, u64 num_leaves, u64* out, hipStream_t stream);
bool bb_fri_leaves_coop(const u32* vals, size_t len, u32 arity_bits, u64 num_leaves, u32* out, hipStream_t stream);
// level-major digests -> the reference's interleaved layout (hash/merkle_tree.rs:50-58)
void gl_digests_to_reference_layout(const u64* levels, u64* out, u32 log_leaves, u32 cap_height, hipStream_t stream);
// gather one row (width elements at stride col_stride) into dst[0..width)
void gl_gather_row(const u64* cols, size_t col_stride, u32 width, u64 index, u64* dst, hipStream_t stream);
// siblings of MerkleTree::prove(leaf) from level-major digests: dst[i] = level_i[(leaf >> i) ^ 1], i < layers
void gl_gather_siblings(const u64* levels, u32 log_leaves, u32 cap_height, u64 leaf, u64* dst, hipStream_t stream);
// dst[j] = src[bitrev_bits(j)] for `ncols` columns of 2^bits elements (salt columns -> leaf order)
void u64_bitrev_copy(const u64* src, u64* dst, u32 bits, size_t ncols, hipStream_t stream);
// leaf-order column-major [width][N] -> row-major leaves [N][width] (debug / parity export)
void u64_transpose_to_rows(const u64* cols, size_t col_stride, u32 width, u64 rows, u64* dst, hipStream_t stream);
// raw permutation of `count` states (tests / microbenchmarks)
void gl_poseidon_permute(const u64* in, u64* out, u64 count, hipStream_t stream);

// ---------------------------------------------------------------- BabyBear (kernels_bb.hip); element data in Montgomery form
struct BbNttTables {
    u32 log_n;
    const u32 *tw4096_fwd, *tw4096_inv, *tw_lo_fwd, *tw_hi_fwd, *tw_lo_inv, *tw_hi_inv;
    u32 n_inv;
    const u32* tw16k_inv;      // log_n 21, 22: see GlNttTables
    const BbNttTables* wide;   // log_n 22: the tables of the 2^20-row transform (k_bb_lde_pa16x2w's twiddles)
    const BbNttTables* sub;    // log_n > 22
    u32 outer_bits;
    const u32* tw_top_fwd;     // log_n > 22: see GlNttTables
};
struct BbCosetTables {
    u32 rate_bits;
    const u32 *pow_lo, *pow_hi;
    const BbCosetTables* fine;   // log_n 22: the cosets of the rate 2^(rate_bits + 2) over 2^20 rows, same shift (k_bb_lde_pa16x2w)
    const BbCosetTables* sub;    // log_n > 22
    void* const* work;
    const size_t* work_bytes;
};
void bb_intt_columns(const u32* src, u32* coeffs, u32* scratch, size_t ncols, const BbNttTables& t, hipStream_t stream);
bool bb_intt_columns_canonical(u32* vals, u32* coeffs, u32* scratch, size_t ncols, size_t mont_cols, const BbNttTables& t, hipStream_t stream);
void bb_lde_columns(const u32* coeffs, u32* lde, size_t ncols, const BbNttTables& t, const BbCosetTables& ct, hipStream_t stream);
void bb_merkle_leaves(const u32* cols, size_t col_stride, u32 width, u64 num_leaves, u32* out, hipStream_t stream);
void bb_merkle_leaves_segment(const u32* cols, size_t col_stride, u32 c_begin, u32 c_end, u64 num_leaves, u32* state, bool last,
                              u32 next_cols, u32* out, hipStream_t stream);  // state: [8 + 8 - keep_from][num_leaves]
void bb_merkle_level(const u32* in, u32* out, u64 num_out, hipStream_t stream);
void bb_poseidon2_permute(const u32* in, u32* out, u64 count, hipStream_t stream);  // canonical in/out
void bb_to_mont(const u32* src, u32* dst, size_t n, hipStream_t stream);
void bb_reduce_words(u32* p, size_t n, hipStream_t stream);   // any u32 -> the residue below p, in place (p3 words from a host)
void bb_from_mont(const u32* src, u32* dst, size_t n, hipStream_t stream);
void bb_gather_row(const u32* cols, size_t col_stride, u32 width, u64 index, u32* dst, hipStream_t stream);  // -> canonical
void bb_bitrev_copy_to_mont(const u32* src, u32* dst, u32 bits, size_t ncols, hipStream_t stream);
void bb_transpose_to_rows(const u32* cols, size_t col_stride, u32 width, u64 rows, u32* dst, hipStream_t stream);  // -> canonical

// ---------------------------------------------------------------- prover (kernels_prover.hip)
// Templated on the field traits of field_traits.hpp (GlF / BbF); element pointers are in the field's device form.

static constexpr u32 MAX_CHUNKS = 32, MAX_CHALLENGES = 16, MAX_RATE = 16;

template <class F>
struct PowTab {      // base^e = lo[e & (2^lo_bits - 1)] * hi[e >> lo_bits]
    const typename F::T* lo;
    const typename F::T* hi;
    u32 lo_bits;
};
template <class F>
struct ExtPowTab {   // same for an extension-field base
    const typename F::E* lo;
    const typename F::E* hi;
    u32 lo_bits;
};
template <class F>
struct CosetPow {    // per coset c: shift_c^t = lo[c][t % nlo] * hi[c][t / nlo]
    const typename F::T* lo;
    const typename F::T* hi;
    u32 nlo, nhi;
};
template <class F>
struct ZsParams {
    u32 log_n, num_routed, num_challenges, chunk /* quotient_degree_factor */, nchunks;
    PowTab<F> w_n;   // subgroup generator powers
    typename F::T betas[MAX_CHALLENGES], gammas[MAX_CHALLENGES];   // device form; by value: no upload between the transcript and the launch
};
template <class F>
struct QuotientParams {
    // rate_bits: log2 of the QUOTIENT domain's blow-up, quotient_degree_bits of prover.rs:735 (the first 2^rate_bits coset blocks of
    // the commitments' leaf-order LDEs ARE every step-th LDE point, step = 2^(fri rate_bits - quotient_degree_bits));
    // num_challenges: the challenges this launch computes
    u32 log_n, rate_bits, num_challenges, num_routed, num_constants /* selectors + constants */, num_selectors;
    u32 chunk, nchunks, nterms;
    u32 gate_constant, gate_pi, num_gate_consts;
    PowTab<F> w_N;   // LDE domain generator powers
    const typename F::T* l0;  // [N] L_0 on the LDE domain, leaf order (l0_table)
    u32 ext_gates;            // 1: the gate terms are already in qv (gate_constraints); 0: the dummy gate set, evaluated inline
    u32 stride_bits;          // log2 of a column's length in cs / wires / zs: log_n + the FRI rate_bits
    u32 k0, total_challenges; // a slice [k0, k0 + num_challenges) of total_challenges (sliced launches only; else 0, num_challenges)
};
template <class F>
struct GateParams {
    u32 log_n, rate_bits /* of the quotient domain, as QuotientParams */, num_challenges, nterms, t0 /* index of the first gate term */;
    gates::GateSet gs;
    u32 stride_bits;   // log2 of a column's length in cs / wires
};
template <class F>
struct PolyGroups {
    const typename F::T* ptr[4];
    u32 ncols[4];
    u32 ngroups;
};
template <class F>
struct PowState {    // canonical sponge state before the candidate is written at `pos`
    typename F::T s[F::SPONGE_W];
    u32 pos;
};

template <class F>
void zs_partial_products(const ZsParams<F>& p, const typename F::T* witness, const typename F::T* sigma, const typename F::T* k_is,
                         typename F::T* q_tmp, typename F::T* zloc_tmp, typename F::T* totals_tmp, u32* err, typename F::T* out, hipStream_t st);
// false if (chunk, num_challenges) cannot be run (see quotient_shape_supported); challenge counts without a specialisation of
// their own run as slices of compiled widths (the uniforms are laid out for the total either way)
template <class F>
bool quotient_values(const QuotientParams<F>& p, const typename F::T* cs, const typename F::T* wires, const typename F::T* zs,
                     const typename F::T* uniforms, typename F::T* qv, hipStream_t st);
bool quotient_shape_supported(u32 field, u32 chunk, u32 num_challenges);
// qv <- the alpha-folded gate constraints of a general gate set (kernels_gates.hip), any num_challenges <= MAX_CHALLENGES (slices)
template <class F>
bool gate_constraints(const GateParams<F>& p, const typename F::T* cs, const typename F::T* wires, const typename F::T* apow,
                      const typename F::T* pi_hash, typename F::T* qv, hipStream_t st);
// l0[j] = Z_H(x_j) / (n (x_j - 1)), zh = device copy of the 2^rate_bits values of Z_H on the cosets
template <class F>
void l0_table(u32 log_n, u32 rate_bits, const PowTab<F>& w_N, const typename F::T* zh, typename F::T* l0, hipStream_t st);
template <class F>
void quotient_combine(u32 log_n, u32 rate_bits, u32 num_challenges, const typename F::T* a, const typename F::T* mat,
                      const CosetPow<F>& inv_shift, typename F::T* out, hipStream_t st);
// lo[e] = z^e (e < nlo), hi[h] = z^(1024 h) (h < nhi): the split tables of ExtPowTab (nlo = 1024) - or a plain list of powers
// (nhi = 0) - built on the device, up to six tables per launch (a proof needs zeta, zeta_next, their inverses and FRI's alpha at once)
template <class F>
struct ExtPowJob {
    typename F::E z;
    typename F::E *lo, *hi;
    u32 nlo, nhi;
};
static constexpr u32 EXT_POW_JOBS = 6;
template <class F>
struct ExtPowJobs {
    ExtPowJob<F> j[EXT_POW_JOBS];
};
template <class F>
void ext_powtabs(const ExtPowJobs<F>& jobs, u32 njobs, hipStream_t st);
// table_k[t] = z_k^t (t < n) from the split tables, for up to two points in one launch
template <class F>
struct ExtPowTables {
    ExtPowTab<F> z[2];
    typename F::E* table[2];
};
template <class F>
void ext_pow_tables(const ExtPowTables<F>& z, u32 count, size_t n, hipStream_t st);
// out[col] = sum_t coeffs_col[t] ztab[t] for every column of up to five batches (columns numbered through the jobs in order);
// partial_tmp: ceil(n / 4096) elements per column
template <class F>
struct EvalJob {
    const typename F::T* coeffs;   // [ncols][n]
    const typename F::E* ztab;     // the point's powers (ext_pow_tables)
    u32 ncols;
};
static constexpr u32 EVAL_JOBS = 5;
template <class F>
struct EvalJobs {
    EvalJob<F> j[EVAL_JOBS];
};
template <class F>
void eval_columns(const EvalJobs<F>& jobs, u32 njobs, size_t n, typename F::E* partial_tmp, typename F::E* out, hipStream_t st);
template <class F>
void reduce_polys(const PolyGroups<F>& g, size_t n, const typename F::E* apow, typename F::E* comp, hipStream_t st);
template <class F>
void divide_by_linear_accumulate(const typename F::E* comp, size_t n, const ExtPowTab<F>& z, const ExtPowTab<F>& zinv,
                                 typename F::E shift, int first, typename F::E* sloc_tmp, typename F::E* totals_tmp,
                                 typename F::E* final_poly, hipStream_t st);
template <class F>
void ext_split(const typename F::E* src, size_t n, typename F::T* dst, hipStream_t st);
// vals = [D][len] coordinate columns in leaf order
template <class F>
void fri_leaves(const typename F::T* vals, size_t len, u32 arity_bits, u64 num_leaves, typename F::T* out, hipStream_t st);
template <class F>
void fri_fold(const typename F::T* in, size_t in_len, u32 arity_bits, typename F::E beta, typename F::T* out, hipStream_t st);
template <class F>
void pow_grind(const PowState<F>& s, u64 start, u64 count, u32 min_lz, u64* result, hipStream_t st);
// Query rounds (fri/prover.rs:190-255): every opened row and Merkle path of the proof in ONE launch per twelve trees and one
// device buffer (one read-back) - four oracle batches, then the FRI layers.  Job j writes its nidx rows at out + j.out (CANONICAL
// values) and the nidx paths (layers x H words each, level i = sibling of leaf >> i) behind them.
template <class F>
struct QueryJob {
    const typename F::T* vals;    // the oracle's LDE batch, column-major with `stride`; a FRI layer: [D][stride] coordinate columns, leaf order
    const typename F::T* levels;  // the tree's digest levels
    u64 stride;
    u64 out;
    u32 width;                    // base elements per opened row (FRI layer: D << arity_bits)
    u32 log_leaves, layers;       // layers = log_leaves - cap_height
    u32 shift;                    // leaf = query index >> shift
    u32 arity_bits, fri;
};
static constexpr u32 QUERY_JOBS = 12;
template <class F>
struct QueryJobs {
    QueryJob<F> j[QUERY_JOBS];
};
static constexpr u32 QUERY_IDX_INLINE = 64;   // up to this many query indices travel as a launch argument (idx_dev may be null then)
struct QueryIdx {
    u64 v[QUERY_IDX_INLINE];
};
template <class F>
void query_gather(const QueryJobs<F>& jobs, u32 njobs, const u64* idx_host, const u64* idx_dev, u32 nidx, typename F::T* out, hipStream_t st);

}  // namespace gbk
