// Prover kernels for gfx950: everything of prove() between the commitments.  Written once against the field
// traits of field_traits.hpp and instantiated for Goldilocks (D = 2, Poseidon-12, canonical u64) and BabyBear
// (D = 4, Poseidon2-16, 32-bit Montgomery words).
//
//   k_zs_*          wires_permutation_partial_products_and_zs        plonk/prover.rs:480-546
//   k_quotient      compute_quotient_polys + eval_vanishing_poly_base_batch for the gate set
//                   {Noop, Constant, PublicInput}                    plonk/prover.rs:712-926, vanishing_poly.rs:177-346
//   k_quotient_combine  the size-N coset_ifft's last radix-2^r step + chunking   prover.rs:921-925, :361-374
//   k_ext_powtabs / k_ext_pow_tables / k_eval_*   OpeningSet::new: the powers of zeta and g*zeta (and FRI's alpha), every batch's
//                   polynomials evaluated in two launches            plonk/proof.rs:346-387
//   k_reduce_polys / k_divide_* / k_final_poly   prove_openings      fri/oracle.rs:187-231
//   k_fri_*         fri_committed_trees fold + leaf hashing          fri/prover.rs:83-133
//   k_pow_grind     fri_proof_of_work (minimum nonce)                fri/prover.rs:136-188
//   k_query_gather  fri_prover_query_rounds: every opened row and path, one launch per twelve trees   fri/prover.rs:190-255
// All data is column-major; LDE matrices are in leaf order (see kernels_ntt.hip).  Element data is in the
// field's device form (F::T); digests and everything gathered for the proof bytes are canonical.
#include "kernels.hpp"
#include "poseidon2_bb.hpp"
#include "poseidon_gl_grouped.hpp"

namespace gbk {

__device__ __forceinline__ u32 brev32(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

template <class F>
__device__ __forceinline__ typename F::T pow_split(const PowTab<F>& t, u64 e) {
    typename F::T lo = t.lo[e & ((1u << t.lo_bits) - 1)];
    u64 h = e >> t.lo_bits;
    return h ? F::mul(lo, t.hi[h]) : lo;
}
template <class F>
__device__ __forceinline__ typename F::E pow_split(const ExtPowTab<F>& t, u64 e) {
    typename F::E l = t.lo[e & ((1u << t.lo_bits) - 1)];
    u64 h = e >> t.lo_bits;
    return h ? F::emul(l, t.hi[h]) : l;
}

// ------------------------------------------------------------------ Z and partial products

// grid ceil(n/256) * c workgroups (mapping below). q[ch][m][row] = prod_{j in chunk m} (w_j + beta k_j x + gamma) / (w_j + beta sigma_j + gamma)
template <class F>
__global__ __launch_bounds__(256) void k_zs_quotients(ZsParams<F> p, const typename F::T* __restrict__ witness,
                                                      const typename F::T* __restrict__ sigma, const typename F::T* __restrict__ k_is,
                                                      typename F::T* __restrict__ q, u32* __restrict__ err) {
    typedef typename F::T T;
    // 1-D grid of nblk * num_challenges workgroups.  The challenges of one block of 256 rows read the same witness and sigma
    // values: they are given linear ids 8 apart - the same XCD, dispatched back to back - so that all but the first find them in
    // that XCD's L2 instead of HBM (a (rows, challenge) grid re-read 1.3 GB per challenge at 2^20 rows).
    const u32 nb = (u32)((((size_t)1 << p.log_n) + 255) >> 8), L = blockIdx.x;
    u32 rb, ch;
    if ((nb & 7) == 0) {
        const u32 group = L / (8 * p.num_challenges), in = L % (8 * p.num_challenges);
        rb = group * 8 + (in & 7);
        ch = in >> 3;
    } else {
        rb = L % nb;
        ch = L / nb;
    }
    const u32 row = rb * 256 + threadIdx.x;
    const size_t n = (size_t)1 << p.log_n;
    if (row >= n) return;
    const T beta = p.betas[ch], gamma = p.gammas[ch];
    const T bx = F::mul(beta, pow_split(p.w_n, row));
    T N[MAX_CHUNKS], Dn[MAX_CHUNKS];
    for (u32 m = 0; m < p.nchunks; m++) {
        T np = F::one(), dp = F::one();
        const u32 j1 = min((m + 1) * p.chunk, p.num_routed);
        for (u32 j = m * p.chunk; j < j1; j++) {
            T w = witness[(size_t)j * n + row];
            const T wg = F::add(w, gamma);
            T num = F::add_lazy(wg, F::mul(bx, k_is[j]));
            T den = F::add_lazy(wg, F::mul(beta, sigma[(size_t)j * n + row]));
            np = F::mul_lazy(np, num);
            dp = F::mul_lazy(dp, den);
        }
        N[m] = np;
        Dn[m] = dp;
    }
    // Montgomery batch inversion of the chunk denominators
    T pref[MAX_CHUNKS];
    T acc = F::one();
    for (u32 m = 0; m < p.nchunks; m++) {
        pref[m] = acc;
        acc = F::mul(acc, Dn[m]);
    }
    if (acc == F::zero()) {  // some denominator is zero: ProverError::InvZeroPermArg (prover.rs:512-514)
        atomicOr(err, 1u);
        return;
    }
    T inv_run = F::inv(acc);
    for (u32 m = p.nchunks; m-- > 0;) {
        T dinv = F::mul(inv_run, pref[m]);
        inv_run = F::mul(inv_run, Dn[m]);
        q[((size_t)ch * p.nchunks + m) * n + row] = F::mul(N[m], dinv);
    }
}

// exclusive prefix product over rows of R(row) = prod_m q[ch][m][row], blocks of 1024 rows.
// grid (ceil(n/1024), c): zloc[ch][row] = prod of R over earlier rows of the same block; totals[ch][block]
template <class F>
__global__ __launch_bounds__(256) void k_zs_scan_local(ZsParams<F> p, const typename F::T* __restrict__ q,
                                                       typename F::T* __restrict__ zloc, typename F::T* __restrict__ totals) {
    typedef typename F::T T;
    __shared__ T sh[256];
    const size_t n = (size_t)1 << p.log_n;
    const u32 ch = blockIdx.y;
    const size_t row0 = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    T r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        T v = F::one();
        if (row0 + k < n)
            for (u32 m = 0; m < p.nchunks; m++) v = F::mul(v, q[((size_t)ch * p.nchunks + m) * n + row0 + k]);
        r[k] = v;
    }
    T mine = F::mul(F::mul(r[0], r[1]), F::mul(r[2], r[3]));
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
        T v = sh[threadIdx.x];
        T o = threadIdx.x >= off ? sh[threadIdx.x - off] : F::one();
        __syncthreads();
        sh[threadIdx.x] = F::mul(v, o);
        __syncthreads();
    }
    T excl = threadIdx.x ? sh[threadIdx.x - 1] : F::one();
    if (threadIdx.x == 255) totals[(size_t)ch * gridDim.x + blockIdx.x] = sh[255];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (row0 + k < n) zloc[(size_t)ch * n + row0 + k] = excl;
        excl = F::mul(excl, r[k]);
    }
}

// grid (c), 1024 threads: totals[ch][b] <- exclusive prefix product.  A thread takes C = ceil(nblocks / 1024) consecutive blocks
// (one up to 2^20 rows, two at 2^21, four at 2^22)
template <class F>
__global__ __launch_bounds__(1024) void k_zs_scan_totals(typename F::T* __restrict__ totals, u32 nblocks) {
    typedef typename F::T T;
    __shared__ T sh[1024];
    T* t = totals + (size_t)blockIdx.x * nblocks;
    const u32 C = (nblocks + 1023) / 1024, b0 = threadIdx.x * C;
    T mine = F::one();
    for (u32 k = 0; k < C; k++)
        if (b0 + k < nblocks) mine = F::mul(mine, t[b0 + k]);
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        T v = sh[threadIdx.x];
        T o = threadIdx.x >= off ? sh[threadIdx.x - off] : F::one();
        __syncthreads();
        sh[threadIdx.x] = F::mul(v, o);
        __syncthreads();
    }
    T run = threadIdx.x ? sh[threadIdx.x - 1] : F::one();   // product of the blocks of earlier threads
    for (u32 k = 0; k < C; k++)
        if (b0 + k < nblocks) {
            const T v = t[b0 + k];
            t[b0 + k] = run;
            run = F::mul(run, v);
        }
}

// grid (ceil(n/256), c): Z(row) = carry * zloc; partial products p_m = Z * q_0..q_m (m < num_prods)
// output columns: [Z_0..Z_{c-1}, pp_{0,*}, pp_{1,*}, ...] (prover.rs:311-317)
template <class F>
__global__ __launch_bounds__(256) void k_zs_finalize(ZsParams<F> p, const typename F::T* __restrict__ q,
                                                     const typename F::T* __restrict__ zloc, const typename F::T* __restrict__ totals,
                                                     u32 nblocks, typename F::T* __restrict__ out) {
    typedef typename F::T T;
    const size_t n = (size_t)1 << p.log_n;
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 ch = blockIdx.y;
    if (row >= n) return;
    T z = F::mul(totals[(size_t)ch * nblocks + (row >> 10)], zloc[(size_t)ch * n + row]);
    out[(size_t)ch * n + row] = z;
    const u32 num_prods = p.nchunks - 1;
    T acc = z;
    for (u32 m = 0; m < num_prods; m++) {
        acc = F::mul(acc, q[((size_t)ch * p.nchunks + m) * n + row]);
        out[((size_t)p.num_challenges + (size_t)ch * num_prods + m) * n + row] = acc;
    }
}

// ------------------------------------------------------------------ quotient

// One thread per LDE point, addressed by its leaf index j.  Writes the (unshifted) quotient value to
// qv[(k * R + coset) * n + il], il = natural index of the point inside its coset block.
// C = num_challenges and CH = chunk size (quotient_degree_factor) are compile-time so that the per-challenge
// accumulators stay in registers and a chunk's 2*CH loads are issued together.
// waves per SIMD of k_quotient: the Goldilocks instance needs 97 VGPRs unconstrained - one more than five waves allow; held to 96
// it runs 5.14 -> 5.05 ms at 2^20 rows (tools/ab_kernel_times.sh)
#define GB_QUOTIENT_OCC(F) (sizeof(typename F::T) == 8 ? 5 : 4)
template <class F, u32 C, u32 CH, bool SLICE>
__global__ __launch_bounds__(256, GB_QUOTIENT_OCC(F)) void k_quotient(QuotientParams<F> p, const typename F::T* __restrict__ cs,
                                                  const typename F::T* __restrict__ wires, const typename F::T* __restrict__ zs,
                                                  const typename F::T* __restrict__ uni, typename F::T* __restrict__ qv) {
    typedef typename F::T T;
    const u32 lgn = p.log_n, r = p.rate_bits;
    const size_t n = (size_t)1 << lgn, N = (size_t)1 << p.stride_bits;   // N: column stride of cs / wires / zs (the FRI LDE)
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= (n << r)) return;
    // SLICE: this launch owns the alpha-fold accumulators of challenges [k0, k0 + C) of CT.  reduce_with_powers_multi
    // (plonk_common.rs:105-122) folds EVERY challenge's Z and partial-product terms into every alpha's sum, so the terms of all CT
    // challenges are generated here, C at a time; the uniforms, the zs columns and the term indices are laid out for all CT.
    const u32 CT = SLICE ? p.total_challenges : C, k0 = SLICE ? p.k0 : 0;
    const u32 cidx = (u32)(j >> lgn), jl = (u32)(j & (n - 1));
    const u32 il = brev32(jl, lgn);
    const u32 imod = brev32(cidx, r);
    const u64 i = ((u64)il << r) | imod;
    const T x = F::mul(F::generator(), pow_split(p.w_N, i));  // shifted_x = g * w_N^i
    const size_t jn = ((size_t)cidx << lgn) | brev32((il + 1) & (u32)(n - 1), lgn);  // leaf of i + 2^r

    // uniform tables: [betas c][gammas c][bk c*routed][apow c*nterms][zh R][zh_inv R][pi_hash H][apow again, constant form];
    // betas and bk = beta * k_j in the field's CONSTANT form (F::cform / F::mulc: Montgomery for Goldilocks)
    const u32 nr = p.num_routed, nterms = p.nterms, R = 1u << r;
    const T* betas = uni;
    const T* gammas = uni + CT;
    const T* bk = uni + 2 * CT;
    const T* zh = uni + 2 * CT + (size_t)CT * nr + (size_t)CT * nterms;   // behind the plain alpha powers (read by the gate kernels)
    const T* zh_inv = zh + R;
    const T* pi_hash = zh_inv + R;
    const T* apow = pi_hash + F::H + (size_t)k0 * nterms;     // [c][nterms] alpha powers in constant form (F::mulc), from challenge k0
    if (SLICE) qv += ((size_t)k0 << r) * n;

    // the alpha fold: acc[k2] = sum over the constraint terms of term_t alpha_k2^t, as F::Acc sums (BabyBear keeps them unreduced)
    typename F::Acc acc[C];
#pragma unroll
    for (u32 k = 0; k < C; k++) acc[k] = F::acc_from(p.ext_gates ? qv[(((size_t)k << r) + cidx) * n + il] : F::zero());
    auto fold = [&](T term, u32 tt) {
#pragma unroll
        for (u32 k2 = 0; k2 + 1 < C; k2 += 2) F::acc_mac2(acc[k2], acc[k2 + 1], term, apow[k2 * nterms + tt], apow[(k2 + 1) * nterms + tt]);
        if (C & 1) F::acc_mac(acc[C - 1], term, apow[(C - 1) * nterms + tt]);
    };
    u32 t = 0;
    const T l0 = p.l0[j];   // L_0 on the quotient domain is a per-circuit table (k_l0_table)
    const u32 num_prods = p.nchunks - 1;
    for (u32 kt0 = 0; kt0 < CT; kt0 += C) {   // one pass unless SLICE
    // the challenge whose terms slot k generates: a slot past CT recomputes the last challenge and is not folded
    auto kt = [&](u32 k) -> u32 { return SLICE ? min(kt0 + k, CT - 1) : k; };
    auto is_live = [&](u32 k) -> bool { return !SLICE || kt0 + k < CT; };
    // L_0(x) (Z(x) - 1): eval_l_0 (zero_poly_coset.rs:58-61); term index = the challenge
    T zk[C];
#pragma unroll
    for (u32 k = 0; k < C; k++) zk[k] = zs[(size_t)kt(k) * N + j];
#pragma unroll
    for (u32 k = 0; k < C; k++) {
        if (is_live(k)) fold(F::mul(l0, F::sub(zk[k], F::one())), kt(k));
    }
    // partial-product checks (util/partial_products.rs:53-77); term index = CT + challenge * nchunks + m
    T prev[C];
#pragma unroll
    for (u32 k = 0; k < C; k++) prev[k] = zk[k];
    for (u32 m = 0; m < p.nchunks; m++) {
        T wv[CH], sg[CH];
        const u32 w0 = m * CH;
#pragma unroll
        for (u32 q = 0; q < CH; q++) {
            const u32 w = w0 + q < nr ? w0 + q : nr - 1;  // clamp: the tail chunk re-reads the last wire, masked below
            wv[q] = wires[(size_t)w * N + j];
            sg[q] = cs[(size_t)(p.num_constants + w) * N + j];
        }
        T next[C];
#pragma unroll
        for (u32 k = 0; k < C; k++)
            next[k] = m == num_prods ? zs[(size_t)kt(k) * N + jn] : zs[((size_t)CT + (size_t)kt(k) * num_prods + m) * N + j];
        T np[C], dp[C];
        const u32 live = min(CH, nr - w0);  // wires in this chunk (uniform); only the tail chunk has fewer than CH
#pragma unroll
        for (u32 k = 0; k < C; k++) np[k] = dp[k] = F::chain_one(live);  // F::mul_chain: `live` Montgomery steps end on the plain product
        if (live == CH) {
#pragma unroll
            for (u32 q = 0; q < CH; q++) {
#pragma unroll
                for (u32 k = 0; k < C; k++) {
                    const T wg = F::add(wv[q], gammas[kt(k)]);  // shared by numerator and denominator
                    T num = F::add_lazy(wg, F::mulc(x, bk[kt(k) * nr + w0 + q]));
                    T den = F::add_lazy(wg, F::mulc(sg[q], betas[kt(k)]));
                    np[k] = F::mul_chain(np[k], num);  // product chains: only multiplied again
                    dp[k] = F::mul_chain(dp[k], den);
                }
            }
        } else {
#pragma unroll
            for (u32 q = 0; q < CH; q++) {
                if (q >= live) continue;
#pragma unroll
                for (u32 k = 0; k < C; k++) {
                    const T wg = F::add(wv[q], gammas[kt(k)]);  // shared by numerator and denominator
                    T num = F::add_lazy(wg, F::mulc(x, bk[kt(k) * nr + w0 + q]));
                    T den = F::add_lazy(wg, F::mulc(sg[q], betas[kt(k)]));
                    np[k] = F::mul_chain(np[k], num);  // product chains: only multiplied again
                    dp[k] = F::mul_chain(dp[k], den);
                }
            }
        }
#pragma unroll
        for (u32 k = 0; k < C; k++) {
            const T term = F::sub(F::mul(prev[k], np[k]), F::mul(next[k], dp[k]));
            if (is_live(k)) fold(term, CT + kt(k) * p.nchunks + m);
            prev[k] = next[k];
        }
    }
    }
    t = CT + CT * p.nchunks;
    // gate constraints: filter * unfiltered, summed per constraint index (vanishing_poly.rs:741-774,
    // gates/gate.rs:188-215,391-404).  One selector group {0,1,2}; no UNUSED factor (single selector).
    // PublicInputGate<H> has H constraints, ConstantGate num_gate_consts <= H.
    if (!p.ext_gates) {
        const T s = cs[j];  // constants[0] = selector
        T f[3];
#pragma unroll
        for (u32 g = 0; g < 3; g++) {
            T v = F::one();
#pragma unroll
            for (u32 ii = 0; ii < 3; ii++)
                if (ii != g) v = F::mul(v, F::sub(F::enc(ii), s));
            f[g] = v;
        }
        const T f_pi = p.gate_pi == 0 ? f[0] : (p.gate_pi == 1 ? f[1] : f[2]);
        const T f_c = p.gate_constant == 0 ? f[0] : (p.gate_constant == 1 ? f[1] : f[2]);
#pragma unroll
        for (u32 cj = 0; cj < F::H; cj++, t++) {
            const T wv = wires[(size_t)cj * N + j];
            T term = F::mul(f_pi, F::sub(wv, pi_hash[cj]));
            if (cj < p.num_gate_consts) {
                const T kc = cs[(size_t)(p.num_selectors + cj) * N + j];
                term = F::add(term, F::mul(f_c, F::sub(kc, wv)));
            }
            fold(term, t);
        }
    }
#pragma unroll
    for (u32 k = 0; k < C; k++) qv[(((size_t)k << r) + cidx) * n + il] = F::mul(F::acc_finish(acc[k]), zh_inv[imod]);
}

// l0[j] = L_0(x_j) = Z_H(x_j) / (n (x_j - 1)) for every LDE point in leaf order (once per circuit: the quotient kernel
// would otherwise spend a field inversion, ~127 multiplications, per point and proof)
template <class F>
__global__ __launch_bounds__(256) void k_l0_table(u32 log_n, u32 rate_bits, PowTab<F> w_N, const typename F::T* __restrict__ zh,
                                                  typename F::T* __restrict__ l0) {
    typedef typename F::T T;
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const u32 cidx = (u32)(j >> log_n), jl = (u32)(j & (n - 1));
    const u32 imod = brev32(cidx, rate_bits);
    const u64 i = ((u64)brev32(jl, log_n) << rate_bits) | imod;
    const T x = F::mul(F::generator(), pow_split(w_N, i));
    l0[j] = F::mul(zh[imod], F::inv(F::mul(F::enc((u64)n), F::sub(x, F::one()))));
}

// After the per-block natural->natural inverse NTTs: a_c[t] are the coefficients of R_c(s_c X).
// chunk_m[t] = g^(-n m) / R * sum_c zeta_c^(-m) * s_c^(-t) * a_c[t]     (mat[m][c] holds the constant part)
template <class F>
__global__ __launch_bounds__(256) void k_quotient_combine(u32 log_n, u32 rate_bits, const typename F::T* __restrict__ a,
                                                          const typename F::T* __restrict__ mat, CosetPow<F> inv_shift,
                                                          typename F::T* __restrict__ out) {
    typedef typename F::T T;
    const size_t n = (size_t)1 << log_n;
    const u32 R = 1u << rate_bits;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 k = blockIdx.y;
    if (t >= n) return;
    T v[MAX_RATE];
    for (u32 c = 0; c < R; c++) {
        T s = inv_shift.lo[(size_t)c * inv_shift.nlo + (t & (inv_shift.nlo - 1))];
        size_t h = t / inv_shift.nlo;
        if (h) s = F::mul(s, inv_shift.hi[(size_t)c * inv_shift.nhi + h]);
        v[c] = F::mul(a[((size_t)k * R + c) * n + t], s);
    }
    for (u32 m = 0; m < R; m++) {
        T acc = F::zero();
        for (u32 c = 0; c < R; c++) acc = F::add(acc, F::mul(v[c], mat[m * R + c]));
        out[((size_t)k * R + m) * n + t] = acc;
    }
}

// ------------------------------------------------------------------ openings: sum_t c_t z^t

// the split power tables of an extension element on the device (round 6: on the host they were 8 x 1024 extension products and
// 16 small uploads per proof - a third of a millisecond of a 4 ms recursion-shaped proof): lo[e] = z^e, e < 1024; hi[h] = z^(1024 h)
template <class F>
__global__ __launch_bounds__(256) void k_ext_powtabs(ExtPowJobs<F> jobs) {   // blockIdx.y = table
    const ExtPowJob<F>& J = jobs.j[blockIdx.y];
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t < J.nlo) J.lo[t] = epow<F>(J.z, t);
    else if (t - J.nlo < J.nhi) J.hi[t - J.nlo] = epow<F>(J.z, (u64)(t - J.nlo) << 10);
}

// table_k[t] = z_k^t, t < n  (blockIdx.y = k)
template <class F>
__global__ __launch_bounds__(256) void k_ext_pow_tables(ExtPowTables<F> z, size_t n) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) z.table[blockIdx.y][t] = pow_split(z.z[blockIdx.y], t);
}

template <class F>
__device__ __forceinline__ typename F::E block_reduce_add(typename F::E v, typename F::E* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (u32 off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = F::eadd(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    typename F::E r = sh[0];
    __syncthreads();
    return r;
}

// Evaluation of every column of up to five coefficient batches at their points (OpeningSet::new: constants/sigmas, wires, Z at
// zeta and g zeta, quotient) in two launches; columns are numbered through the jobs in order.
// grid (nchunks = ceil(n / 4096), total columns): partial[col][chunk] = sum over the chunk of c_t * z^t
template <class F>
__global__ __launch_bounds__(256) void k_eval_partial(EvalJobs<F> jobs, size_t n, typename F::E* __restrict__ partial) {
    typedef typename F::E E;
    __shared__ E sh[256];
    u32 col = blockIdx.y, j = 0;
    while (col >= jobs.j[j].ncols) col -= jobs.j[j++].ncols;
    const typename F::T* c = jobs.j[j].coeffs + (size_t)col * n;
    const E* ztab = jobs.j[j].ztab;
    const size_t base = (size_t)blockIdx.x * 4096;
    E acc = F::ezero();
    for (u32 k = 0; k < 16; k++) {
        size_t t = base + k * 256 + threadIdx.x;
        if (t < n) acc = F::eadd(acc, F::escale(ztab[t], c[t]));
    }
    E r = block_reduce_add<F>(acc, sh);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = r;
}
// grid (total columns): out[col] = sum of partial[col][*]
template <class F>
__global__ __launch_bounds__(256) void k_eval_final(const typename F::E* __restrict__ partial, u32 nchunks,
                                                    typename F::E* __restrict__ out) {
    typedef typename F::E E;
    __shared__ E sh[256];
    E acc = F::ezero();
    for (u32 k = threadIdx.x; k < nchunks; k += 256) acc = F::eadd(acc, partial[(size_t)blockIdx.x * nchunks + k]);
    E r = block_reduce_add<F>(acc, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// ------------------------------------------------------------------ prove_openings

// comp[t] = sum_j alpha^j * poly_j[t]  (reduce_polys_base, util/reducing.rs:89-103) over up to 4 column groups
template <class F>
__global__ __launch_bounds__(256) void k_reduce_polys(PolyGroups<F> g, size_t n, const typename F::E* __restrict__ apow,
                                                      typename F::E* __restrict__ comp) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    typename F::E acc = F::ezero();
    u32 jj = 0;
    for (u32 o = 0; o < g.ngroups; o++) {
        const typename F::T* base = g.ptr[o];
        for (u32 j = 0; j < g.ncols[o]; j++, jj++) acc = F::eadd(acc, F::escale(apow[jj], base[(size_t)j * n + t]));
    }
    comp[t] = acc;
}

// divide_by_linear (polynomial/division.rs:75-88): q[t] = sum_{u > t} c_u z^(u-t-1) = z^-(t+1) * S_{t+1},
// S_t = sum_{u >= t} c_u z^u.  Step 1: w_u = c_u z^u and block-local suffix sums (blocks of 1024).
template <class F>
__global__ __launch_bounds__(256) void k_divide_local(const typename F::E* __restrict__ comp, size_t n, ExtPowTab<F> z,
                                                      typename F::E* __restrict__ sloc, typename F::E* __restrict__ totals) {
    typedef typename F::E E;
    __shared__ E sh[256];
    const size_t u0 = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    E w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = u0 + k < n ? F::emul(comp[u0 + k], pow_split(z, u0 + k)) : F::ezero();
    E mine = F::eadd(F::eadd(w[0], w[1]), F::eadd(w[2], w[3]));
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {  // inclusive SUFFIX scan
        E v = sh[threadIdx.x];
        E o = threadIdx.x + off < 256 ? sh[threadIdx.x + off] : F::ezero();
        __syncthreads();
        sh[threadIdx.x] = F::eadd(v, o);
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = sh[0];
    E run = threadIdx.x + 1 < 256 ? sh[threadIdx.x + 1] : F::ezero();  // sum of later threads in this block
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        run = F::eadd(run, w[k]);
        if (u0 + k < n) sloc[u0 + k] = run;  // local S_t (this block only)
    }
}
// single block, 1024 threads: totals[b] <- sum of totals of LATER blocks (exclusive suffix); a thread takes C = ceil(nblocks / 1024)
// consecutive blocks (one up to 2^20 coefficients)
template <class F>
__global__ __launch_bounds__(1024) void k_divide_totals(typename F::E* __restrict__ totals, u32 nblocks) {
    typedef typename F::E E;
    __shared__ E sh[1024];
    const u32 C = (nblocks + 1023) / 1024, b0 = threadIdx.x * C;
    E mine = F::ezero();
    for (u32 k = 0; k < C; k++)
        if (b0 + k < nblocks) mine = F::eadd(mine, totals[b0 + k]);
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        E v = sh[threadIdx.x];
        E o = threadIdx.x + off < 1024 ? sh[threadIdx.x + off] : F::ezero();
        __syncthreads();
        sh[threadIdx.x] = F::eadd(v, o);
        __syncthreads();
    }
    E run = threadIdx.x + 1 < 1024 ? sh[threadIdx.x + 1] : F::ezero();   // sum over the blocks of later threads
    for (u32 k = C; k-- > 0;)
        if (b0 + k < nblocks) {
            const E v = totals[b0 + k];
            totals[b0 + k] = run;
            run = F::eadd(run, v);
        }
}
// final[t] = final[t] * shift + q[t], q[t] = zinv^(t+1) * S_{t+1}, q[n-1] = 0   (fri/oracle.rs:218-223)
template <class F>
__global__ __launch_bounds__(256) void k_divide_apply(const typename F::E* __restrict__ sloc, const typename F::E* __restrict__ totals,
                                                      size_t n, ExtPowTab<F> zinv, typename F::E shift, int first,
                                                      typename F::E* __restrict__ final_poly) {
    typedef typename F::E E;
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    E q = F::ezero();
    if (t + 1 < n) {
        size_t u = t + 1;
        E S = F::eadd(sloc[u], totals[u >> 10]);
        q = F::emul(S, pow_split(zinv, u));
    }
    final_poly[t] = first ? q : F::eadd(F::emul(final_poly[t], shift), q);
}
// split an extension array into D base columns [D][n] (for the coordinate-wise NTT)
template <class F>
__global__ __launch_bounds__(256) void k_ext_split(const typename F::E* __restrict__ src, size_t n, typename F::T* __restrict__ dst) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    typename F::E v = src[t];
#pragma unroll
    for (u32 k = 0; k < F::D; k++) dst[(size_t)k * n + t] = F::coord(v, k);
}

// ------------------------------------------------------------------ FRI

// Sponge over the field's permutation, rate 8, fed with DEVICE-form elements; out() gives canonical words.
template <class F>
struct Sponge;
// The Goldilocks sponge runs the permutation with its MDS layers on the matrix pipe and its partial rounds in groups
// (poseidon_gl_grouped.hpp), like the tree kernels: init() and permute() must be reached by all 64 lanes of a wave, so the kernels below clamp the index of lanes past
// the end instead of returning early.
template <>
struct Sponge<GlF> {
    static constexpr int LDS_V4 = poseidon_gl::GROUP_LDS_V4;   // the workgroup's operand table of the grouped partial rounds
    u64 s[12];
    poseidon_gl::MdsOperand amat;
    const poseidon_gl::v4i* gops;
    __device__ __forceinline__ void init(poseidon_gl::v4i* lds) {   // all threads of the workgroup (fills the table, one barrier)
        amat = poseidon_gl::mds_mfma_matrix();
        poseidon_gl::group_ops_init(lds);
        gops = lds + (threadIdx.x & 63);
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = 0;
    }
    __device__ __forceinline__ void permute() {   // plain lazy residues in and out, as poseidon_gl::permute_lazy
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = poseidon_gl::to_mont(s[i]);
        poseidon_gl::permute_mont_mfma_grouped(s, amat, gops);
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = poseidon_gl::mont_fold((u32)s[i], (u32)(s[i] >> 32), 0u, 0u);
    }
    __device__ __forceinline__ u64 out(int i) { return poseidon_gl::to_canonical(s[i]); }
    __device__ __forceinline__ void set_canonical(int i, u64 v) { s[i] = v; }
};
template <>
struct Sponge<BbF> {
    static constexpr int LDS_V4 = 1;   // none needed
    u32 s[16];
    __device__ __forceinline__ void init(poseidon_gl::v4i*) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = 0;
    }
    __device__ __forceinline__ void permute() { poseidon2_bb::permute(s); }
    __device__ __forceinline__ u32 out(int i) { return bb::from_mont(s[i]); }
    __device__ __forceinline__ void set_canonical(int i, u32 v) { s[i] = bb::to_mont(v); }
};

// leaf m = flatten(values[arity*m .. arity*(m+1))) in bit-reversed (= leaf) order (fri/prover.rs:101-107);
// vals = [D][len] coordinate columns, out = H canonical words per leaf.
template <class F>
__global__ __launch_bounds__(256) void k_fri_leaves(const typename F::T* __restrict__ vals, size_t len, u32 arity_bits,
                                                    u64 num_leaves, typename F::T* __restrict__ out) {
    typedef typename F::T T;
    constexpr u32 D = F::D, H = F::H, PER = 8 / D;  // extension elements per absorption of 8 base elements
    const u64 m0 = (u64)blockIdx.x * 256 + threadIdx.x;
    const bool live = m0 < num_leaves;
    const u64 m = live ? m0 : num_leaves - 1;      // no early exit: the sponge's permutation is wave-wide (MFMA)
    const u32 arity = 1u << arity_bits;
    const T* a = vals + (m << arity_bits);
    T* o = out + (size_t)H * m;
    if (D * arity <= H) {  // hash_or_noop (plonk/config.rs:70-84)
        if (!live) return;
        for (u32 i = 0; i < H; i++) o[i] = 0;
        for (u32 k = 0; k < arity; k++)
            for (u32 d = 0; d < D; d++) o[D * k + d] = (T)F::dec(a[(size_t)d * len + k]);
        return;
    }
    __shared__ poseidon_gl::v4i sponge_lds[Sponge<F>::LDS_V4];
    Sponge<F> sp;
    sp.init(sponge_lds);
    for (u32 k0 = 0; k0 < arity; k0 += PER) {
#pragma unroll
        for (u32 k = 0; k < PER; k++)
            if (k0 + k < arity) {
#pragma unroll
                for (u32 d = 0; d < D; d++) sp.s[D * k + d] = a[(size_t)d * len + k0 + k];
            }
        sp.permute();
    }
    if (!live) return;
#pragma unroll
    for (u32 i = 0; i < H; i++) o[i] = sp.out(i);
}

// coeffs' [m] = sum_t coeffs[arity*m + t] beta^t  (reduce_with_powers, fri/prover.rs:112-121); in/out as [D][len] columns
template <class F>
__global__ __launch_bounds__(256) void k_fri_fold(const typename F::T* __restrict__ in, size_t in_len, u32 arity_bits,
                                                  typename F::E beta, typename F::T* __restrict__ out) {
    typedef typename F::E E;
    const size_t out_len = in_len >> arity_bits;
    size_t m = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= out_len) return;
    const u32 arity = 1u << arity_bits;
    E acc = F::ezero();
    for (u32 t = arity; t-- > 0;) {
        size_t idx = (m << arity_bits) + t;
        E v = F::ezero();
#pragma unroll
        for (u32 k = 0; k < F::D; k++) F::set_coord(v, k, in[(size_t)k * in_len + idx]);
        acc = F::eadd(F::emul(acc, beta), v);
    }
#pragma unroll
    for (u32 k = 0; k < F::D; k++) out[(size_t)k * out_len + m] = F::coord(acc, k);
}

// proof of work: candidates start .. start+count; result = min satisfying candidate (fri/prover.rs:169-180)
template <class F>
__global__ __launch_bounds__(256) void k_pow_grind(PowState<F> st, u64 start, u64 count, u32 min_leading_zeros,
                                                   u64* __restrict__ result) {
    const u64 g0 = (u64)blockIdx.x * 256 + threadIdx.x;
    const bool live = g0 < count;
    const u64 g = live ? g0 : count - 1;           // no early exit (wave-wide permutation)
    u64 cand = start + g;
    __shared__ poseidon_gl::v4i sponge_lds[Sponge<F>::LDS_V4];
    Sponge<F> sp;
    sp.init(sponge_lds);
    // runtime position, static register indexing
#pragma unroll
    for (int i = 0; i < (int)F::SPONGE_W; i++) sp.set_canonical(i, (u32)i == st.pos ? (typename F::T)cand : st.s[i]);
    sp.permute();
    u64 resp = sp.out(7);
    u32 lz = resp ? (u32)__clzll((long long)resp) : 64;
    if (live && lz >= min_leading_zeros) atomicMin(result, cand);
}

// ------------------------------------------------------------------ query gathers (outputs canonical)

// blockIdx.y = job.  rows[q][e] = vals[e * stride + leaf_q]  (FRI layer: v_(e % D)[arity * leaf_q + e / D]);  path[q][i][0..H) =
// level_i[(leaf_q >> i) ^ 1];  leaf_q = idx[q] >> shift
template <class F, bool INLINE>
__global__ __launch_bounds__(256) void k_query_gather(QueryJobs<F> jobs, QueryIdx inl, const u64* __restrict__ idx, u32 nidx,
                                                      typename F::T* __restrict__ out) {
    constexpr u32 H = F::H, D = F::D;
    const QueryJob<F>& J = jobs.j[blockIdx.y];
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 nrow = nidx * J.width;
    const bool row = g < nrow;
    if (!row) g -= nrow;
    if (!row && g >= nidx * J.layers * H) return;
    const u32 q = row ? g / J.width : (g / H) / J.layers;
    u64 x = 0;
    if (INLINE) {   // the indices came with the launch (no upload): static indexing of the argument, one select per slot
#pragma unroll
        for (u32 i = 0; i < QUERY_IDX_INLINE; i++) x = q == i ? inl.v[i] : x;
    } else {
        x = idx[q];
    }
    const u64 leaf = x >> J.shift;
    if (row) {
        const u32 e = g % J.width;
        const size_t src = J.fri ? (size_t)(e % D) * J.stride + (leaf << J.arity_bits) + e / D : (size_t)e * J.stride + leaf;
        out[J.out + g] = (typename F::T)F::dec(J.vals[src]);
        return;
    }
    const u32 e = g % H, i = (g / H) % J.layers;
    const u64 N = (u64)1 << J.log_leaves;
    const u64 off = 2 * N - ((2 * N) >> i);
    out[J.out + nrow + g] = J.levels[H * (off + ((leaf >> i) ^ 1)) + e];
}

// ------------------------------------------------------------------ launchers

static inline u32 nblk(size_t n, u32 bs) { return (u32)((n + bs - 1) / bs); }

template <class F>
void zs_partial_products(const ZsParams<F>& p, const typename F::T* witness, const typename F::T* sigma, const typename F::T* k_is,
                         typename F::T* q_tmp, typename F::T* zloc_tmp, typename F::T* totals_tmp, u32* err, typename F::T* out, hipStream_t st) {
    const size_t n = (size_t)1 << p.log_n;
    const u32 nb1024 = nblk(n, 1024);
    hipLaunchKernelGGL(k_zs_quotients<F>, dim3(nblk(n, 256) * p.num_challenges), dim3(256), 0, st, p, witness, sigma, k_is, q_tmp, err);
    hipLaunchKernelGGL(k_zs_scan_local<F>, dim3(nb1024, p.num_challenges), dim3(256), 0, st, p, q_tmp, zloc_tmp, totals_tmp);
    hipLaunchKernelGGL(k_zs_scan_totals<F>, dim3(p.num_challenges), dim3(1024), 0, st, totals_tmp, nb1024);
    hipLaunchKernelGGL(k_zs_finalize<F>, dim3(nblk(n, 256), p.num_challenges), dim3(256), 0, st, p, q_tmp, zloc_tmp, totals_tmp,
                       nb1024, out);
}

// A challenge count with no specialisation of its own runs as slices of compiled widths <= WMAX (balanced: every slice
// floor or ceil of count / slices); the hot counts (Goldilocks 1..4, BabyBear 6..10) are one plain launch, as before.
#define GB_Q(FF, CC, HH, SL) hipLaunchKernelGGL((k_quotient<FF, CC, HH, SL>), grid, block, 0, st, q, cs, wires, zs, uniforms, qv)
// Instances: the stock factor 8 with every hot challenge count as one plain launch (Goldilocks 1 .. 4, BabyBear 4 .. 10, slices 5 .. 8
// wide); the other factors the reference's CircuitConfig allows (2, 4, 16: plonk/circuit_data.rs:86, prover.rs:735-749; round 6) with
// a reduced set - Goldilocks 1 .. 2, BabyBear 4 .. 8 - every other count as slices of those.
template <class F, u32 CH>
static bool quotient_slice(QuotientParams<F> q, u32 width, bool slice, const typename F::T* cs, const typename F::T* wires,
                           const typename F::T* zs, const typename F::T* uniforms, typename F::T* qv, dim3 grid, dim3 block, hipStream_t st) {
    q.num_challenges = width;
#define GB_QW(W) case W: if (slice) GB_Q(F, W, CH, true); else GB_Q(F, W, CH, false); return true
    if constexpr (F::TAG == 0) {
        switch (width) {
            GB_QW(1);
            GB_QW(2);
            case 3: if constexpr (CH == 8) { if (slice) GB_Q(F, 3, CH, true); else GB_Q(F, 3, CH, false); return true; } else return false;
            case 4: if constexpr (CH == 8) { if (slice) GB_Q(F, 4, CH, true); else GB_Q(F, 4, CH, false); return true; } else return false;
            default: return false;
        }
    } else if constexpr (CH == 8) {
        if (slice) switch (width) {
            case 5: GB_Q(F, 5, CH, true); return true;
            case 6: GB_Q(F, 6, CH, true); return true;
            case 7: GB_Q(F, 7, CH, true); return true;
            case 8: GB_Q(F, 8, CH, true); return true;
            default: return false;
        }
        switch (width) {
            case 4: GB_Q(F, 4, CH, false); return true;
            case 5: GB_Q(F, 5, CH, false); return true;
            case 6: GB_Q(F, 6, CH, false); return true;
            case 7: GB_Q(F, 7, CH, false); return true;
            case 8: GB_Q(F, 8, CH, false); return true;
            case 9: GB_Q(F, 9, CH, false); return true;
            case 10: GB_Q(F, 10, CH, false); return true;
            default: return false;
        }
    } else {
        switch (width) {
            GB_QW(4);
            GB_QW(5);
            GB_QW(6);
            GB_QW(7);
            GB_QW(8);
            default: return false;
        }
    }
#undef GB_QW
}
#undef GB_Q
static inline u32 quotient_gl_wmax(u32 chunk) { return chunk == 8 ? 4 : 2; }
static inline u32 quotient_bb_wmax(u32 chunk) { return chunk == 8 ? 10 : 8; }
template <class F>
bool quotient_values(const QuotientParams<F>& p, const typename F::T* cs, const typename F::T* wires, const typename F::T* zs,
                     const typename F::T* uniforms, typename F::T* qv, hipStream_t st) {
    const size_t NQ = (size_t)1 << (p.log_n + p.rate_bits);
    const dim3 grid(nblk(NQ, 256)), block(256);
    u32 widths[MAX_CHALLENGES];
    const u32 ns = challenge_slices(F::TAG, quotient_gl_wmax(p.chunk), p.num_challenges, widths, quotient_bb_wmax(p.chunk));
    if (!ns || !quotient_shape_supported(F::TAG, p.chunk, p.num_challenges)) return false;
    QuotientParams<F> q = p;
    q.total_challenges = p.num_challenges;
    q.k0 = 0;
    for (u32 i = 0; i < ns; q.k0 += widths[i], i++) {
        bool ok;
        switch (p.chunk) {
            case 2: ok = quotient_slice<F, 2>(q, widths[i], ns > 1, cs, wires, zs, uniforms, qv, grid, block, st); break;
            case 4: ok = quotient_slice<F, 4>(q, widths[i], ns > 1, cs, wires, zs, uniforms, qv, grid, block, st); break;
            case 8: ok = quotient_slice<F, 8>(q, widths[i], ns > 1, cs, wires, zs, uniforms, qv, grid, block, st); break;
            case 16: ok = quotient_slice<F, 16>(q, widths[i], ns > 1, cs, wires, zs, uniforms, qv, grid, block, st); break;
            default: ok = false;
        }
        if (!ok) return false;
    }
    return true;
}
template bool quotient_values<GlF>(const QuotientParams<GlF>&, const u64*, const u64*, const u64*, const u64*, u64*, hipStream_t);
template bool quotient_values<BbF>(const QuotientParams<BbF>&, const u32*, const u32*, const u32*, const u32*, u32*, hipStream_t);
// max_quotient_degree_factor 2, 4, 8, 16 (a power of two up to 2^rate_bits, plonk/prover.rs:736-740);
// BabyBear: (31 - degree_bits) * c >= 100 (circuit_builder.rs:1190-1192) needs c >= 4; Goldilocks c >= 2
bool quotient_shape_supported(u32 field, u32 chunk, u32 num_challenges) {
    u32 widths[MAX_CHALLENGES];
    if (num_challenges == 0 || num_challenges > MAX_CHALLENGES) return false;
    if (chunk != 2 && chunk != 4 && chunk != 8 && chunk != 16) return false;
    return challenge_slices(field, quotient_gl_wmax(chunk), num_challenges, widths, quotient_bb_wmax(chunk)) != 0;
}

template <class F>
void l0_table(u32 log_n, u32 rate_bits, const PowTab<F>& w_N, const typename F::T* zh, typename F::T* l0, hipStream_t st) {
    const size_t N = (size_t)1 << (log_n + rate_bits);
    hipLaunchKernelGGL(k_l0_table<F>, dim3(nblk(N, 256)), dim3(256), 0, st, log_n, rate_bits, w_N, zh, l0);
}

template <class F>
void quotient_combine(u32 log_n, u32 rate_bits, u32 num_challenges, const typename F::T* a, const typename F::T* mat,
                      const CosetPow<F>& inv_shift, typename F::T* out, hipStream_t st) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_quotient_combine<F>, dim3(nblk(n, 256), num_challenges), dim3(256), 0, st, log_n, rate_bits, a, mat,
                       inv_shift, out);
}

template <class F>
void ext_powtabs(const ExtPowJobs<F>& jobs, u32 njobs, hipStream_t st) {
    u32 most = 0;
    for (u32 j = 0; j < njobs; j++) most = std::max(most, jobs.j[j].nlo + jobs.j[j].nhi);
    if (most) hipLaunchKernelGGL(k_ext_powtabs<F>, dim3(nblk(most, 256), njobs), dim3(256), 0, st, jobs);
}

template <class F>
void ext_pow_tables(const ExtPowTables<F>& z, u32 count, size_t n, hipStream_t st) {
    if (count && n) hipLaunchKernelGGL(k_ext_pow_tables<F>, dim3(nblk(n, 256), count), dim3(256), 0, st, z, n);
}

template <class F>
void eval_columns(const EvalJobs<F>& jobs, u32 njobs, size_t n, typename F::E* partial_tmp, typename F::E* out, hipStream_t st) {
    u32 total = 0;
    for (u32 j = 0; j < njobs; j++) total += jobs.j[j].ncols;
    if (!total) return;
    const u32 nch = nblk(n, 4096);
    hipLaunchKernelGGL(k_eval_partial<F>, dim3(nch, total), dim3(256), 0, st, jobs, n, partial_tmp);
    hipLaunchKernelGGL(k_eval_final<F>, dim3(total), dim3(256), 0, st, partial_tmp, nch, out);
}

template <class F>
void reduce_polys(const PolyGroups<F>& g, size_t n, const typename F::E* apow, typename F::E* comp, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_polys<F>, dim3(nblk(n, 256)), dim3(256), 0, st, g, n, apow, comp);
}

template <class F>
void divide_by_linear_accumulate(const typename F::E* comp, size_t n, const ExtPowTab<F>& z, const ExtPowTab<F>& zinv,
                                 typename F::E shift, int first, typename F::E* sloc_tmp, typename F::E* totals_tmp,
                                 typename F::E* final_poly, hipStream_t st) {
    const u32 nb = nblk(n, 1024);
    hipLaunchKernelGGL(k_divide_local<F>, dim3(nb), dim3(256), 0, st, comp, n, z, sloc_tmp, totals_tmp);
    hipLaunchKernelGGL(k_divide_totals<F>, dim3(1), dim3(1024), 0, st, totals_tmp, nb);
    hipLaunchKernelGGL(k_divide_apply<F>, dim3(nblk(n, 256)), dim3(256), 0, st, sloc_tmp, totals_tmp, n, zinv, shift, first,
                       final_poly);
}

template <class F>
void ext_split(const typename F::E* src, size_t n, typename F::T* dst, hipStream_t st) {
    hipLaunchKernelGGL(k_ext_split<F>, dim3(nblk(n, 256)), dim3(256), 0, st, src, n, dst);
}

template <class F>
void fri_leaves(const typename F::T* vals, size_t len, u32 arity_bits, u64 num_leaves, typename F::T* out, hipStream_t st) {
    if constexpr (F::TAG == 0)   // small trees: one state per 16-lane row (poseidon_gl_coop.hpp)
        if (gl_fri_leaves_coop(vals, len, arity_bits, num_leaves, out, st)) return;
    if constexpr (F::TAG == 1)
        if (bb_fri_leaves_coop(vals, len, arity_bits, num_leaves, out, st)) return;
    hipLaunchKernelGGL(k_fri_leaves<F>, dim3(nblk(num_leaves, 256)), dim3(256), 0, st, vals, len, arity_bits, num_leaves, out);
}

template <class F>
void fri_fold(const typename F::T* in, size_t in_len, u32 arity_bits, typename F::E beta, typename F::T* out, hipStream_t st) {
    hipLaunchKernelGGL(k_fri_fold<F>, dim3(nblk(in_len >> arity_bits, 256)), dim3(256), 0, st, in, in_len, arity_bits, beta, out);
}

template <class F>
void pow_grind(const PowState<F>& s, u64 start, u64 count, u32 min_lz, u64* result, hipStream_t st) {
    hipLaunchKernelGGL(k_pow_grind<F>, dim3(nblk(count, 256)), dim3(256), 0, st, s, start, count, min_lz, result);
}

template <class F>
void query_gather(const QueryJobs<F>& jobs, u32 njobs, const u64* idx_host, const u64* idx_dev, u32 nidx, typename F::T* out, hipStream_t st) {
    size_t most = 0;
    for (u32 j = 0; j < njobs; j++) most = std::max(most, (size_t)nidx * (jobs.j[j].width + (size_t)jobs.j[j].layers * F::H));
    if (!most || !njobs) return;
    QueryIdx inl{};
    if (nidx <= QUERY_IDX_INLINE) {
        std::copy(idx_host, idx_host + nidx, inl.v);
        hipLaunchKernelGGL((k_query_gather<F, true>), dim3(nblk(most, 256), njobs), dim3(256), 0, st, jobs, inl, idx_dev, nidx, out);
    } else {
        hipLaunchKernelGGL((k_query_gather<F, false>), dim3(nblk(most, 256), njobs), dim3(256), 0, st, jobs, inl, idx_dev, nidx, out);
    }
}

#define GB_INSTANTIATE(F)                                                                                                           \
    template void zs_partial_products<F>(const ZsParams<F>&, const F::T*, const F::T*, const F::T*, F::T*, F::T*, F::T*, u32*, F::T*,  \
                                         hipStream_t);                                                                              \
    template void quotient_combine<F>(u32, u32, u32, const F::T*, const F::T*, const CosetPow<F>&, F::T*, hipStream_t);             \
    template void l0_table<F>(u32, u32, const PowTab<F>&, const F::T*, F::T*, hipStream_t);                                         \
    template void ext_powtabs<F>(const ExtPowJobs<F>&, u32, hipStream_t);                                                           \
    template void ext_pow_tables<F>(const ExtPowTables<F>&, u32, size_t, hipStream_t);                                              \
    template void eval_columns<F>(const EvalJobs<F>&, u32, size_t, F::E*, F::E*, hipStream_t);                                      \
    template void reduce_polys<F>(const PolyGroups<F>&, size_t, const F::E*, F::E*, hipStream_t);                                   \
    template void divide_by_linear_accumulate<F>(const F::E*, size_t, const ExtPowTab<F>&, const ExtPowTab<F>&, F::E, int, F::E*,   \
                                                 F::E*, F::E*, hipStream_t);                                                        \
    template void ext_split<F>(const F::E*, size_t, F::T*, hipStream_t);                                                            \
    template void fri_leaves<F>(const F::T*, size_t, u32, u64, F::T*, hipStream_t);                                                 \
    template void fri_fold<F>(const F::T*, size_t, u32, F::E, F::T*, hipStream_t);                                                  \
    template void pow_grind<F>(const PowState<F>&, u64, u64, u32, u64*, hipStream_t);                                               \
    template void query_gather<F>(const QueryJobs<F>&, u32, const u64*, const u64*, u32, F::T*, hipStream_t);
GB_INSTANTIATE(GlF)
GB_INSTANTIATE(BbF)
#undef GB_INSTANTIATE

}  // namespace gbk
