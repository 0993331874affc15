// Prover kernels for gfx950 (Goldilocks, D = 2): everything of prove() between the commitments.
//
//   k_zs_*          wires_permutation_partial_products_and_zs        plonk/prover.rs:480-546
//   k_quotient      compute_quotient_polys + eval_vanishing_poly_base_batch for the gate set
//                   {Noop, Constant, PublicInput}                    plonk/prover.rs:712-926, vanishing_poly.rs:177-346
//   k_quotient_combine  the size-N coset_ifft's last radix-2^r step + chunking   prover.rs:921-925, :361-374
//   k_eval_*        OpeningSet::new (Horner at zeta, g*zeta)         plonk/proof.rs:346-387
//   k_reduce_polys / k_divide_* / k_final_poly   prove_openings      fri/oracle.rs:187-231
//   k_fri_*         fri_committed_trees fold + leaf hashing          fri/prover.rs:83-133
//   k_pow_grind     fri_proof_of_work (minimum nonce)                fri/prover.rs:136-188
//   k_gather_*      fri_prover_query_rounds                          fri/prover.rs:190-255
// All data is column-major; LDE matrices are in leaf order (see kernels_ntt.hip).
#include "kernels.hpp"
#include "poseidon_gl.hpp"

namespace gbk {

using gl::ext2;

__device__ __forceinline__ u32 brev32(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

__device__ __forceinline__ u64 pow_split(const PowTab& t, u64 e) {
    u64 lo = t.lo[e & ((1u << t.lo_bits) - 1)];
    u64 h = e >> t.lo_bits;
    return h ? gl::mul(lo, t.hi[h]) : lo;
}
__device__ __forceinline__ ext2 pow_split(const ExtPowTab& t, u64 e) {
    const ext2* lo = reinterpret_cast<const ext2*>(t.lo);
    const ext2* hi = reinterpret_cast<const ext2*>(t.hi);
    ext2 l = lo[e & ((1u << t.lo_bits) - 1)];
    u64 h = e >> t.lo_bits;
    return h ? gl::mul(l, hi[h]) : l;
}

// ------------------------------------------------------------------ Z and partial products

// grid (ceil(n/256), c). q[ch][m][row] = prod_{j in chunk m} (w_j + beta k_j x + gamma) / (w_j + beta sigma_j + gamma)
__global__ __launch_bounds__(256) void k_zs_quotients(ZsParams p, const u64* __restrict__ witness, const u64* __restrict__ sigma,
                                                      const u64* __restrict__ k_is, const u64* __restrict__ betas,
                                                      const u64* __restrict__ gammas, u64* __restrict__ q, u32* __restrict__ err) {
    const u32 row = blockIdx.x * 256 + threadIdx.x;
    const u32 ch = blockIdx.y;
    const size_t n = (size_t)1 << p.log_n;
    if (row >= n) return;
    const u64 beta = betas[ch], gamma = gammas[ch];
    const u64 bx = gl::mul(beta, pow_split(p.w_n, row));
    u64 N[MAX_CHUNKS], Dn[MAX_CHUNKS];
    for (u32 m = 0; m < p.nchunks; m++) {
        u64 np = 1, dp = 1;
        const u32 j1 = min((m + 1) * p.chunk, p.num_routed);
        for (u32 j = m * p.chunk; j < j1; j++) {
            u64 w = witness[(size_t)j * n + row];
            u64 num = gl::add(gl::add(w, gl::mul(bx, k_is[j])), gamma);
            u64 den = gl::add(gl::add(w, gl::mul(beta, sigma[(size_t)j * n + row])), gamma);
            np = gl::mul(np, num);
            dp = gl::mul(dp, den);
        }
        N[m] = np;
        Dn[m] = dp;
    }
    // Montgomery batch inversion of the chunk denominators
    u64 pref[MAX_CHUNKS];
    u64 acc = 1;
    for (u32 m = 0; m < p.nchunks; m++) {
        pref[m] = acc;
        acc = gl::mul(acc, Dn[m]);
    }
    if (acc == 0) {  // some denominator is zero: ProverError::InvZeroPermArg (prover.rs:512-514)
        atomicOr(err, 1u);
        return;
    }
    u64 inv_run = gl::inv(acc);
    for (u32 m = p.nchunks; m-- > 0;) {
        u64 dinv = gl::mul(inv_run, pref[m]);
        inv_run = gl::mul(inv_run, Dn[m]);
        q[((size_t)ch * p.nchunks + m) * n + row] = gl::mul(N[m], dinv);
    }
}

// exclusive prefix product over rows of R(row) = prod_m q[ch][m][row], blocks of 1024 rows.
// grid (ceil(n/1024), c): zloc[ch][row] = prod of R over earlier rows of the same block; totals[ch][block]
__global__ __launch_bounds__(256) void k_zs_scan_local(ZsParams p, const u64* __restrict__ q, u64* __restrict__ zloc,
                                                       u64* __restrict__ totals) {
    __shared__ u64 sh[256];
    const size_t n = (size_t)1 << p.log_n;
    const u32 ch = blockIdx.y;
    const size_t row0 = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    u64 r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u64 v = 1;
        if (row0 + k < n)
            for (u32 m = 0; m < p.nchunks; m++) v = gl::mul(v, q[((size_t)ch * p.nchunks + m) * n + row0 + k]);
        r[k] = v;
    }
    u64 mine = gl::mul(gl::mul(r[0], r[1]), gl::mul(r[2], r[3]));
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
        u64 v = sh[threadIdx.x];
        u64 o = threadIdx.x >= off ? sh[threadIdx.x - off] : 1;
        __syncthreads();
        sh[threadIdx.x] = gl::mul(v, o);
        __syncthreads();
    }
    u64 excl = threadIdx.x ? sh[threadIdx.x - 1] : 1;
    if (threadIdx.x == 255) totals[(size_t)ch * gridDim.x + blockIdx.x] = sh[255];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (row0 + k < n) zloc[(size_t)ch * n + row0 + k] = excl;
        excl = gl::mul(excl, r[k]);
    }
}

// grid (c), 1024 threads: totals[ch][b] <- exclusive prefix product (nblocks <= 1024)
__global__ __launch_bounds__(1024) void k_zs_scan_totals(u64* __restrict__ totals, u32 nblocks) {
    __shared__ u64 sh[1024];
    u64* t = totals + (size_t)blockIdx.x * nblocks;
    u64 mine = threadIdx.x < nblocks ? t[threadIdx.x] : 1;
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        u64 v = sh[threadIdx.x];
        u64 o = threadIdx.x >= off ? sh[threadIdx.x - off] : 1;
        __syncthreads();
        sh[threadIdx.x] = gl::mul(v, o);
        __syncthreads();
    }
    if (threadIdx.x < nblocks) t[threadIdx.x] = threadIdx.x ? sh[threadIdx.x - 1] : 1;
}

// grid (ceil(n/256), c): Z(row) = carry * zloc; partial products p_m = Z * q_0..q_m (m < num_prods)
// output columns: [Z_0..Z_{c-1}, pp_{0,*}, pp_{1,*}, ...] (prover.rs:311-317)
__global__ __launch_bounds__(256) void k_zs_finalize(ZsParams p, const u64* __restrict__ q, const u64* __restrict__ zloc,
                                                     const u64* __restrict__ totals, u32 nblocks, u64* __restrict__ out) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 ch = blockIdx.y;
    if (row >= n) return;
    u64 z = gl::mul(totals[(size_t)ch * nblocks + (row >> 10)], zloc[(size_t)ch * n + row]);
    out[(size_t)ch * n + row] = z;
    const u32 num_prods = p.nchunks - 1;
    u64 acc = z;
    for (u32 m = 0; m < num_prods; m++) {
        acc = gl::mul(acc, q[((size_t)ch * p.nchunks + m) * n + row]);
        out[((size_t)p.num_challenges + (size_t)ch * num_prods + m) * n + row] = acc;
    }
}

// ------------------------------------------------------------------ quotient

// One thread per LDE point, addressed by its leaf index j.  Writes the (unshifted) quotient value to
// qv[(k * R + coset) * n + il], il = natural index of the point inside its coset block.
// C = num_challenges and CH = chunk size (quotient_degree_factor) are compile-time so that the per-challenge
// accumulators stay in registers and a chunk's 2*CH loads are issued together.
template <u32 C, u32 CH>
__global__ __launch_bounds__(256) void k_quotient(QuotientParams p, const u64* __restrict__ cs, const u64* __restrict__ wires,
                                                  const u64* __restrict__ zs, const u64* __restrict__ uni,
                                                  u64* __restrict__ qv) {
    const u32 lgn = p.log_n, r = p.rate_bits;
    const size_t n = (size_t)1 << lgn, N = n << r;
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const u32 cidx = (u32)(j >> lgn), jl = (u32)(j & (n - 1));
    const u32 il = brev32(jl, lgn);
    const u32 imod = brev32(cidx, r);
    const u64 i = ((u64)il << r) | imod;
    const u64 x = gl::mul(gl::GENERATOR, pow_split(p.w_N, i));  // shifted_x = 7 * w_N^i
    const size_t jn = ((size_t)cidx << lgn) | brev32((il + 1) & (u32)(n - 1), lgn);  // leaf of i + 2^r

    // uniform tables: [betas c][gammas c][bk c*routed][apow c*nterms][zh R][zh_inv R][pi_hash 4]
    const u32 nr = p.num_routed, nterms = p.nterms, R = 1u << r;
    const u64* betas = uni;
    const u64* gammas = uni + C;
    const u64* bk = gammas + C;
    const u64* apow = bk + (size_t)C * nr;
    const u64* zh = apow + (size_t)C * nterms;
    const u64* zh_inv = zh + R;
    const u64* pi_hash = zh_inv + R;

    u64 acc[C];
#pragma unroll
    for (u32 k = 0; k < C; k++) acc[k] = 0;
    u32 t = 0;
    // L_0(x) (Z(x) - 1): eval_l_0 (zero_poly_coset.rs:58-61)
    const u64 l0 = gl::mul(zh[imod], gl::inv(gl::mul((u64)(n % gl::P), gl::sub(x, 1))));
    u64 zk[C];
#pragma unroll
    for (u32 k = 0; k < C; k++) zk[k] = zs[(size_t)k * N + j];
#pragma unroll
    for (u32 k = 0; k < C; k++, t++) {
        u64 term = gl::mul(l0, gl::sub(zk[k], 1));
#pragma unroll
        for (u32 k2 = 0; k2 < C; k2++) acc[k2] = gl::add(acc[k2], gl::mul(term, apow[k2 * nterms + t]));
    }
    // partial-product checks (util/partial_products.rs:53-77); term index = C + k * nchunks + m
    const u32 num_prods = p.nchunks - 1;
    u64 prev[C];
#pragma unroll
    for (u32 k = 0; k < C; k++) prev[k] = zk[k];
    for (u32 m = 0; m < p.nchunks; m++) {
        u64 wv[CH], sg[CH];
        const u32 w0 = m * CH;
#pragma unroll
        for (u32 q = 0; q < CH; q++) {
            const u32 w = w0 + q < nr ? w0 + q : nr - 1;  // clamp: the tail chunk re-reads the last wire, masked below
            wv[q] = wires[(size_t)w * N + j];
            sg[q] = cs[(size_t)(p.num_constants + w) * N + j];
        }
        u64 next[C];
#pragma unroll
        for (u32 k = 0; k < C; k++)
            next[k] = m == num_prods ? zs[(size_t)k * N + jn] : zs[((size_t)C + (size_t)k * num_prods + m) * N + j];
        u64 np[C], dp[C];
#pragma unroll
        for (u32 k = 0; k < C; k++) np[k] = dp[k] = 1;
#pragma unroll
        for (u32 q = 0; q < CH; q++) {
            if (w0 + q < nr) {
#pragma unroll
                for (u32 k = 0; k < C; k++) {
                    u64 num = gl::add(gl::add(wv[q], gl::mul(bk[k * nr + w0 + q], x)), gammas[k]);
                    u64 den = gl::add(gl::add(wv[q], gl::mul(betas[k], sg[q])), gammas[k]);
                    np[k] = gl::mul(np[k], num);
                    dp[k] = gl::mul(dp[k], den);
                }
            }
        }
#pragma unroll
        for (u32 k = 0; k < C; k++) {
            const u64 term = gl::sub(gl::mul(prev[k], np[k]), gl::mul(next[k], dp[k]));
            const u32 tt = C + k * p.nchunks + m;
#pragma unroll
            for (u32 k2 = 0; k2 < C; k2++) acc[k2] = gl::add(acc[k2], gl::mul(term, apow[k2 * nterms + tt]));
            prev[k] = next[k];
        }
    }
    t = C + C * p.nchunks;
    // gate constraints: filter * unfiltered, summed per constraint index (vanishing_poly.rs:741-774,
    // gates/gate.rs:188-215,391-404).  One selector group {0,1,2}; no UNUSED factor (single selector).
    {
        const u64 s = cs[j];  // constants[0] = selector
        u64 f[3];
#pragma unroll
        for (u32 g = 0; g < 3; g++) {
            u64 v = 1;
#pragma unroll
            for (u32 ii = 0; ii < 3; ii++)
                if (ii != g) v = gl::mul(v, gl::sub((u64)ii, s));
            f[g] = v;
        }
        const u64 f_pi = p.gate_pi == 0 ? f[0] : (p.gate_pi == 1 ? f[1] : f[2]);
        const u64 f_c = p.gate_constant == 0 ? f[0] : (p.gate_constant == 1 ? f[1] : f[2]);
#pragma unroll
        for (u32 cj = 0; cj < 4; cj++, t++) {
            const u64 wv = wires[(size_t)cj * N + j];
            u64 term = gl::mul(f_pi, gl::sub(wv, pi_hash[cj]));
            if (cj < p.num_gate_consts) {
                const u64 kc = cs[(size_t)(p.num_selectors + cj) * N + j];
                term = gl::add(term, gl::mul(f_c, gl::sub(kc, wv)));
            }
#pragma unroll
            for (u32 k2 = 0; k2 < C; k2++) acc[k2] = gl::add(acc[k2], gl::mul(term, apow[k2 * nterms + t]));
        }
    }
#pragma unroll
    for (u32 k = 0; k < C; k++) qv[(((size_t)k << r) + cidx) * n + il] = gl::mul(acc[k], zh_inv[imod]);
}

// After the per-block natural->natural inverse NTTs: a_c[t] are the coefficients of R_c(s_c X).
// chunk_m[t] = 7^(-n m) / R * sum_c zeta_c^(-m) * s_c^(-t) * a_c[t]     (mat[m][c] holds the constant part)
__global__ __launch_bounds__(256) void k_quotient_combine(u32 log_n, u32 rate_bits, const u64* __restrict__ a,
                                                          const u64* __restrict__ mat, CosetPow inv_shift, u64* __restrict__ out) {
    const size_t n = (size_t)1 << log_n;
    const u32 R = 1u << rate_bits;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 k = blockIdx.y;
    if (t >= n) return;
    u64 v[MAX_RATE];
    for (u32 c = 0; c < R; c++) {
        u64 s = inv_shift.lo[(size_t)c * inv_shift.nlo + (t & (inv_shift.nlo - 1))];
        size_t h = t / inv_shift.nlo;
        if (h) s = gl::mul(s, inv_shift.hi[(size_t)c * inv_shift.nhi + h]);
        v[c] = gl::mul(a[((size_t)k * R + c) * n + t], s);
    }
    for (u32 m = 0; m < R; m++) {
        u64 acc = 0;
        for (u32 c = 0; c < R; c++) acc = gl::add(acc, gl::mul(v[c], mat[m * R + c]));
        out[((size_t)k * R + m) * n + t] = acc;
    }
}

// ------------------------------------------------------------------ openings: sum_t c_t z^t

// table[t] = z^t, t < n
__global__ __launch_bounds__(256) void k_ext_pow_table(ExtPowTab z, size_t n, ext2* __restrict__ table) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) table[t] = pow_split(z, t);
}

__device__ __forceinline__ ext2 block_reduce_add(ext2 v, ext2* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (u32 off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = gl::add(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    ext2 r = sh[0];
    __syncthreads();
    return r;
}

// grid (nchunks = ceil(n / 4096), ncols): partial[col][chunk] = sum over the chunk of c_t * z^t
__global__ __launch_bounds__(256) void k_eval_partial(const u64* __restrict__ coeffs, size_t n, const ext2* __restrict__ ztab,
                                                      ext2* __restrict__ partial) {
    __shared__ ext2 sh[256];
    const u64* c = coeffs + (size_t)blockIdx.y * n;
    const size_t base = (size_t)blockIdx.x * 4096;
    ext2 acc = gl::e2(0);
    for (u32 k = 0; k < 16; k++) {
        size_t t = base + k * 256 + threadIdx.x;
        if (t < n) acc = gl::add(acc, gl::scale(ztab[t], c[t]));
    }
    ext2 r = block_reduce_add(acc, sh);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = r;
}
// grid (ncols): out[col] = sum of partial[col][*]
__global__ __launch_bounds__(256) void k_eval_final(const ext2* __restrict__ partial, u32 nchunks, ext2* __restrict__ out) {
    __shared__ ext2 sh[256];
    ext2 acc = gl::e2(0);
    for (u32 k = threadIdx.x; k < nchunks; k += 256) acc = gl::add(acc, partial[(size_t)blockIdx.x * nchunks + k]);
    ext2 r = block_reduce_add(acc, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// ------------------------------------------------------------------ prove_openings

// comp[t] = sum_j alpha^j * poly_j[t]  (reduce_polys_base, util/reducing.rs:89-103) over up to 4 column groups
__global__ __launch_bounds__(256) void k_reduce_polys(PolyGroups g, size_t n, const ext2* __restrict__ apow, ext2* __restrict__ comp) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    ext2 acc = gl::e2(0);
    u32 jj = 0;
    for (u32 o = 0; o < g.ngroups; o++) {
        const u64* base = g.ptr[o];
        for (u32 j = 0; j < g.ncols[o]; j++, jj++) acc = gl::add(acc, gl::scale(apow[jj], base[(size_t)j * n + t]));
    }
    comp[t] = acc;
}

// divide_by_linear (polynomial/division.rs:75-88): q[t] = sum_{u > t} c_u z^(u-t-1) = z^-(t+1) * S_{t+1},
// S_t = sum_{u >= t} c_u z^u.  Step 1: w_u = c_u z^u and block-local suffix sums (blocks of 1024).
__global__ __launch_bounds__(256) void k_divide_local(const ext2* __restrict__ comp, size_t n, ExtPowTab z, ext2* __restrict__ sloc,
                                                      ext2* __restrict__ totals) {
    __shared__ ext2 sh[256];
    const size_t u0 = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
    ext2 w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = u0 + k < n ? gl::mul(comp[u0 + k], pow_split(z, u0 + k)) : gl::e2(0);
    ext2 mine = gl::add(gl::add(w[0], w[1]), gl::add(w[2], w[3]));
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {  // inclusive SUFFIX scan
        ext2 v = sh[threadIdx.x];
        ext2 o = threadIdx.x + off < 256 ? sh[threadIdx.x + off] : gl::e2(0);
        __syncthreads();
        sh[threadIdx.x] = gl::add(v, o);
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = sh[0];
    ext2 run = threadIdx.x + 1 < 256 ? sh[threadIdx.x + 1] : gl::e2(0);  // sum of later threads in this block
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        run = gl::add(run, w[k]);
        if (u0 + k < n) sloc[u0 + k] = run;  // local S_t (this block only)
    }
}
// single block, 1024 threads: totals[b] <- sum of totals of LATER blocks (exclusive suffix)
__global__ __launch_bounds__(1024) void k_divide_totals(ext2* __restrict__ totals, u32 nblocks) {
    __shared__ ext2 sh[1024];
    sh[threadIdx.x] = threadIdx.x < nblocks ? totals[threadIdx.x] : gl::e2(0);
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
        ext2 v = sh[threadIdx.x];
        ext2 o = threadIdx.x + off < 1024 ? sh[threadIdx.x + off] : gl::e2(0);
        __syncthreads();
        sh[threadIdx.x] = gl::add(v, o);
        __syncthreads();
    }
    if (threadIdx.x < nblocks) totals[threadIdx.x] = threadIdx.x + 1 < 1024 ? sh[threadIdx.x + 1] : gl::e2(0);
}
// final[t] = final[t] * shift + q[t], q[t] = zinv^(t+1) * S_{t+1}, q[n-1] = 0   (fri/oracle.rs:218-223)
__global__ __launch_bounds__(256) void k_divide_apply(const ext2* __restrict__ sloc, const ext2* __restrict__ totals, size_t n,
                                                      ExtPowTab zinv, ext2 shift, int first, ext2* __restrict__ final_poly) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    ext2 q = gl::e2(0);
    if (t + 1 < n) {
        size_t u = t + 1;
        ext2 S = gl::add(sloc[u], totals[u >> 10]);
        q = gl::mul(S, pow_split(zinv, u));
    }
    final_poly[t] = first ? q : gl::add(gl::mul(final_poly[t], shift), q);
}
// split an ext2 array into two base columns [2][n] (for the coordinate-wise NTT)
__global__ __launch_bounds__(256) void k_ext_split(const ext2* __restrict__ src, size_t n, u64* __restrict__ dst) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    ext2 v = src[t];
    dst[t] = v.c0;
    dst[n + t] = v.c1;
}

// ------------------------------------------------------------------ FRI

// leaf m = flatten(values[arity*m .. arity*(m+1))) in bit-reversed (= leaf) order (fri/prover.rs:101-107)
__global__ __launch_bounds__(256) void k_fri_leaves(const u64* __restrict__ v0, const u64* __restrict__ v1, u32 arity_bits,
                                                    u64 num_leaves, u64* __restrict__ out) {
    u64 m = (u64)blockIdx.x * 256 + threadIdx.x;
    if (m >= num_leaves) return;
    const u32 arity = 1u << arity_bits;
    const u64* a = v0 + (m << arity_bits);
    const u64* b = v1 + (m << arity_bits);
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    if (2 * arity <= 4) {  // hash_or_noop
        for (u32 k = 0; k < arity; k++) {
            s[2 * k] = a[k];
            s[2 * k + 1] = b[k];
        }
    } else {
        for (u32 k0 = 0; k0 < arity; k0 += 4) {  // 8 base elements per absorption
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k0 + k < arity) {
                    s[2 * k] = a[k0 + k];
                    s[2 * k + 1] = b[k0 + k];
                }
            poseidon_gl::permute_lazy(s);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = poseidon_gl::to_canonical(s[i]);
    }
    u64* o = out + 4 * m;
    o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
}

// coeffs' [m] = sum_t coeffs[arity*m + t] beta^t  (reduce_with_powers, fri/prover.rs:112-121); in/out as [2][len] columns
__global__ __launch_bounds__(256) void k_fri_fold(const u64* __restrict__ in, size_t in_len, u32 arity_bits, ext2 beta,
                                                  u64* __restrict__ out) {
    const size_t out_len = in_len >> arity_bits;
    size_t m = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= out_len) return;
    const u32 arity = 1u << arity_bits;
    ext2 acc = gl::e2(0);
    for (u32 t = arity; t-- > 0;) {
        size_t idx = (m << arity_bits) + t;
        acc = gl::add(gl::mul(acc, beta), gl::e2(in[idx], in[in_len + idx]));
    }
    out[m] = acc.c0;
    out[out_len + m] = acc.c1;
}

// proof of work: candidates start .. start+count; result = min satisfying candidate (fri/prover.rs:169-180)
__global__ __launch_bounds__(256) void k_pow_grind(PowState st, u64 start, u64 count, u32 min_leading_zeros, u64* __restrict__ result) {
    u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    if (g >= count) return;
    u64 cand = start + g;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = st.s[i];
    // runtime position, static register indexing
#pragma unroll
    for (int i = 0; i < 12; i++)
        if ((u32)i == st.pos) s[i] = cand;
    poseidon_gl::permute_lazy(s);
    u64 resp = poseidon_gl::to_canonical(s[7]);
    u32 lz = resp ? (u32)__clzll((long long)resp) : 64;
    if (lz >= min_leading_zeros) atomicMin(result, cand);
}

// ------------------------------------------------------------------ query gathers

// rows[q][w] = cols[w * stride + idx[q]]
__global__ void k_gather_rows(const u64* __restrict__ cols, size_t stride, u32 width, const u64* __restrict__ idx, u32 nidx,
                              u64* __restrict__ rows) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nidx * width) return;
    u32 q = g / width, w = g % width;
    rows[g] = cols[(size_t)w * stride + idx[q]];
}
// FRI layer leaf: out[q][2k + comp] = v_comp[arity * idx[q] + k]
__global__ void k_gather_fri_leaves(const u64* __restrict__ v0, const u64* __restrict__ v1, u32 arity_bits,
                                    const u64* __restrict__ idx, u32 nidx, u64* __restrict__ out) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 width = 2u << arity_bits;
    if (g >= nidx * width) return;
    u32 q = g / width, e = g % width;
    const u64* v = (e & 1) ? v1 : v0;
    out[g] = v[(idx[q] << arity_bits) + (e >> 1)];
}
// sib[q][i][0..4) = level_i[(idx[q] >> i) ^ 1]
__global__ void k_gather_siblings_multi(const u64* __restrict__ levels, u32 log_leaves, u32 cap_height,
                                        const u64* __restrict__ idx, u32 nidx, u64* __restrict__ out) {
    const u32 layers = log_leaves - cap_height;
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nidx * layers * 4) return;
    u32 e = g & 3, i = (g >> 2) % layers, q = (g >> 2) / layers;
    const u64 N = (u64)1 << log_leaves;
    const u64 off = 2 * N - ((2 * N) >> i);
    out[g] = levels[4 * (off + ((idx[q] >> i) ^ 1)) + e];
}

// ------------------------------------------------------------------ launchers

static inline u32 nblk(size_t n, u32 bs) { return (u32)((n + bs - 1) / bs); }

void gl_zs_partial_products(const ZsParams& p, const u64* witness, const u64* sigma, const u64* k_is, const u64* betas,
                            const u64* gammas, u64* q_tmp, u64* zloc_tmp, u64* totals_tmp, u32* err, u64* out, hipStream_t st) {
    const size_t n = (size_t)1 << p.log_n;
    const u32 nb1024 = nblk(n, 1024);
    hipLaunchKernelGGL(k_zs_quotients, dim3(nblk(n, 256), p.num_challenges), dim3(256), 0, st, p, witness, sigma, k_is, betas,
                       gammas, q_tmp, err);
    hipLaunchKernelGGL(k_zs_scan_local, dim3(nb1024, p.num_challenges), dim3(256), 0, st, p, q_tmp, zloc_tmp, totals_tmp);
    hipLaunchKernelGGL(k_zs_scan_totals, dim3(p.num_challenges), dim3(1024), 0, st, totals_tmp, nb1024);
    hipLaunchKernelGGL(k_zs_finalize, dim3(nblk(n, 256), p.num_challenges), dim3(256), 0, st, p, q_tmp, zloc_tmp, totals_tmp,
                       nb1024, out);
}

void gl_quotient_values(const QuotientParams& p, const u64* cs, const u64* wires, const u64* zs, const u64* uniforms, u64* qv,
                        hipStream_t st) {
    const size_t N = (size_t)1 << (p.log_n + p.rate_bits);
    const dim3 grid(nblk(N, 256)), block(256);
#define GB_Q(CC, HH) hipLaunchKernelGGL((k_quotient<CC, HH>), grid, block, 0, st, p, cs, wires, zs, uniforms, qv)
    if (p.chunk == 8) {
        switch (p.num_challenges) {
            case 1: GB_Q(1, 8); return;
            case 2: GB_Q(2, 8); return;
            case 3: GB_Q(3, 8); return;
            case 4: GB_Q(4, 8); return;
            default: break;
        }
    }
    if (p.chunk == 16 && p.num_challenges <= 2) {
        if (p.num_challenges == 1) GB_Q(1, 16); else GB_Q(2, 16);
        return;
    }
    // (gb_circuit_create rejects other shapes)
#undef GB_Q
}

void gl_quotient_combine(u32 log_n, u32 rate_bits, u32 num_challenges, const u64* a, const u64* mat, const CosetPow& inv_shift,
                         u64* out, hipStream_t st) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_quotient_combine, dim3(nblk(n, 256), num_challenges), dim3(256), 0, st, log_n, rate_bits, a, mat,
                       inv_shift, out);
}

void gl_ext_pow_table(const ExtPowTab& z, size_t n, u64* table, hipStream_t st) {
    hipLaunchKernelGGL(k_ext_pow_table, dim3(nblk(n, 256)), dim3(256), 0, st, z, n, reinterpret_cast<ext2*>(table));
}

void gl_eval_columns(const u64* coeffs, size_t ncols, size_t n, const u64* ztab, u64* partial_tmp, u64* out, hipStream_t st) {
    if (!ncols) return;
    const u32 nch = nblk(n, 4096);
    hipLaunchKernelGGL(k_eval_partial, dim3(nch, (u32)ncols), dim3(256), 0, st, coeffs, n, reinterpret_cast<const ext2*>(ztab),
                       reinterpret_cast<ext2*>(partial_tmp));
    hipLaunchKernelGGL(k_eval_final, dim3((u32)ncols), dim3(256), 0, st, reinterpret_cast<const ext2*>(partial_tmp), nch,
                       reinterpret_cast<ext2*>(out));
}

void gl_reduce_polys(const PolyGroups& g, size_t n, const u64* apow, u64* comp, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_polys, dim3(nblk(n, 256)), dim3(256), 0, st, g, n, reinterpret_cast<const ext2*>(apow),
                       reinterpret_cast<ext2*>(comp));
}

void gl_divide_by_linear_accumulate(const u64* comp, size_t n, const ExtPowTab& z, const ExtPowTab& zinv, const u64 shift[2],
                                    int first, u64* sloc_tmp, u64* totals_tmp, u64* final_poly, hipStream_t st) {
    const u32 nb = nblk(n, 1024);
    hipLaunchKernelGGL(k_divide_local, dim3(nb), dim3(256), 0, st, reinterpret_cast<const ext2*>(comp), n, z,
                       reinterpret_cast<ext2*>(sloc_tmp), reinterpret_cast<ext2*>(totals_tmp));
    hipLaunchKernelGGL(k_divide_totals, dim3(1), dim3(1024), 0, st, reinterpret_cast<ext2*>(totals_tmp), nb);
    hipLaunchKernelGGL(k_divide_apply, dim3(nblk(n, 256)), dim3(256), 0, st, reinterpret_cast<const ext2*>(sloc_tmp),
                       reinterpret_cast<const ext2*>(totals_tmp), n, zinv, gl::e2(shift[0], shift[1]), first,
                       reinterpret_cast<ext2*>(final_poly));
}

void gl_ext_split(const u64* src, size_t n, u64* dst, hipStream_t st) {
    hipLaunchKernelGGL(k_ext_split, dim3(nblk(n, 256)), dim3(256), 0, st, reinterpret_cast<const ext2*>(src), n, dst);
}

void gl_fri_leaves(const u64* v0, const u64* v1, u32 arity_bits, u64 num_leaves, u64* out, hipStream_t st) {
    hipLaunchKernelGGL(k_fri_leaves, dim3(nblk(num_leaves, 256)), dim3(256), 0, st, v0, v1, arity_bits, num_leaves, out);
}

void gl_fri_fold(const u64* in, size_t in_len, u32 arity_bits, const u64 beta[2], u64* out, hipStream_t st) {
    hipLaunchKernelGGL(k_fri_fold, dim3(nblk(in_len >> arity_bits, 256)), dim3(256), 0, st, in, in_len, arity_bits,
                       gl::e2(beta[0], beta[1]), out);
}

void gl_pow_grind(const PowState& s, u64 start, u64 count, u32 min_lz, u64* result, hipStream_t st) {
    hipLaunchKernelGGL(k_pow_grind, dim3(nblk(count, 256)), dim3(256), 0, st, s, start, count, min_lz, result);
}

void gl_gather_rows_multi(const u64* cols, size_t stride, u32 width, const u64* idx, u32 nidx, u64* rows, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_rows, dim3(nblk((size_t)nidx * width, 256)), dim3(256), 0, st, cols, stride, width, idx, nidx, rows);
}
void gl_gather_fri_leaves(const u64* v0, const u64* v1, u32 arity_bits, const u64* idx, u32 nidx, u64* out, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_fri_leaves, dim3(nblk((size_t)nidx * (2u << arity_bits), 256)), dim3(256), 0, st, v0, v1,
                       arity_bits, idx, nidx, out);
}
void gl_gather_siblings_multi(const u64* levels, u32 log_leaves, u32 cap_height, const u64* idx, u32 nidx, u64* out,
                              hipStream_t st) {
    const u32 layers = log_leaves - cap_height;
    if (!layers) return;
    hipLaunchKernelGGL(k_gather_siblings_multi, dim3(nblk((size_t)nidx * layers * 4, 256)), dim3(256), 0, st, levels, log_leaves,
                       cap_height, idx, nidx, out);
}

}  // namespace gbk
