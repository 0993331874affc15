/* TEST ORACLE - CPU restatement of the reference's Goldilocks hot path.
 *
 * Test infrastructure only (see gl.h).  Every function cites the reference file:line it follows
 * (paths relative to /root/reference).  Pinned by: Poseidon-12 KATs, the serialized regression
 * proof (all Merkle paths / PoW / FRI queries) and the circuit_digest KAT - see
 * tests/test_oracle_*.py.  Parallelised with OpenMP the way the reference uses Rayon (over
 * columns, over Merkle subtrees via recursive join) so that it can double as the "port" CPU
 * baseline in bench.py.
 */
#include "gl.h"
#include "poseidon_constants.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define W 12
#define RATE 8
#define HOUT 4

double gbo_last_cs_commit_seconds = 0.0; /* set by the dummy-circuit provers (prover_impl.h) */
/* the cap of the constants/sigmas commitment the last prover call made ITSELF (prover_impl.h), canonical words of the field: lets
 * a test compare the GPU's build()-time cap with the oracle's own before it seeds the oracle's transcript with it */
unsigned char gbo_last_cs_cap[8192];
size_t gbo_last_cs_cap_bytes = 0;
#define N_PARTIAL 22
#define HALF_FULL 4

static const uint64_t RC[GL_POSEIDON_ALL_ROUND_CONSTANTS_LEN] = {GL_POSEIDON_ALL_ROUND_CONSTANTS_LIST};
static const uint64_t MDS_CIRC[12] = {GL_POSEIDON_MDS_CIRC_LIST};
static const uint64_t MDS_DIAG[12] = {GL_POSEIDON_MDS_DIAG_LIST};
static const uint64_t FP_FIRST[12] = {GL_POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT_LIST};
static const uint64_t FP_RC[22] = {GL_POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS_LIST};
static const uint64_t FP_VS[22][11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_VS_LIST};
static const uint64_t FP_WHATS[22][11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_W_HATS_LIST};
static const uint64_t FP_INIT[11][11] = {GL_POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX_LIST};

/* ------------------------------------------------------------------ Poseidon-12 */

/* hash/poseidon_goldilocks.rs:840-846 */
static inline gl_t sbox(gl_t x) {
    gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}

/* hash/poseidon_goldilocks.rs:547-557 (mds_row_shf_field) applied to all rows (:584-595), computed the way the
 * reference's mds_layer does (:497-528): on the 32-bit halves of the state, whose 12-term sums with the < 2^6
 * entries fit 64 bits, recombined as lo + 2^32 hi and reduced once. */
static void mds_layer(gl_t s[W]) {
    uint64_t lo[2 * W], hi[2 * W];
    for (int i = 0; i < W; i++) {
        lo[i] = lo[i + W] = (uint32_t)s[i];
        hi[i] = hi[i + W] = s[i] >> 32;
    }
    for (int r = 0; r < W; r++) {
        uint64_t sl = lo[r] * MDS_DIAG[r], sh = hi[r] * MDS_DIAG[r];
        for (int i = 0; i < W; i++) {
            sl += lo[i + r] * MDS_CIRC[i];
            sh += hi[i + r] * MDS_CIRC[i];
        }
        s[r] = gl_reduce128((unsigned __int128)sl + ((unsigned __int128)sh << 32));
    }
}

/* hash/poseidon_goldilocks.rs:802-809 */
static inline void constant_layer(gl_t s[W], int round) {
    for (int i = 0; i < W; i++) s[i] = gl_add(s[i], gl_canon(RC[i + W * round]));
}

/* hash/poseidon_goldilocks.rs:889-897 */
static void full_rounds(gl_t s[W], int *round) {
    for (int k = 0; k < HALF_FULL; k++) {
        constant_layer(s, *round);
        for (int i = 0; i < W; i++) s[i] = sbox(s[i]);
        mds_layer(s);
        (*round)++;
    }
}

/* hash/poseidon_goldilocks.rs:927-948 (poseidon_naive: the defining form) */
void gbo_gl_poseidon_naive(const gl_t in[W], gl_t out[W]) {
    gl_t s[W];
    memcpy(s, in, sizeof s);
    int round = 0;
    full_rounds(s, &round);
    for (int k = 0; k < N_PARTIAL; k++) {
        constant_layer(s, round);
        s[0] = sbox(s[0]);
        mds_layer(s);
        round++;
    }
    full_rounds(s, &round);
    memcpy(out, s, sizeof s);
}

/* ---- the form the CPU baseline runs.  The reference's scalar path (hash/poseidon_goldilocks.rs:889-922) keeps the state as
 * arbitrary u64 representatives between layers (reduce128 :254-267 does not canonicalise), accumulates the partial rounds' dot
 * products in 128 bits before one reduction (mds_partial_layer_fast :718-744, reduce_u160) and fuses `s[i] + s0 * v[i]`
 * (multiply_accumulate); restated here the same way.  Output-identical to gbo_gl_poseidon_naive, the defining form
 * (the reference asserts the same at :1196-1198; here tests/test_oracle_kats.py). */
static uint64_t RC_C[GL_POSEIDON_ALL_ROUND_CONSTANTS_LEN], FP_FIRST_C[12], FP_RC_C[22], FP_VS_C[22][11], FP_WHATS_C[22][11],
    FP_INIT_C[11][11];
__attribute__((constructor)) static void poseidon_tables_init(void) {
    for (size_t i = 0; i < GL_POSEIDON_ALL_ROUND_CONSTANTS_LEN; i++) RC_C[i] = gl_canon(RC[i]);
    for (int i = 0; i < 12; i++) FP_FIRST_C[i] = gl_canon(FP_FIRST[i]);
    for (int k = 0; k < 22; k++) {
        FP_RC_C[k] = gl_canon(FP_RC[k]);
        for (int i = 0; i < 11; i++) { FP_VS_C[k][i] = gl_canon(FP_VS[k][i]); FP_WHATS_C[k][i] = gl_canon(FP_WHATS[k][i]); }
    }
    for (int r = 0; r < 11; r++) for (int c = 0; c < 11; c++) FP_INIT_C[r][c] = gl_canon(FP_INIT[r][c]);
}
/* any 128-bit value -> some u64 representative (gl_reduce128 without the final canonicalisation) */
static inline uint64_t nc_reduce128(unsigned __int128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    t0 -= -(uint64_t)(lo < hi_hi) & GL_EPS; /* branch-free: the carries are coin flips on random data */
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t t2 = t0 + t1;
    t2 += -(uint64_t)(t2 < t0) & GL_EPS;
    return t2;
}
static inline uint64_t nc_mul(uint64_t a, uint64_t b) { return nc_reduce128((unsigned __int128)a * b); }
/* a: any representative, c: canonical */
static inline uint64_t nc_add_canon(uint64_t a, uint64_t c) {
    uint64_t s = a + c;
    return s + (-(uint64_t)(s < a) & GL_EPS);
}
static inline uint64_t nc_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    uint64_t t = s + (-(uint64_t)(s < a) & GL_EPS);
    return t + (-(uint64_t)(t < s) & GL_EPS);
}
static inline uint64_t nc_sbox(uint64_t x) {
    uint64_t x2 = nc_mul(x, x), x4 = nc_mul(x2, x2), x3 = nc_mul(x, x2);
    return nc_mul(x3, x4);
}
/* sum of up to 12 products held as (sum of low words, sum of high words) */
typedef struct { unsigned __int128 lo, hi; } acc_t;
static inline void acc_mul(acc_t *a, uint64_t x, uint64_t y) {
    unsigned __int128 p = (unsigned __int128)x * y;
    a->lo += (uint64_t)p;
    a->hi += (uint64_t)(p >> 64);
}
static inline uint64_t acc_reduce(const acc_t *a) {
    uint64_t h = nc_reduce128(a->hi);
    return nc_add(nc_reduce128(a->lo), nc_reduce128((unsigned __int128)h << 64));
}
static inline void nc_mds_layer(uint64_t s[W]) {
    uint64_t lo[2 * W], hi[2 * W];
    for (int i = 0; i < W; i++) {
        lo[i] = lo[i + W] = (uint32_t)s[i];
        hi[i] = hi[i + W] = s[i] >> 32;
    }
    for (int r = 0; r < W; r++) {
        uint64_t sl = lo[r] * MDS_DIAG[r], sh = hi[r] * MDS_DIAG[r];
        for (int i = 0; i < W; i++) {
            sl += lo[i + r] * MDS_CIRC[i];
            sh += hi[i + r] * MDS_CIRC[i];
        }
        s[r] = nc_reduce128((unsigned __int128)sl + ((unsigned __int128)sh << 32));
    }
}
static inline void nc_full_rounds(uint64_t s[W], int *round) {
    for (int k = 0; k < HALF_FULL; k++) {
        for (int i = 0; i < W; i++) s[i] = nc_sbox(nc_add_canon(s[i], RC_C[i + W * *round]));
        nc_mds_layer(s);
        (*round)++;
    }
}
void gbo_gl_poseidon(const gl_t in[W], gl_t out[W]) {
    uint64_t s[W];
    memcpy(s, in, sizeof s);
    int round = 0;
    nc_full_rounds(s, &round);
    for (int i = 0; i < W; i++) s[i] = nc_add_canon(s[i], FP_FIRST_C[i]);
    {
        uint64_t r[W];
        r[0] = s[0];
        for (int c = 1; c < W; c++) {
            acc_t a = {0, 0};
            for (int rr = 1; rr < W; rr++) acc_mul(&a, s[rr], FP_INIT_C[rr - 1][c - 1]);
            r[c] = acc_reduce(&a);
        }
        memcpy(s, r, sizeof r);
    }
    for (int k = 0; k < N_PARTIAL; k++) {
        uint64_t s0 = nc_add_canon(nc_sbox(s[0]), FP_RC_C[k]);
        acc_t a = {0, 0};
        acc_mul(&a, s0, MDS_CIRC[0] + MDS_DIAG[0]);
        for (int i = 1; i < W; i++) acc_mul(&a, s[i], FP_WHATS_C[k][i - 1]);
        for (int i = 1; i < W; i++) s[i] = nc_reduce128((unsigned __int128)s0 * FP_VS_C[k][i - 1] + s[i]);
        s[0] = acc_reduce(&a);
    }
    round += N_PARTIAL;
    nc_full_rounds(s, &round);
    for (int i = 0; i < W; i++) out[i] = gl_canon(s[i]);
}

/* gates/poseidon_goldilocks.rs:223-313 PoseidonGate::eval_unfiltered_base_one: the 123 constraints of one row, in the
 * reference's order.  w = the row's wire values: 0..11 inputs, 12..23 outputs, 24 swap, 25..28 deltas, then the s-box
 * inputs of full rounds 1..3, of the 22 partial rounds (fast form) and of full rounds 4..7 (:44-97). */
void gbo_gl_poseidon_gate_constraints(const gl_t *w, gl_t *out) {
    enum { WIRE_SWAP = 24, START_DELTA = 25, START_FULL_0 = 29, START_PARTIAL = START_FULL_0 + W * (HALF_FULL - 1),
           START_FULL_1 = START_PARTIAL + N_PARTIAL };
    int t = 0, round = 0;
    gl_t swap = w[WIRE_SWAP], s[W];
    out[t++] = gl_mul(swap, gl_sub(swap, 1));
    for (int i = 0; i < 4; i++) out[t++] = gl_sub(gl_mul(swap, gl_sub(w[i + 4], w[i])), w[START_DELTA + i]);
    for (int i = 0; i < 4; i++) {
        s[i] = gl_add(w[i], w[START_DELTA + i]);
        s[i + 4] = gl_sub(w[i + 4], w[START_DELTA + i]);
    }
    for (int i = 8; i < W; i++) s[i] = w[i];
    for (int r = 0; r < HALF_FULL; r++) {
        constant_layer(s, round);
        if (r != 0)
            for (int i = 0; i < W; i++) {
                gl_t in = w[START_FULL_0 + W * (r - 1) + i];
                out[t++] = gl_sub(s[i], in);
                s[i] = in;
            }
        for (int i = 0; i < W; i++) s[i] = sbox(s[i]);
        mds_layer(s);
        round++;
    }
    for (int i = 0; i < W; i++) s[i] = gl_add(s[i], gl_canon(FP_FIRST[i]));
    {
        gl_t r[W];
        r[0] = s[0];
        for (int c = 1; c < W; c++) {
            gl_t sum = 0;
            for (int rr = 1; rr < W; rr++) sum = gl_add(sum, gl_mul(s[rr], gl_canon(FP_INIT[rr - 1][c - 1])));
            r[c] = sum;
        }
        memcpy(s, r, sizeof r);
    }
    for (int k = 0; k < N_PARTIAL; k++) {
        gl_t in = w[START_PARTIAL + k];
        out[t++] = gl_sub(s[0], in);
        s[0] = sbox(in);
        if (k != N_PARTIAL - 1) s[0] = gl_add(s[0], gl_canon(FP_RC[k]));
        gl_t d = gl_mul(s[0], MDS_CIRC[0] + MDS_DIAG[0]);
        for (int i = 1; i < W; i++) d = gl_add(d, gl_mul(s[i], gl_canon(FP_WHATS[k][i - 1])));
        gl_t r[W];
        r[0] = d;
        for (int i = 1; i < W; i++) r[i] = gl_add(s[i], gl_mul(s[0], gl_canon(FP_VS[k][i - 1])));
        memcpy(s, r, sizeof r);
    }
    round += N_PARTIAL;
    for (int r = 0; r < HALF_FULL; r++) {
        constant_layer(s, round);
        for (int i = 0; i < W; i++) {
            gl_t in = w[START_FULL_1 + W * r + i];
            out[t++] = gl_sub(s[i], in);
            s[i] = in;
        }
        for (int i = 0; i < W; i++) s[i] = sbox(s[i]);
        mds_layer(s);
        round++;
    }
    for (int i = 0; i < W; i++) out[t++] = gl_sub(s[i], w[W + i]);
}

/* ------------------------------------------------------------------ sponge / compression */

/* hash/hashing.rs:100-123 hash_n_to_m_no_pad with num_outputs = 4 (overwrite mode, rate 8) */
void gbo_gl_hash_no_pad(const gl_t *in, size_t n, gl_t out[HOUT]) {
    gl_t st[W] = {0};
    for (size_t off = 0; off < n; off += RATE) {
        size_t k = n - off < RATE ? n - off : RATE;
        memcpy(st, in + off, k * sizeof(gl_t));
        gbo_gl_poseidon(st, st);
    }
    memcpy(out, st, HOUT * sizeof(gl_t));
}

/* generic m-output squeeze (hash/hashing.rs:112-122) */
void gbo_gl_hash_n_to_m_no_pad(const gl_t *in, size_t n, gl_t *out, size_t m) {
    gl_t st[W] = {0};
    for (size_t off = 0; off < n; off += RATE) {
        size_t k = n - off < RATE ? n - off : RATE;
        memcpy(st, in + off, k * sizeof(gl_t));
        gbo_gl_poseidon(st, st);
    }
    size_t got = 0;
    for (;;) {
        for (int i = 0; i < RATE; i++) {
            out[got++] = st[i];
            if (got == m) return;
        }
        gbo_gl_poseidon(st, st);
    }
}

/* plonk/config.rs:70-84 hash_or_noop: <= 4 elements are zero-padded, not hashed */
void gbo_gl_hash_or_noop(const gl_t *in, size_t n, gl_t out[HOUT]) {
    if (n <= HOUT) {
        memset(out, 0, HOUT * sizeof(gl_t));
        memcpy(out, in, n * sizeof(gl_t));
    } else {
        gbo_gl_hash_no_pad(in, n, out);
    }
}

/* hash/hashing.rs:76-96 compress == Hasher::two_to_one (poseidon_goldilocks.rs:1108-1110) */
void gbo_gl_two_to_one(const gl_t l[HOUT], const gl_t r[HOUT], gl_t out[HOUT]) {
    gl_t st[W] = {0};
    memcpy(st, l, HOUT * sizeof(gl_t));
    memcpy(st + HOUT, r, HOUT * sizeof(gl_t));
    gbo_gl_poseidon(st, st);
    memcpy(out, st, HOUT * sizeof(gl_t));
}

/* ------------------------------------------------------------------ bit reversal */

static inline size_t rev_bits(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

/* util/src/lib.rs:189-238 reverse_index_bits_in_place (semantics: swap i <-> bitrev(i)) */
void gbo_reverse_index_bits_u64(uint64_t *a, unsigned lg_n) {
    size_t n = (size_t)1 << lg_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = rev_bits(i, lg_n);
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

/* ------------------------------------------------------------------ NTT */

/* field/src/fft.rs:12-31 fft_root_table: row lg_m-1 holds powers of w_{2^lg_m}, max(half_m,2) entries */
typedef struct { unsigned lg_n; gl_t **rows; } root_table_t;

static root_table_t *root_table_new(unsigned lg_n) {
    root_table_t *t = malloc(sizeof *t);
    t->lg_n = lg_n;
    t->rows = lg_n ? malloc(lg_n * sizeof(gl_t *)) : NULL;
    for (unsigned lg_m = 1; lg_m <= lg_n; lg_m++) {
        size_t half_m = (size_t)1 << (lg_m - 1);
        size_t len = half_m < 2 ? 2 : half_m;
        gl_t base = gl_two_adic_generator(lg_m);
        gl_t *row = malloc(len * sizeof(gl_t));
        gl_t x = 1;
        for (size_t i = 0; i < len; i++) { row[i] = x; x = gl_mul(x, base); }
        t->rows[lg_m - 1] = row;
    }
    return t;
}
static void root_table_free(root_table_t *t) {
    for (unsigned i = 0; i < t->lg_n; i++) free(t->rows[i]);
    free(t->rows);
    free(t);
}

/* field/src/fft.rs:168-205 fft_classic + :98-160 fft_classic_simd (scalar instantiation):
 * bit-reverse, zero-tail replication for the first r layers, then DIT layers r..lg_n. */
static void fft_classic(gl_t *v, unsigned lg_n, unsigned r, const root_table_t *t) {
    size_t n = (size_t)1 << lg_n;
    gbo_reverse_index_bits_u64(v, lg_n);
    if (r > 0) {
        size_t mask = ~(((size_t)1 << r) - 1);
        for (size_t i = 0; i < n; i++) v[i] = v[i & mask];
    }
    for (unsigned lg_half_m = r; lg_half_m < lg_n; lg_half_m++) {
        size_t half_m = (size_t)1 << lg_half_m, m = half_m * 2;
        const gl_t *om = t->rows[lg_half_m];
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half_m; j++) {
                gl_t tt = gl_mul(om[j], v[k + half_m + j]);
                gl_t u = v[k + j];
                v[k + j] = gl_add(u, tt);
                v[k + half_m + j] = gl_sub(u, tt);
            }
    }
}

/* fft_with_options (fft.rs:55-63); table==NULL -> built per call like fft_dispatch :39-46 */
static void fft_opt(gl_t *v, unsigned lg_n, unsigned zero_factor, const root_table_t *table) {
    if (lg_n == 0) return;
    root_table_t *own = NULL;
    if (!table) table = own = root_table_new(lg_n);
    fft_classic(v, lg_n, zero_factor, table);
    if (own) root_table_free(own);
}

void gbo_gl_fft(gl_t *v, unsigned lg_n, unsigned zero_factor) { fft_opt(v, lg_n, zero_factor, NULL); }

/* field/src/fft.rs:70-94 ifft_with_options: forward transform, then reverse all but first and scale by 1/n */
void gbo_gl_ifft(gl_t *v, unsigned lg_n) {
    size_t n = (size_t)1 << lg_n;
    gl_t n_inv = gl_pow(gl_inv(2), lg_n);
    fft_opt(v, lg_n, 0, NULL);
    v[0] = gl_mul(v[0], n_inv);
    if (n > 1) v[n / 2] = gl_mul(v[n / 2], n_inv);
    for (size_t i = 1; i < n / 2; i++) {
        size_t j = n - i;
        gl_t ci = gl_mul(v[j], n_inv), cj = gl_mul(v[i], n_inv);
        v[i] = ci;
        v[j] = cj;
    }
}

/* field/src/polynomial/mod.rs:282-295 coset_fft_with_options: c_i *= shift^i, then fft */
static void coset_fft_opt(gl_t *v, unsigned lg_n, gl_t shift, unsigned zero_factor, const root_table_t *t) {
    size_t n = (size_t)1 << lg_n;
    gl_t p = 1;
    for (size_t i = 0; i < n; i++) { v[i] = gl_mul(v[i], p); p = gl_mul(p, shift); }
    fft_opt(v, lg_n, zero_factor, t);
}
void gbo_gl_coset_fft(gl_t *v, unsigned lg_n, gl_t shift, unsigned zero_factor) {
    coset_fft_opt(v, lg_n, shift, zero_factor, NULL);
}

/* field/src/polynomial/mod.rs:62-72 coset_ifft */
void gbo_gl_coset_ifft(gl_t *v, unsigned lg_n, gl_t shift) {
    size_t n = (size_t)1 << lg_n;
    gbo_gl_ifft(v, lg_n);
    gl_t si = gl_inv(shift), p = 1;
    for (size_t i = 0; i < n; i++) { v[i] = gl_mul(v[i], p); p = gl_mul(p, si); }
}

/* ------------------------------------------------------------------ Merkle tree */

/* hash/merkle_tree.rs:86-113 fill_subtree, same in-order interleaved digest layout (:50-58).
 * digests_len counts hashes. rayon::join -> omp task above a size cut-off. */
static void fill_subtree(gl_t *digests, size_t digests_len, const gl_t *leaves, size_t nleaves, size_t width,
                         gl_t out[HOUT]) {
    if (digests_len == 0) {
        gbo_gl_hash_or_noop(leaves, width, out);
        return;
    }
    size_t half = digests_len / 2;
    gl_t *left_buf = digests, *left_mem = digests + (half - 1) * HOUT;
    gl_t *right_mem = digests + half * HOUT, *right_buf = digests + (half + 1) * HOUT;
    gl_t l[HOUT], r[HOUT];
    if (nleaves >= 512) {
#pragma omp task shared(l)
        fill_subtree(left_buf, half - 1, leaves, nleaves / 2, width, l);
#pragma omp task shared(r)
        fill_subtree(right_buf, half - 1, leaves + (nleaves / 2) * width, nleaves / 2, width, r);
#pragma omp taskwait
    } else {
        fill_subtree(left_buf, half - 1, leaves, nleaves / 2, width, l);
        fill_subtree(right_buf, half - 1, leaves + (nleaves / 2) * width, nleaves / 2, width, r);
    }
    memcpy(left_mem, l, sizeof l);
    memcpy(right_mem, r, sizeof r);
    gbo_gl_two_to_one(l, r, out);
}

/* hash/merkle_tree.rs:152-181 MerkleTree::new (+ fill_digests_buf :115-149).
 * leaves [L][width] row-major; digests [2(L-2^cap)][4]; cap [2^cap][4]. Returns 0, or -1 on the
 * reference's assert (cap_height > log2 L). */
int gbo_gl_merkle_tree(const gl_t *leaves, size_t log_l, size_t width, unsigned cap_height, gl_t *digests, gl_t *cap) {
    if (cap_height > log_l) return -1;
    size_t L = (size_t)1 << log_l, ncap = (size_t)1 << cap_height;
    size_t num_digests = 2 * (L - ncap);
    if (num_digests == 0) {
        for (size_t i = 0; i < L; i++) gbo_gl_hash_or_noop(leaves + i * width, width, cap + i * HOUT);
        return 0;
    }
    size_t sub_d = num_digests >> cap_height, sub_l = L >> cap_height;
#pragma omp parallel
#pragma omp single
    for (size_t s = 0; s < ncap; s++) {
#pragma omp task
        fill_subtree(digests + s * sub_d * HOUT, sub_d, leaves + s * sub_l * width, sub_l, width, cap + s * HOUT);
    }
    return 0;
}

/* hash/merkle_tree.rs:188-222 MerkleTree::prove; returns number of siblings written */
int gbo_gl_merkle_prove(const gl_t *digests, size_t log_l, unsigned cap_height, size_t leaf_index, gl_t *siblings) {
    size_t L = (size_t)1 << log_l;
    unsigned num_layers = (unsigned)log_l - cap_height;
    size_t num_digests = 2 * (L - ((size_t)1 << cap_height));
    size_t tree_index = leaf_index >> num_layers;
    size_t tree_len = num_digests >> cap_height;
    const gl_t *tree = digests + tree_len * tree_index * HOUT;
    size_t pair_index = leaf_index & (((size_t)1 << num_layers) - 1);
    for (unsigned i = 0; i < num_layers; i++) {
        size_t parity = pair_index & 1;
        pair_index >>= 1;
        size_t siblings_index = (pair_index << (i + 1)) + ((size_t)1 << i) - 1;
        size_t sibling_index = 2 * siblings_index + (1 - parity);
        memcpy(siblings + i * HOUT, tree + sibling_index * HOUT, HOUT * sizeof(gl_t));
    }
    return (int)num_layers;
}

/* hash/merkle_proofs.rs:54-76 verify_merkle_proof_to_cap; returns 1 if ok */
int gbo_gl_merkle_verify(const gl_t *leaf, size_t width, size_t leaf_index, const gl_t *cap, const gl_t *siblings, unsigned nsib) {
    gl_t cur[HOUT];
    gbo_gl_hash_or_noop(leaf, width, cur);
    size_t idx = leaf_index;
    for (unsigned i = 0; i < nsib; i++) {
        const gl_t *sib = siblings + i * HOUT;
        gl_t nxt[HOUT];
        if (idx & 1) gbo_gl_two_to_one(sib, cur, nxt); else gbo_gl_two_to_one(cur, sib, nxt);
        memcpy(cur, nxt, sizeof cur);
        idx >>= 1;
    }
    return memcmp(cur, cap + idx * HOUT, sizeof cur) == 0;
}

/* ------------------------------------------------------------------ PolynomialBatch */

/* fri/oracle.rs:68-123 from_values / from_coeffs.
 *   cols      [ncols][n]   column-major input (values on H_n, or coefficients if is_coeffs)
 *   salts     NULL or [4][N] (the F::rand_vec columns of :144-148, host-supplied)
 *   coeffs    out [ncols][n]
 *   leaves    out [N][ncols+nsalt]  leaf i = LDE point bitrev(i)   (:108-109)
 *   digests   out [2(N-2^cap)][4], cap out [2^cap][4]
 * "IFFT" :76-80 builds a root table per column (v.ifft() passes None); "FFT + blinding"
 * :125-150 shares one table of size N (circuit_builder.rs:1227-1228). */
int gbo_gl_commit(const gl_t *cols, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height,
                  int is_coeffs, const gl_t *salts, gl_t *coeffs, gl_t *leaves, gl_t *digests, gl_t *cap) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    unsigned log_N = log_n + rate_bits;
    size_t nsalt = salts ? 4 : 0, width = ncols + nsalt;
    if (cap_height > log_N) return -1;
    memcpy(coeffs, cols, ncols * n * sizeof(gl_t));
    if (!is_coeffs) {
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t c = 0; c < ncols; c++) gbo_gl_ifft(coeffs + c * n, log_n);
    }
    root_table_t *table = root_table_new(log_N);
    gl_t *lde = malloc(ncols ? ncols * N * sizeof(gl_t) : 1);
    if (!lde) return -2;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t c = 0; c < ncols; c++) {
        gl_t *v = lde + c * N;
        memcpy(v, coeffs + c * n, n * sizeof(gl_t));
        memset(v + n, 0, (N - n) * sizeof(gl_t)); /* p.lde(rate_bits): polynomial/mod.rs:201-203 */
        coset_fft_opt(v, log_N, GL_GENERATOR, rate_bits, table);
    }
    root_table_free(table);
    /* transpose (util/mod.rs:25-31) + reverse_index_bits_in_place on the leaf vector */
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; i++) {
        size_t src = rev_bits(i, log_N);
        gl_t *row = leaves + i * width;
        for (size_t c = 0; c < ncols; c++) row[c] = lde[c * N + src];
        for (size_t s = 0; s < nsalt; s++) row[ncols + s] = salts[s * N + src];
    }
    free(lde);
    return gbo_gl_merkle_tree(leaves, log_N, width, cap_height, digests, cap);
}

/* ------------------------------------------------------------------ Challenger */

/* iop/challenger.rs:18-150 */
typedef struct {
    gl_t state[W];
    gl_t in[RATE];
    int nin;
    gl_t out[RATE];
    int nout;
} gbo_gl_challenger;

void gbo_gl_challenger_init(gbo_gl_challenger *c) { memset(c, 0, sizeof *c); }
static void duplexing(gbo_gl_challenger *c) {
    for (int i = 0; i < c->nin; i++) c->state[i] = c->in[i];
    c->nin = 0;
    gbo_gl_poseidon(c->state, c->state);
    memcpy(c->out, c->state, RATE * sizeof(gl_t));
    c->nout = RATE;
}
void gbo_gl_challenger_observe(gbo_gl_challenger *c, const gl_t *e, size_t n) {
    for (size_t i = 0; i < n; i++) {
        c->nout = 0;
        c->in[c->nin++] = e[i];
        if (c->nin == RATE) duplexing(c);
    }
}
gl_t gbo_gl_challenger_get(gbo_gl_challenger *c) {
    if (c->nin != 0 || c->nout == 0) duplexing(c);
    return c->out[--c->nout];
}
size_t gbo_gl_challenger_sizeof(void) { return sizeof(gbo_gl_challenger); }

/* ------------------------------------------------------------------ small field helpers for the python checker */
gl_t gbo_gl_mul(gl_t a, gl_t b) { return gl_mul(a, b); }
gl_t gbo_gl_powu(gl_t a, uint64_t e) { return gl_pow(a, e); }
gl_t gbo_gl_inv(gl_t a) { return gl_inv(a); }
gl_t gbo_gl_two_adic_generator(unsigned bits) { return gl_two_adic_generator(bits); }
/* the host may grant fewer CPUs than it shows (a cgroup quota): the caller sizes the pool to its share */
void gbo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int gbo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* out[i] = base^i ; out[i] = a[i] * k  (vector helpers for the python circuit restatement) */
void gbo_gl_powers(gl_t base, size_t n, gl_t *out) {
    gl_t x = 1;
    for (size_t i = 0; i < n; i++) { out[i] = x; x = gl_mul(x, base); }
}
void gbo_gl_scale_vec(const gl_t *a, gl_t k, size_t n, gl_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = gl_mul(a[i], k);
}
