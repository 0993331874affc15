/* TEST ORACLE - CPU restatement of the reference prover for the dummy circuit's gate set
 * {NoopGate, ConstantGate, PublicInputGate} over Goldilocks (D = 2, Poseidon-12), no lookups.
 *
 * Test infrastructure only (see gl.h).  Follows, in order (paths relative to
 * /root/reference/plonky2/src):
 *   plonk/prover.rs:228-447      internal_prove_with_partition_witness
 *   plonk/prover.rs:480-546      wires_permutation_partial_products_and_zs
 *   plonk/prover.rs:712-926      compute_quotient_polys
 *   plonk/vanishing_poly.rs:177-346 eval_vanishing_poly_base_batch, gates/gate.rs:188-215,391-404
 *   plonk/proof.rs:346-440       OpeningSet::new / to_fri_openings
 *   fri/oracle.rs:187-246        prove_openings
 *   fri/prover.rs:22-255         fri_proof (commit phase, PoW with the MINIMUM nonce, queries)
 *   util/serialization/mod.rs:2103-2151 proof byte layout
 * Checked by oracle/verifier.py (itself pinned by the reference's serialized regression proof).
 */
#include "gl.h"
#include <stdlib.h>
#include <string.h>

#define HOUT 4
#define D 2

/* from oracle_gl.c */
void gbo_gl_hash_no_pad(const gl_t *in, size_t n, gl_t out[HOUT]);
int gbo_gl_commit(const gl_t *cols, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height, int is_coeffs,
                  const gl_t *salts, gl_t *coeffs, gl_t *leaves, gl_t *digests, gl_t *cap);
int gbo_gl_merkle_tree(const gl_t *leaves, size_t log_l, size_t width, unsigned cap_height, gl_t *digests, gl_t *cap);
int gbo_gl_merkle_prove(const gl_t *digests, size_t log_l, unsigned cap_height, size_t leaf_index, gl_t *siblings);
void gbo_gl_coset_ifft(gl_t *v, unsigned lg_n, gl_t shift);
void gbo_gl_coset_fft(gl_t *v, unsigned lg_n, gl_t shift, unsigned zero_factor);
void gbo_gl_poseidon(const gl_t in[12], gl_t out[12]);
typedef struct { gl_t state[12]; gl_t in[8]; int nin; gl_t out[8]; int nout; } challenger_t;
void gbo_gl_challenger_init(challenger_t *c);
void gbo_gl_challenger_observe(challenger_t *c, const gl_t *e, size_t n);
gl_t gbo_gl_challenger_get(challenger_t *c);

typedef struct {
    unsigned num_wires, num_routed, num_constants /* const/sigma constants incl. selectors */, num_challenges;
    unsigned rate_bits, cap_height, pow_bits, num_queries, arity_bits, final_poly_bits, quotient_degree_factor;
    unsigned degree_bits;
    unsigned num_selectors;      /* 1 */
    unsigned gate_noop, gate_constant, gate_pi; /* indices in the sorted gate list (selector values) */
    unsigned num_gate_consts;    /* ConstantGate num_consts */
} gbo_circuit_cfg;

typedef struct {
    size_t ncols;
    unsigned log_n, rate_bits, cap_height;
    gl_t *coeffs, *leaves, *digests, *cap;
} batch_t;

static size_t rev_bits_sz(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

static int batch_commit(batch_t *b, const gl_t *cols, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height,
                        int is_coeffs) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    b->ncols = ncols; b->log_n = log_n; b->rate_bits = rate_bits; b->cap_height = cap_height;
    b->coeffs = malloc(ncols * n * sizeof(gl_t));
    b->leaves = malloc(N * ncols * sizeof(gl_t));
    b->digests = malloc((2 * (N - ((size_t)1 << cap_height)) + 1) * HOUT * sizeof(gl_t));
    b->cap = malloc(((size_t)HOUT << cap_height) * sizeof(gl_t));
    if (!b->coeffs || !b->leaves || !b->digests || !b->cap) return -2;
    return gbo_gl_commit(cols, ncols, log_n, rate_bits, cap_height, is_coeffs, NULL, b->coeffs, b->leaves, b->digests, b->cap);
}
static void batch_free(batch_t *b) { free(b->coeffs); free(b->leaves); free(b->digests); free(b->cap); }
/* fri/oracle.rs:153-158 */
static const gl_t *batch_lde(const batch_t *b, size_t index, size_t step) {
    unsigned bits = b->log_n + b->rate_bits;
    return b->leaves + rev_bits_sz(index * step, bits) * b->ncols;
}

static gl2_t challenger_ext(challenger_t *c) { gl2_t r; r.c[0] = gbo_gl_challenger_get(c); r.c[1] = gbo_gl_challenger_get(c); return r; }

typedef struct { uint8_t *p; size_t len, cap; } buf_t;
static void put(buf_t *b, const void *src, size_t n) {
    if (b->len + n <= b->cap) memcpy(b->p + b->len, src, n);
    b->len += n;
}
static void put_u64(buf_t *b, uint64_t x) { put(b, &x, 8); }
static void put_u8(buf_t *b, uint8_t x) { put(b, &x, 1); }
static void put_ext(buf_t *b, gl2_t x) { put_u64(b, x.c[0]); put_u64(b, x.c[1]); }

/* poly eval of base coefficients at an extension point: p.to_extension().eval(z) (plonk/proof.rs:359-363) */
static gl2_t eval_base_poly_ext(const gl_t *c, size_t n, gl2_t z) {
    gl2_t acc = gl2_from(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), gl2_from(c[i]));
    return acc;
}

/* Status: 0 ok, 1 = InvZeroPermArg (plonk/prover.rs:512-514), 2 = opening point in subgroup, <0 internal */
int gbo_gl_prove_dummy(const gbo_circuit_cfg *cfg, const gl_t *constants_sigmas /*[ncs][n] values*/,
                       const gl_t *circuit_digest, const gl_t *k_is, const gl_t *witness /*[num_wires][n]*/,
                       const gl_t *public_inputs, size_t num_public_inputs, uint8_t *out, size_t out_cap, size_t *out_len,
                       gl_t *debug_out /* optional: [betas c][gammas c][alphas c][zeta 2][fri_alpha 2][pow 1] */) {
    const unsigned c = cfg->num_challenges, r = cfg->rate_bits, lg = cfg->degree_bits, capH = cfg->cap_height;
    const size_t n = (size_t)1 << lg, N = n << r;
    const unsigned lgN = lg + r;
    const unsigned nw = cfg->num_wires, nr = cfg->num_routed, ncs = cfg->num_constants + nr;
    const unsigned qdf = cfg->quotient_degree_factor;
    const unsigned num_prods = (nr + qdf - 1) / qdf - 1; /* util/partial_products.rs:41-48 */
    const unsigned nchunks = num_prods + 1;
    if (((size_t)1 << r) != qdf) return -10; /* step = 1 case only (prover.rs:746-749) */
    buf_t ob = {out, 0, out_cap};
    int rc = 0;
    gl_t *qvals = NULL, *qchunks = NULL, *fc0 = NULL, *fc1 = NULL, *fri_caps = NULL;
    gl2_t *final_poly = NULL, *values = NULL, *o_cs = NULL, *o_w = NULL, *o_z = NULL, *o_zn = NULL, *o_q = NULL;
    gl_t **tree_leaves = NULL, **tree_digests = NULL;
    unsigned *tree_log = NULL;
    unsigned narity = 0, arity_bits_list[32];

    gl_t pi_hash[HOUT];
    gbo_gl_hash_no_pad(public_inputs, num_public_inputs, pi_hash); /* prover.rs:244 */

    batch_t cs = {0}, wires = {0}, zs = {0}, quot = {0};
    if ((rc = batch_commit(&cs, constants_sigmas, ncs, lg, r, capH, 0))) return rc;   /* circuit_builder.rs:1230-1239 */
    if ((rc = batch_commit(&wires, witness, nw, lg, r, capH, 0))) return rc;          /* prover.rs:261-272 */

    challenger_t ch;
    gbo_gl_challenger_init(&ch);
    gbo_gl_challenger_observe(&ch, circuit_digest, HOUT);
    gbo_gl_challenger_observe(&ch, pi_hash, HOUT);
    gbo_gl_challenger_observe(&ch, wires.cap, (size_t)HOUT << capH);
    gl_t *betas = malloc(c * sizeof(gl_t)), *gammas = malloc(c * sizeof(gl_t)), *alphas = malloc(c * sizeof(gl_t));
    for (unsigned i = 0; i < c; i++) betas[i] = gbo_gl_challenger_get(&ch);
    for (unsigned i = 0; i < c; i++) gammas[i] = gbo_gl_challenger_get(&ch);

    /* sigma values per row: ProverOnlyCircuitData.sigmas[row][j] = sigma_vecs[j][row] */
    const gl_t *sigma_cols = constants_sigmas + (size_t)cfg->num_constants * n;
    gl_t *subgroup = malloc(n * sizeof(gl_t));
    { gl_t w = gl_two_adic_generator(lg), x = 1; for (size_t i = 0; i < n; i++) { subgroup[i] = x; x = gl_mul(x, w); } }

    /* ---- prover.rs:480-546: Z and partial products.  zs_pp columns: [Z_0..Z_{c-1}, pp_{0,0..}, pp_{1,0..}, ...] */
    const size_t nzs = (size_t)c * (1 + num_prods);
    gl_t *zs_vals = malloc(nzs * n * sizeof(gl_t));
    {
        /* per-row chunk products in parallel (Rayon par_iter over the subgroup, prover.rs:497-528), then the
         * sequential running product (:531-539) */
        gl_t *cp = malloc((size_t)nchunks * n * sizeof(gl_t));
        for (unsigned i = 0; i < c && !rc; i++) {
            int bad = 0;
#pragma omp parallel for schedule(static) reduction(|:bad)
            for (size_t row = 0; row < n; row++) {
                gl_t x = subgroup[row];
                gl_t chunk_prod[64];
                for (unsigned m = 0; m < nchunks; m++) chunk_prod[m] = 1;
                for (unsigned j = 0; j < nr; j++) {
                    gl_t wv = witness[(size_t)j * n + row];
                    gl_t num = gl_add(gl_add(wv, gl_mul(betas[i], gl_mul(k_is[j], x))), gammas[i]);
                    gl_t den = gl_add(gl_add(wv, gl_mul(betas[i], sigma_cols[(size_t)j * n + row])), gammas[i]);
                    if (den == 0) { bad = 1; den = 1; }
                    gl_t q = gl_mul(num, gl_inv(den));
                    chunk_prod[j / qdf] = gl_mul(chunk_prod[j / qdf], q);
                }
                for (unsigned m = 0; m < nchunks; m++) cp[(size_t)m * n + row] = chunk_prod[m];
            }
            if (bad) { rc = 1; break; }
            /* partial_products_and_z_gx (util/partial_products.rs:29-38) then swap Z(gx) <-> Z(x) (:537-538) */
            gl_t z_x = 1;
            gl_t *Z = zs_vals + (size_t)i * n;
            for (size_t row = 0; row < n; row++) {
                gl_t acc = z_x;
                Z[row] = z_x;
                for (unsigned m = 0; m < nchunks; m++) {
                    acc = gl_mul(acc, cp[(size_t)m * n + row]);
                    if (m < num_prods) zs_vals[((size_t)c + (size_t)i * num_prods + m) * n + row] = acc;
                }
                z_x = acc;
            }
        }
        free(cp);
    }
    if (rc) goto done_early;
    if ((rc = batch_commit(&zs, zs_vals, nzs, lg, r, capH, 0))) goto done_early;    /* prover.rs:328-339 */
    gbo_gl_challenger_observe(&ch, zs.cap, (size_t)HOUT << capH);
    for (unsigned i = 0; i < c; i++) alphas[i] = gbo_gl_challenger_get(&ch);

    /* ---- prover.rs:712-926 compute_quotient_polys: step = 1, next_step = 2^r, lde_size = N */
    qvals = malloc((size_t)c * N * sizeof(gl_t));
    {
        /* ZeroPolyOnCoset (field/src/zero_poly_coset.rs:22-62) */
        gl_t g_pow_n = gl_pow(GL_GENERATOR, n);
        gl_t zh[64], zh_inv[64];
        gl_t wr = gl_two_adic_generator(r), xr = 1;
        for (unsigned i = 0; i < (1u << r); i++) { zh[i] = gl_sub(gl_mul(g_pow_n, xr), 1); zh_inv[i] = gl_inv(zh[i]); xr = gl_mul(xr, wr); }
        gl_t wN = gl_two_adic_generator(lgN);
        const unsigned nterms = c + c * nchunks + HOUT; /* z_1 terms, partial product terms, gate constraints (max = 4) */
        const unsigned nsel = cfg->num_selectors;
        /* Rayon par_chunks(BATCH_SIZE = 32) over the points (prover.rs:791-797) */
#pragma omp parallel for schedule(static)
        for (size_t i0 = 0; i0 < N; i0 += 32) {
        gl_t terms[256];
        gl_t pt = gl_pow(wN, i0);
        for (size_t i = i0; i < i0 + 32 && i < N; i++, pt = gl_mul(pt, wN)) {
            gl_t x = gl_mul(GL_GENERATOR, pt); /* shifted_x */
            size_t i_next = (i + ((size_t)1 << r)) % N;
            const gl_t *lcs = batch_lde(&cs, i, 1), *lw = batch_lde(&wires, i, 1), *lz = batch_lde(&zs, i, 1), *nz = batch_lde(&zs, i_next, 1);
            const gl_t *consts = lcs, *sig = lcs + cfg->num_constants;
            unsigned t = 0;
            /* eval_l_0 (zero_poly_coset.rs:58-61) */
            gl_t l0 = gl_mul(zh[i % (1u << r)], gl_inv(gl_mul((gl_t)(n % GL_P), gl_sub(x, 1))));
            for (unsigned k = 0; k < c; k++) terms[t++] = gl_mul(l0, gl_sub(lz[k], 1));
            for (unsigned k = 0; k < c; k++) {
                /* check_partial_products (util/partial_products.rs:53-77) */
                for (unsigned m = 0; m < nchunks; m++) {
                    gl_t np = 1, dp = 1;
                    for (unsigned j = m * qdf; j < nr && j < (m + 1) * qdf; j++) {
                        np = gl_mul(np, gl_add(gl_add(lw[j], gl_mul(betas[k], gl_mul(k_is[j], x))), gammas[k]));
                        dp = gl_mul(dp, gl_add(gl_add(lw[j], gl_mul(betas[k], sig[j])), gammas[k]));
                    }
                    gl_t prev = m == 0 ? lz[k] : lz[c + k * num_prods + m - 1];
                    gl_t next = m == nchunks - 1 ? nz[k] : lz[c + k * num_prods + m];
                    terms[t++] = gl_sub(gl_mul(prev, np), gl_mul(next, dp));
                }
            }
            /* gate constraints (vanishing_poly.rs:741-774): filter * unfiltered, summed per constraint index */
            {
                gl_t s = consts[0]; /* selector polynomial of the single group */
                const gl_t *gc = consts + nsel; /* remove_prefix(num_selectors) */
                unsigned ng = 3;
                gl_t cons[HOUT] = {0, 0, 0, 0};
                for (unsigned g = 0; g < ng; g++) {
                    /* compute_filter (gates/gate.rs:391-404): prod_{i in group, i != g} (i - s); single selector => no UNUSED term */
                    gl_t f = 1;
                    for (unsigned ii = 0; ii < ng; ii++) if (ii != g) f = gl_mul(f, gl_sub((gl_t)ii, s));
                    if (g == cfg->gate_constant)
                        for (unsigned j = 0; j < cfg->num_gate_consts; j++) cons[j] = gl_add(cons[j], gl_mul(f, gl_sub(gc[j], lw[j])));
                    else if (g == cfg->gate_pi)
                        for (unsigned j = 0; j < HOUT; j++) cons[j] = gl_add(cons[j], gl_mul(f, gl_sub(lw[j], pi_hash[j])));
                }
                for (unsigned j = 0; j < HOUT; j++) terms[t++] = cons[j];
            }
            /* reduce_with_powers_multi (plonk_common.rs:105-122), then * 1/Z_H (prover.rs:909-916) */
            for (unsigned k = 0; k < c; k++) {
                gl_t cum = 0;
                for (unsigned tt = t; tt-- > 0;) cum = gl_add(terms[tt], gl_mul(cum, alphas[k]));
                qvals[(size_t)k * N + i] = gl_mul(cum, zh_inv[i % (1u << r)]);
            }
        }
        }
        if (nterms > 256) rc = -11;
    }
    /* coset_ifft (prover.rs:921-925), trim_to_len + chunks (:361-374): c * qdf chunk polys of n coefficients */
    qchunks = malloc((size_t)c * qdf * n * sizeof(gl_t));
    for (unsigned k = 0; k < c; k++) {
        gbo_gl_coset_ifft(qvals + (size_t)k * N, lgN, GL_GENERATOR);
        memcpy(qchunks + (size_t)k * qdf * n, qvals + (size_t)k * N, (size_t)qdf * n * sizeof(gl_t));
    }
    if ((rc = batch_commit(&quot, qchunks, (size_t)c * qdf, lg, r, capH, 1))) goto done;   /* prover.rs:376-387 */
    gbo_gl_challenger_observe(&ch, quot.cap, (size_t)HOUT << capH);
    gl2_t zeta = challenger_ext(&ch);
    {
        gl2_t zn = zeta;
        for (unsigned i = 0; i < lg; i++) zn = gl2_mul(zn, zn);
        if (zn.c[0] == 1 && zn.c[1] == 0) { rc = 2; goto done; }
    }
    gl2_t g_ext = gl2_from(gl_two_adic_generator(lg));
    gl2_t zeta_next = gl2_mul(g_ext, zeta);

    /* ---- OpeningSet::new (plonk/proof.rs:346-387) */
    const size_t nq = (size_t)c * qdf;
    o_cs = malloc(ncs * sizeof(gl2_t)); o_w = malloc(nw * sizeof(gl2_t)); o_z = malloc(nzs * sizeof(gl2_t));
    o_zn = malloc(nzs * sizeof(gl2_t)); o_q = malloc(nq * sizeof(gl2_t));
#pragma omp parallel for
    for (size_t j = 0; j < ncs; j++) o_cs[j] = eval_base_poly_ext(cs.coeffs + j * n, n, zeta);
#pragma omp parallel for
    for (size_t j = 0; j < nw; j++) o_w[j] = eval_base_poly_ext(wires.coeffs + j * n, n, zeta);
#pragma omp parallel for
    for (size_t j = 0; j < nzs; j++) { o_z[j] = eval_base_poly_ext(zs.coeffs + j * n, n, zeta); o_zn[j] = eval_base_poly_ext(zs.coeffs + j * n, n, zeta_next); }
#pragma omp parallel for
    for (size_t j = 0; j < nq; j++) o_q[j] = eval_base_poly_ext(quot.coeffs + j * n, n, zeta);

    /* proof bytes so far: caps + openings (serialization/mod.rs:2103-2118, 1514-1529) */
    put(&ob, wires.cap, ((size_t)HOUT << capH) * 8);
    put(&ob, zs.cap, ((size_t)HOUT << capH) * 8);
    put(&ob, quot.cap, ((size_t)HOUT << capH) * 8);
    for (size_t j = 0; j < cfg->num_constants; j++) put_ext(&ob, o_cs[j]);           /* constants */
    for (size_t j = cfg->num_constants; j < ncs; j++) put_ext(&ob, o_cs[j]);         /* plonk_sigmas */
    for (size_t j = 0; j < nw; j++) put_ext(&ob, o_w[j]);                            /* wires */
    for (size_t j = 0; j < c; j++) put_ext(&ob, o_z[j]);                             /* plonk_zs */
    for (size_t j = 0; j < c; j++) put_ext(&ob, o_zn[j]);                            /* plonk_zs_next */
    /* lookup_zs, lookup_zs_next: empty */
    for (size_t j = c; j < nzs; j++) put_ext(&ob, o_z[j]);                           /* partial_products */
    for (size_t j = 0; j < nq; j++) put_ext(&ob, o_q[j]);                            /* quotient_polys */

    /* observe_openings(to_fri_openings) (plonk/proof.rs:388-440, fri/challenges.rs:15-23) */
    for (size_t j = 0; j < ncs; j++) gbo_gl_challenger_observe(&ch, o_cs[j].c, 2);
    for (size_t j = 0; j < nw; j++) gbo_gl_challenger_observe(&ch, o_w[j].c, 2);
    for (size_t j = 0; j < c; j++) gbo_gl_challenger_observe(&ch, o_z[j].c, 2);
    for (size_t j = c; j < nzs; j++) gbo_gl_challenger_observe(&ch, o_z[j].c, 2);
    for (size_t j = 0; j < nq; j++) gbo_gl_challenger_observe(&ch, o_q[j].c, 2);
    for (size_t j = 0; j < c; j++) gbo_gl_challenger_observe(&ch, o_zn[j].c, 2);

    /* ---- prove_openings (fri/oracle.rs:187-246) */
    gl2_t fri_alpha = challenger_ext(&ch);
    final_poly = calloc(N, sizeof(gl2_t)); /* lde(rate_bits): zero padded to N */
    {
        const batch_t *oracles[4] = {&cs, &wires, &zs, &quot};
        const size_t counts[4] = {ncs, nw, nzs, nq};
        gl2_t *comp = malloc(n * sizeof(gl2_t));
        for (int batch = 0; batch < 2; batch++) {
            /* reduce_polys_base (util/reducing.rs:89-103): sum_j alpha^j * poly_j, powers restart at 1 */
            for (size_t t = 0; t < n; t++) comp[t] = gl2_from(0);
            gl2_t ap = gl2_from(1);
            size_t count = 0;
            for (int o = 0; o < 4; o++) {
                size_t lo = 0, hi = counts[o];
                if (batch == 1) { if (o != 2) continue; hi = c; } /* fri_zs_polys: zs_range of oracle 2 (circuit_data.rs:760-767) */
                for (size_t j = lo; j < hi; j++) {
                    const gl_t *p = oracles[o]->coeffs + j * n;
                    for (size_t t = 0; t < n; t++) comp[t] = gl2_add(comp[t], gl2_scale(ap, p[t]));
                    ap = gl2_mul(ap, fri_alpha);
                    count++;
                }
            }
            /* divide_by_linear (field/src/polynomial/division.rs:75-88) + push zero */
            gl2_t point = batch == 0 ? zeta : zeta_next;
            gl2_t *q = malloc(n * sizeof(gl2_t));
            gl2_t acc = gl2_from(0);
            for (size_t t = n; t-- > 0;) { acc = gl2_add(gl2_mul(acc, point), comp[t]); if (t > 0) q[t - 1] = acc; }
            q[n - 1] = gl2_from(0);
            /* alpha.shift_poly(&mut final_poly); final_poly += quotient (oracle.rs:222-223) */
            gl2_t sh = gl2_pow(fri_alpha, count);
            for (size_t t = 0; t < n; t++) final_poly[t] = gl2_add(gl2_mul(final_poly[t], sh), q[t]);
            free(q);
        }
        free(comp);
    }
    /* coset_fft in the extension field == base NTT on each coordinate (oracle.rs:226-231) */
    fc0 = malloc(N * sizeof(gl_t)); fc1 = malloc(N * sizeof(gl_t));
    values = malloc(N * sizeof(gl2_t));
    gl2_t *coeffs = final_poly;
    size_t cur_len = N;
    unsigned cur_lg = lgN;

    /* ---- fri_committed_trees (fri/prover.rs:83-133) */
    { /* ConstantArityBits (fri/reduction_strategies.rs:44-56) */
        unsigned db = lg;
        while (db > cfg->final_poly_bits && db + r >= capH + cfg->arity_bits) { arity_bits_list[narity++] = cfg->arity_bits; db -= cfg->arity_bits; }
    }
    tree_leaves = calloc(narity + 1, sizeof(gl_t *)); tree_digests = calloc(narity + 1, sizeof(gl_t *));
    tree_log = calloc(narity + 1, sizeof(unsigned));
    gl_t shift = GL_GENERATOR;
    {
        for (size_t t = 0; t < N; t++) { fc0[t] = coeffs[t].c[0]; fc1[t] = coeffs[t].c[1]; }
        gbo_gl_coset_fft(fc0, lgN, shift, 0);
        gbo_gl_coset_fft(fc1, lgN, shift, 0);
        for (size_t t = 0; t < N; t++) { values[t].c[0] = fc0[t]; values[t].c[1] = fc1[t]; }
    }
    fri_caps = malloc(narity * ((size_t)HOUT << capH) * sizeof(gl_t) + 8);
    for (unsigned li = 0; li < narity; li++) {
        unsigned ab = arity_bits_list[li];
        size_t arity = (size_t)1 << ab, nleaves = cur_len >> ab, width = arity * D;
        gl_t *lv = malloc(nleaves * width * sizeof(gl_t));
        /* reverse_index_bits_in_place(values); chunk by arity; flatten */
        for (size_t i = 0; i < cur_len; i++) {
            size_t src = rev_bits_sz(i, cur_lg);
            lv[i * D] = values[src].c[0];
            lv[i * D + 1] = values[src].c[1];
        }
        gl_t *dg = malloc((2 * (nleaves - ((size_t)1 << capH)) + 1) * HOUT * sizeof(gl_t));
        gl_t *cap = fri_caps + li * ((size_t)HOUT << capH);
        if ((rc = gbo_gl_merkle_tree(lv, cur_lg - ab, width, capH, dg, cap))) goto done;
        tree_leaves[li] = lv; tree_digests[li] = dg; tree_log[li] = cur_lg - ab;
        gbo_gl_challenger_observe(&ch, cap, (size_t)HOUT << capH);
        gl2_t beta = challenger_ext(&ch);
        /* fold: reduce_with_powers(chunk, beta) (plonk_common.rs:124-136) */
        size_t new_len = cur_len >> ab;
        for (size_t m = 0; m < new_len; m++) {
            gl2_t s = gl2_from(0);
            for (size_t t = arity; t-- > 0;) s = gl2_add(gl2_mul(s, beta), coeffs[m * arity + t]);
            coeffs[m] = s;
        }
        cur_len = new_len; cur_lg -= ab;
        shift = gl_pow(shift, arity);
        for (size_t t = 0; t < cur_len; t++) { fc0[t] = coeffs[t].c[0]; fc1[t] = coeffs[t].c[1]; }
        gbo_gl_coset_fft(fc0, cur_lg, shift, 0);
        gbo_gl_coset_fft(fc1, cur_lg, shift, 0);
        for (size_t t = 0; t < cur_len; t++) { values[t].c[0] = fc0[t]; values[t].c[1] = fc1[t]; }
    }
    size_t final_len = cur_len >> r;
    for (size_t t = 0; t < final_len; t++) gbo_gl_challenger_observe(&ch, coeffs[t].c, 2);

    /* ---- fri_proof_of_work (fri/prover.rs:136-188): minimum nonce (== find_any with one thread) */
    gl_t pow_witness = 0;
    {
        unsigned min_lz = cfg->pow_bits; /* + (64 - order.bits()) = 0 for Goldilocks */
        gl_t st[12];
        memcpy(st, ch.state, sizeof st);
        for (int i = 0; i < ch.nin; i++) st[i] = ch.in[i];
        int pos = ch.nin;
        for (gl_t cand = 0;; cand++) {
            gl_t s2[12];
            memcpy(s2, st, sizeof s2);
            s2[pos] = cand;
            gbo_gl_poseidon(s2, s2);
            gl_t resp = s2[7]; /* squeeze().last(): rate 8 */
            unsigned lz = resp ? (unsigned)__builtin_clzll(resp) : 64;
            if (lz >= min_lz) { pow_witness = cand; break; }
        }
        gbo_gl_challenger_observe(&ch, &pow_witness, 1);
        gl_t resp = gbo_gl_challenger_get(&ch);
        unsigned lz = resp ? (unsigned)__builtin_clzll(resp) : 64;
        if (lz < min_lz) { rc = -20; goto done; }
        if (debug_out) debug_out[3 * c + 4] = resp;
    }

    /* FRI proof bytes (serialization/mod.rs:1679-1695): caps, query rounds, final poly, pow witness */
    put(&ob, fri_caps, narity * ((size_t)HOUT << capH) * 8);
    {
        const batch_t *oracles[4] = {&cs, &wires, &zs, &quot};
        gl_t sib[64 * HOUT];
        for (unsigned qi = 0; qi < cfg->num_queries; qi++) {
            size_t x_index = (size_t)(gbo_gl_challenger_get(&ch) % N);
            for (int o = 0; o < 4; o++) {
                const batch_t *b = oracles[o];
                put(&ob, b->leaves + x_index * b->ncols, b->ncols * 8);
                int ns = gbo_gl_merkle_prove(b->digests, lgN, capH, x_index, sib);
                put_u8(&ob, (uint8_t)ns);
                put(&ob, sib, (size_t)ns * HOUT * 8);
            }
            size_t xi = x_index;
            for (unsigned li = 0; li < narity; li++) {
                unsigned ab = arity_bits_list[li];
                size_t width = ((size_t)1 << ab) * D;
                size_t leaf = xi >> ab;
                put(&ob, tree_leaves[li] + leaf * width, width * 8);
                int ns = gbo_gl_merkle_prove(tree_digests[li], tree_log[li], capH, leaf, sib);
                put_u8(&ob, (uint8_t)ns);
                put(&ob, sib, (size_t)ns * HOUT * 8);
                xi = leaf;
            }
        }
    }
    for (size_t t = 0; t < final_len; t++) put_ext(&ob, coeffs[t]);
    put_u64(&ob, pow_witness);
    /* ProofWithPublicInputs (serialization/mod.rs:2134-2151) */
    put_u64(&ob, num_public_inputs);
    put(&ob, public_inputs, num_public_inputs * 8);
    *out_len = ob.len;
    if (ob.len > ob.cap) rc = -30;

    if (debug_out) {
        for (unsigned i = 0; i < c; i++) { debug_out[i] = betas[i]; debug_out[c + i] = gammas[i]; debug_out[2 * c + i] = alphas[i]; }
        debug_out[3 * c] = zeta.c[0]; debug_out[3 * c + 1] = zeta.c[1];
        debug_out[3 * c + 2] = fri_alpha.c[0]; debug_out[3 * c + 3] = fri_alpha.c[1];
    }
done:
    if (tree_leaves) for (unsigned li = 0; li < narity; li++) { free(tree_leaves[li]); free(tree_digests[li]); }
    free(tree_leaves); free(tree_digests); free(tree_log); free(fri_caps);
    free(fc0); free(fc1); free(values);
    free(o_cs); free(o_w); free(o_z); free(o_zn); free(o_q);
    free(final_poly);
    free(qvals); free(qchunks);
done_early:
    free(zs_vals); free(subgroup); free(betas); free(gammas); free(alphas);
    batch_free(&cs); batch_free(&wires); batch_free(&zs); batch_free(&quot);
    return rc;
}
