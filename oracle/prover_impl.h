/* TEST ORACLE - field-generic body of the CPU restatement of the reference prover for the dummy circuit's
 * gate set {NoopGate, ConstantGate, PublicInputGate}.  Included by prover_gl.c (Goldilocks, D = 2, Poseidon-12)
 * and prover_bb.c (BabyBear, D = 4, Poseidon2-16) after they define the F_ / E_ / X_ macros.
 *
 * Test infrastructure only.  Follows, in order (paths relative to /root/reference/plonky2/src):
 *   plonk/prover.rs:228-447      internal_prove_with_partition_witness
 *   plonk/prover.rs:480-546      wires_permutation_partial_products_and_zs
 *   plonk/prover.rs:712-926      compute_quotient_polys
 *   plonk/vanishing_poly.rs:177-346 eval_vanishing_poly_base_batch, gates/gate.rs:188-215,391-404
 *   plonk/proof.rs:346-440       OpeningSet::new / to_fri_openings
 *   fri/oracle.rs:187-246        prove_openings
 *   fri/prover.rs:22-255         fri_proof (commit phase, PoW with the MINIMUM nonce, queries)
 *   util/serialization/mod.rs:2103-2151 proof byte layout
 * Checked by oracle/verifier.py (pinned, for Goldilocks, by the reference's serialized regression proof).
 */
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* GBO_TIMING=1: wall time of each phase to stderr (named like the reference's timed!() scopes) */
#ifndef GBO_SCOPE
#define GBO_SCOPE(name)                                                                              \
    do {                                                                                             \
        if (gbo_timing_on) {                                                                         \
            double t_now = omp_get_wtime();                                                          \
            fprintf(stderr, "[oracle prove] %-44s %8.3f s\n", name, t_now - gbo_t_last);             \
            gbo_t_last = t_now;                                                                      \
        }                                                                                            \
    } while (0)
#endif

extern double gbo_last_cs_commit_seconds; /* defined in oracle_gl.c */
extern unsigned char gbo_last_cs_cap[8192];
extern size_t gbo_last_cs_cap_bytes;

#ifndef GBO_CIRCUIT_CFG_DEFINED
#define GBO_CIRCUIT_CFG_DEFINED
typedef struct {
    unsigned num_wires, num_routed, num_constants /* const/sigma constants incl. selectors */, num_challenges;
    unsigned rate_bits, cap_height, pow_bits, num_queries, arity_bits, final_poly_bits, quotient_degree_factor;
    unsigned degree_bits;
    unsigned num_selectors;      /* 1 */
    unsigned gate_noop, gate_constant, gate_pi; /* indices in the sorted gate list (selector values) */
    unsigned num_gate_consts;    /* ConstantGate num_consts */
    /* CommonCircuitData.gates sorted by (degree, id) with selectors_info flattened: {kind, param, selector_index,
     * group_start, group_end}; kind 0 Noop, 1 Constant{param}, 2 PublicInput, 3 Arithmetic{param = num_ops}, 4 Poseidon,
     * 5 Poseidon2BabyBear{param = num_ops} */
    unsigned num_gates;
    unsigned gates[16][5];
} gbo_circuit_cfg;
#endif


#define GBO_MAX_GATE_CONSTRAINTS 160
#define GBO_MAX_TERMS 512

typedef struct {
    size_t ncols, width; /* width = ncols + salt columns */
    unsigned log_n, rate_bits, cap_height;
    F_T *coeffs, *leaves, *digests, *cap;
} batch_t;

static size_t rev_bits_sz(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

/* salts: NULL (blinding = false) or the SALT_SIZE = 4 salt columns [4][N] (fri/oracle.rs:133-148) */
static int batch_commit(batch_t *b, const F_T *cols, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height,
                        int is_coeffs, const F_T *salts) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    b->ncols = ncols; b->log_n = log_n; b->rate_bits = rate_bits; b->cap_height = cap_height;
    b->width = ncols + (salts ? 4 : 0);
    b->coeffs = malloc(ncols * n * sizeof(F_T));
    b->leaves = malloc(N * b->width * sizeof(F_T));
    b->digests = malloc((2 * (N - ((size_t)1 << cap_height)) + 1) * HOUT * sizeof(F_T));
    b->cap = malloc(((size_t)HOUT << cap_height) * sizeof(F_T));
    if (!b->coeffs || !b->leaves || !b->digests || !b->cap) return -2;
    return X_COMMIT(cols, ncols, log_n, rate_bits, cap_height, is_coeffs, salts, b->coeffs, b->leaves, b->digests, b->cap);
}
static void batch_free(batch_t *b) { free(b->coeffs); free(b->leaves); free(b->digests); free(b->cap); }
/* fri/oracle.rs:153-158 */
static const F_T *batch_lde(const batch_t *b, size_t index, size_t step) {
    unsigned bits = b->log_n + b->rate_bits;
    return b->leaves + rev_bits_sz(index * step, bits) * b->width;
}

static E_T challenger_ext(challenger_t *c) { E_T r; for (int k = 0; k < D; k++) r.c[k] = X_CH_GET(c); return r; }

typedef struct { uint8_t *p; size_t len, cap; } buf_t;
static void put(buf_t *b, const void *src, size_t n) {
    if (b->len + n <= b->cap) memcpy(b->p + b->len, src, n);
    b->len += n;
}
static void put_u64(buf_t *b, uint64_t x) { put(b, &x, 8); }
static void put_f(buf_t *b, F_T x) { put(b, &x, sizeof(F_T)); }
static void put_u8(buf_t *b, uint8_t x) { put(b, &x, 1); }
static void put_ext(buf_t *b, E_T x) { for (int k = 0; k < D; k++) put_f(b, x.c[k]); }

/* poly eval of base coefficients at an extension point: p.to_extension().eval(z) (plonk/proof.rs:359-363) */
static E_T eval_base_poly_ext(const F_T *c, size_t n, E_T z) {
    E_T acc = E_FROM(0);
    for (size_t i = n; i-- > 0;) acc = E_ADD(E_MUL(acc, z), E_FROM(c[i]));
    return acc;
}

/* Test hook: when set, the next proofs copy their intermediate values out - the Z / partial-product VALUES handed to
 * from_values (prover.rs:318-329) and the quotient chunk COEFFICIENTS handed to from_coeffs (:361-376) - so that the GPU's
 * stage-level entry points can be compared with the oracle prover's own intermediates (tests/test_gpu_stage_abi.py). */
static F_T *dump_zs_values = NULL, *dump_quotient_chunks = NULL;
#define X_CAT2(a, b) a##b
#define X_CAT(a, b) X_CAT2(a, b)
void X_CAT(X_PROVE_DUMMY, _set_dump)(F_T *zs_values, F_T *quotient_chunks) { dump_zs_values = zs_values; dump_quotient_chunks = quotient_chunks; }
/* Gate sets beyond the five kinds evaluated below (the recursion circuits' gates): the caller hands over the gate part of
 * eval_vanishing_poly_base_batch (plonk/vanishing_poly.rs:741-774) - for every point i of the quotient domain (shift * w_Q^i, Q = n * quotient_degree_factor) the
 * num_gate_constraints sums  sum_gates filter(selector) * unfiltered_constraint_j  - computed by oracle/plonk_dummy.py
 * gate_constraint_terms() with the evaluators of oracle/gates.py (the ones the reference's recursion proof pins);
 * [Q][nterms], used by the next proof instead of the inline evaluators. */
/* FriParams.reduction_arity_bits of a circuit whose FriReductionStrategy is Fixed(..) or MinSize(..) (fri/reduction_strategies.rs:
 * 29-56): when set, the next proofs use this list instead of deriving ConstantArityBits' from the configuration. */
static unsigned ext_arity_bits[32], ext_narity = 0;
static int ext_arity_set = 0;
void X_CAT(X_PROVE_DUMMY, _set_reduction_arity_bits)(const unsigned *bits, unsigned n) {
    ext_arity_set = bits != NULL;
    ext_narity = 0;
    for (unsigned i = 0; bits && i < n && i < 32; i++) ext_arity_bits[ext_narity++] = bits[i];
}
static const F_T *ext_gate_terms = NULL;
static unsigned ext_gate_nterms = 0;
void X_CAT(X_PROVE_DUMMY, _set_gate_terms)(const F_T *terms, unsigned nterms) { ext_gate_terms = terms; ext_gate_nterms = nterms; }

/* Status: 0 ok, 1 = InvZeroPermArg (plonk/prover.rs:512-514), 2 = opening point in subgroup, <0 internal */
int X_PROVE_DUMMY_SALTED(const gbo_circuit_cfg *cfg, const F_T *constants_sigmas /*[ncs][n] values*/,
                       const F_T *circuit_digest, const F_T *k_is, const F_T *witness /*[num_wires][n]*/,
                       const F_T *public_inputs, size_t num_public_inputs, uint8_t *out, size_t out_cap, size_t *out_len,
                       F_T *debug_out /* optional: [betas c][gammas c][alphas c][zeta D][fri_alpha D][pow 1] */,
                       const F_T *salts /* NULL, or zero-knowledge (prover.rs:267,334,382): [3][4][N] for wires, zs, quotient */) {
    const unsigned c = cfg->num_challenges, r = cfg->rate_bits, lg = cfg->degree_bits, capH = cfg->cap_height;
    const size_t n = (size_t)1 << lg, N = n << r;
    const unsigned lgN = lg + r;
    const unsigned nw = cfg->num_wires, nr = cfg->num_routed, ncs = cfg->num_constants + nr;
    const unsigned qdf = cfg->quotient_degree_factor;
    const unsigned num_prods = (nr + qdf - 1) / qdf - 1; /* util/partial_products.rs:41-48 */
    const unsigned nchunks = num_prods + 1;
    /* prover.rs:735-749: quotient_degree_bits = log2_ceil(quotient_degree_factor) <= rate_bits; the quotient is computed on every
     * step-th point of the commitments' LDE, step = 2^(rate_bits - quotient_degree_bits).  Power-of-two factors only (every
     * stock configuration has max_quotient_degree_factor = 8). */
    unsigned qb = 0;
    while (((size_t)1 << qb) < qdf) qb++;
    if (((size_t)1 << qb) != qdf || qb > r || qb > 6) return -10;
    const size_t qstep = (size_t)1 << (r - qb), Q = n << qb; /* step, lde_size */
    buf_t ob = {out, 0, out_cap};
    int rc = 0;
    const int gbo_timing_on = getenv("GBO_TIMING") != NULL;
    double gbo_t_last = omp_get_wtime();
    (void)gbo_timing_on; (void)gbo_t_last;
    F_T *qvals = NULL, *qchunks = NULL, *fc0 = NULL, *fc1 = NULL, *fri_caps = NULL;
    E_T *final_poly = NULL, *values = NULL, *o_cs = NULL, *o_w = NULL, *o_z = NULL, *o_zn = NULL, *o_q = NULL;
    F_T **tree_leaves = NULL, **tree_digests = NULL;
    unsigned *tree_log = NULL;
    unsigned narity = 0, arity_bits_list[32];

    F_T pi_hash[HOUT];
    X_HASH_NO_PAD(public_inputs, num_public_inputs, pi_hash); /* prover.rs:244 */

    batch_t cs = {0}, wires = {0}, zs = {0}, quot = {0};
    /* the constants/sigmas commitment is build() work (circuit_builder.rs:1230-1239), redone here only because this
     * oracle keeps no circuit object; its wall time is reported so that a prove() timing can leave it out */
    double t_cs = omp_get_wtime();
    if ((rc = batch_commit(&cs, constants_sigmas, ncs, lg, r, capH, 0, NULL))) return rc;
    gbo_last_cs_commit_seconds = omp_get_wtime() - t_cs;
    gbo_last_cs_cap_bytes = ((size_t)HOUT << capH) * sizeof(F_T);
    if (gbo_last_cs_cap_bytes <= sizeof gbo_last_cs_cap) memcpy(gbo_last_cs_cap, cs.cap, gbo_last_cs_cap_bytes);
    else gbo_last_cs_cap_bytes = 0;
    GBO_SCOPE("constants/sigmas commit (build() work)");
    if ((rc = batch_commit(&wires, witness, nw, lg, r, capH, 0, salts))) return rc;          /* prover.rs:261-272 */

    challenger_t ch;
    X_CH_INIT(&ch);
    X_CH_OBSERVE(&ch, circuit_digest, HOUT);
    X_CH_OBSERVE(&ch, pi_hash, HOUT);
    X_CH_OBSERVE(&ch, wires.cap, (size_t)HOUT << capH);
    F_T *betas = malloc(c * sizeof(F_T)), *gammas = malloc(c * sizeof(F_T)), *alphas = malloc(c * sizeof(F_T));
    for (unsigned i = 0; i < c; i++) betas[i] = X_CH_GET(&ch);
    for (unsigned i = 0; i < c; i++) gammas[i] = X_CH_GET(&ch);

    /* sigma values per row: ProverOnlyCircuitData.sigmas[row][j] = sigma_vecs[j][row] */
    const F_T *sigma_cols = constants_sigmas + (size_t)cfg->num_constants * n;
    F_T *subgroup = malloc(n * sizeof(F_T));
    { F_T w = F_TWO_ADIC(lg), x = 1; for (size_t i = 0; i < n; i++) { subgroup[i] = x; x = F_MUL(x, w); } }

    GBO_SCOPE("wires commit");
    /* ---- prover.rs:480-546: Z and partial products.  zs_pp columns: [Z_0..Z_{c-1}, pp_{0,0..}, pp_{1,0..}, ...] */
    const size_t nzs = (size_t)c * (1 + num_prods);
    F_T *zs_vals = malloc(nzs * n * sizeof(F_T));
    {
        /* per-row chunk products in parallel (Rayon par_iter over the subgroup, prover.rs:497-528), then the
         * sequential running product (:531-539) */
        F_T *cp = malloc((size_t)nchunks * n * sizeof(F_T));
        for (unsigned i = 0; i < c && !rc; i++) {
            int bad = 0;
#pragma omp parallel for schedule(static) reduction(|:bad)
            for (size_t row = 0; row < n; row++) {
                F_T x = subgroup[row];
                F_T chunk_prod[64];
                for (unsigned m = 0; m < nchunks; m++) chunk_prod[m] = 1;
                for (unsigned j = 0; j < nr; j++) {
                    F_T wv = witness[(size_t)j * n + row];
                    F_T num = F_ADD(F_ADD(wv, F_MUL(betas[i], F_MUL(k_is[j], x))), gammas[i]);
                    F_T den = F_ADD(F_ADD(wv, F_MUL(betas[i], sigma_cols[(size_t)j * n + row])), gammas[i]);
                    if (den == 0) { bad = 1; den = 1; }
                    F_T q = F_MUL(num, F_INV(den));
                    chunk_prod[j / qdf] = F_MUL(chunk_prod[j / qdf], q);
                }
                for (unsigned m = 0; m < nchunks; m++) cp[(size_t)m * n + row] = chunk_prod[m];
            }
            if (bad) { rc = 1; break; }
            /* partial_products_and_z_gx (util/partial_products.rs:29-38) then swap Z(gx) <-> Z(x) (:537-538) */
            F_T z_x = 1;
            F_T *Z = zs_vals + (size_t)i * n;
            for (size_t row = 0; row < n; row++) {
                F_T acc = z_x;
                Z[row] = z_x;
                for (unsigned m = 0; m < nchunks; m++) {
                    acc = F_MUL(acc, cp[(size_t)m * n + row]);
                    if (m < num_prods) zs_vals[((size_t)c + (size_t)i * num_prods + m) * n + row] = acc;
                }
                z_x = acc;
            }
        }
        free(cp);
    }
    if (rc) goto done_early;
    GBO_SCOPE("Z and partial products");
    if (dump_zs_values) memcpy(dump_zs_values, zs_vals, nzs * n * sizeof(F_T));
    if ((rc = batch_commit(&zs, zs_vals, nzs, lg, r, capH, 0, salts ? salts + 4 * N : NULL))) goto done_early;    /* prover.rs:328-339 */
    X_CH_OBSERVE(&ch, zs.cap, (size_t)HOUT << capH);
    for (unsigned i = 0; i < c; i++) alphas[i] = X_CH_GET(&ch);

    GBO_SCOPE("zs commit");
    /* ---- prover.rs:712-926 compute_quotient_polys: next_step = 2^qb, lde_size = Q = n 2^qb */
    qvals = malloc((size_t)c * Q * sizeof(F_T));
    {
        /* ZeroPolyOnCoset (field/src/zero_poly_coset.rs:22-62) */
        F_T g_pow_n = F_POW(F_GENERATOR, n);
        F_T zh[64], zh_inv[64];
        F_T wr = F_TWO_ADIC(qb), xr = 1;
        for (unsigned i = 0; i < (1u << qb); i++) { zh[i] = F_SUB(F_MUL(g_pow_n, xr), 1); zh_inv[i] = F_INV(zh[i]); xr = F_MUL(xr, wr); }
        F_T wN = F_TWO_ADIC(lg + qb); /* points = two_adic_subgroup(degree_bits + quotient_degree_bits) */
        /* num_gate_constraints = max over the gate set (circuit_builder.rs:1286-1290) */
        unsigned ngc = 0;
        for (unsigned g = 0; g < cfg->num_gates; g++) {
            const unsigned kind = cfg->gates[g][0], param = cfg->gates[g][1];
            unsigned m = kind == 1 || kind == 3 ? param : (kind == 2 ? HOUT : (kind == 4 ? 123 : (kind == 5 ? 150 * param : 0)));
            if (m > ngc) ngc = m;
        }
        if (ext_gate_terms) ngc = ext_gate_nterms;
        const unsigned nterms = c + c * nchunks + ngc; /* z_1 terms, partial product terms, gate constraints */
        const unsigned nsel = cfg->num_selectors;
        if (nterms > GBO_MAX_TERMS || ngc > GBO_MAX_GATE_CONSTRAINTS || (!ext_gate_terms && cfg->num_gates > 16)) { rc = -11; goto done; }
        /* Rayon par_chunks(BATCH_SIZE = 32) over the points (prover.rs:791-797) */
#pragma omp parallel for schedule(static)
        for (size_t i0 = 0; i0 < Q; i0 += 32) {
        F_T terms[GBO_MAX_TERMS];
        F_T pt = F_POW(wN, i0);
        for (size_t i = i0; i < i0 + 32 && i < Q; i++, pt = F_MUL(pt, wN)) {
            F_T x = F_MUL(F_GENERATOR, pt); /* shifted_x */
            size_t i_next = (i + ((size_t)1 << qb)) % Q;
            const F_T *lcs = batch_lde(&cs, i, qstep), *lw = batch_lde(&wires, i, qstep), *lz = batch_lde(&zs, i, qstep),
                      *nz = batch_lde(&zs, i_next, qstep); /* get_lde_values(i, step) (prover.rs:819-831) */
            const F_T *consts = lcs, *sig = lcs + cfg->num_constants;
            unsigned t = 0;
            /* eval_l_0 (zero_poly_coset.rs:58-61) */
            F_T l0 = F_MUL(zh[i % (1u << qb)], F_INV(F_MUL(F_FROM_U64(n), F_SUB(x, 1))));
            for (unsigned k = 0; k < c; k++) terms[t++] = F_MUL(l0, F_SUB(lz[k], 1));
            for (unsigned k = 0; k < c; k++) {
                /* check_partial_products (util/partial_products.rs:53-77) */
                for (unsigned m = 0; m < nchunks; m++) {
                    F_T np = 1, dp = 1;
                    for (unsigned j = m * qdf; j < nr && j < (m + 1) * qdf; j++) {
                        np = F_MUL(np, F_ADD(F_ADD(lw[j], F_MUL(betas[k], F_MUL(k_is[j], x))), gammas[k]));
                        dp = F_MUL(dp, F_ADD(F_ADD(lw[j], F_MUL(betas[k], sig[j])), gammas[k]));
                    }
                    F_T prev = m == 0 ? lz[k] : lz[c + k * num_prods + m - 1];
                    F_T next = m == nchunks - 1 ? nz[k] : lz[c + k * num_prods + m];
                    terms[t++] = F_SUB(F_MUL(prev, np), F_MUL(next, dp));
                }
            }
            /* gate constraints (vanishing_poly.rs:741-774): filter * unfiltered, summed per constraint index */
            if (ext_gate_terms) {
                for (unsigned j = 0; j < ngc; j++) terms[t++] = ext_gate_terms[i * ngc + j];
            } else {
                const F_T *gc = consts + nsel; /* remove_prefix(num_selectors) */
                F_T cons[GBO_MAX_GATE_CONSTRAINTS], gcons[GBO_MAX_GATE_CONSTRAINTS];
                for (unsigned j = 0; j < ngc; j++) cons[j] = 0;
                for (unsigned g = 0; g < cfg->num_gates; g++) {
                    const unsigned kind = cfg->gates[g][0], param = cfg->gates[g][1];
                    /* compute_filter (gates/gate.rs:391-404): prod_{i in group, i != g} (i - s) [* (UNUSED - s)] */
                    F_T s = consts[cfg->gates[g][2]];
                    F_T f = 1;
                    for (unsigned ii = cfg->gates[g][3]; ii < cfg->gates[g][4]; ii++) if (ii != g) f = F_MUL(f, F_SUB(F_FROM_U64(ii), s));
                    if (nsel > 1) f = F_MUL(f, F_SUB(F_FROM_U64(0xFFFFFFFFull), s));
                    unsigned m = 0;
                    if (kind == 1)         /* gates/constant.rs:64-72 */
                        for (; m < param; m++) gcons[m] = F_SUB(gc[m], lw[m]);
                    else if (kind == 2)    /* gates/public_input.rs:52-60 */
                        for (; m < HOUT; m++) gcons[m] = F_SUB(lw[m], pi_hash[m]);
                    else if (kind == 3)    /* gates/arithmetic_base.rs:209-226 */
                        for (; m < param; m++)
                            gcons[m] = F_SUB(lw[4 * m + 3], F_ADD(F_MUL(F_MUL(lw[4 * m], lw[4 * m + 1]), gc[0]), F_MUL(lw[4 * m + 2], gc[1])));
                    else if (kind == 4) {  /* gates/poseidon_goldilocks.rs:223-313 */
                        X_POSEIDON_GATE(lw, gcons);
                        m = 123;
                    } else if (kind == 5) { /* gates/poseidon2_babybear.rs:315-413 */
                        X_POSEIDON2_GATE(lw, param, gcons);
                        m = 150 * param;
                    }
                    for (unsigned j = 0; j < m; j++) cons[j] = F_ADD(cons[j], F_MUL(f, gcons[j]));
                }
                for (unsigned j = 0; j < ngc; j++) terms[t++] = cons[j];
            }
            /* reduce_with_powers_multi (plonk_common.rs:105-122), then * 1/Z_H (prover.rs:909-916) */
            for (unsigned k = 0; k < c; k++) {
                F_T cum = 0;
                for (unsigned tt = t; tt-- > 0;) cum = F_ADD(terms[tt], F_MUL(cum, alphas[k]));
                qvals[(size_t)k * Q + i] = F_MUL(cum, zh_inv[i % (1u << qb)]);
            }
        }
        }
    }
    /* coset_ifft (prover.rs:921-925), trim_to_len + chunks (:361-374): c * qdf chunk polys of n coefficients */
    qchunks = malloc((size_t)c * qdf * n * sizeof(F_T));
    for (unsigned k = 0; k < c; k++) {
        X_COSET_IFFT(qvals + (size_t)k * Q, lg + qb, F_GENERATOR);
        memcpy(qchunks + (size_t)k * qdf * n, qvals + (size_t)k * Q, (size_t)qdf * n * sizeof(F_T));
    }
    GBO_SCOPE("quotient values + coset_ifft");
    if (dump_quotient_chunks) memcpy(dump_quotient_chunks, qchunks, (size_t)c * qdf * n * sizeof(F_T));
    if ((rc = batch_commit(&quot, qchunks, (size_t)c * qdf, lg, r, capH, 1, salts ? salts + 8 * N : NULL))) goto done;   /* prover.rs:376-387 */
    X_CH_OBSERVE(&ch, quot.cap, (size_t)HOUT << capH);
    E_T zeta = challenger_ext(&ch);
    {
        E_T zn = zeta;
        for (unsigned i = 0; i < lg; i++) zn = E_MUL(zn, zn);
        { int is_one = zn.c[0] == 1; for (int k = 1; k < D; k++) is_one &= zn.c[k] == 0; if (is_one) { rc = 2; goto done; } }
    }
    E_T g_ext = E_FROM(F_TWO_ADIC(lg));
    E_T zeta_next = E_MUL(g_ext, zeta);

    GBO_SCOPE("quotient commit");
    /* ---- OpeningSet::new (plonk/proof.rs:346-387) */
    const size_t nq = (size_t)c * qdf;
    o_cs = malloc(ncs * sizeof(E_T)); o_w = malloc(nw * sizeof(E_T)); o_z = malloc(nzs * sizeof(E_T));
    o_zn = malloc(nzs * sizeof(E_T)); o_q = malloc(nq * sizeof(E_T));
#pragma omp parallel for
    for (size_t j = 0; j < ncs; j++) o_cs[j] = eval_base_poly_ext(cs.coeffs + j * n, n, zeta);
#pragma omp parallel for
    for (size_t j = 0; j < nw; j++) o_w[j] = eval_base_poly_ext(wires.coeffs + j * n, n, zeta);
#pragma omp parallel for
    for (size_t j = 0; j < nzs; j++) { o_z[j] = eval_base_poly_ext(zs.coeffs + j * n, n, zeta); o_zn[j] = eval_base_poly_ext(zs.coeffs + j * n, n, zeta_next); }
#pragma omp parallel for
    for (size_t j = 0; j < nq; j++) o_q[j] = eval_base_poly_ext(quot.coeffs + j * n, n, zeta);

    /* proof bytes so far: caps + openings (serialization/mod.rs:2103-2118, 1514-1529) */
    put(&ob, wires.cap, ((size_t)HOUT << capH) * sizeof(F_T));
    put(&ob, zs.cap, ((size_t)HOUT << capH) * sizeof(F_T));
    put(&ob, quot.cap, ((size_t)HOUT << capH) * sizeof(F_T));
    for (size_t j = 0; j < cfg->num_constants; j++) put_ext(&ob, o_cs[j]);           /* constants */
    for (size_t j = cfg->num_constants; j < ncs; j++) put_ext(&ob, o_cs[j]);         /* plonk_sigmas */
    for (size_t j = 0; j < nw; j++) put_ext(&ob, o_w[j]);                            /* wires */
    for (size_t j = 0; j < c; j++) put_ext(&ob, o_z[j]);                             /* plonk_zs */
    for (size_t j = 0; j < c; j++) put_ext(&ob, o_zn[j]);                            /* plonk_zs_next */
    /* lookup_zs, lookup_zs_next: empty */
    for (size_t j = c; j < nzs; j++) put_ext(&ob, o_z[j]);                           /* partial_products */
    for (size_t j = 0; j < nq; j++) put_ext(&ob, o_q[j]);                            /* quotient_polys */

    /* observe_openings(to_fri_openings) (plonk/proof.rs:388-440, fri/challenges.rs:15-23) */
    for (size_t j = 0; j < ncs; j++) X_CH_OBSERVE(&ch, o_cs[j].c, D);
    for (size_t j = 0; j < nw; j++) X_CH_OBSERVE(&ch, o_w[j].c, D);
    for (size_t j = 0; j < c; j++) X_CH_OBSERVE(&ch, o_z[j].c, D);
    for (size_t j = c; j < nzs; j++) X_CH_OBSERVE(&ch, o_z[j].c, D);
    for (size_t j = 0; j < nq; j++) X_CH_OBSERVE(&ch, o_q[j].c, D);
    for (size_t j = 0; j < c; j++) X_CH_OBSERVE(&ch, o_zn[j].c, D);

    GBO_SCOPE("opening set");
    /* ---- prove_openings (fri/oracle.rs:187-246) */
    E_T fri_alpha = challenger_ext(&ch);
    final_poly = calloc(N, sizeof(E_T)); /* lde(rate_bits): zero padded to N */
    {
        const batch_t *oracles[4] = {&cs, &wires, &zs, &quot};
        const size_t counts[4] = {ncs, nw, nzs, nq};
        E_T *comp = malloc(n * sizeof(E_T));
        for (int batch = 0; batch < 2; batch++) {
            /* reduce_polys_base (util/reducing.rs:89-103): sum_j alpha^j * poly_j, powers restart at 1 */
            for (size_t t = 0; t < n; t++) comp[t] = E_FROM(0);
            E_T ap = E_FROM(1);
            size_t count = 0;
            for (int o = 0; o < 4; o++) {
                size_t lo = 0, hi = counts[o];
                if (batch == 1) { if (o != 2) continue; hi = c; } /* fri_zs_polys: zs_range of oracle 2 (circuit_data.rs:760-767) */
                for (size_t j = lo; j < hi; j++) {
                    const F_T *p = oracles[o]->coeffs + j * n;
                    for (size_t t = 0; t < n; t++) comp[t] = E_ADD(comp[t], E_SCALE(ap, p[t]));
                    ap = E_MUL(ap, fri_alpha);
                    count++;
                }
            }
            /* divide_by_linear (field/src/polynomial/division.rs:75-88) + push zero */
            E_T point = batch == 0 ? zeta : zeta_next;
            E_T *q = malloc(n * sizeof(E_T));
            E_T acc = E_FROM(0);
            for (size_t t = n; t-- > 0;) { acc = E_ADD(E_MUL(acc, point), comp[t]); if (t > 0) q[t - 1] = acc; }
            q[n - 1] = E_FROM(0);
            /* alpha.shift_poly(&mut final_poly); final_poly += quotient (oracle.rs:222-223) */
            E_T sh = E_POW(fri_alpha, count);
            for (size_t t = 0; t < n; t++) final_poly[t] = E_ADD(E_MUL(final_poly[t], sh), q[t]);
            free(q);
        }
        free(comp);
    }
    /* coset_fft in the extension field == base NTT on each coordinate (oracle.rs:226-231) */
    fc0 = malloc((size_t)D * N * sizeof(F_T)); fc1 = NULL;
    values = malloc(N * sizeof(E_T));
    E_T *coeffs = final_poly;
    size_t cur_len = N;
    unsigned cur_lg = lgN;

    GBO_SCOPE("prove_openings");
    /* ---- fri_committed_trees (fri/prover.rs:83-133) */
    if (ext_arity_set) {
        for (unsigned i = 0; i < ext_narity; i++) arity_bits_list[narity++] = ext_arity_bits[i];
    } else { /* ConstantArityBits (fri/reduction_strategies.rs:44-56) */
        unsigned db = lg;
        while (db > cfg->final_poly_bits && db + r >= capH + cfg->arity_bits) { arity_bits_list[narity++] = cfg->arity_bits; db -= cfg->arity_bits; }
    }
    tree_leaves = calloc(narity + 1, sizeof(F_T *)); tree_digests = calloc(narity + 1, sizeof(F_T *));
    tree_log = calloc(narity + 1, sizeof(unsigned));
    F_T shift = F_GENERATOR;
    {
        for (int k = 0; k < D; k++) {
            for (size_t t = 0; t < N; t++) fc0[(size_t)k * N + t] = coeffs[t].c[k];
            X_COSET_FFT(fc0 + (size_t)k * N, lgN, shift, 0);
            for (size_t t = 0; t < N; t++) values[t].c[k] = fc0[(size_t)k * N + t];
        }
    }
    fri_caps = malloc(narity * ((size_t)HOUT << capH) * sizeof(F_T) + 8);
    for (unsigned li = 0; li < narity; li++) {
        unsigned ab = arity_bits_list[li];
        size_t arity = (size_t)1 << ab, nleaves = cur_len >> ab, width = arity * D;
        F_T *lv = malloc(nleaves * width * sizeof(F_T));
        /* reverse_index_bits_in_place(values); chunk by arity; flatten */
        for (size_t i = 0; i < cur_len; i++) {
            size_t src = rev_bits_sz(i, cur_lg);
            for (int k = 0; k < D; k++) lv[i * D + k] = values[src].c[k];
        }
        F_T *dg = malloc((2 * (nleaves - ((size_t)1 << capH)) + 1) * HOUT * sizeof(F_T));
        F_T *cap = fri_caps + li * ((size_t)HOUT << capH);
        if ((rc = X_MERKLE_TREE(lv, cur_lg - ab, width, capH, dg, cap))) goto done;
        tree_leaves[li] = lv; tree_digests[li] = dg; tree_log[li] = cur_lg - ab;
        X_CH_OBSERVE(&ch, cap, (size_t)HOUT << capH);
        E_T beta = challenger_ext(&ch);
        /* fold: reduce_with_powers(chunk, beta) (plonk_common.rs:124-136) */
        size_t new_len = cur_len >> ab;
        for (size_t m = 0; m < new_len; m++) {
            E_T s = E_FROM(0);
            for (size_t t = arity; t-- > 0;) s = E_ADD(E_MUL(s, beta), coeffs[m * arity + t]);
            coeffs[m] = s;
        }
        cur_len = new_len; cur_lg -= ab;
        shift = F_POW(shift, arity);
        for (int k = 0; k < D; k++) {
            for (size_t t = 0; t < cur_len; t++) fc0[(size_t)k * cur_len + t] = coeffs[t].c[k];
            X_COSET_FFT(fc0 + (size_t)k * cur_len, cur_lg, shift, 0);
            for (size_t t = 0; t < cur_len; t++) values[t].c[k] = fc0[(size_t)k * cur_len + t];
        }
    }
    size_t final_len = cur_len >> r;
    for (size_t t = 0; t < final_len; t++) X_CH_OBSERVE(&ch, coeffs[t].c, D);

    GBO_SCOPE("fri_committed_trees");
    /* ---- fri_proof_of_work (fri/prover.rs:136-188): minimum nonce (== find_any with one thread) */
    F_T pow_witness = 0;
    {
        unsigned min_lz = cfg->pow_bits + (64 - F_ORDER_BITS); /* fri/prover.rs:147 */
        F_T st[SPONGE_W];
        memcpy(st, ch.state, sizeof st);
        for (int i = 0; i < ch.nin; i++) st[i] = ch.in[i];
        int pos = ch.nin;
        for (F_T cand = 0;; cand++) {
            F_T s2[SPONGE_W];
            memcpy(s2, st, sizeof s2);
            s2[pos] = cand;
            X_PERMUTE(s2, s2);
            F_T resp = s2[7]; /* squeeze().last(): rate 8 */
            unsigned lz = resp ? (unsigned)__builtin_clzll((unsigned long long)resp) : 64;
            if (lz >= min_lz) { pow_witness = cand; break; }
        }
        X_CH_OBSERVE(&ch, &pow_witness, 1);
        F_T resp = X_CH_GET(&ch);
        unsigned lz = resp ? (unsigned)__builtin_clzll((unsigned long long)resp) : 64;
        if (lz < min_lz) { rc = -20; goto done; }
        if (debug_out) debug_out[3 * c + 2 * D] = resp;
    }

    /* FRI proof bytes (serialization/mod.rs:1679-1695): caps, query rounds, final poly, pow witness */
    put(&ob, fri_caps, narity * ((size_t)HOUT << capH) * sizeof(F_T));
    {
        const batch_t *oracles[4] = {&cs, &wires, &zs, &quot};
        F_T sib[64 * HOUT];
        for (unsigned qi = 0; qi < cfg->num_queries; qi++) {
            size_t x_index = (size_t)((uint64_t)X_CH_GET(&ch) % N);
            for (int o = 0; o < 4; o++) {
                const batch_t *b = oracles[o];
                put(&ob, b->leaves + x_index * b->width, b->width * sizeof(F_T)); /* salted rows when hiding */
                int ns = X_MERKLE_PROVE(b->digests, lgN, capH, x_index, sib);
                put_u8(&ob, (uint8_t)ns);
                put(&ob, sib, (size_t)ns * HOUT * sizeof(F_T));
            }
            size_t xi = x_index;
            for (unsigned li = 0; li < narity; li++) {
                unsigned ab = arity_bits_list[li];
                size_t width = ((size_t)1 << ab) * D;
                size_t leaf = xi >> ab;
                put(&ob, tree_leaves[li] + leaf * width, width * sizeof(F_T));
                int ns = X_MERKLE_PROVE(tree_digests[li], tree_log[li], capH, leaf, sib);
                put_u8(&ob, (uint8_t)ns);
                put(&ob, sib, (size_t)ns * HOUT * sizeof(F_T));
                xi = leaf;
            }
        }
    }
    for (size_t t = 0; t < final_len; t++) put_ext(&ob, coeffs[t]);
    put_f(&ob, pow_witness);
    /* ProofWithPublicInputs (serialization/mod.rs:2134-2151) */
    put_u64(&ob, num_public_inputs);
    put(&ob, public_inputs, num_public_inputs * sizeof(F_T));
    *out_len = ob.len;
    if (ob.len > ob.cap) rc = -30;

    if (debug_out) {
        for (unsigned i = 0; i < c; i++) { debug_out[i] = betas[i]; debug_out[c + i] = gammas[i]; debug_out[2 * c + i] = alphas[i]; }
        for (int k = 0; k < D; k++) { debug_out[3 * c + k] = zeta.c[k]; debug_out[3 * c + D + k] = fri_alpha.c[k]; }
    }
    GBO_SCOPE("pow + queries + serialise");
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

int X_PROVE_DUMMY(const gbo_circuit_cfg *cfg, const F_T *constants_sigmas, const F_T *circuit_digest, const F_T *k_is,
                  const F_T *witness, const F_T *public_inputs, size_t num_public_inputs, uint8_t *out, size_t out_cap,
                  size_t *out_len, F_T *debug_out) {
    return X_PROVE_DUMMY_SALTED(cfg, constants_sigmas, circuit_digest, k_is, witness, public_inputs, num_public_inputs, out, out_cap,
                                out_len, debug_out, NULL);
}
