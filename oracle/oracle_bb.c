/* TEST ORACLE - CPU restatement of the reference's BabyBear hot path (Poseidon2 width 16, H = 8).
 *
 * Test infrastructure only.  PARITY PARTLY PINNED: the reference takes BabyBear from the un-vendored Plonky3 fork
 * (p3-baby-bear / p3-monty-31 / p3-poseidon2, branch goldilocks_improvements, no pinned rev) and holds no serialized
 * BabyBear proof.  Its one numeric BabyBear KAT (hash/poseidon2_risc0_babybear.rs:321-342, a width-24 Poseidon2) is
 * restated at the end of this file and matches: that pins the modulus, the canonical arithmetic, the x^7 s-box and the
 * Poseidon2 round order.  The generator 31, two_adic_generator(27) and the extension non-residue 11 stay UNPINNED
 * (recalled from upstream, self-consistency only).  What IS in the reference and followed here line by line:
 *   plonky2/src/hash/poseidon2_babybear.rs:18-67            round counts and constants
 *   plonky2/src/gates/poseidon2_babybear.rs:41-42,609-672   permutation order (asserted == p3 at :958-1004)
 *   plonky2/src/gates/poseidon2_babybear.rs:736-740,787-832,903-917  add_rc, M_I, M_E, apply_mat4
 *   plonky2/src/hash/hashing.rs:76-123, plonk/config.rs:70-84, hash/merkle_tree.rs:86-222, fri/oracle.rs:68-158
 * Recalled from upstream Plonky3 and only checked for self-consistency (tests/test_oracle_bb.py):
 *   p = 2^31 - 2^27 + 1, F::generator() = 31, two_adic_generator(27) = 0x1a427a41.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "poseidon_constants.h"

typedef uint32_t bb_t;
#define BB_P 2013265921u
#define BB_GENERATOR 31u
#define BB_TWO_ADIC_GEN_27 0x1a427a41u
#define BB_TWO_ADICITY 27
#define W 16
#define RATE 8
#define HOUT 8

/* selections as masks (the carries are coin flips on random data; a mispredicted branch costs more than the arithmetic) */
static inline bb_t bb_add(bb_t a, bb_t b) { uint32_t s = a + b; return s - (-(uint32_t)(s >= BB_P) & BB_P); }
static inline bb_t bb_sub(bb_t a, bb_t b) { return a - b + (-(uint32_t)(a < b) & BB_P); }
static inline bb_t bb_mul(bb_t a, bb_t b) { return (bb_t)(((uint64_t)a * b) % BB_P); }
static inline bb_t bb_pow(bb_t b, uint64_t e) { bb_t r = 1; while (e) { if (e & 1) r = bb_mul(r, b); b = bb_mul(b, b); e >>= 1; } return r; }
static inline bb_t bb_inv(bb_t a) { return bb_pow(a, BB_P - 2); }
static inline bb_t bb_two_adic_generator(unsigned bits) { bb_t g = BB_TWO_ADIC_GEN_27; for (unsigned i = bits; i < BB_TWO_ADICITY; i++) g = bb_mul(g, g); return g; }

static const uint32_t EXT_RC[8][16] = {BB_POSEIDON2_EXTERNAL_CONSTANTS_LIST};
static const uint32_t INT_RC[13] = {BB_POSEIDON2_INTERNAL_CONSTANTS_LIST};
static const unsigned DIAG_SHIFTS[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15}; /* gates/poseidon2_babybear.rs:41-42 */

static inline bb_t sbox7(bb_t x) { bb_t x2 = bb_mul(x, x), x4 = bb_mul(x2, x2), x3 = bb_mul(x, x2); return bb_mul(x3, x4); }

/* gates/poseidon2_babybear.rs:903-917 */
static void apply_mat4(bb_t x[4]) {
    bb_t t01 = bb_add(x[0], x[1]), t23 = bb_add(x[2], x[3]), t0123 = bb_add(t01, t23);
    bb_t t01123 = bb_add(t0123, x[1]), t01233 = bb_add(t0123, x[3]);
    bb_t n3 = bb_add(t01233, bb_add(x[0], x[0]));
    bb_t n1 = bb_add(t01123, bb_add(x[2], x[2]));
    bb_t n0 = bb_add(t01123, t01);
    bb_t n2 = bb_add(t01233, t23);
    x[0] = n0; x[1] = n1; x[2] = n2; x[3] = n3;
}
/* :804-832 */
static void permute_external(bb_t s[W]) {
    for (int i = 0; i < W; i += 4) apply_mat4(s + i);
    bb_t sums[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; k++) for (int j = 0; j < W; j += 4) sums[k] = bb_add(sums[k], s[j + k]);
    for (int i = 0; i < W; i++) s[i] = bb_add(s[i], sums[i % 4]);
}
/* :787-802 */
static void permute_internal(bb_t s[W]) {
    for (int i = 0; i < W; i++) s[i] = bb_mul(s[i], 943718400u);
    bb_t part = 0;
    for (int i = 1; i < W; i++) part = bb_add(part, s[i]);
    bb_t full = bb_add(part, s[0]);
    s[0] = bb_sub(part, s[0]);
    for (int i = 0; i < 15; i++) s[i + 1] = bb_add(full, bb_mul(s[i + 1], (bb_t)(1u << DIAG_SHIFTS[i])));
}
/* gates/poseidon2_babybear.rs:609-672 (the generator's run_once without the wire writes) */
void gbo_bb_poseidon2(const bb_t in[W], bb_t out[W]) {
    bb_t s[W];
    memcpy(s, in, sizeof s);
    permute_external(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < W; i++) s[i] = sbox7(bb_add(s[i], EXT_RC[r][i]));
        permute_external(s);
    }
    for (int r = 0; r < 13; r++) {
        s[0] = sbox7(bb_add(s[0], INT_RC[r]));
        permute_internal(s);
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < W; i++) s[i] = sbox7(bb_add(s[i], EXT_RC[r][i]));
        permute_external(s);
    }
    memcpy(out, s, sizeof s);
}

/* hash/hashing.rs:100-123, rate 8, 8 outputs */
/* gates/poseidon2_babybear.rs:315-413 Poseidon2BabyBearGate::eval_unfiltered_base_one: 150 constraints per operation, in the
 * reference's order.  w = the row's wires: per op 33 routed (16 inputs, 16 outputs, swap) for all ops first, then per op 133
 * non-routed (8 deltas; s-box inputs of full rounds 1..3, of the 13 internal rounds, of full rounds 4..7) (:56-147). */
void gbo_bb_poseidon2_gate_constraints(const bb_t *w, unsigned num_ops, bb_t *out) {
    unsigned t = 0;
    for (unsigned op = 0; op < num_ops; op++) {
        const unsigned in0 = 33 * op, out0 = in0 + W, start_delta = num_ops * 33 + op * 133;
        const unsigned start_full_0 = start_delta + 8, start_partial = start_full_0 + W * 3, start_full_1 = start_partial + 13;
        bb_t swap = w[in0 + 32], s[W];
        out[t++] = bb_mul(swap, bb_sub(swap, 1));
        for (int i = 0; i < 8; i++) out[t++] = bb_sub(bb_mul(swap, bb_sub(w[in0 + i + 8], w[in0 + i])), w[start_delta + i]);
        for (int i = 0; i < 8; i++) {
            s[i] = bb_add(w[in0 + i], w[start_delta + i]);
            s[i + 8] = bb_sub(w[in0 + i + 8], w[start_delta + i]);
        }
        permute_external(s);
        for (int r = 0; r < 4; r++) {
            for (int i = 0; i < W; i++) s[i] = bb_add(s[i], EXT_RC[r][i]);
            if (r > 0)
                for (int i = 0; i < W; i++) {
                    bb_t in = w[start_full_0 + W * (r - 1) + i];
                    out[t++] = bb_sub(s[i], in);
                    s[i] = in;
                }
            for (int i = 0; i < W; i++) s[i] = sbox7(s[i]);
            permute_external(s);
        }
        for (int r = 0; r < 13; r++) {
            s[0] = bb_add(s[0], INT_RC[r]);
            bb_t in = w[start_partial + r];
            out[t++] = bb_sub(s[0], in);
            s[0] = sbox7(in);
            permute_internal(s);
        }
        for (int r = 4; r < 8; r++) {
            for (int i = 0; i < W; i++) s[i] = bb_add(s[i], EXT_RC[r][i]);
            for (int i = 0; i < W; i++) {
                bb_t in = w[start_full_1 + W * (r - 4) + i];
                out[t++] = bb_sub(s[i], in);
                s[i] = in;
            }
            for (int i = 0; i < W; i++) s[i] = sbox7(s[i]);
            permute_external(s);
        }
        for (int i = 0; i < W; i++) out[t++] = bb_sub(s[i], w[out0 + i]);
    }
}

void gbo_bb_hash_no_pad(const bb_t *in, size_t n, bb_t out[HOUT]) {
    bb_t st[W] = {0};
    for (size_t off = 0; off < n; off += RATE) {
        size_t k = n - off < RATE ? n - off : RATE;
        memcpy(st, in + off, k * sizeof(bb_t));
        gbo_bb_poseidon2(st, st);
    }
    memcpy(out, st, HOUT * sizeof(bb_t));
}
/* plonk/config.rs:70-84: <= NUM_HASH_OUT_ELTS (8) elements are zero padded */
void gbo_bb_hash_or_noop(const bb_t *in, size_t n, bb_t out[HOUT]) {
    if (n <= HOUT) { memset(out, 0, HOUT * sizeof(bb_t)); memcpy(out, in, n * sizeof(bb_t)); }
    else gbo_bb_hash_no_pad(in, n, out);
}
/* hash/hashing.rs:76-96 (poseidon2_babybear.rs two_to_one) */
void gbo_bb_two_to_one(const bb_t l[HOUT], const bb_t r[HOUT], bb_t out[HOUT]) {
    bb_t st[W];
    memcpy(st, l, HOUT * sizeof(bb_t));
    memcpy(st + HOUT, r, HOUT * sizeof(bb_t));
    gbo_bb_poseidon2(st, st);
    memcpy(out, st, HOUT * sizeof(bb_t));
}

static inline size_t rev_bits(size_t x, unsigned bits) { size_t r = 0; for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }

/* field/src/fft.rs:168-205 with the root table of :12-31 folded in (rows computed on the fly) */
static void fft_classic(bb_t *v, unsigned lg_n, unsigned r) {
    size_t n = (size_t)1 << lg_n;
    for (size_t i = 0; i < n; i++) { size_t j = rev_bits(i, lg_n); if (i < j) { bb_t t = v[i]; v[i] = v[j]; v[j] = t; } }
    if (r > 0) { size_t mask = ~(((size_t)1 << r) - 1); for (size_t i = 0; i < n; i++) v[i] = v[i & mask]; }
    for (unsigned lg_half = r; lg_half < lg_n; lg_half++) {
        size_t half = (size_t)1 << lg_half, m = half * 2;
        bb_t base = bb_two_adic_generator(lg_half + 1);
        bb_t *om = malloc(half * sizeof(bb_t));
        bb_t x = 1;
        for (size_t j = 0; j < half; j++) { om[j] = x; x = bb_mul(x, base); }
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                bb_t t = bb_mul(om[j], v[k + half + j]), u = v[k + j];
                v[k + j] = bb_add(u, t);
                v[k + half + j] = bb_sub(u, t);
            }
        free(om);
    }
}
void gbo_bb_fft(bb_t *v, unsigned lg_n, unsigned zero_factor) { if (lg_n) fft_classic(v, lg_n, zero_factor); }
/* field/src/fft.rs:70-94 */
void gbo_bb_ifft(bb_t *v, unsigned lg_n) {
    size_t n = (size_t)1 << lg_n;
    bb_t n_inv = bb_pow(bb_inv(2), lg_n);
    gbo_bb_fft(v, lg_n, 0);
    v[0] = bb_mul(v[0], n_inv);
    if (n > 1) v[n / 2] = bb_mul(v[n / 2], n_inv);
    for (size_t i = 1; i < n / 2; i++) { size_t j = n - i; bb_t ci = bb_mul(v[j], n_inv), cj = bb_mul(v[i], n_inv); v[i] = ci; v[j] = cj; }
}
/* field/src/polynomial/mod.rs:282-295 */
void gbo_bb_coset_fft(bb_t *v, unsigned lg_n, bb_t shift, unsigned zero_factor) {
    size_t n = (size_t)1 << lg_n;
    bb_t p = 1;
    for (size_t i = 0; i < n; i++) { v[i] = bb_mul(v[i], p); p = bb_mul(p, shift); }
    gbo_bb_fft(v, lg_n, zero_factor);
}

/* hash/merkle_tree.rs:86-113 */
static void fill_subtree(bb_t *digests, size_t digests_len, const bb_t *leaves, size_t nleaves, size_t width, bb_t out[HOUT]) {
    if (digests_len == 0) { gbo_bb_hash_or_noop(leaves, width, out); return; }
    size_t half = digests_len / 2;
    bb_t l[HOUT], r[HOUT];
    if (nleaves >= 512) {
#pragma omp task shared(l)
        fill_subtree(digests, half - 1, leaves, nleaves / 2, width, l);
#pragma omp task shared(r)
        fill_subtree(digests + (half + 1) * HOUT, half - 1, leaves + (nleaves / 2) * width, nleaves / 2, width, r);
#pragma omp taskwait
    } else {
        fill_subtree(digests, half - 1, leaves, nleaves / 2, width, l);
        fill_subtree(digests + (half + 1) * HOUT, half - 1, leaves + (nleaves / 2) * width, nleaves / 2, width, r);
    }
    memcpy(digests + (half - 1) * HOUT, l, sizeof l);
    memcpy(digests + half * HOUT, r, sizeof r);
    gbo_bb_two_to_one(l, r, out);
}
/* hash/merkle_tree.rs:152-181 */
int gbo_bb_merkle_tree(const bb_t *leaves, size_t log_l, size_t width, unsigned cap_height, bb_t *digests, bb_t *cap) {
    if (cap_height > log_l) return -1;
    size_t L = (size_t)1 << log_l, ncap = (size_t)1 << cap_height, nd = 2 * (L - ncap);
    if (nd == 0) { for (size_t i = 0; i < L; i++) gbo_bb_hash_or_noop(leaves + i * width, width, cap + i * HOUT); return 0; }
    size_t sub_d = nd >> cap_height, sub_l = L >> cap_height;
#pragma omp parallel
#pragma omp single
    for (size_t s = 0; s < ncap; s++) {
#pragma omp task
        fill_subtree(digests + s * sub_d * HOUT, sub_d, leaves + s * sub_l * width, sub_l, width, cap + s * HOUT);
    }
    return 0;
}
/* hash/merkle_tree.rs:188-222 */
int gbo_bb_merkle_prove(const bb_t *digests, size_t log_l, unsigned cap_height, size_t leaf_index, bb_t *siblings) {
    size_t L = (size_t)1 << log_l;
    unsigned layers = (unsigned)log_l - cap_height;
    size_t nd = 2 * (L - ((size_t)1 << cap_height)), tree_len = nd >> cap_height;
    const bb_t *tree = digests + tree_len * (leaf_index >> layers) * HOUT;
    size_t pair = leaf_index & (((size_t)1 << layers) - 1);
    for (unsigned i = 0; i < layers; i++) {
        size_t parity = pair & 1;
        pair >>= 1;
        size_t sidx = 2 * ((pair << (i + 1)) + ((size_t)1 << i) - 1) + (1 - parity);
        memcpy(siblings + i * HOUT, tree + sidx * HOUT, HOUT * sizeof(bb_t));
    }
    return (int)layers;
}
/* hash/merkle_proofs.rs:54-76 */
int gbo_bb_merkle_verify(const bb_t *leaf, size_t width, size_t leaf_index, const bb_t *cap, const bb_t *siblings, unsigned nsib) {
    bb_t cur[HOUT], nxt[HOUT];
    gbo_bb_hash_or_noop(leaf, width, cur);
    size_t idx = leaf_index;
    for (unsigned i = 0; i < nsib; i++) {
        const bb_t *sib = siblings + i * HOUT;
        if (idx & 1) gbo_bb_two_to_one(sib, cur, nxt); else gbo_bb_two_to_one(cur, sib, nxt);
        memcpy(cur, nxt, sizeof cur);
        idx >>= 1;
    }
    return memcmp(cur, cap + idx * HOUT, sizeof cur) == 0;
}

/* fri/oracle.rs:68-123 (see gbo_gl_commit) */
int gbo_bb_commit(const bb_t *cols, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height, int is_coeffs,
                  const bb_t *salts, bb_t *coeffs, bb_t *leaves, bb_t *digests, bb_t *cap) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    unsigned log_N = log_n + rate_bits;
    size_t nsalt = salts ? 4 : 0, width = ncols + nsalt;
    if (cap_height > log_N) return -1;
    memcpy(coeffs, cols, ncols * n * sizeof(bb_t));
    if (!is_coeffs) {
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t c = 0; c < ncols; c++) gbo_bb_ifft(coeffs + c * n, log_n);
    }
    bb_t *lde = malloc(ncols ? ncols * N * sizeof(bb_t) : 1);
    if (!lde) return -2;
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t c = 0; c < ncols; c++) {
        bb_t *v = lde + c * N;
        memcpy(v, coeffs + c * n, n * sizeof(bb_t));
        memset(v + n, 0, (N - n) * sizeof(bb_t));
        gbo_bb_coset_fft(v, log_N, BB_GENERATOR, rate_bits);
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; i++) {
        size_t src = rev_bits(i, log_N);
        bb_t *row = leaves + i * width;
        for (size_t c = 0; c < ncols; c++) row[c] = lde[c * N + src];
        for (size_t s = 0; s < nsalt; s++) row[ncols + s] = salts[s * N + src];
    }
    free(lde);
    return gbo_bb_merkle_tree(leaves, log_N, width, cap_height, digests, cap);
}

uint32_t gbo_bb_mul(uint32_t a, uint32_t b) { return bb_mul(a, b); }
uint32_t gbo_bb_powu(uint32_t a, uint64_t e) { return bb_pow(a, e); }
uint32_t gbo_bb_two_adic_generator(unsigned bits) { return bb_two_adic_generator(bits); }

/* ------------------------------------------------------------------ Challenger (iop/challenger.rs:18-150), width 16, rate 8 */
typedef struct { bb_t state[W]; bb_t in[RATE]; int nin; bb_t out[RATE]; int nout; } gbo_bb_challenger;
void gbo_bb_challenger_init(gbo_bb_challenger *c) { memset(c, 0, sizeof *c); }
static void bb_duplexing(gbo_bb_challenger *c) {
    for (int i = 0; i < c->nin; i++) c->state[i] = c->in[i];
    c->nin = 0;
    gbo_bb_poseidon2(c->state, c->state);
    memcpy(c->out, c->state, RATE * sizeof(bb_t));
    c->nout = RATE;
}
void gbo_bb_challenger_observe(gbo_bb_challenger *c, const bb_t *e, size_t n) {
    for (size_t i = 0; i < n; i++) {
        c->nout = 0;
        c->in[c->nin++] = e[i];
        if (c->nin == RATE) bb_duplexing(c);
    }
}
bb_t gbo_bb_challenger_get(gbo_bb_challenger *c) {
    if (c->nin != 0 || c->nout == 0) bb_duplexing(c);
    return c->out[--c->nout];
}
/* field/src/polynomial/mod.rs:62-72 */
void gbo_bb_coset_ifft(bb_t *v, unsigned lg_n, bb_t shift) {
    size_t n = (size_t)1 << lg_n;
    gbo_bb_ifft(v, lg_n);
    bb_t si = bb_inv(shift), p = 1;
    for (size_t i = 0; i < n; i++) { v[i] = bb_mul(v[i], p); p = bb_mul(p, si); }
}
void gbo_bb_powers(bb_t base, size_t n, bb_t *out) { bb_t x = 1; for (size_t i = 0; i < n; i++) { out[i] = x; x = bb_mul(x, base); } }
void gbo_bb_scale_vec(const bb_t *a, bb_t k, size_t n, bb_t *out) { for (size_t i = 0; i < n; i++) out[i] = bb_mul(a[i], k); }

/* ------------------------------------------------------------------ Poseidon2-24 with the RISC0 parameters.
 * NOT on the hot path.  Restated because it is the one BabyBear computation the reference pins with numbers
 * (hash/poseidon2_risc0_babybear.rs:321-342): matching that KAT pins, from the reference's own data, the modulus
 * p = 2^31 - 2^27 + 1, the x^7 s-box and the generic Poseidon2 round order the width-16 permutation above shares
 * (p3_poseidon2::Poseidon2::permute_mut drives both; restated in gates/poseidon2_risc0_babybear.rs:552-600).
 * Layers: gates/poseidon2_risc0_babybear.rs:678-682 (add_rc), :731-736 (internal: sum + diag_i x_i),
 * :738-766 (external: apply_hl_mat4 per block + column-class sums), :841-857 (apply_hl_mat4). */
static const uint32_t R0_EXT_RC[8][24] = {BB_R0_EXTERNAL_CONSTANTS_LIST};
static const uint32_t R0_INT_RC[21] = {BB_R0_INTERNAL_CONSTANTS_LIST};
static const uint32_t R0_DIAG[24] = {BB_R0_M_INT_DIAG_HZN_LIST};

static void r0_apply_hl_mat4(bb_t x[4]) {
    bb_t t0 = bb_add(x[0], x[1]), t1 = bb_add(x[2], x[3]);
    bb_t t2 = bb_add(bb_add(x[1], x[1]), t1), t3 = bb_add(bb_add(x[3], x[3]), t0);
    bb_t t1_4 = bb_add(bb_add(t1, t1), bb_add(t1, t1)), t0_4 = bb_add(bb_add(t0, t0), bb_add(t0, t0));
    bb_t t4 = bb_add(t1_4, t3), t5 = bb_add(t0_4, t2);
    bb_t t6 = bb_add(t3, t5), t7 = bb_add(t2, t4);
    x[0] = t6; x[1] = t5; x[2] = t7; x[3] = t4;
}
static void r0_external(bb_t s[24]) {
    for (int i = 0; i < 24; i += 4) r0_apply_hl_mat4(s + i);
    bb_t sums[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; k++) for (int j = 0; j < 24; j += 4) sums[k] = bb_add(sums[k], s[j + k]);
    for (int i = 0; i < 24; i++) s[i] = bb_add(s[i], sums[i % 4]);
}
static void r0_internal(bb_t s[24]) {
    bb_t sum = 0;
    for (int i = 0; i < 24; i++) sum = bb_add(sum, s[i]);
    for (int i = 0; i < 24; i++) s[i] = bb_add(sum, bb_mul(R0_DIAG[i], s[i]));
}
void gbo_bb_poseidon2_r0(const bb_t in[24], bb_t out[24]) {
    bb_t s[24];
    memcpy(s, in, sizeof s);
    r0_external(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 24; i++) s[i] = sbox7(bb_add(s[i], R0_EXT_RC[r][i]));
        r0_external(s);
    }
    for (int r = 0; r < 21; r++) {
        s[0] = sbox7(bb_add(s[0], R0_INT_RC[r]));
        r0_internal(s);
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 24; i++) s[i] = sbox7(bb_add(s[i], R0_EXT_RC[r][i]));
        r0_external(s);
    }
    memcpy(out, s, sizeof s);
}
