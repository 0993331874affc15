/* TEST ORACLE - BabyBear instantiation of prover_impl.h (D = 4 with x^4 - 11, Poseidon2-16, H = 8).
 * PARITY UNPINNED for the field constants (see oracle_bb.c).  The extension non-residue W = 11 is recalled from
 * upstream Plonky3 (BinomiallyExtendable<4> for BabyBear); 11 is checked to be a non-residue of the required kind
 * in tests/test_oracle_bb.py. */
#include <stdint.h>
#include <stddef.h>

typedef uint32_t bb_t;
#define BB_P 2013265921u
#define BB_W 11u
static inline bb_t bb_add(bb_t a, bb_t b) { uint32_t s = a + b; return s >= BB_P ? s - BB_P : s; }
static inline bb_t bb_sub(bb_t a, bb_t b) { return a >= b ? a - b : a + BB_P - b; }
static inline bb_t bb_mul(bb_t a, bb_t b) { return (bb_t)(((uint64_t)a * b) % BB_P); }
static inline bb_t bb_pow(bb_t b, uint64_t e) { bb_t r = 1; while (e) { if (e & 1) r = bb_mul(r, b); b = bb_mul(b, b); e >>= 1; } return r; }
static inline bb_t bb_inv(bb_t a) { return bb_pow(a, BB_P - 2); }
static inline bb_t bb_two_adic_generator(unsigned bits) { bb_t g = 0x1a427a41u; for (unsigned i = bits; i < 27; i++) g = bb_mul(g, g); return g; }

typedef struct { bb_t c[4]; } bb4_t;
static inline bb4_t bb4_from(bb_t a) { bb4_t r = {{a, 0, 0, 0}}; return r; }
static inline bb4_t bb4_add(bb4_t a, bb4_t b) { bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_add(a.c[i], b.c[i]); return r; }
static inline bb4_t bb4_sub(bb4_t a, bb4_t b) { bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_sub(a.c[i], b.c[i]); return r; }
static inline bb4_t bb4_mul(bb4_t a, bb4_t b) {
    uint64_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t[i + j] = (t[i + j] + (uint64_t)a.c[i] * b.c[j]) % BB_P;
    bb4_t r;
    for (int k = 0; k < 4; k++) r.c[k] = (bb_t)((t[k] + (k < 3 ? (uint64_t)BB_W * t[k + 4] : 0)) % BB_P);
    return r;
}
static inline bb4_t bb4_scale(bb4_t a, bb_t s) { bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_mul(a.c[i], s); return r; }
static inline bb4_t bb4_pow(bb4_t b, uint64_t e) { bb4_t r = bb4_from(1); while (e) { if (e & 1) r = bb4_mul(r, b); b = bb4_mul(b, b); e >>= 1; } return r; }

#define HOUT 8
#define D 4
#define SPONGE_W 16
#define F_ORDER_BITS 31
#define F_T bb_t
#define E_T bb4_t
#define F_ADD bb_add
#define F_SUB bb_sub
#define F_MUL bb_mul
#define F_INV bb_inv
#define F_POW bb_pow
#define F_TWO_ADIC bb_two_adic_generator
#define F_GENERATOR 31u
#define F_FROM_U64(x) ((bb_t)((x) % BB_P))
#define E_FROM bb4_from
#define E_ADD bb4_add
#define E_SUB bb4_sub
#define E_MUL bb4_mul
#define E_SCALE bb4_scale
#define E_POW bb4_pow
#define X_HASH_NO_PAD gbo_bb_hash_no_pad
#define X_COMMIT gbo_bb_commit
#define X_MERKLE_TREE gbo_bb_merkle_tree
#define X_MERKLE_PROVE gbo_bb_merkle_prove
#define X_COSET_IFFT gbo_bb_coset_ifft
#define X_COSET_FFT gbo_bb_coset_fft
#define X_PERMUTE gbo_bb_poseidon2
#define X_CH_INIT gbo_bb_challenger_init
#define X_CH_OBSERVE gbo_bb_challenger_observe
#define X_CH_GET gbo_bb_challenger_get
#define X_PROVE_DUMMY gbo_bb_prove_dummy
#define X_PROVE_DUMMY_SALTED gbo_bb_prove_dummy_salted

/* from oracle_bb.c */
void gbo_bb_hash_no_pad(const bb_t *in, size_t n, bb_t out[HOUT]);
int gbo_bb_commit(const bb_t *cols, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height, int is_coeffs,
                  const bb_t *salts, bb_t *coeffs, bb_t *leaves, bb_t *digests, bb_t *cap);
int gbo_bb_merkle_tree(const bb_t *leaves, size_t log_l, size_t width, unsigned cap_height, bb_t *digests, bb_t *cap);
int gbo_bb_merkle_prove(const bb_t *digests, size_t log_l, unsigned cap_height, size_t leaf_index, bb_t *siblings);
void gbo_bb_coset_ifft(bb_t *v, unsigned lg_n, bb_t shift);
void gbo_bb_coset_fft(bb_t *v, unsigned lg_n, bb_t shift, unsigned zero_factor);
void gbo_bb_poseidon2(const bb_t in[16], bb_t out[16]);
typedef struct { bb_t state[16]; bb_t in[8]; int nin; bb_t out[8]; int nout; } challenger_t;
void gbo_bb_challenger_init(challenger_t *c);
void gbo_bb_challenger_observe(challenger_t *c, const bb_t *e, size_t n);
bb_t gbo_bb_challenger_get(challenger_t *c);

#define X_POSEIDON_GATE(w, out) do { (void)(w); for (unsigned q_ = 0; q_ < 123; q_++) (out)[q_] = 0; rc = -12; } while (0) /* Goldilocks gate */
void gbo_bb_poseidon2_gate_constraints(const uint32_t *w, unsigned num_ops, uint32_t *out);
#define X_POSEIDON2_GATE(w, nops, out) gbo_bb_poseidon2_gate_constraints(w, nops, out)
#include "prover_impl.h"
