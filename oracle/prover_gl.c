/* TEST ORACLE - Goldilocks instantiation of prover_impl.h (D = 2, Poseidon-12, H = 4). */
#include "gl.h"

#define HOUT 4
#define D 2
#define SPONGE_W 12
#define F_ORDER_BITS 64
#define F_T gl_t
#define E_T gl2_t
#define F_ADD gl_add
#define F_SUB gl_sub
#define F_MUL gl_mul
#define F_INV gl_inv
#define F_POW gl_pow
#define F_TWO_ADIC gl_two_adic_generator
#define F_GENERATOR GL_GENERATOR
#define F_FROM_U64(x) ((gl_t)((x) % GL_P))
#define E_FROM gl2_from
#define E_ADD gl2_add
#define E_SUB gl2_sub
#define E_MUL gl2_mul
#define E_SCALE gl2_scale
#define E_POW gl2_pow
#define X_HASH_NO_PAD gbo_gl_hash_no_pad
#define X_COMMIT gbo_gl_commit
#define X_MERKLE_TREE gbo_gl_merkle_tree
#define X_MERKLE_PROVE gbo_gl_merkle_prove
#define X_COSET_IFFT gbo_gl_coset_ifft
#define X_COSET_FFT gbo_gl_coset_fft
#define X_PERMUTE gbo_gl_poseidon
#define X_CH_INIT gbo_gl_challenger_init
#define X_CH_OBSERVE gbo_gl_challenger_observe
#define X_CH_GET gbo_gl_challenger_get
#define X_PROVE_DUMMY gbo_gl_prove_dummy
#define X_PROVE_DUMMY_SALTED gbo_gl_prove_dummy_salted

/* from oracle_gl.c */
void gbo_gl_hash_no_pad(const gl_t *in, size_t n, gl_t out[HOUT]);
int gbo_gl_commit(const gl_t *cols, size_t ncols, unsigned log_n, unsigned rate_bits, unsigned cap_height, int is_coeffs,
                  const gl_t *salts, gl_t *coeffs, gl_t *leaves, gl_t *digests, gl_t *cap);
int gbo_gl_merkle_tree(const gl_t *leaves, size_t log_l, size_t width, unsigned cap_height, gl_t *digests, gl_t *cap);
int gbo_gl_merkle_prove(const gl_t *digests, size_t log_l, unsigned cap_height, size_t leaf_index, gl_t *siblings);
void gbo_gl_coset_ifft(gl_t *v, unsigned lg_n, gl_t shift);
void gbo_gl_coset_fft(gl_t *v, unsigned lg_n, gl_t shift, unsigned zero_factor);
void gbo_gl_poseidon(const gl_t in[12], gl_t out[12]);
void gbo_gl_poseidon_gate_constraints(const gl_t *w, gl_t *out);
typedef struct { gl_t state[12]; gl_t in[8]; int nin; gl_t out[8]; int nout; } challenger_t;
void gbo_gl_challenger_init(challenger_t *c);
void gbo_gl_challenger_observe(challenger_t *c, const gl_t *e, size_t n);
gl_t gbo_gl_challenger_get(challenger_t *c);

#define X_POSEIDON_GATE(w, out) gbo_gl_poseidon_gate_constraints(w, out)
#define X_POSEIDON2_GATE(w, nops, out) do { (void)(w); (void)(nops); (out)[0] = 0; rc = -12; } while (0) /* BabyBear gate */
#include "prover_impl.h"
