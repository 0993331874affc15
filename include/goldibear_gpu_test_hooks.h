/* goldibear_gpu_test_hooks.h - exports of libgoldibear_gpu.so that exist for its test suite only.  Not part of the product ABI
 * (include/goldibear_gpu.h): no binding of the reference calls them, the Rust shim does not declare them. */
#ifndef GOLDIBEAR_GPU_TEST_HOOKS_H
#define GOLDIBEAR_GPU_TEST_HOOKS_H

#include "goldibear_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Arm a one-shot fault: the next gb_prove* on `c` behaves as if the permutation argument had found a zero denominator
 * (plonk/prover.rs:512-514) once its Z computation is done - it returns GB_ERR_PERM_ARG_ZERO and keeps what gb_prove_retry
 * builds on, exactly as the real error does.  In a 64-bit field the real error has probability ~2^-37 per 2^20-row proof: this
 * is how the retry path of a Goldilocks circuit is exercised (tests/test_gpu_prove.py). */
gb_status gb_test_arm_perm_arg_failure(gb_circuit* c);

#ifdef __cplusplus
}
#endif
#endif /* GOLDIBEAR_GPU_TEST_HOOKS_H */
