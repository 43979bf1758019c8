/* gbp_mi355x_debug.h — TEST HOOKS of the GBP engine.  Not part of the product ABI: the symbols below exist only in
 * libgbp_mi355x_test.so (the product sources compiled with -DGBP_BUILD_TEST_HOOKS, `python -m gbp_poplar_amd.build`
 * builds it beside the product library) and are used by tests/ and profiles/ to look at internal state in the
 * reference's tensor layouts and to run the device math layer on caller-supplied vectors. */
#ifndef GBP_MI355X_DEBUG_H
#define GBP_MI355X_DEBUG_H

#include "gbp_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Raw internal state in the reference's tensor layouts (ba.cpp:665-687,759-775):
 *   what 0: a = factor_potentials_eta [9E],  b = factor_potentials_lambda [81E] = [cc36|cl18|lc18|ll9]
 *   what 1: a = cam message eta [6E],        b = cam message Lambda [36E] (lower triangle as stored; upper 0)
 *   what 2: a = lmk message eta [3E],        b = lmk message Lambda [9E]
 *   what 3: a = mu [9E],                     b = dmu [E]                                          */
int gbp_debug_get(gbp_ctx* ctx, int what, float* a, float* b);
/* Timing experiment: average us per launch of an ablated sweep kernel (1 = no landmark-message
 * gather/scatter, 2 = no landmark-belief gather, 4 = no arithmetic, 8 = streaming landmark messages;
 * bits combine).  Leaves garbage in the ctx. */
int gbp_debug_time_sweep(gbp_ctx* ctx, int ablation, int reps, double* avg_us);
/* Overwrite the factor potentials from reference-layout arrays (inverse of what 0). Test hook. */
int gbp_debug_set_factor_potentials(gbp_ctx* ctx, const float* eta9E, const float* lambda81E);

/* The device math layer on caller-supplied vectors, one GPU lane per vector (no ctx): lets a test compare the HIP
 * routines directly with outputs of the reference's own matlib.cpp / bafuncs.cpp.  in/out are [n][width] fp32:
 *   op 0 inv3x3 9 -> 9 (matlib.cpp:143-161)        op 1 inv6x6 36 -> 36 (matlib.cpp:180-222)
 *   op 2 so3exp 3 -> 9 (bafuncs.cpp:31-55)         op 3 hfunc + Jac: cam6 lmk3 K9 -> hx2 Jkf12 Jlmk6 (bafuncs.cpp:82-213)
 *   op 4 P(6x3) += B(6x6) A(6x3), op 5 P(3x6) += A^T B: A18 B36 P18 -> 18;  op 6 P(6x6) += A A^T: A18 P36 -> 36
 *        (matMul and its transpose modes, matlib.cpp:47-89)
 *   op 7 inf2mean6x6: eta6 Lambda36 -> 6;  op 8 inf2mean3x3: eta3 Lambda9 -> 3 (bafuncs.cpp:2-15)
 *   op 9 inv6x6 again, but in the SUB-WAVE mapping: 16 lanes cooperate on one matrix (operands in LDS, lane = output
 *        element, reference order): 36 -> 36, bit-identical to op 1; exists to be measured against it
 *   op 10 div_shared: x9 m1 -> 9 quotients x[i] / m — the shared-reciprocal division of gbp_device_math.hpp, which must equal
 *        the IEEE fp32 division bit for bit (tests/test_gpu_device_math.py)                                             */
int gbp_debug_math(int op, const float* in, float* out, int n);
/* same, then `reps` back-to-back launches timed with hipEvents: average microseconds per launch */
int gbp_debug_math_timed(int op, const float* in, float* out, int n, int reps, double* avg_us);

#ifdef __cplusplus
}
#endif
#endif /* GBP_MI355X_DEBUG_H */
