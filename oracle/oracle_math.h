/* oracle_math.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * The small dense-math layer of the reference's device code (reference ba/matlib.cpp and
 * ba/bafuncs.cpp), as a C interface with two interchangeable implementations:
 *   oracle_math.c     our CPU restatement (travels with the repo, used on the GPU box)
 *   ref_adapter.cpp   thin wrappers that call the reference's own templates, compiled from
 *                     /root/reference where they lie, out of tree (make ref; this container only)
 * tests/test_oracle_math.py checks the two bit-for-bit.
 *
 * All matrices are row-major fp32.  Like the reference's matMul, products ACCUMULATE into the
 * output (matlib.cpp:54,64,74), so "out" arguments documented as pre-zeroed must be zero on entry
 * to get a plain product.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H
#ifdef __cplusplus
extern "C" {
#endif

/* P(pr x pc) += op(A) * op(B);  A is ar x ac, B is br x bc (matlib.cpp:47-89). */
void om_matmul(const float* A, int ar, int ac, const float* B, int br, int bc,
               float* P, int pc, int tA, int tB);
void om_inv3x3(const float* M, float* inv);              /* matlib.cpp:143-161 */
void om_inv6x6(const float* A, float* Ainv_zeroed);      /* matlib.cpp:180-222 */
void om_so3exp(const float* v, float* R_zeroed);         /* bafuncs.cpp:31-55  */
void om_inf2mean6x6(const float* eta6, const float* lambda36, float* mean6_zeroed);   /* bafuncs.cpp:2-7   */
void om_inf2mean3x3(const float* eta3, const float* lambda9, float* mean3_zeroed);    /* bafuncs.cpp:10-15 */
void om_hfunc(const float* cam6, const float* lmk3, const float* K9, float* hx2);   /* bafuncs.cpp:82-103 */
void om_jac(const float* cam6, const float* lmk3, const float* K9,
            float* Jkf12_zeroed, float* Jlmk6_zeroed);   /* bafuncs.cpp:106-213 */
const char* om_impl_name(void);
void om_set_trig_mode(int mode);   /* restatement only: 0 = host libm sinf/cosf, 1 = correctly rounded */

#ifdef __cplusplus
}
#endif
#endif
