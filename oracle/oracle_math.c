/* oracle_math.c — TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's dense-math layer.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this; the product
 * (gbp_poplar_amd/) never links or calls it.
 *
 * Every function follows the operation ORDER of the reference function it cites, because fp32
 * results are compared bit-for-bit (build with -ffp-contract=off).  Pinned against the reference's
 * own code by tests/test_oracle_math.py through oracle/_ref (ref_adapter.cpp).
 */
#include "oracle_math.h"
#include <math.h>

const char* om_impl_name(void) { return "restatement"; }

/* trig mode 0: sinf/cosf of the host libm (literal restatement of std::sin(float)).
 * trig mode 1: correctly rounded (fp64 evaluation rounded once) — what the HIP kernels use; the
 * reference's own target (IPU) has its own libm, so neither mode is more "reference" than the other. */
static int g_trig_mode = 0;
void om_set_trig_mode(int m) { g_trig_mode = m; }
static float o_sin(float x) { return g_trig_mode ? (float)sin((double)x) : sinf(x); }
static float o_cos(float x) { return g_trig_mode ? (float)cos((double)x) : cosf(x); }

/* reference ba/matlib.cpp:47-89 — accumulate, k innermost, the three transpose modes in use. */
void om_matmul(const float* A, int ar, int ac, const float* B, int br, int bc,
               float* P, int pc, int tA, int tB) {
  int i, j, k;
  if (!tA && !tB) {
    for (i = 0; i < ar; ++i)
      for (j = 0; j < bc; ++j)
        for (k = 0; k < ac; ++k) P[i * pc + j] += A[i * ac + k] * B[k * bc + j];
  } else if (tA && !tB) {
    for (i = 0; i < ac; ++i)
      for (j = 0; j < bc; ++j)
        for (k = 0; k < ar; ++k) P[i * pc + j] += A[k * ac + i] * B[k * bc + j];
  } else if (!tA && tB) {
    for (i = 0; i < ar; ++i)
      for (j = 0; j < br; ++j)
        for (k = 0; k < ac; ++k) P[i * pc + j] += A[i * ac + k] * B[j * bc + k];
  } else {
    for (i = 0; i < ac; ++i)
      for (j = 0; j < br; ++j)
        for (k = 0; k < ar; ++k) P[i * pc + j] += A[k * ac + i] * B[j * bc + k];
  }
}

/* reference ba/matlib.cpp:143-161 — cofactor inverse, nine divisions by det. */
void om_inv3x3(const float* M, float* inv) {
#define m(r, c) M[(r) * 3 + (c)]
  float det = m(0, 0) * (m(1, 1) * m(2, 2) - m(2, 1) * m(1, 2)) -
              m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
              m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
  inv[0] = (m(1, 1) * m(2, 2) - m(2, 1) * m(1, 2)) / det;
  inv[1] = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) / det;
  inv[2] = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) / det;
  inv[3] = (m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2)) / det;
  inv[4] = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) / det;
  inv[5] = (m(1, 0) * m(0, 2) - m(0, 0) * m(1, 2)) / det;
  inv[6] = (m(1, 0) * m(2, 1) - m(2, 0) * m(1, 1)) / det;
  inv[7] = (m(2, 0) * m(0, 1) - m(0, 0) * m(2, 1)) / det;
  inv[8] = (m(0, 0) * m(1, 1) - m(1, 0) * m(0, 1)) / det;
#undef m
}

/* reference ba/matlib.cpp:163-178 — inverse of an upper-triangular matrix into a zeroed output. */
static void inv_upper(const float* U, int n, float* Ui) {
  int i, j, k, q;
  for (j = 0; j < n; ++j) {
    Ui[j * n + j] = 1 / U[j * n + j];
    for (i = 0; i < j; ++i)
      for (k = 0; k < j; ++k) Ui[i * n + j] += Ui[i * n + k] * U[k * n + j];
    for (q = 0; q < j; ++q) Ui[q * n + j] /= -U[j * n + j];
  }
}

/* reference ba/matlib.cpp:180-222 — un-pivoted LDL^T on the LOWER triangle of A (line 201 reads
 * A(i,j), i>j), D^-1, LT^-1, then Ainv += (LT^-1 D^-1) LT^-T through two dense 6x6x6 products. */
void om_inv6x6(const float* A, float* Ainv) {
  float D[36] = {0}, LT[36] = {0}, LTi[36] = {0}, W[36] = {0};
  int i, j, k;
  for (j = 0; j < 6; ++j) {
    LT[j * 6 + j] = 1.0;
    D[j * 6 + j] = A[j * 6 + j];
    for (k = 0; k < j; ++k) D[j * 6 + j] -= LT[k * 6 + j] * LT[k * 6 + j] * D[k * 6 + k];
    for (i = j + 1; i < 6; ++i) {
      LT[j * 6 + i] = (1 / D[j * 6 + j]) * A[i * 6 + j];
      for (k = 0; k < j; ++k)
        LT[j * 6 + i] -= (1 / D[j * 6 + j]) * LT[k * 6 + i] * LT[k * 6 + j] * D[k * 6 + k];
    }
  }
  for (j = 0; j < 6; ++j) D[j * 6 + j] = 1 / D[j * 6 + j];
  inv_upper(LT, 6, LTi);
  om_matmul(LTi, 6, 6, D, 6, 6, W, 6, 0, 0);
  om_matmul(W, 6, 6, LTi, 6, 6, Ainv, 6, 0, 1);
}

/* reference ba/bafuncs.cpp:2-15: Sigma = Lambda^-1, mean += Sigma eta */
void om_inf2mean6x6(const float* eta, const float* lambda, float* mean) {
  float S[36] = {0};
  om_inv6x6(lambda, S);
  om_matmul(S, 6, 6, eta, 6, 1, mean, 1, 0, 0);
}
void om_inf2mean3x3(const float* eta, const float* lambda, float* mean) {
  float S[9] = {0};
  om_inv3x3(lambda, S);
  om_matmul(S, 3, 3, eta, 3, 1, mean, 1, 0, 0);
}

/* reference ba/bafuncs.cpp:19-28 */
static void hat3(const float* v, float* H /* zeroed */) {
  H[1] = -v[2]; H[2] = v[1];
  H[3] = v[2];  H[5] = -v[0];
  H[6] = -v[1]; H[7] = v[0];
}

/* reference ba/bafuncs.cpp:31-55 — Rodrigues; identity below 1e-6. */
void om_so3exp(const float* v, float* R) {
  float theta;
  R[0] = 1.f; R[4] = 1.f; R[8] = 1.f;
  theta = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (theta > 1e-6f) {
    float s = o_sin(theta), c = o_cos(theta);
    float H[9] = {0}, H2[9] = {0};
    int i;
    hat3(v, H);
    om_matmul(H, 3, 3, H, 3, 3, H2, 3, 0, 0);
    for (i = 0; i < 9; ++i) {
      R[i] += (s / theta) * H[i];
      R[i] += ((1 - c) / (theta * theta)) * H2[i];
    }
  }
}

/* reference ba/bafuncs.cpp:58-80 — 4x4 [R t; 0 0 0 0] (bottom-right is 0, line 61). */
static void w2c(const float* x, float* T /* zeroed 16 */) {
  float v[3], R[9] = {0};
  int i, j;
  T[15] = 0.f; T[3] = x[0]; T[7] = x[1]; T[11] = x[2];
  for (i = 0; i < 3; ++i) v[i] = x[i + 3];
  om_so3exp(v, R);
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
}

/* reference ba/bafuncs.cpp:82-103 */
void om_hfunc(const float* cam, const float* lmk, const float* K, float* hx) {
  float T[16] = {0}, yh[4], yc[4] = {0};
  w2c(cam, T);
  yh[0] = lmk[0]; yh[1] = lmk[1]; yh[2] = lmk[2]; yh[3] = 1.0;
  om_matmul(T, 4, 4, yh, 4, 1, yc, 1, 0, 0);
  hx[0] = K[0] * (yc[0] / yc[2]) + K[2];
  hx[1] = K[4] * (yc[1] / yc[2]) + K[5];
}

/* reference ba/bafuncs.cpp:106-213 */
void om_jac(const float* cam, const float* lmk, const float* K, float* Jkf, float* Jlmk) {
  float T[16] = {0}, R[9], yh[4], yc[4] = {0}, Jp[6] = {0};
  float v[3], dR[9] = {0}, Jrot[6] = {0};
  float vh[9] = {0}, yhat[9] = {0}, vv[9], RtI[9] = {0}, Ry[9] = {0}, num[9] = {0};
  float den;
  int i, j;
  w2c(cam, T);
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) R[i * 3 + j] = T[i * 4 + j];
  yh[0] = lmk[0]; yh[1] = lmk[1]; yh[2] = lmk[2]; yh[3] = 1.0;
  om_matmul(T, 4, 4, yh, 4, 1, yc, 1, 0, 0);

  Jp[0] = K[0] / yc[2];
  Jp[2] = -(K[0] * yc[0]) / (yc[2] * yc[2]);
  Jp[4] = K[4] / yc[2];
  Jp[5] = -(K[4] * yc[1]) / (yc[2] * yc[2]);

  om_matmul(Jp, 2, 3, R, 3, 3, Jlmk, 3, 0, 0);          /* landmark Jacobian */
  for (i = 0; i < 2; ++i)
    for (j = 0; j < 3; ++j) Jkf[i * 6 + j] = Jp[i * 3 + j];

  for (i = 0; i < 3; ++i) v[i] = cam[i + 3];
  hat3(v, vh);
  hat3(lmk, yhat);
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) vv[i * 3 + j] = v[i] * v[j];
  for (i = 0; i < 3; ++i) {
    RtI[i * 3 + i] = -1.f;
    for (j = 0; j < 3; ++j) RtI[i * 3 + j] += R[j * 3 + i];
  }
  om_matmul(R, 3, 3, yhat, 3, 3, Ry, 3, 0, 0);
  om_matmul(RtI, 3, 3, vh, 3, 3, num, 3, 0, 0);
  for (i = 0; i < 9; ++i) num[i] += vv[i];
  den = 0;
  for (i = 0; i < 3; ++i) den += v[i] * v[i];
  om_matmul(Ry, 3, 3, num, 3, 3, dR, 3, 0, 0);
  for (i = 0; i < 9; ++i) dR[i] = -dR[i] / den;
  om_matmul(Jp, 2, 3, dR, 3, 3, Jrot, 3, 0, 0);
  for (i = 0; i < 2; ++i)
    for (j = 0; j < 3; ++j) Jkf[i * 6 + j + 3] = Jrot[i * 3 + j];
}
