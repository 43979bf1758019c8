// gbp_device_math.hpp — register-resident small dense math for the GBP kernels (gfx950).
//
// Mapping: ONE LANE OWNS ONE FACTOR.  Every routine below is straight-line code over compile-time
// indexed arrays (fully unrolled -> VGPRs, no scratch), evaluating fp32 operations in exactly the
// order of the reference routine it cites, so that a sweep is bit-comparable with the CPU oracle
// when compiled with -ffp-contract=off (no FMA) and IEEE division / sqrt (hipcc default).
// Terms that are structurally zero in the reference's dense loops (triangular factors, hat
// matrices) are skipped: adding a +-0 product to a partial sum does not change an fp32 value
// (sign of an exact-zero result aside; non-finite inputs excepted).
#pragma once
#include <hip/hip_runtime.h>

namespace gbpdev {

#define GBP_UNROLL _Pragma("unroll")
#define GBP_DEV __device__ __forceinline__

// SLP fence: an empty asm the value passes through (no instruction).  hipcc's SLP vectoriser pairs independent fp32 adds /
// multiplies into v_pk_* instructions and prices the register shuffles that build the operand pairs at nothing; where the
// operands come from differently laid-out records (a packed triangle against a full matrix) it spends up to four v_mov on
// one saved add.  The fence ends the vectoriser's tree at this value, so the arithmetic that feeds it stays scalar
// (measured per site with the executed-instruction counters, profiles/r04_alu_diet.md).
#define GBP_SLP_FENCE(x) asm("" : "+v"(x))

// packed lower-triangle index, i >= j
GBP_DEV constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
GBP_DEV constexpr int trisym(int i, int j) { return i >= j ? tri(i, j) : tri(j, i); }

// v[i] / m for N values and ONE divisor, each quotient the correctly rounded IEEE fp32 quotient (bit for bit what
// `v[i] / m` gives), through one fp64 reciprocal: q = (float)((double)v * (1.0 / (double)m)).
// Why it is exact: the product carries a relative error <= 2^-52 (two fp64 roundings).  A quotient of two 24-bit
// significands is never a midpoint of two fp32 numbers and lies at least 2^-49 (relative) away from the nearest one
// (|X 2^(24+s) - (2k+1) M| >= 1 for integers X, M < 2^24), so the fp64 value and the exact quotient sit on the same
// side of every fp32 rounding boundary and round alike.  Zeros, infinities and NaNs behave like the division.  The
// bound needs a NORMAL fp32 result: a set that may hold a subnormal quotient (or one that underflows to zero from a
// non-zero numerator) is divided the slow way.  That is decided BEFORE the quotients, from the numerators alone:
//   |v| >= thr, thr >= FLT_MIN |m| (1 + 2^-10)   =>   |v / m| >= FLT_MIN (1 + 2^-10): a normal quotient
// with thr = (|m| (1 + 2^-9)) FLT_MIN rounded + 2 ulps (the first product rounds once, the scaling is exact unless it is
// subnormal, where it loses < 1 ulp: hence the 2).  "Some non-zero |v| < thr" is one integer comparison of the smallest
// u = 2 bits(|v|) - 1 (zero -> 0xffffffff; fp32 magnitudes order like their bit patterns) — v_lshl_add + half a v_min3
// per value instead of two compares and two scalar mask operations on every quotient; an infinite or NaN threshold
// (m infinite / NaN) compares as huge and takes the slow path, which is the division itself.  An fp32 IEEE division
// costs 11 instructions, five of them quarter-rate; this costs 3 full-rate ones per value + 1.5 for the test (measured
// in situ on fr1xyz: 29.9 -> 26.4 us per iteration for the 54 divisions of the Huber rescale alone,
// profiles/r03_small_graphs.md; the test: profiles/r04_alu_diet.md).
template <int N>
GBP_DEV void div_shared(const float (&v)[N], float m, float (&q)[N]) {
  const double r = 1.0 / (double)m;
  uint32_t least = 0xffffffffu;
  GBP_UNROLL
  for (int i = 0; i < N; ++i) {
    q[i] = (float)((double)v[i] * r);
    const uint32_t u = (__float_as_uint(v[i]) << 1) - 1u;
    least = u < least ? u : least;
  }
  const float thr = (__builtin_fabsf(m) * 1.001953125f) * 1.17549435e-38f;
  const bool redo = least < 2u * (__float_as_uint(thr) + 2u) - 1u;
  // (the slow path behind a WAVE-UNIFORM test, so that it is a branch no wavefront takes instead of nine divisions the
  // compiler might if-convert into every lane's instruction stream)
  if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
    if (redo) {
      GBP_UNROLL
      for (int i = 0; i < N; ++i) q[i] = v[i] / m;
    }
  }
}

// reference ba/matlib.cpp:143-161 — cofactor inverse, nine IEEE divisions by det (div_shared: same bits).
GBP_DEV void inv3x3(const float (&M)[9], float (&inv)[9]) {
  const float det = M[0] * (M[4] * M[8] - M[7] * M[5]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
                    M[2] * (M[3] * M[7] - M[4] * M[6]);
  const float cof[9] = {M[4] * M[8] - M[7] * M[5], M[2] * M[7] - M[1] * M[8], M[1] * M[5] - M[2] * M[4],
                        M[5] * M[6] - M[3] * M[8], M[0] * M[8] - M[2] * M[6], M[3] * M[2] - M[0] * M[5],
                        M[3] * M[7] - M[6] * M[4], M[6] * M[1] - M[0] * M[7], M[0] * M[4] - M[3] * M[1]};
  div_shared(cof, det, inv);
}

// reference ba/matlib.cpp:180-222 — un-pivoted LDL^T of the LOWER triangle (packed, 21 entries),
// D^-1, inverse of the unit upper factor (matlib.cpp:163-178), Ainv = (LT^-1 D^-1) LT^-T.
// All 36 entries of Ainv are produced separately: (a*d)*b and (b*d)*a round differently, so the
// reference's result is not bit-symmetric.
// The factorisation part: D^-1 and the inverse of the unit upper factor.  `A(i, j)`, i >= j, supplies the lower triangle
// (an array, or straight from LDS where registers are scarce).
template <class AF>
GBP_DEV void ldl6_lower(AF&& A, float (&rD)[6], float (&Ui)[6][6]) {
  float D[6];
  float U[6][6];   // U[j][i], i > j : LT(j,i)
  GBP_UNROLL
  for (int j = 0; j < 6; ++j) {
    float d = A(j, j);
    GBP_UNROLL
    for (int k = 0; k < j; ++k) d -= U[k][j] * U[k][j] * D[k];
    D[j] = d;
    rD[j] = 1 / d;
    GBP_UNROLL
    for (int i = j + 1; i < 6; ++i) {
      float u = rD[j] * A(i, j);
      GBP_UNROLL
      for (int k = 0; k < j; ++k) u -= rD[j] * U[k][i] * U[k][j] * D[k];
      U[j][i] = u;
    }
  }
  GBP_UNROLL
  for (int j = 1; j < 6; ++j) {
    GBP_UNROLL
    for (int i = 0; i < j; ++i) {
      float acc = 0.f;
      acc += U[i][j];  // k = i: LTinv(i,i) * LT(i,j) = 1 * LT(i,j)
      GBP_UNROLL
      for (int k = i + 1; k < j; ++k) acc += Ui[i][k] * U[k][j];
      Ui[i][j] = acc / -1.f;   // Ui[i][j], j > i : LTinv(i,j)
    }
  }
}
// entry (i, j) of Ainv = (LT^-1 D^-1) LT^-T from the factors
GBP_DEV float inv6_entry(const float (&rD)[6], const float (&Ui)[6][6], int i, int j) {
  const int k0 = i > j ? i : j;
  float acc = 0.f;
  GBP_UNROLL
  for (int k = 0; k < 6; ++k) {
    if (k < k0) continue;
    const float w = (k == i) ? rD[k] : Ui[i][k] * rD[k];  // (LTinv Dinv)(i,k); LTinv(i,i) = 1
    const float b = (k == j) ? 1.f : Ui[j][k];            // LTinv(j,k)
    acc += (k == j) ? w : w * b;
  }
  return acc;
}
GBP_DEV void inv6x6_lower(const float (&A)[21], float (&Ainv)[36]) {
  float rD[6], Ui[6][6];
  ldl6_lower([&](int i, int j) { return A[tri(i, j)]; }, rD, Ui);
  GBP_UNROLL
  for (int i = 0; i < 6; ++i) {
    GBP_UNROLL
    for (int j = 0; j < 6; ++j) {
      Ainv[i * 6 + j] = inv6_entry(rD, Ui, i, j);
    }
  }
}
// inf2mean6x6 (bafuncs.cpp:2-9) without materialising the inverse: x = A^-1 eta, every entry of A^-1 and every partial sum
// evaluated exactly as inv6x6_lower + the row-times-vector loop do (fewer live registers: used by the belief kernels).
template <class AF, class EF>
GBP_DEV void solve6_lower(AF&& A, EF&& eta, float (&x)[6]) {
  float rD[6], Ui[6][6];
  ldl6_lower(A, rD, Ui);
  GBP_UNROLL
  for (int i = 0; i < 6; ++i) {
    float acc = 0.f;
    GBP_UNROLL
    for (int k = 0; k < 6; ++k) acc += inv6_entry(rD, Ui, i, k) * eta(k);
    x[i] = acc;
  }
}

// NB on the two shortcuts above: 1*x and x*1 are exact, so multiplying by the unit diagonal is
// dropped; "0 + x" is kept (acc starts at 0.f) because 0 + (-0) = +0 in the reference too.

// reference ba/bafuncs.cpp:31-55 — Rodrigues, identity if theta <= 1e-6.
GBP_DEV void so3exp(const float (&v)[3], float (&R)[9]) {
  const float theta = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  GBP_UNROLL
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.f : 0.f;
  if (theta > 1e-6f) {
    // sin/cos are taken correctly rounded (fp64 evaluation rounded once to fp32): the reference's
    // own std::sin/std::cos resolve to whatever libm its target ships (Poplar's on the IPU), so no
    // libm is "the" reference; a correctly rounded value is the one every good libm approximates.
    double sd, cd;     // one argument reduction for both (the device library's sincos): the same fp64 values as sin() and cos()
    sincos((double)theta, &sd, &cd);
    const float s = (float)sd, c = (float)cd;
    const float H[9] = {0.f, -v[2], v[1], v[2], 0.f, -v[0], -v[1], v[0], 0.f};
    const float a = s / theta;
    const float b = (1 - c) / (theta * theta);
    GBP_UNROLL
    for (int i = 0; i < 3; ++i) {
      GBP_UNROLL
      for (int j = 0; j < 3; ++j) {
        float h2 = 0.f;
        GBP_UNROLL
        for (int k = 0; k < 3; ++k)
          if (k != i && k != j) h2 += H[i * 3 + k] * H[k * 3 + j];
        float r = R[i * 3 + j];
        if (i != j) r += a * H[i * 3 + j];
        r += b * h2;
        R[i * 3 + j] = r;
      }
    }
  }
}

// Shared front end of hfunc (bafuncs.cpp:82-103) and Jac (bafuncs.cpp:106-213): both build the
// same Tw2c and the same camera-frame point from the same inputs, so they are evaluated once.
struct Lin {
  float Jkf[12];  // 2x6
  float Jl[6];    // 2x3
  float hx[2];
};

// The part of Jac that depends on the CAMERA's linearisation point only (bafuncs.cpp:31-55, 166-199):
//   R = so3exp(w)                       (fp64-evaluated sin / cos: the long serial chain of a relinearisation)
//   N = (R^T - I) [w]x + w w^T          (the camera-only factor of dRy/dw = -R [y]x N / |w|^2)
//   d = |w|^2
// Every factor of a camera that relinearises in one sweep evaluates these with identical inputs — ~1 000 times per camera
// on the 1M-factor graph.  They are computed ONCE per camera where the hoisted mean is produced (k_beliefs camera part,
// k_persist camera role), with the very operations below, and stored as 20 floats per camera (5 float4: R 0-8, N 9-17,
// d 18, pad); relinearising lanes load them.  k_linearise and the per-factor-mu mode call cam_lin themselves: same bits.
struct CamLin {
  float R[9];
  float N[9];
  float den;
};
constexpr int kCamLin4 = 5;   // float4 per camera of the CAM_LIN array

GBP_DEV void cam_lin(const float (&v)[3], CamLin& c) {
  so3exp(v, c.R);
  const float vh[9] = {0.f, -v[2], v[1], v[2], 0.f, -v[0], -v[1], v[0], 0.f};
  float RtI[9];
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) {
    GBP_UNROLL
    for (int j = 0; j < 3; ++j) {
      float t = (i == j) ? -1.f : 0.f;
      t += c.R[j * 3 + i];
      RtI[i * 3 + j] = t;
    }
  }
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) {
    GBP_UNROLL
    for (int j = 0; j < 3; ++j) {
      float b = 0.f;
      GBP_UNROLL
      for (int k = 0; k < 3; ++k)
        if (k != j) b += RtI[i * 3 + k] * vh[k * 3 + j];  // hat matrices have a zero diagonal
      c.N[i * 3 + j] = b + v[i] * v[j];
    }
  }
  float den = 0.f;
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) den += v[i] * v[i];
  c.den = den;
}
GBP_DEV void cam_lin_pack(const CamLin& c, float4 (&q)[kCamLin4]) {
  q[0] = make_float4(c.R[0], c.R[1], c.R[2], c.R[3]);
  q[1] = make_float4(c.R[4], c.R[5], c.R[6], c.R[7]);
  q[2] = make_float4(c.R[8], c.N[0], c.N[1], c.N[2]);
  q[3] = make_float4(c.N[3], c.N[4], c.N[5], c.N[6]);
  q[4] = make_float4(c.N[7], c.N[8], c.den, 0.f);
}
GBP_DEV void cam_lin_unpack(const float4 (&q)[kCamLin4], CamLin& c) {
  c.R[0] = q[0].x; c.R[1] = q[0].y; c.R[2] = q[0].z; c.R[3] = q[0].w;
  c.R[4] = q[1].x; c.R[5] = q[1].y; c.R[6] = q[1].z; c.R[7] = q[1].w;
  c.R[8] = q[2].x; c.N[0] = q[2].y; c.N[1] = q[2].z; c.N[2] = q[2].w;
  c.N[3] = q[3].x; c.N[4] = q[3].y; c.N[5] = q[3].z; c.N[6] = q[3].w;
  c.N[7] = q[4].x; c.N[8] = q[4].y; c.den = q[4].z;
}

// hfunc + Jac of one factor from the camera-only terms `cl` of its camera (cl == cam_lin(cam[3..5]))
GBP_DEV void jac_hfunc_lin(const float (&cam)[6], const float (&lmk)[3], const float (&K)[9], const CamLin& cl, Lin& o) {
  const float (&R)[9] = cl.R;
  float yc[3];
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) {
    float acc = 0.f;
    GBP_UNROLL
    for (int k = 0; k < 3; ++k) acc += R[i * 3 + k] * lmk[k];
    acc += cam[i];  // T(i,3) * 1.0
    yc[i] = acc;
  }
  // six divisions by two divisors (bafuncs.cpp:96-99,150-160): yc[2] and yc[2]^2
  const float nz[4] = {yc[0], yc[1], K[0], K[4]};
  float qz[4];
  div_shared(nz, yc[2], qz);
  o.hx[0] = K[0] * qz[0] + K[2];
  o.hx[1] = K[4] * qz[1] + K[5];
  const float nzz[2] = {-(K[0] * yc[0]), -(K[4] * yc[1])};
  float qzz[2];
  div_shared(nzz, yc[2] * yc[2], qzz);
  const float jp00 = qz[2];
  const float jp02 = qzz[0];
  const float jp11 = qz[3];
  const float jp12 = qzz[1];
  // Jlmk = J_proj * R  (J_proj has structural zeros at (0,1) and (1,0))
  GBP_UNROLL
  for (int j = 0; j < 3; ++j) {
    float a0 = 0.f, a1 = 0.f;
    a0 += jp00 * R[j];
    a0 += jp02 * R[6 + j];
    a1 += jp11 * R[3 + j];
    a1 += jp12 * R[6 + j];
    o.Jl[j] = a0;
    o.Jl[3 + j] = a1;
  }
  o.Jkf[0] = jp00; o.Jkf[1] = 0.f;  o.Jkf[2] = jp02;
  o.Jkf[6] = 0.f;  o.Jkf[7] = jp11; o.Jkf[8] = jp12;

  // rotation block: dRy/dw = -R [y]x (w w^T + (R^T - I)[w]x) / |w|^2   (bafuncs.cpp:178-204); the bracket and |w|^2 are cl.N, cl.den
  const float yh[9] = {0.f, -lmk[2], lmk[1], lmk[2], 0.f, -lmk[0], -lmk[1], lmk[0], 0.f};
  float Ry[9], dR[9];
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) {
    GBP_UNROLL
    for (int j = 0; j < 3; ++j) {
      float a = 0.f;
      GBP_UNROLL
      for (int k = 0; k < 3; ++k)
        if (k != j) a += R[i * 3 + k] * yh[k * 3 + j];  // hat matrices have a zero diagonal
      Ry[i * 3 + j] = a;
    }
  }
  float ndR[9];
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) {
    GBP_UNROLL
    for (int j = 0; j < 3; ++j) {
      float a = 0.f;
      GBP_UNROLL
      for (int k = 0; k < 3; ++k) a += Ry[i * 3 + k] * cl.N[k * 3 + j];
      ndR[i * 3 + j] = -a;
    }
  }
  div_shared(ndR, cl.den, dR);
  GBP_UNROLL
  for (int j = 0; j < 3; ++j) {
    float a0 = 0.f, a1 = 0.f;
    a0 += jp00 * dR[j];
    a0 += jp02 * dR[6 + j];
    a1 += jp11 * dR[3 + j];
    a1 += jp12 * dR[6 + j];
    o.Jkf[3 + j] = a0;
    o.Jkf[9 + j] = a1;
  }
}

GBP_DEV void jac_hfunc(const float (&cam)[6], const float (&lmk)[3], const float (&K)[9], Lin& o) {
  CamLin cl;
  const float v[3] = {cam[3], cam[4], cam[5]};
  cam_lin(v, cl);
  jac_hfunc_lin(cam, lmk, K, cl, o);
}

}  // namespace gbpdev
