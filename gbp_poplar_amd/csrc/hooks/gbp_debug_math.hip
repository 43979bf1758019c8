// hooks/gbp_debug_math.hip — TEST HOOK (include/gbp_mi355x_debug.h: gbp_debug_math): the device math layer of
// gbp_device_math.hpp on caller-supplied vectors.  Included by gbp_kernels.hip inside namespace gbp when the library is built
// with -DGBP_BUILD_TEST_HOOKS (libgbp_mi355x_test.so); the product library does not contain it.
#ifdef GBP_BUILD_EXPERIMENTS
void lab_launch_inv6_coop(const float* in, float* out, int n, hipStream_t s);
#endif
// =================================================================================================
// k_debug_math: the device math layer (gbp_device_math.hpp) on caller-supplied vectors, one lane per
// vector — lets a test compare HIP directly with the reference's own matlib.cpp / bafuncs.cpp outputs
// (tests/golden/math_vectors.npz), without the restated vertex layer in between.  Test hook only.
//   op 0 inv3x3      in 9        out 9      matlib.cpp:143-161
//   op 1 inv6x6      in 36       out 36     matlib.cpp:180-222 (reads the lower triangle)
//   op 2 so3exp      in 3        out 9      bafuncs.cpp:31-55
//   op 3 hfunc+Jac   in 6+3+9    out 2+12+6 bafuncs.cpp:82-213
//   op 4 P(6x3) += B(6x6) A(6x3)    in A18 B36 P18  out 18    matMul, matlib.cpp:47-56
//   op 5 P(3x6) += A^T B            in A18 B36 P18  out 18    matMul transposeA, matlib.cpp:57-66
//   op 6 P(6x6) += A A^T            in A18 P36      out 36    matMul transposeB, matlib.cpp:67-76
//   op 7 inf2mean6x6 in eta6 L36  out 6 ; op 8 inf2mean3x3 in eta3 L9 out 3   bafuncs.cpp:2-15
//   op 10 div_shared  in x9 m1     out 9      (gbp_device_math.hpp: IEEE quotients through one fp64 reciprocal)
// =================================================================================================
__global__ __launch_bounds__(64) void k_debug_math(int op, const float* __restrict__ in, float* __restrict__ out, int n,
                                                   int in_w, int out_w) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n) return;
  const float* x = in + (size_t)t * in_w;
  float* y = out + (size_t)t * out_w;
  if (op == 0) {
    float M[9], I[9];
    GBP_UNROLL
    for (int i = 0; i < 9; ++i) M[i] = x[i];
    inv3x3(M, I);
    GBP_UNROLL
    for (int i = 0; i < 9; ++i) y[i] = I[i];
  } else if (op == 1) {
    float Al[21], I[36];
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) {
      GBP_UNROLL
      for (int j = 0; j <= i; ++j) Al[tri(i, j)] = x[i * 6 + j];
    }
    inv6x6_lower(Al, I);
    GBP_UNROLL
    for (int i = 0; i < 36; ++i) y[i] = I[i];
  } else if (op == 2) {
    const float v[3] = {x[0], x[1], x[2]};
    float R[9];
    so3exp(v, R);
    GBP_UNROLL
    for (int i = 0; i < 9; ++i) y[i] = R[i];
  } else if (op == 3) {
    float cam[6], lmk[3], K[9];
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) cam[i] = x[i];
    GBP_UNROLL
    for (int i = 0; i < 3; ++i) lmk[i] = x[6 + i];
    GBP_UNROLL
    for (int i = 0; i < 9; ++i) K[i] = x[9 + i];
    Lin L;
    jac_hfunc(cam, lmk, K, L);
    y[0] = L.hx[0]; y[1] = L.hx[1];
    GBP_UNROLL
    for (int i = 0; i < 12; ++i) y[2 + i] = L.Jkf[i];
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) y[14 + i] = L.Jl[i];
  } else if (op == 4) {   // the same loop shape as the message products of k_sweep (k sequential, acc starts at P)
    const float* A = x; const float* B = x + 18; const float* P0 = x + 54;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 3; ++j) {
        float acc = P0[i * 3 + j];
        for (int k = 0; k < 6; ++k) acc += B[i * 6 + k] * A[k * 3 + j];
        y[i * 3 + j] = acc;
      }
  } else if (op == 5) {
    const float* A = x; const float* B = x + 18; const float* P0 = x + 54;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 6; ++j) {
        float acc = P0[i * 6 + j];
        for (int k = 0; k < 6; ++k) acc += A[k * 3 + i] * B[k * 6 + j];
        y[i * 6 + j] = acc;
      }
  } else if (op == 6) {
    const float* A = x; const float* P0 = x + 18;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        float acc = P0[i * 6 + j];
        for (int k = 0; k < 3; ++k) acc += A[i * 3 + k] * A[j * 3 + k];
        y[i * 6 + j] = acc;
      }
  } else if (op == 7 || op == 8) {
    float cb[44], lb[16], x0c[6], x0l[3];
    GBP_UNROLL
    for (int i = 0; i < 44; ++i) cb[i] = 0.f;
    GBP_UNROLL
    for (int i = 0; i < 16; ++i) lb[i] = 0.f;
    if (op == 7) {
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) cb[i] = x[i];
      GBP_UNROLL
      for (int i = 0; i < 36; ++i) cb[8 + i] = x[6 + i];
      lb[4] = lb[8] = lb[12] = 1.f;
    } else {
      GBP_UNROLL
      for (int i = 0; i < 3; ++i) lb[i] = x[i];
      GBP_UNROLL
      for (int i = 0; i < 9; ++i) lb[4 + i] = x[3 + i];
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) cb[8 + i * 7] = 1.f;
    }
    belief_means(cb, lb, x0c, x0l);
    if (op == 7) { GBP_UNROLL for (int i = 0; i < 6; ++i) y[i] = x0c[i]; }
    else { GBP_UNROLL for (int i = 0; i < 3; ++i) y[i] = x0l[i]; }
  } else if (op == 10) {   // div_shared: 9 numerators, one divisor -> 9 quotients (must equal IEEE x / m bit for bit)
    float num[9], quo[9];
    GBP_UNROLL
    for (int i = 0; i < 9; ++i) num[i] = x[i];
    div_shared(num, x[9], quo);
    GBP_UNROLL
    for (int i = 0; i < 9; ++i) y[i] = quo[i];
  }
}


bool debug_math_widths(int op, int* in_w, int* out_w) {
  static const int iw[11] = {9, 36, 3, 18, 72, 72, 54, 42, 12, 36, 10}, ow[11] = {9, 36, 9, 20, 18, 18, 36, 6, 3, 36, 9};
  if (op < 0 || op > 10) return false;
  *in_w = iw[op]; *out_w = ow[op];
  return true;
}
void launch_debug_math(int op, const float* in, float* out, int n, hipStream_t s) {
  int in_w = 0, out_w = 0;
  if (!debug_math_widths(op, &in_w, &out_w) || n <= 0) return;
#ifdef GBP_BUILD_EXPERIMENTS
  if (op == 9) { lab_launch_inv6_coop(in, out, n, s); return; }    // 16 lanes per matrix (experiments/gbp_lab_kernels.hip)
#else
  if (op == 9) return;                                             // the sub-wave inverse exists in the experiments build only
#endif
  hipLaunchKernelGGL(k_debug_math, dim3((n + 63) / 64), dim3(64), 0, s, op, in, out, n, in_w, out_w);
}
