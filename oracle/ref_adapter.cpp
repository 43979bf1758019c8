// ref_adapter.cpp — TEST INFRASTRUCTURE ONLY, built in the survey/build container only.
//
// Implements the om_* interface of oracle_math.h by calling the REFERENCE's own templates,
// compiled from the sources where they lie (no copy is made into this repository; the objects go to
// $TMPDIR/gbp_oracle_ref, OUTSIDE the repository, so they never travel with a snapshot of it).  matlib.cpp and bafuncs.cpp are self-contained
// header-style C++ (they need only <cmath>), so this is g++ on the reference's own files —
// no stand-in headers.  gbp_codelets.cpp (the Poplar vertex classes) is NOT built: it needs
// <poplar/Vertex.hpp>, which the image lacks, so it is restated in oracle_gbp.c instead.
#include <cmath>
#include "/root/reference/ba/matlib.cpp"
#include "/root/reference/ba/bafuncs.cpp"
#include "oracle_math.h"

extern "C" {

const char* om_impl_name(void) { return "reference"; }
void om_set_trig_mode(int) {}  // the reference code calls std::sin/std::cos, nothing to select

void om_matmul(const float* A, int ar, int ac, const float* B, int br, int bc,
               float* P, int pc, int tA, int tB) {
  Mat<float> a(const_cast<float*>(A), ar, ac), b(const_cast<float*>(B), br, bc);
  unsigned pr = tA ? ac : ar;
  Mat<float> p(P, pr, pc);
  matMul(a, b, p, tA != 0, tB != 0);
}

void om_inv3x3(const float* M, float* inv) {
  Mat<float> m(const_cast<float*>(M), 3, 3), o(inv, 3, 3);
  inv3x3(m, o);
}

void om_inv6x6(const float* A, float* Ainv) {
  Mat<float> a(const_cast<float*>(A), 6, 6), o(Ainv, 6, 6);
  inv6x6(a, o);
}

void om_inf2mean6x6(const float* eta, const float* lambda, float* mean) {
  float S[36] = {0};
  Mat<float> e(const_cast<float*>(eta), 6, 1), l(const_cast<float*>(lambda), 6, 6), m(mean, 6, 1), s(S, 6, 6);
  inf2mean6x6(e, l, m, s);
}

void om_inf2mean3x3(const float* eta, const float* lambda, float* mean) {
  float S[9] = {0};
  Mat<float> e(const_cast<float*>(eta), 3, 1), l(const_cast<float*>(lambda), 3, 3), m(mean, 3, 1), s(S, 3, 3);
  inf2mean3x3(e, l, m, s);
}

void om_so3exp(const float* v, float* R) {
  Mat<float> vv(const_cast<float*>(v), 3, 1), r(R, 3, 3);
  so3exp(vv, r);
}

void om_hfunc(const float* cam, const float* lmk, const float* K, float* hx) {
  Mat<float> c(const_cast<float*>(cam), 6, 1), l(const_cast<float*>(lmk), 3, 1);
  Mat<float> k(const_cast<float*>(K), 3, 3), h(hx, 2, 1);
  hfunc(c, l, k, h);
}

void om_jac(const float* cam, const float* lmk, const float* K, float* Jkf, float* Jlmk) {
  Mat<float> c(const_cast<float*>(cam), 6, 1), l(const_cast<float*>(lmk), 3, 1);
  Mat<float> k(const_cast<float*>(K), 3, 3), jk(Jkf, 2, 6), jl(Jlmk, 2, 3);
  Jac(c, l, k, jk, jl);
}

}  // extern "C"
