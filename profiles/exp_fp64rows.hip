// exp_fp64rows.hip — does the time of an fp64 / conversion instruction depend on WHICH lanes are active?
// One wavefront per SIMD-ish launch (64 blocks x 64 threads), a dependent chain of the three instructions of
// div_shared (v_cvt_f64_f32, v_mul_f64, v_cvt_f32_f64) and, for comparison, of v_fma_f32 / v_fma_f64 /
// v_rcp_f32, executed under different EXEC masks.  ns per instruction of one wave.
//   hipcc -O3 --offload-arch=gfx950 -o profiles/_bin/exp_fp64rows profiles/exp_fp64rows.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ void k_chain(float* out, unsigned long long mask, int n, float x0 = 1.0f) {
  const unsigned lane = threadIdx.x & 63;
  float x = x0 * (1.0f + lane * 1e-3f);
  double d = (double)x0 * (1.0 + lane * 1e-3);
  if ((mask >> lane) & 1ull) {
    for (int i = 0; i < n; ++i) {
      if (OP == 0) {          // div_shared's body: cvt, mul, cvt
#pragma unroll
        for (int k = 0; k < 16; ++k) x = (float)((double)x * 1.0000001);
      } else if (OP == 1) {   // fp32 fma
#pragma unroll
        for (int k = 0; k < 16; ++k) { x = __builtin_fmaf(x, 1.0000001f, 1e-9f); x = __builtin_fmaf(x, 0.9999999f, 1e-9f); x = __builtin_fmaf(x, 1.0000001f, -1e-9f); }
      } else if (OP == 2) {   // fp64 fma
#pragma unroll
        for (int k = 0; k < 16; ++k) { d = __builtin_fma(d, 1.0000001, 1e-9); d = __builtin_fma(d, 0.9999999, 1e-9); d = __builtin_fma(d, 1.0000001, -1e-9); }
      } else {                // quarter-rate transcendental
#pragma unroll
        for (int k = 0; k < 16; ++k) { x = __builtin_amdgcn_rcpf(x); x = __builtin_amdgcn_rcpf(x); x = __builtin_amdgcn_rcpf(x); }
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + (float)d;
}

int main() {
  float* out;
  CK(hipMalloc(&out, 64 * 512 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct { const char* name; unsigned long long m; } masks[] = {
      {"1 lane (row 0)", 1ull}, {"5 lanes in row 0", 0x1111ull | 0x8000ull}, {"16 lanes = row 0", 0xffffull},
      {"1 lane in each of 2 rows", 1ull | (1ull << 16)}, {"1 lane in each of 4 rows", 1ull | (1ull << 16) | (1ull << 32) | (1ull << 48)},
      {"5 lanes over 4 rows", 1ull | (1ull << 20) | (1ull << 21) | (1ull << 40) | (1ull << 63)}, {"all 64 lanes", ~0ull}};
  const char* ops[] = {"cvt_f64_f32 + mul_f64 + cvt_f32_f64", "3 x fma_f32", "3 x fma_f64", "3 x rcp_f32"};
  printf("| active lanes | %s | %s | %s | %s |\n|---|---|---|---|---|\n", ops[0], ops[1], ops[2], ops[3]);
  const int n = 2000;
  for (auto& mk : masks) {
    printf("| %s |", mk.name);
    for (int op = 0; op < 4; ++op) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, nullptr));
        if (op == 0) hipLaunchKernelGGL(k_chain<0>, dim3(64), dim3(64), 0, nullptr, out, mk.m, n, 1.0f);
        if (op == 1) hipLaunchKernelGGL(k_chain<1>, dim3(64), dim3(64), 0, nullptr, out, mk.m, n, 1.0f);
        if (op == 2) hipLaunchKernelGGL(k_chain<2>, dim3(64), dim3(64), 0, nullptr, out, mk.m, n, 1.0f);
        if (op == 3) hipLaunchKernelGGL(k_chain<3>, dim3(64), dim3(64), 0, nullptr, out, mk.m, n, 1.0f);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf(" %.2f ns |", 1e6 * ms / ((double)n * 16 * 3));
    }
    printf("\n");
  }
  // ---- does a CU share its fp64 / conversion / transcendental pipe between its four SIMDs?  1, 2, 4 wavefronts per
  // workgroup (one per SIMD), all 64 lanes active, 64 workgroups: ns per instruction of one wave ----
  printf("\n| waves per workgroup (one per SIMD) | %s | %s | %s | %s |\n|---|---|---|---|---|\n", ops[0], ops[1], ops[2], ops[3]);
  for (int wpb = 1; wpb <= 8; wpb *= 2) {
    printf("| %d |", wpb);
    for (int op = 0; op < 4; ++op) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, nullptr));
        if (op == 0) hipLaunchKernelGGL(k_chain<0>, dim3(64), dim3(64 * wpb), 0, nullptr, out, ~0ull, n, 1.0f);
        if (op == 1) hipLaunchKernelGGL(k_chain<1>, dim3(64), dim3(64 * wpb), 0, nullptr, out, ~0ull, n, 1.0f);
        if (op == 2) hipLaunchKernelGGL(k_chain<2>, dim3(64), dim3(64 * wpb), 0, nullptr, out, ~0ull, n, 1.0f);
        if (op == 3) hipLaunchKernelGGL(k_chain<3>, dim3(64), dim3(64 * wpb), 0, nullptr, out, ~0ull, n, 1.0f);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf(" %.2f ns |", 1e6 * ms / ((double)n * 16 * 3));
    }
    printf("\n");
  }
  // ---- data dependence: the same chains on normal, fp32-subnormal, zero, infinite and NaN operands (all 64 lanes) ----
  printf("\n| operand | %s | %s | %s | %s |\n|---|---|---|---|---|\n", ops[0], ops[1], ops[2], ops[3]);
  struct { const char* name; float v; } vals[] = {{"1.0", 1.0f}, {"1e-20", 1e-20f}, {"1e-40 (fp32 subnormal)", 1e-40f}, {"0", 0.f},
                                                  {"inf", __builtin_inff()}, {"NaN", __builtin_nanf("")}};
  for (auto& vv : vals) {
    printf("| %s |", vv.name);
    for (int op = 0; op < 4; ++op) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, nullptr));
        if (op == 0) hipLaunchKernelGGL(k_chain<0>, dim3(64), dim3(64), 0, nullptr, out, ~0ull, n, vv.v);
        if (op == 1) hipLaunchKernelGGL(k_chain<1>, dim3(64), dim3(64), 0, nullptr, out, ~0ull, n, vv.v);
        if (op == 2) hipLaunchKernelGGL(k_chain<2>, dim3(64), dim3(64), 0, nullptr, out, ~0ull, n, vv.v);
        if (op == 3) hipLaunchKernelGGL(k_chain<3>, dim3(64), dim3(64), 0, nullptr, out, ~0ull, n, vv.v);
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf(" %.2f ns |", 1e6 * ms / ((double)n * 16 * 3));
    }
    printf("\n");
  }
  return 0;
}
