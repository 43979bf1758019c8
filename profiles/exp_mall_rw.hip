// Can the 256 MiB Infinity Cache keep a read-modify-write working set T resident across iterations while a read-only stream S
// passes with the non-temporal hint?  One "iteration" = ONE kernel that updates T in place (load, +1, store) and reads S.
//   ./exp_mall_rw [T MiB] [S MiB]   -> us per iteration for the four (T load, T store) policies, and for S alone / T alone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int TL, int TS>   // T load / store policy: 0 default, 1 non-temporal
__global__ __launch_bounds__(256) void k_iter(v4f* T, size_t tn4, const v4f* S, size_t sn4, float* out) {
  const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  // interleave: each thread alternates one T element and (sn4 / tn4) S elements, like a sweep tile touching all its streams
  const size_t ratio = tn4 ? (sn4 + tn4 - 1) / tn4 : 0;
  for (size_t i = gid; i < tn4; i += gsz) {
    v4f v = TL ? __builtin_nontemporal_load(T + i) : T[i];
    for (size_t r = 0; r < ratio; ++r) {
      const size_t j = i * ratio + r;
      if (j < sn4) { const v4f s = __builtin_nontemporal_load(S + j); acc += s.x + s.y + s.z + s.w; }
    }
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    if (TS) __builtin_nontemporal_store(v, T + i); else T[i] = v;
  }
  if (tn4 == 0) for (size_t j = gid; j < sn4; j += gsz) { const v4f s = __builtin_nontemporal_load(S + j); acc += s.x + s.y + s.z + s.w; }
  if (acc == 123.456f) out[0] = acc;
}
int main(int argc, char** argv) {
  const size_t tmb = argc > 1 ? atoi(argv[1]) : 176, smb = argc > 2 ? atoi(argv[2]) : 224;
  const size_t tn4 = tmb * (1 << 20) / 16, sn4 = smb * (1 << 20) / 16;
  v4f *T, *S; float* out;
  CK(hipMalloc(&T, tn4 * 16 + 16)); CK(hipMalloc(&S, sn4 * 16 + 16)); CK(hipMalloc(&out, 4));
  CK(hipMemset(T, 0, tn4 * 16)); CK(hipMemset(S, 0, sn4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 16, reps = 30;
  auto run = [&](int mode) {
    for (int w = 0; w < 3; ++w) k_iter<0, 0><<<grid, 256>>>(T, tn4, S, sn4, out);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) {
      switch (mode) {
        case 0: k_iter<0, 0><<<grid, 256>>>(T, tn4, S, sn4, out); break;
        case 1: k_iter<0, 1><<<grid, 256>>>(T, tn4, S, sn4, out); break;
        case 2: k_iter<1, 0><<<grid, 256>>>(T, tn4, S, sn4, out); break;
        case 3: k_iter<1, 1><<<grid, 256>>>(T, tn4, S, sn4, out); break;
        case 4: k_iter<0, 0><<<grid, 256>>>(T, 0, S, sn4, out); break;          // S alone
        case 5: k_iter<0, 0><<<grid, 256>>>(T, tn4, S, 0, out); break;          // T alone, default / default
        case 6: k_iter<1, 1><<<grid, 256>>>(T, tn4, S, 0, out); break;          // T alone, nt / nt
      }
    }
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
  };
  const char* names[] = {"T default/default + S", "T default load, nt store + S", "T nt load, default store + S", "T nt/nt + S", "S alone (nt loads)", "T alone default/default", "T alone nt/nt"};
  printf("T = %zu MiB read-modify-write, S = %zu MiB read with nt; us per iteration (bytes moved if nothing is cached: %zu MB)\n", tmb, smb, (2 * tmb + smb) * 1048576 / 1000000);
  for (int m = 0; m < 7; ++m) { const float us = run(m); printf("  %-32s %8.1f us\n", names[m], us); }
  return 0;
}
