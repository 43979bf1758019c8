// Does a non-temporal stream displace a table from the 256 MiB Infinity Cache?  (MI355X; hipcc --offload-arch=gfx950 -O3)
//   read T (default policy) | stream S (variant) | read T again: the time of the second read tells whether T survived.
//   ./exp_mall_policy [T MiB = 128] [S MiB = 512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>   // 0 default load, 1 nt load
__global__ __launch_bounds__(256) void k_read(const v4f* p, size_t n4, float* out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const v4f v = MODE ? __builtin_nontemporal_load(p + i) : p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
template <int MODE>   // 0 default store, 1 nt store
__global__ __launch_bounds__(256) void k_write(v4f* p, size_t n4, float s) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const v4f v = {s, s + 1, s + 2, s + 3};
    if (MODE) __builtin_nontemporal_store(v, p + i); else p[i] = v;
  }
}
int main(int argc, char** argv) {
  const size_t tmb = argc > 1 ? atoi(argv[1]) : 128, smb = argc > 2 ? atoi(argv[2]) : 512;
  const size_t tn4 = tmb * (1 << 20) / 16, sn4 = smb * (1 << 20) / 16;
  v4f *T, *S; float* out;
  CK(hipMalloc(&T, tn4 * 16)); CK(hipMalloc(&S, sn4 * 16)); CK(hipMalloc(&out, 4));
  CK(hipMemset(T, 0, tn4 * 16)); CK(hipMemset(S, 0, sn4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  auto time_read_T = [&](float& us) { hipEventRecord(e0); k_read<0><<<grid, 256>>>(T, tn4, out); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); us = ms * 1e3f; return 0; };
  const char* names[] = {"nothing in between (T re-read at once)", "default loads of S", "nt loads of S", "default stores to S", "nt stores to S"};
  printf("T = %zu MiB, S = %zu MiB; second read of T after ...\n", tmb, smb);
  for (int v = 0; v < 5; ++v) {
    float best = 1e30f, sum = 0.f, sus = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
      float us;
      time_read_T(us);                      // brings T in
      hipEventRecord(e0);
      if (v == 1) k_read<0><<<grid, 256>>>(S, sn4, out);
      if (v == 2) k_read<1><<<grid, 256>>>(S, sn4, out);
      if (v == 3) k_write<0><<<grid, 256>>>(S, sn4, (float)rep);
      if (v == 4) k_write<1><<<grid, 256>>>(S, sn4, (float)rep);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); sus += ms * 1e3f;
      time_read_T(us);
      if (us < best) best = us;
      sum += us;
    }
    printf("  %-42s: T read in %.1f us best, %.1f mean = %.2f TB/s (best); the stream itself %.1f us = %.2f TB/s\n", names[v], best, sum / 5, tn4 * 16 / best / 1e6,
           sus / 5, v ? sn4 * 16 / (sus / 5) / 1e6 : 0.0);
  }
  return 0;
}
