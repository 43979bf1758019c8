// exp_first_copy.hip — what do the FIRST stream, the first allocation and the first host -> device copies of a process cost on this stack?
//   hipcc -O2 --offload-arch=gfx950 -o profiles/_bin/exp_first_copy profiles/exp_first_copy.hip && profiles/_bin/exp_first_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using clk = std::chrono::steady_clock;
static clk::time_point t;
static void lap(const char* what) { auto n = clk::now(); std::printf("%-58s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count()); t = clk::now(); }
__global__ void k_copy(const float4* s, float4* d, size_t n) { for (size_t i = blockIdx.x * 256ul + threadIdx.x; i < n; i += gridDim.x * 256ul) d[i] = s[i]; }
int main() {
  t = clk::now();
  int n = 0; (void)hipGetDeviceCount(&n); lap("hipGetDeviceCount (runtime init)");
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); lap("hipStreamCreateWithFlags (first stream)");
  void *d0, *d1; (void)hipMalloc(&d0, 1 << 20); lap("hipMalloc 1 MB (first)"); (void)hipMalloc(&d1, 16 << 20); lap("hipMalloc 16 MB");
  (void)hipMemsetAsync(d0, 0, 1 << 20, s); (void)hipStreamSynchronize(s); lap("hipMemsetAsync 1 MB + sync (first fill)");
  std::vector<char> h(16 << 20, 1);
  (void)hipMemcpy(d0, h.data(), 4, hipMemcpyHostToDevice); lap("hipMemcpy H2D 4 B, pageable (first copy)");
  (void)hipMemcpy(d0, h.data(), 4, hipMemcpyHostToDevice); lap("hipMemcpy H2D 4 B again");
  (void)hipMemcpy(d0, h.data(), 256 << 10, hipMemcpyHostToDevice); lap("hipMemcpy H2D 256 KB, pageable");
  (void)hipMemcpy(d0, h.data(), 256 << 10, hipMemcpyHostToDevice); lap("hipMemcpy H2D 256 KB again");
  (void)hipMemcpy(d1, h.data(), 4 << 20, hipMemcpyHostToDevice); lap("hipMemcpy H2D 4 MB, pageable");
  (void)hipMemcpy(d1, h.data(), 4 << 20, hipMemcpyHostToDevice); lap("hipMemcpy H2D 4 MB again");
  void* p; (void)hipHostMalloc(&p, 4 << 20, hipHostMallocMapped); lap("hipHostMalloc 4 MB mapped (first pinned)");
  void* pd; (void)hipHostGetDevicePointer(&pd, p, 0);
  hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s, (const float4*)pd, (float4*)d1, (size_t)(256 << 10) / 16); (void)hipStreamSynchronize(s); lap("copy kernel 256 KB from mapped host memory + sync (first launch)");
  hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s, (const float4*)pd, (float4*)d1, (size_t)(4 << 20) / 16); (void)hipStreamSynchronize(s); lap("copy kernel 4 MB from mapped host memory + sync");
  (void)hipMemcpyAsync(d1, p, 4 << 20, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); lap("hipMemcpyAsync H2D 4 MB from pinned + sync");
  (void)hipMemcpy(h.data(), d1, 256 << 10, hipMemcpyDeviceToHost); lap("hipMemcpy D2H 256 KB, pageable (first D2H)");
  (void)hipMemcpy(h.data(), d1, 256 << 10, hipMemcpyDeviceToHost); lap("hipMemcpy D2H 256 KB again");
  return 0;
}
