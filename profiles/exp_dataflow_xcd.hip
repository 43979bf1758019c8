// exp_dataflow_xcd.hip — tagged-record hand-offs between waves of the SAME XCD: can the records stay in that XCD's L2?
//
// k_persist_flow reads and writes every cross-wave record with the sc1 cache policy (write-through / re-fetched through the fabric:
// the 8 XCDs' L2s are not coherent with each other).  Producer and consumer on one XCD share an L2; this measures the round time of
// exp_dataflow.hip's tagged exchange for partners on the same XCD (block b <-> block b + 8 x k) with each cache-policy pair, and
// counts rounds that never saw their data (time-outs) — a policy that is not coherent across CUs shows up there, not as wrong data
// (the tag is in the record).
//   build:  hipcc -O3 --offload-arch=gfx950 -o profiles/_bin/exp_dataflow_xcd profiles/exp_dataflow_xcd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int LD, int ST>
struct Xw {
  __amdgpu_buffer_rsrc_t r;
  __device__ explicit Xw(void* base) : r(__builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000)) {}
  __device__ v4u ld(unsigned i4) const { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i4 * 16u), 0, LD); }
  __device__ void st(unsigned i4, v4u v) const { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(i4 * 16u), 0, ST); }
};

// grid = nb blocks of 256; partner of wave (b, w) is wave (pb, w) with pb = b + hop (mod nb): hop = 8 keeps the XCD (round-robin dispatch)
template <int LD, int ST, int K>
__global__ __launch_bounds__(256) void k_rounds(void* buf_, unsigned* errors, int rounds, unsigned hop, int work) {
  const unsigned nb = gridDim.x, nw = nb * 4, wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * 4 + wib;
  // pair blocks: even "group" sends to odd and back, so that the skew between the two is bounded by the exchange itself
  const unsigned grp = blockIdx.x / hop, pos = blockIdx.x % hop;
  const unsigned pb = ((grp ^ 1u) * hop + pos);
  if (pb >= nb) return;
  const unsigned other = pb * 4 + wib;
  const Xw<LD, ST> X(buf_);
  unsigned bad = 0, lost = 0;
  float acc = (float)lane;
  for (int r = 1; r <= rounds; ++r) {
    const unsigned half = ((unsigned)r & 1u) * nw * 64u * (unsigned)K;
#pragma unroll 1
    for (int i = 0; i < work; ++i)
      asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(1.0000001f), "v"(0.5f));
    const unsigned a = __float_as_uint(acc);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const v4u v = {a, wave, lane * 64u + (unsigned)k, (unsigned)r};
      X.st(half + (wave * (unsigned)K + (unsigned)k) * 64u + lane, v);
    }
    v4u v[K];
    bool got = false;
    for (unsigned spin = 0; spin < (1u << 14); ++spin) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = X.ld(half + (other * (unsigned)K + (unsigned)k) * 64u + lane);
      bool ok = true;
#pragma unroll
      for (int k = 0; k < K; ++k) ok = ok && v[k].w == (unsigned)r;
      if (__all(ok)) { got = true; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!got) { ++lost; if (lost > 3) break; }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (got && (v[k].y != other || v[k].z != lane * 64u + (unsigned)k)) ++bad;
      acc += __uint_as_float(v[k].x) * 1e-30f;
    }
  }
  if (bad) atomicAdd(errors, bad);
  if (lost && lane == 0) atomicAdd(errors + 1, lost);
  if (acc == 123.456f) errors[2] = 1;
}

template <int LD, int ST, int K>
static void run(const char* name, int nb, unsigned hop, int work, void* buf, unsigned* errors, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  const int rounds = 2000;
  double us = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemsetAsync(errors, 0, 64, s));
    CK(hipMemsetAsync(buf, 0, (size_t)2 * 4096 * 64 * K * 16, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((k_rounds<LD, ST, K>), dim3(nb), dim3(256), 0, s, buf, errors, rounds, hop, work);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    us = 1e3 * ms / rounds;
  }
  unsigned h[16];
  CK(hipMemcpy(h, errors, 64, hipMemcpyDeviceToHost));
  printf("| %s | %d | hop %u (%s) | %d | 4 x %d fma | %.2f | %u | %u |\n", name, nb, hop, hop % 8 == 0 ? "same XCD" : "another XCD", K, work, us, h[0], h[1]);
  fflush(stdout);
}

int main() {
  unsigned* errors;
  void* buf;
  CK(hipMalloc(&errors, 64));
  CK(hipMalloc(&buf, (size_t)2 * 4096 * 64 * 16 * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  printf("| load / store policy | blocks | partner | K | work | us per round | wrong records | rounds that never saw their data |\n|---|---|---|---|---|---|---|---|\n");
  for (int nb : {64, 256})
    for (unsigned hop : {8u, 4u})
      for (int work : {0, 100}) {
        run<16, 16, 16>("sc1 / sc1 (product)", nb, hop, work, buf, errors, s, e0, e1);
        run<1, 1, 16>("sc0 / sc0", nb, hop, work, buf, errors, s, e0, e1);
        run<1, 0, 16>("sc0 / default", nb, hop, work, buf, errors, s, e0, e1);
        run<1, 16, 16>("sc0 / sc1", nb, hop, work, buf, errors, s, e0, e1);
        run<17, 17, 16>("sc0 sc1 / sc0 sc1", nb, hop, work, buf, errors, s, e0, e1);
      }
  return 0;
}
