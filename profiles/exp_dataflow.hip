// exp_dataflow.hip — device-wide hand-off of a persistent kernel: counter barrier against TAGGED RECORDS (MI355X, 8 XCDs).
//
// k_persist pays two device-wide hand-offs per GBP iteration: sc1 (write-through) stores, s_waitcnt vmcnt(0), one arrival per
// workgroup on a counter, a poll, then sc1 loads of what other XCDs wrote.  Alternative measured here: no counter at all — every
// 16-byte record carries the round number in its fourth word, a consumer re-loads its K records until all of them carry the round
// it waits for (a 16-byte aligned store of one lane is not torn; records are double-buffered by the parity of the round, which the
// data dependencies of the algorithm make sufficient).
//   build:  hipcc -O3 --offload-arch=gfx950 -o profiles/_bin/exp_dataflow profiles/exp_dataflow.hip
//   run:    profiles/_bin/exp_dataflow
// One round = every wave stores K records per lane, then reads the K records per lane of a wave of ANOTHER workgroup
// (round-robin placement: another XCD) and checks them.  Two rounds = one GBP iteration (messages -> beliefs -> messages).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Xw {
  __amdgpu_buffer_rsrc_t r;
  __device__ explicit Xw(void* base) : r(__builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000)) {}
  __device__ v4u ld(unsigned i4) const { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i4 * 16u), 0, 16); }
  __device__ void st(unsigned i4, v4u v) const { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(i4 * 16u), 0, 16); }
};

__device__ __forceinline__ void grid_sync(unsigned* sync, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spin = 0;
    while ((int)(__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && ++spin < (1u << 24)) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// MODE 0: counter barrier (the product's hand-off);  MODE 1: tagged records, no barrier;  WORK: dependent fma chain between load and store
template <int MODE, int K>
__global__ __launch_bounds__(256) void k_rounds(void* buf_, unsigned* sync, unsigned* errors, unsigned* spins, int rounds, unsigned spread, int work) {
  if (blockIdx.x % spread) return;
  const unsigned bid = blockIdx.x / spread, nblk = gridDim.x / spread;
  const unsigned nw = nblk * 4, wave = bid * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const Xw X(buf_);
  const unsigned other = (wave + nw / 2u) % nw;        // its partner (whose partner this wave is: the skew between the two is bounded by the
  // exchange itself, as between a factor's wave and its variables' owners) half the grid on: another workgroup, another CU
  unsigned bad = 0, nspin = 0;
  float acc = (float)lane;
  for (int r = 1; r <= rounds; ++r) {
    const unsigned half = ((unsigned)r & 1u) * nw * 64u * (unsigned)K;
#pragma unroll 1
    for (int i = 0; i < work; ++i)        // the phase's arithmetic: a dependent chain, the same instructions in every instantiation
      asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(1.0000001f), "v"(0.5f));
    const unsigned a = __float_as_uint(acc);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const v4u v = {a, wave, lane * 64u + (unsigned)k, (unsigned)r};
      X.st(half + (wave * (unsigned)K + (unsigned)k) * 64u + lane, v);
    }
    if (MODE == 0) grid_sync(sync, (unsigned)r * nblk);
    v4u v[K];
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = X.ld(half + (other * (unsigned)K + (unsigned)k) * 64u + lane);
    } else {
      for (unsigned spin = 0; spin < (1u << 16); ++spin) {
        asm volatile("" ::: "memory");      // the loads are re-issued every time round (the intrinsic is not volatile)
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = X.ld(half + (other * (unsigned)K + (unsigned)k) * 64u + lane);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < K; ++k) ok = ok && v[k].w == (unsigned)r;
        if (__all(ok)) break;
        ++nspin;
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (v[k].y != other || v[k].z != lane * 64u + (unsigned)k || v[k].w != (unsigned)r) ++bad;
      acc += __uint_as_float(v[k].x) * 1e-30f;
    }
  }
  if (bad) atomicAdd(errors, bad);
  if (lane == 0) atomicAdd(spins, nspin);
  if (acc == 123.456f) errors[1] = 1;
}

template <int MODE, int K>
static void run(int nb, int spread, int work, void* buf, unsigned* sync, unsigned* errors, hipStream_t s, hipEvent_t e0, hipEvent_t e1, double& us, unsigned& err, double& spins_per_round) {
  const int rounds = 2000;
  unsigned* spins = errors + 8;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemsetAsync(sync, 0, 4096, s));
    CK(hipMemsetAsync(errors, 0, 64, s));
    CK(hipMemsetAsync(buf, 0, (size_t)2 * 4096 * 64 * K * 16, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((k_rounds<MODE, K>), dim3(nb * spread), dim3(256), 0, s, buf, sync, errors, spins, rounds, (unsigned)spread, work);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    us = 1e3 * ms / rounds;
  }
  unsigned h[16];
  CK(hipMemcpy(h, errors, 64, hipMemcpyDeviceToHost));
  err = h[0];
  spins_per_round = (double)h[8] / ((double)nb * 4 * rounds);
}

int main() {
  unsigned *sync, *errors;
  void* buf;
  CK(hipMalloc(&sync, 4096));
  CK(hipMalloc(&errors, 64));
  CK(hipMalloc(&buf, (size_t)2 * 4096 * 64 * 16 * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  printf("| workgroups (x spread) | arithmetic per round | K records per lane | counter barrier: us/round | tagged records: us/round | re-loads per wave and round | errors |\n|---|---|---|---|---|---|---|\n");
  const int cfgs[][2] = {{14, 4}, {51, 4}, {61, 4}, {96, 2}, {128, 2}, {256, 1}};
  for (auto& c : cfgs)
    for (int work : {0, 100}) {
      double u0, u1, sp0, sp1;
      unsigned er0, er1;
      run<0, 4>(c[0], c[1], work, buf, sync, errors, s, e0, e1, u0, er0, sp0);
      run<1, 4>(c[0], c[1], work, buf, sync, errors, s, e0, e1, u1, er1, sp1);
      printf("| %d (x %d) | 4 x %d fma | 4 | %.2f | %.2f | %.2f | %u / %u |\n", c[0], c[1], work, u0, u1, sp1, er0, er1);
      fflush(stdout);
      run<0, 16>(c[0], c[1], work, buf, sync, errors, s, e0, e1, u0, er0, sp0);
      run<1, 16>(c[0], c[1], work, buf, sync, errors, s, e0, e1, u1, er1, sp1);
      printf("| %d (x %d) | 4 x %d fma | 16 | %.2f | %.2f | %.2f | %u / %u |\n", c[0], c[1], work, u0, u1, sp1, er0, er1);
      fflush(stdout);
    }
  return 0;
}
