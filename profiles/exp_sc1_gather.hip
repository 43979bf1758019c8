// exp_sc1_gather.hip — what does a wave-wide gather of 64-byte records cost in k_persist's belief phase?
//
// The landmark waves of k_persist gather 15 (or 30) records of 64 B per landmark, 16 landmarks per wave, 4 lanes x 16 B per
// record, every load `sc1` (written through / re-fetched: the arrays cross XCDs inside the launch).  A wave whose landmarks
// need the second batch of 15 takes +0.9 us although all 30 loads are in flight together (profiles/persist_trace.py), i.e.
// ~140 cycles per load instruction.  Is that the access shape, the sc1 policy, or the number of requests?
//   build:  hipcc -O3 --offload-arch=gfx950 -o profiles/_bin/exp_sc1_gather profiles/exp_sc1_gather.hip
//   run:    profiles/_bin/exp_sc1_gather
// One wave per SIMD, `nblk` workgroups of 4 waves spread like k_persist (every 4th workgroup works), each wave times R
// rounds of N independent loads (all issued, then consumed).  Variants:
//   quad16   4 lanes x 16 B per record, 16 random records per instruction            (k_persist today)
//   row16x4  16 lanes x 4 B per record (dword loads), 4 random records per instruction, 4x the instructions
//   lane64   1 lane reads a whole record with 4 dwordx4 loads, 64 random records per 4 instructions
//   seq      64 lanes x 16 B contiguous (1 KiB per instruction, a different KiB each)  (the coalesced reference)
// each with plain loads and with sc1 loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool SC1>
__device__ __forceinline__ v4u ld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, SC1 ? 16 : 0);
}
template <bool SC1>
__device__ __forceinline__ unsigned ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, SC1 ? 16 : 0);
}

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

constexpr int N = 15;

template <int SHAPE, bool SC1>
__global__ __launch_bounds__(256) void k(const float* table, unsigned n_rec, int rounds, unsigned spread, unsigned long long* ticks, float* sink) {
  if (blockIdx.x % spread) return;
  const unsigned bid = blockIdx.x / spread, wib = threadIdx.x >> 6, lane = threadIdx.x & 63, w = bid * 4 + wib;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(table), 0, 0x7fffffff, 0x00020000);
  float acc = 0.f;
  unsigned long long t0 = 0, t1 = 0;
  for (int it = -1; it < rounds; ++it) {
    if (it == 0) t0 = wall_clock64();
    if (SHAPE == 0) {            // quad16
      v4u m[N];
#pragma unroll
      for (int k2 = 0; k2 < N; ++k2) {
        const unsigned rec = hash((w * 64 + (lane >> 2)) * 131u + (unsigned)k2 * 7919u + (unsigned)(it + 1) * 104729u) & (n_rec - 1u);
        m[k2] = ld16<SC1>(r, rec * 64u + (lane & 3u) * 16u);
      }
#pragma unroll
      for (int k2 = 0; k2 < N; ++k2) acc += __uint_as_float(m[k2].x) + __uint_as_float(m[k2].w);
    } else if (SHAPE == 1) {     // row16x4: same bytes, 4 x N dword instructions
      unsigned m[4 * N];
#pragma unroll
      for (int k2 = 0; k2 < 4 * N; ++k2) {
        const unsigned rec = hash((w * 64 + (lane >> 4)) * 131u + (unsigned)k2 * 7919u + (unsigned)(it + 1) * 104729u) & (n_rec - 1u);
        m[k2] = ld4<SC1>(r, rec * 64u + (lane & 15u) * 4u);
      }
#pragma unroll
      for (int k2 = 0; k2 < 4 * N; ++k2) acc += __uint_as_float(m[k2]);
    } else if (SHAPE == 2) {     // lane64: 64 records per 4 instructions (4 x the bytes of quad16 per N... keep N/4 rounded up records per lane)
      v4u m[16];
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        const unsigned rec = hash((w * 64 + lane) * 131u + (unsigned)k2 * 7919u + (unsigned)(it + 1) * 104729u) & (n_rec - 1u);
#pragma unroll
        for (int g = 0; g < 4; ++g) m[k2 * 4 + g] = ld16<SC1>(r, rec * 64u + (unsigned)g * 16u);
      }
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) acc += __uint_as_float(m[k2].x) + __uint_as_float(m[k2].w);
    } else {                     // seq
      v4u m[N];
#pragma unroll
      for (int k2 = 0; k2 < N; ++k2) {
        const unsigned kib = hash(w * 131u + (unsigned)k2 * 7919u + (unsigned)(it + 1) * 104729u) & (n_rec / 16u - 1u);
        m[k2] = ld16<SC1>(r, kib * 1024u + lane * 16u);
      }
#pragma unroll
      for (int k2 = 0; k2 < N; ++k2) acc += __uint_as_float(m[k2].x) + __uint_as_float(m[k2].w);
    }
  }
  t1 = wall_clock64();
  if (lane == 0) ticks[w] = t1 - t0;
  if (acc == 12345.678f) sink[0] = acc;
}

template <int SHAPE, bool SC1>
static void run(const char* label, const float* table, unsigned n_rec, unsigned nblk, unsigned long long* ticks, float* sink, int instr_per_round, double kib_per_round) {
  const int rounds = 200;
  const unsigned spread = 4;
  hipLaunchKernelGGL((k<SHAPE, SC1>), dim3(nblk * spread), dim3(256), 0, 0, table, n_rec, rounds, spread, ticks, sink);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(nblk * 4);
  CK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0, mx = 0;
  for (auto v : h) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
  const double us = sum / h.size() / 100.0 / rounds, us_max = mx / 100.0 / rounds;   // 100 MHz clock
  printf("| %-8s | %-5s | %5.2f | %5.2f | %6.0f | %5.1f |\n", label, SC1 ? "sc1" : "plain", us, us_max, us * 2400.0 / instr_per_round, kib_per_round);
}

int main() {
  const unsigned n_rec = 16384;            // a power of two (index = hash & mask); fr1xyz has 12 908 factors; 64 B each = 1 MiB
  float* table; unsigned long long* ticks; float* sink;
  CK(hipMalloc(&table, (size_t)n_rec * 64));
  CK(hipMemset(table, 0, (size_t)n_rec * 64));
  CK(hipMalloc(&ticks, 4096 * 8));
  CK(hipMalloc(&sink, 64));
  for (unsigned nblk : {1u, 13u, 52u}) {
    printf("\n%u working workgroups (4 waves each, one per SIMD), table of %u records x 64 B\n", nblk, n_rec);
    printf("| shape | loads | us per round (mean) | (slowest wave) | cycles per load instruction | KiB per wave and round |\n|---|---|---|---|---|---|\n");
    run<0, false>("quad16", table, n_rec, nblk, ticks, sink, N, N * 1.0);
    run<0, true>("quad16", table, n_rec, nblk, ticks, sink, N, N * 1.0);
    run<1, false>("row16x4", table, n_rec, nblk, ticks, sink, 4 * N, N * 1.0);
    run<1, true>("row16x4", table, n_rec, nblk, ticks, sink, 4 * N, N * 1.0);
    run<2, false>("lane64", table, n_rec, nblk, ticks, sink, 16, 16.0);
    run<2, true>("lane64", table, n_rec, nblk, ticks, sink, 16, 16.0);
    run<3, false>("seq", table, n_rec, nblk, ticks, sink, N, N * 1.0);
    run<3, true>("seq", table, n_rec, nblk, ticks, sink, N, N * 1.0);
  }
  return 0;
}
