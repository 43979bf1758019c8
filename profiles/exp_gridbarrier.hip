// exp_gridbarrier.hip — what does a device-wide barrier cost on MI355X (8 XCDs, private L2s)?
//
// Floor of a persistent (one launch, many iterations) GBP kernel for small graphs: every iteration needs two
// grid-wide hand-offs (messages -> beliefs -> messages) through memory that another XCD wrote.
//   build:  hipcc -O3 --offload-arch=gfx950 -o profiles/_bin/exp_gridbarrier profiles/exp_gridbarrier.hip
//   run:    profiles/_bin/exp_gridbarrier            (prints one table)
// Each "round" = every block writes 1 KiB per wave, release, barrier, acquire, reads the 1 KiB of the NEXT block's wave
// (written on another XCD under round-robin placement) and checks it.  Variants:
//   flat     one monotonically increasing counter, agent-scope atomics, every wave's lane 0 polls it
//   mono1    the same, only thread 0 of the block arrives / polls, __syncthreads around it
//   launch   the same rounds as separate kernel launches (what a hipGraph replays), for comparison
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, bool one_per_block) {
  if (one_per_block) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope: L2 write-back so that other XCDs see this block's stores
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (unsigned spin = 0; __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spin < (1u << 22); ++spin)
        __builtin_amdgcn_s_sleep(1);             // bounded: a scheduling surprise must not hang the box
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  } else {
    if ((threadIdx.x & 63) == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (unsigned spin = 0; __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spin < (1u << 22); ++spin)
        __builtin_amdgcn_s_sleep(1);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
}

// hierarchical: 8 group counters (block b -> group b & 7, each on its own 128-B line) + one top counter bumped by the last
// arrival of every group; one thread per block arrives and polls.  counter layout: [g * 32] groups, [8 * 32] top.
__device__ __forceinline__ void grid_barrier_tree(unsigned* counter, unsigned epoch, unsigned nblocks) {
  __atomic_thread_fence(__ATOMIC_RELEASE);     // every wave: its own stores out to memory (agent scope)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x & 7u, ngroups = nblocks < 8u ? nblocks : 8u;
    const unsigned n_g = (nblocks - g + 7u) / 8u;
    const unsigned old = __hip_atomic_fetch_add(counter + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == epoch * n_g) __hip_atomic_fetch_add(counter + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spin = 0; __hip_atomic_load(counter + 8 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * ngroups && spin < (1u << 22); ++spin)
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

// flags: no read-modify-write at all.  Block b stores the epoch into flag[b]; the first wave of every block polls ALL
// flags (lane i reads flag[i], flag[i + 64], ...) until every one has reached the epoch.
__device__ __forceinline__ void grid_barrier_flags(unsigned* flags, unsigned epoch, unsigned nblocks) {
  __atomic_thread_fence(__ATOMIC_RELEASE);
  __syncthreads();
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spin = 0; spin < (1u << 22); ++spin) {
      bool ok = true;
      for (unsigned i = threadIdx.x; i < nblocks; i += 64)
        ok = ok && __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

// light: no L2 write-back / invalidate at all.  The exchanged DATA is written and read with agent-scope relaxed
// atomics (64-bit: global_store/load_dwordx2 sc1 — write-through / re-fetched per access), so the barrier only has to
// wait for the stores of the workgroup (s_waitcnt vmcnt(0) + s_barrier), then one arrival per workgroup on one counter.
__device__ __forceinline__ void grid_barrier_light(unsigned* counter, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have been acknowledged
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spin = 0; __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spin < (1u << 22); ++spin)
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void grid_barrier_agent(unsigned* counter, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // buffer_wbl2 sc1: agent scope, not the system-scope default of __atomic_thread_fence
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spin = 0; __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spin < (1u << 22); ++spin)
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int MODE>   // 0 flat (per wave), 1 one thread per block, 2 no barrier at all (one round per launch), 3 tree, 4 flags
__global__ void k_rounds(float4* buf, unsigned* counter, unsigned* errors, int rounds, int round0) {
  const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
  const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned arrivals = MODE == 1 ? gridDim.x : nwaves;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    const int rr = round0 + r;
    float4* mine = buf + ((size_t)(rr & 1) * nwaves + wave) * 64;
    if (MODE == 5) {
      unsigned long long* m64 = reinterpret_cast<unsigned long long*>(mine + lane);
      const float2 lo = make_float2((float)rr, (float)wave), hi = make_float2((float)lane, 1.f);
      __hip_atomic_store(m64, *reinterpret_cast<const unsigned long long*>(&lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(m64 + 1, *reinterpret_cast<const unsigned long long*>(&hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else
    mine[lane] = make_float4((float)rr, (float)wave, (float)lane, 1.f);
    if (MODE == 5) grid_barrier_light(counter, (unsigned)(r + 1) * gridDim.x);
    else if (MODE == 6) grid_barrier_agent(counter, (unsigned)(r + 1) * gridDim.x);
    else if (MODE == 3) grid_barrier_tree(counter, (unsigned)(r + 1), gridDim.x);
    else if (MODE == 4) grid_barrier_flags(counter, (unsigned)(r + 1), gridDim.x);
    else if (MODE != 2) grid_barrier(counter, (unsigned)(r + 1) * arrivals, MODE == 1);
    if (MODE != 2) {
      const unsigned other = (wave + (blockDim.x >> 6)) % nwaves;   // a wave of the next block: another XCD
      float4 v;
      if (MODE == 5) {
        const unsigned long long* o64 = reinterpret_cast<const unsigned long long*>(buf + ((size_t)(rr & 1) * nwaves + other) * 64 + lane);
        const unsigned long long a0 = __hip_atomic_load(o64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long a1 = __hip_atomic_load(o64 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float2 lo = *reinterpret_cast<const float2*>(&a0), hi = *reinterpret_cast<const float2*>(&a1);
        v = make_float4(lo.x, lo.y, hi.x, hi.y);
      } else
      v = buf[((size_t)(rr & 1) * nwaves + other) * 64 + lane];
      if (v.x != (float)rr || v.y != (float)other || v.z != (float)lane) ++bad;
    }
  }
  if (bad) atomicAdd(errors, bad);
}

int main() {
  unsigned *counter, *errors;
  float4* buf;
  CK(hipMalloc(&counter, 16384));
  CK(hipMalloc(&errors, 64));
  CK(hipMalloc(&buf, 2 * 4096 * 64 * sizeof(float4)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int rounds = 2000;
  printf("| blocks x threads | waves | flat: us/round | one-thread-per-block: us/round | separate launches: us/round | tree (8 groups): us/round | flags (no RMW): us/round | light (sc1 data, no L2 flush): us/round | one thread per block, agent-scope fences: us/round | errors |\n|---|---|---|---|---|---|---|---|---|---|\n");
  const int cfgs[][2] = {{14, 256}, {19, 256}, {51, 256}, {128, 256}, {256, 256}, {512, 256}, {56, 64}, {202, 64}, {1024, 64}};
  for (auto& cfg : cfgs) {
    const int nb = cfg[0], nt = cfg[1];
    double us[7];
    unsigned herr = 0;
    CK(hipMemset(errors, 0, 4));
    for (int mode = 0; mode < 7; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {   // rep 0 = warm-up
        CK(hipMemsetAsync(counter, 0, 16384, s));
        CK(hipEventRecord(e0, s));
        if (mode == 0) hipLaunchKernelGGL(k_rounds<0>, dim3(nb), dim3(nt), 0, s, buf, counter, errors, rounds, 0);
        else if (mode == 1) hipLaunchKernelGGL(k_rounds<1>, dim3(nb), dim3(nt), 0, s, buf, counter, errors, rounds, 0);
        else if (mode == 2) for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_rounds<2>, dim3(nb), dim3(nt), 0, s, buf, counter, errors, 1, r);
        else if (mode == 3) hipLaunchKernelGGL(k_rounds<3>, dim3(nb), dim3(nt), 0, s, buf, counter, errors, rounds, 0);
        else if (mode == 4) hipLaunchKernelGGL(k_rounds<4>, dim3(nb), dim3(nt), 0, s, buf, counter, errors, rounds, 0);
        else if (mode == 5) hipLaunchKernelGGL(k_rounds<5>, dim3(nb), dim3(nt), 0, s, buf, counter, errors, rounds, 0);
        else hipLaunchKernelGGL(k_rounds<6>, dim3(nb), dim3(nt), 0, s, buf, counter, errors, rounds, 0);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us[mode] = 1e3 * ms / rounds;
      }
    }
    CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
    printf("| %d x %d | %d | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %u |\n", nb, nt, nb * nt / 64, us[0], us[1], us[2], us[3], us[4], us[5], us[6], herr);
  }
  return 0;
}
