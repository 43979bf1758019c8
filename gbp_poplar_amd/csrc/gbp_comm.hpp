// gbp_comm.hpp — the one exchange step of a landmark-sharded GBP iteration, from the C++ host.
//
// Replaces what Poplar compiles for `--ipus N` (reference ba/ba.cpp:414-417,617-649: one graph over N x 1216 tiles,
// inter-IPU exchange generated from the static graph).  Here every rank (one process per GPU) holds a [C x 44] fp32
// buffer of camera partial sums; one ALL-GATHER per iteration gives every rank all of them, and each rank adds
// prior + partials in rank order (k_beliefs) — deterministic, bit-identical camera beliefs on all ranks.
//
// Two transports behind one interface:
//   * RCCL (xGMI): ncclAllGather on a HIP stream — stream-ordered, capturable into the iteration's hipGraph.  librccl
//     is dlopen'ed on first use (no link-time dependency: a single-GPU user never loads it, and inside a PyTorch
//     process the already-loaded librccl is reused instead of a second copy).
//   * host-staged: ranks that SHARE a GPU (fewer GPUs than ranks: test rigs, `--ipus 2` on a one-GPU box) cannot form
//     an RCCL communicator ("duplicate GPU"); their partials travel through a MAP_SHARED region (D2H, barrier, H2D).
//     It moves the same bytes in the same layout, only slower; nothing is computed on the host.
// The shared region also carries the rendezvous of a forked launcher (RCCL unique id, per-rank GPU identity, barrier).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

namespace gbp {

constexpr int kCommIdBytes = 128;      // NCCL_UNIQUE_ID_BYTES
constexpr int kCommMaxWorld = 64;

class Comm {
 public:
  virtual ~Comm() {}
  // recv[r][0..n) = send of rank r, for all r.  Stream-ordered transports enqueue on `s`; the host-staged one
  // synchronises `s`, exchanges and returns with recv complete.
  virtual int all_gather(const float* send_dev, float* recv_dev, size_t n, hipStream_t s, std::string& err) = 0;
  virtual bool stream_ordered() const = 0;
  // small host-side gather (metric sums): all[r*n + i] = mine[i] of rank r
  virtual int all_gather_host(const double* mine, double* all, int n, std::string& err) = 0;
  virtual int barrier(std::string& err) = 0;
  virtual const char* name() const = 0;
  // where the collective library was loaded from and its version ("" / 0 for the host-staged transport): what a first
  // multi-GPU run wants on record next to its numbers
  virtual std::string library_path() const { return ""; }
  virtual int library_version() const { return 0; }
  int rank = 0, world = 1;
};

// RCCL
int comm_unique_id(void* id128, std::string& err);
Comm* comm_create_rccl(const void* id128, int rank, int world, std::string& err);

// shared rendezvous / staging region (created by the launcher before the ranks start, e.g. mmap MAP_SHARED|MAP_ANONYMOUS)
size_t comm_region_bytes(uint32_t n_cams, int world);
int comm_region_init(void* region, size_t bytes, uint32_t n_cams, int world);
void comm_region_abort(void* region);
int comm_region_selftest(void* region, int rank, int world, int rounds, std::string& err);   // protocol check, no device   // a supervisor saw a rank die: wake every rank waiting in the region with an error
// transport: 0 = auto (RCCL when every rank sits on its own GPU, host-staged otherwise), 1 = RCCL, 2 = host-staged
Comm* comm_create_from_region(void* region, int rank, int world, int transport, std::string& err);

}  // namespace gbp
