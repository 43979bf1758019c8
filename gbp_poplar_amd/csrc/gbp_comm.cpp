// gbp_comm.cpp — transports of the per-iteration camera-partial all-gather (see gbp_comm.hpp).
#include "gbp_comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen'ed, nothing links against it
#include <sched.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace gbp {
namespace {

// ---- RCCL through dlopen ------------------------------------------------------------------------------------
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;     // optional
  std::string path;                                    // resolved file the symbols come from (dladdr)
  std::string error;
};

RcclApi& rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  // Inside a PyTorch process torch's own librccl (loaded as "librccl.so") is already there: reuse it, two RCCL
  // runtimes in one process would each bring their own bootstrap and HIP state.
  // GBP_RCCL_LIB names the library explicitly: then ONLY that path is tried (no silent substitute).
  const char* env = std::getenv("GBP_RCCL_LIB");
  void* h = nullptr;
  std::string tried_names;
  auto open = [&](const char* name, int flags) {
    if (h) return;
    (void)dlerror();
    h = dlopen(name, flags);
    if (!h && !(flags & RTLD_NOLOAD)) {
      const char* de = dlerror();     // read ONCE: the call clears the message
      tried_names += std::string(tried_names.empty() ? "" : "; ") + name + ": " + (de ? de : "?");
    }
  };
  if (env && *env) {
    open(env, RTLD_NOW | RTLD_LOCAL);
  } else {
    open("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    open("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) open(name, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) {
    api.error = "librccl not found (dlopen: " + tried_names + ")";
    return api;
  }
  api.handle = h;
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(h, "ncclGetVersion"));
  Dl_info info;
  if (api.AllGather && dladdr(reinterpret_cast<void*>(api.AllGather), &info) && info.dli_fname) api.path = info.dli_fname;
  if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy || !api.GetErrorString) {
    api.error = "librccl lacks an expected symbol";
    api.handle = nullptr;
  }
  return api;
}

std::string nccl_err(const char* what, ncclResult_t r) {
  return std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
}

// ---- shared region ----------------------------------------------------------------------------------------------
constexpr uint32_t kMagic = 0x47425043u;   // "GBPC"
constexpr double kTimeoutS = 300.0;

struct RegionHeader {
  uint32_t magic, world, n_cams, pad;
  std::atomic<uint32_t> bar_count, bar_sense, abort_flag, id_ready;
  char id[kCommIdBytes];
  char gpu_id[kCommMaxWorld][64];
  double scratch[kCommMaxWorld][16];
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "cross-process barrier needs lock-free atomics");

inline size_t header_bytes() { return (sizeof(RegionHeader) + 255) / 256 * 256; }
inline float* region_data(void* region) { return reinterpret_cast<float*>(static_cast<char*>(region) + header_bytes()); }

struct RegionPeer {   // one rank's view of the region: sense-reversing barrier, abortable and bounded in time
  RegionHeader* h = nullptr;
  uint32_t sense = 0;
  int world = 1;
  bool wait_until(const std::atomic<uint32_t>& a, uint32_t want, std::string& err, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (a.load(std::memory_order_acquire) != want) {
      if (h->abort_flag.load(std::memory_order_acquire)) { err = std::string(what) + ": another rank aborted"; return false; }
      if (++spins > 2000) {
        usleep(50);
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) {
          h->abort_flag.store(1, std::memory_order_release);
          err = std::string(what) + ": timed out waiting for the other ranks";
          return false;
        }
      } else {
        sched_yield();
      }
    }
    return true;
  }
  bool barrier(std::string& err) {
    sense ^= 1u;
    if (h->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
      h->bar_count.store(0, std::memory_order_relaxed);
      h->bar_sense.store(sense, std::memory_order_release);
      return true;
    }
    return wait_until(h->bar_sense, sense, err, "barrier");
  }
  bool gather_host(int rank, const double* mine, double* all, int n, std::string& err) {
    if (n > 16) { err = "all_gather_host: at most 16 values"; return false; }
    std::memcpy(h->scratch[rank], mine, sizeof(double) * n);
    if (!barrier(err)) return false;
    for (int r = 0; r < world; ++r) std::memcpy(all + (size_t)r * n, h->scratch[r], sizeof(double) * n);
    return barrier(err);    // nobody overwrites the scratch before everyone has read it
  }
};

// ---- RCCL transport ---------------------------------------------------------------------------------------------
class RcclComm : public Comm {
 public:
  ncclComm_t comm = nullptr;
  RegionPeer peer;                 // launcher-made groups: host-side gather / barrier through the region
  double* d_scratch = nullptr;     // id-made groups (no region): small gathers through RCCL itself
  ~RcclComm() override {
    if (d_scratch) (void)hipFree(d_scratch);
    if (comm && rccl().CommDestroy) (void)rccl().CommDestroy(comm);
  }
  int all_gather(const float* send, float* recv, size_t n, hipStream_t s, std::string& err) override {
    const ncclResult_t r = rccl().AllGather(send, recv, n, ncclFloat, comm, s);
    if (r != ncclSuccess) { err = nccl_err("ncclAllGather", r); return -1; }
    return 0;
  }
  bool stream_ordered() const override { return true; }
  int all_gather_host(const double* mine, double* all, int n, std::string& err) override {
    if (peer.h) return peer.gather_host(rank, mine, all, n, err) ? 0 : -1;
    if (n > 16) { err = "all_gather_host: at most 16 values"; return -1; }
    if (!d_scratch && hipMalloc(&d_scratch, sizeof(double) * 16 * (size_t)(world + 1)) != hipSuccess) { err = "hipMalloc"; return -1; }
    if (hipMemcpy(d_scratch, mine, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy"; return -1; }
    const ncclResult_t r = rccl().AllGather(d_scratch, d_scratch + 16, (size_t)n, ncclDouble, comm, nullptr);
    if (r != ncclSuccess) { err = nccl_err("ncclAllGather", r); return -1; }
    if (hipStreamSynchronize(nullptr) != hipSuccess ||
        hipMemcpy(all, d_scratch + 16, sizeof(double) * n * (size_t)world, hipMemcpyDeviceToHost) != hipSuccess) { err = "hip sync/copy"; return -1; }
    return 0;
  }
  int barrier(std::string& err) override {
    if (peer.h) return peer.barrier(err) ? 0 : -1;
    double x = 0, all[kCommMaxWorld];
    return all_gather_host(&x, all, 1, err);
  }
  const char* name() const override { return "rccl"; }
  std::string library_path() const override { return rccl().path; }
  int library_version() const override {
    int v = 0;
    if (rccl().GetVersion && rccl().GetVersion(&v) == ncclSuccess) return v;
    return 0;
  }
};

// ---- host-staged transport (ranks sharing a GPU) -----------------------------------------------------------------------
class StagedComm : public Comm {
 public:
  RegionPeer peer;
  float* data = nullptr;        // [2][world][n_floats]
  size_t n_floats = 0;
  int parity = 0;
  std::vector<float> stage;
  int all_gather(const float* send, float* recv, size_t n, hipStream_t s, std::string& err) override {
    if (n > n_floats) { err = "all_gather: message larger than the staging region"; return -1; }
    float* slot = data + ((size_t)parity * world + rank) * n_floats;
    if (hipMemcpyAsync(slot, send, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
      err = "all_gather: device -> region copy failed";
      return -1;
    }
    if (!peer.barrier(err)) return -1;
    // a rank can be at most one exchange ahead of the slowest one (the next barrier needs everybody), so the
    // two parities never collide
    stage.resize((size_t)world * n);
    for (int r = 0; r < world; ++r) std::memcpy(&stage[(size_t)r * n], data + ((size_t)parity * world + r) * n_floats, n * 4);
    if (hipMemcpyAsync(recv, stage.data(), stage.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
      err = "all_gather: region -> device copy failed";
      return -1;
    }
    parity ^= 1;
    return 0;
  }
  bool stream_ordered() const override { return false; }
  int all_gather_host(const double* mine, double* all, int n, std::string& err) override {
    return peer.gather_host(rank, mine, all, n, err) ? 0 : -1;
  }
  int barrier(std::string& err) override { return peer.barrier(err) ? 0 : -1; }
  const char* name() const override { return "host-staged"; }
};

}  // namespace

int comm_unique_id(void* id128, std::string& err) {
  RcclApi& api = rccl();
  if (!api.handle) { err = api.error; return -1; }
  ncclUniqueId id;
  const ncclResult_t r = api.GetUniqueId(&id);
  if (r != ncclSuccess) { err = nccl_err("ncclGetUniqueId", r); return -1; }
  std::memcpy(id128, id.internal, kCommIdBytes);
  return 0;
}

Comm* comm_create_rccl(const void* id128, int rank, int world, std::string& err) {
  RcclApi& api = rccl();
  if (!api.handle) { err = api.error; return nullptr; }
  if (world < 1 || world > kCommMaxWorld || rank < 0 || rank >= world) { err = "bad rank / world"; return nullptr; }
  ncclUniqueId id;
  std::memcpy(id.internal, id128, kCommIdBytes);
  RcclComm* c = new (std::nothrow) RcclComm();
  if (!c) { err = "out of memory"; return nullptr; }
  c->rank = rank; c->world = world;
  const ncclResult_t r = api.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    err = nccl_err("ncclCommInitRank", r);
    c->comm = nullptr;
    delete c;
    return nullptr;
  }
  return c;
}

size_t comm_region_bytes(uint32_t n_cams, int world) {
  if (world < 1) world = 1;
  return header_bytes() + (size_t)2 * world * n_cams * 44 * sizeof(float);
}

int comm_region_init(void* region, size_t bytes, uint32_t n_cams, int world) {
  if (!region || world < 1 || world > kCommMaxWorld || bytes < comm_region_bytes(n_cams, world)) return -1;
  RegionHeader* h = new (region) RegionHeader();
  h->magic = kMagic; h->world = (uint32_t)world; h->n_cams = n_cams; h->pad = 0;
  h->bar_count.store(0); h->bar_sense.store(0); h->abort_flag.store(0); h->id_ready.store(0);
  std::memset(h->id, 0, sizeof(h->id));
  std::memset(h->gpu_id, 0, sizeof(h->gpu_id));
  return 0;
}

void comm_region_abort(void* region) {
  RegionHeader* h = static_cast<RegionHeader*>(region);
  if (h && h->magic == kMagic) h->abort_flag.store(1, std::memory_order_release);
}

// The cross-process protocol of the region on its own (no device): `rounds` x {gather of two doubles per rank, barrier},
// every rank checks what it received.  Lets the rendezvous / staging logic be tested on a CPU-only box.
int comm_region_selftest(void* region, int rank, int world, int rounds, std::string& err) {
  RegionHeader* h = static_cast<RegionHeader*>(region);
  if (!h || h->magic != kMagic || (int)h->world != world || rank < 0 || rank >= world) { err = "bad communication region"; return -1; }
  RegionPeer peer;
  peer.h = h; peer.world = world;
  std::vector<double> all((size_t)world * 2);
  for (int r = 0; r < rounds; ++r) {
    const double mine[2] = {(double)(rank * 1000 + r), (double)(r - rank)};
    if (!peer.gather_host(rank, mine, all.data(), 2, err)) return -1;
    for (int k = 0; k < world; ++k)
      if (all[(size_t)k * 2] != (double)(k * 1000 + r) || all[(size_t)k * 2 + 1] != (double)(r - k)) { err = "selftest: wrong data from a rank"; return -2; }
    if (!peer.barrier(err)) return -1;
  }
  return 0;
}

Comm* comm_create_from_region(void* region, int rank, int world, int transport, std::string& err) {
  RegionHeader* h = static_cast<RegionHeader*>(region);
  if (!h || h->magic != kMagic || (int)h->world != world || rank < 0 || rank >= world) { err = "bad communication region"; return nullptr; }
  RegionPeer peer;
  peer.h = h; peer.world = world;
  // 1. every rank publishes the identity of its GPU; a GPU shared by two ranks rules RCCL out
  int dev = 0;
  char bus[64] = {0};
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof(bus), dev) != hipSuccess) { err = "hipDeviceGetPCIBusId failed"; return nullptr; }
  std::memcpy(h->gpu_id[rank], bus, sizeof(bus));
  if (!peer.barrier(err)) return nullptr;
  bool shared = false;
  for (int a = 0; a < world; ++a)
    for (int b = a + 1; b < world; ++b)
      if (std::strncmp(h->gpu_id[a], h->gpu_id[b], 64) == 0) shared = true;
  if (transport == 1 && shared) { err = "RCCL needs one GPU per rank, but two ranks share a GPU"; h->abort_flag.store(1); return nullptr; }
  const bool use_rccl = transport == 1 || (transport == 0 && !shared);
  if (!use_rccl) {
    StagedComm* c = new (std::nothrow) StagedComm();
    if (!c) { err = "out of memory"; return nullptr; }
    c->rank = rank; c->world = world; c->peer = peer;
    c->data = region_data(region);
    c->n_floats = (size_t)h->n_cams * 44;
    if (!c->peer.barrier(err)) { delete c; return nullptr; }
    return c;
  }
  // 2. RCCL: rank 0 draws the unique id, the region hands it to the others
  if (rank == 0) {
    if (comm_unique_id(h->id, err) != 0) { h->abort_flag.store(1); return nullptr; }
    h->id_ready.store(1, std::memory_order_release);
  } else if (!peer.wait_until(h->id_ready, 1u, err, "RCCL unique id")) {
    return nullptr;
  }
  Comm* c = comm_create_rccl(h->id, rank, world, err);
  if (!c) { h->abort_flag.store(1); return nullptr; }
  static_cast<RcclComm*>(c)->peer = peer;
  if (!static_cast<RcclComm*>(c)->peer.barrier(err)) { delete c; return nullptr; }
  return c;
}

}  // namespace gbp
