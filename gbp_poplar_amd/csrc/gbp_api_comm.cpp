// gbp_api_comm.cpp — the landmark-sharded ctx (include/gbp_mi355x_multi.h): communicator glue, the sharded iteration and
// gbp_eval_global.  Replaces `--ipus N` (reference ba/ba.cpp:414-417,617-649: Poplar compiles the inter-IPU exchange into the
// program): one process per GPU, each rank owns a landmark range and every factor incident to it; per iteration ONE all-gather of
// the [C x 44] camera partial sums (csrc/gbp_comm.*: RCCL over xGMI, or the host-staged transport when ranks share a GPU), every
// rank then adds prior + partials in rank order — bit-identical camera beliefs on all ranks.
#include "gbp_ctx.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace gbp;
using namespace gbp::api;

namespace gbp {
namespace api {

// plain exchange on the ctx's stream (LINEARISE, refreshes): partials in send_dev -> recv_dev of every rank
int exchange_now(gbp_ctx* c) {
  COMMCHK(c, c->comm->all_gather(static_cast<const float*>(c->send_dev), static_cast<float*>(c->recv_dev),
                                 (size_t)c->C * kCamRec, c->stream, e_));
  return GBP_OK;
}

// One iteration of a sharded ctx: sweep + local camera partials, the ALL-GATHER of the partials on the communication
// stream while the rank-local landmark half of the belief update runs, then the camera combine.  With a stream-ordered
// transport (RCCL) nothing here blocks the host, so the sequence can be captured into a hipGraph.
static int enqueue_sharded_iteration(gbp_ctx* c, const SweepArgs& a) {
  if (c->profile_stages) {
    if (c->pending_sweep_ev.size() >= 256) drain_sweep_events(c);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(c, hipEventCreate(&e0));
    if (hipError_t e_ = hipEventCreate(&e1); e_ != hipSuccess) {
      (void)hipEventDestroy(e0);
      return fail(c, GBP_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e_));
    }
    c->pending_sweep_ev.emplace_back(e0, e1);
    HIPCHK(c, hipEventRecord(e0, c->stream));
    launch_sweep(a, c->n_tiles, c->hoist, c->stream);
    HIPCHK(c, hipEventRecord(e1, c->stream));
  } else {
    launch_sweep(a, c->n_tiles, c->hoist, c->stream);
  }
  // The communication stream (highest priority) takes the whole camera side of the exchange — local partial sums, then the
  // all-gather — right after the sweep; the landmark half of the belief update runs beside it on the main stream.
  const bool ordered = c->comm->stream_ordered() && !c->comm_single_stream;
  hipEvent_t x0 = nullptr, x1 = nullptr;     // profiling: how long the camera side of the exchange takes on its stream
  if (c->profile_stages && c->comm->stream_ordered()) {
    HIPCHK(c, hipEventCreate(&x0));
    if (hipError_t e_ = hipEventCreate(&x1); e_ != hipSuccess) { (void)hipEventDestroy(x0); return fail(c, GBP_ERR_HIP, "hipEventCreate"); }
    c->pending_exch_ev.emplace_back(x0, x1);
  }
  if (ordered) {
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_fork, 0));
    if (x0) HIPCHK(c, hipEventRecord(x0, c->comm_stream));
    enqueue_cam_partials(c, static_cast<float*>(c->send_dev), c->comm_stream);
    COMMCHK(c, c->comm->all_gather(static_cast<const float*>(c->send_dev), static_cast<float*>(c->recv_dev),
                                   (size_t)c->C * kCamRec, c->comm_stream, e_));
    if (x1) HIPCHK(c, hipEventRecord(x1, c->comm_stream));
    HIPCHK(c, hipEventRecord(c->ev_join, c->comm_stream));
  }
  if (ordered) {
    {  // the landmark half needs nothing from other ranks: it runs beside the all-gather
      BeliefArgs b = belief_args(c);
      b.roll = 1;
      launch_beliefs(b, false, true, c->stream);
    }
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  } else {
    // ONE stream: the local camera partial sums ride in the launch of the landmark half (camera blocks write send_dev and
    // leave, landmark blocks do the rank-local belief update), so an iteration is sweep -> beliefs -> all-gather -> combine:
    // three kernels and one collective.  Still no host wait with a stream-ordered transport.
    {
      BeliefArgs b = belief_args(c);
      b.cam_local = static_cast<float*>(c->send_dev);
      b.partial_only = 1;
      b.roll = 1;
      launch_beliefs(b, true, true, c->stream);
    }
    if (x0) HIPCHK(c, hipEventRecord(x0, c->stream));
    if (int rc = exchange_now(c)) return rc;
    if (x1) HIPCHK(c, hipEventRecord(x1, c->stream));
  }
  {
    BeliefArgs b = belief_args(c);
    b.gathered = static_cast<const float*>(c->recv_dev);
    b.roll = 1;
    launch_beliefs(b, true, false, c->stream);
  }
  HIPCHK(c, hipGetLastError());
  return GBP_OK;
}

int iterate_sharded(gbp_ctx* c, int n) {
  const SweepArgs a = sweep_args(c);
  gbp_ctx::Span sp{};
  if (int rc = span_begin(c, sp)) return rc;
  int left = n;
  // Measured (config-5 shard shape, 1-rank communicator): direct launches 0.186 ms per iteration, the captured graph with
  // its cross-stream fork/join nodes 0.191 ms — the path is not host-bound, so the graph is opt-in (graph_unroll > 0).
  const bool can_graph = c->comm->stream_ordered() && !c->profile_stages && c->sharded_graph && !c->graph_failed &&
                         c->stream == c->own_stream;
  // RCCL sets itself up lazily (channels, proxy threads): a few direct iterations must have run before a capture
  while (left > 0 && (!can_graph || c->comm_warm < 3 || left < c->prm.graph_unroll)) {
    if (int rc = enqueue_sharded_iteration(c, a)) return rc;
    c->comm_warm++;
    --left;
  }
  if (left >= c->prm.graph_unroll && can_graph) {
    if (!c->graph_exec) {
      HIPCHK(c, hipStreamSynchronize(c->stream));
      hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed);
      int rc = GBP_OK;
      if (e == hipSuccess) {
        for (int i = 0; i < c->prm.graph_unroll && rc == GBP_OK; ++i) rc = enqueue_sharded_iteration(c, a);
        e = hipStreamEndCapture(c->stream, &c->graph);
        if (e == hipSuccess && rc == GBP_OK) e = hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0);
      }
      if (e != hipSuccess || rc != GBP_OK) {     // no graph for this ctx: direct launches (identical results)
        (void)hipGetLastError();
        drop_graph(c);
        c->graph_failed = true;
      } else {
        c->graph_iters = c->prm.graph_unroll;
      }
    }
    while (c->graph_exec && left >= c->graph_iters) {
      HIPCHK(c, hipGraphLaunch(c->graph_exec, c->stream));
      left -= c->graph_iters;
    }
  }
  for (; left > 0; --left)
    if (int rc = enqueue_sharded_iteration(c, a)) return rc;
  if (int rc = span_end(c, sp)) return rc;
  if (!c->comm->stream_ordered() || c->profile_stages) HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!c->profile_stages) c->timed_iters += (uint64_t)n;    // with profiling the sweep brackets count the iterations
  c->beliefs_valid = true;
  return GBP_OK;
}

}  // namespace api
}  // namespace gbp

// ---- communicator: the exchange step owned by the library (RCCL over xGMI from the C++ host) ------------------------
static int comm_attach(gbp_ctx* c, gbp::Comm* comm) {
  if (int rc = settle(c)) { delete comm; return rc; }
  c->comm = comm;
  // A second HSA queue makes every dispatch of the main queue slower (measured: +10 us per sharded iteration on the
  // config-5 shard shape, 0.183 vs 0.174 ms with a 1-rank communicator), so overlapping the all-gather with the
  // landmark beliefs (~20 us of cover) only pays once the all-gather itself takes longer than that: 4 ranks and more.
  // GBP_COMM_SINGLE_STREAM=0/1 overrides the rule (measurements).
  const char* ss = std::getenv("GBP_COMM_SINGLE_STREAM");
  c->comm_single_stream = ss ? ss[0] == '1' : c->world <= 2;
  drop_graph(c);
  // IN-PLACE all-gather: this rank's partial sums are written straight into its own slot of the gathered buffer
  // (sendbuff == recvbuff + rank * count is the in-place form of ncclAllGather: the collective then moves only what comes from
  // other ranks — with a 1-rank communicator nothing at all, where the out-of-place form was a 5 us copy kernel per iteration,
  // profiles/r06_sharded_timeline.md).  (zero-filled on the ctx's stream, in front of everything that uses it)
  if (!c->xrecv.p)
    if (int rc = dev_alloc(c, c->xrecv, (size_t)c->world * c->C * kCamRec * 4)) return rc;
  c->recv_dev = c->xrecv.p;
  c->send_dev = static_cast<float*>(c->xrecv.p) + (size_t)c->rank * c->C * kCamRec;
  if (!c->comm_stream && !c->comm_single_stream) {
    // highest priority: the all-gather is issued while the landmark half of k_beliefs fills the GPU; it must not queue
    // behind those blocks (the camera combine of every rank waits for it)
    int least = 0, greatest = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(c, hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, greatest));
  }
  if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  if (!c->ev_join) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  c->comm_warm = 0;
  return GBP_OK;
}

GBP_EXPORT_T(int, 0, gbp_device_count, (void), ()) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

GBP_EXPORT(gbp_set_device, nullptr, (int device), (device)) {
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, GBP_ERR_NO_DEVICE, "gbp_set_device: no such device");
  return GBP_OK;
}

GBP_EXPORT_T(size_t, 0, gbp_comm_region_bytes, (uint32_t n_cams, int world), (n_cams, world)) {
  if (world < 1 || world > gbp::kCommMaxWorld) return 0;      // (gbp_comm_region_init refuses such a world as well)
  return gbp::comm_region_bytes(n_cams, world);
}

GBP_EXPORT(gbp_comm_region_init, nullptr, (void* region, size_t bytes, uint32_t n_cams, int world), (region, bytes, n_cams, world)) {
  if (!region) return GBP_ERR_INVALID;
  return gbp::comm_region_init(region, bytes, n_cams, world) == 0 ? GBP_OK : GBP_ERR_INVALID;
}

GBP_EXPORT_VOID(gbp_comm_region_abort, (void* region), (region)) { if (region) gbp::comm_region_abort(region); }

GBP_EXPORT(gbp_comm_region_selftest, nullptr, (void* region, int rank, int world, int rounds), (region, rank, world, rounds)) {
  if (!region || world < 1 || rank < 0 || rank >= world) return GBP_ERR_INVALID;
  std::string err;
  const int rc = gbp::comm_region_selftest(region, rank, world, rounds, err);
  if (rc != 0) return fail(nullptr, GBP_ERR_COMM, "gbp_comm_region_selftest: " + err);
  return GBP_OK;
}

GBP_EXPORT(gbp_comm_init, c, (gbp_ctx* c, void* region, int transport), (c, region, transport)) {
  if (!c || !region) return GBP_ERR_INVALID;
  if (c->comm) return fail(c, GBP_ERR_STATE, "gbp_comm_init: the ctx already has a communicator");
  std::string err;
  gbp::Comm* comm = gbp::comm_create_from_region(region, c->rank, c->world, transport, err);
  if (!comm) return fail(c, GBP_ERR_COMM, "gbp_comm_init: " + err);
  return comm_attach(c, comm);
}

GBP_EXPORT(gbp_comm_unique_id, nullptr, (void* id128), (id128)) {
  std::string err;
  if (!id128) return GBP_ERR_INVALID;
  if (gbp::comm_unique_id(id128, err) != 0) return fail(nullptr, GBP_ERR_COMM, "gbp_comm_unique_id: " + err);
  return GBP_OK;
}

GBP_EXPORT(gbp_comm_init_rccl, c, (gbp_ctx* c, const void* id128), (c, id128)) {
  if (!c || !id128) return GBP_ERR_INVALID;
  if (c->comm) return fail(c, GBP_ERR_STATE, "gbp_comm_init_rccl: the ctx already has a communicator");
  std::string err;
  gbp::Comm* comm = gbp::comm_create_rccl(id128, c->rank, c->world, err);
  if (!comm) return fail(c, GBP_ERR_COMM, "gbp_comm_init_rccl: " + err);
  return comm_attach(c, comm);
}

// ---- what a first multi-GPU run wants on record (bench.py preflight) ---------------------------------------------------
GBP_EXPORT(gbp_comm_describe, c, (gbp_ctx* c, char* buf, size_t cap), (c, buf, cap)) {
  if (!c || !buf || cap == 0) return GBP_ERR_INVALID;
  int dev = 0;
  char bus[64] = {0};
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetPCIBusId(bus, sizeof(bus), dev);
  const std::string lib = c->comm ? c->comm->library_path() : "";
  std::snprintf(buf, cap, "{\"rank\": %d, \"world\": %d, \"device\": %d, \"pci_bus_id\": \"%s\", \"transport\": \"%s\", \"library\": \"%s\", "
                          "\"library_version\": %d, \"two_streams\": %s}",
                c->rank, c->world, dev, bus, c->comm ? c->comm->name() : "none", lib.c_str(), c->comm ? c->comm->library_version() : 0,
                (c->comm && !c->comm_single_stream) ? "true" : "false");
  return GBP_OK;
}

// 0: the whole sharded iteration on ONE stream (sweep, beliefs + local partials, all-gather, combine); 1: the camera side of
// the exchange (local partials, all-gather) on a second, highest-priority stream beside the landmark beliefs.  Results are
// identical; which is faster depends on what the all-gather costs against ~10 us of second-queue overhead — measure
// (bench.py does, 20 iterations each) instead of guessing.
GBP_EXPORT(gbp_comm_set_schedule, c, (gbp_ctx* c, int two_streams), (c, two_streams)) {
  if (!c || !c->comm) return fail(c, GBP_ERR_STATE, "gbp_comm_set_schedule: the ctx has no communicator");
  if (int rc = settle(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));
  drop_graph(c);
  c->comm_single_stream = two_streams == 0;
  if (!c->comm_stream && !c->comm_single_stream) {
    int least = 0, greatest = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(c, hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, greatest));
  }
  return GBP_OK;
}

// `reps` all-gathers of the camera partial buffers back to back on the ctx's stream (collective: every rank calls it);
// *avg_us = mean duration between two events.  The buffers keep their content (the gather of the same partials).
GBP_EXPORT(gbp_comm_probe, c, (gbp_ctx* c, int reps, double* avg_us), (c, reps, avg_us)) {
  if (!c || !c->comm || !avg_us || reps < 1) return fail(c, GBP_ERR_STATE, "gbp_comm_probe: needs a communicator, reps >= 1");
  if (!c->comm->stream_ordered()) {      // host-staged: wall clock around blocking exchanges
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i)
      if (int rc = exchange_now(c)) return rc;
    *avg_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    return GBP_OK;
  }
  if (int rc = exchange_now(c)) return rc;     // warm (lazy channel set-up)
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  for (int i = 0; i < reps; ++i)
    if (int rc = exchange_now(c)) return rc;
  HIPCHK(c, hipEventRecord(c->ev2, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev2));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev1, c->ev2));
  *avg_us = 1e3 * ms / reps;
  return GBP_OK;
}

GBP_EXPORT_T(const char*, "none", gbp_comm_transport, (const gbp_ctx* c), (c)) { return (c && c->comm) ? c->comm->name() : "none"; }

GBP_EXPORT(gbp_comm_barrier, c, (gbp_ctx* c), (c)) {
  if (!c || !c->comm) return GBP_ERR_STATE;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  COMMCHK(c, c->comm->barrier(e_));
  return GBP_OK;
}

// gbp_eval over ALL shards: local sums gathered over the ranks and added in rank order (same bits on every rank)
GBP_EXPORT(gbp_eval_global, c, (gbp_ctx* c, gbp_eval_out* o), (c, o)) {
  if (!c || !o) return GBP_ERR_INVALID;
  const int rc = eval(c, o);
  if (rc != GBP_OK || !c->comm) return rc;
  const double mine[7] = {o->sum_norm, o->sum_half_sq, (double)o->n_active, (double)o->n_relin, (double)o->n_robust,
                          (double)o->n_nonfinite, (double)o->n_nonpd};
  double all[7 * gbp::kCommMaxWorld];
  COMMCHK(c, c->comm->all_gather_host(mine, all, 7, e_));
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < c->world; ++r)
    for (int i = 0; i < 7; ++i) acc[i] = acc[i] + all[r * 7 + i];
  o->sum_norm = acc[0]; o->sum_half_sq = acc[1]; o->n_active = (uint64_t)(acc[2] + 0.5); o->n_relin = (uint64_t)(acc[3] + 0.5);
  o->n_robust = (uint64_t)(acc[4] + 0.5); o->n_nonfinite = (uint64_t)(acc[5] + 0.5); o->n_nonpd = (uint64_t)(acc[6] + 0.5);
  return GBP_OK;
}
