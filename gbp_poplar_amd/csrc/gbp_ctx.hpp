// gbp_ctx.hpp — the ONE internal header of the C-ABI implementation (include/gbp_mi355x*.h): the context, the helpers every
// translation unit shares, and the macro through which every exported function is defined.
//
//   gbp_api_ctx.cpp      life cycle + the host-stream programs: gbp_create / gbp_destroy, WRITE, READ, READ_PRIORS, NEW_KEYFRAME,
//                        gbp_sync, gbp_set_stream, gbp_timing                                   (ba.cpp:659-937, slam.cpp:913-928)
//   gbp_api_launch.cpp   what a program launches: LINEARISE, GBP_PROG (hipGraph replay / direct launches), WEAKEN_PRIORS, the
//                        split-phase iteration                                                 (ba.cpp:863-865,890-905)
//   gbp_api_persist.cpp  the persistent kernel's launches: co-residency probe, snapshot, log of unvalidated launches, recovery
//   gbp_api_eval.cpp     the metric on the device (util.cpp:74-144) and the loops that carry it: gbp_eval*, gbp_iterate_eval*,
//                        gbp_ba_loop                                                           (ba.cpp:1001-1053)
//   gbp_api_comm.cpp     sharded ctx: communicator glue, the sharded iteration, gbp_eval_global  (ba.cpp:414-417,617-649)
//   gbp_api_debug.cpp    the test hooks of include/gbp_mi355x_debug.h (test-hooks builds only)
//
// Replaces the Poplar graph / compute-set wiring of the reference (ba/ba.cpp:45-371, 659-937): where the reference maps vertices
// to IPU tiles and connects tensor slices, gbp_create sorts factors into device order (gbp_layout.cpp), builds the tile-coalesced
// arrays the kernels stream, and the launch TUs record the sequence of one iteration as a hipGraph.
#pragma once
#include "../../include/gbp_mi355x.h"
#include "../../include/gbp_mi355x_multi.h"
#include "../../include/gbp_mi355x_compat.h"
#ifdef GBP_BUILD_TEST_HOOKS
#include "../../include/gbp_mi355x_debug.h"
#endif
#include "gbp_comm.hpp"
#include "gbp_export.hpp"
#include "gbp_kernels.h"
#include "gbp_layout.hpp"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace gbp {
namespace api {
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};
}  // namespace api
}  // namespace gbp

struct gbp_ctx {
  uint32_t C = 0, L = 0, E = 0;       // global sizes
  uint32_t lmk_begin = 0, lmk_end = 0, L_loc = 0, E_loc = 0;
  int rank = 0, world = 1;
  float K[9];
  gbp_params prm;
  gbp::Layout lay;                         // device order (host side): position <-> file edge, rows, slots, tile order
  uint32_t Ep = 0, n_tiles = 0, n_rows = 0;
  // device memory
  using DevBuf = gbp::api::DevBuf;
  std::vector<DevBuf*> all;
  DevBuf row_cam, lmk_idx, fac, cmsg, mu, lmsg, camb, camp, lmkb, lmkp, rowp, local, d_cam_row_ptr, d_row_slot, d_lmk_ptr, cwf, lwf,
      cscale, lscale, cam_mu, lmk_mu, dK, evalp, hmu_c, hmu_l, clin, d_lmk_fpos, d_lmk_ix, health, tile_perm;
  // host -> device copies of small graphs go through this pinned, device-mapped buffer and a copy kernel (gbp_api_ctx.cpp: H2D)
  void* stage_host = nullptr;
  void* stage_dev = nullptr;
  size_t stage_cap = 0;
  size_t stage_hint = 0;               // what gbp_upload will stage: the buffer is pinned once, at that size (gbp_create)
  DevBuf idx_arena;                    // the index arrays of the device order (row_cam, lmk_idx, lmk_fpos, lmk_ix, the row / landmark pointers, row_slot, K, tile_perm: views into it)
  DevBuf st_a, st_b;                   // [Ep] scratch of the per-factor state get / set kernels
  std::vector<uint8_t> active_host;    // [Ep] host shadow of the active flags (hoist guard of gbp_new_keyframe)
  bool use_tile_perm = false;
  DevBuf seg_live;                     // [n_tiles] which 64-byte segments of a tile hold a factor (view into idx_arena; SweepArgs.seg_live)
  bool use_seg_live = false;
  uint32_t sweep_policy = 0;           // kPol* bits of SweepArgs.policy for this graph's shape (sweep_policy_for)
  bool hoist = true;  // per-variable belief means (k_sweep<true>); false = literal per-factor mu/oldmu tensors
  void* send_dev = nullptr;
  void* recv_dev = nullptr;
  // library-owned exchange (gbp_comm_init*): communicator, buffers, a second stream so the all-gather overlaps the
  // rank-local landmark half of the belief update (fork / join through two events: capturable into the hipGraph)
  gbp::Comm* comm = nullptr;
  DevBuf xrecv;                        // [world][C][44] gathered camera partials; this rank writes its own slot (in-place all-gather)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int comm_warm = 0;                   // sharded iterations run directly so far (RCCL must have run before a capture)
  bool comm_single_stream = false;     // all-gather on the main stream, no second queue (default for world <= 2)
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  int graph_iters = 0;
  bool graph_failed = false;           // a capture / instantiation failed once: direct launches from then on
  // the same for iterations that carry the metric (k_sweep<EV> + k_beliefs<EV>: gbp_iterate_eval_each beyond k_persist)
  hipGraph_t graph_ev = nullptr;
  hipGraphExec_t graph_exec_ev = nullptr;
  DevBuf ev_cam, ev_lmk, ev_part, ev_ctl;   // metric records of the belief owners, ring of per-tile partial sums, counter + health words
  uint32_t ev_depth = 0;               // slots of the ring = iterations per piece of a burst
  void* ev_host = nullptr;             // pinned + device-mapped: one gbp_eval_out per iteration of a burst (k_eval_fold)
  size_t ev_host_cap = 0;
  void* ev_host_dev = nullptr;
  bool sharded_graph = false;          // gbp_params.graph_unroll > 0 was asked for explicitly (see iterate_sharded)
  bool uploaded = false, beliefs_valid = false;
  bool lmk_half_done = false;          // gbp_iterate_local already refreshed the landmark beliefs of this iteration
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  // gbp_iterate does not block the host: each call is bracketed by an event pair that is read later (gbp_timing, or
  // when the ring is full), so a caller that evaluates the metric every iteration pays ONE host synchronisation per
  // iteration (inside gbp_eval), not two
  struct Span { hipEvent_t a, b; };
  std::vector<Span> spans;             // recorded, not yet read
  std::vector<Span> span_pool;         // reusable event pairs
  void* eval_host = nullptr;           // pinned + device-mapped: k_eval writes the metric partials + health counters here
  void* eval_host_dev = nullptr;
  void* series_host = nullptr;         // gbp_iterate_eval_each: [kSeriesMax metrics][1 + workgroups] slots, same kind of memory
  void* series_dev = nullptr;
  int eval_parity = 0, eval_pending = 0;
  bool eval_per_wave[2] = {false, false};   // result area written by k_persist (one record per tile wave) or by k_eval (one per workgroup)
  hipEvent_t eval_ev[2] = {nullptr, nullptr};
  bool profile_stages = false;
  // k_persist (small graphs): n iterations in one launch
  bool persist_ok = false;             // bursts run inside k_persist (eligible, co-resident, no time-out since the last upload)
  bool persist_eligible = false;       // what persist_ok returns to at the next gbp_upload after a recovered time-out
  bool persist_coop = false;           // launched with hipLaunchCooperativeKernel (co-residency guaranteed by the runtime / driver)
  // A launch whose barrier timed out (workgroups not co-resident: e.g. another process holds CUs) is UNDONE and replayed on the
  // two-kernel path: every launch is preceded by a snapshot of the arrays it mutates (one copy kernel, skipped once the abort
  // word is set), later launches of the ctx return at once, and the host — at the next point where it synchronises anyway —
  // restores the snapshot and replays the logged launches from the first failed one on.
  // mode 0 = gbp_iterate, 1 = gbp_iterate_eval (metric in eval area `area`), 2 = eval_each / gbp_ba_loop with metrics (blocking);
  // w_steps2 != 0: the launch weakens priors itself (gbp_ba_loop: loop index of its first iteration, twice the --steps)
  struct Burst { unsigned seq; int n; int mode; int area; unsigned w_first = 0, w_steps2 = 0; };
  DevBuf pflow;                        // tagged shadows of k_persist_flow (PersistFlow), one allocation
  gbp::PersistFlow flow{};                  // the tagged shadows of k_persist_flow
  uint32_t flow_total4 = 0;            // test-hooks builds: float4 of the shadows (the redundant copies of gbp_debug_persist_verify follow them)
  bool persist_flow = true;            // test-hooks build: gbp_debug_persist_flow(ctx, 0) / GBP_PERSIST_FLOW=0 run the barrier kernel of rounds 3-4 instead
  std::vector<Burst> persist_log;      // launched, completion not yet validated
  unsigned persist_seq = 0;
  DevBuf psnap;                        // snapshot arena
  gbp::CopySegs snap_save{}, snap_restore{};
  std::string warn;                    // text of the last recovered incident (also left in `err`, the call returns GBP_OK)
  uint64_t persist_recoveries = 0;
  double probe_ms = 0;                 // gbp_create: what the co-residency probe took (the first kernel launch of the process: code-object load included)
  DevBuf psync;                        // barrier words
  void* pstatus_host = nullptr;        // pinned + device-mapped: raised by the kernel if a barrier gave up
  void* pstatus_dev = nullptr;
  uint64_t persist_launches = 0;
  unsigned persist_epoch_base = 0;     // arrivals the barrier counter has seen so far (it keeps counting across launches)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_sweep_ev;  // split-phase profiling: brackets not yet read
  double sweep_ms = 0, belief_ms = 0, total_ms = 0, exchange_ms = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_exch_ev;     // profiling: brackets of partials + all-gather
  uint64_t timed_iters = 0, dev_bytes = 0;
  std::string err;
};

namespace gbp {
namespace api {

// ---- errors -----------------------------------------------------------------------------------------------------------------
// (fail(), guarded() and the GBP_EXPORT macros: gbp_export.hpp)

#define HIPCHK(ctx, expr)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return ::gbp::api::fail(ctx, GBP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

#define COMMCHK(ctx, expr)                                                       \
  do {                                                                           \
    std::string e_;                                                              \
    if ((expr) != 0) return ::gbp::api::fail(ctx, GBP_ERR_COMM, "exchange: " + e_); \
  } while (0)

// ---- gbp_api_ctx.cpp --------------------------------------------------------------------------------------------------------
std::string& create_error();                                 // thread-local text of the last failed gbp_create / ctx-less call
int dev_alloc(gbp_ctx* c, DevBuf& b, size_t bytes);         // zero-filled device memory owned by the ctx (fill ordered on c->stream)
template <class T> inline T* P(DevBuf& b) { return static_cast<T*>(b.p); }
// does this ctx combine camera partials through exchange buffers (sharded, or a 1-rank communicator)?
inline bool exch(const gbp_ctx* c) { return c->world > 1 || c->comm != nullptr; }
// The host -> device copies of ONE call (gbp_create, gbp_upload, gbp_new_keyframe).  The first small and the first medium-sized
// hipMemcpy of a process cost 11 - 15 ms and 7 - 9 ms on this stack (set-up of two copy paths inside the runtime, profiles/exp_first_copy.hip)
// — more than the 1 500 iterations of a `ba fr1xyz` run.  A call that moves at most kStageMax bytes therefore stages its pieces in one
// pinned, device-mapped buffer and moves them with k_copy_segments on the ctx's stream (first launch 0.4 ms, then PCIe speed); larger
// calls use hipMemcpy, where those milliseconds do not matter.  put() copies out of `src` before it returns; end() returns with
// everything on the device (it synchronises the ctx's stream).
struct H2D {
  static constexpr size_t kStageMax = (size_t)32 << 20;
  gbp_ctx* c = nullptr;
  bool direct = true;
  size_t used = 0;
  CopySegs segs{};
  int begin(gbp_ctx* ctx, size_t total_bytes, int pieces);
  int put(void* dst_dev, const void* src, size_t bytes);      // dst: a whole DevBuf or a 16-byte aligned piece whose padded size is inside one
  int stage(const void* src, size_t bytes, const void** dev);  // staged only (!direct): the bytes where a kernel of the caller's reads them
  int end();
  int flush();
};
// ... and the device -> host copies of one call (gbp_read, gbp_read_priors), the same way round: get() queues, end() returns with every
// destination filled (the first device -> host hipMemcpy of a process pays the same set-up when no host -> device one came before it)
struct D2H {
  struct Out { void* dst; size_t off, bytes; };
  H2D up;
  std::vector<Out> pending;
  int begin(gbp_ctx* ctx, size_t total_bytes, int pieces);
  int get(void* dst_host, const void* src_dev, size_t bytes);
  int end();
};
// timing brackets of the iterate calls (read later: gbp_timing, or when the ring is full)
int resolve_spans(gbp_ctx* c, bool wait);
int span_begin(gbp_ctx* c, gbp_ctx::Span& sp);
int span_end(gbp_ctx* c, const gbp_ctx::Span& sp);
void drain_sweep_events(gbp_ctx* c);
// a pair of timing events for per-stage profiling: created (or fails with the ctx's error text)
int event_pair(gbp_ctx* c, hipEvent_t* a, hipEvent_t* b);
// The construction knobs of the device order and the cache policy of the sweep: the defaults are the product.  Only the
// test-hooks build can change them (gbp_debug_layout_options / gbp_debug_force_sweep_policy: measurements and tests).
extern LayoutOptions g_layout_options;
extern int g_force_sweep_policy;
extern int g_force_seg_skip;      // -1: by shape; 0 / 1: never / always skip the all-pad segments in the sweep (gbp_debug_force_seg_skip)

// Per-factor scalar state lives in the pad slots of the LMSG records (gbp_kernels.h): {damping, count, flags, variance} per
// device position, as the host code sees it.
struct HostState { float damping; int32_t count; uint32_t flags; float var; };
inline HostState get_state(const std::vector<float>& rec, size_t p) {
  HostState h;
  int32_t packed;
  std::memcpy(&packed, &rec[p * 16 + 13], 4);
  h.damping = rec[p * 16 + 3]; h.count = packed >> 3; h.flags = (uint32_t)packed & 7u; h.var = rec[p * 16 + 14];
  return h;
}
inline void put_state(std::vector<float>& rec, size_t p, const HostState& h) {
  const int32_t packed = (int32_t)(((uint32_t)h.count << 3) | (h.flags & 7u));
  rec[p * 16 + 3] = h.damping;
  std::memcpy(&rec[p * 16 + 13], &packed, 4);
  rec[p * 16 + 14] = h.var;
}
inline size_t tile_off(uint32_t p, int G, int f) {  // float offset of float f of position p in a G-group tiled array
  return (((size_t)(p >> 6) * G + (f >> 2)) * 64 + (p & 63)) * 4 + (f & 3);
}

// ---- gbp_api_launch.cpp -----------------------------------------------------------------------------------------------------
SweepArgs sweep_args(gbp_ctx* c);
BeliefArgs belief_args(gbp_ctx* c);
void drop_graph(gbp_ctx* c);
// camera beliefs from stored partials (single GPU: the local sums; sharded: recv_dev) + landmark beliefs re-summed.  roll = true
// at the end of an iteration, false for prior-only refreshes (WEAKEN_PRIORS, NEW_KEYFRAME, LINEARISE).
int refresh_beliefs_from_partials(gbp_ctx* c, bool roll, bool do_lmk = true, bool weaken = false);
void enqueue_iteration(gbp_ctx* c, const SweepArgs& a, bool ev = false, bool weaken_after = false);
void enqueue_cam_partials(gbp_ctx* c, float* dst, hipStream_t s = nullptr);
int iterate_plain(gbp_ctx* c, const SweepArgs& a, int n, bool ev = false);                      // hipGraph replays + remainder
int iterate_weaken_plain(gbp_ctx* c, const SweepArgs& a, int n, unsigned i0, unsigned steps2);  // the same with the loop's weakenings riding
int iterate(gbp_ctx* c, int n);                              // GBP_PROG x n: persistent kernel / hipGraph / sharded
int weaken_priors(gbp_ctx* c);                               // WEAKEN_PRIORS

// ---- gbp_api_persist.cpp ----------------------------------------------------------------------------------------------------
constexpr int kPersistChunk = 4096;      // iterations per k_persist launch (a launch cannot be pre-empted: ~60 ms at 15 us each)
constexpr int kNotLaunched = 1;          // launch_persist_burst: nothing ran, the ctx has left the persistent path
int persist_setup(gbp_ctx* c, const gbp_params* prm, bool sharded);   // gbp_create: probe, snapshot arena, tagged shadows (or none: not eligible)
void persist_forget(gbp_ctx* c);                            // gbp_destroy: its launches have ended, nobody waits for this ctx again
int persist_reset(gbp_ctx* c);                              // gbp_upload: a fresh start for the persistent path
// Every entry point that enqueues other device work, or changes what the launches in flight depend on, first makes sure they
// completed without a time-out (and repairs the state if one did): free when nothing is in flight.
int settle(gbp_ctx* c);
int persist_check(gbp_ctx* c, unsigned upto);               // the stream has been synchronised / an event behind launch `upto` completed
bool stream_is_capturing(gbp_ctx* c);
int persist_ready(gbp_ctx* c, bool* yes);                   // may the next burst run inside the persistent kernel?
int launch_persist_burst(gbp_ctx* c, const SweepArgs& a, int n, const PersistEval* ev, int mode, int area,
                         uint32_t w_first = 0, uint32_t w_steps2 = 0);

// ---- gbp_api_eval.cpp -------------------------------------------------------------------------------------------------------
int eval_enqueue(gbp_ctx* c, int area);                     // k_means + k_eval of the current beliefs into result area `area`
int eval_begin(gbp_ctx* c);
int eval_end(gbp_ctx* c, gbp_eval_out* o);
int eval(gbp_ctx* c, gbp_eval_out* o);

// ---- gbp_api_comm.cpp -------------------------------------------------------------------------------------------------------
int exchange_now(gbp_ctx* c);                               // plain all-gather of the camera partials on the ctx's stream
int iterate_sharded(gbp_ctx* c, int n);

}  // namespace api
}  // namespace gbp
