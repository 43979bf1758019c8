// gbp_capi.cpp — C-ABI implementation (include/gbp_mi355x.h): context, HBM layout construction,
// program list (WRITE / LINEARISE / GBP / WEAKEN_PRIORS / READ / READ_PRIORS / NEW_KEYFRAME),
// hipGraph capture of the iteration, split-phase multi-GPU iteration.
//
// Replaces the Poplar graph/compute-set wiring of the reference (ba/ba.cpp:45-371, 659-937): where
// the reference maps vertices to IPU tiles and connects tensor slices, this file sorts factors
// into device order, builds the tile-coalesced arrays the kernels stream, and records the launch
// sequence of one iteration as a hipGraph.
#include "../../include/gbp_mi355x.h"
#ifdef GBP_BUILD_TEST_HOOKS
#include "../../include/gbp_mi355x_debug.h"
#endif
#include "gbp_comm.hpp"
#include "gbp_kernels.h"
#include "gbp_layout.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

using namespace gbp;

namespace {
thread_local std::string g_create_error;
// Serialisation of k_persist launches across the ctxs / streams of a process (launch_persist_burst): a library-owned event
// per device is recorded behind every launch, the next launch from another ctx or stream waits for it.  No stream handle of
// another ctx is ever touched (it may have been destroyed by its owner); the two "last" words are compared, never used.
std::mutex g_persist_mu;
hipEvent_t g_persist_event[16] = {};
const void* g_persist_last_ctx[16] = {};
const void* g_persist_last_stream[16] = {};
constexpr int kPersistChunk = 4096;      // iterations per k_persist launch (a launch cannot be pre-empted: ~60 ms at 15 us each)
constexpr size_t kPersistLogMax = 8;     // launches in flight without a validated completion

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};
static_assert(kLayoutTile == (uint32_t)kTile && kLayoutRow == (uint32_t)kRow, "gbp_layout.hpp and gbp_kernels.h disagree on the tile shape");
// The construction knobs of the device order and the cache policy of the sweep: the defaults are the product.  Only the
// test-hooks build can change them (gbp_debug_layout_options / gbp_debug_force_sweep_policy: measurements and tests).
LayoutOptions g_layout_options;
int g_force_sweep_policy = -1;

// Cache policy of the sweep's two message streams for a graph of C cameras and n_tiles tiles (SweepArgs.policy).
// Camera messages loaded with the default policy instead of the non-temporal hint: few cameras (their belief table small beside
// an XCD's 4 MiB L2) AND both message streams of this rank (176 B per factor slot) within ~3/4 of the 256 MiB Infinity Cache,
// where the lines loaded this sweep are still found by the next: measured on 1 M factors x 100 000 landmarks +1.3 % iterations/s
// with 500 cameras, +0.6 % with 1 000, +0.25 % with 2 000, -0.1 % with 4 000, -0.3 % with 8 000 (-1.4 % on the config-5 shard
// shape); 1 000 cameras, factor count scanned: +1.4 % at 0.5 M, +0...2 % at 1 M (the edge), -0.5 % at 1.25 M, -2 % at 1.5 M,
// -5.5 % at 2 M (profiles/r04_alu_diet.md section 6).
uint32_t sweep_policy_for(uint32_t C, uint32_t n_tiles) {
  uint32_t pol = 0;
  if (C <= 2048u && (uint64_t)n_tiles * 64u * 176u <= 200000000ull) pol |= kPolCmsgLoadCached;
  return pol;
}
}  // namespace

struct gbp_ctx {
  uint32_t C = 0, L = 0, E = 0;       // global sizes
  uint32_t lmk_begin = 0, lmk_end = 0, L_loc = 0, E_loc = 0;
  int rank = 0, world = 1;
  float K[9];
  gbp_params prm;
  Layout lay;                         // device order (host side): position <-> file edge, rows, slots, tile order
  uint32_t Ep = 0, n_tiles = 0, n_rows = 0;
  // device memory
  std::vector<DevBuf*> all;
  DevBuf row_cam, lmk_idx, fac, cmsg, mu, lmsg, camb, camp, lmkb, lmkp, rowp, local, d_cam_row_ptr, d_row_slot, d_lmk_ptr, cwf, lwf,
      cscale, lscale, cam_mu, lmk_mu, dK, evalp, hmu_c, hmu_l, clin, d_lmk_fpos, d_lmk_ix, health, tile_perm;
  DevBuf st_a, st_b;                   // [Ep] scratch of the per-factor state get / set kernels
  std::vector<uint8_t> active_host;    // [Ep] host shadow of the active flags (hoist guard of gbp_new_keyframe)
  bool use_tile_perm = false;
  uint32_t sweep_policy = 0;           // kPol* bits of SweepArgs.policy for this graph's shape (sweep_policy_for)
  bool hoist = true;  // per-variable belief means (k_sweep<true>); false = literal per-factor mu/oldmu tensors
  void* send_dev = nullptr;
  void* recv_dev = nullptr;
  // library-owned exchange (gbp_comm_init*): communicator, buffers, a second stream so the all-gather overlaps the
  // rank-local landmark half of the belief update (fork / join through two events: capturable into the hipGraph)
  gbp::Comm* comm = nullptr;
  DevBuf xsend, xrecv;
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
  PersistFlow flow{};                  // the tagged shadows of k_persist_flow
  bool persist_flow = true;            // test-hooks build: gbp_debug_persist_flow(ctx, 0) / GBP_PERSIST_FLOW=0 run the barrier kernel of rounds 3-4 instead
  std::vector<Burst> persist_log;      // launched, completion not yet validated
  unsigned persist_seq = 0;
  DevBuf psnap;                        // snapshot arena
  CopySegs snap_save{}, snap_restore{};
  std::string warn;                    // text of the last recovered incident (also left in `err`, the call returns GBP_OK)
  uint64_t persist_recoveries = 0;
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

namespace {

int fail(gbp_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}

#define HIPCHK(ctx, expr)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(ctx, GBP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
  } while (0)

int dev_alloc(gbp_ctx* c, DevBuf& b, size_t bytes) {
  b.bytes = bytes < 256 ? 256 : bytes;   // >= one camera / landmark record: pad lanes of an empty shard read index 0
  HIPCHK(c, hipMalloc(&b.p, b.bytes));
  HIPCHK(c, hipMemset(b.p, 0, b.bytes));
  c->dev_bytes += b.bytes;
  c->all.push_back(&b);
  return GBP_OK;
}

template <class T> T* P(DevBuf& b) { return static_cast<T*>(b.p); }

SweepArgs sweep_args(gbp_ctx* c) {
  SweepArgs a;
  a.row_cam = P<uint32_t>(c->row_cam); a.lmk_idx = P<uint32_t>(c->lmk_idx); a.fac = P<float4>(c->fac); a.cmsg = P<float4>(c->cmsg);
  a.mu = P<float4>(c->mu); a.lmsg = P<float4>(c->lmsg); a.camb = P<float4>(c->camb); a.lmkb = P<float4>(c->lmkb);
  a.rowp = P<float4>(c->rowp);
  a.cam_mu = P<float4>(c->hmu_c); a.lmk_mu = P<float4>(c->hmu_l); a.cam_lin = P<float4>(c->clin);
  std::memcpy(a.K, c->K, sizeof(a.K));
  a.hp.maxeta_damping = c->prm.maxeta_damping; a.hp.num_undamped_iters = c->prm.num_undamped_iters;
  a.hp.dmu_threshold = c->prm.dmu_threshold; a.hp.min_linear_iters = c->prm.min_linear_iters;
  a.hp.nstds = c->prm.nstds; a.hp.relin_mode = c->prm.relin_mode;
  a.variant = c->prm.reserved[0];      // read by the experiments build only
  a.tile_perm = c->use_tile_perm ? P<uint32_t>(c->tile_perm) : nullptr;
  a.policy = c->sweep_policy;
  a.ev = EvalRide{};
  return a;
}

void drop_graph(gbp_ctx* c) {
  if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
  if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
  if (c->graph_exec_ev) { (void)hipGraphExecDestroy(c->graph_exec_ev); c->graph_exec_ev = nullptr; }
  if (c->graph_ev) { (void)hipGraphDestroy(c->graph_ev); c->graph_ev = nullptr; }
  c->graph_iters = 0;
}

BeliefArgs belief_args(gbp_ctx* c) {
  BeliefArgs b{};
  b.rowp = P<float>(c->rowp); b.cam_row_ptr = P<uint32_t>(c->d_cam_row_ptr); b.cam_prior = P<float>(c->camp);
  b.row_slot = c->lay.row_slot.empty() ? nullptr : P<uint32_t>(c->d_row_slot);
  b.cam_local = P<float>(c->local); b.gathered = nullptr; b.world = c->world;
  b.camb = P<float>(c->camb); b.cam_mu = P<float4>(c->hmu_c); b.cam_lin = P<float4>(c->clin); b.n_cams = c->C;
  b.lmk_prior = P<float4>(c->lmkp); b.lmsg = P<float4>(c->lmsg); b.lmk_ptr = P<uint32_t>(c->d_lmk_ptr);
  b.lmk_fpos = P<uint32_t>(c->d_lmk_fpos); b.lmk_ix = P<uint32_t>(c->d_lmk_ix);
  b.lmkb = P<float4>(c->lmkb); b.lmk_mu = P<float4>(c->hmu_l); b.n_lmks = c->L_loc;
  b.partial_only = 0; b.hoist = c->hoist ? 1 : 0; b.roll = 0;
  b.lmk_blocks = 0; b.lmk_xcd_order = c->prm.tile_order != 1 ? 1 : 0;
  return b;
}

// does this ctx combine camera partials through exchange buffers (sharded, or a 1-rank communicator)?
inline bool exch(const gbp_ctx* c) { return c->world > 1 || c->comm != nullptr; }

// camera beliefs from stored partials (single GPU: d_local; multi: recv_dev) + landmark beliefs re-summed.
// roll = true at the end of an iteration (the sweep has consumed the current means), false for
// prior-only refreshes (WEAKEN_PRIORS, NEW_KEYFRAME, LINEARISE).
int refresh_beliefs_from_partials(gbp_ctx* c, bool roll, bool do_lmk = true, bool weaken = false) {
  BeliefArgs b = belief_args(c);
  if (weaken) {
    b.weaken = 1;
    b.cam_prior_rw = P<float>(c->camp); b.cam_scale = P<float>(c->cscale); b.cam_wflag = P<uint32_t>(c->cwf);
    b.lmk_prior_rw = P<float4>(c->lmkp); b.lmk_scale = P<float>(c->lscale); b.lmk_wflag = P<uint32_t>(c->lwf);
  }
  if (!exch(c)) {
    b.gathered = P<float>(c->local); b.world = 1;
  } else {
    if (!c->recv_dev) return fail(c, GBP_ERR_STATE, "exchange buffers not set");
    b.gathered = static_cast<const float*>(c->recv_dev);
  }
  b.roll = roll ? 1 : 0;
  launch_beliefs(b, true, do_lmk, c->stream);
  HIPCHK(c, hipGetLastError());
  return GBP_OK;
}

// the riding metric's view of the ctx (gbp_kernels.h: EvalRide); valid once ev_alloc has run
EvalRide eval_ride(gbp_ctx* c) {
  EvalRide e{};
  e.cam_rec = P<float4>(c->ev_cam); e.lmk_mean = P<float4>(c->ev_lmk); e.part = P<EvalRec>(c->ev_part);
  e.counter = P<unsigned>(c->ev_ctl);
  e.health = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->ev_ctl.p) + 8);
  e.slot_health = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->ev_ctl.p) + 64);
  e.n_tiles = c->n_tiles; e.num_undamped = c->prm.num_undamped_iters;
  return e;
}

// one iteration: k_sweep + k_beliefs; ev: the instantiations that carry the metric (a.ev filled in by the caller)
// weaken_after: WEAKEN_PRIORS follows this iteration with nothing reading the beliefs in between — its belief update takes the
// weakened priors straight away (WeakenPriorVertex rides in k_beliefs as in gbp_weaken_priors): the same beliefs, means and
// mean changes as {k_beliefs; k_beliefs(weaken)} leave, in one launch
void enqueue_iteration(gbp_ctx* c, const SweepArgs& a, bool ev = false, bool weaken_after = false) {
  launch_sweep(a, c->n_tiles, c->hoist, c->stream, ev);
  BeliefArgs b = belief_args(c);
  b.roll = 1;
  if (weaken_after) {
    b.weaken = 1;
    b.cam_prior_rw = P<float>(c->camp); b.cam_scale = P<float>(c->cscale); b.cam_wflag = P<uint32_t>(c->cwf);
    b.lmk_prior_rw = P<float4>(c->lmkp); b.lmk_scale = P<float>(c->lscale); b.lmk_wflag = P<uint32_t>(c->lwf);
  }
  if (ev) b.ev = a.ev;
  launch_beliefs(b, true, true, c->stream, ev);
}

// local camera partials only (before an exchange / before a prior-only refresh)
void enqueue_cam_partials(gbp_ctx* c, float* dst, hipStream_t s = nullptr) {
  BeliefArgs b = belief_args(c);
  b.cam_local = dst; b.partial_only = 1;
  launch_beliefs(b, true, false, s ? s : c->stream);
}

void pack_cam(const float* eta, const float* lam, uint32_t C, std::vector<float>& out) {
  out.assign((size_t)C * kCamRec, 0.f);
  for (uint32_t c = 0; c < C; ++c) {
    std::memcpy(&out[(size_t)c * kCamRec], eta + (size_t)c * 6, 6 * 4);
    std::memcpy(&out[(size_t)c * kCamRec + 8], lam + (size_t)c * 36, 36 * 4);
  }
}
void pack_lmk(const float* eta, const float* lam, uint32_t l0, uint32_t n, std::vector<float>& out) {
  out.assign((size_t)n * 16, 0.f);
  for (uint32_t i = 0; i < n; ++i) {
    std::memcpy(&out[(size_t)i * 16], eta + (size_t)(l0 + i) * 3, 3 * 4);
    std::memcpy(&out[(size_t)i * 16 + 4], lam + (size_t)(l0 + i) * 9, 9 * 4);
  }
}

// Per-factor scalar state lives in the pad slots of the LMSG records (gbp_kernels.h): these two helpers
// expose it to the host code as {damping, count, flags, variance} per device position.
struct HostState { float damping; int32_t count; uint32_t flags; float var; };

int download_lmsg(gbp_ctx* c, std::vector<float>& rec) {
  rec.resize((size_t)c->Ep * 16);
  HIPCHK(c, hipMemcpy(rec.data(), c->lmsg.p, rec.size() * 4, hipMemcpyDeviceToHost));
  return GBP_OK;
}
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

// timing brackets of gbp_iterate calls: read the finished ones without blocking anything that is still queued
int resolve_spans(gbp_ctx* c, bool wait) {
  size_t done = 0;
  for (; done < c->spans.size(); ++done) {
    const gbp_ctx::Span sp = c->spans[done];
    if (wait) { if (hipEventSynchronize(sp.b) != hipSuccess) break; }
    else if (hipEventQuery(sp.b) != hipSuccess) break;
    float ms = 0;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) c->total_ms += ms;
    c->span_pool.push_back(sp);
  }
  c->spans.erase(c->spans.begin(), c->spans.begin() + (long)done);
  return GBP_OK;
}
int span_begin(gbp_ctx* c, gbp_ctx::Span& sp) {
  if (c->spans.size() >= 64) resolve_spans(c, true);
  if (!c->span_pool.empty()) { sp = c->span_pool.back(); c->span_pool.pop_back(); }
  else {
    HIPCHK(c, hipEventCreate(&sp.a));
    if (hipError_t e_ = hipEventCreate(&sp.b); e_ != hipSuccess) { (void)hipEventDestroy(sp.a); return fail(c, GBP_ERR_HIP, "hipEventCreate"); }
  }
  HIPCHK(c, hipEventRecord(sp.a, c->stream));
  return GBP_OK;
}
int span_end(gbp_ctx* c, const gbp_ctx::Span& sp) {
  HIPCHK(c, hipEventRecord(sp.b, c->stream));
  c->spans.push_back(sp);
  return GBP_OK;
}

// split-phase profiling: read (and free) the sweep brackets recorded by gbp_iterate_begin
void drain_sweep_events(gbp_ctx* c) {
  for (auto& pr : c->pending_sweep_ev) {
    float ms = 0;
    if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      c->sweep_ms += ms;
      c->timed_iters += 1;
    }
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  c->pending_sweep_ev.clear();
  for (auto& pr : c->pending_exch_ev) {
    float ms = 0;
    if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->exchange_ms += ms;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  c->pending_exch_ev.clear();
}

// No C++ exception crosses the C-ABI: every entry point that allocates host memory runs inside this guard.
template <class F> int guarded(gbp_ctx* c, const char* what, F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(c, GBP_ERR_NOMEM, std::string(what) + ": out of host memory");
  } catch (const std::exception& e) {
    return fail(c, GBP_ERR_INVALID, std::string(what) + ": " + e.what());
  }
}

inline size_t tile_off(uint32_t p, int G, int f) {  // float offset of float f of position p in a G-group tiled array
  return (((size_t)(p >> 6) * G + (f >> 2)) * 64 + (p & 63)) * 4 + (f & 3);
}

}  // namespace

extern "C" {

int gbp_abi_version(void) { return GBP_ABI_VERSION; }

void gbp_default_params(gbp_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->maxeta_damping = 0.4f; p->num_undamped_iters = 8; p->dmu_threshold = 3e-3f; p->min_linear_iters = 10;
  p->nstds = 2.5f; p->relin_mode = 0; p->graph_unroll = 0;
}

const char* gbp_last_error(const gbp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void gbp_destroy(gbp_ctx* c) {
  if (!c) return;
  drop_graph(c);
  if (c->persist_eligible) {   // its launches have ended before its memory goes away; the "last launcher" words are only ever compared
    (void)hipStreamSynchronize(c->stream);
    std::lock_guard<std::mutex> lock(g_persist_mu);
    for (const void*& p : g_persist_last_ctx) if (p == c) p = &g_persist_mu;      // "someone else": the next launcher waits for the device's event
  }
  if (c->comm) { (void)hipStreamSynchronize(c->stream); if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream); delete c->comm; c->comm = nullptr; }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  for (auto* v : {&c->pending_sweep_ev, &c->pending_exch_ev})
    for (auto& pr : *v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (DevBuf* b : c->all) if (b->p) (void)hipFree(b->p);
  for (auto& v : {&c->spans, &c->span_pool})
    for (auto& sp : *v) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  if (c->eval_host) (void)hipHostFree(c->eval_host);
  if (c->ev_host) (void)hipHostFree(c->ev_host);
  if (c->series_host) (void)hipHostFree(c->series_host);
  if (c->pstatus_host) (void)hipHostFree(c->pstatus_host);
  for (hipEvent_t e : c->eval_ev) if (e) (void)hipEventDestroy(e);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev2) (void)hipEventDestroy(c->ev2);
  if (c->ev3) (void)hipEventDestroy(c->ev3);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

static int create_impl(const gbp_problem* pr, const gbp_params* prm, const gbp_shard* sh, gbp_ctx** out) {
  if (!pr || !out || !pr->cam_id || !pr->lmk_id || pr->n_cams == 0 || pr->n_lmks == 0 || pr->n_edges == 0)
    return fail(nullptr, GBP_ERR_INVALID, "gbp_create: null or empty problem");
  // ---- device order: pure host code (gbp_layout.cpp), built and validated before anything touches the GPU ----
  Layout lay;
  {
    gbp_params dflt;
    gbp_default_params(&dflt);
    std::string lerr;
    if (int lrc = layout_build(pr, (prm ? prm : &dflt)->tile_order, sh, g_layout_options, lay, lerr)) return fail(nullptr, lrc, lerr);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, GBP_ERR_NO_DEVICE, "gbp_create: no HIP device (the product has no CPU fallback)");
  gbp_ctx* c = new gbp_ctx();
  struct Owner { gbp_ctx* p; ~Owner() { if (p) gbp_destroy(p); } } owner{c};   // released on success only
  c->lay = std::move(lay);
  c->C = pr->n_cams; c->L = pr->n_lmks; c->E = pr->n_edges;
  std::memcpy(c->K, pr->K, sizeof(c->K));
  if (prm) c->prm = *prm; else gbp_default_params(&c->prm);
  if (const char* gu = prm ? nullptr : std::getenv("GBP_GRAPH_UNROLL")) c->prm.graph_unroll = std::atoi(gu);   // measurements through callers that pass NO params (the CLIs); explicit params always win
  c->sharded_graph = c->prm.graph_unroll > 0;                // a sharded iteration is captured only on explicit request
  if (c->prm.graph_unroll == 0) c->prm.graph_unroll = 10;   // < 0: never capture, always direct launches
  c->hoist = c->prm.per_factor_mu == 0;
  c->rank = sh ? sh->rank : 0;
  c->world = sh ? sh->world : 1;
  c->lmk_begin = sh ? sh->lmk_begin : 0;
  c->lmk_end = sh ? sh->lmk_end : c->L;
  c->L_loc = c->lmk_end - c->lmk_begin;      // (the shard was validated by layout_build)

  const Layout& y = c->lay;
  c->E_loc = y.E_loc; c->n_rows = y.n_rows; c->Ep = y.Ep; c->n_tiles = y.n_tiles;
  const uint32_t C = c->C;
  c->sweep_policy = g_force_sweep_policy >= 0 ? (uint32_t)g_force_sweep_policy : sweep_policy_for(c->C, c->n_tiles);

  // ---- device allocations ----
  int rc = GBP_OK;
  auto A = [&](DevBuf& b, size_t bytes) { if (rc == GBP_OK) rc = dev_alloc(c, b, bytes); };
  const size_t Ep = c->Ep;
  A(c->row_cam, (Ep / kRow) * 4); A(c->lmk_idx, Ep * 4); A(c->fac, Ep * kFacG * 16); A(c->cmsg, Ep * kCmsgG * 16);
  A(c->mu, c->hoist ? 0 : Ep * kMuG * 16);   // literal mu/oldmu tensor: only with per_factor_mu
  A(c->lmsg, Ep * 64); A(c->d_lmk_fpos, (size_t)c->E_loc * 4); A(c->d_lmk_ix, (size_t)c->L_loc * 64);
  A(c->camb, (size_t)C * kCamRec * 4); A(c->camp, (size_t)C * kCamRec * 4); A(c->local, (size_t)C * kCamRec * 4);
  A(c->lmkb, (size_t)c->L_loc * 64); A(c->lmkp, (size_t)c->L_loc * 64);
  A(c->rowp, (Ep / kRow) * kCamRec * 4);
  A(c->d_cam_row_ptr, (size_t)(C + 1) * 4); A(c->d_lmk_ptr, (size_t)(c->L_loc + 1) * 4);
  A(c->d_row_slot, y.row_slot.size() * 4);
  A(c->cwf, (size_t)C * 4); A(c->lwf, (size_t)c->L_loc * 4); A(c->cscale, (size_t)C * 4); A(c->lscale, (size_t)c->L_loc * 4);
  A(c->cam_mu, (size_t)C * 6 * 4 * 2); A(c->lmk_mu, (size_t)c->L_loc * 3 * 4 * 2);   // metric means; k_persist alternates between the two halves
  A(c->dK, 16 * 4);
  A(c->evalp, sizeof(DeviceEval) * 16); A(c->health, 32);
  A(c->hmu_c, (size_t)C * 4 * 16); A(c->hmu_l, (size_t)c->L_loc * 2 * 16); A(c->clin, (size_t)C * 5 * 16);
  A(c->st_a, Ep * 4); A(c->st_b, Ep * 4);
  if (rc != GBP_OK) { g_create_error = c->err; return rc; }
  auto CK = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == GBP_OK) { g_create_error = std::string(what) + ": " + hipGetErrorString(e); rc = GBP_ERR_HIP; }
  };
  CK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreate");
  c->stream = c->own_stream;
  CK(hipEventCreate(&c->ev0), "hipEventCreate"); CK(hipEventCreate(&c->ev1), "hipEventCreate");
  CK(hipEventCreate(&c->ev2), "hipEventCreate"); CK(hipEventCreate(&c->ev3), "hipEventCreate");
  CK(hipMemcpy(c->d_cam_row_ptr.p, y.cam_row_ptr.data(), (size_t)(C + 1) * 4, hipMemcpyHostToDevice), "copy cam_row_ptr");
  CK(hipMemcpy(c->d_lmk_ptr.p, y.lmk_ptr.data(), (size_t)(c->L_loc + 1) * 4, hipMemcpyHostToDevice), "copy lmk_ptr");
  if (!y.row_slot.empty()) CK(hipMemcpy(c->d_row_slot.p, y.row_slot.data(), y.row_slot.size() * 4, hipMemcpyHostToDevice), "copy row_slot");
  CK(hipMemcpy(c->dK.p, c->K, 9 * 4, hipMemcpyHostToDevice), "copy K");
  if (c->E_loc) CK(hipMemcpy(c->d_lmk_fpos.p, y.lmk_fpos.data(), (size_t)c->E_loc * 4, hipMemcpyHostToDevice), "copy lmk_fpos");
  if (c->L_loc) CK(hipMemcpy(c->d_lmk_ix.p, y.lmk_ix.data(), (size_t)c->L_loc * 64, hipMemcpyHostToDevice), "copy lmk_ix");
  CK(hipMemcpy(c->row_cam.p, y.row_cam.data(), y.row_cam.size() * 4, hipMemcpyHostToDevice), "copy row_cam");
  CK(hipMemcpy(c->lmk_idx.p, y.pos_lmk_loc.data(), Ep * 4, hipMemcpyHostToDevice), "copy lmk_idx");
  if (rc == GBP_OK && !y.tile_perm.empty()) {
    // the XCD-aware execution order of the sweep: wave slot -> tile (gbp_layout.cpp); read once per wave with a scalar load
    rc = dev_alloc(c, c->tile_perm, (size_t)y.n_tiles * 4);
    if (rc == GBP_OK) CK(hipMemcpy(c->tile_perm.p, y.tile_perm.data(), (size_t)y.n_tiles * 4, hipMemcpyHostToDevice), "copy tile_perm");
    else g_create_error = c->err;
    c->use_tile_perm = rc == GBP_OK;
  }
  if (rc != GBP_OK) return rc;
  // ---- persistent iteration kernel: only where every workgroup of the graph is resident at once ----
  {
    const char* pe = prm ? nullptr : std::getenv("GBP_PERSIST");     // measurements through the CLIs (they pass no params): -1 / 0 / 1 like gbp_params.persistent
    const int mode = pe ? std::atoi(pe) : c->prm.persistent;
    const char* pc = prm ? nullptr : std::getenv("GBP_PERSIST_COOP");
    const int coop_mode = pc ? std::atoi(pc) : c->prm.persist_coop;  // 1 = cooperative launch, else (default) plain launch + probe + recovery
    const uint32_t nb = persist_blocks(c->n_tiles, c->C, c->L_loc, true);
    // measured (profiles/persist_crossover.py, round 5): with hand-offs through tagged records the persistent kernel is faster than
    // the two-kernel path on every graph it is co-resident for — the shipped sequences (14 - 61 workgroups) 1.36 - 1.55x, synthetic
    // graphs 1.16 - 1.43x up to 250 workgroups (64 000 factors: 12.2 against 16.2 us per iteration; round 4's barrier kernel broke
    // even at 125 workgroups and took 24.8 us there).  So: every graph of at most one workgroup per CU of an MI355X.
    const uint32_t auto_limit = 256;
    // (k_persist sweeps tile w on wave w and its camera role adds rows cam_row_ptr[c] .. cam_row_ptr[c + 1] where camera-major order
    // puts them: a graph with a tile permutation or with rows placed by landmark class never runs in it, whatever the size
    // thresholds of the three features say)
    if (mode >= 0 && !sh && c->hoist && !c->use_tile_perm && c->lay.row_slot.empty() && nb <= (mode > 0 ? 1u << 30 : auto_limit)) {
      const int resident = persist_max_resident_blocks();
      if (resident > 0 && nb <= (uint32_t)resident) {
        rc = dev_alloc(c, c->psync, kPersistSyncWords * sizeof(unsigned));
        if (rc == GBP_OK) {
          CK(hipHostMalloc(&c->pstatus_host, 64, hipHostMallocMapped), "hipHostMalloc");
          if (rc == GBP_OK) {
            std::memset(c->pstatus_host, 0, 64);
            CK(hipHostGetDevicePointer(&c->pstatus_dev, c->pstatus_host, 0), "hipHostGetDevicePointer");
          }
        } else {
          g_create_error = c->err;
        }
        if (rc == GBP_OK) {
          // Co-residency.  Plain launch (default): the occupancy query says the workgroups fit; a probe (the placement + three
          // barriers, no work) checks that THIS device's dispatcher keeps them resident together — under the process-wide lock
          // and behind the device's last k_persist launch, so that it does not compete with one; what another PROCESS does to
          // the GPU later is caught by the bounded barrier wait and undone by persist_recover.  Cooperative launch (persist_coop
          // = 1): the runtime refuses a grid that cannot be resident at once and the driver never runs two cooperative grids
          // side by side, whichever process they belong to — measured on MI355X: two concurrent `ba fr1xyz` then take turns
          // (44 ms each instead of 22), and every launch costs 30-60 us more (profiles/r04_persist_launch.md), which is why
          // it is the option and not the default.  A failing probe leaves the ctx on the two-kernel path and says so in
          // gbp_last_error.
          int dev = 0, coop_attr = 0;
          (void)hipGetDevice(&dev);
          (void)hipDeviceGetAttribute(&coop_attr, hipDeviceAttributeCooperativeLaunch, dev);
          std::lock_guard<std::mutex> lock(g_persist_mu);
          if (g_persist_event[dev & 15] && g_persist_last_ctx[dev & 15]) (void)hipStreamWaitEvent(c->stream, g_persist_event[dev & 15], 0);
          bool ok = false;
          if (coop_mode > 0 && coop_attr) {
            ok = persist_probe(c->n_tiles, c->C, c->L_loc, P<unsigned>(c->psync), static_cast<unsigned*>(c->pstatus_dev),
                               static_cast<volatile unsigned*>(c->pstatus_host), true, c->stream);
            c->persist_coop = ok;
            if (!ok) c->err = "k_persist: the cooperative launch was refused or its barriers timed out";
          }
          if (coop_mode > 0 && !coop_attr) c->err = "k_persist: this device does not offer cooperative launches";
          if (coop_mode <= 0) {
            ok = persist_probe(c->n_tiles, c->C, c->L_loc, P<unsigned>(c->psync), static_cast<unsigned*>(c->pstatus_dev),
                               static_cast<volatile unsigned*>(c->pstatus_host), false, c->stream);
            if (!ok) c->err = "k_persist: the workgroups of this graph are not co-resident under the spread placement on this device (probe timed out)";
          }
          if (!ok) c->err += "; iterations run on the two-kernel path";
          c->persist_ok = c->persist_eligible = ok;
          if (ok) CK(hipMemset(c->psync.p, 0, kPersistSyncWords * sizeof(unsigned)), "hipMemset");   // counter back to 0 after the probe
          else (void)hipGetLastError();
        }
        if (rc == GBP_OK && c->persist_ok) {
          // snapshot arena: one slot for every array a k_persist launch mutates
          // (+ the priors and the weaken flags: a launch of gbp_ba_loop weakens priors itself)
          DevBuf* segs[] = {&c->lmsg, &c->cmsg, &c->fac, &c->rowp, &c->camb, &c->lmkb, &c->hmu_c, &c->hmu_l, &c->clin, &c->local,
                            &c->camp, &c->lmkp, &c->cwf, &c->lwf};
          size_t total = 0;
          for (DevBuf* b : segs) total += (b->bytes + 15) / 16 * 16;
          rc = dev_alloc(c, c->psnap, total);
          if (rc == GBP_OK) {
            size_t off = 0;
            int i = 0;
            for (DevBuf* b : segs) {
              void* slot = static_cast<char*>(c->psnap.p) + off;
              c->snap_save.src[i] = b->p; c->snap_save.dst[i] = slot; c->snap_save.n4[i] = (b->bytes + 15) / 16;      // (hipMalloc granules are larger)
              c->snap_restore.src[i] = slot; c->snap_restore.dst[i] = b->p; c->snap_restore.n4[i] = (b->bytes + 15) / 16;
              off += (b->bytes + 15) / 16 * 16;
              ++i;
            }
            c->snap_save.n = c->snap_restore.n = i;
          } else {
            g_create_error = c->err;
          }
        }
        if (rc == GBP_OK && c->persist_ok) {
          // tagged shadows of the arrays that cross waves inside a launch (k_persist_flow): two halves each
          const size_t Ep = (size_t)c->n_tiles * 64, C_ = c->C, L_ = c->L_loc;
          const size_t n4[10] = {2 * Ep * 4, 2 * (Ep / 16) * kFlowRow4, 2 * C_ * kFlowCam4, 2 * C_ * 2, 2 * C_ * kFlowClin4, 2 * L_ * kFlowLmk4, 2 * L_,
                                 2 * C_ * 4, 2 * L_, (size_t)kSeriesMax};      // (the last: [kSeriesMax][2] 64-bit health words = kSeriesMax float4)
          size_t total = 0;
          for (size_t n : n4) total += n;
          rc = dev_alloc(c, c->pflow, total * 16);
          if (rc == GBP_OK) {
            float4* q = static_cast<float4*>(c->pflow.p);
            float4** dst[9] = {&c->flow.lmsg, &c->flow.rowp, &c->flow.camb, &c->flow.cmu, &c->flow.clin, &c->flow.lmkb, &c->flow.lmu, &c->flow.emc, &c->flow.eml};
            for (int i = 0; i < 9; ++i) { *dst[i] = q; q += n4[i]; }
            c->flow.health_iter = reinterpret_cast<unsigned long long*>(q);
            // the host-mapped slots of gbp_iterate_eval_each / gbp_ba_loop: here, not inside the first timed burst (pinning 5 MB takes
            // ~0.5 ms, a twentieth of a default `ba fr1xyz` run)
            if (!c->series_host) {
              CK(hipHostMalloc(&c->series_host, sizeof(DeviceEval) * (size_t)(c->n_tiles + 1) * kSeriesMax, hipHostMallocMapped), "hipHostMalloc");
              if (rc == GBP_OK) CK(hipHostGetDevicePointer(&c->series_dev, c->series_host, 0), "hipHostGetDevicePointer");
            }
#ifdef GBP_BUILD_TEST_HOOKS
            const char* pf = prm ? nullptr : std::getenv("GBP_PERSIST_FLOW");      // (the barrier kernel exists in the test-hooks build only)
            if (pf && std::atoi(pf) == 0) c->persist_flow = false;
#endif
          } else {
            g_create_error = c->err;
          }
        }
      }
    }
    if (rc != GBP_OK) return rc;
  }
  // (the zero-fills of dev_alloc ran on the NULL stream, which the ctx's non-blocking stream does not order against)
  CK(hipDeviceSynchronize(), "hipDeviceSynchronize");
  if (rc != GBP_OK) return rc;
  owner.p = nullptr;
  *out = c;
  return GBP_OK;
}

// ---- k_persist launches in flight: validation and recovery (defined with launch_persist_burst below) ----
static int persist_check(gbp_ctx* c, unsigned upto);
// Every entry point that enqueues other device work, or changes what the launches in flight depend on, first makes sure
// they completed without a barrier time-out (and repairs the state if one did): free when nothing is in flight.
static bool stream_is_capturing(gbp_ctx* c);
static int settle(gbp_ctx* c) {
  if (c->persist_log.empty()) return GBP_OK;
  // unvalidated k_persist launches and a caller who has begun capturing the stream: synchronising would invalidate their capture
  if (stream_is_capturing(c))
    return fail(c, GBP_ERR_STATE, "bursts of the persistent kernel are still in flight on this stream: call gbp_sync before beginning a stream capture");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return persist_check(c, 0);
}

int gbp_set_stream(gbp_ctx* c, void* s) {
  if (!c) return GBP_ERR_INVALID;
  if (int rc = settle(c)) return rc;
  drop_graph(c);
  c->stream = s ? static_cast<hipStream_t>(s) : c->own_stream;
  return GBP_OK;
}

int gbp_set_exchange_buffers(gbp_ctx* c, void* send_dev, void* recv_dev) {
  if (!c) return GBP_ERR_INVALID;
  c->send_dev = send_dev; c->recv_dev = recv_dev;
  return GBP_OK;
}

int gbp_sync(gbp_ctx* c) {
  if (!c) return GBP_ERR_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return persist_check(c, 0);
}

// WRITE_PROG (ba.cpp:868-886).  Also zeroes every tensor the reference leaves uninitialised
// (messages, factor potentials, beliefs: ba.cpp:668-687,759-775).
static int upload_impl(gbp_ctx* c, const gbp_state_in* in) {
  if (!c || !in) return GBP_ERR_INVALID;
  if (!in->cam_priors_eta || !in->cam_priors_lambda || !in->lmk_priors_eta || !in->lmk_priors_lambda ||
      !in->measurements || !in->meas_variances || !in->active_flag)
    return fail(c, GBP_ERR_INVALID, "gbp_upload: priors, measurements, meas_variances and active_flag are required");
  if (in->mu && in->oldmu && std::memcmp(in->mu, in->oldmu, (size_t)c->E * 9 * 4) != 0)
    return fail(c, GBP_ERR_INVALID, "gbp_upload: mu != oldmu is not supported (the reference uploads zeros for both, ba.cpp:582-583)");
  if (c->hoist) {
    const float* om = in->oldmu ? in->oldmu : in->mu;
    if (om)
      for (size_t i = 0; i < (size_t)c->E * 9; ++i)
        if (om[i] != 0.f)
          return fail(c, GBP_ERR_INVALID, "gbp_upload: non-zero oldmu needs gbp_params.per_factor_mu = 1 (the reference uploads zeros, ba.cpp:582-583)");
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->pstatus_host) {
    // a fresh start for the persistent kernel too: whatever its launches in flight did is overwritten below; a ctx that left
    // the persistent path after a recovered time-out gets it back
    c->persist_log.clear();
    HIPCHK(c, hipMemset(c->psync.p, 0, kPersistSyncWords * sizeof(unsigned)));
    *static_cast<volatile unsigned*>(c->pstatus_host) = 0u;
    c->persist_epoch_base = 0;
    c->persist_ok = c->persist_eligible;
  }
  const size_t Ep = c->Ep;
  std::vector<float> rec0(Ep * 16, 0.f), fac(Ep * kFacG * 4, 0.f), mu(c->hoist ? 0 : Ep * kMuG * 4, 0.f);
  c->active_host.assign(Ep, 0);
  for (size_t p = 0; p < Ep; ++p) {
    const uint32_t e = c->lay.pos_edge[p];
    HostState h{0.f, 0, kFlagPad, 0.f};
    if (e != ~0u) {
      h.flags = (in->active_flag[e] == 1) ? kFlagActive : 0u;
      c->active_host[p] = in->active_flag[e] == 1;
      h.damping = in->damping ? in->damping[e] : 0.f;
      h.count = in->damping_count ? in->damping_count[e] : 0;
      h.var = in->meas_variances[e];
      fac[tile_off((uint32_t)p, kFacG, 54)] = in->measurements[2 * (size_t)e];
      fac[tile_off((uint32_t)p, kFacG, 55)] = in->measurements[2 * (size_t)e + 1];
      const float* om = in->oldmu ? in->oldmu : in->mu;
      if (om && !c->hoist) for (int i = 0; i < 9; ++i) mu[tile_off((uint32_t)p, kMuG, i)] = om[(size_t)e * 9 + i];
    }
    put_state(rec0, p, h);
  }
  HIPCHK(c, hipMemcpy(c->lmsg.p, rec0.data(), rec0.size() * 4, hipMemcpyHostToDevice));   // zero messages + state
  HIPCHK(c, hipMemcpy(c->fac.p, fac.data(), fac.size() * 4, hipMemcpyHostToDevice));
  if (!c->hoist) HIPCHK(c, hipMemcpy(c->mu.p, mu.data(), mu.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemset(c->cmsg.p, 0, c->cmsg.bytes));
  HIPCHK(c, hipMemset(c->rowp.p, 0, c->rowp.bytes));
  HIPCHK(c, hipMemset(c->local.p, 0, c->local.bytes));
  HIPCHK(c, hipMemset(c->camb.p, 0, c->camb.bytes));
  HIPCHK(c, hipMemset(c->lmkb.p, 0, c->lmkb.bytes));
  HIPCHK(c, hipMemset(c->hmu_c.p, 0, c->hmu_c.bytes));
  HIPCHK(c, hipMemset(c->hmu_l.p, 0, c->hmu_l.bytes));
  HIPCHK(c, hipMemset(c->clin.p, 0, c->clin.bytes));
  std::vector<float> rec;
  pack_cam(in->cam_priors_eta, in->cam_priors_lambda, c->C, rec);
  HIPCHK(c, hipMemcpy(c->camp.p, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
  pack_lmk(in->lmk_priors_eta, in->lmk_priors_lambda, c->lmk_begin, c->L_loc, rec);
  if (c->L_loc) HIPCHK(c, hipMemcpy(c->lmkp.p, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> zf(std::max(c->C, c->L), 0.f);
  std::vector<uint32_t> zu(std::max(c->C, c->L), 0u);
  HIPCHK(c, hipMemcpy(c->cscale.p, in->cam_scaling ? in->cam_scaling : zf.data(), (size_t)c->C * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->cwf.p, in->cam_weaken_flag ? in->cam_weaken_flag : zu.data(), (size_t)c->C * 4, hipMemcpyHostToDevice));
  if (c->L_loc) {
    HIPCHK(c, hipMemcpy(c->lscale.p, in->lmk_scaling ? in->lmk_scaling + c->lmk_begin : zf.data(), (size_t)c->L_loc * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->lwf.p, in->lmk_weaken_flag ? in->lmk_weaken_flag + c->lmk_begin : zu.data(), (size_t)c->L_loc * 4, hipMemcpyHostToDevice));
  }
  if (exch(c) && c->recv_dev) HIPCHK(c, hipMemsetAsync(c->recv_dev, 0, (size_t)c->world * c->C * kCamRec * 4, c->stream));
  c->uploaded = true;
  c->beliefs_valid = false;
  return GBP_OK;
}

int gbp_refresh_begin(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  float4* dst = exch(c) ? static_cast<float4*>(c->send_dev) : P<float4>(c->local);
  if (!dst) return fail(c, GBP_ERR_STATE, "exchange buffers not set");
  enqueue_cam_partials(c, reinterpret_cast<float*>(dst));
  HIPCHK(c, hipGetLastError());
  return GBP_OK;
}

int gbp_refresh_end(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  const int rc = refresh_beliefs_from_partials(c, false);
  if (rc == GBP_OK) c->beliefs_valid = true;
  return rc;
}

int gbp_linearise_factors(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  launch_linearise(sweep_args(c), c->n_tiles, c->stream);
  HIPCHK(c, hipGetLastError());
  return GBP_OK;
}

// ---- sharded ctx with a library-owned communicator (gbp_comm_init*) ----------------------------------------------------
#define COMMCHK(ctx, expr)                                                       \
  do {                                                                           \
    std::string e_;                                                              \
    if ((expr) != 0) return fail(ctx, GBP_ERR_COMM, "exchange: " + e_);          \
  } while (0)

// plain exchange on the ctx's stream (LINEARISE, refreshes): partials in send_dev -> recv_dev of every rank
static int exchange_now(gbp_ctx* c) {
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

static int iterate_sharded(gbp_ctx* c, int n) {
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

// LINEARISE_PROG (ba.cpp:890-893): prog_ub, then RelineariseFactorVertex on every factor.
static int linearise_impl(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_linearise: upload first");
  if (c->world > 1 && !c->comm)
    return fail(c, GBP_ERR_STATE, "sharded ctx without a communicator: gbp_comm_init first, or use refresh_begin / exchange / refresh_end / linearise_factors");
  int rc = gbp_refresh_begin(c);
  if (rc == GBP_OK && c->comm) rc = exchange_now(c);
  if (rc == GBP_OK) rc = gbp_refresh_end(c);
  if (rc == GBP_OK) rc = gbp_linearise_factors(c);
  return rc;
}

static int iterate_begin_impl(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  float4* dst = exch(c) ? static_cast<float4*>(c->send_dev) : P<float4>(c->local);
  if (!dst) return fail(c, GBP_ERR_STATE, "exchange buffers not set");
  if (c->profile_stages) {  // bracket the sweep launch; the pair is read (and timed_iters counted) by gbp_timing
    if (c->pending_sweep_ev.size() >= 256) drain_sweep_events(c);   // bounded: long profiled runs never pile up events
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(c, hipEventCreate(&e0));
    if (hipError_t e_ = hipEventCreate(&e1); e_ != hipSuccess) {
      (void)hipEventDestroy(e0);
      return fail(c, GBP_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e_));
    }
    c->pending_sweep_ev.emplace_back(e0, e1);
    HIPCHK(c, hipEventRecord(e0, c->stream));
    launch_sweep(sweep_args(c), c->n_tiles, c->hoist, c->stream);
    HIPCHK(c, hipEventRecord(e1, c->stream));
  } else {
    launch_sweep(sweep_args(c), c->n_tiles, c->hoist, c->stream);
  }
  enqueue_cam_partials(c, reinterpret_cast<float*>(dst));
  HIPCHK(c, hipGetLastError());
  return GBP_OK;
}

// The landmark half of the belief update needs nothing from other ranks: a caller may run it while the
// all-gather of the camera partials is in flight (between gbp_iterate_begin and gbp_iterate_end).
int gbp_iterate_local(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  BeliefArgs b = belief_args(c);
  b.roll = 1;
  launch_beliefs(b, false, true, c->stream);
  HIPCHK(c, hipGetLastError());
  c->lmk_half_done = true;
  return GBP_OK;
}

int gbp_iterate_end(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  const int rc = refresh_beliefs_from_partials(c, true, !c->lmk_half_done);
  c->lmk_half_done = false;
  if (rc == GBP_OK) c->beliefs_valid = true;
  return rc;
}

// Capture `graph_unroll` single-GPU iterations once (nothing is executed by a capture).  Any failure leaves the stream
// out of capture mode, drops the partial graph and falls back to direct launches for the life of the ctx (results are
// identical either way).
static bool ensure_graph(gbp_ctx* c, const SweepArgs& a, bool ev = false) {
  hipGraph_t& g = ev ? c->graph_ev : c->graph;
  hipGraphExec_t& x = ev ? c->graph_exec_ev : c->graph_exec;
  if (x) return true;
  if (c->graph_failed || c->prm.graph_unroll <= 0 || c->stream != c->own_stream) return false;
  hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
  if (e == hipSuccess) {
    for (int i = 0; i < c->prm.graph_unroll; ++i) enqueue_iteration(c, a, ev);
    e = hipStreamEndCapture(c->stream, &g);          // also ends a capture that was invalidated on the way
    if (e == hipSuccess) e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    drop_graph(c);
    c->graph_failed = true;
    return false;
  }
  c->graph_iters = c->prm.graph_unroll;
  return true;
}

// One-off costs of the multi-iteration path, paid on request instead of inside the first gbp_iterate(n >= graph_unroll):
// graph capture + instantiation + upload of the executable graph.  Executes no iteration.
int gbp_prepare(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_prepare: upload first");
  if (int rc = settle(c)) return rc;
  if (c->comm || c->world > 1) return GBP_OK;               // sharded iterations run from direct launches by default
  if (c->persist_ok) return GBP_OK;                         // multi-iteration bursts run inside k_persist: nothing to capture
  if (ensure_graph(c, sweep_args(c))) (void)hipGraphUpload(c->graph_exec, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GBP_OK;
}

// GBP_PROG x n on the two-kernel path: replay of a captured hipGraph of `graph_unroll` iterations, remainder launched directly.
// (ev: the iterations carry the metric — a.ev set, see eval_each_ride; their launches depend on the iteration only through a
// counter in device memory, so they replay from a graph of their own)
static int iterate_plain(gbp_ctx* c, const SweepArgs& a, int n, bool ev = false) {
  int left = n;
  bool use_graph = (c->stream == c->own_stream) && c->prm.graph_unroll > 0 && n >= c->prm.graph_unroll && !c->graph_failed;
  if (use_graph && !(ev ? c->graph_exec_ev : c->graph_exec)) use_graph = ensure_graph(c, a, ev);
  if (use_graph) {
    while (left >= c->graph_iters) {
      HIPCHK(c, hipGraphLaunch(ev ? c->graph_exec_ev : c->graph_exec, c->stream));
      left -= c->graph_iters;
    }
  }
  for (; left > 0; --left) enqueue_iteration(c, a, ev);
  HIPCHK(c, hipGetLastError());
  return GBP_OK;
}

static int eval_enqueue(gbp_ctx* c, int area);

// Passes i0 .. i0 + n - 1 of the reference's loop WITHOUT the metric on the two-kernel path of a single-GPU ctx, the weakening in
// front of pass i0 (if any) already done by the caller: a weakening in front of a later pass rides in the belief update of the
// iteration before it (enqueue_iteration: weaken_after), the runs between them replay from the hipGraph.
static int iterate_weaken_plain(gbp_ctx* c, const SweepArgs& a, int n, unsigned i0, unsigned steps2) {
  const auto weak = [&](unsigned i) { return ((i + 1u) % 2u == 0u) && i < steps2; };
  int k = 0;
  while (k < n) {
    int run = 0;      // iterations whose successor (inside this call) is not weakened
    while (k + run < n && !(k + run + 1 < n && weak(i0 + (unsigned)(k + run) + 1u))) ++run;
    if (run)
      if (int rc = iterate_plain(c, a, run)) return rc;
    k += run;
    if (k < n) {      // ... and the one whose belief update takes the weakened priors
      enqueue_iteration(c, a, false, true);
      HIPCHK(c, hipGetLastError());
      ++k;
    }
  }
  return GBP_OK;
}

// A k_persist launch gave up at a barrier (*pstatus_host = its number): undo it and everything queued behind it, replay
// on the two-kernel path.  The snapshot kernel of every later launch saw the abort word and left the arena alone, so the
// arena holds the state the first failed launch started from.
static int persist_recover(gbp_ctx* c) {
  HIPCHK(c, hipStreamSynchronize(c->stream));        // the failed launch and the no-op launches behind it have ended
  const unsigned first = *static_cast<volatile unsigned*>(c->pstatus_host);
  std::vector<gbp_ctx::Burst> redo;
  for (const gbp_ctx::Burst& b : c->persist_log)
    if (b.seq >= first) redo.push_back(b);
  c->persist_log.clear();
  launch_copy_segments(c->snap_restore, nullptr, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemsetAsync(c->psync.p, 0, kPersistSyncWords * sizeof(unsigned), c->stream));
  HIPCHK(c, hipMemsetAsync(c->health.p, 0, 32, c->stream));       // both areas are zero between evaluations
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *static_cast<volatile unsigned*>(c->pstatus_host) = 0u;
  c->persist_epoch_base = 0;
  c->persist_ok = false;                              // until the next gbp_upload
  c->persist_recoveries += 1;
  const SweepArgs a = sweep_args(c);
  long iters = 0;
  for (const gbp_ctx::Burst& b : redo) {
    if (b.mode == 2) continue;                        // gbp_iterate_eval_each is blocking: it replays its own burst
    if (b.w_steps2) { if (int rc = iterate_weaken_plain(c, a, b.n, b.w_first, b.w_steps2)) return rc; }
    else if (int rc = iterate_plain(c, a, b.n)) return rc;
    iters += b.n;
    if (b.mode == 1)
      if (int rc = eval_enqueue(c, b.area)) return rc;
  }
  // the replay has completed when this returns: the callers (gbp_sync, gbp_read*, gbp_new_keyframe, gbp_set_stream, the debug
  // accessors) go on to blocking copies on the NULL stream, which a non-blocking stream does not order against
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->warn = "warning: a device-wide barrier of the persistent kernel timed out in launch " + std::to_string(first) +
            " of this ctx (its workgroups were not co-resident: is another process using the GPU?); the state was restored and " +
            std::to_string(iters) + " iterations were replayed on the two-kernel path (identical results); the ctx stays on that path until the next gbp_upload";
  c->err = c->warn;
  return GBP_OK;
}

// The stream has been synchronised, or an event recorded behind launch `upto` has completed (0 = everything queued has).
static int persist_check(gbp_ctx* c, unsigned upto) {
  if (!c->pstatus_host || c->persist_log.empty()) return GBP_OK;
  if (*static_cast<volatile unsigned*>(c->pstatus_host) != 0u) return persist_recover(c);
  if (upto == 0) c->persist_log.clear();
  else
    while (!c->persist_log.empty() && c->persist_log.front().seq <= upto) c->persist_log.erase(c->persist_log.begin());
  return GBP_OK;
}

static bool stream_is_capturing(gbp_ctx* c) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(c->stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}

// May the next burst of this ctx run inside k_persist?  Makes room in the log of unvalidated launches first (which may
// find a time-out, recover, and take the ctx off the persistent path).  Not while the stream is being captured: the
// barrier targets are launch arguments computed by the host, a replayed graph would wait for arrivals long past.
static int persist_ready(gbp_ctx* c, bool* yes) {
  *yes = false;
  if (!c->persist_ok || c->comm || c->world != 1 || c->profile_stages) return GBP_OK;
  if (stream_is_capturing(c)) return GBP_OK;
  if (c->persist_log.size() >= kPersistLogMax)
    if (int rc = settle(c)) return rc;
  *yes = c->persist_ok;
  return GBP_OK;
}

constexpr int kNotLaunched = 1;    // launch_persist_burst: nothing ran, the ctx has left the persistent path

// n iterations inside ONE k_persist launch (+ the metric phases when `ev` is given).  The barrier counter keeps counting
// across the launches of a ctx (no memset per launch): the host tracks how many arrivals it has seen.
static int launch_persist_burst(gbp_ctx* c, const SweepArgs& a, int n, const PersistEval* ev, int mode, int area,
                                uint32_t w_first = 0, uint32_t w_steps2 = 0) {
  PersistArgs A{};
  A.s = a;
  A.b = belief_args(c);
  A.b.roll = 1;
  if (w_steps2) {      // WEAKEN_PRIORS inside the launch (gbp_ba_loop)
    A.w_first = w_first; A.w_steps2 = w_steps2;
    A.b.cam_prior_rw = P<float>(c->camp); A.b.cam_scale = P<float>(c->cscale); A.b.cam_wflag = P<uint32_t>(c->cwf);
    A.b.lmk_prior_rw = P<float4>(c->lmkp); A.b.lmk_scale = P<float>(c->lscale); A.b.lmk_wflag = P<uint32_t>(c->lwf);
  }
  A.n_tiles = c->n_tiles;
  A.n_iters = n;
  A.sync = P<unsigned>(c->psync);
  A.status = static_cast<unsigned*>(c->pstatus_dev);
  A.epoch_base = c->persist_epoch_base;
  A.seq = c->persist_seq + 1;
  if (ev) A.ev = *ev;
  const bool flow = c->persist_flow && c->flow.lmsg != nullptr;      // hand-offs through tagged records (k_persist_flow)
  if (flow) {
    A.f = c->flow;
    A.f.tag0 = (A.seq & 0x7ffffu) << 13;      // + iteration (<= kPersistChunk) + 1: never the tag of a record an earlier launch left behind
  }
  {
    // Two k_persist launches must never compete for CUs (each spins at its barriers until ALL its workgroups are resident).
    // Across processes that is the cooperative launch's guarantee; inside a process a launch from another ctx or stream than
    // the previous one waits for the event recorded behind that one (launches on one stream are ordered anyway).
    std::lock_guard<std::mutex> lock(g_persist_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipEvent_t& e = g_persist_event[dev & 15];
    if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (g_persist_last_ctx[dev & 15] && (g_persist_last_ctx[dev & 15] != c || g_persist_last_stream[dev & 15] != c->stream))
      HIPCHK(c, hipStreamWaitEvent(c->stream, e, 0));
    launch_copy_segments(c->snap_save, P<unsigned>(c->psync) + 32, c->stream);       // skipped on the device once the abort word is set (5-7 us per launch)
    HIPCHK(c, hipGetLastError());
    const hipError_t le = launch_persist(A, c->persist_coop, c->stream);
    if (le != hipSuccess) {
      (void)hipGetLastError();
      if (!c->persist_coop) return fail(c, GBP_ERR_HIP, std::string("k_persist launch: ") + hipGetErrorString(le));
      // the runtime refused the cooperative grid: nothing ran, nothing is lost — this ctx continues on the two-kernel path
      c->persist_ok = c->persist_eligible = false;
      c->warn = std::string("warning: the cooperative launch of the persistent kernel was refused (") + hipGetErrorString(le) +
                "); iterations run on the two-kernel path";
      c->err = c->warn;
      return kNotLaunched;
    }
    HIPCHK(c, hipEventRecord(e, c->stream));
    g_persist_last_ctx[dev & 15] = c;
    g_persist_last_stream[dev & 15] = c->stream;
  }
  const unsigned nb = persist_blocks(c->n_tiles, c->C, c->L_loc, ev != nullptr && ev->each != 0);      // the grid launch_persist used
  // arrivals of this launch (n <= kPersistChunk; the counter wraps, grid_sync compares wrap-safe): two hand-offs per iteration — with
  // tagged records none, and ONE barrier at the end of a launch that carries the metric
  c->persist_epoch_base += flow ? (ev ? nb : 0u) : nb * (unsigned)(2 * n - 1 + (ev ? 1 : 0));
  c->persist_seq += 1;
  c->persist_log.push_back(gbp_ctx::Burst{c->persist_seq, n, mode, area, w_first, w_steps2});
  c->persist_launches += 1;
  return GBP_OK;
}

// GBP_PROG x n (ba.cpp:895-905) on one GPU: inside the persistent kernel (small graphs), else hipGraph replay / direct launches.
static int iterate_impl(gbp_ctx* c, int n) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_iterate: upload first");
  if (n <= 0) return GBP_OK;
  if (c->comm) {
    if (int rc = settle(c)) return rc;
    return iterate_sharded(c, n);
  }
  if (c->world > 1)
    return fail(c, GBP_ERR_STATE, "sharded ctx without a communicator: gbp_comm_init first, or use gbp_iterate_begin / exchange / gbp_iterate_end");
  bool persist = false;
  if (n >= 2)                          // a single iteration is as fast from two launches (measured)
    if (int rc = persist_ready(c, &persist)) return rc;
  if (!persist)
    if (int rc = settle(c)) return rc;
  const SweepArgs a = sweep_args(c);
  if (c->stream != c->own_stream && stream_is_capturing(c)) {
    // the caller is capturing its own stream (gbp_set_stream) into a graph: plain kernel launches only — no persistent kernel
    // (host-computed barrier targets), no timing events that would become graph nodes, no capture of our own inside theirs
    for (int i = 0; i < n; ++i) enqueue_iteration(c, a);
    HIPCHK(c, hipGetLastError());
    c->beliefs_valid = true;
    return GBP_OK;
  }
  gbp_ctx::Span sp{};
  if (int rc = span_begin(c, sp)) return rc;
  if (c->profile_stages) {
    // Per-stage timing: all n iterations are queued back to back with a hipEvent before / between / after the
    // two kernels, and read after ONE synchronisation, so a bracket holds the kernel (plus the ~1 us
    // dependent-launch gap), not the idle-queue start-up latency a per-iteration sync would add.
    struct Events {   // freed on every exit path
      std::vector<hipEvent_t> v;
      ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
    } evs;
    evs.v.assign(2 * (size_t)n + 1, nullptr);
    std::vector<hipEvent_t>& ev = evs.v;
    for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
    HIPCHK(c, hipEventRecord(ev[0], c->stream));
    for (int i = 0; i < n; ++i) {
      launch_sweep(a, c->n_tiles, c->hoist, c->stream);
      HIPCHK(c, hipEventRecord(ev[2 * i + 1], c->stream));
      BeliefArgs b = belief_args(c);
      b.roll = 1;
      launch_beliefs(b, true, true, c->stream);
      HIPCHK(c, hipEventRecord(ev[2 * i + 2], c->stream));
    }
    HIPCHK(c, hipEventSynchronize(ev[2 * (size_t)n]));
    for (int i = 0; i < n; ++i) {
      float a_ms = 0, b_ms = 0;
      HIPCHK(c, hipEventElapsedTime(&a_ms, ev[2 * i], ev[2 * i + 1]));
      HIPCHK(c, hipEventElapsedTime(&b_ms, ev[2 * i + 1], ev[2 * i + 2]));
      c->sweep_ms += a_ms; c->belief_ms += b_ms;
    }
  } else if (persist) {
    // small graph: the whole burst in one launch (k_persist); very long bursts in pieces, a launch cannot be pre-empted
    for (int left = n; left > 0;) {
      const int m = std::min(left, kPersistChunk);
      int rc = launch_persist_burst(c, a, m, nullptr, 0, 0);
      if (rc == GBP_OK) {
        left -= m;
        if (left > 0) {
          rc = persist_ready(c, &persist);
          if (rc == GBP_OK && !persist) rc = kNotLaunched;
        }
      }
      if (rc == kNotLaunched) {        // the ctx left the persistent path: the rest on the two-kernel path
        rc = settle(c);
        if (rc == GBP_OK) rc = iterate_plain(c, a, left);
        left = 0;
      }
      if (rc != GBP_OK) { c->span_pool.push_back(sp); return rc; }
    }
  } else {
    if (int rc = iterate_plain(c, a, n)) { c->span_pool.push_back(sp); return rc; }
  }
  HIPCHK(c, hipGetLastError());
  if (int rc = span_end(c, sp)) return rc;
  c->timed_iters += (uint64_t)n;
  c->beliefs_valid = true;
  return GBP_OK;
}

// WEAKEN_PRIORS (ba.cpp:863-865): WeakenPriorVertex on every variable, then prog_ub.
static int weaken_priors_impl(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_weaken_priors: upload first");
  if (int rc = settle(c)) return rc;
  return refresh_beliefs_from_partials(c, false, true, /*weaken=*/true);      // ONE launch: the prior owners scale on their way into the sums
}
int gbp_weaken_priors(gbp_ctx* c) { return weaken_priors_impl(c); }

// READ_PROG (ba.cpp:908-916)
static int read_impl(gbp_ctx* c, gbp_state_out* o) {
  if (!c || !o) return GBP_ERR_INVALID;
  if (int rc = gbp_sync(c)) return rc;
  if (o->cam_beliefs_eta || o->cam_beliefs_lambda) {
    std::vector<float> rec((size_t)c->C * kCamRec);
    HIPCHK(c, hipMemcpy(rec.data(), c->camb.p, rec.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < c->C; ++k) {
      if (o->cam_beliefs_eta) std::memcpy(o->cam_beliefs_eta + (size_t)k * 6, &rec[(size_t)k * kCamRec], 6 * 4);
      if (o->cam_beliefs_lambda) std::memcpy(o->cam_beliefs_lambda + (size_t)k * 36, &rec[(size_t)k * kCamRec + 8], 36 * 4);
    }
  }
  if ((o->lmk_beliefs_eta || o->lmk_beliefs_lambda) && c->L_loc) {
    std::vector<float> rec((size_t)c->L_loc * 16);
    HIPCHK(c, hipMemcpy(rec.data(), c->lmkb.p, rec.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t l = 0; l < c->L_loc; ++l) {
      if (o->lmk_beliefs_eta) std::memcpy(o->lmk_beliefs_eta + (size_t)(c->lmk_begin + l) * 3, &rec[(size_t)l * 16], 3 * 4);
      if (o->lmk_beliefs_lambda) std::memcpy(o->lmk_beliefs_lambda + (size_t)(c->lmk_begin + l) * 9, &rec[(size_t)l * 16 + 4], 9 * 4);
    }
  }
  if (o->damping || o->damping_count || o->robust_flag) {
    // per-factor scalars ride in the message records: a small kernel extracts them into two compact arrays
    launch_state_get(P<float4>(c->lmsg), P<float>(c->st_a), P<int>(c->st_b), c->Ep, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<float> damp(c->Ep);
    std::vector<int32_t> packed(c->Ep);
    HIPCHK(c, hipMemcpy(damp.data(), c->st_a.p, (size_t)c->Ep * 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(packed.data(), c->st_b.p, (size_t)c->Ep * 4, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < c->Ep; ++p) {
      const uint32_t e = c->lay.pos_edge[p];
      if (e == ~0u) continue;
      if (o->damping) o->damping[e] = damp[p];
      if (o->damping_count) o->damping_count[e] = packed[p] >> 3;
      if (o->robust_flag) o->robust_flag[e] = ((uint32_t)packed[p] & kFlagRobust) ? 1u : 0u;
    }
  }
  return GBP_OK;
}

// READ_PRIORS (slam.cpp:913-917)
static int read_priors_impl(gbp_ctx* c, gbp_priors_out* o) {
  if (!c || !o) return GBP_ERR_INVALID;
  if (int rc = gbp_sync(c)) return rc;
  std::vector<float> rec((size_t)c->C * kCamRec);
  HIPCHK(c, hipMemcpy(rec.data(), c->camp.p, rec.size() * 4, hipMemcpyDeviceToHost));
  for (uint32_t k = 0; k < c->C; ++k) {
    if (o->cam_priors_eta) std::memcpy(o->cam_priors_eta + (size_t)k * 6, &rec[(size_t)k * kCamRec], 6 * 4);
    if (o->cam_priors_lambda) std::memcpy(o->cam_priors_lambda + (size_t)k * 36, &rec[(size_t)k * kCamRec + 8], 36 * 4);
  }
  if (c->L_loc) {
    rec.resize((size_t)c->L_loc * 16);
    HIPCHK(c, hipMemcpy(rec.data(), c->lmkp.p, rec.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t l = 0; l < c->L_loc; ++l) {
      if (o->lmk_priors_eta) std::memcpy(o->lmk_priors_eta + (size_t)(c->lmk_begin + l) * 3, &rec[(size_t)l * 16], 3 * 4);
      if (o->lmk_priors_lambda) std::memcpy(o->lmk_priors_lambda + (size_t)(c->lmk_begin + l) * 9, &rec[(size_t)l * 16 + 4], 9 * 4);
    }
  }
  return GBP_OK;
}

// NEW_KEYFRAME (slam.cpp:919-928): re-upload damping_count, priors, flags; then prog_ub.
static int new_keyframe_impl(gbp_ctx* c, const gbp_kf_update* u) {
  if (!c || !u || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_new_keyframe: upload first");
  if (int rc = gbp_sync(c)) return rc;
  if (u->damping_count || u->active_flag) {
    // edit the per-factor scalars in place on the device: 8 bytes per factor go over PCIe, not the 64-byte records
    std::vector<int32_t> cnt(c->Ep, 0);
    std::vector<uint32_t> ctl(c->Ep, 0u);
    const int thr = c->prm.min_linear_iters - c->prm.num_undamped_iters;
    for (size_t p = 0; p < c->Ep; ++p) {
      const uint32_t e = c->lay.pos_edge[p];
      if (e == ~0u) continue;
      if (u->damping_count) { cnt[p] = u->damping_count[e]; ctl[p] |= 1u; }
      if (u->active_flag) {
        const bool on = u->active_flag[e] == 1;
        // Hoisted means (k_sweep<HOIST>): a factor's FIRST active sweep measures dmu against the variable's previous
        // mean where the reference measures it against the factor's own zero-initialised oldmu (ba.cpp:582-583).  The
        // two agree as long as that sweep cannot relinearise, i.e. count + 1 <= min_linear_iters - num_undamped_iters
        // (gbp_codelets.cpp:280) — true for the reference's re-arm value -15 (slam.cpp:1040) and its defaults.
        if (c->hoist && on && !c->active_host[p] && u->damping_count && u->damping_count[e] + 1 > thr)
          return fail(c, GBP_ERR_INVALID, "gbp_new_keyframe: a factor activated with damping_count + 1 > min_linear_iters - "
                                          "num_undamped_iters could relinearise on its first sweep; that needs gbp_params.per_factor_mu = 1");
        ctl[p] |= 2u | (on ? 4u : 0u);
      }
    }
    HIPCHK(c, hipMemcpy(c->st_b.p, cnt.data(), (size_t)c->Ep * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->st_a.p, ctl.data(), (size_t)c->Ep * 4, hipMemcpyHostToDevice));
    launch_state_set(P<float4>(c->lmsg), P<int>(c->st_b), P<uint32_t>(c->st_a), c->Ep, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (u->active_flag)
      for (size_t p = 0; p < c->Ep; ++p)
        if (c->lay.pos_edge[p] != ~0u) c->active_host[p] = u->active_flag[c->lay.pos_edge[p]] == 1;
  }
  if (u->cam_priors_eta && u->cam_priors_lambda) {
    std::vector<float> rec;
    pack_cam(u->cam_priors_eta, u->cam_priors_lambda, c->C, rec);
    HIPCHK(c, hipMemcpy(c->camp.p, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
  }
  if (u->lmk_priors_eta && u->lmk_priors_lambda && c->L_loc) {
    std::vector<float> rec;
    pack_lmk(u->lmk_priors_eta, u->lmk_priors_lambda, c->lmk_begin, c->L_loc, rec);
    HIPCHK(c, hipMemcpy(c->lmkp.p, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
  }
  if (u->cam_weaken_flag) HIPCHK(c, hipMemcpy(c->cwf.p, u->cam_weaken_flag, (size_t)c->C * 4, hipMemcpyHostToDevice));
  if (u->lmk_weaken_flag && c->L_loc)
    HIPCHK(c, hipMemcpy(c->lwf.p, u->lmk_weaken_flag + c->lmk_begin, (size_t)c->L_loc * 4, hipMemcpyHostToDevice));
  return refresh_beliefs_from_partials(c, false);
}

// eval_reprojection_error (util.cpp:74-144) + counters (ba.cpp:1011-1020) over the local shard
// The metric in two halves, so that a caller printing it every iteration (the reference's default loop) can queue the
// NEXT GBP iteration before it waits for the previous metric: begin enqueues k_means + k_eval, which write their result
// DIRECTLY into pinned, device-mapped host memory (slot 0 = the two health counters, slots 1..nb = per-block partials;
// no copy launch) and records an event; end waits for that event only and sums the partials in block order.  Two
// evaluations may be in flight (two result areas).  The health counters are accumulated with atomics in device memory,
// double-buffered so that no memset launch is needed: k_means zeroes the pair the next evaluation will use.
static int eval_alloc(gbp_ctx* c) {
  if (c->eval_host) return GBP_OK;
  HIPCHK(c, hipHostMalloc(&c->eval_host, sizeof(DeviceEval) * 1025 * 2, hipHostMallocMapped));
  HIPCHK(c, hipHostGetDevicePointer(&c->eval_host_dev, c->eval_host, 0));
  HIPCHK(c, hipEventCreateWithFlags(&c->eval_ev[0], hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->eval_ev[1], hipEventDisableTiming));
  return GBP_OK;
}

// k_means + k_eval of the current beliefs into result area `area`, its event recorded behind them
static int eval_enqueue(gbp_ctx* c, int area) {
  DeviceEval* slots = static_cast<DeviceEval*>(c->eval_host_dev) + 1025 * area;
  unsigned long long* h_cur = P<unsigned long long>(c->health) + 2 * area;
  unsigned long long* h_next = P<unsigned long long>(c->health) + 2 * (area ^ 1);
  launch_means(P<float4>(c->camb), P<float4>(c->lmkb), P<float>(c->cam_mu), P<float>(c->lmk_mu), c->C, c->L_loc,
               h_cur, h_next, /*count_cams=*/c->rank == 0, c->stream);
  launch_eval(P<uint32_t>(c->row_cam), P<uint32_t>(c->lmk_idx), P<float4>(c->lmsg), P<float4>(c->fac), P<float>(c->cam_mu), P<float>(c->lmk_mu),
              P<float>(c->dK), c->prm.num_undamped_iters, slots + 1, h_cur, reinterpret_cast<unsigned long long*>(slots), c->n_tiles, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->eval_ev[area], c->stream));
  c->eval_per_wave[area] = false;
  return GBP_OK;
}

static int eval_begin_impl(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_eval: upload first");
  if (c->eval_pending >= 2) return fail(c, GBP_ERR_STATE, "gbp_eval_begin: two evaluations already in flight, call gbp_eval_end first");
  if (int rc = settle(c)) return rc;
  if (int rc = eval_alloc(c)) return rc;
  const int area = c->eval_parity & 1;
  if (int rc = eval_enqueue(c, area)) return rc;
  c->eval_parity ^= 1;
  c->eval_pending += 1;
  return GBP_OK;
}

// part[0] = the two health counters; then one record per workgroup of k_eval (per_wave = false: nb of them), or one per tile
// wave of k_persist (per_wave = true: n_tiles of them) — the four waves of a workgroup added as k_eval's block reduction adds
// them, ((w0 + w1) + w2) + w3, then the workgroups in order: the same fp64 additions in the same order either way.
static int sum_eval(gbp_ctx* c, const DeviceEval* part, uint32_t nb, gbp_eval_out* o, bool per_wave) {
  std::memset(o, 0, sizeof(*o));
  if (per_wave) {
    for (uint32_t b = 0; b < nb; ++b) {
      DeviceEval w[4] = {};
      for (uint32_t k = 0; k < 4; ++k)
        if (b * 4 + k < c->n_tiles) w[k] = part[1 + b * 4 + k];
      o->sum_norm += ((w[0].sum_norm + w[1].sum_norm) + w[2].sum_norm) + w[3].sum_norm;
      o->sum_half_sq += ((w[0].sum_half_sq + w[1].sum_half_sq) + w[2].sum_half_sq) + w[3].sum_half_sq;
      for (uint32_t k = 0; k < 4; ++k) { o->n_active += w[k].n_active; o->n_relin += w[k].n_relin; o->n_robust += w[k].n_robust; }
    }
  } else {
    for (uint32_t b = 1; b <= nb; ++b) {
      o->sum_norm += part[b].sum_norm; o->sum_half_sq += part[b].sum_half_sq;
      o->n_active += part[b].n_active; o->n_relin += part[b].n_relin; o->n_robust += part[b].n_robust;
    }
  }
  // non-finite guard (replaces the Poplar FP traps of ba.cpp:888-891) + non-PD belief count;
  // cameras are replicated, so only rank 0 counts them
  unsigned long long h[2];
  std::memcpy(h, part, 16);
  o->n_nonfinite = h[0];
  o->n_nonpd = h[1];
  return GBP_OK;
}

static int eval_end_impl(gbp_ctx* c, gbp_eval_out* o) {
  if (!c || !o) return GBP_ERR_INVALID;
  if (c->eval_pending < 1) return fail(c, GBP_ERR_STATE, "gbp_eval_end: no evaluation in flight");
  std::memset(o, 0, sizeof(*o));
  const int area = (c->eval_parity + (c->eval_pending == 2 ? 0 : 1)) & 1;   // the OLDEST pending evaluation
  HIPCHK(c, hipEventSynchronize(c->eval_ev[area]));
  if (!c->persist_log.empty()) {
    // the metric may have come out of a k_persist launch: that launch (the oldest logged one for this area) and everything
    // before it have completed — validate them; after a time-out the recovery has re-queued the metric behind the replay
    unsigned upto = 0;
    for (const gbp_ctx::Burst& b : c->persist_log)
      if (b.mode == 1 && b.area == area) { upto = b.seq; break; }
    const bool failed = *static_cast<volatile unsigned*>(c->pstatus_host) != 0u;
    if (failed || upto)
      if (int rc = persist_check(c, upto)) return rc;
    if (failed) HIPCHK(c, hipEventSynchronize(c->eval_ev[area]));
  }
  c->eval_pending -= 1;
  return sum_eval(c, static_cast<const DeviceEval*>(c->eval_host) + 1025 * area, eval_blocks(c->n_tiles), o, c->eval_per_wave[area]);
}

static int eval_impl(gbp_ctx* c, gbp_eval_out* o) {
  if (!c || !o || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_eval: upload first");
  if (c->eval_pending) return fail(c, GBP_ERR_STATE, "gbp_eval: finish the evaluations in flight (gbp_eval_end) first");
  if (int rc = eval_begin_impl(c)) return rc;
  return eval_end_impl(c, o);
}

// gbp_iterate(n) followed by gbp_eval_begin() in one call.  On a graph that runs in k_persist the metric rides in the same
// launch (two more phases after the last belief update: what k_means and k_eval compute, bit for bit) — the reference's
// default loop prints the metric after EVERY iteration (ba.cpp:1009-1028), which otherwise costs four launches per iteration.
static int iterate_eval_impl(gbp_ctx* c, int n) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_iterate_eval: upload first");
  if (n <= 0) return eval_begin_impl(c);
  bool fused = false;
  if (c->eval_pending < 2 && n <= kPersistChunk && c->n_tiles <= 1024)
    if (int rc = persist_ready(c, &fused)) return rc;
  if (fused) {
    if (int rc = eval_alloc(c)) return rc;
    const int area = c->eval_parity & 1;
    DeviceEval* slots = static_cast<DeviceEval*>(c->eval_host_dev) + 1025 * area;
    PersistEval ev{};
    ev.on = 1;
    ev.cam_mu = P<float>(c->cam_mu); ev.lmk_mu = P<float>(c->lmk_mu);
    ev.num_undamped = c->prm.num_undamped_iters;
    ev.slots = slots;                    // [0] = health copy, [1 + tile wave] = partial sums
    ev.health = P<unsigned long long>(c->health) + 2 * area;
    ev.health_next = P<unsigned long long>(c->health) + 2 * (area ^ 1);
    ev.health_each = P<unsigned long long>(c->health);
    gbp_ctx::Span sp{};
    if (int rc = span_begin(c, sp)) return rc;
    // A long burst with ONE metric at its end: all but the last iteration in the launch that carries no metric code at all (the
    // instantiation with the metric runs every iteration ~0.5 us slower: 59 us per 100 iterations on fr1xyz against ~10 us for
    // one more launch), the last iteration and the metric in a launch of their own.
    int head = n >= 16 ? n - 1 : 0;
    int lrc = head ? launch_persist_burst(c, sweep_args(c), head, nullptr, 0, 0) : GBP_OK;
    if (lrc == kNotLaunched) head = 0;      // nothing ran: everything on the two-kernel path below
    if (lrc == GBP_OK) lrc = launch_persist_burst(c, sweep_args(c), n - head, &ev, 1, area);
    if (lrc == kNotLaunched && head) {      // the head ran, the ctx then left the persistent path: the last iteration and the metric on the two-kernel path
      if (int rc = span_end(c, sp)) return rc;
      c->timed_iters += (uint64_t)head;
      c->beliefs_valid = true;
      if (int rc = iterate_impl(c, n - head)) return rc;
      return eval_begin_impl(c);
    }
    if (lrc == GBP_OK) {
      if (int rc = span_end(c, sp)) return rc;
      c->timed_iters += (uint64_t)n;
      c->beliefs_valid = true;
      HIPCHK(c, hipEventRecord(c->eval_ev[area], c->stream));
      c->eval_per_wave[area] = true;
      c->eval_parity ^= 1;
      c->eval_pending += 1;
      return GBP_OK;
    }
    c->span_pool.push_back(sp);
    if (lrc != kNotLaunched) return lrc;
  }
  if (int rc = iterate_impl(c, n)) return rc;
  return eval_begin_impl(c);
}

// gbp_iterate_eval_each on a graph that does not run in k_persist: the metric of iteration k rides in the sweep of iteration
// k + 1 (k_sweep<EV>, k_beliefs<EV>: EvalRide in gbp_kernels.h), the iterations replay from a hipGraph like gbp_iterate's, the
// host is not involved until the burst has ended: per piece of at most ev_depth iterations (the ring of per-tile records) one
// k_eval_ride for the piece's last iteration and one k_eval_fold, which reduces every slot to one 56-byte result in host-mapped
// memory.  Bit-identical to gbp_iterate(1) + gbp_eval per iteration (same operations, same order of the sums).
static int ev_alloc(gbp_ctx* c) {
  if (c->ev_depth) return GBP_OK;
  const size_t slot_bytes = (size_t)c->n_tiles * sizeof(EvalRec);
  const uint32_t depth = (uint32_t)std::min<size_t>(256, std::max<size_t>(2, ((size_t)128 << 20) / slot_bytes));      // <= 128 MB of ring
  if (int rc = dev_alloc(c, c->ev_cam, (size_t)c->C * 3 * 16)) return rc;
  if (int rc = dev_alloc(c, c->ev_lmk, (size_t)c->L_loc * 16)) return rc;
  if (int rc = dev_alloc(c, c->ev_part, slot_bytes * depth)) return rc;
  if (int rc = dev_alloc(c, c->ev_ctl, 64 + (size_t)depth * 16)) return rc;
  // dev_alloc zero-fills on the NULL stream, which the ctx's non-blocking stream does not order against — and hipMemset of device
  // memory may return before the fill has run: without this wait the fill raced the first burst's records (seen: a first metric
  // over 3 062 of 200 000 factors)
  HIPCHK(c, hipDeviceSynchronize());
  c->ev_depth = depth;
  return GBP_OK;
}
static int eval_each_ride(gbp_ctx* c, int n, gbp_eval_out* out) {
  if (int rc = settle(c)) return rc;
  if (int rc = ev_alloc(c)) return rc;
  if ((size_t)n > c->ev_host_cap) {       // one 56-byte result per iteration of the burst, host-mapped
    if (c->ev_host) { (void)hipHostFree(c->ev_host); c->ev_host = nullptr; c->ev_host_cap = 0; }
    const size_t cap = std::max<size_t>(1024, (size_t)n);
    HIPCHK(c, hipHostMalloc(&c->ev_host, sizeof(gbp_eval_out) * cap, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer(&c->ev_host_dev, c->ev_host, 0));
    c->ev_host_cap = cap;
  }
  static_assert(sizeof(gbp_eval_out) == 56, "k_eval_fold writes gbp_eval_out records");
  SweepArgs a = sweep_args(c);
  a.ev = eval_ride(c);
  gbp_ctx::Span sp{};
  if (int rc = span_begin(c, sp)) return rc;
  for (int done = 0; done < n;) {         // pieces of at most ev_depth iterations, queued behind each other: no host wait in between
    const int m = std::min(n - done, (int)c->ev_depth);
    HIPCHK(c, hipMemsetAsync(c->ev_ctl.p, 0, 64, c->stream));      // iteration counter and health words of this piece
    if (int rc = iterate_plain(c, a, m, true)) { c->span_pool.push_back(sp); return rc; }
    launch_eval_ride(a.ev, P<uint32_t>(c->row_cam), P<uint32_t>(c->lmk_idx), P<float4>(c->lmsg), P<float4>(c->fac), P<float>(c->dK), c->stream);
    launch_eval_fold(a.ev, (uint32_t)m, static_cast<gbp_eval_out*>(c->ev_host_dev) + done, c->stream);
    HIPCHK(c, hipGetLastError());
    done += m;
  }
  if (int rc = span_end(c, sp)) return rc;
  c->timed_iters += (uint64_t)n;
  c->beliefs_valid = true;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(out, c->ev_host, sizeof(gbp_eval_out) * (size_t)n);
  return GBP_OK;
}

// n iterations with the metric after EVERY one of them (the reference's default loop, ba.cpp:1009-1028 / slam.cpp), blocking:
// out[k] = what gbp_iterate(1) + gbp_eval_global() would have returned for the k-th of them.  On a graph that runs in
// k_persist a burst is ONE launch: the metric of iteration k rides in the sweep phase of iteration k + 1 (both only read the
// beliefs), its partial sums go to host-mapped memory.  Everywhere else it is the loop it replaces, two metrics in flight.
static int iterate_eval_each_impl(gbp_ctx* c, int n, gbp_eval_out* out) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_iterate_eval_each: upload first");
  if (n < 0 || (n > 0 && !out)) return fail(c, GBP_ERR_INVALID, "gbp_iterate_eval_each: n >= 0 and an array of n results");
  if (c->eval_pending) return fail(c, GBP_ERR_STATE, "gbp_iterate_eval_each: finish the evaluations in flight (gbp_eval_end) first");
  if (int rc = settle(c)) return rc;                  // blocking call: nothing of this ctx stays in flight across it
  const uint32_t nb = eval_blocks(c->n_tiles);
  bool fused = false;
  if (nb == (c->n_tiles + 3) / 4)
    if (int rc = persist_ready(c, &fused)) return rc;
  int done = 0;
  if (fused) {
    const uint32_t stride = c->n_tiles + 1;          // [0] = health copy, then one record per tile wave
    if (!c->series_host) {
      HIPCHK(c, hipHostMalloc(&c->series_host, sizeof(DeviceEval) * (size_t)stride * kSeriesMax, hipHostMallocMapped));
      HIPCHK(c, hipHostGetDevicePointer(&c->series_dev, c->series_host, 0));
    }
    const int area = c->eval_parity & 1;   // both health areas are zero between evaluations; this launch leaves them so
    while (done < n && fused) {
      const int m = std::min(n - done, (int)kSeriesMax);
      PersistEval ev{};
      ev.on = 1; ev.each = 1; ev.stride = stride;
      ev.cam_mu = P<float>(c->cam_mu); ev.lmk_mu = P<float>(c->lmk_mu);
      ev.num_undamped = c->prm.num_undamped_iters;
      ev.slots = static_cast<DeviceEval*>(c->series_dev);
      ev.health = P<unsigned long long>(c->health) + 2 * area;
      ev.health_next = P<unsigned long long>(c->health) + 2 * (area ^ 1);
      ev.health_each = P<unsigned long long>(c->health);
      gbp_ctx::Span sp{};
      if (int rc = span_begin(c, sp)) return rc;
      const int lrc = launch_persist_burst(c, sweep_args(c), m, &ev, 2, area);
      if (lrc != GBP_OK) {
        c->span_pool.push_back(sp);
        if (lrc != kNotLaunched) return lrc;
        fused = false;
        break;
      }
      if (int rc = span_end(c, sp)) return rc;
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const bool failed = *static_cast<volatile unsigned*>(c->pstatus_host) != 0u;
      if (int rc = persist_check(c, 0)) return rc;      // a time-out: state restored to the start of this burst
      if (failed) { fused = false; break; }             // ... which the plain loop below now runs
      c->timed_iters += (uint64_t)m;
      c->beliefs_valid = true;
      for (int k = 0; k < m; ++k) sum_eval(c, static_cast<const DeviceEval*>(c->series_host) + (size_t)k * stride, nb, out + done + k, true);
      done += m;
    }
  }
  if (done < n && !c->comm && c->world == 1 && c->hoist && !c->profile_stages && !stream_is_capturing(c))
    return eval_each_ride(c, n - done, out + done);
  int collected = done;
  for (int k = done; k < n; ++k) {
    if (int rc = iterate_impl(c, 1)) return rc;
    if (int rc = eval_begin_impl(c)) return rc;
    if (c->eval_pending == 2) { if (int rc = eval_end_impl(c, out + collected)) return rc; ++collected; }
  }
  while (collected < n) { if (int rc = eval_end_impl(c, out + collected)) return rc; ++collected; }
  return GBP_OK;
}

// n passes of the body of the reference's iteration loop (ba.cpp:1001-1028) from loop index iter0: WEAKEN_PRIORS in front of pass i
// iff (i + 1) % 2 == 0 and i < 2 * steps, GBP_PROG, the metric.  On a graph that runs in the persistent kernel the passes between two
// host events are ONE launch however many weakenings lie between them (k_persist_flow applies WeakenPriorVertex itself, in front of
// the iterations the loop weakens before; only a weakening in front of a launch's FIRST iteration is a launch of its own);
// everywhere else — and after a recovered time-out — it is the calls it stands for, in the loop's order.
static int ba_loop_impl(gbp_ctx* c, int n, unsigned iter0, unsigned steps, gbp_eval_out* out) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_ba_loop: upload first");
  if (n < 0) return fail(c, GBP_ERR_INVALID, "gbp_ba_loop: n >= 0");
  const auto weak = [&](unsigned i) { return ((i + 1u) % 2u == 0u) && i < 2u * steps; };
  if (!out) {
    // without the metric (out == NULL): not blocking, like gbp_iterate.  The weakening in front of the first pass is a launch of its
    // own; every later one rides in the launch of the persistent kernel, or — two-kernel path, single-GPU ctx — in the belief update
    // of the iteration before it.  A sharded ctx takes the calls one by one.
    int done = 0;
    while (done < n) {
      const unsigned i0 = iter0 + (unsigned)done;
      if (weak(i0))
        if (int rc = weaken_priors_impl(c)) return rc;
      if (c->comm || c->world != 1 || c->profile_stages || stream_is_capturing(c)) {      // (per-stage timing and a caller's capture keep their own paths)
        int m = 1;
        while (done + m < n && !weak(iter0 + (unsigned)(done + m))) ++m;
        if (int rc = iterate_impl(c, m)) return rc;
        done += m;
        continue;
      }
      const int m = std::min(n - done, kPersistChunk);
      bool persist = false;
      if (m >= 2 && c->persist_flow && c->flow.lmsg)      // (a single iteration is as fast from two launches: gbp_iterate's rule)
        if (int rc = persist_ready(c, &persist)) return rc;
      if (!persist)
        if (int rc = settle(c)) return rc;
      gbp_ctx::Span sp{};
      if (int rc = span_begin(c, sp)) return rc;
      int lrc = kNotLaunched;
      if (persist) lrc = launch_persist_burst(c, sweep_args(c), m, nullptr, 0, 0, i0, 2u * steps);
      if (lrc == kNotLaunched) lrc = iterate_weaken_plain(c, sweep_args(c), m, i0, 2u * steps);
      if (lrc != GBP_OK) { c->span_pool.push_back(sp); return lrc; }
      if (int rc = span_end(c, sp)) return rc;
      c->timed_iters += (uint64_t)m;
      c->beliefs_valid = true;
      done += m;
    }
    return GBP_OK;
  }
  if (c->eval_pending) return fail(c, GBP_ERR_STATE, "gbp_ba_loop: finish the evaluations in flight (gbp_eval_end) first");
  int done = 0;
  while (done < n) {
    const unsigned i0 = iter0 + (unsigned)done;
    if (weak(i0))
      if (int rc = weaken_priors_impl(c)) return rc;
    bool fused = false;
    const uint32_t nb = eval_blocks(c->n_tiles);
    if (int rc = settle(c)) return rc;
    if (nb == (c->n_tiles + 3) / 4 && c->persist_flow && c->flow.lmsg)
      if (int rc = persist_ready(c, &fused)) return rc;
    if (fused) {
      const uint32_t stride = c->n_tiles + 1;
      if (!c->series_host) {
        HIPCHK(c, hipHostMalloc(&c->series_host, sizeof(DeviceEval) * (size_t)stride * kSeriesMax, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer(&c->series_dev, c->series_host, 0));
      }
      const int area = c->eval_parity & 1;
      const int m = std::min(n - done, (int)kSeriesMax);
      PersistEval ev{};
      ev.on = 1; ev.each = 1; ev.stride = stride;
      ev.cam_mu = P<float>(c->cam_mu); ev.lmk_mu = P<float>(c->lmk_mu);
      ev.num_undamped = c->prm.num_undamped_iters;
      ev.slots = static_cast<DeviceEval*>(c->series_dev);
      ev.health = P<unsigned long long>(c->health) + 2 * area;
      ev.health_next = P<unsigned long long>(c->health) + 2 * (area ^ 1);
      ev.health_each = P<unsigned long long>(c->health);
      gbp_ctx::Span sp{};
      if (int rc = span_begin(c, sp)) return rc;
      const int lrc = launch_persist_burst(c, sweep_args(c), m, &ev, 2, area, i0, 2u * steps);
      if (lrc == GBP_OK) {
        if (int rc = span_end(c, sp)) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const bool failed = *static_cast<volatile unsigned*>(c->pstatus_host) != 0u;
        if (int rc = persist_check(c, 0)) return rc;      // a time-out: state (priors and flags too) restored to the start of this burst
        if (!failed) {
          c->timed_iters += (uint64_t)m;
          c->beliefs_valid = true;
          for (int k = 0; k < m; ++k) sum_eval(c, static_cast<const DeviceEval*>(c->series_host) + (size_t)k * stride, nb, out + done + k, true);
          done += m;
          continue;
        }
      } else {
        c->span_pool.push_back(sp);
        if (lrc != kNotLaunched) return lrc;
      }
    }
    // the calls the loop stands for, up to (not including) its next weakening
    int m = 1;
    while (done + m < n && !weak(iter0 + (unsigned)(done + m))) ++m;
    if (int rc = iterate_eval_each_impl(c, m, out + done)) return rc;
    done += m;
  }
  return GBP_OK;
}

int gbp_timing(gbp_ctx* c, gbp_timing_out* t, int reset) {
  if (!c || !t) return GBP_ERR_INVALID;
  drain_sweep_events(c);   // split-phase brackets recorded by gbp_iterate_begin
  resolve_spans(c, true);  // gbp_iterate brackets still in flight
  t->sweep_ms = c->sweep_ms; t->belief_ms = c->belief_ms; t->total_ms = c->total_ms; t->iterations = c->timed_iters;
  t->exchange_ms = c->exchange_ms;
  t->algorithmic_bytes_per_iter = 1112ull * c->E_loc + 336ull * c->C + 96ull * c->L_loc;
  t->device_bytes_allocated = c->dev_bytes;
  if (reset) { c->sweep_ms = c->belief_ms = c->total_ms = c->exchange_ms = 0; c->timed_iters = 0; }
  return GBP_OK;
}

// ---- extras declared below the main program list -------------------------------------------------
int gbp_set_profiling(gbp_ctx* c, int per_stage_events) {
  if (!c) return GBP_ERR_INVALID;
  c->profile_stages = per_stage_events != 0;
  return GBP_OK;
}

#ifdef GBP_BUILD_TEST_HOOKS   // ---- test hooks (include/gbp_mi355x_debug.h): only in libgbp_mi355x_test.so ----
// Raw internal state in the reference's tensor layouts, for stage-level parity tests.
//   what = 0: factor_potentials_eta [9E] + factor_potentials_lambda [81E] = [cc36|cl18|lc18|ll9] (ba.cpp:93-96)
//   what = 1: factor->camera messages as stored: eta [6E] + Lambda [36E] (lower triangle; upper = 0)
//   what = 2: factor->landmark messages: eta [3E] + Lambda [9E]
//   what = 3: mu [9E] + dmu [E]
// Entries of factors outside the local shard are left untouched.
static int debug_get_impl(gbp_ctx* c, int what, float* a, float* b) {
  if (!c || !a || !b) return GBP_ERR_INVALID;
  if (int rc = gbp_sync(c)) return rc;
  auto tri = [](int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; };
  if (what == 0) {
    std::vector<float> f((size_t)c->Ep * kFacG * 4);
    HIPCHK(c, hipMemcpy(f.data(), c->fac.p, f.size() * 4, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < c->Ep; ++p) {
      const uint32_t e = c->lay.pos_edge[p];
      if (e == ~0u) continue;
      auto F = [&](int i) { return f[tile_off((uint32_t)p, kFacG, i)]; };
      for (int i = 0; i < 9; ++i) a[(size_t)e * 9 + i] = F(i);
      float* lam = b + (size_t)e * 81;
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) lam[i * 6 + j] = F(9 + tri(i, j));
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) { lam[36 + i * 3 + j] = F(30 + i * 3 + j); lam[54 + j * 6 + i] = F(30 + i * 3 + j); }
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) lam[72 + i * 3 + j] = F(48 + tri(i, j));
    }
  } else if (what == 1) {
    std::vector<float> f((size_t)c->Ep * kCmsgG * 4);
    HIPCHK(c, hipMemcpy(f.data(), c->cmsg.p, f.size() * 4, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < c->Ep; ++p) {
      const uint32_t e = c->lay.pos_edge[p];
      if (e == ~0u) continue;
      for (int i = 0; i < 6; ++i) a[(size_t)e * 6 + i] = f[tile_off((uint32_t)p, kCmsgG, i)];
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j)
        b[(size_t)e * 36 + i * 6 + j] = (i >= j) ? f[tile_off((uint32_t)p, kCmsgG, 6 + tri(i, j))] : 0.f;
    }
  } else if (what == 2) {
    std::vector<float> f((size_t)c->Ep * 16);
    HIPCHK(c, hipMemcpy(f.data(), c->lmsg.p, f.size() * 4, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < c->Ep; ++p) {
      const uint32_t e = c->lay.pos_edge[p];
      if (e == ~0u) continue;
      const float* r = &f[p * 16];
      for (int i = 0; i < 3; ++i) a[(size_t)e * 3 + i] = r[i];
      for (int i = 0; i < 9; ++i) b[(size_t)e * 9 + i] = r[4 + i];
    }
  } else if (what == 3 && c->hoist) {
    // hoisted mode: mu of a factor = the per-variable means its last sweep used; dmu is not kept per factor
    std::vector<float> mc((size_t)c->C * 16), ml((size_t)c->L_loc * 8);
    HIPCHK(c, hipMemcpy(mc.data(), c->hmu_c.p, mc.size() * 4, hipMemcpyDeviceToHost));
    if (c->L_loc) HIPCHK(c, hipMemcpy(ml.data(), c->hmu_l.p, ml.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> rec;
    if (int rc = download_lmsg(c, rec)) return rc;
    for (size_t p = 0; p < c->Ep; ++p) {
      const uint32_t e = c->lay.pos_edge[p];
      if (e == ~0u) continue;
      if (!(get_state(rec, p).flags & kFlagActive)) continue;  // inactive factors never update mu (gbp_codelets.cpp:242)
      for (int i = 0; i < 6; ++i) a[(size_t)e * 9 + i] = mc[(size_t)c->lay.pos_cam[p] * 16 + 8 + i];
      for (int i = 0; i < 3; ++i) a[(size_t)e * 9 + 6 + i] = ml[(size_t)c->lay.pos_lmk_loc[p] * 8 + 4 + i];
      b[e] = 0.f;
    }
  } else if (what == 3) {
    std::vector<float> f((size_t)c->Ep * kMuG * 4);
    HIPCHK(c, hipMemcpy(f.data(), c->mu.p, f.size() * 4, hipMemcpyDeviceToHost));
    for (size_t p = 0; p < c->Ep; ++p) {
      const uint32_t e = c->lay.pos_edge[p];
      if (e == ~0u) continue;
      for (int i = 0; i < 9; ++i) a[(size_t)e * 9 + i] = f[tile_off((uint32_t)p, kMuG, i)];
      b[e] = f[tile_off((uint32_t)p, kMuG, 9)];
    }
  } else {
    return fail(c, GBP_ERR_INVALID, "gbp_debug_get: unknown selector");
  }
  return GBP_OK;
}

// Timing experiment: average duration (us) of `reps` launches of an ablated k_sweep (see gbp_kernels.hip).
// The ctx state is garbage afterwards; upload again before using it.
int gbp_debug_time_sweep(gbp_ctx* c, int ablation, int reps, double* avg_us) {
  if (!c || !avg_us || reps <= 0 || !c->uploaded) return GBP_ERR_INVALID;
  if (int rc = settle(c)) return rc;
  const SweepArgs a = sweep_args(c);
  bool built = true;
  auto one = [&]() {
    if (ablation >= 100 && ablation <= 102) {  // 100: k_beliefs, 101: camera part only, 102: landmark part only
      launch_beliefs(belief_args(c), ablation != 102, ablation != 101, c->stream);
    } else if (ablation == 0) {
      launch_sweep(a, c->n_tiles, c->hoist, c->stream);
    } else {
#ifdef GBP_BUILD_EXPERIMENTS
      built = lab_launch_sweep_ablated(a, c->n_tiles, ablation, c->stream) && built;
#else
      built = false;
#endif
    }
  };
  one();
  if (!built) return fail(c, GBP_ERR_INVALID, "gbp_debug_time_sweep: ablated sweeps exist in the experiments build only (python -m gbp_poplar_amd.build --experiments)");
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  for (int i = 0; i < reps; ++i) one();
  HIPCHK(c, hipEventRecord(c->ev2, c->stream));
  HIPCHK(c, hipEventSynchronize(c->ev2));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev1, c->ev2));
  *avg_us = 1e3 * ms / reps;
  return GBP_OK;
}

// ---- the device order without a device (gbp_layout.cpp): what gbp_create builds, handed out for CPU property tests ----
struct gbp_layout { Layout y; };
static LayoutOptions to_options(const gbp_layout_options* o) {
  LayoutOptions r;
  if (o) {
    r.row_placement = o->row_placement; r.row_window = o->row_window; r.row_place_max_deg = o->row_place_max_deg;
    r.row_key_lane = o->row_key_lane; r.classes = o->classes; r.tile_window = o->tile_window; r.tile_min_tiles = o->tile_min_tiles;
    r.tile_identity = o->tile_identity; r.row_sort_in_class = o->row_sort_in_class;
  }
  return r;
}
void gbp_debug_layout_default_options(gbp_layout_options* o) {
  if (!o) return;
  const LayoutOptions d;
  o->row_placement = d.row_placement; o->row_window = d.row_window; o->row_place_max_deg = d.row_place_max_deg;
  o->row_key_lane = d.row_key_lane; o->classes = d.classes; o->tile_window = d.tile_window; o->tile_min_tiles = d.tile_min_tiles;
  o->tile_identity = d.tile_identity; o->row_sort_in_class = d.row_sort_in_class;
}
int gbp_debug_layout_options(const gbp_layout_options* o) { g_layout_options = to_options(o); return GBP_OK; }
int gbp_debug_force_sweep_policy(int policy) { g_force_sweep_policy = policy; return GBP_OK; }
int gbp_debug_persist_roles(uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, int with_metric, uint32_t* dims, uint32_t* role, uint32_t cap) {
  if (!dims) return GBP_ERR_INVALID;
  const PersistGrid pg = persist_grid(n_tiles, n_cams, n_lmks, with_metric != 0);
  dims[0] = pg.nb; dims[1] = pg.separate; dims[2] = pg.n_met; dims[3] = (n_lmks + 15) / 16;
  if (role) {
    if (cap < pg.nb * 4u) return GBP_ERR_INVALID;
    for (uint32_t b = 0; b < pg.nb; ++b)
      for (uint32_t w = 0; w < 4; ++w) role[b * 4 + w] = persist_role(b, w, pg.nb, n_tiles, n_cams, dims[3], pg.n_met, pg.separate);
  }
  return GBP_OK;
}
int gbp_debug_persist_flow(gbp_ctx* c, int on) {
  if (!c) return GBP_ERR_INVALID;
  c->persist_flow = on != 0;
  return GBP_OK;
}
int gbp_debug_layout_build(const gbp_problem* pr, int tile_order, const gbp_shard* sh, const gbp_layout_options* o, gbp_layout** out) {
  if (!out) return GBP_ERR_INVALID;
  return guarded(nullptr, "gbp_debug_layout_build", [&] {
    gbp_layout* h = new gbp_layout();
    std::string err;
    const int rc = layout_build(pr, tile_order, sh, o ? to_options(o) : g_layout_options, h->y, err);
    if (rc != GBP_OK) { delete h; return fail(nullptr, rc, err); }
    *out = h;
    return (int)GBP_OK;
  });
}
int gbp_debug_layout_dims(const gbp_layout* h, uint32_t* d) {
  if (!h || !d) return GBP_ERR_INVALID;
  const Layout& y = h->y;
  const uint32_t v[11] = {y.C, y.L, y.E, y.lmk_begin, y.lmk_end, y.L_loc, y.E_loc, y.n_rows, y.n_tiles, y.Ep, y.row_window};
  std::memcpy(d, v, sizeof(v));
  return GBP_OK;
}
int gbp_debug_layout_array(const gbp_layout* h, int which, const uint32_t** data, size_t* n) {
  if (!h || !data || !n) return GBP_ERR_INVALID;
  const Layout& y = h->y;
  const std::vector<uint32_t>* a[11] = {&y.pos_edge, &y.pos_cam, &y.pos_lmk_loc, &y.pos_lpos, &y.cam_row_ptr, &y.row_slot, &y.row_cam,
                                        &y.lmk_ptr, &y.lmk_fpos, &y.lmk_ix, &y.tile_perm};
  if (which < 0 || which > 10) return GBP_ERR_INVALID;
  *data = a[which]->data(); *n = a[which]->size();
  return GBP_OK;
}
void gbp_debug_layout_free(gbp_layout* h) { delete h; }
int gbp_debug_tile_order_local(const uint8_t* tile_class, uint32_t n_tiles, uint32_t window, uint32_t n_classes, uint32_t* perm) {
  if (!tile_class || !perm || window == 0 || n_classes == 0) return GBP_ERR_INVALID;
  return guarded(nullptr, "gbp_debug_tile_order_local", [&] { tile_order_local(tile_class, n_tiles, window, n_classes, perm); return (int)GBP_OK; });
}

// Inverse of gbp_debug_get(what = 0): overwrite the factor potentials (lower triangles of the
// symmetric blocks and Lambda_cl are taken; Lambda_lc is implied).  Test hook only.
static int debug_set_factor_potentials_impl(gbp_ctx* c, const float* eta9E, const float* lam81E) {
  if (!c || !eta9E || !lam81E) return GBP_ERR_INVALID;
  if (int rc = gbp_sync(c)) return rc;
  auto tri = [](int i, int j) { return i * (i + 1) / 2 + j; };
  std::vector<float> f((size_t)c->Ep * kFacG * 4);
  HIPCHK(c, hipMemcpy(f.data(), c->fac.p, f.size() * 4, hipMemcpyDeviceToHost));
  for (size_t p = 0; p < c->Ep; ++p) {
    const uint32_t e = c->lay.pos_edge[p];
    if (e == ~0u) continue;
    const float* lam = lam81E + (size_t)e * 81;
    for (int i = 0; i < 9; ++i) f[tile_off((uint32_t)p, kFacG, i)] = eta9E[(size_t)e * 9 + i];
    for (int i = 0; i < 6; ++i) for (int j = 0; j <= i; ++j) f[tile_off((uint32_t)p, kFacG, 9 + tri(i, j))] = lam[i * 6 + j];
    for (int i = 0; i < 18; ++i) f[tile_off((uint32_t)p, kFacG, 30 + i)] = lam[36 + i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j <= i; ++j) f[tile_off((uint32_t)p, kFacG, 48 + tri(i, j))] = lam[72 + i * 3 + j];
  }
  HIPCHK(c, hipMemcpy(c->fac.p, f.data(), f.size() * 4, hipMemcpyHostToDevice));
  return GBP_OK;
}

#endif  // GBP_BUILD_TEST_HOOKS

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
  int rc = GBP_OK;
  if (!c->xsend.p) rc = dev_alloc(c, c->xsend, (size_t)c->C * kCamRec * 4);
  if (rc == GBP_OK && !c->xrecv.p) rc = dev_alloc(c, c->xrecv, (size_t)c->world * c->C * kCamRec * 4);
  if (rc != GBP_OK) return rc;
  HIPCHK(c, hipDeviceSynchronize());      // (dev_alloc's zero-fill runs on the NULL stream)
  c->send_dev = c->xsend.p; c->recv_dev = c->xrecv.p;
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

int gbp_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int gbp_set_device(int device) {
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, GBP_ERR_NO_DEVICE, "gbp_set_device: no such device");
  return GBP_OK;
}

size_t gbp_comm_region_bytes(uint32_t n_cams, int world) { return gbp::comm_region_bytes(n_cams, world); }

int gbp_comm_region_init(void* region, size_t bytes, uint32_t n_cams, int world) {
  return gbp::comm_region_init(region, bytes, n_cams, world) == 0 ? GBP_OK : GBP_ERR_INVALID;
}

void gbp_comm_region_abort(void* region) { gbp::comm_region_abort(region); }

int gbp_comm_region_selftest(void* region, int rank, int world, int rounds) {
  std::string err;
  const int rc = gbp::comm_region_selftest(region, rank, world, rounds, err);
  if (rc != 0) return fail(nullptr, GBP_ERR_COMM, "gbp_comm_region_selftest: " + err);
  return GBP_OK;
}

int gbp_comm_init(gbp_ctx* c, void* region, int transport) {
  if (!c || !region) return GBP_ERR_INVALID;
  if (c->comm) return fail(c, GBP_ERR_STATE, "gbp_comm_init: the ctx already has a communicator");
  std::string err;
  gbp::Comm* comm = gbp::comm_create_from_region(region, c->rank, c->world, transport, err);
  if (!comm) return fail(c, GBP_ERR_COMM, "gbp_comm_init: " + err);
  return comm_attach(c, comm);
}

int gbp_comm_unique_id(void* id128) {
  std::string err;
  if (!id128) return GBP_ERR_INVALID;
  if (gbp::comm_unique_id(id128, err) != 0) return fail(nullptr, GBP_ERR_COMM, "gbp_comm_unique_id: " + err);
  return GBP_OK;
}

int gbp_comm_init_rccl(gbp_ctx* c, const void* id128) {
  if (!c || !id128) return GBP_ERR_INVALID;
  if (c->comm) return fail(c, GBP_ERR_STATE, "gbp_comm_init_rccl: the ctx already has a communicator");
  std::string err;
  gbp::Comm* comm = gbp::comm_create_rccl(id128, c->rank, c->world, err);
  if (!comm) return fail(c, GBP_ERR_COMM, "gbp_comm_init_rccl: " + err);
  return comm_attach(c, comm);
}

// ---- what a first multi-GPU run wants on record (bench.py preflight) ---------------------------------------------------
int gbp_comm_describe(gbp_ctx* c, char* buf, size_t cap) {
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
int gbp_comm_set_schedule(gbp_ctx* c, int two_streams) {
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
int gbp_comm_probe(gbp_ctx* c, int reps, double* avg_us) {
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

int gbp_graph_state(const gbp_ctx* c) { return !c ? 0 : (c->persist_ok ? 2 : ((c->graph_exec || c->graph_exec_ev) ? 1 : (c->graph_failed ? -1 : 0))); }

const char* gbp_comm_transport(const gbp_ctx* c) { return (c && c->comm) ? c->comm->name() : "none"; }

int gbp_comm_barrier(gbp_ctx* c) {
  if (!c || !c->comm) return GBP_ERR_STATE;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  COMMCHK(c, c->comm->barrier(e_));
  return GBP_OK;
}

// gbp_eval over ALL shards: local sums gathered over the ranks and added in rank order (same bits on every rank)
static int eval_impl(gbp_ctx* c, gbp_eval_out* o);
int gbp_eval_global(gbp_ctx* c, gbp_eval_out* o) {
  if (!c || !o) return GBP_ERR_INVALID;
  const int rc = guarded(c, "gbp_eval_global", [&] { return eval_impl(c, o); });
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

#ifdef GBP_BUILD_TEST_HOOKS
// Device math layer on caller-supplied vectors (test hook, see k_debug_math): HIP vs the reference's own
// matlib.cpp / bafuncs.cpp outputs, no ctx and no restated vertex layer involved.
static int debug_math_run(int op, const float* in, float* out, int n, int reps, double* avg_us) {
  int in_w = 0, out_w = 0;
  if (!in || !out || n <= 0 || reps < 1 || !debug_math_widths(op, &in_w, &out_w))
    return fail(nullptr, GBP_ERR_INVALID, "gbp_debug_math: bad op / arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, GBP_ERR_NO_DEVICE, "gbp_debug_math: no HIP device (the product has no CPU fallback)");
  float *d_in = nullptr, *d_out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto done = [&](int rc, const char* what, hipError_t e) {
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc == GBP_OK ? rc : fail(nullptr, rc, std::string(what) + ": " + hipGetErrorString(e));
  };
  hipError_t e;
  if ((e = hipMalloc(&d_in, (size_t)n * in_w * 4)) != hipSuccess) return done(GBP_ERR_HIP, "hipMalloc", e);
  if ((e = hipMalloc(&d_out, (size_t)n * out_w * 4)) != hipSuccess) return done(GBP_ERR_HIP, "hipMalloc", e);
  if ((e = hipMemcpy(d_in, in, (size_t)n * in_w * 4, hipMemcpyHostToDevice)) != hipSuccess) return done(GBP_ERR_HIP, "hipMemcpy", e);
  launch_debug_math(op, d_in, d_out, n, nullptr);
  if ((e = hipGetLastError()) != hipSuccess) return done(GBP_ERR_HIP, "k_debug_math", e);
  if (avg_us) {   // back-to-back launches between two events
    if ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) return done(GBP_ERR_HIP, "hipEventCreate", e);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) launch_debug_math(op, d_in, d_out, n, nullptr);
    (void)hipEventRecord(e1, nullptr);
    if ((e = hipEventSynchronize(e1)) != hipSuccess) return done(GBP_ERR_HIP, "hipEventSynchronize", e);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *avg_us = 1e3 * ms / reps;
  }
  if ((e = hipMemcpy(out, d_out, (size_t)n * out_w * 4, hipMemcpyDeviceToHost)) != hipSuccess) return done(GBP_ERR_HIP, "hipMemcpy", e);
  return done(GBP_OK, "", hipSuccess);
}

int gbp_debug_math(int op, const float* in, float* out, int n) { return debug_math_run(op, in, out, n, 1, nullptr); }
int gbp_debug_math_timed(int op, const float* in, float* out, int n, int reps, double* avg_us) {
  if (!avg_us) return GBP_ERR_INVALID;
  return debug_math_run(op, in, out, n, reps, avg_us);
}

#endif  // GBP_BUILD_TEST_HOOKS

// ---- exported wrappers of the entry points that allocate host memory -----------------------------------------
int gbp_create(const gbp_problem* pr, const gbp_params* prm, const gbp_shard* sh, gbp_ctx** out) {
  return guarded(nullptr, "gbp_create", [&] { return create_impl(pr, prm, sh, out); });
}
int gbp_upload(gbp_ctx* c, const gbp_state_in* in) { return guarded(c, "gbp_upload", [&] { return upload_impl(c, in); }); }
int gbp_iterate(gbp_ctx* c, int n) { return guarded(c, "gbp_iterate", [&] { return iterate_impl(c, n); }); }
int gbp_linearise(gbp_ctx* c) { return guarded(c, "gbp_linearise", [&] { return linearise_impl(c); }); }
int gbp_iterate_begin(gbp_ctx* c) { return guarded(c, "gbp_iterate_begin", [&] { return iterate_begin_impl(c); }); }
int gbp_read(gbp_ctx* c, gbp_state_out* o) { return guarded(c, "gbp_read", [&] { return read_impl(c, o); }); }
int gbp_read_priors(gbp_ctx* c, gbp_priors_out* o) { return guarded(c, "gbp_read_priors", [&] { return read_priors_impl(c, o); }); }
int gbp_new_keyframe(gbp_ctx* c, const gbp_kf_update* u) { return guarded(c, "gbp_new_keyframe", [&] { return new_keyframe_impl(c, u); }); }
int gbp_eval(gbp_ctx* c, gbp_eval_out* o) { return guarded(c, "gbp_eval", [&] { return eval_impl(c, o); }); }
int gbp_eval_begin(gbp_ctx* c) { return guarded(c, "gbp_eval_begin", [&] { return eval_begin_impl(c); }); }
int gbp_eval_end(gbp_ctx* c, gbp_eval_out* o) { return guarded(c, "gbp_eval_end", [&] { return eval_end_impl(c, o); }); }
int gbp_iterate_eval(gbp_ctx* c, int n) { return guarded(c, "gbp_iterate_eval", [&] { return iterate_eval_impl(c, n); }); }
int gbp_iterate_eval_each(gbp_ctx* c, int n, gbp_eval_out* out) {
  return guarded(c, "gbp_iterate_eval_each", [&] { return iterate_eval_each_impl(c, n, out); });
}
int gbp_ba_loop(gbp_ctx* c, int n, unsigned iter0, unsigned steps, gbp_eval_out* out) {
  return guarded(c, "gbp_ba_loop", [&] { return ba_loop_impl(c, n, iter0, steps, out); });
}
#ifdef GBP_BUILD_TEST_HOOKS
int gbp_debug_get(gbp_ctx* c, int what, float* a, float* b) { return guarded(c, "gbp_debug_get", [&] { return debug_get_impl(c, what, a, b); }); }
int gbp_debug_set_factor_potentials(gbp_ctx* c, const float* eta9E, const float* lam81E) {
  return guarded(c, "gbp_debug_set_factor_potentials", [&] { return debug_set_factor_potentials_impl(c, eta9E, lam81E); });
}
#endif  // GBP_BUILD_TEST_HOOKS

}  // extern "C"
