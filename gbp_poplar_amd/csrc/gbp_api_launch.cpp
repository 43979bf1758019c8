// gbp_api_launch.cpp — what the programs of the list launch (include/gbp_mi355x.h):
//   gbp_linearise        LINEARISE_PROG   reference ba/ba.cpp:890-893   belief refresh + k_linearise
//   gbp_iterate          GBP_PROG x n     ba.cpp:895-905                persistent kernel / hipGraph replay / direct launches
//   gbp_weaken_priors    WEAKEN_PRIORS    ba.cpp:863-865                ONE k_beliefs launch (the prior owners scale on their way into the sums)
//   gbp_prepare          Engine::load     ba.cpp:936-937                hipGraph capture + instantiation, runs nothing
// and the split-phase form of the iteration (include/gbp_mi355x_multi.h) for callers that run the exchange themselves.
#include "gbp_ctx.hpp"

#include <algorithm>

using namespace gbp;
using namespace gbp::api;

namespace gbp {
namespace api {

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
  a.seg_live = c->use_seg_live ? P<uint32_t>(c->seg_live) : nullptr;
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


// camera beliefs from stored partials (single GPU: d_local; multi: recv_dev) + landmark beliefs re-summed.
// roll = true at the end of an iteration (the sweep has consumed the current means), false for
// prior-only refreshes (WEAKEN_PRIORS, NEW_KEYFRAME, LINEARISE).
int refresh_beliefs_from_partials(gbp_ctx* c, bool roll, bool do_lmk, bool weaken) {
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

// one iteration: k_sweep + k_beliefs; ev: the instantiations that carry the metric (a.ev filled in by the caller)
// weaken_after: WEAKEN_PRIORS follows this iteration with nothing reading the beliefs in between — its belief update takes the
// weakened priors straight away (WeakenPriorVertex rides in k_beliefs as in gbp_weaken_priors): the same beliefs, means and
// mean changes as {k_beliefs; k_beliefs(weaken)} leave, in one launch
void enqueue_iteration(gbp_ctx* c, const SweepArgs& a, bool ev, bool weaken_after) {
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
void enqueue_cam_partials(gbp_ctx* c, float* dst, hipStream_t s) {
  BeliefArgs b = belief_args(c);
  b.cam_local = dst; b.partial_only = 1;
  launch_beliefs(b, true, false, s ? s : c->stream);
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

// GBP_PROG x n on the two-kernel path: replay of a captured hipGraph of `graph_unroll` iterations, remainder launched directly.
// (ev: the iterations carry the metric — a.ev set, see eval_each_ride; their launches depend on the iteration only through a
// counter in device memory, so they replay from a graph of their own)
int iterate_plain(gbp_ctx* c, const SweepArgs& a, int n, bool ev) {
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

// Passes i0 .. i0 + n - 1 of the reference's loop WITHOUT the metric on the two-kernel path of a single-GPU ctx, the weakening in
// front of pass i0 (if any) already done by the caller: a weakening in front of a later pass rides in the belief update of the
// iteration before it (enqueue_iteration: weaken_after), the runs between them replay from the hipGraph.
int iterate_weaken_plain(gbp_ctx* c, const SweepArgs& a, int n, unsigned i0, unsigned steps2) {
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

// GBP_PROG x n (ba.cpp:895-905) on one GPU: inside the persistent kernel (small graphs), else hipGraph replay / direct launches.
int iterate(gbp_ctx* c, int n) {
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
int weaken_priors(gbp_ctx* c) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_weaken_priors: upload first");
  if (int rc = settle(c)) return rc;
  return refresh_beliefs_from_partials(c, false, true, /*weaken=*/true);      // ONE launch: the prior owners scale on their way into the sums
}

}  // namespace api
}  // namespace gbp

GBP_EXPORT(gbp_iterate, c, (gbp_ctx* c, int n), (c, n)) { return iterate(c, n); }
GBP_EXPORT(gbp_weaken_priors, c, (gbp_ctx* c), (c)) { return weaken_priors(c); }

GBP_EXPORT(gbp_refresh_begin, c, (gbp_ctx* c), (c)) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  float4* dst = exch(c) ? static_cast<float4*>(c->send_dev) : P<float4>(c->local);
  if (!dst) return fail(c, GBP_ERR_STATE, "exchange buffers not set");
  enqueue_cam_partials(c, reinterpret_cast<float*>(dst));
  HIPCHK(c, hipGetLastError());
  return GBP_OK;
}

GBP_EXPORT(gbp_refresh_end, c, (gbp_ctx* c), (c)) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  const int rc = refresh_beliefs_from_partials(c, false);
  if (rc == GBP_OK) c->beliefs_valid = true;
  return rc;
}

GBP_EXPORT(gbp_linearise_factors, c, (gbp_ctx* c), (c)) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  launch_linearise(sweep_args(c), c->n_tiles, c->stream);
  HIPCHK(c, hipGetLastError());
  return GBP_OK;
}

// LINEARISE_PROG (ba.cpp:890-893): prog_ub, then RelineariseFactorVertex on every factor.
GBP_EXPORT(gbp_linearise, c, (gbp_ctx* c), (c)) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_linearise: upload first");
  if (c->world > 1 && !c->comm)
    return fail(c, GBP_ERR_STATE, "sharded ctx without a communicator: gbp_comm_init first, or use refresh_begin / exchange / refresh_end / linearise_factors");
  int rc = gbp_refresh_begin(c);
  if (rc == GBP_OK && c->comm) rc = exchange_now(c);
  if (rc == GBP_OK) rc = gbp_refresh_end(c);
  if (rc == GBP_OK) rc = gbp_linearise_factors(c);
  return rc;
}

GBP_EXPORT(gbp_iterate_begin, c, (gbp_ctx* c), (c)) {
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
GBP_EXPORT(gbp_iterate_local, c, (gbp_ctx* c), (c)) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  BeliefArgs b = belief_args(c);
  b.roll = 1;
  launch_beliefs(b, false, true, c->stream);
  HIPCHK(c, hipGetLastError());
  c->lmk_half_done = true;
  return GBP_OK;
}

GBP_EXPORT(gbp_iterate_end, c, (gbp_ctx* c), (c)) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "upload first");
  if (int rc = settle(c)) return rc;
  const int rc = refresh_beliefs_from_partials(c, true, !c->lmk_half_done);
  c->lmk_half_done = false;
  if (rc == GBP_OK) c->beliefs_valid = true;
  return rc;
}

// One-off costs of the multi-iteration path, paid on request instead of inside the first gbp_iterate(n >= graph_unroll):
// graph capture + instantiation + upload of the executable graph.  Executes no iteration.
GBP_EXPORT(gbp_prepare, c, (gbp_ctx* c), (c)) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_prepare: upload first");
  if (int rc = settle(c)) return rc;
  if (c->comm || c->world > 1) return GBP_OK;               // sharded iterations run from direct launches by default
  if (c->persist_ok) return GBP_OK;                         // multi-iteration bursts run inside k_persist: nothing to capture
  if (ensure_graph(c, sweep_args(c))) (void)hipGraphUpload(c->graph_exec, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GBP_OK;
}
