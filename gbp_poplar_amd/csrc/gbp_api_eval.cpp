// gbp_api_eval.cpp — the metric on the device and the loops that carry it.
//   gbp_eval                      eval_reprojection_error (reference ba/util.cpp:74-144) + the counters of ba.cpp:1011-1020
//   gbp_ba_loop                   the BODY of the reference's iteration loop (ba.cpp:1001-1028, slam.cpp:1048-1103): prior weakening,
//                                 GBP_PROG, the metric — one launch of the persistent kernel per burst on small graphs, the metric
//                                 riding in the sweeps (k_sweep<EV> / k_beliefs_ev) elsewhere
//   include/gbp_mi355x_compat.h   gbp_eval_begin / _end, gbp_iterate_eval, gbp_iterate_eval_each: earlier forms of the same loop
#include "gbp_ctx.hpp"

#include <algorithm>

using namespace gbp;
using namespace gbp::api;

namespace gbp {
namespace api {

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
int eval_enqueue(gbp_ctx* c, int area) {
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

int eval_begin(gbp_ctx* c) {
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

int eval_end(gbp_ctx* c, gbp_eval_out* o) {
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

int eval(gbp_ctx* c, gbp_eval_out* o) {
  if (!c || !o || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_eval: upload first");
  if (c->eval_pending) return fail(c, GBP_ERR_STATE, "gbp_eval: finish the evaluations in flight (gbp_eval_end) first");
  if (int rc = eval_begin(c)) return rc;
  return eval_end(c, o);
}

}  // namespace api
}  // namespace gbp

namespace {

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

// gbp_iterate(n) followed by gbp_eval_begin() in one call.  On a graph that runs in k_persist the metric rides in the same
// launch (two more phases after the last belief update: what k_means and k_eval compute, bit for bit) — the reference's
// default loop prints the metric after EVERY iteration (ba.cpp:1009-1028), which otherwise costs four launches per iteration.
static int iterate_eval(gbp_ctx* c, int n) {
  if (!c || !c->uploaded) return fail(c, GBP_ERR_STATE, "gbp_iterate_eval: upload first");
  if (n <= 0) return eval_begin(c);
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
      if (int rc = iterate(c, n - head)) return rc;
      return eval_begin(c);
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
  if (int rc = iterate(c, n)) return rc;
  return eval_begin(c);
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
  // (dev_alloc zero-fills on the ctx's stream, in front of the first burst that uses the ring: a fill on the NULL stream once raced
  // the first burst's records — a first metric over 3 062 of 200 000 factors — and a device-wide wait stalls every other stream of the process)
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
    if (int rc = iterate(c, 1)) return rc;
    if (int rc = eval_begin(c)) return rc;
    if (c->eval_pending == 2) { if (int rc = eval_end(c, out + collected)) return rc; ++collected; }
  }
  while (collected < n) { if (int rc = eval_end(c, out + collected)) return rc; ++collected; }
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
  if (steps >= (1u << 30)) return fail(c, GBP_ERR_INVALID, "gbp_ba_loop: steps < 2^30 (the loop's test is iter < 2 * steps)");
  const auto weak = [&](unsigned i) { return ((i + 1u) % 2u == 0u) && i < 2u * steps; };
  if (!out) {
    // without the metric (out == NULL): not blocking, like gbp_iterate.  The weakening in front of the first pass is a launch of its
    // own; every later one rides in the launch of the persistent kernel, or — two-kernel path, single-GPU ctx — in the belief update
    // of the iteration before it.  A sharded ctx takes the calls one by one.
    int done = 0;
    while (done < n) {
      const unsigned i0 = iter0 + (unsigned)done;
      if (weak(i0))
        if (int rc = weaken_priors(c)) return rc;
      if (c->comm || c->world != 1 || c->profile_stages || stream_is_capturing(c)) {      // (per-stage timing and a caller's capture keep their own paths)
        int m = 1;
        while (done + m < n && !weak(iter0 + (unsigned)(done + m))) ++m;
        if (int rc = iterate(c, m)) return rc;
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
      if (int rc = weaken_priors(c)) return rc;
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

}  // namespace

GBP_EXPORT(gbp_eval, c, (gbp_ctx* c, gbp_eval_out* o), (c, o)) { return eval(c, o); }
GBP_EXPORT(gbp_eval_begin, c, (gbp_ctx* c), (c)) { return eval_begin(c); }
GBP_EXPORT(gbp_eval_end, c, (gbp_ctx* c, gbp_eval_out* o), (c, o)) { return eval_end(c, o); }
GBP_EXPORT(gbp_iterate_eval, c, (gbp_ctx* c, int n), (c, n)) { return iterate_eval(c, n); }
GBP_EXPORT(gbp_iterate_eval_each, c, (gbp_ctx* c, int n, gbp_eval_out* out), (c, n, out)) { return iterate_eval_each_impl(c, n, out); }
GBP_EXPORT(gbp_ba_loop, c, (gbp_ctx* c, int n, unsigned iter0, unsigned steps, gbp_eval_out* out), (c, n, iter0, steps, out)) {
  return ba_loop_impl(c, n, iter0, steps, out);
}
