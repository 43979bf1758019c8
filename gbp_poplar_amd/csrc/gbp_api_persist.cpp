// gbp_api_persist.cpp — the launches of the persistent kernel (k_persist_flow: n iterations of gbp_iter_prog, reference
// ba/ba.cpp:895-905, in ONE launch on graphs whose workgroups are all resident at once).
//
// What lives here is everything that makes such a launch safe to use from the program list: the creation-time probe of the
// placement, the serialisation of launches across the ctxs / streams of a process, the snapshot taken in front of every launch, the
// log of launches whose completion has not been validated yet, and the recovery (restore + replay on the two-kernel path) when a
// wait inside a launch timed out.
#include "gbp_ctx.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <mutex>

using namespace gbp;
using namespace gbp::api;

namespace {
// Serialisation of k_persist launches across the ctxs / streams of a process (launch_persist_burst): a library-owned event
// per device is recorded behind every launch, the next launch from another ctx or stream waits for it.  No stream handle of
// another ctx is ever touched (it may have been destroyed by its owner); the two "last" words are compared, never used.
std::mutex g_persist_mu;
hipEvent_t g_persist_event[16] = {};
const void* g_persist_last_ctx[16] = {};
const void* g_persist_last_stream[16] = {};
constexpr size_t kPersistLogMax = 8;     // launches in flight without a validated completion
}  // namespace

namespace gbp {
namespace api {

void persist_forget(gbp_ctx* c) {
  if (!c->persist_eligible) return;
  // its launches have ended before its memory goes away; the "last launcher" words are only ever compared
  (void)hipStreamSynchronize(c->stream);
  std::lock_guard<std::mutex> lock(g_persist_mu);
  for (const void*& p : g_persist_last_ctx) if (p == c) p = &g_persist_mu;      // "someone else": the next launcher waits for the device's event
}

// gbp_upload: whatever the launches in flight did is overwritten by the upload; a ctx that left the persistent path after a
// recovered time-out gets it back
int persist_reset(gbp_ctx* c) {
  if (!c->pstatus_host) return GBP_OK;
  c->persist_log.clear();
  HIPCHK(c, hipMemsetAsync(c->psync.p, 0, kPersistSyncWords * sizeof(unsigned), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *static_cast<volatile unsigned*>(c->pstatus_host) = 0u;
  c->persist_epoch_base = 0;
  c->persist_ok = c->persist_eligible;
  return GBP_OK;
}

// The persistent kernel for this ctx, or nothing (not eligible / not co-resident: the ctx stays on the two-kernel path, with the
// reason in gbp_last_error where there is one).  Returns an error only when a HIP call failed.
int persist_setup(gbp_ctx* c, const gbp_params* prm, bool sharded) {
  int rc = GBP_OK;
  auto CK = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == GBP_OK) { create_error() = std::string(what) + ": " + hipGetErrorString(e); rc = GBP_ERR_HIP; }
  };
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
  if (mode >= 0 && !sharded && c->hoist && !c->use_tile_perm && c->lay.row_slot.empty() && nb <= (mode > 0 ? 1u << 30 : auto_limit)) {
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
        create_error() = c->err;
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
        const auto t_probe = std::chrono::steady_clock::now();
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
        c->probe_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_probe).count();
        if (!ok) c->err += "; iterations run on the two-kernel path";
        c->persist_ok = c->persist_eligible = ok;
        if (ok) CK(hipMemsetAsync(c->psync.p, 0, kPersistSyncWords * sizeof(unsigned), c->stream), "hipMemsetAsync");   // counter back to 0 after the probe
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
          create_error() = c->err;
        }
      }
      if (rc == GBP_OK && c->persist_ok) {
        // tagged shadows of the arrays that cross waves inside a launch (k_persist_flow): two halves each
        const size_t Ep = (size_t)c->n_tiles * 64, C_ = c->C, L_ = c->L_loc;
        const size_t n4[10] = {2 * Ep * 4, 2 * (Ep / 16) * kFlowRow4, 2 * C_ * kFlowCam4, 2 * C_ * 2, 2 * C_ * kFlowClin4, 2 * L_ * kFlowLmk4, 2 * L_,
                               2 * C_ * 4, 2 * L_, (size_t)kSeriesMax};      // (the last: [kSeriesMax][2] 64-bit health words = kSeriesMax float4)
        size_t total = 0;
        for (size_t n : n4) total += n;
#ifdef GBP_BUILD_TEST_HOOKS
        // gbp_debug_persist_verify: room for the redundant copy of every record + the mismatch counter (the product allocates neither)
        rc = dev_alloc(c, c->pflow, 2 * total * 16 + 64);
        c->flow_total4 = (uint32_t)total;
#else
        rc = dev_alloc(c, c->pflow, total * 16);
#endif
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
          create_error() = c->err;
        }
      }
    }
  }
  return rc;
}

int settle(gbp_ctx* c) {
  if (c->persist_log.empty()) return GBP_OK;
  // unvalidated k_persist launches and a caller who has begun capturing the stream: synchronising would invalidate their capture
  if (stream_is_capturing(c))
    return fail(c, GBP_ERR_STATE, "bursts of the persistent kernel are still in flight on this stream: call gbp_sync before beginning a stream capture");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return persist_check(c, 0);
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
int persist_check(gbp_ctx* c, unsigned upto) {
  if (!c->pstatus_host || c->persist_log.empty()) return GBP_OK;
  if (*static_cast<volatile unsigned*>(c->pstatus_host) != 0u) return persist_recover(c);
  if (upto == 0) c->persist_log.clear();
  else
    while (!c->persist_log.empty() && c->persist_log.front().seq <= upto) c->persist_log.erase(c->persist_log.begin());
  return GBP_OK;
}

bool stream_is_capturing(gbp_ctx* c) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(c->stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}

// May the next burst of this ctx run inside k_persist?  Makes room in the log of unvalidated launches first (which may
// find a time-out, recover, and take the ctx off the persistent path).  Not while the stream is being captured: the
// barrier targets are launch arguments computed by the host, a replayed graph would wait for arrivals long past.
int persist_ready(gbp_ctx* c, bool* yes) {
  *yes = false;
  if (!c->persist_ok || c->comm || c->world != 1 || c->profile_stages) return GBP_OK;
  if (stream_is_capturing(c)) return GBP_OK;
  if (c->persist_log.size() >= kPersistLogMax)
    if (int rc = settle(c)) return rc;
  *yes = c->persist_ok;
  return GBP_OK;
}

// n iterations inside ONE k_persist launch (+ the metric phases when `ev` is given).  The barrier counter keeps counting
// across the launches of a ctx (no memset per launch): the host tracks how many arrivals it has seen.
int launch_persist_burst(gbp_ctx* c, const SweepArgs& a, int n, const PersistEval* ev, int mode, int area,
                         uint32_t w_first, uint32_t w_steps2) {
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

}  // namespace api
}  // namespace gbp
