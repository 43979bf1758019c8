// gbp_api_ctx.cpp — life cycle of a ctx and the programs that move host streams (include/gbp_mi355x.h):
//   gbp_create / gbp_destroy   graph build + Engine ctor/load        reference ba/ba.cpp:659-937
//   gbp_upload                 WRITE_PROG                            ba.cpp:868-886
//   gbp_read                   READ_PROG                             ba.cpp:908-916
//   gbp_read_priors            READ_PRIORS                           slam.cpp:913-917
//   gbp_new_keyframe           NEW_KEYFRAME                          slam.cpp:919-928
//   gbp_sync, gbp_set_stream, gbp_timing, gbp_set_profiling
// gbp_create sorts the factors into device order (gbp_layout.cpp, pure host code) and only then touches the GPU.
#include "gbp_ctx.hpp"
#include "gbp_threads.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace gbp;
using namespace gbp::api;

namespace gbp {
namespace api {

LayoutOptions g_layout_options;
int g_force_sweep_policy = -1;
int g_force_seg_skip = -1;

std::string& create_error() {
  thread_local std::string g_create_error;
  return g_create_error;
}

int fail(gbp_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else create_error() = msg;
  return code;
}

// Zero-filled device memory owned by the ctx.  The fill runs on the ctx's stream — the stream every later use of the buffer is
// queued on — so no device-wide synchronisation is needed behind an allocation (a fill on the NULL stream is not ordered against a
// non-blocking stream and may still be running when hipMemset returns; blocking copies into a fresh buffer wait for the stream first).
int dev_alloc(gbp_ctx* c, DevBuf& b, size_t bytes) {
  b.bytes = bytes < 256 ? 256 : (bytes + 15) / 16 * 16;   // >= one camera / landmark record (pad lanes of an empty shard read index 0); whole float4s (H2D)
  HIPCHK(c, hipMalloc(&b.p, b.bytes));
  c->all.push_back(&b);
  c->dev_bytes += b.bytes;
  HIPCHK(c, hipMemsetAsync(b.p, 0, b.bytes, c->stream));
  return GBP_OK;
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

int H2D::begin(gbp_ctx* ctx, size_t total_bytes, int pieces) {
  c = ctx; used = 0; segs = CopySegs{};
  const size_t need = total_bytes + 16 * (size_t)pieces;
  direct = need > kStageMax;
  if (direct) return GBP_OK;
  if (c->stage_cap < need) {
    if (c->stage_host) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->stage_host); c->stage_host = nullptr; c->stage_cap = 0; }
    const size_t cap = std::max<size_t>(std::max(need, c->stage_hint <= kStageMax ? c->stage_hint : 0), (size_t)1 << 20);
    HIPCHK(c, hipHostMalloc(&c->stage_host, cap, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer(&c->stage_dev, c->stage_host, 0));
    c->stage_cap = cap;
  }
  return GBP_OK;
}
int H2D::flush() {
  if (segs.n == 0) return GBP_OK;
  launch_copy_segments(segs, nullptr, c->stream);
  HIPCHK(c, hipGetLastError());
  segs.n = 0;
  return GBP_OK;
}
int H2D::put(void* dst_dev, const void* src, size_t bytes) {
  if (bytes == 0) return GBP_OK;
  if (direct) { HIPCHK(c, hipMemcpy(dst_dev, src, bytes, hipMemcpyHostToDevice)); return GBP_OK; }
  const size_t padded = (bytes + 15) / 16 * 16;
  if (used + padded > c->stage_cap) return fail(c, GBP_ERR_INVALID, "H2D: staging buffer too small (internal)");
  char* h = static_cast<char*>(c->stage_host) + used;
  std::memcpy(h, src, bytes);
  if (padded > bytes) std::memset(h + bytes, 0, padded - bytes);
  if (segs.n == kMaxCopySegs)
    if (int rc = flush()) return rc;
  segs.src[segs.n] = static_cast<char*>(c->stage_dev) + used; segs.dst[segs.n] = dst_dev; segs.n4[segs.n] = padded / 16;
  segs.n += 1;
  used += padded;
  return GBP_OK;
}
// bytes into the staging buffer for a kernel of the caller's to read (no copy segment): *dev = their device address, valid until end()
int H2D::stage(const void* src, size_t bytes, const void** dev) {
  const size_t padded = (bytes + 15) / 16 * 16;
  if (direct || used + padded > c->stage_cap) return fail(c, GBP_ERR_INVALID, "H2D: staging buffer too small (internal)");
  std::memcpy(static_cast<char*>(c->stage_host) + used, src, bytes);
  *dev = static_cast<const char*>(c->stage_dev) + used;
  used += padded;
  return GBP_OK;
}
int H2D::end() {
  if (!direct)
    if (int rc = flush()) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));      // the staging buffer is free again, the data is on the device
  return GBP_OK;
}

int D2H::begin(gbp_ctx* ctx, size_t total_bytes, int pieces) {
  if (int rc = up.begin(ctx, total_bytes, pieces)) return rc;
  pending.clear();
  return GBP_OK;
}
int D2H::get(void* dst_host, const void* src_dev, size_t bytes) {
  gbp_ctx* c = up.c;
  if (bytes == 0) return GBP_OK;
  if (up.direct) { HIPCHK(c, hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost)); return GBP_OK; }
  const size_t padded = (bytes + 15) / 16 * 16;
  if (up.used + padded > c->stage_cap) return fail(c, GBP_ERR_INVALID, "D2H: staging buffer too small (internal)");
  if (up.segs.n == kMaxCopySegs) {
    launch_copy_segments(up.segs, nullptr, c->stream);
    HIPCHK(c, hipGetLastError());
    up.segs.n = 0;
  }
  up.segs.src[up.segs.n] = src_dev; up.segs.dst[up.segs.n] = static_cast<char*>(c->stage_dev) + up.used; up.segs.n4[up.segs.n] = padded / 16;
  up.segs.n += 1;
  pending.push_back(Out{dst_host, up.used, bytes});
  up.used += padded;
  return GBP_OK;
}
int D2H::end() {
  if (int rc = up.end()) return rc;      // launches what is queued, synchronises the stream
  if (!up.direct)
    for (const Out& o : pending) std::memcpy(o.dst, static_cast<const char*>(up.c->stage_host) + o.off, o.bytes);
  pending.clear();
  return GBP_OK;
}

int event_pair(gbp_ctx* c, hipEvent_t* a, hipEvent_t* b) {
  HIPCHK(c, hipEventCreate(a));
  if (hipError_t e_ = hipEventCreate(b); e_ != hipSuccess) {
    (void)hipEventDestroy(*a);
    *a = nullptr;
    return fail(c, GBP_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e_));
  }
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

}  // namespace api
}  // namespace gbp

namespace {

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

// what one gbp_upload stages: the compact per-factor streams (20 B per position), per_factor_mu's tensor, the packed priors, scalings and flags
size_t upload_stage_bytes(const gbp_ctx* c) {
  return (size_t)c->Ep * 20 + (c->hoist ? 0 : (size_t)c->Ep * kMuG * 16) + ((size_t)c->C * kCamRec + (size_t)c->L_loc * 16 + 2 * ((size_t)c->C + c->L_loc)) * 4;
}

}  // namespace

GBP_EXPORT_T(int, 0, gbp_abi_version, (void), ()) { return GBP_ABI_VERSION; }

GBP_EXPORT_VOID(gbp_default_params, (gbp_params* p), (p)) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->maxeta_damping = 0.4f; p->num_undamped_iters = 8; p->dmu_threshold = 3e-3f; p->min_linear_iters = 10;
  p->nstds = 2.5f; p->relin_mode = 0; p->graph_unroll = 0;
}

GBP_EXPORT_T(const char*, "", gbp_last_error, (const gbp_ctx* ctx), (ctx)) { return ctx ? ctx->err.c_str() : create_error().c_str(); }

GBP_EXPORT_VOID(gbp_destroy, (gbp_ctx* c), (c)) {
  if (!c) return;
  drop_graph(c);
  persist_forget(c);
  if (c->comm) { (void)hipStreamSynchronize(c->stream); if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream); delete c->comm; c->comm = nullptr; }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  for (auto* v : {&c->pending_sweep_ev, &c->pending_exch_ev})
    for (auto& pr : *v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (DevBuf* b : c->all) if (b->p) (void)hipFree(b->p);
  for (auto& v : {&c->spans, &c->span_pool})
    for (auto& sp : *v) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  if (c->stage_host) (void)hipHostFree(c->stage_host);
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


GBP_EXPORT(gbp_create, nullptr, (const gbp_problem* pr, const gbp_params* prm, const gbp_shard* sh, gbp_ctx** out), (pr, prm, sh, out)) {
  if (!pr || !out || !pr->cam_id || !pr->lmk_id || pr->n_cams == 0 || pr->n_lmks == 0 || pr->n_edges == 0)
    return fail(nullptr, GBP_ERR_INVALID, "gbp_create: null or empty problem");
  // what the call spends where (left in gbp_last_error(ctx) as an `info:` line: the CLIs' --profile report quotes it)
  using clk = std::chrono::steady_clock;
  const auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  clk::time_point t_phase = clk::now();
  // GBP_CREATE_TRACE=1: one line per step on stderr (what the first stream, the first allocation, the first fill, the first copy of a
  // process cost: profiles/r06_configs.md)
  const bool trace = std::getenv("GBP_CREATE_TRACE") != nullptr;
  clk::time_point t_step = clk::now();
  const auto step = [&](const char* what) {
    if (trace) std::fprintf(stderr, "gbp_create: %-28s %8.3f ms\n", what, ms_since(t_step));
    t_step = clk::now();
  };
  // ---- device order: pure host code (gbp_layout.cpp), built and validated before anything touches the GPU ----
  Layout lay;
  {
    gbp_params dflt;
    gbp_default_params(&dflt);
    std::string lerr;
    if (int lrc = layout_build(pr, (prm ? prm : &dflt)->tile_order, sh, g_layout_options, lay, lerr)) return fail(nullptr, lrc, lerr);
  }
  const double layout_ms = ms_since(t_phase);
  t_phase = clk::now();
  step("device order");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, GBP_ERR_NO_DEVICE, "gbp_create: no HIP device (the product has no CPU fallback)");
  const double runtime_ms = ms_since(t_phase);      // (the HIP runtime comes up here unless the caller has touched it before)
  t_phase = clk::now();
  step("hipGetDeviceCount");
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


  int rc = GBP_OK;
  auto CK = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == GBP_OK) { create_error() = std::string(what) + ": " + hipGetErrorString(e); rc = GBP_ERR_HIP; }
  };
  CK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreate");
  if (rc != GBP_OK) return rc;
  c->stream = c->own_stream;
  step("hipStreamCreate");
  // ---- device allocations (zero-filled on the ctx's stream) ----
  auto A = [&](DevBuf& b, size_t bytes) { if (rc == GBP_OK) rc = dev_alloc(c, b, bytes); };
  const size_t Ep = c->Ep;
  // The index arrays of the device order live in ONE allocation and go up in ONE copy (H2D, gbp_ctx.hpp)
  struct Piece { DevBuf* b; const void* src; size_t bytes; size_t off; };
  Piece pieces[] = {{&c->row_cam, y.row_cam.data(), y.row_cam.size() * 4, 0}, {&c->lmk_idx, y.pos_lmk_loc.data(), Ep * 4, 0},
                    {&c->d_lmk_fpos, y.lmk_fpos.data(), (size_t)c->E_loc * 4, 0}, {&c->d_lmk_ix, y.lmk_ix.data(), (size_t)c->L_loc * 64, 0},
                    {&c->d_cam_row_ptr, y.cam_row_ptr.data(), (size_t)(C + 1) * 4, 0}, {&c->d_lmk_ptr, y.lmk_ptr.data(), (size_t)(c->L_loc + 1) * 4, 0},
                    {&c->d_row_slot, y.row_slot.data(), y.row_slot.size() * 4, 0}, {&c->dK, c->K, 9 * 4, 0},
                    {&c->tile_perm, y.tile_perm.data(), y.tile_perm.size() * 4, 0}, {&c->seg_live, nullptr, 0, 0}};
  // Which 64-byte segments (four positions) of a tile hold a factor at all?  A camera is padded to whole rows of 16: the tail of its
  // last row is empty — on a graph of many small cameras (BASELINE config 5: ~156 factors per camera and rank) the all-pad segments
  // are 3.7 % of all positions, which the sweep then neither streams in nor out (k_sweep<..., SEG>).  Used where it is worth a kernel
  // of its own: at least 1 % of the positions, a graph that runs on the two-kernel path with the default cache policy.
  std::vector<uint32_t> seg_live(y.n_tiles, 0u);
  size_t dead = 0;
  for (size_t t = 0; t < y.n_tiles; ++t) {
    uint32_t m = 0;
    for (uint32_t sgi = 0; sgi < 16; ++sgi)
      for (uint32_t k = 0; k < 4; ++k)
        if (y.pos_edge[t * 64 + sgi * 4 + k] != kNoEdge) { m |= 1u << sgi; break; }
    seg_live[t] = m;
    dead += 16u - (uint32_t)__builtin_popcount(m);
  }
  const char* seg_env = std::getenv("GBP_SEG_SKIP");      // A/B measurements (bench.py, the CLIs): 0 = never, 1 = always; results are identical
  const int seg_force = g_force_seg_skip >= 0 ? g_force_seg_skip : (seg_env ? (seg_env[0] == '1') : -1);
  const bool use_seg = c->hoist && c->sweep_policy == 0u && (seg_force >= 0 ? seg_force == 1 : (dead * 4 * 100 >= (size_t)c->Ep && y.n_tiles >= 2048));
  if (use_seg) { pieces[9].src = seg_live.data(); pieces[9].bytes = seg_live.size() * 4; }
  size_t idx_bytes = 0;
  for (Piece& pc : pieces) { pc.off = idx_bytes; idx_bytes += (std::max<size_t>(pc.bytes, 256) + 255) / 256 * 256; }      // (>= 256 B each: pad lanes of an empty shard read index 0)
  A(c->idx_arena, idx_bytes);
  step("first hipMalloc + fill");
  for (Piece& pc : pieces) { pc.b->p = rc == GBP_OK ? static_cast<char*>(c->idx_arena.p) + pc.off : nullptr; pc.b->bytes = pc.bytes; }      // views: freed with the arena
  A(c->fac, Ep * kFacG * 16); A(c->cmsg, Ep * kCmsgG * 16);
  A(c->mu, c->hoist ? 0 : Ep * kMuG * 16);   // literal mu/oldmu tensor: only with per_factor_mu
  A(c->lmsg, Ep * 64);
  A(c->camb, (size_t)C * kCamRec * 4); A(c->camp, (size_t)C * kCamRec * 4); A(c->local, (size_t)C * kCamRec * 4);
  A(c->lmkb, (size_t)c->L_loc * 64); A(c->lmkp, (size_t)c->L_loc * 64);
  A(c->rowp, (Ep / kRow) * kCamRec * 4);
  A(c->cwf, (size_t)C * 4); A(c->lwf, (size_t)c->L_loc * 4); A(c->cscale, (size_t)C * 4); A(c->lscale, (size_t)c->L_loc * 4);
  A(c->cam_mu, (size_t)C * 6 * 4 * 2); A(c->lmk_mu, (size_t)c->L_loc * 3 * 4 * 2);   // metric means; k_persist alternates between the two halves
  A(c->evalp, sizeof(DeviceEval) * 16); A(c->health, 32);
  A(c->hmu_c, (size_t)C * 4 * 16); A(c->hmu_l, (size_t)c->L_loc * 2 * 16); A(c->clin, (size_t)C * 5 * 16);
  A(c->st_a, Ep * 4); A(c->st_b, Ep * 4);
  if (rc != GBP_OK) { create_error() = c->err; return rc; }
  step("the other allocations");
  CK(hipEventCreate(&c->ev0), "hipEventCreate"); CK(hipEventCreate(&c->ev1), "hipEventCreate");
  CK(hipEventCreate(&c->ev2), "hipEventCreate"); CK(hipEventCreate(&c->ev3), "hipEventCreate");
  CK(hipStreamSynchronize(c->stream), "hipStreamSynchronize");      // the fills have landed: blocking copies into the fresh buffers follow
  const double alloc_ms = ms_since(t_phase);
  t_phase = clk::now();
  step("events + stream sync");
  {
    std::vector<char> stage(idx_bytes, 0);
    for (const Piece& pc : pieces)
      if (pc.bytes) std::memcpy(stage.data() + pc.off, pc.src, pc.bytes);
    // (pinned once, large enough for the gbp_upload that follows: pinning costs ~0.5 ms per MB, a second buffer would cost that again)
    c->stage_hint = upload_stage_bytes(c) + 16 * 9;
    H2D up;
    if (rc == GBP_OK) rc = up.begin(c, idx_bytes, 1);
    if (rc == GBP_OK) rc = up.put(c->idx_arena.p, stage.data(), idx_bytes);
    if (rc == GBP_OK) rc = up.end();
    if (rc != GBP_OK) create_error() = c->err;
  }
  step("the copy of the device order");
  // the XCD-aware execution order of the sweep: wave slot -> tile (gbp_layout.cpp); read once per wave with a scalar load
  c->use_tile_perm = rc == GBP_OK && !y.tile_perm.empty();
  c->use_seg_live = rc == GBP_OK && use_seg;
  if (rc != GBP_OK) return rc;
  const double upload_ms = ms_since(t_phase);
  t_phase = clk::now();
  // ---- persistent iteration kernel: only where every workgroup of the graph is resident at once (gbp_api_persist.cpp) ----
  rc = persist_setup(c, prm, sh != nullptr);
  if (rc != GBP_OK) { if (create_error().empty()) create_error() = c->err; return rc; }
  CK(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  if (rc != GBP_OK) return rc;
  step("persistent kernel set-up");
  {
    char info[320];
    std::snprintf(info, sizeof(info), "info: gbp_create: device order %.3f ms, runtime %.3f ms, allocation %.3f ms, layout upload %.3f ms, "
                                      "persistent kernel %.3f ms (probe %.3f ms)",
                  layout_ms, runtime_ms, alloc_ms, upload_ms, ms_since(t_phase), c->probe_ms);
    c->err = c->err.empty() ? std::string(info) : c->err + " | " + info;      // (a probe that failed has left its reason in front)
  }
  owner.p = nullptr;
  *out = c;
  return GBP_OK;
}

GBP_EXPORT(gbp_set_stream, c, (gbp_ctx* c, void* s), (c, s)) {
  if (!c) return GBP_ERR_INVALID;
  if (int rc = settle(c)) return rc;
  drop_graph(c);
  c->stream = s ? static_cast<hipStream_t>(s) : c->own_stream;
  return GBP_OK;
}

GBP_EXPORT(gbp_set_exchange_buffers, c, (gbp_ctx* c, void* send_dev, void* recv_dev), (c, send_dev, recv_dev)) {
  if (!c) return GBP_ERR_INVALID;
  c->send_dev = send_dev; c->recv_dev = recv_dev;
  return GBP_OK;
}

GBP_EXPORT(gbp_sync, c, (gbp_ctx* c), (c)) {
  if (!c) return GBP_ERR_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return persist_check(c, 0);
}

// WRITE_PROG (ba.cpp:868-886).  Also zeroes every tensor the reference leaves uninitialised
// (messages, factor potentials, beliefs: ba.cpp:668-687,759-775).
GBP_EXPORT(gbp_upload, c, (gbp_ctx* c, const gbp_state_in* in), (c, in)) {
  if (!c || !in) return GBP_ERR_INVALID;
  if (!in->cam_priors_eta || !in->cam_priors_lambda || !in->lmk_priors_eta || !in->lmk_priors_lambda ||
      !in->measurements || !in->meas_variances || !in->active_flag)
    return fail(c, GBP_ERR_INVALID, "gbp_upload: priors, measurements, meas_variances and active_flag are required");
  if (in->mu && in->oldmu && in->mu != in->oldmu && std::memcmp(in->mu, in->oldmu, (size_t)c->E * 9 * 4) != 0)
    return fail(c, GBP_ERR_INVALID, "gbp_upload: mu != oldmu is not supported (the reference uploads zeros for both, ba.cpp:582-583)");
  if (c->hoist) {
    const float* om = in->oldmu ? in->oldmu : in->mu;
    if (om)
      for (size_t i = 0; i < (size_t)c->E * 9; ++i)
        if (om[i] != 0.f)
          return fail(c, GBP_ERR_INVALID, "gbp_upload: non-zero oldmu needs gbp_params.per_factor_mu = 1 (the reference uploads zeros, ba.cpp:582-583)");
  }
  const bool trace = std::getenv("GBP_HOST_TRACE") != nullptr;      // milliseconds of every step on stderr
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "gbp_upload: %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  };
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (int rc = persist_reset(c)) return rc;      // a fresh start for the persistent kernel too
  lap("argument checks, stream sync, persistent-kernel reset");
  const size_t Ep = c->Ep;
  // The per-factor streams in their compact form, in device order: {damping, count << 3 | flags, z0, z1} + the variance = 20 bytes per
  // position.  k_upload_scatter writes them into the records they belong to (the whole 64-byte LMSG record: zero messages + state; the
  // measurement slots of the FAC tile, the other 216 bytes per position zeroed on the device) — 288 bytes per position used to be built on
  // the host and cross PCIe: 0.11 s of a `bin/ba` run on a file of 10^6 factors (profiles/r06_configs.md section 2).
  struct St { float damping; int32_t packed; float z0, z1; };
  static_assert(sizeof(St) == 16, "k_upload_scatter reads one float4 per position");
  std::vector<St> st(Ep);
  std::vector<float> var(Ep), mu(c->hoist ? 0 : Ep * kMuG * 4, 0.f);
  c->active_host.assign(Ep, 0);
  const float* om = in->oldmu ? in->oldmu : in->mu;
  const unsigned T = gbp::host::host_threads(Ep, 1u << 17);      // (a gather by device position: every position is written by one thread)
  gbp::host::on_threads(T, [&](unsigned t) {
    for (size_t p = Ep * t / T, p1 = Ep * (t + 1) / T; p < p1; ++p) {
      const uint32_t e = c->lay.pos_edge[p];
      St o{0.f, (int32_t)kFlagPad, 0.f, 0.f};
      float v = 0.f;
      if (e != ~0u) {
        const bool on = in->active_flag[e] == 1;
        c->active_host[p] = on;
        const int32_t count = in->damping_count ? in->damping_count[e] : 0;
        o.damping = in->damping ? in->damping[e] : 0.f;
        o.packed = (int32_t)(((uint32_t)count << 3) | (on ? kFlagActive : 0u));
        o.z0 = in->measurements[2 * (size_t)e];
        o.z1 = in->measurements[2 * (size_t)e + 1];
        v = in->meas_variances[e];
        if (om && !c->hoist) for (int i = 0; i < 9; ++i) mu[tile_off((uint32_t)p, kMuG, i)] = om[(size_t)e * 9 + i];
      }
      st[p] = o;
      var[p] = v;
    }
  });
  lap("gather by device position");
  HIPCHK(c, hipMemsetAsync(c->fac.p, 0, c->fac.bytes, c->stream));
  lap("FAC fill queued");
  H2D up;
  if (int rc = up.begin(c, upload_stage_bytes(c), 9)) return rc;
  lap("staging buffer ready");
  {
    // the compact streams: out of the pinned staging buffer (the kernel reads host memory), or — too large for it — through a device
    // buffer that lives for this call
    DevBuf tmp;
    const float4* st_dev = nullptr;
    const float* var_dev = nullptr;
    if (!up.direct) {
      const void *a = nullptr, *b = nullptr;
      if (int rc = up.stage(st.data(), Ep * 16, &a)) return rc;
      if (int rc = up.stage(var.data(), Ep * 4, &b)) return rc;
      st_dev = static_cast<const float4*>(a); var_dev = static_cast<const float*>(b);
      lap("compact streams copied into the staging buffer");
    } else {
      HIPCHK(c, hipMalloc(&tmp.p, Ep * 20));
      tmp.bytes = Ep * 20;
      const hipError_t e1 = hipMemcpy(tmp.p, st.data(), Ep * 16, hipMemcpyHostToDevice);
      const hipError_t e2 = e1 == hipSuccess ? hipMemcpy(static_cast<char*>(tmp.p) + Ep * 16, var.data(), Ep * 4, hipMemcpyHostToDevice) : e1;
      if (e2 != hipSuccess) { (void)hipFree(tmp.p); HIPCHK(c, e2); }
      st_dev = static_cast<const float4*>(tmp.p); var_dev = reinterpret_cast<const float*>(static_cast<char*>(tmp.p) + Ep * 16);
    }
    launch_upload_scatter(P<float4>(c->lmsg), P<float4>(c->fac), st_dev, var_dev, (uint32_t)Ep, c->stream);
    const hipError_t le = hipGetLastError();
    if (tmp.p) { (void)hipStreamSynchronize(c->stream); (void)hipFree(tmp.p); }
    HIPCHK(c, le);
  }
  lap("staging + k_upload_scatter queued");
  if (!c->hoist)
    if (int rc = up.put(c->mu.p, mu.data(), mu.size() * 4)) return rc;
  HIPCHK(c, hipMemsetAsync(c->cmsg.p, 0, c->cmsg.bytes, c->stream));
  HIPCHK(c, hipMemsetAsync(c->rowp.p, 0, c->rowp.bytes, c->stream));
  HIPCHK(c, hipMemsetAsync(c->local.p, 0, c->local.bytes, c->stream));
  HIPCHK(c, hipMemsetAsync(c->camb.p, 0, c->camb.bytes, c->stream));
  HIPCHK(c, hipMemsetAsync(c->lmkb.p, 0, c->lmkb.bytes, c->stream));
  HIPCHK(c, hipMemsetAsync(c->hmu_c.p, 0, c->hmu_c.bytes, c->stream));
  HIPCHK(c, hipMemsetAsync(c->hmu_l.p, 0, c->hmu_l.bytes, c->stream));
  HIPCHK(c, hipMemsetAsync(c->clin.p, 0, c->clin.bytes, c->stream));
  std::vector<float> rec;
  pack_cam(in->cam_priors_eta, in->cam_priors_lambda, c->C, rec);
  if (int rc = up.put(c->camp.p, rec.data(), rec.size() * 4)) return rc;
  pack_lmk(in->lmk_priors_eta, in->lmk_priors_lambda, c->lmk_begin, c->L_loc, rec);
  if (c->L_loc)
    if (int rc = up.put(c->lmkp.p, rec.data(), rec.size() * 4)) return rc;
  std::vector<float> zf(std::max(c->C, c->L), 0.f);
  std::vector<uint32_t> zu(std::max(c->C, c->L), 0u);
  if (int rc = up.put(c->cscale.p, in->cam_scaling ? in->cam_scaling : zf.data(), (size_t)c->C * 4)) return rc;
  if (int rc = up.put(c->cwf.p, in->cam_weaken_flag ? in->cam_weaken_flag : zu.data(), (size_t)c->C * 4)) return rc;
  if (c->L_loc) {
    if (int rc = up.put(c->lscale.p, in->lmk_scaling ? in->lmk_scaling + c->lmk_begin : zf.data(), (size_t)c->L_loc * 4)) return rc;
    if (int rc = up.put(c->lwf.p, in->lmk_weaken_flag ? in->lmk_weaken_flag + c->lmk_begin : zu.data(), (size_t)c->L_loc * 4)) return rc;
  }
  lap("fills, priors, scalings queued");
  if (int rc = up.end()) return rc;
  lap("everything on the device");
  if (exch(c) && c->recv_dev) HIPCHK(c, hipMemsetAsync(c->recv_dev, 0, (size_t)c->world * c->C * kCamRec * 4, c->stream));
  c->uploaded = true;
  c->beliefs_valid = false;
  return GBP_OK;
}

// READ_PROG (ba.cpp:908-916)
GBP_EXPORT(gbp_read, c, (gbp_ctx* c, gbp_state_out* o), (c, o)) {
  if (!c || !o) return GBP_ERR_INVALID;
  if (int rc = gbp_sync(c)) return rc;
  const bool want_cam = o->cam_beliefs_eta || o->cam_beliefs_lambda, want_lmk = (o->lmk_beliefs_eta || o->lmk_beliefs_lambda) && c->L_loc;
  const bool want_state = o->damping || o->damping_count || o->robust_flag;
  std::vector<float> rec_c(want_cam ? (size_t)c->C * kCamRec : 0), rec_l(want_lmk ? (size_t)c->L_loc * 16 : 0), damp(want_state ? c->Ep : 0);
  std::vector<int32_t> packed(want_state ? c->Ep : 0);
  if (want_state) {
    // per-factor scalars ride in the message records: a small kernel extracts them into two compact arrays
    launch_state_get(P<float4>(c->lmsg), P<float>(c->st_a), P<int>(c->st_b), c->Ep, c->stream);
    HIPCHK(c, hipGetLastError());
  }
  D2H down;
  if (int rc = down.begin(c, (rec_c.size() + rec_l.size() + damp.size() + packed.size()) * 4, 4)) return rc;
  if (want_state && down.up.direct) HIPCHK(c, hipStreamSynchronize(c->stream));      // (a large graph: the copies below are blocking hipMemcpy calls, which do not order against the ctx's stream)
  if (int rc = down.get(rec_c.data(), c->camb.p, rec_c.size() * 4)) return rc;
  if (int rc = down.get(rec_l.data(), c->lmkb.p, rec_l.size() * 4)) return rc;
  if (int rc = down.get(damp.data(), c->st_a.p, damp.size() * 4)) return rc;
  if (int rc = down.get(packed.data(), c->st_b.p, packed.size() * 4)) return rc;
  if (int rc = down.end()) return rc;
  if (want_cam) {
    const std::vector<float>& rec = rec_c;
    for (uint32_t k = 0; k < c->C; ++k) {
      if (o->cam_beliefs_eta) std::memcpy(o->cam_beliefs_eta + (size_t)k * 6, &rec[(size_t)k * kCamRec], 6 * 4);
      if (o->cam_beliefs_lambda) std::memcpy(o->cam_beliefs_lambda + (size_t)k * 36, &rec[(size_t)k * kCamRec + 8], 36 * 4);
    }
  }
  if (want_lmk) {
    const std::vector<float>& rec = rec_l;
    for (uint32_t l = 0; l < c->L_loc; ++l) {
      if (o->lmk_beliefs_eta) std::memcpy(o->lmk_beliefs_eta + (size_t)(c->lmk_begin + l) * 3, &rec[(size_t)l * 16], 3 * 4);
      if (o->lmk_beliefs_lambda) std::memcpy(o->lmk_beliefs_lambda + (size_t)(c->lmk_begin + l) * 9, &rec[(size_t)l * 16 + 4], 9 * 4);
    }
  }
  if (want_state) {
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
GBP_EXPORT(gbp_read_priors, c, (gbp_ctx* c, gbp_priors_out* o), (c, o)) {
  if (!c || !o) return GBP_ERR_INVALID;
  if (int rc = gbp_sync(c)) return rc;
  std::vector<float> rec((size_t)c->C * kCamRec), rec_l((size_t)c->L_loc * 16);
  D2H down;
  if (int rc = down.begin(c, (rec.size() + rec_l.size()) * 4, 2)) return rc;
  if (int rc = down.get(rec.data(), c->camp.p, rec.size() * 4)) return rc;
  if (int rc = down.get(rec_l.data(), c->lmkp.p, rec_l.size() * 4)) return rc;
  if (int rc = down.end()) return rc;
  for (uint32_t k = 0; k < c->C; ++k) {
    if (o->cam_priors_eta) std::memcpy(o->cam_priors_eta + (size_t)k * 6, &rec[(size_t)k * kCamRec], 6 * 4);
    if (o->cam_priors_lambda) std::memcpy(o->cam_priors_lambda + (size_t)k * 36, &rec[(size_t)k * kCamRec + 8], 36 * 4);
  }
  if (c->L_loc) {
    rec.swap(rec_l);
    for (uint32_t l = 0; l < c->L_loc; ++l) {
      if (o->lmk_priors_eta) std::memcpy(o->lmk_priors_eta + (size_t)(c->lmk_begin + l) * 3, &rec[(size_t)l * 16], 3 * 4);
      if (o->lmk_priors_lambda) std::memcpy(o->lmk_priors_lambda + (size_t)(c->lmk_begin + l) * 9, &rec[(size_t)l * 16 + 4], 9 * 4);
    }
  }
  return GBP_OK;
}

// NEW_KEYFRAME (slam.cpp:919-928): re-upload damping_count, priors, flags; then prog_ub.
GBP_EXPORT(gbp_new_keyframe, c, (gbp_ctx* c, const gbp_kf_update* u), (c, u)) {
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
    H2D up;
    if (int rc = up.begin(c, (size_t)c->Ep * 8, 2)) return rc;
    if (int rc = up.put(c->st_b.p, cnt.data(), (size_t)c->Ep * 4)) return rc;
    if (int rc = up.put(c->st_a.p, ctl.data(), (size_t)c->Ep * 4)) return rc;
    if (int rc = up.end()) return rc;
    launch_state_set(P<float4>(c->lmsg), P<int>(c->st_b), P<uint32_t>(c->st_a), c->Ep, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (u->active_flag)
      for (size_t p = 0; p < c->Ep; ++p)
        if (c->lay.pos_edge[p] != ~0u) c->active_host[p] = u->active_flag[c->lay.pos_edge[p]] == 1;
  }
  H2D up;
  if (int rc = up.begin(c, ((size_t)c->C + c->L_loc) * (kCamRec + 16 + 1) * 4, 4)) return rc;
  if (u->cam_priors_eta && u->cam_priors_lambda) {
    std::vector<float> rec;
    pack_cam(u->cam_priors_eta, u->cam_priors_lambda, c->C, rec);
    if (int rc = up.put(c->camp.p, rec.data(), rec.size() * 4)) return rc;
  }
  if (u->lmk_priors_eta && u->lmk_priors_lambda && c->L_loc) {
    std::vector<float> rec;
    pack_lmk(u->lmk_priors_eta, u->lmk_priors_lambda, c->lmk_begin, c->L_loc, rec);
    if (int rc = up.put(c->lmkp.p, rec.data(), rec.size() * 4)) return rc;
  }
  if (u->cam_weaken_flag)
    if (int rc = up.put(c->cwf.p, u->cam_weaken_flag, (size_t)c->C * 4)) return rc;
  if (u->lmk_weaken_flag && c->L_loc)
    if (int rc = up.put(c->lwf.p, u->lmk_weaken_flag + c->lmk_begin, (size_t)c->L_loc * 4)) return rc;
  if (int rc = up.end()) return rc;
  return refresh_beliefs_from_partials(c, false);
}

GBP_EXPORT(gbp_timing, c, (gbp_ctx* c, gbp_timing_out* t, int reset), (c, t, reset)) {
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
GBP_EXPORT(gbp_set_profiling, c, (gbp_ctx* c, int per_stage_events), (c, per_stage_events)) {
  if (!c) return GBP_ERR_INVALID;
  c->profile_stages = per_stage_events != 0;
  return GBP_OK;
}

GBP_EXPORT_T(int, 0, gbp_graph_state, (const gbp_ctx* c), (c)) {
  return !c ? 0 : (c->persist_ok ? 2 : ((c->graph_exec || c->graph_exec_ev) ? 1 : (c->graph_failed ? -1 : 0)));
}
