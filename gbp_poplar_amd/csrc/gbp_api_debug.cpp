// gbp_api_debug.cpp — the test hooks of include/gbp_mi355x_debug.h (libgbp_mi355x_test.so / _exp.so only; the product library
// compiles this file to nothing): internal tensors in the reference's layouts, the device math on caller vectors, the device order
// without a device, timing experiments.
#ifdef GBP_BUILD_TEST_HOOKS
#include "gbp_ctx.hpp"

#include <algorithm>

using namespace gbp;
using namespace gbp::api;

namespace {
int download_lmsg(gbp_ctx* c, std::vector<float>& rec) {
  rec.resize((size_t)c->Ep * 16);
  HIPCHK(c, hipMemcpy(rec.data(), c->lmsg.p, rec.size() * 4, hipMemcpyDeviceToHost));
  return GBP_OK;
}
}  // namespace

// Raw internal state in the reference's tensor layouts, for stage-level parity tests.
//   what = 0: factor_potentials_eta [9E] + factor_potentials_lambda [81E] = [cc36|cl18|lc18|ll9] (ba.cpp:93-96)
//   what = 1: factor->camera messages as stored: eta [6E] + Lambda [36E] (lower triangle; upper = 0)
//   what = 2: factor->landmark messages: eta [3E] + Lambda [9E]
//   what = 3: mu [9E] + dmu [E]
// Entries of factors outside the local shard are left untouched.
GBP_EXPORT(gbp_debug_get, c, (gbp_ctx* c, int what, float* a, float* b), (c, what, a, b)) {
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
GBP_EXPORT(gbp_debug_time_sweep, c, (gbp_ctx* c, int ablation, int reps, double* avg_us), (c, ablation, reps, avg_us)) {
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
GBP_EXPORT_VOID(gbp_debug_layout_default_options, (gbp_layout_options* o), (o)) {
  if (!o) return;
  const LayoutOptions d;
  o->row_placement = d.row_placement; o->row_window = d.row_window; o->row_place_max_deg = d.row_place_max_deg;
  o->row_key_lane = d.row_key_lane; o->classes = d.classes; o->tile_window = d.tile_window; o->tile_min_tiles = d.tile_min_tiles;
  o->tile_identity = d.tile_identity; o->row_sort_in_class = d.row_sort_in_class;
}
GBP_EXPORT(gbp_debug_layout_options, nullptr, (const gbp_layout_options* o), (o)) { g_layout_options = to_options(o); return GBP_OK; }
GBP_EXPORT(gbp_debug_force_sweep_policy, nullptr, (int policy), (policy)) { g_force_sweep_policy = policy; return GBP_OK; }
GBP_EXPORT(gbp_debug_force_seg_skip, nullptr, (int mode), (mode)) { g_force_seg_skip = mode; return GBP_OK; }
GBP_EXPORT(gbp_debug_persist_roles, nullptr, (uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, int with_metric, uint32_t* dims, uint32_t* role, uint32_t cap),
           (n_tiles, n_cams, n_lmks, with_metric, dims, role, cap)) {
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
GBP_EXPORT(gbp_debug_persist_flow, c, (gbp_ctx* c, int on), (c, on)) {
  if (!c) return GBP_ERR_INVALID;
  c->persist_flow = on != 0;
  return GBP_OK;
}
// Redundant records in the persistent kernel (k_persist_flow<EV, VER = true>): on != 0 — every tagged record of the following launches
// is published twice (the second copy with its payload complemented) and a consumer accepts a record only when both copies carry the
// same tag; a payload mismatch between the two is what a torn or stale 16-byte record would look like.  *mismatches (may be NULL)
// receives the count since the last call and resets it.  A ctx that does not run in the persistent kernel: GBP_ERR_STATE.
GBP_EXPORT(gbp_debug_persist_verify, c, (gbp_ctx* c, int on, uint64_t* mismatches), (c, on, mismatches)) {
  if (!c) return GBP_ERR_INVALID;
  if (!c->flow.lmsg || !c->flow_total4) return fail(c, GBP_ERR_STATE, "gbp_debug_persist_verify: this ctx does not run in the persistent kernel");
  if (int rc = gbp_sync(c)) return rc;
  unsigned long long* counter = reinterpret_cast<unsigned long long*>(static_cast<float4*>(c->pflow.p) + 2 * (size_t)c->flow_total4);
  if (mismatches) {
    unsigned long long v = 0;
    HIPCHK(c, hipMemcpy(&v, counter, 8, hipMemcpyDeviceToHost));
    *mismatches = v;
  }
  HIPCHK(c, hipMemsetAsync(counter, 0, 8, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->flow.mirror4 = on ? c->flow_total4 : 0u;
  c->flow.verify_errors = counter;
  return GBP_OK;
}

// The detector under the tagged records (hooks/gbp_flow_torture.hip): `rounds` rounds of K 16-byte records per lane between partner
// workgroups bid and bid ^ mask of a grid of `blocks` (a power of two, 16 .. 256, >= 2 * mask; K = 4 or 16); out[4] = records seen torn
// (tag of the awaited round, payload words of the slot's previous content), otherwise corrupt, waits that timed out, records checked.
// inject != 0: the control — some records are stored tag first, payload later: the detector must report them.
GBP_EXPORT(gbp_debug_flow_torture, nullptr, (int blocks, int K, int rounds, unsigned mask, int inject, uint64_t* out), (blocks, K, rounds, mask, inject, out)) {
  if (!out || blocks < 16 || blocks > 256 || (blocks & (blocks - 1)) || (K != 4 && K != 16) || rounds < 1 || mask == 0 || 2 * mask > (unsigned)blocks)
    return fail(nullptr, GBP_ERR_INVALID, "gbp_debug_flow_torture: blocks a power of two in [16, 256], K 4 or 16, rounds >= 1, 0 < 2 * mask <= blocks");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, GBP_ERR_NO_DEVICE, "gbp_debug_flow_torture: no HIP device");
  const size_t n4 = (size_t)2 * blocks * 4 * 64 * K;
  float4* buf = nullptr;
  unsigned long long* d_out = nullptr;
  hipStream_t s = nullptr;
  auto done = [&](int rc, const char* what, hipError_t e) {
    if (buf) (void)hipFree(buf);
    if (d_out) (void)hipFree(d_out);
    if (s) (void)hipStreamDestroy(s);
    return rc == GBP_OK ? rc : fail(nullptr, rc, std::string(what) + ": " + hipGetErrorString(e));
  };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) return done(GBP_ERR_HIP, "hipStreamCreate", e);
  if ((e = hipMalloc(&buf, n4 * 16)) != hipSuccess) return done(GBP_ERR_HIP, "hipMalloc", e);
  if ((e = hipMalloc(&d_out, 32)) != hipSuccess) return done(GBP_ERR_HIP, "hipMalloc", e);
  if ((e = hipMemsetAsync(buf, 0, n4 * 16, s)) != hipSuccess) return done(GBP_ERR_HIP, "hipMemsetAsync", e);
  if ((e = hipMemsetAsync(d_out, 0, 32, s)) != hipSuccess) return done(GBP_ERR_HIP, "hipMemsetAsync", e);
  if (!launch_flow_torture(buf, d_out, blocks, K, rounds, mask, 0x40000000u, inject, s)) return done(GBP_ERR_INVALID, "launch_flow_torture", hipErrorInvalidValue);
  if ((e = hipGetLastError()) != hipSuccess) return done(GBP_ERR_HIP, "k_flow_torture", e);
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return done(GBP_ERR_HIP, "hipStreamSynchronize", e);
  unsigned long long h[4];
  if ((e = hipMemcpy(h, d_out, 32, hipMemcpyDeviceToHost)) != hipSuccess) return done(GBP_ERR_HIP, "hipMemcpy", e);
  for (int i = 0; i < 4; ++i) out[i] = h[i];
  return done(GBP_OK, "", hipSuccess);
}

GBP_EXPORT(gbp_debug_layout_build, nullptr, (const gbp_problem* pr, int tile_order, const gbp_shard* sh, const gbp_layout_options* o, gbp_layout** out),
           (pr, tile_order, sh, o, out)) {
  if (!out) return GBP_ERR_INVALID;
  gbp_layout* h = new gbp_layout();
  std::string err;
  const int rc = layout_build(pr, tile_order, sh, o ? to_options(o) : g_layout_options, h->y, err);
  if (rc != GBP_OK) { delete h; return fail(nullptr, rc, err); }
  *out = h;
  return GBP_OK;
}
GBP_EXPORT(gbp_debug_layout_dims, nullptr, (const gbp_layout* h, uint32_t* d), (h, d)) {
  if (!h || !d) return GBP_ERR_INVALID;
  const Layout& y = h->y;
  const uint32_t v[11] = {y.C, y.L, y.E, y.lmk_begin, y.lmk_end, y.L_loc, y.E_loc, y.n_rows, y.n_tiles, y.Ep, y.row_window};
  std::memcpy(d, v, sizeof(v));
  return GBP_OK;
}
GBP_EXPORT(gbp_debug_layout_array, nullptr, (const gbp_layout* h, int which, const uint32_t** data, size_t* n), (h, which, data, n)) {
  if (!h || !data || !n) return GBP_ERR_INVALID;
  const Layout& y = h->y;
  const std::vector<uint32_t>* a[11] = {&y.pos_edge, &y.pos_cam, &y.pos_lmk_loc, &y.pos_lpos, &y.cam_row_ptr, &y.row_slot, &y.row_cam,
                                        &y.lmk_ptr, &y.lmk_fpos, &y.lmk_ix, &y.tile_perm};
  if (which < 0 || which > 10) return GBP_ERR_INVALID;
  *data = a[which]->data(); *n = a[which]->size();
  return GBP_OK;
}
GBP_EXPORT_VOID(gbp_debug_layout_free, (gbp_layout* h), (h)) { delete h; }
GBP_EXPORT(gbp_debug_tile_order_local, nullptr, (const uint8_t* tile_class, uint32_t n_tiles, uint32_t window, uint32_t n_classes, uint32_t* perm),
           (tile_class, n_tiles, window, n_classes, perm)) {
  if (!tile_class || !perm || window == 0 || n_classes == 0) return GBP_ERR_INVALID;
  tile_order_local(tile_class, n_tiles, window, n_classes, perm);
  return GBP_OK;
}

// Inverse of gbp_debug_get(what = 0): overwrite the factor potentials (lower triangles of the
// symmetric blocks and Lambda_cl are taken; Lambda_lc is implied).  Test hook only.
GBP_EXPORT(gbp_debug_set_factor_potentials, c, (gbp_ctx* c, const float* eta9E, const float* lam81E), (c, eta9E, lam81E)) {
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

GBP_EXPORT(gbp_debug_math, nullptr, (int op, const float* in, float* out, int n), (op, in, out, n)) { return debug_math_run(op, in, out, n, 1, nullptr); }
GBP_EXPORT(gbp_debug_math_timed, nullptr, (int op, const float* in, float* out, int n, int reps, double* avg_us), (op, in, out, n, reps, avg_us)) {
  if (!avg_us) return GBP_ERR_INVALID;
  return debug_math_run(op, in, out, n, reps, avg_us);
}


#endif  // GBP_BUILD_TEST_HOOKS
