// Negative tests of the C-ABI under AddressSanitizer + UBSan (tests/test_host_sanitizers.py): every export of include/gbp_mi355x.h,
// gbp_mi355x_multi.h and gbp_mi355x_compat.h that can be reached without a device is called with what a careless host passes —
// NULL ctx, NULL outputs, negative counts, calls out of order, a region that is too small — and must answer with a status (never a
// crash, never a C++ exception: the exports run inside csrc/gbp_export.hpp's guard).  "Calls out of order" need a ctx: this file is
// part of the library's build (it includes the internal header) and makes one WITHOUT a device, as gbp_create leaves it before the
// first upload.
#include "../../gbp_poplar_amd/csrc/gbp_ctx.hpp"

#include <cstdio>
#include <cstring>
#include <vector>

#define REQUIRE(cond)                                                          \
  do {                                                                         \
    if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

static int null_ctx_calls() {
  gbp_eval_out ev{};
  gbp_state_in in{};
  gbp_state_out out{};
  gbp_priors_out pout{};
  gbp_kf_update kf{};
  gbp_timing_out tm{};
  double us = 0;
  char buf[64];
  REQUIRE(gbp_abi_version() == GBP_ABI_VERSION);
  gbp_default_params(nullptr);
  gbp_destroy(nullptr);
  REQUIRE(gbp_last_error(nullptr) != nullptr);
  REQUIRE(gbp_upload(nullptr, &in) == GBP_ERR_INVALID);
  REQUIRE(gbp_linearise(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_iterate(nullptr, 3) == GBP_ERR_STATE);
  REQUIRE(gbp_prepare(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_weaken_priors(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_read(nullptr, &out) == GBP_ERR_INVALID);
  REQUIRE(gbp_read_priors(nullptr, &pout) == GBP_ERR_INVALID);
  REQUIRE(gbp_new_keyframe(nullptr, &kf) == GBP_ERR_STATE);
  REQUIRE(gbp_eval(nullptr, &ev) == GBP_ERR_STATE);
  REQUIRE(gbp_ba_loop(nullptr, 3, 0, 5, &ev) == GBP_ERR_STATE);
  REQUIRE(gbp_sync(nullptr) == GBP_ERR_INVALID);
  REQUIRE(gbp_timing(nullptr, &tm, 0) == GBP_ERR_INVALID);
  REQUIRE(gbp_graph_state(nullptr) == 0);
  REQUIRE(gbp_set_profiling(nullptr, 1) == GBP_ERR_INVALID);
  // gbp_mi355x_compat.h
  REQUIRE(gbp_eval_begin(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_eval_end(nullptr, &ev) == GBP_ERR_INVALID);
  REQUIRE(gbp_iterate_eval(nullptr, 2) == GBP_ERR_STATE);
  REQUIRE(gbp_iterate_eval_each(nullptr, 2, &ev) == GBP_ERR_STATE);
  // gbp_mi355x_multi.h
  REQUIRE(gbp_set_stream(nullptr, nullptr) == GBP_ERR_INVALID);
  REQUIRE(gbp_set_exchange_buffers(nullptr, nullptr, nullptr) == GBP_ERR_INVALID);
  REQUIRE(gbp_iterate_begin(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_iterate_local(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_iterate_end(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_refresh_begin(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_refresh_end(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_linearise_factors(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_comm_init(nullptr, buf, 0) == GBP_ERR_INVALID);
  REQUIRE(gbp_comm_init_rccl(nullptr, buf) == GBP_ERR_INVALID);
  REQUIRE(gbp_comm_unique_id(nullptr) == GBP_ERR_INVALID);
  REQUIRE(std::strcmp(gbp_comm_transport(nullptr), "none") == 0);
  REQUIRE(gbp_comm_barrier(nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_comm_describe(nullptr, buf, sizeof(buf)) == GBP_ERR_INVALID);
  REQUIRE(gbp_comm_set_schedule(nullptr, 1) == GBP_ERR_STATE);
  REQUIRE(gbp_comm_probe(nullptr, 3, &us) == GBP_ERR_STATE);
  REQUIRE(gbp_eval_global(nullptr, &ev) == GBP_ERR_INVALID);
  REQUIRE(gbp_landmark_partition(nullptr, 2, nullptr) == GBP_ERR_INVALID);
  return 0;
}

static int create_without_a_device() {
  // a valid 2 x 3 problem: the device order is built and validated on the host, then gbp_create finds no GPU (this harness runs on
  // the CPU box; on a GPU box the call succeeds and the ctx is destroyed again)
  const uint32_t cam[6] = {0, 0, 0, 1, 1, 1}, lmk[6] = {0, 1, 2, 0, 1, 2};
  gbp_problem pr{};
  pr.n_cams = 2; pr.n_lmks = 3; pr.n_edges = 6; pr.cam_id = cam; pr.lmk_id = lmk;
  const float K[9] = {500, 0, 320, 0, 500, 240, 0, 0, 1};
  std::memcpy(pr.K, K, sizeof(K));
  gbp_ctx* c = nullptr;
  REQUIRE(gbp_create(nullptr, nullptr, nullptr, &c) == GBP_ERR_INVALID && c == nullptr);
  REQUIRE(gbp_create(&pr, nullptr, nullptr, nullptr) == GBP_ERR_INVALID);
  gbp_problem bad = pr;
  bad.n_edges = 0;
  REQUIRE(gbp_create(&bad, nullptr, nullptr, &c) == GBP_ERR_INVALID);
  const uint32_t lmk_bad[6] = {0, 1, 7, 0, 1, 2};          // landmark index out of range: refused by the layout builder
  bad = pr; bad.lmk_id = lmk_bad;
  REQUIRE(gbp_create(&bad, nullptr, nullptr, &c) == GBP_ERR_INVALID && std::strlen(gbp_last_error(nullptr)) > 0);
  gbp_shard sh{2, 2, 0, 3};                                 // rank >= world
  REQUIRE(gbp_create(&pr, nullptr, &sh, &c) == GBP_ERR_INVALID);
  const int rc = gbp_create(&pr, nullptr, nullptr, &c);
  if (gbp_device_count() == 0) {
    REQUIRE(rc == GBP_ERR_NO_DEVICE && c == nullptr && std::strstr(gbp_last_error(nullptr), "no HIP device") != nullptr);
    REQUIRE(gbp_set_device(0) == GBP_ERR_NO_DEVICE);
  } else {
    REQUIRE(rc == GBP_OK && c != nullptr);
    gbp_destroy(c);
  }
  return 0;
}

static int calls_out_of_order() {
  // a ctx as gbp_create leaves it before the first gbp_upload — built here WITHOUT a device (no stream, no buffers): every program of
  // the list must refuse to run on it, nothing may touch the (absent) device
  gbp_ctx* c = new gbp_ctx();
  c->C = 2; c->L = 3; c->E = 6; c->L_loc = 3; c->E_loc = 6; c->lmk_end = 3;
  gbp_default_params(&c->prm);
  gbp_eval_out ev[4] = {};
  gbp_kf_update kf{};
  gbp_state_in in{};
  REQUIRE(gbp_linearise(c) == GBP_ERR_STATE && std::strstr(gbp_last_error(c), "upload first") != nullptr);
  REQUIRE(gbp_iterate(c, 5) == GBP_ERR_STATE);
  REQUIRE(gbp_prepare(c) == GBP_ERR_STATE);
  REQUIRE(gbp_weaken_priors(c) == GBP_ERR_STATE);
  REQUIRE(gbp_new_keyframe(c, &kf) == GBP_ERR_STATE);
  REQUIRE(gbp_new_keyframe(c, nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_eval(c, ev) == GBP_ERR_STATE);
  REQUIRE(gbp_eval(c, nullptr) == GBP_ERR_STATE);
  REQUIRE(gbp_ba_loop(c, 4, 0, 5, ev) == GBP_ERR_STATE);
  REQUIRE(gbp_eval_begin(c) == GBP_ERR_STATE);
  REQUIRE(gbp_eval_end(c, ev) == GBP_ERR_STATE);                    // no evaluation in flight
  REQUIRE(gbp_eval_end(c, nullptr) == GBP_ERR_INVALID);
  REQUIRE(gbp_iterate_eval(c, 3) == GBP_ERR_STATE);
  REQUIRE(gbp_iterate_eval_each(c, 3, ev) == GBP_ERR_STATE);
  REQUIRE(gbp_iterate_begin(c) == GBP_ERR_STATE);
  REQUIRE(gbp_iterate_local(c) == GBP_ERR_STATE);
  REQUIRE(gbp_iterate_end(c) == GBP_ERR_STATE);
  REQUIRE(gbp_refresh_begin(c) == GBP_ERR_STATE);
  REQUIRE(gbp_refresh_end(c) == GBP_ERR_STATE);
  REQUIRE(gbp_linearise_factors(c) == GBP_ERR_STATE);
  REQUIRE(gbp_comm_set_schedule(c, 1) == GBP_ERR_STATE);            // no communicator
  double us = 0;
  REQUIRE(gbp_comm_probe(c, 3, &us) == GBP_ERR_STATE);
  REQUIRE(gbp_comm_barrier(c) == GBP_ERR_STATE);
  REQUIRE(gbp_comm_init(c, nullptr, 0) == GBP_ERR_INVALID);
  REQUIRE(gbp_comm_init_rccl(c, nullptr) == GBP_ERR_INVALID);
  REQUIRE(std::strcmp(gbp_comm_transport(c), "none") == 0);
  REQUIRE(gbp_graph_state(c) == 0);
  REQUIRE(gbp_upload(c, nullptr) == GBP_ERR_INVALID);
  REQUIRE(gbp_upload(c, &in) == GBP_ERR_INVALID && std::strstr(gbp_last_error(c), "required") != nullptr);   // priors / measurements missing
  REQUIRE(gbp_read(c, nullptr) == GBP_ERR_INVALID);
  REQUIRE(gbp_read_priors(c, nullptr) == GBP_ERR_INVALID);
  REQUIRE(gbp_timing(c, nullptr, 0) == GBP_ERR_INVALID);
  REQUIRE(gbp_eval_global(c, nullptr) == GBP_ERR_INVALID);
  REQUIRE(gbp_set_exchange_buffers(c, nullptr, nullptr) == GBP_OK);
  REQUIRE(gbp_set_profiling(c, 1) == GBP_OK && gbp_set_profiling(c, 0) == GBP_OK);
  // pretend the upload happened: the argument checks behind it (still nothing that would touch a device)
  c->uploaded = true;
  REQUIRE(gbp_ba_loop(c, -1, 0, 5, ev) == GBP_ERR_INVALID);
  REQUIRE(gbp_ba_loop(c, 2, 0, 1u << 30, ev) == GBP_ERR_INVALID);      // 2 * steps would wrap
  REQUIRE(gbp_iterate_eval_each(c, -2, ev) == GBP_ERR_INVALID);
  REQUIRE(gbp_iterate_eval_each(c, 2, nullptr) == GBP_ERR_INVALID);
  REQUIRE(gbp_iterate(c, 0) == GBP_OK && gbp_iterate(c, -3) == GBP_OK);      // nothing to do
  REQUIRE(gbp_eval_end(c, ev) == GBP_ERR_STATE);
  c->world = 2;                                                      // a sharded ctx without a communicator
  REQUIRE(gbp_iterate(c, 2) == GBP_ERR_STATE && std::strstr(gbp_last_error(c), "communicator") != nullptr);
  REQUIRE(gbp_linearise(c) == GBP_ERR_STATE);
  REQUIRE(gbp_iterate_begin(c) == GBP_ERR_STATE && std::strstr(gbp_last_error(c), "exchange buffers") != nullptr);
  REQUIRE(gbp_refresh_begin(c) == GBP_ERR_STATE);
  c->world = 1;
  c->uploaded = false;
  gbp_destroy(c);
  return 0;
}

static int region_calls() {
  const size_t need = gbp_comm_region_bytes(8, 4);
  REQUIRE(need > 0 && gbp_comm_region_bytes(8, 0) == 0 && gbp_comm_region_bytes(8, 1000) == 0);
  std::vector<unsigned char> mem(need + 64, 0xAB);
  REQUIRE(gbp_comm_region_init(nullptr, need, 8, 4) == GBP_ERR_INVALID);
  REQUIRE(gbp_comm_region_init(mem.data(), need - 1, 8, 4) == GBP_ERR_INVALID);      // too small
  REQUIRE(gbp_comm_region_init(mem.data(), need, 8, 0) == GBP_ERR_INVALID);
  REQUIRE(gbp_comm_region_init(mem.data(), need, 8, 4) == GBP_OK);
  for (size_t i = need; i < mem.size(); ++i) REQUIRE(mem[i] == 0xAB);               // nothing written behind the region
  REQUIRE(gbp_comm_region_selftest(nullptr, 0, 1, 1) == GBP_ERR_INVALID);
  REQUIRE(gbp_comm_region_selftest(mem.data(), 4, 4, 1) == GBP_ERR_INVALID);        // rank >= world
  gbp_comm_region_abort(nullptr);
  gbp_comm_region_abort(mem.data());
  REQUIRE(gbp_comm_region_selftest(mem.data(), 0, 4, 1) == GBP_ERR_COMM);           // aborted: the waiting rank leaves with an error
  {  // a 1-rank region: the protocol alone
    const size_t n1 = gbp_comm_region_bytes(8, 1);
    std::vector<unsigned char> m1(n1);
    REQUIRE(gbp_comm_region_init(m1.data(), n1, 8, 1) == GBP_OK);
    REQUIRE(gbp_comm_region_selftest(m1.data(), 0, 1, 5) == GBP_OK);
  }
  // host helpers with NULL / inconsistent arguments
  uint32_t bounds[3];
  gbp_problem pr{};
  REQUIRE(gbp_landmark_partition(&pr, 0, bounds) == GBP_ERR_INVALID);
  return 0;
}

int api_negative() {
  if (null_ctx_calls()) return 1;
  if (create_without_a_device()) return 1;
  if (calls_out_of_order()) return 1;
  if (region_calls()) return 1;
  return 0;
}
