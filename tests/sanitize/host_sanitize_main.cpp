// AddressSanitizer / UBSan harness for the CPU-side code (GPU ASan is not available on this pool):
//   * the host helpers exported through the C-ABI (gbp_poplar_amd/csrc/gbp_host.cpp),
//   * the device-order builder gbp_create runs before it touches the GPU (gbp_poplar_amd/csrc/gbp_layout.cpp; layout_sanitize.cpp), and
//   * the oracle (oracle/oracle_gbp.c, oracle_math.c — test infrastructure whose answers the GPU is judged by).
// Built and run by tests/test_host_sanitizers.py:
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all gbp_host.cpp oracle_*.c host_sanitize_main.cpp
// Exit code 0 = every call returned what it should and the sanitizers stayed silent.
#include "../../include/gbp_mi355x.h"
#include "../../include/gbp_mi355x_multi.h"
#include "../../oracle/oracle.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define REQUIRE(cond)                                                          \
  do {                                                                         \
    if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

struct Bal {
  gbp_bal b{};
  std::vector<uint32_t> cam, lmk;
  std::vector<double> obs, cams, pts;
  void alloc() {
    cam.assign(b.n_edges, 0); lmk.assign(b.n_edges, 0); obs.assign(2ull * b.n_edges, 0);
    cams.assign(6ull * b.n_cams, 0); pts.assign(3ull * b.n_lmks, 0);
    b.cam_id = cam.data(); b.lmk_id = lmk.data(); b.observations = obs.data(); b.cameras = cams.data(); b.points = pts.data();
  }
};

int layout_sanitize();   // layout_sanitize.cpp: the device-order builder of gbp_create (gbp_layout.cpp)
int api_negative();      // api_negative.cpp: every export that needs no device, called with what a careless host passes

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  REQUIRE(layout_sanitize() == 0);
  REQUIRE(api_negative() == 0);
  // ---- synthetic graph -> file -> loader round trip ----
  Bal s;
  s.b.n_cams = 6; s.b.n_lmks = 50; s.b.n_edges = 50 * 4;
  s.alloc();
  REQUIRE(gbp_synth_generate(6, 50, 4, 99, &s.b, nullptr, nullptr) == GBP_OK);
  const std::string path = dir + "/sanitize_bal.txt";
  REQUIRE(gbp_bal_write(path.c_str(), &s.b) == GBP_OK);
  Bal r;
  REQUIRE(gbp_bal_read_header(path.c_str(), &r.b) == GBP_OK);
  REQUIRE(r.b.n_cams == 6 && r.b.n_lmks == 50 && r.b.n_edges == 200);
  r.alloc();
  REQUIRE(gbp_bal_read(path.c_str(), &r.b) == GBP_OK);
  REQUIRE(r.cam == s.cam && r.lmk == s.lmk && r.obs == s.obs);
  REQUIRE(gbp_bal_read_header((dir + "/does_not_exist.txt").c_str(), &r.b) == GBP_ERR_IO);
  {  // truncated file: header promises more than the body holds
    const std::string t = dir + "/sanitize_truncated.txt";
    FILE* f = std::fopen(t.c_str(), "w");
    REQUIRE(f != nullptr);
    std::fprintf(f, "6 50 200\n500 500 320 240\n0 0 1.0 2.0\n");
    std::fclose(f);
    REQUIRE(gbp_bal_read(t.c_str(), &r.b) == GBP_ERR_IO);
    REQUIRE(orc_bal_read(t.c_str(), &r.b) != GBP_OK);
  }
  {  // standard 9-parameter BAL file (hand-written, 2 cameras x 3 points)
    const std::string t = dir + "/sanitize_standard.txt";
    FILE* f = std::fopen(t.c_str(), "w");
    REQUIRE(f != nullptr);
    std::fprintf(f, "2 3 6\n");
    for (int l = 0; l < 3; ++l) for (int c = 0; c < 2; ++c) std::fprintf(f, "%d %d %f %f\n", c, l, 10.0 * l - 5 * c, 3.0 * c + l);
    for (int c = 0; c < 2; ++c) std::fprintf(f, "%f\n%f\n%f\n0\n0\n-5\n%f\n1e-2\n1e-3\n", 0.01 * c, 0.02, -0.01, 500.0 + 20 * c);
    for (int i = 0; i < 9; ++i) std::fprintf(f, "%f\n", 0.1 * i);
    std::fclose(f);
    Bal st;
    REQUIRE(gbp_bal_import_standard_header(t.c_str(), &st.b) == GBP_OK);
    st.alloc();
    REQUIRE(gbp_bal_import_standard(t.c_str(), &st.b) == GBP_OK);
    REQUIRE(st.b.fx == 510.0 && st.cam[0] == 0 && st.cam[5] == 1);
    st.b.n_edges = 7;   // header mismatch must be refused, not overrun
    REQUIRE(gbp_bal_import_standard(t.c_str(), &st.b) == GBP_ERR_IO);
  }

  // ---- priors, scalings, SLAM flags, metric ----
  const uint32_t C = r.b.n_cams, L = r.b.n_lmks, E = r.b.n_edges;
  gbp_problem prob{C, L, E, r.cam.data(), r.lmk.data(), {(float)r.b.fx, 0, (float)r.b.cx, 0, (float)r.b.fy, (float)r.b.cy, 0, 0, 1}};
  std::vector<float> camf(r.cams.begin(), r.cams.end()), lmkf(r.pts.begin(), r.pts.end());
  std::vector<float> ce(6 * C), cl(36 * C), le(3 * L), ll(9 * L), oe(6 * C), ol(36 * C), oe3(3 * L), ol9(9 * L);
  REQUIRE(gbp_set_prior_lambda(&prob, 4.f, camf.data(), lmkf.data(), camf.data(), lmkf.data(), ce.data(), cl.data(), le.data(), ll.data()) == GBP_OK);
  REQUIRE(orc_set_prior_lambda(&prob, 4.f, camf.data(), lmkf.data(), camf.data(), lmkf.data(), oe.data(), ol.data(), oe3.data(), ol9.data()) == GBP_OK);
  REQUIRE(ce == oe && cl == ol && le == oe3 && ll == ol9);
  std::vector<float> cs(C), ls(L), ocs(C), ols(L);
  REQUIRE(gbp_prior_scalings(C, L, cl.data(), 5.f, 100.f, 0.01f, cs.data(), ls.data()) == GBP_OK);
  REQUIRE(orc_prior_scalings(C, L, cl.data(), 5.f, 100.f, 0.01f, ocs.data(), ols.data()) == GBP_OK);
  REQUIRE(cs == ocs && ls == ols);
  std::vector<uint32_t> act(E), cwf(C), lwf(L), laf(L);
  REQUIRE(gbp_slam_create_flags(&prob, 5, act.data(), cwf.data(), lwf.data(), laf.data()) == GBP_OK);
  int32_t n_new = -1;
  for (uint32_t dc = 1; dc + 1 < C; ++dc)
    REQUIRE(gbp_slam_update_flags(&prob, 5, dc, act.data(), lwf.data(), cwf.data(), laf.data(), &n_new) == GBP_OK && n_new >= 0);

  // initialisation options + landmark partition (round 2 additions to the host library)
  {
    std::vector<float> cn = camf, ln = lmkf, cn2 = camf, ln2 = lmkf;
    REQUIRE(gbp_init_add_noise(C, L, 0.05f, 2.f, 0.1f, 7, cn.data(), ln.data()) == GBP_OK);
    REQUIRE(gbp_init_add_noise(C, L, 0.05f, 2.f, 0.1f, 7, cn2.data(), ln2.data()) == GBP_OK);
    REQUIRE(cn == cn2 && ln == ln2 && cn != camf);
    REQUIRE(std::equal(cn.begin(), cn.begin() + 12, camf.begin()));            // cameras 0 and 1 are the gauge anchors
    std::vector<float> ld = lmkf;
    REQUIRE(gbp_init_av_depth(&prob, camf.data(), ld.data()) == GBP_OK);
    REQUIRE(gbp_init_av_depth(&prob, nullptr, ld.data()) == GBP_ERR_INVALID);
    for (int world : {1, 2, 3, 8, 64}) {
      std::vector<uint32_t> bounds((size_t)world + 1, 12345u);
      REQUIRE(gbp_landmark_partition(&prob, world, bounds.data()) == GBP_OK);
      REQUIRE(bounds.front() == 0 && bounds.back() == L);
      for (int k = 0; k < world; ++k) REQUIRE(bounds[k] <= bounds[k + 1]);
    }
    REQUIRE(gbp_landmark_partition(&prob, 0, nullptr) == GBP_ERR_INVALID);
    const size_t bytes = gbp_comm_region_bytes(C, 4);
    std::vector<char> region(bytes);
    REQUIRE(gbp_comm_region_init(region.data(), bytes, C, 4) == GBP_OK);
    REQUIRE(gbp_comm_region_init(region.data(), bytes - 1, C, 4) == GBP_ERR_INVALID);
    gbp_comm_region_abort(region.data());
  }

  // ---- oracle: BA flow with weakening, sharded split-phase flow, read-backs ----
  std::vector<uint32_t> active(E, 1u), cw(C, 5u), lw(L, 5u);
  std::vector<int32_t> count(E, -2);
  std::vector<float> meas(2 * E), var(E, 4.f);
  for (size_t i = 0; i < meas.size(); ++i) meas[i] = (float)r.obs[i];
  gbp_state_in in{};
  in.damping_count = count.data(); in.active_flag = active.data(); in.cam_scaling = cs.data(); in.lmk_scaling = ls.data();
  in.cam_weaken_flag = cw.data(); in.lmk_weaken_flag = lw.data(); in.cam_priors_eta = ce.data(); in.cam_priors_lambda = cl.data();
  in.lmk_priors_eta = le.data(); in.lmk_priors_lambda = ll.data(); in.measurements = meas.data(); in.meas_variances = var.data();
  std::vector<float> be(6 * C), bl(36 * C), ble(3 * L), bll(9 * L), damp(E);
  std::vector<int32_t> cnt(E);
  std::vector<uint32_t> rob(E);
  gbp_state_out out{be.data(), bl.data(), ble.data(), bll.data(), damp.data(), cnt.data(), rob.data()};
  gbp_eval_out ev0{}, ev1{};
  for (int order = 0; order < 2; ++order) {
    orc_ctx* o = orc_create(&prob, nullptr);
    REQUIRE(o != nullptr);
    REQUIRE(orc_set_sum_order(o, order, 1, nullptr) == GBP_OK);
    REQUIRE(orc_upload(o, &in) == GBP_OK && orc_linearise(o) == GBP_OK && orc_eval(o, &ev0) == GBP_OK);
    for (int it = 0; it < 14; ++it) {
      if ((it + 1) % 2 == 0 && it < 10) REQUIRE(orc_weaken_priors(o) == GBP_OK);
      REQUIRE(orc_iterate(o, 1) == GBP_OK);
    }
    REQUIRE(orc_eval(o, &ev1) == GBP_OK && orc_read(o, &out) == GBP_OK);
    REQUIRE(ev1.n_active == E && ev1.n_nonfinite == 0 && ev1.sum_half_sq < ev0.sum_half_sq);
    double sn = 0, sh = 0;
    uint64_t na = 0;
    REQUIRE(gbp_eval_host(&prob, active.data(), meas.data(), be.data(), bl.data(), ble.data(), bll.data(), &sn, &sh, &na) == GBP_OK);
    REQUIRE(na == E && std::fabs(sh - ev1.sum_half_sq) <= 1e-4 * ev1.sum_half_sq);
    std::vector<float> pe(6 * C), pl(36 * C), ple(3 * L), pll(9 * L), fe(9ull * E), fl(81ull * E), m1(6ull * E), m2(36ull * E), m3(3ull * E), m4(9ull * E), mu(9ull * E), dmu(E);
    gbp_priors_out po{pe.data(), pl.data(), ple.data(), pll.data()};
    REQUIRE(orc_read_priors(o, &po) == GBP_OK);
    REQUIRE(orc_get_factor_potentials(o, fe.data(), fl.data()) == GBP_OK);
    REQUIRE(orc_get_messages(o, m1.data(), m2.data(), m3.data(), m4.data()) == GBP_OK && orc_get_mu(o, mu.data(), dmu.data()) == GBP_OK);
    REQUIRE(gbp_slam_initialise_new_kf(1, be.data(), bl.data(), pl.data(), pe.data()) == GBP_OK);
    gbp_kf_update kf{};
    kf.damping_count = count.data(); kf.cam_priors_eta = pe.data(); kf.cam_priors_lambda = pl.data();
    kf.lmk_priors_eta = ple.data(); kf.lmk_priors_lambda = pll.data(); kf.active_flag = act.data();
    kf.cam_weaken_flag = cwf.data(); kf.lmk_weaken_flag = lwf.data();
    REQUIRE(orc_new_keyframe(o, &kf) == GBP_OK && orc_iterate(o, 2) == GBP_OK);
    orc_destroy(o);
  }
  {  // two shards in one process, exchange by copying the 44-float camera records
    const uint32_t bounds[3] = {0, L / 2, L};
    orc_ctx* sh[2];
    std::vector<float> send[2], recv(2ull * C * 44);
    for (int k = 0; k < 2; ++k) {
      sh[k] = orc_create(&prob, nullptr);
      REQUIRE(sh[k] != nullptr && orc_set_sum_order(sh[k], 1, 2, bounds) == GBP_OK);
      REQUIRE(orc_set_shard(sh[k], k, 2, bounds[k], bounds[k + 1]) == GBP_OK && orc_upload(sh[k], &in) == GBP_OK);
      send[k].assign((size_t)C * 44, 0.f);
    }
    auto exchange = [&]() { for (int k = 0; k < 2; ++k) std::copy(send[k].begin(), send[k].end(), recv.begin() + (size_t)k * C * 44); };
    for (int k = 0; k < 2; ++k) REQUIRE(orc_refresh_begin(sh[k], send[k].data()) == GBP_OK);
    exchange();
    for (int k = 0; k < 2; ++k) REQUIRE(orc_refresh_end(sh[k], recv.data()) == GBP_OK && orc_linearise_factors(sh[k]) == GBP_OK);
    for (int it = 0; it < 6; ++it) {
      if (it == 1) for (int k = 0; k < 2; ++k) REQUIRE(orc_weaken_priors_sharded(sh[k], recv.data()) == GBP_OK);
      for (int k = 0; k < 2; ++k) REQUIRE(orc_iterate_begin(sh[k], send[k].data()) == GBP_OK);
      exchange();
      for (int k = 0; k < 2; ++k) REQUIRE(orc_iterate_end(sh[k], recv.data()) == GBP_OK);
    }
    gbp_kf_update kf{};
    kf.damping_count = count.data(); kf.active_flag = active.data(); kf.cam_weaken_flag = cw.data(); kf.lmk_weaken_flag = lw.data();
    kf.lmk_priors_eta = le.data(); kf.lmk_priors_lambda = ll.data();
    for (int k = 0; k < 2; ++k) REQUIRE(orc_new_keyframe_sharded(sh[k], &kf, recv.data()) == GBP_OK);
    gbp_eval_out a{}, b{};
    REQUIRE(orc_eval(sh[0], &a) == GBP_OK && orc_eval(sh[1], &b) == GBP_OK && a.n_active + b.n_active == E);
    orc_destroy(sh[0]);
    orc_destroy(sh[1]);
  }
  std::puts("sanitize: ok");
  return 0;
}
