// ThreadSanitizer harness of the host code that runs on several threads (gbp_host.cpp: read_number_file behind gbp_bal_read /
// gbp_bal_import_standard, the per-thread maxima of gbp_set_prior_lambda): a file above the 1 MB threshold read with 7 threads must
// equal what was written, the prior strengths must not depend on the thread count, and TSan must have nothing to report.
// Built and run by tests/test_host_sanitizers.py (gbp_host.cpp needs nothing of the device-facing translation units but fail()).
#include "../../include/gbp_mi355x.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace gbp { namespace api { int fail(gbp_ctx*, int code, const std::string&) { return code; } } }

#define REQUIRE(x) do { if (!(x)) { std::fprintf(stderr, "host_threads_main.cpp:%d: %s\n", __LINE__, #x); return 1; } } while (0)

struct Bal {
  gbp_bal b{};
  std::vector<uint32_t> cam, lmk;
  std::vector<double> obs, cams, pts;
  void alloc() {
    cam.assign(b.n_edges, 0); lmk.assign(b.n_edges, 0); obs.assign(2ull * b.n_edges, 0); cams.assign(6ull * b.n_cams, 0); pts.assign(3ull * b.n_lmks, 0);
    b.cam_id = cam.data(); b.lmk_id = lmk.data(); b.observations = obs.data(); b.cameras = cams.data(); b.points = pts.data();
  }
};

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  Bal s;
  s.b.n_cams = 60; s.b.n_lmks = 9000; s.b.n_edges = 9000 * 8;
  s.alloc();
  REQUIRE(gbp_synth_generate(60, 9000, 8, 4, &s.b, nullptr, nullptr) == GBP_OK);
  const std::string path = dir + "/tsan_bal.txt";
  REQUIRE(gbp_bal_write(path.c_str(), &s.b) == GBP_OK);
  setenv("GBP_HOST_THREADS", "7", 1);
  Bal r;
  REQUIRE(gbp_bal_read_header(path.c_str(), &r.b) == GBP_OK);
  REQUIRE(r.b.n_cams == 60 && r.b.n_lmks == 9000 && r.b.n_edges == 72000);
  r.alloc();
  REQUIRE(gbp_bal_read(path.c_str(), &r.b) == GBP_OK);
  REQUIRE(r.cam == s.cam && r.lmk == s.lmk && r.obs == s.obs && r.cams == s.cams && r.pts == s.pts);

  Bal g;      // 270 000 factors: four ranges of factors (one thread per 65 536 at most)
  g.b.n_cams = 60; g.b.n_lmks = 9000; g.b.n_edges = 9000 * 30;
  g.alloc();
  REQUIRE(gbp_synth_generate(60, 9000, 30, 5, &g.b, nullptr, nullptr) == GBP_OK);
  gbp_problem prob{};
  prob.n_cams = 60; prob.n_lmks = 9000; prob.n_edges = g.b.n_edges; prob.cam_id = g.cam.data(); prob.lmk_id = g.lmk.data();
  const float K[9] = {(float)g.b.fx, 0.f, (float)g.b.cx, 0.f, (float)g.b.fy, (float)g.b.cy, 0.f, 0.f, 1.f};
  for (int i = 0; i < 9; ++i) prob.K[i] = K[i];
  std::vector<float> camf(g.cams.begin(), g.cams.end()), lmkf(g.pts.begin(), g.pts.end());
  std::vector<float> out[2][4];
  const char* threads[2] = {"1", "5"};
  for (int k = 0; k < 2; ++k) {
    setenv("GBP_HOST_THREADS", threads[k], 1);
    out[k][0].assign(6 * 60, 0.f); out[k][1].assign(36 * 60, 0.f); out[k][2].assign(3 * 9000, 0.f); out[k][3].assign(9 * 9000, 0.f);
    REQUIRE(gbp_set_prior_lambda(&prob, 4.f, camf.data(), lmkf.data(), camf.data(), lmkf.data(), out[k][0].data(), out[k][1].data(),
                                 out[k][2].data(), out[k][3].data()) == GBP_OK);
  }
  for (int i = 0; i < 4; ++i) REQUIRE(out[0][i] == out[1][i]);
  std::printf("tsan: ok\n");
  return 0;
}
