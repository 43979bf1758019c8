// bal_convert_main.cpp — `./bal_convert <standard_bal.txt> <out.txt>`: rewrites a standard 9-parameter "Bundle
// Adjustment in the Large" file in the reference's text format (ba/sequences/README.md:5-16) so that `./ba` and
// `./slam` can read it.  The conversion itself is gbp_bal_import_standard (include/gbp_mi355x.h); SURVEY 8f-3.
#include "../../include/gbp_mi355x.h"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "usage: %s <standard_bal_file> <output_file>\n", argv[0]);
    return 2;
  }
  gbp_bal b{};
  if (gbp_bal_import_standard_header(argv[1], &b) != GBP_OK) {
    std::fprintf(stderr, "ERROR: unable to open file %s\n", argv[1]);   // wording of dataio.cpp:20-23
    return 1;
  }
  std::vector<uint32_t> cam(b.n_edges), lmk(b.n_edges);
  std::vector<double> obs(2ull * b.n_edges), cams(6ull * b.n_cams), pts(3ull * b.n_lmks);
  b.cam_id = cam.data(); b.lmk_id = lmk.data();
  b.observations = obs.data(); b.cameras = cams.data(); b.points = pts.data();
  if (gbp_bal_import_standard(argv[1], &b) != GBP_OK) {
    std::fprintf(stderr, "Invalid BAL data file: %s\n", argv[1]);
    return 1;
  }
  if (gbp_bal_write(argv[2], &b) != GBP_OK) {
    std::fprintf(stderr, "ERROR: unable to write file %s\n", argv[2]);
    return 1;
  }
  std::printf("%u cameras, %u landmarks, %u observations; shared focal length %.6f\n", b.n_cams, b.n_lmks, b.n_edges, b.fx);
  return 0;
}
