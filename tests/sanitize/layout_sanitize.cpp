// The device-order builder (gbp_poplar_amd/csrc/gbp_layout.cpp: what gbp_create runs before it touches the GPU) under ASan +
// UBSan, on the graph families of tests/test_layout.py: random / unsorted / duplicate-edge files, empty shards, one camera, one
// landmark, the config-5 shard shape — with the product's options and with shrunken thresholds that switch the row placement and
// the tile permutation on for graphs of a few tiles.  The checks are a C++ restatement of the PROPERTIES (every factor placed once,
// rows whole and in the camera's file order, pads flagged, landmark slots in file order — ba/ba.cpp:267-279 —, row_slot a
// bijection inside each window, tile_perm a bijection), not of the construction.
#include "../../gbp_poplar_amd/csrc/gbp_layout.hpp"

#include <algorithm>
#include <cstdio>
#include <numeric>
#include <string>
#include <vector>

#define REQUIRE(cond)                                                          \
  do {                                                                         \
    if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

namespace {
struct Rng {                      // splitmix64
  uint64_t s;
  uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
  uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
};

int check(const gbp_problem& pr, const gbp_shard* sh, const gbp::LayoutOptions& opt, int tile_order, const gbp::Layout& y) {
  using gbp::kNoEdge;
  const uint32_t lo = sh ? sh->lmk_begin : 0, hi = sh ? sh->lmk_end : pr.n_lmks;
  REQUIRE(y.C == pr.n_cams && y.L == pr.n_lmks && y.E == pr.n_edges && y.lmk_begin == lo && y.lmk_end == hi && y.L_loc == hi - lo);
  REQUIRE(y.Ep % 256 == 0 && y.n_tiles * 64 == y.Ep && y.Ep >= 256 && (size_t)y.n_rows * 16 <= y.Ep);
  REQUIRE(y.pos_edge.size() == y.Ep && y.pos_cam.size() == y.Ep && y.pos_lmk_loc.size() == y.Ep && y.pos_lpos.size() == y.Ep);
  REQUIRE(y.cam_row_ptr.size() == (size_t)pr.n_cams + 1 && y.lmk_ptr.size() == (size_t)y.L_loc + 1 && y.row_cam.size() == y.Ep / 16);
  REQUIRE(y.row_slot.empty() || y.row_slot.size() == y.n_rows);
  REQUIRE(y.lmk_fpos.size() == y.E_loc && y.lmk_ix.size() == (size_t)y.L_loc * 16);
  // walk the file: the i-th local factor of camera c must sit in lane i % 16 of the camera's (i / 16)-th row; the k-th local
  // factor of landmark l must be slot k of l
  std::vector<uint32_t> cfill(pr.n_cams, 0), lfill(y.L_loc, 0);
  std::vector<uint8_t> hit(y.Ep, 0);
  uint32_t n_local = 0;
  for (uint32_t e = 0; e < pr.n_edges; ++e) {
    const uint32_t c = pr.cam_id[e], l = pr.lmk_id[e];
    if (l < lo || l >= hi) continue;
    ++n_local;
    const uint32_t i = cfill[c]++, k = lfill[l - lo]++;
    const uint32_t r = y.cam_row_ptr[c] + i / 16;
    REQUIRE(r < y.cam_row_ptr[c + 1]);
    const uint32_t dr = y.row_slot.empty() ? r : y.row_slot[r];
    REQUIRE(dr < y.n_rows);
    const uint32_t p = dr * 16 + i % 16;
    REQUIRE(y.pos_edge[p] == e && !hit[p]);
    hit[p] = 1;
    REQUIRE(y.pos_cam[p] == c && y.row_cam[dr] == c && y.pos_lmk_loc[p] == l - lo);
    REQUIRE(y.pos_lpos[p] == y.lmk_ptr[l - lo] + k && y.lmk_fpos[y.lmk_ptr[l - lo] + k] == p);
    if (k < 15) REQUIRE(y.lmk_ix[(size_t)(l - lo) * 16 + 1 + k] == p);
  }
  REQUIRE(n_local == y.E_loc);
  for (uint32_t c = 0; c < pr.n_cams; ++c) REQUIRE(y.cam_row_ptr[c + 1] - y.cam_row_ptr[c] == (cfill[c] + 15) / 16);
  for (uint32_t l = 0; l < y.L_loc; ++l) REQUIRE(y.lmk_ptr[l + 1] - y.lmk_ptr[l] == lfill[l] && y.lmk_ix[(size_t)l * 16] == lfill[l]);
  for (size_t p = 0; p < y.Ep; ++p)
    if (!hit[p]) REQUIRE(y.pos_edge[p] == kNoEdge && y.pos_lpos[p] == y.E_loc && y.pos_cam[p] < pr.n_cams && y.pos_lmk_loc[p] < std::max(y.L_loc, 1u));
  if (!y.row_slot.empty()) {      // a bijection inside every window of cameras
    REQUIRE(y.row_window == opt.row_window);
    for (uint32_t c0 = 0; c0 < pr.n_cams; c0 += y.row_window) {
      const uint32_t c1 = std::min(pr.n_cams, c0 + y.row_window), R0 = y.cam_row_ptr[c0], R1 = y.cam_row_ptr[c1];
      std::vector<uint32_t> s(y.row_slot.begin() + R0, y.row_slot.begin() + R1);
      std::sort(s.begin(), s.end());
      for (uint32_t r = R0; r < R1; ++r) REQUIRE(s[r - R0] == r);
    }
  } else {
    REQUIRE(y.row_window == 0);
  }
  if (!y.tile_perm.empty()) {
    REQUIRE(y.tile_perm.size() == y.n_tiles);
    std::vector<uint32_t> s(y.tile_perm);
    std::sort(s.begin(), s.end());
    for (uint32_t t = 0; t < y.n_tiles; ++t) REQUIRE(s[t] == t);
    if (tile_order != 2)
      for (uint32_t t = 0; t < y.n_tiles; ++t) {
        const long d = (long)y.tile_perm[t] - (long)t;
        REQUIRE(d <= (long)opt.tile_window + 32 && -d <= (long)opt.tile_window + 32);
      }
  }
  return 0;
}

int run(const std::vector<uint32_t>& cam, const std::vector<uint32_t>& lmk, uint32_t C, uint32_t L, const gbp_shard* sh) {
  gbp_problem pr{C, L, (uint32_t)cam.size(), cam.data(), lmk.data(), {1, 0, 0, 0, 1, 0, 0, 0, 1}};
  gbp::LayoutOptions small;
  small.tile_min_tiles = 4; small.row_window = 3; small.row_place_max_deg = 1000000;
  gbp::LayoutOptions fine = small;
  fine.classes = 16; fine.row_key_lane = 8; fine.tile_window = 7;
  const gbp::LayoutOptions opts[3] = {gbp::LayoutOptions(), small, fine};
  for (const gbp::LayoutOptions& o : opts)
    for (int tile_order = 0; tile_order <= 3; ++tile_order) {
      gbp::Layout y;
      std::string err;
      REQUIRE(gbp::layout_build(&pr, tile_order, sh, o, y, err) == GBP_OK);
      if (check(pr, sh, o, tile_order, y)) return 1;
    }
  return 0;
}
}  // namespace

int layout_sanitize() {
  Rng rng{20200303};
  for (int g = 0; g < 12; ++g) {     // random files: unsorted, with duplicate edges, ragged degrees, variables without factors
    const uint32_t C = 1 + rng.below(40), L = 1 + rng.below(300), E = 1 + rng.below(4000);
    std::vector<uint32_t> cam(E), lmk(E);
    for (uint32_t e = 0; e < E; ++e) { cam[e] = rng.below(C); lmk[e] = rng.below(L); }
    if (g % 3 == 0) for (uint32_t e = 1; e < E; e += 3) { cam[e] = cam[e - 1]; lmk[e] = lmk[e - 1]; }
    if (g % 2 == 0) {               // sorted by (camera, landmark) like the shipped files
      std::vector<uint32_t> o(E);
      std::iota(o.begin(), o.end(), 0u);
      std::sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) { return cam[a] != cam[b] ? cam[a] < cam[b] : lmk[a] < lmk[b]; });
      std::vector<uint32_t> c2(E), l2(E);
      for (uint32_t e = 0; e < E; ++e) { c2[e] = cam[o[e]]; l2[e] = lmk[o[e]]; }
      cam.swap(c2); lmk.swap(l2);
    }
    if (run(cam, lmk, C, L, nullptr)) return 1;
    const uint32_t cut = rng.below(L + 1);
    const gbp_shard a{0, 3, 0, cut}, b{1, 3, cut, cut}, c{2, 3, cut, L};     // the middle shard is EMPTY
    if (run(cam, lmk, C, L, &a) || run(cam, lmk, C, L, &b) || run(cam, lmk, C, L, &c)) return 1;
  }
  {  // ONE camera; one landmark seen by every camera (slots beyond the index record)
    std::vector<uint32_t> cam(700, 0u), lmk(700);
    std::iota(lmk.begin(), lmk.end(), 0u);
    if (run(cam, lmk, 1, 700, nullptr)) return 1;
    std::vector<uint32_t> cam2(50), lmk2(50, 0u);
    std::iota(cam2.begin(), cam2.end(), 0u);
    if (run(cam2, lmk2, 50, 1, nullptr)) return 1;
  }
  {  // the config-5 shard shape, scaled: many cameras with ~100 factors each, >= 2 048 tiles: the product's thresholds apply
    const uint32_t C = 2048, L = 20000, obs = 10;
    std::vector<std::pair<uint32_t, uint32_t>> ed;
    for (uint32_t l = 0; l < L; ++l)
      for (uint32_t k = 0; k < obs; ++k) ed.emplace_back(rng.below(C), l);
    std::sort(ed.begin(), ed.end());
    std::vector<uint32_t> cam(ed.size()), lmk(ed.size());
    for (size_t e = 0; e < ed.size(); ++e) { cam[e] = ed[e].first; lmk[e] = ed[e].second; }
    gbp_problem pr{C, L, (uint32_t)cam.size(), cam.data(), lmk.data(), {1, 0, 0, 0, 1, 0, 0, 0, 1}};
    gbp::Layout y;
    std::string err;
    REQUIRE(gbp::layout_build(&pr, 0, nullptr, gbp::LayoutOptions(), y, err) == GBP_OK);
    REQUIRE(y.n_tiles >= 2048 && y.row_slot.size() == y.n_rows && y.row_window == 32 && y.tile_perm.size() == y.n_tiles);
    if (check(pr, nullptr, gbp::LayoutOptions(), 0, y)) return 1;
  }
  {  // refusals
    const uint32_t cam[2] = {0, 1}, lmk[2] = {0, 5};
    gbp_problem bad{2, 5, 2, cam, lmk, {0}};
    gbp::Layout y;
    std::string err;
    REQUIRE(gbp::layout_build(&bad, 0, nullptr, gbp::LayoutOptions(), y, err) == GBP_ERR_INVALID && !err.empty());
    REQUIRE(gbp::layout_build(nullptr, 0, nullptr, gbp::LayoutOptions(), y, err) == GBP_ERR_INVALID);
    const uint32_t lmk2[2] = {0, 1};
    gbp_problem ok{2, 5, 2, cam, lmk2, {0}};
    const gbp_shard s1{0, 2, 3, 9}, s2{2, 2, 0, 5};
    REQUIRE(gbp::layout_build(&ok, 0, &s1, gbp::LayoutOptions(), y, err) == GBP_ERR_INVALID);
    REQUIRE(gbp::layout_build(&ok, 0, &s2, gbp::LayoutOptions(), y, err) == GBP_ERR_INVALID);
    gbp::LayoutOptions o;
    o.classes = 0;
    REQUIRE(gbp::layout_build(&ok, 0, nullptr, o, y, err) == GBP_ERR_INVALID);
  }
  return 0;
}
