// gbp_layout.cpp — construction of the device order (see gbp_layout.hpp).  Pure host code, O(E).
#include "gbp_layout.hpp"
#include "gbp_threads.hpp"

#include <algorithm>

namespace gbp {

void tile_order_local(const uint8_t* tile_class, uint32_t n_tiles, uint32_t window, uint32_t n_classes, uint32_t* perm) {
  std::vector<uint8_t> placed(n_tiles, 0);
  std::vector<uint32_t> next_of_class(n_classes, 0);      // per class: where to continue looking
  uint32_t oldest = 0;
  for (uint32_t slot = 0; slot < n_tiles; ++slot) {
    while (oldest < n_tiles && placed[oldest]) ++oldest;
    const uint32_t want = (slot / 4) % n_classes;
    uint32_t& cur = next_of_class[want];
    if (cur < oldest) cur = oldest;
    while (cur < n_tiles && cur < oldest + window && (placed[cur] || tile_class[cur] != want)) ++cur;
    uint32_t pick = oldest;
    if (cur < n_tiles && cur < oldest + window && !placed[cur] && tile_class[cur] == want) pick = cur;
    perm[slot] = pick;
    placed[pick] = 1;
  }
}

namespace {

inline uint32_t lmk_class(uint32_t l_loc, uint32_t L_loc, uint32_t classes) {
  const uint64_t k = (uint64_t)l_loc * classes / std::max<uint32_t>(L_loc, 1u);
  return (uint32_t)std::min<uint64_t>(k, classes - 1);
}

// Row placement (graphs of many cameras with few factors each: BASELINE config 5 has 156 per camera and rank).
// A row — 16 consecutive factors of one camera, in file order: the unit of the camera sums, never split or reordered — may sit
// in any device row: the sweep writes its sums where the row sits, the camera part of k_beliefs finds a camera's rows through
// row_slot and adds them in the camera's own order (same sums, same bits).  The file lists a camera's factors by landmark, so a
// row covers 16 / deg of the landmark range, but the four rows of a TILE cover four times that: 41 % on config 5 — no tile
// belongs to one landmark class, the XCD-aware tile order has nothing to work with, and nearly every landmark-belief gather
// misses its L2.  Inside windows of `row_window` cameras the rows are therefore placed by the landmark class of their key factor
// (stable: camera-major order within a class), so that a tile holds rows of neighbouring cameras from ONE class.
void place_rows(const gbp_problem* pr, const LayoutOptions& opt, Layout& y, unsigned T, const std::vector<uint32_t>& cam_base /* [T][C]: a camera's factors in the chunks before chunk t */,
                const std::vector<uint32_t>& deg /* [C] */) {
  const uint32_t C = y.C, W = opt.row_window, K = opt.classes;
  std::vector<uint8_t> key(y.n_rows, 0);
  std::vector<uint32_t> key_lmk(opt.row_sort_in_class ? y.n_rows : 0, 0);
  {
    std::vector<uint32_t> fill_all(cam_base);      // (a copy: the positions are filled from the same starting counts afterwards)
    host::on_threads(T, [&](unsigned t) {
      uint32_t* cfill = fill_all.data() + (size_t)t * C;
      for (uint64_t e = (uint64_t)y.E * t / T, e1 = (uint64_t)y.E * (t + 1) / T; e < e1; ++e) {
        const uint32_t cam = pr->cam_id[e], l = pr->lmk_id[e];
        if (l < y.lmk_begin || l >= y.lmk_end) continue;
        const uint32_t i = cfill[cam]++;
        // ONE factor sets a row's key: the one in the key lane, or — a camera's short last row without that lane — the row's first
        // (what "the first factor sets it, the key lane overrides it" comes to, with one writer per row whatever the chunking)
        const uint32_t in_row = std::min<uint32_t>(kLayoutRow, deg[cam] - (i / kLayoutRow) * kLayoutRow);
        const uint32_t key_lane = opt.row_key_lane < in_row ? opt.row_key_lane : 0u;
        if (i % kLayoutRow == key_lane) {
          const uint32_t r = y.cam_row_ptr[cam] + i / kLayoutRow;
          key[r] = (uint8_t)lmk_class(l - y.lmk_begin, y.L_loc, K);
          if (opt.row_sort_in_class) key_lmk[r] = l - y.lmk_begin;
        }
      }
    });
  }
  y.row_slot.assign(y.n_rows, 0);
  std::vector<uint32_t> cnt(K + 1), order;
  for (uint32_t c0 = 0; c0 < C; c0 += W) {
    const uint32_t c1 = std::min<uint32_t>(C, c0 + W), R0 = y.cam_row_ptr[c0], R1 = y.cam_row_ptr[c1];
    if (opt.row_sort_in_class) {        // (measurement option) by class, then by key landmark, then camera-major
      order.resize(R1 - R0);
      for (uint32_t r = R0; r < R1; ++r) order[r - R0] = r;
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] < key[b] : key_lmk[a] < key_lmk[b]; });
      for (uint32_t i = 0; i < R1 - R0; ++i) y.row_slot[order[i]] = R0 + i;
      continue;
    }
    std::fill(cnt.begin(), cnt.end(), 0u);
    for (uint32_t r = R0; r < R1; ++r) cnt[key[r] + 1]++;
    for (uint32_t k = 0; k < K; ++k) cnt[k + 1] += cnt[k];
    for (uint32_t r = R0; r < R1; ++r) y.row_slot[r] = R0 + cnt[key[r]]++;      // counting sort: stable
  }
  y.row_window = W;
}

// tile_order 2 (kept for measurements): tiles ranked by the octile of their lowest landmark, the ranking cut into 8 equal runs,
// one per XCD (workgroup w -> XCD w mod 8).  Every XCD then has a stream front of its own: 11 % less fabric traffic but ~2 %
// slower than the sequential order on the 1M-factor graph.
void tile_perm_global(Layout& y) {
  const uint32_t nt = y.n_tiles, nb = nt / 4;
  std::vector<uint64_t> key(nt);
  for (uint32_t t = 0; t < nt; ++t) {
    uint32_t lo = ~0u;
    for (uint32_t i = 0; i < kLayoutTile; ++i) {
      const size_t p = (size_t)t * kLayoutTile + i;
      if (y.pos_edge[p] != kNoEdge) lo = std::min(lo, y.pos_lmk_loc[p]);
    }
    const uint64_t run = lo == ~0u ? 8u : lmk_class(lo, y.L_loc, 8u);
    key[t] = (run << 32) | t;   // memory order inside a run
  }
  std::sort(key.begin(), key.end());
  y.tile_perm.assign(nt, 0);
  const uint32_t q = nb / 8, r = nb % 8;   // XCD group g owns q + 1 workgroups if g < r, else q (bijective for any nb)
  uint32_t next = 0;
  for (uint32_t g = 0; g < 8; ++g) {
    const uint32_t n_wg = q + (g < r ? 1u : 0u);
    for (uint32_t k = 0; k < n_wg; ++k)
      for (uint32_t v = 0; v < 4; ++v) y.tile_perm[(size_t)(8 * k + g) * 4 + v] = (uint32_t)(key[next++] & 0xffffffffu);
  }
}

// tile_order 3 (and 0 on large graphs): tiles classed by the landmark class of their median factor, then tile_order_local.
// A camera's factors are sorted by landmark, so its tiles walk through the classes in order and the permutation only shuffles
// tiles of two or three neighbouring cameras: every stream keeps ONE compact front while each private L2 serves 1 / 8 of the
// gathered landmark table.
void tile_perm_local(const LayoutOptions& opt, Layout& y) {
  const uint32_t nt = y.n_tiles, K = opt.classes;
  std::vector<uint8_t> cls(nt, (uint8_t)K);            // a tile of pads only: a class nobody asks for (it is taken as "the oldest")
  const unsigned T = host::host_threads(nt, 1u << 12);
  host::on_threads(T, [&](unsigned th) {
    for (uint32_t t = (uint32_t)((uint64_t)nt * th / T), t1 = (uint32_t)((uint64_t)nt * (th + 1) / T); t < t1; ++t) {
      uint32_t l[kLayoutTile], n = 0;
      for (uint32_t i = 0; i < kLayoutTile; ++i) {
        const size_t p = (size_t)t * kLayoutTile + i;
        if (y.pos_edge[p] != kNoEdge) l[n++] = y.pos_lmk_loc[p];
      }
      if (opt.tile_identity) { cls[t] = 0; continue; }
      if (n) {
        std::nth_element(l, l + n / 2, l + n);
        cls[t] = (uint8_t)lmk_class(l[n / 2], y.L_loc, K);
      }
    }
  });
  y.tile_perm.assign(nt, 0);
  tile_order_local(cls.data(), nt, opt.tile_window, K, y.tile_perm.data());
}

}  // namespace

int layout_build(const gbp_problem* pr, int tile_order, const gbp_shard* sh, const LayoutOptions& opt, Layout& y, std::string& err) {
  auto bad = [&](const char* what) { err = what; return (int)GBP_ERR_INVALID; };
  if (!pr || !pr->cam_id || !pr->lmk_id || pr->n_cams == 0 || pr->n_lmks == 0 || pr->n_edges == 0) return bad("gbp_create: null or empty problem");
  if (opt.classes == 0 || opt.classes > 64 || opt.row_window == 0 || opt.tile_window == 0 || opt.row_key_lane >= kLayoutRow)
    return bad("gbp_create: bad layout options");
  y = Layout();
  y.C = pr->n_cams; y.L = pr->n_lmks; y.E = pr->n_edges;
  y.lmk_begin = sh ? sh->lmk_begin : 0;
  y.lmk_end = sh ? sh->lmk_end : y.L;
  if (sh && (sh->world < 1 || sh->rank < 0 || sh->rank >= sh->world)) return bad("gbp_create: bad shard");
  if (y.lmk_begin > y.lmk_end || y.lmk_end > y.L) return bad("gbp_create: bad shard");
  y.L_loc = y.lmk_end - y.lmk_begin;
  const uint32_t C = y.C, E = y.E;

  // ---- degrees; camera-major rows of 16, tiles of 64, whole workgroups of 256 ----
  // The file is cut into T chunks of consecutive factors (T = 1 below 2^19 factors; at most 8: T x (C + L) counters).  Every chunk counts
  // its factors per camera and per landmark; summed over the chunks in order these are the degrees, and the running sums in front of chunk t
  // are where chunk t continues a camera's rows and a landmark's slots — the FILE order of ba.cpp:267-279, whatever T is.
  const unsigned T = std::min(8u, host::host_threads(E, 1u << 18));
  const size_t Ll = y.L_loc;
  std::vector<uint32_t> cam_base((size_t)T * C, 0), lmk_base((size_t)T * Ll, 0);
  {
    std::vector<uint32_t> n_loc(T, 0);
    std::vector<int> out_of_range(T, 0);
    host::on_threads(T, [&](unsigned t) {
      uint32_t* cc = cam_base.data() + (size_t)t * C;
      uint32_t* lc = lmk_base.data() + (size_t)t * Ll;
      uint32_t n = 0;
      for (uint64_t e = (uint64_t)E * t / T, e1 = (uint64_t)E * (t + 1) / T; e < e1; ++e) {
        const uint32_t cam = pr->cam_id[e], l = pr->lmk_id[e];
        if (cam >= C || l >= y.L) { out_of_range[t] = 1; return; }
        if (l >= y.lmk_begin && l < y.lmk_end) { cc[cam]++; lc[l - y.lmk_begin]++; n++; }
      }
      n_loc[t] = n;
    });
    for (unsigned t = 0; t < T; ++t) {
      if (out_of_range[t]) return bad("gbp_create: index out of range");
      y.E_loc += n_loc[t];
    }
  }
  std::vector<uint32_t> deg(C, 0), ldeg(Ll, 0);
  host::on_threads(T, [&](unsigned th) {      // counts -> running sums in front of every chunk (exclusive, over the chunks), totals = degrees
    for (size_t k = (size_t)C * th / T, k1 = (size_t)C * (th + 1) / T; k < k1; ++k) {
      uint32_t run = 0;
      for (unsigned t = 0; t < T; ++t) { const uint32_t n = cam_base[(size_t)t * C + k]; cam_base[(size_t)t * C + k] = run; run += n; }
      deg[k] = run;
    }
    for (size_t l = Ll * th / T, l1 = Ll * (th + 1) / T; l < l1; ++l) {
      uint32_t run = 0;
      for (unsigned t = 0; t < T; ++t) { const uint32_t n = lmk_base[(size_t)t * Ll + l]; lmk_base[(size_t)t * Ll + l] = run; run += n; }
      ldeg[l] = run;
    }
  });
  {  // device positions are 32-bit
    uint64_t rows = 0;
    for (uint32_t k = 0; k < C; ++k) rows += (deg[k] + kLayoutRow - 1) / kLayoutRow;
    if (((rows * kLayoutRow + kLayoutBlock - 1) / kLayoutBlock) * kLayoutBlock >= (1ull << 32))
      return bad("gbp_create: more than 2^32 padded factor positions on one GPU; shard by landmark (gbp_shard)");
  }
  y.cam_row_ptr.assign(C + 1, 0);
  for (uint32_t k = 0; k < C; ++k) y.cam_row_ptr[k + 1] = y.cam_row_ptr[k] + (deg[k] + kLayoutRow - 1) / kLayoutRow;
  y.n_rows = y.cam_row_ptr[C];
  y.Ep = ((y.n_rows * kLayoutRow + kLayoutBlock - 1) / kLayoutBlock) * kLayoutBlock;
  if (y.Ep == 0) y.Ep = kLayoutBlock;
  y.n_tiles = y.Ep / kLayoutTile;
  y.lmk_ptr.assign(y.L_loc + 1, 0);
  for (uint32_t l = 0; l < y.L_loc; ++l) y.lmk_ptr[l + 1] = y.lmk_ptr[l] + ldeg[l];

  // ---- where the rows sit ----
  const bool order_on = tile_order == 3 || (tile_order == 0 && y.n_tiles >= opt.tile_min_tiles);
  if (opt.row_placement && (tile_order == 0 || tile_order == 3) && y.n_tiles >= opt.tile_min_tiles && C >= 2 * opt.row_window &&
      y.L_loc >= opt.classes && (uint64_t)y.E_loc < (uint64_t)C * opt.row_place_max_deg)
    place_rows(pr, opt, y, T, cam_base, deg);
  auto dev_row = [&](uint32_t r) -> uint32_t { return y.row_slot.empty() ? r : y.row_slot[r]; };

  // ---- factors into positions: a camera's factors in file order along its rows; a landmark's slots in file order ----
  y.pos_edge.assign(y.Ep, kNoEdge);
  y.pos_cam.assign(y.Ep, 0);
  y.pos_lmk_loc.assign(y.Ep, 0);
  y.pos_lpos.assign(y.Ep, y.E_loc);            // pads point behind the last slot
  host::on_threads(T, [&](unsigned th) {      // (a camera's rows are its own: every position has one writer)
    for (uint32_t k = (uint32_t)((uint64_t)C * th / T), k1 = (uint32_t)((uint64_t)C * (th + 1) / T); k < k1; ++k)
      for (uint32_t r = y.cam_row_ptr[k]; r < y.cam_row_ptr[k + 1]; ++r)
        for (uint32_t i = 0; i < kLayoutRow; ++i) y.pos_cam[(size_t)dev_row(r) * kLayoutRow + i] = k;
  });
  host::on_threads(T, [&](unsigned t) {       // chunk t continues every camera's rows and every landmark's slots where the chunks before it stop
    uint32_t* cfill = cam_base.data() + (size_t)t * C;
    uint32_t* lfill = lmk_base.data() + (size_t)t * Ll;
    for (uint64_t e = (uint64_t)E * t / T, e1 = (uint64_t)E * (t + 1) / T; e < e1; ++e) {
      const uint32_t cam = pr->cam_id[e], l = pr->lmk_id[e];
      if (l < y.lmk_begin || l >= y.lmk_end) continue;
      const uint32_t ll = l - y.lmk_begin;
      const uint32_t ci = cfill[cam]++;
      const uint32_t p = dev_row(y.cam_row_ptr[cam] + ci / kLayoutRow) * kLayoutRow + ci % kLayoutRow;
      y.pos_edge[p] = (uint32_t)e;
      y.pos_lmk_loc[p] = ll;
      y.pos_lpos[p] = y.lmk_ptr[ll] + lfill[ll]++;
    }
  });
  y.row_cam.resize(y.Ep / kLayoutRow);
  for (size_t r = 0; r < y.row_cam.size(); ++r) y.row_cam[r] = y.pos_cam[r * kLayoutRow];

  // ---- landmark side: slot list and the 64-B index record (degree + positions of the first 15 slots) ----
  y.lmk_fpos.assign(y.E_loc, 0u);
  host::on_threads(T, [&](unsigned th) {      // (pos_lpos is a bijection onto the slots)
    for (size_t p = (size_t)y.Ep * th / T, p1 = (size_t)y.Ep * (th + 1) / T; p < p1; ++p)
      if (y.pos_edge[p] != kNoEdge) y.lmk_fpos[y.pos_lpos[p]] = (uint32_t)p;
  });
  y.lmk_ix.assign((size_t)y.L_loc * 16, 0u);
  host::on_threads(T, [&](unsigned th) {
    for (uint32_t l = (uint32_t)((uint64_t)Ll * th / T), l1 = (uint32_t)((uint64_t)Ll * (th + 1) / T); l < l1; ++l) {
      const uint32_t s0 = y.lmk_ptr[l], d = y.lmk_ptr[l + 1] - s0;
      y.lmk_ix[(size_t)l * 16] = d;
      for (uint32_t k = 0; k < d && k < 15u; ++k) y.lmk_ix[(size_t)l * 16 + 1 + k] = y.lmk_fpos[s0 + k];
    }
  });

  // ---- execution order of the sweep's tiles ----
  if (tile_order == 2) tile_perm_global(y);
  else if (order_on) tile_perm_local(opt, y);
  return GBP_OK;
}

}  // namespace gbp
