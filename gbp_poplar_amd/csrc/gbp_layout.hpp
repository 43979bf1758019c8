// gbp_layout.hpp — the DEVICE ORDER of a factor graph: which factor sits at which device position, where a camera's rows and a
// landmark's message records are, in which order the sweep's wavefronts take the tiles.
//
// Pure host code (no HIP call, no device needed): gbp_create builds a Layout first and only then touches the GPU, so the whole
// construction runs under the CPU sanitizers and in CPU property tests (tests/test_layout.py, tests/sanitize/).
//
// Replaces the vertex-to-tile mapping and the tensor-slice wiring of the reference (ba/ba.cpp:71-97, 243-366): where Poplar is
// told on which IPU tile a vertex runs and which slice of a message tensor it reads, this is told where a factor's lane is.
// The one thing the reference fixes and the layout must keep: the message-slot order of a variable is the FILE order of its
// incident factors (ba/ba.cpp:267-279).
#pragma once
#include "../../include/gbp_mi355x.h"

#include <cstdint>
#include <string>
#include <vector>

namespace gbp {

constexpr uint32_t kLayoutTile = 64;   // lanes of a tile (one wavefront); == kTile of gbp_kernels.h
constexpr uint32_t kLayoutRow = 16;    // lanes of a row (the unit of the camera sums); == kRow
constexpr uint32_t kLayoutBlock = 256; // factor positions are padded to whole workgroups of the sweep
constexpr uint32_t kNoEdge = ~0u;      // pos_edge of a padding lane

// Knobs of the construction.  The defaults ARE the product; other values exist for measurements and tests only and are reachable
// only through the test-hooks build (include/gbp_mi355x_debug.h: gbp_debug_layout_*).
struct LayoutOptions {
  uint32_t row_placement = 1;        // 0: rows always in camera-major order
  uint32_t row_window = 32;          // cameras whose rows are placed together (16 .. 48 measured equal, 64 .. 256 worse: profiles/HISTORY.md)
  uint32_t row_place_max_deg = 512;  // rows are placed by landmark class where a camera has fewer factors than this on average
  uint32_t row_key_lane = 8;         // the factor of a row whose landmark classes the row: the middle one (a short last row: its first).
                                     // Config-5 shard shape: 153.3 -> 150.8 us per iteration, 835 -> 822 MB per sweep against lane 0 (4, 12: between)
  uint32_t classes = 8;              // landmark classes of rows and tiles (8 = one per XCD; multiples of 8 give finer runs inside an XCD's share)
  uint32_t tile_window = 96;         // look-ahead of the local tile permutation, in tiles (32 / 48 / 192 measured: no difference)
  uint32_t tile_min_tiles = 2048;    // tile_order 0 permutes (and places rows) only on graphs of at least this many tiles
  uint32_t tile_identity = 0;        // 1: every tile in class 0 (what does the look-up itself cost?)
  uint32_t row_sort_in_class = 0;    // 1: the rows of a class (inside a window) ordered by their key landmark instead of camera-major
};

struct Layout {
  // sizes
  uint32_t C = 0, L = 0, E = 0;                       // the global problem
  uint32_t lmk_begin = 0, lmk_end = 0, L_loc = 0;     // this shard's landmark range
  uint32_t E_loc = 0;                                 // factors incident to it
  uint32_t n_rows = 0, n_tiles = 0, Ep = 0;           // rows of 16 in use, tiles of 64, padded factor positions (multiple of 256)
  uint32_t row_window = 0;                            // cameras per window of the row placement (0: camera-major rows)
  // per device position [Ep]
  std::vector<uint32_t> pos_edge;                     // file index of the factor, kNoEdge = pad
  std::vector<uint32_t> pos_cam;                      // camera of the position's row (pads of a used row: that camera; unused rows: 0)
  std::vector<uint32_t> pos_lmk_loc;                  // local landmark index (pads: 0)
  std::vector<uint32_t> pos_lpos;                     // landmark-major slot (lmk_ptr[l] + slot of the factor at l); pads: E_loc
  // cameras
  std::vector<uint32_t> cam_row_ptr;                  // [C + 1] a camera's rows, camera-major numbering
  std::vector<uint32_t> row_slot;                     // [n_rows] device row of camera-major row r; EMPTY = identity
  std::vector<uint32_t> row_cam;                      // [Ep / 16] camera of every device row
  // landmarks
  std::vector<uint32_t> lmk_ptr;                      // [L_loc + 1] slots of each local landmark
  std::vector<uint32_t> lmk_fpos;                     // [E_loc] landmark-major slot -> device position
  std::vector<uint32_t> lmk_ix;                       // [L_loc][16] degree, positions of slots 0 .. 14 (one 64-B record per landmark)
  // sweep order
  std::vector<uint32_t> tile_perm;                    // [n_tiles] wave slot -> tile; EMPTY = identity
};

// Builds the layout of `pr` for shard `sh` (NULL = the whole graph).  tile_order as gbp_params.tile_order.
// Returns GBP_OK or GBP_ERR_INVALID (+ text in err): null / empty problem, index out of range, bad shard, more than 2^32 positions.
int layout_build(const gbp_problem* pr, int tile_order, const gbp_shard* sh, const LayoutOptions& opt, Layout& out, std::string& err);

// The local XCD-aware execution order of the sweep (tile_order 3) as a pure function of the tiles' landmark classes:
// perm[wave slot] = tile.  Workgroup w = wave slots 4w .. 4w + 3 lands on XCD w mod 8 and is filled with the earliest not yet
// placed tiles of class (w mod n_classes), looking at most `window` tiles ahead of the oldest unplaced one (else: the oldest
// unplaced tile, whatever its class).  A bijection that keeps every tile within window + 32 slots of its sequential place.
void tile_order_local(const uint8_t* tile_class, uint32_t n_tiles, uint32_t window, uint32_t n_classes, uint32_t* perm);

}  // namespace gbp
