// gbp_kernels.h — device data layout + kernel launchers shared by gbp_kernels.hip and the C-ABI translation units (gbp_api_*.cpp, gbp_ctx.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gbp {

// ---- HBM layout ---------------------------------------------------------------------------------
// Factors live in DEVICE ORDER: camera-major (a camera's local factors in file order), each camera
// padded to whole ROWS of 16 lanes, the total padded to whole TILES of 64 lanes (= one wavefront).
// Per-factor arrays are tile-coalesced AoSoA: float4 element (tile, group g, lane) sits at
//   base4[(tile * G + g) * 64 + lane]
// so that a wave's g-th access is one contiguous, 16-byte-per-lane, 1 KiB transaction.
constexpr int kTile = 64;
constexpr int kRow = 16;

// FAC: factor potential, 56 floats / 14 groups (symmetric blocks packed; lossless because the
// reference's Lambda_cc / Lambda_ll are bit-symmetric and Lambda_lc is a copy of Lambda_cl^T,
// gbp_codelets.cpp:158-162,363-367):
//   [0..8] eta (c0..5,l0..2)  [9..29] Lambda_cc lower  [30..47] Lambda_cl 6x3  [48..53] Lambda_ll lower
//   [54,55] measurement z
constexpr int kFacG = 14;
// CMSG: factor->camera message as the sweep re-reads it, 28 floats / 7 groups:
//   [0..5] eta  [6..26] Lambda lower triangle (only the lower triangle is ever read back:
//   inv6x6 reads A(i,j), i>=j, matlib.cpp:195-201; the full 6x6 goes into the row partial sums)
constexpr int kCmsgG = 7;
// MU: [0..8] mu (== oldmu between sweeps, ba.cpp:898)  [9] dmu
constexpr int kMuG = 3;
// Per-factor scalar state rides in the pad slots of the landmark-message record (below); flags:
constexpr uint32_t kFlagActive = 1u, kFlagRobust = 2u, kFlagPad = 4u;
// Indices: ROW_CAM[p/16] = camera of a row, LMK_IDX[p] = landmark (local index) of a factor.
//
// Landmark-side records are AoS of 16 floats (one 64-byte sector each):
//   LMSG[p] (factor->landmark message of the factor at DEVICE position p; a landmark's records are found
//   through LMK_FPOS[lmk_ptr[l] .. lmk_ptr[l+1]) in slot order), LMKB[l] / LMKP[l] beliefs / priors:
//   [0..2] eta  [3] pad  [4..12] Lambda 3x3  [13..15] pad
//   LMKB pads carry the hoisted dmu^2 pieces; LMSG pads carry the factor's scalar state:
//   [3] damping  [13] (damping_count << 3) | flags (int bits)  [14] measurement variance
constexpr int kLmkRec4 = 4;
// Camera-side records are 44 floats (11 float4):  [0..5] eta  [6,7] pad  [8..43] Lambda 6x6
//   CAMB[c] beliefs, CAMP[c] priors, ROWP[row] row partial sums, exchange buffers [rank][c]
constexpr int kCamRec4 = 11;
constexpr int kCamRec = 44;

struct Hyper {  // gbp_codelets.cpp:11-16
  float maxeta_damping;
  int num_undamped_iters;
  float dmu_threshold;
  int min_linear_iters;
  float nstds;
  int relin_mode;
};

// The metric riding in the two-kernel path (gbp_iterate_eval_each on graphs that do not run in k_persist): eval_reprojection_error
// (util.cpp:74-144) + the counters of ba.cpp:1011-1020 after EVERY iteration without a launch of their own and without the host.
//   k_beliefs<EV> of iteration k   also computes what k_means computes — the fp64 pivoted means of its new beliefs — and leaves
//                                  them as METRIC RECORDS: per camera the rotation and translation the residuals need (the rotation
//                                  evaluated once per camera instead of once per factor: same operations, same bits), per landmark
//                                  its mean; it counts the non-finite / non-PD beliefs and advances the iteration counter;
//   k_sweep<EV> of iteration k+1   holds every factor's state as sweep k left it and its measurement in registers anyway: two
//                                  L2-resident gathers more give the residual; one partial-sum record per wave goes to slot k of a
//                                  device ring (index = TILE, not wave slot: the order of the sums is a property of the graph);
//   k_eval_fold, once per burst    reduces the per-tile records of every slot in the order of every other metric (the lane tree of
//                                  a wave, ((w0 + w1) + w2) + w3 for the four tiles of a 256-factor block, blocks j, j + 1024,
//                                  j + 2048 ... serially = the <= 1 024 block sums k_eval produces, then those serially as the
//                                  host adds them) and writes ONE result per iteration to host-mapped memory.
// The slot comes from a counter in DEVICE memory, so the launches of a burst are the same for every iteration and replay from a
// hipGraph.  The metric of a piece's last iteration, which no sweep follows, comes from k_eval_ride (the same records from the same
// metric records, one wave per tile); nothing of a burst waits for the host.
struct EvalRec {           // partial sums of one tile (one wavefront of the sweep)
  double sum_norm, sum_half_sq;
  uint32_t n_active, n_relin, n_robust, pad;
};
struct EvalRide {
  float4* cam_rec;             // [C][3]   (R0 R1 R2 t0) (R3 R4 R5 t1) (R6 R7 R8 t2): eigenso3exp of the camera's metric mean + its translation
  float4* lmk_mean;            // [L]      metric mean of the landmark (x y z 0)
  EvalRec* part;               // [depth][n_tiles] ring of per-tile partial sums
  unsigned long long* health;  // [2] non-finite means / non-PD beliefs counted by the CURRENT belief update (the next sweep collects and zeroes them)
  unsigned long long* slot_health;  // [depth][2] what the sweep collected for each slot
  unsigned* counter;           // iterations of this burst completed so far (advanced by k_beliefs<EV>); sweep k + 1 fills slot counter - 1
  uint32_t n_tiles;
  int num_undamped;
};

struct SweepArgs {
  const uint32_t* row_cam;   // [Ep/16] camera of each 16-lane row
  const uint32_t* lmk_idx;   // [Ep]    landmark (local index) of each factor
  float4* fac;
  float4* cmsg;
  float4* mu;
  float4* lmsg;
  const float4* camb;
  const float4* lmkb;
  float4* rowp;
  const float4* cam_mu;   // [C][4]: hoisted camera means (current x2 float4, used-by-last-sweep x2)
  const float4* lmk_mu;   // [L][2]
  const float4* cam_lin;  // [C][5]: camera-only Jacobian terms of the current hoisted mean (R 9, N 9, |w|^2; gbp_device_math.hpp CamLin)
  float K[9];
  Hyper hp;
  int variant;               // experiments build only (gbp_params.reserved[0]): a mapping experiment instead of k_sweep
  const uint32_t* tile_perm; // [n_tiles] or NULL: wave slot (4 * block + wave) -> tile.  The XCD-aware execution order (gbp_layout.cpp):
                             // workgroups are dealt round-robin over the 8 XCDs, the table hands every XCD the tiles of one landmark
                             // class so that its private L2 holds that slice of the gathered landmark tables
  const uint32_t* seg_live;  // [n_tiles] or NULL: bit s = the 64-byte segment (lanes 4s .. 4s + 3) of the tile holds a factor; with it the sweep
                             // neither loads nor stores the all-pad segments (k_sweep<..., SEG>): graphs of many small cameras, where the unused
                             // tails of the cameras' last rows are a few per cent of all positions
  uint32_t policy;           // kPol* bits: cache policy of the two message streams, chosen per graph shape (gbp_api_ctx.cpp: sweep_policy_for)
  EvalRide ev;               // k_sweep<EV> only
};
// SweepArgs.policy.  The potentials (read once per sweep) and every tile STORE of the camera messages carry the non-temporal
// hint on every graph; what varies with the shape is how the two message streams, which a tile reads and rewrites in place, are
// loaded / stored:
constexpr uint32_t kPolCmsgLoadCached = 1u;   // camera-message tiles LOADED with the default policy (few cameras, both streams within the Infinity Cache)
constexpr uint32_t kPolLmsgLoadNt = 2u;       // landmark-message tiles loaded with the non-temporal hint (default: default policy — k_beliefs gathers them next)
constexpr uint32_t kPolLmsgStoreNt = 4u;      // ... and stored with it

struct BeliefArgs {
  // camera part
  const float* rowp; const uint32_t* cam_row_ptr; const float* cam_prior;
  const uint32_t* row_slot;  // [n_rows] device row of the camera-major row r, or NULL: rows sit in camera-major order (gbp_layout.cpp: row placement)
  float* cam_local;          // [C][44] local row sums (kept for prior-only refreshes / the exchange buffer)
  const float* gathered;     // != nullptr: belief = prior + sum_r gathered[r] ([world][C][44]) instead of the row sums
  int world;
  float* camb; float4* cam_mu; float4* cam_lin; uint32_t n_cams;
  // landmark part
  const float4* lmk_prior; const float4* lmsg; const uint32_t* lmk_ptr; const uint32_t* lmk_fpos;
  const uint32_t* lmk_ix;   // [L][16]: degree, device positions of slots 1..15 (one 64-B index record per landmark)
  float4* lmkb; float4* lmk_mu; uint32_t n_lmks;
  // control
  uint32_t cam_blocks;
  uint32_t lmk_blocks;       // landmark blocks of this launch
  int lmk_xcd_order;         // 1: blocks with equal (index mod 8) — one XCD — take a contiguous landmark range
  int partial_only;          // camera part writes cam_local only (multi-GPU: before the exchange)
  int hoist;                 // compute per-variable means + dmu^2 pieces
  int roll;                  // end of an iteration: "means used by the last sweep" := current means, then recompute
  // WEAKEN_PRIORS (ba.cpp:863-865) = WeakenPriorVertex on every variable + this belief refresh, in ONE launch: weaken != 0 makes
  // the owner of every prior element with flag in 1..5 scale it (written back through *_prior_rw) and count the flag down
  int weaken;
  float* cam_prior_rw; const float* cam_scale; uint32_t* cam_wflag;
  float4* lmk_prior_rw; const float* lmk_scale; uint32_t* lmk_wflag;
  EvalRide ev;               // k_beliefs<EV> only
};

// k_persist: n GBP iterations in ONE launch for graphs small enough that every workgroup is resident at once
// (the launch-bound configs: fr1xyz = 51 workgroups).  Wave w sweeps tile w (per-factor state stays in registers across
// iterations), a device-wide barrier, wave w then owns camera w or a group of 16 landmarks in the belief update
// (its index records and priors stay in registers), a second barrier.
struct DeviceEval;
// optional metric phases at the end of a k_persist launch (gbp_iterate_eval): what k_means + k_eval compute, same bits
struct PersistEval {
  int on;
  int each;                            // 0: one metric, after the last iteration;  1: after EVERY iteration (gbp_iterate_eval_each)
  uint32_t stride;                     // DeviceEval slots per metric: [0] = health copy, [1 + tile wave] = that wave's partial sums
  float* cam_mu;                       // [2][C][6] metric means (util.cpp:103-108) of iteration k in half k & 1, written by the camera / metric waves
  float* lmk_mu;                       // [2][L][3]
  int num_undamped;
  DeviceEval* slots;                   // [metrics][stride], host-mapped memory; metric k of the launch -> slots + k * stride
  unsigned long long* health;          // [2] non-finite means / non-PD beliefs of THIS evaluation (zero on entry); each == 0
  unsigned long long* health_each;     // [2][2] each == 1: the counters of iteration k in pair k & 1 (all zero on entry and on exit)
  unsigned long long* health_next;     // zeroed for the next evaluation
};
// k_persist_flow: the same launch without device-wide barriers.  Every array that crosses waves inside the launch has a TAGGED
// shadow: 16-byte records of three payload floats and the number of the iteration that produced them (a 16-byte aligned store of
// one lane is not torn), two halves alternating by the parity of the iteration.  A consumer re-loads its records until all of them
// carry the iteration it waits for; what an iteration waits for is exactly what it reads (profiles/exp_dataflow.hip: 6.9 -> 4.7 us
// per hand-off at 51 workgroups).  The shadows live only inside a launch: its prologue publishes the beliefs it starts from, its
// last iteration writes the ordinary arrays.
constexpr uint32_t kFlowRow4 = 15;    // float4 per 16-lane row: lane q of the row's last quad stores 4 (q = 3: 3) of them
constexpr uint32_t kFlowCam4 = 10;    // camera belief: eta 6, S, lower triangle of Lambda 21 (+ 2 unused slots)
constexpr uint32_t kFlowClin4 = 7;    // CAM_LIN: 20 floats
constexpr uint32_t kFlowLmk4 = 5;     // landmark belief: eta 3, Lambda 9, the three squared mean changes
struct PersistFlow {
  float4* lmsg;    // [2][Ep][4]         eta | Lambda rows 0, 1, 2 of the factor -> landmark message
  float4* rowp;    // [2][Ep / 16][15]   row sums of the factor -> camera messages
  float4* camb;    // [2][C][10]
  float4* cmu;     // [2][C][2]          hoisted camera mean
  float4* clin;    // [2][C][7]
  float4* lmkb;    // [2][L][5]
  float4* lmu;     // [2][L][1]          hoisted landmark mean
  float4* emc;     // [2][C][4]          metric mean of a camera: its rotation (9), its translation (3) (launches that carry the metric)
  float4* eml;     // [2][L][1]          ... of a landmark
  unsigned long long* health_iter;   // [kSeriesMax][2] non-finite means / non-PD beliefs of iteration k of the launch (zero between launches)
  unsigned tag0;   // tags of this launch: tag0 (what the prologue publishes), tag0 + 1 + it (what iteration it produces)
  // test-hooks builds (gbp_debug_persist_verify): mirror4 != 0 = every record is published a second time, payload complemented, this
  // many float4 further on (the allocation is twice the size), and a consumer accepts a record only when both copies agree; mismatches
  // are counted in *verify_errors.  0 in the product.
  uint32_t mirror4;
  unsigned long long* verify_errors;
};

// Which belief-phase role the wave (workgroup bid of nblk, wave wib of 4) of a k_persist_flow launch has, in k_persist's numbering:
// v < C camera v;  C <= v < C + G landmark group v - C;  C + G <= v < 2C + G metric mean of camera v - (C + G);  ~0u none.
// separate = 0: v = wib * nblk + bid (roles and tiles share waves).  separate = 1: the roles are dealt to the waves WITHOUT a tile first
// (whole workgroups behind the n_tiles / 4 that hold tiles, one role per workgroup before a second one): cameras, then n_met metric
// means, then landmark groups.  Host and device (gbp_debug_persist_roles evaluates it on the host: tests/test_persist_roles.py).
#if defined(__HIPCC__) || defined(__CUDACC__)
#define GBP_HD __host__ __device__ inline
#else
#define GBP_HD inline
#endif
GBP_HD uint32_t persist_role(uint32_t bid, uint32_t wib, uint32_t nblk, uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmk_groups,
                             uint32_t n_met, uint32_t separate) {
  if (!separate) return wib * nblk + bid;
  const uint32_t v_met0 = n_cams + n_lmk_groups;
  const uint32_t tw = n_tiles / 4u;                                     // workgroups that hold tiles (n_tiles is a multiple of 4)
  const uint32_t r = bid >= tw ? wib * (nblk - tw) + (bid - tw) : (nblk - tw) * 4u + (bid * 4u + wib);
  if (r < n_cams) return r;
  if (r < n_cams + n_met) return v_met0 + (r - n_cams);
  const uint32_t g = r - n_cams - n_met;
  return g < n_lmk_groups ? n_cams + g : ~0u;                           // no role left for this wave
}
// grid of a launch: workgroups, whether the roles are separated, how many metric roles (persist_blocks / launch_persist use it)
struct PersistGrid { uint32_t nb, separate, n_met; };
PersistGrid persist_grid(uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, bool with_metric);

struct PersistArgs {
  SweepArgs s;
  BeliefArgs b;
  PersistEval ev;
  PersistFlow f;           // f.lmsg != NULL: k_persist_flow (tagged records) instead of k_persist (barriers)
  uint32_t separate;       // k_persist_flow: 1 = cameras (and metric means) are owned by waves WITHOUT a tile (persist_blocks sized the grid for it)
  uint32_t n_met;          // ... and this many metric roles follow the camera roles (0 or C)
  // WEAKEN_PRIORS inside the launch (gbp_ba_loop): iteration `it` of the launch is loop index w_first + it of the reference's loop
  // (ba.cpp:1001-1008); WeakenPriorVertex runs in front of it iff (w_first + it + 1) % 2 == 0 and w_first + it < w_steps2 — for it >= 1
  // (a weakening in front of the launch's first iteration is the host's: an ordinary gbp_weaken_priors).  w_steps2 = 0: none.
  // Scalings, flags and the priors to write back: b.cam_scale / cam_wflag / cam_prior_rw and the lmk_ ones.
  uint32_t w_first, w_steps2;
  unsigned epoch_base;     // arrivals the barrier counter has already seen (launches of one ctx keep counting: no memset per launch)
  uint32_t n_tiles;        // sweep tiles = waves with a phase-A role
  uint32_t n_lmk_groups;   // ceil(L / 16): waves [C, C + n_lmk_groups) own 16 landmarks each in phase B
  int n_iters;
  uint32_t spread;         // the grid is `spread` times larger than the work and only every spread-th workgroup works (4: see launch_persist)
  uint32_t n_work_blocks;  // working workgroups (== gridDim.x unless spread)
  unsigned* sync;          // [kPersistSyncWords] barrier words: [0] arrival counter (monotonic over launches), [32] abort word
  unsigned* status;        // host-mapped: set to `seq` by the first workgroup that gives up at a barrier (a workgroup was not resident)
  unsigned seq;            // number of this launch in the ctx (>= 1): tells the host WHICH launch failed first (later ones return at once)
};
// snapshot / restore of the arrays a k_persist launch mutates (k_copy_segments)
constexpr int kMaxCopySegs = 16;
struct CopySegs {
  int n;
  const void* src[kMaxCopySegs];
  void* dst[kMaxCopySegs];
  size_t n4[kMaxCopySegs];     // float4 elements of each segment
};
constexpr int kSeriesMax = 512;         // metrics per launch of gbp_iterate_eval_each (longer bursts are split)
constexpr int kPersistSyncWords = 16 * 32;

struct DeviceEval {  // per-block partials, summed on the host in block order
  double sum_norm, sum_half_sq;
  unsigned long long n_active, n_relin, n_robust, pad;
};

void launch_sweep(const SweepArgs& a, uint32_t n_tiles, bool hoist, hipStream_t s, bool ev = false);
void launch_linearise(const SweepArgs& a, uint32_t n_tiles, hipStream_t s);
void launch_beliefs(BeliefArgs b, bool do_cam, bool do_lmk, hipStream_t s, bool ev = false);
// per-tile records of ring slots [0, n_slots) -> out[slot]: one gbp_eval_out-shaped result per slot (may be mapped host memory)
void launch_eval_fold(const EvalRide& ev, uint32_t n_slots, void* out, hipStream_t s);
// the riding metric of the CURRENT beliefs (a piece's last iteration: no sweep follows) into ring slot counter - 1
void launch_eval_ride(const EvalRide& ev, const uint32_t* row_cam, const uint32_t* lmk_idx, const float4* lmsg, const float4* fac, const float* K9_dev, hipStream_t s);
// workgroups of a k_persist launch for a graph; with_metric: + one wave per camera for the metric roles where the placement allows
uint32_t persist_blocks(uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, bool with_metric);
int persist_max_resident_blocks();                                            // how many of them this GPU keeps resident at once
// cooperative != 0: hipLaunchCooperativeKernel — the runtime refuses a grid that cannot be co-resident on the device and the
// driver never runs two cooperative grids (of any process) side by side; 0: plain launch (the creation-time probe vouches for
// the placement inside this process only).  Returns the launch status.
hipError_t launch_persist(PersistArgs a, bool cooperative, hipStream_t s);
void launch_copy_segments(const CopySegs& t, const unsigned* guard /* abort word or NULL */, hipStream_t s);
// runs the placement + barriers of k_persist for this graph once (blocking); false = the workgroups are not co-resident here
bool persist_probe(uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, unsigned* sync, unsigned* status_dev, volatile unsigned* status_host,
                   bool cooperative, hipStream_t s);
void launch_state_get(const float4* lmsg, float* damping, int* packed, uint32_t n, hipStream_t s);
void launch_state_set(float4* lmsg, const int* new_count, const uint32_t* ctl, uint32_t n, hipStream_t s);
// WRITE_PROG's per-factor streams (ba.cpp:868-886) from their compact host form — st[p] = {damping, count << 3 | flags, z0, z1}, var[p], both possibly
// host-mapped — into the records of the device order: the whole LMSG record of position p (zero messages + state) and the measurement
// slots of its FAC tile (the rest of FAC is zeroed by the caller)
void launch_upload_scatter(float4* lmsg, float4* fac, const float4* st, const float* var, uint32_t n, hipStream_t s);
void launch_means(const float4* camb, const float4* lmkb, float* cam_mu, float* lmk_mu, uint32_t n_cams,
                  uint32_t n_lmks, unsigned long long* health2 /* [0] non-finite means, [1] non-PD beliefs: zero on entry */,
                  unsigned long long* health2_next /* zeroed by this launch for the next evaluation */,
                  bool count_cams, hipStream_t s);
void launch_eval(const uint32_t* row_cam, const uint32_t* lmk_idx, const float4* lmsg, const float4* fac, const float* cam_mu,
                 const float* lmk_mu, const float* K9_dev, int num_undamped_iters, DeviceEval* partials /* may be mapped host memory */,
                 unsigned long long* health2, unsigned long long* health2_out, uint32_t n_tiles, hipStream_t s);
uint32_t eval_blocks(uint32_t n_tiles);
// experiments build (csrc/experiments/gbp_lab_kernels.hip): timing ablations of the sweep; false = unknown ablation
bool lab_launch_sweep_ablated(const SweepArgs& a, uint32_t n_tiles, int abl, hipStream_t s);
// test hook (hooks/gbp_flow_torture.hip): `rounds` rounds of K tagged 16-byte records per lane exchanged between partner workgroups
// bid and bid ^ mask; out[4] += {torn, corrupt, time-outs, records checked}; inject: the control (tag stored ahead of its record).  false: unsupported K
bool launch_flow_torture(float4* buf, unsigned long long* out, int blocks, int K, int rounds, unsigned mask, unsigned tag0, int inject, hipStream_t s);
bool debug_math_widths(int op, int* in_w, int* out_w);   // floats per vector of k_debug_math's op
void launch_debug_math(int op, const float* in, float* out, int n, hipStream_t s);  // test hook

}  // namespace gbp
