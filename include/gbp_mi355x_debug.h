/* gbp_mi355x_debug.h — TEST HOOKS of the GBP engine.  Not part of the product ABI: the symbols below exist only in
 * libgbp_mi355x_test.so (the product sources compiled with -DGBP_BUILD_TEST_HOOKS, `python -m gbp_poplar_amd.build`
 * builds it beside the product library) and are used by tests/ and profiles/ to look at internal state in the
 * reference's tensor layouts and to run the device math layer on caller-supplied vectors. */
#ifndef GBP_MI355X_DEBUG_H
#define GBP_MI355X_DEBUG_H

#include "gbp_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Raw internal state in the reference's tensor layouts (ba.cpp:665-687,759-775):
 *   what 0: a = factor_potentials_eta [9E],  b = factor_potentials_lambda [81E] = [cc36|cl18|lc18|ll9]
 *   what 1: a = cam message eta [6E],        b = cam message Lambda [36E] (lower triangle as stored; upper 0)
 *   what 2: a = lmk message eta [3E],        b = lmk message Lambda [9E]
 *   what 3: a = mu [9E],                     b = dmu [E]                                          */
GBP_API int gbp_debug_get(gbp_ctx* ctx, int what, float* a, float* b);
/* Timing experiment: average us per launch of an ablated sweep kernel (1 = no landmark-message
 * gather/scatter, 2 = no landmark-belief gather, 4 = no arithmetic, 8 = streaming landmark messages;
 * bits combine; ablations other than 0 and 100-102 exist in the experiments build only).  Leaves garbage in the ctx. */
GBP_API int gbp_debug_time_sweep(gbp_ctx* ctx, int ablation, int reps, double* avg_us);
/* Overwrite the factor potentials from reference-layout arrays (inverse of what 0). Test hook. */
GBP_API int gbp_debug_set_factor_potentials(gbp_ctx* ctx, const float* eta9E, const float* lambda81E);

/* The device math layer on caller-supplied vectors, one GPU lane per vector (no ctx): lets a test compare the HIP
 * routines directly with outputs of the reference's own matlib.cpp / bafuncs.cpp.  in/out are [n][width] fp32:
 *   op 0 inv3x3 9 -> 9 (matlib.cpp:143-161)        op 1 inv6x6 36 -> 36 (matlib.cpp:180-222)
 *   op 2 so3exp 3 -> 9 (bafuncs.cpp:31-55)         op 3 hfunc + Jac: cam6 lmk3 K9 -> hx2 Jkf12 Jlmk6 (bafuncs.cpp:82-213)
 *   op 4 P(6x3) += B(6x6) A(6x3), op 5 P(3x6) += A^T B: A18 B36 P18 -> 18;  op 6 P(6x6) += A A^T: A18 P36 -> 36
 *        (matMul and its transpose modes, matlib.cpp:47-89)
 *   op 7 inf2mean6x6: eta6 Lambda36 -> 6;  op 8 inf2mean3x3: eta3 Lambda9 -> 3 (bafuncs.cpp:2-15)
 *   op 9 inv6x6 again, but in the SUB-WAVE mapping: 16 lanes cooperate on one matrix (operands in LDS, lane = output
 *        element, reference order): 36 -> 36, bit-identical to op 1; exists to be measured against it
 *   op 10 div_shared: x9 m1 -> 9 quotients x[i] / m — the shared-reciprocal division of gbp_device_math.hpp, which must equal
 *        the IEEE fp32 division bit for bit (tests/test_gpu_device_math.py)                                             */
GBP_API int gbp_debug_math(int op, const float* in, float* out, int n);
/* same, then `reps` back-to-back launches timed with hipEvents: average microseconds per launch */
GBP_API int gbp_debug_math_timed(int op, const float* in, float* out, int n, int reps, double* avg_us);

/* ---- the device order without a device ---------------------------------------------------------------------------------
 * gbp_create first builds the DEVICE ORDER of the graph — which factor sits at which device position, where a camera's rows
 * and a landmark's message records are, in which order the sweep's wavefronts take the tiles — as pure host code
 * (csrc/gbp_layout.cpp; replaces the vertex-to-tile mapping of ba.cpp:243-366).  These hooks hand that construction out so that
 * CPU property tests (tests/test_layout.py; tests/sanitize/ under ASan + UBSan) can check it on any graph: no HIP call is made.
 * gbp_layout_options: the construction's knobs; the defaults ARE the product (gbp_debug_layout_default_options). */
typedef struct {
  uint32_t row_placement;      /* 1; 0 = rows always in camera-major order                                                     */
  uint32_t row_window;         /* 32: cameras whose rows are placed together                                                   */
  uint32_t row_place_max_deg;  /* 512: rows are placed by landmark class where a camera has fewer factors than this on average  */
  uint32_t row_key_lane;       /* 8: the factor of a row whose landmark classes the row (the middle one)                       */
  uint32_t classes;            /* 8: landmark classes of rows and tiles                                                        */
  uint32_t tile_window;        /* 96: look-ahead of the local tile permutation, in tiles                                        */
  uint32_t tile_min_tiles;     /* 2048: tile_order 0 permutes tiles (and places rows) only on graphs of at least this many      */
  uint32_t tile_identity;      /* 0; 1 = every tile in class 0                                                                 */
  uint32_t row_sort_in_class;  /* 0; 1 = the rows of a class (inside a window) ordered by their key landmark (measurement)      */
} gbp_layout_options;
typedef struct gbp_layout gbp_layout;
GBP_API void gbp_debug_layout_default_options(gbp_layout_options* opt);
/* the options LATER gbp_create calls of this process build their device order with (NULL = the defaults again); measurements */
GBP_API int gbp_debug_layout_options(const gbp_layout_options* opt);
/* the cache policy of the sweep's message streams (SweepArgs.policy bits: 1 = camera messages loaded cached, 2 / 4 = landmark
 * messages loaded / stored non-temporal) later gbp_create calls use instead of the choice by graph shape; -1 = by shape */
GBP_API int gbp_debug_force_sweep_policy(int policy);
/* the sweep of later gbp_create calls skips the all-pad 64-byte segments of its tiles (k_sweep<..., SEG>): -1 = by shape (the default:
 * graphs of >= 2 048 tiles where they are >= 1 % of the positions), 0 = never, 1 = always; identical results — A/B measurements and tests */
GBP_API int gbp_debug_force_seg_skip(int mode);
/* bursts without the metric on a graph that runs in the persistent kernel: 1 (default) = k_persist_flow (hand-offs through tagged
 * records, no device-wide barrier), 0 = k_persist<false> (counter barriers); identical results — A/B measurements and tests */
GBP_API int gbp_debug_persist_flow(gbp_ctx* ctx, int on);
/* ---- the tagged records of the persistent kernel ("a 16-byte aligned store of one lane is not observed torn") ----
 * gbp_debug_persist_verify: on != 0 — the following launches of k_persist_flow publish every record twice (the second copy with its
 * payload complemented, same tag) and a consumer accepts a record only when both copies carry the same tag; a payload mismatch is what
 * a torn or stale record would look like.  *mismatches (may be NULL) = the count since the last call (then reset).  Results are
 * identical to the plain kernel's.
 * gbp_debug_flow_torture: the detector under that assumption, no ctx involved — `rounds` rounds of K (4 or 16) 16-byte records per lane
 * exchanged with the product's own store / load instructions between partner workgroups bid and bid ^ mask (odd mask: another XCD) of
 * a grid of `blocks` (a power of two, 16 .. 256, >= 2 * mask), every word of a record a function of its round;
 * out[4] = {records seen torn, otherwise corrupt, waits that timed out, records checked}.  inject != 0: the control — one record in
 * 64 is stored tag first, payload a little later (what a tearing memory system would show): the detector must report torn records. */
GBP_API int gbp_debug_persist_verify(gbp_ctx* ctx, int on, uint64_t* mismatches);
GBP_API int gbp_debug_flow_torture(int blocks, int K, int rounds, unsigned mask, int inject, uint64_t* out /* [4] */);
/* the grid and the belief-phase roles of a launch of the persistent kernel for a graph of n_tiles sweep tiles (a multiple of 4), n_cams
 * cameras, n_lmks landmarks, without / with the metric after every iteration (host evaluation of the function the kernel uses; no GPU):
 * dims[4] = workgroups, roles separated from the tiles (0 / 1), metric roles, landmark groups;  role[4 * workgroups] (may be NULL): per
 * wave 4 * workgroup + wave-in-workgroup: v < n_cams camera v, < n_cams + groups landmark group, < 2 n_cams + groups metric mean, ~0 none */
GBP_API int gbp_debug_persist_roles(uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, int with_metric, uint32_t* dims, uint32_t* role, uint32_t cap);
GBP_API int gbp_debug_layout_build(const gbp_problem* problem, int tile_order, const gbp_shard* shard /*NULL = whole graph*/,
                           const gbp_layout_options* opt /*NULL = the process's current options*/, gbp_layout** out);
/* dims[11] = C, L, E, lmk_begin, lmk_end, L_loc, E_loc, n_rows, n_tiles, Ep, row_window */
GBP_API int gbp_debug_layout_dims(const gbp_layout* lay, uint32_t* dims);
/* which: 0 pos_edge [Ep] (~0 = pad), 1 pos_cam [Ep], 2 pos_lmk_loc [Ep], 3 pos_lpos [Ep], 4 cam_row_ptr [C+1], 5 row_slot
 * [n_rows] or empty, 6 row_cam [Ep/16], 7 lmk_ptr [L_loc+1], 8 lmk_fpos [E_loc], 9 lmk_ix [L_loc][16], 10 tile_perm [n_tiles] or
 * empty.  The pointer stays valid until gbp_debug_layout_free. */
GBP_API int gbp_debug_layout_array(const gbp_layout* lay, int which, const uint32_t** data, size_t* n);
GBP_API void gbp_debug_layout_free(gbp_layout* lay);
/* The local XCD-aware execution order of the sweep (gbp_params.tile_order = 3) as a pure function of the tiles' landmark
 * classes (0 .. n_classes - 1): perm[wave slot] = tile — a bijection that keeps every tile within `window` + 32 slots of its
 * sequential place. */
GBP_API int gbp_debug_tile_order_local(const uint8_t* tile_class /*[n_tiles]*/, uint32_t n_tiles, uint32_t window, uint32_t n_classes,
                               uint32_t* perm /*[n_tiles]*/);

#ifdef __cplusplus
}
#endif
#endif /* GBP_MI355X_DEBUG_H */
