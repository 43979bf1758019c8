/* gbp_mi355x.h — C-ABI of the MI355X-native GBP bundle-adjustment engine.
 *
 * Drop-in boundary for ONE path of joeaortiz/gbp-poplar: the synchronous GBP sweep that the
 * reference expresses as a Poplar program list + named host data streams
 * (reference ba/ba.cpp:925-934 `progs = {WRITE, LINEARISE, GBP, WEAKEN_PRIORS, READ}`,
 * ba/slam.cpp:937-948 adds READ_PRIORS and NEW_KEYFRAME).  The reference has no FFI; each
 * entry point below replaces one `engine.run(<PROG>)` call (+ its connected streams) and cites it.
 *
 * This header is the whole single-GPU boundary: the program list, the loop body (gbp_ba_loop), the metric, timing and the host
 * helpers of the path's callers.  gbp_mi355x_multi.h adds the landmark-sharded (multi-GPU) entry points, gbp_mi355x_compat.h keeps
 * earlier forms of the loop for hosts written against ABI 2-5, gbp_mi355x_debug.h the test hooks (test builds of the library only).
 *
 * Conventions: plain C, no exceptions across the ABI, every function returns 0 on success and a
 * negative gbp_status otherwise (text via gbp_last_error).  All pointers are HOST memory owned by
 * the caller unless the name ends in `_dev`.  A ctx is not thread-safe; calls are blocking unless
 * stated.  All floating point is IEEE fp32, integers are 32-bit, layouts are the reference's
 * row-major AoS host layouts (ba/ba.cpp:690-713,778-790).  All device work of a ctx is ordered on ITS stream (its own, or the one handed
 * over with gbp_set_stream), copies to and from the host included: nothing goes through the NULL stream, so work a caller queues on other
 * streams is the caller's to synchronise with.
 */
#ifndef GBP_MI355X_H
#define GBP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: GBP_API marks what it exports (nm -D shows these functions and nothing else). */
#if defined(__GNUC__) || defined(__clang__)
#define GBP_API __attribute__((visibility("default")))
#else
#define GBP_API
#endif

/* 2: gbp_timing_out gained exchange_ms, gbp_status gained GBP_ERR_COMM, the gbp_debug_* test hooks moved to
 *    gbp_mi355x_debug.h / libgbp_mi355x_test.so; gbp_params.persistent, gbp_iterate_eval.
 * 3: gbp_iterate_eval_each.
 * 4: gbp_params.persist_coop (was reserved[1]); a barrier time-out of the persistent kernel is recovered inside the library
 *    (state restored, burst replayed on the two-kernel path, GBP_OK + a warning in gbp_last_error) instead of GBP_ERR_HIP.
 * 5: the library exports exactly the gbp_* functions declared here (hidden visibility for everything else); the pipelined
 *    exchange (gbp_set_exchange_chunks / gbp_iterate_begin_chunk: measured slower than the plain one in every configuration) is
 *    gone, gbp_tile_order_local moved to the test-hooks header; gbp_iterate_eval_each keeps the metric on the device on graphs
 *    of any size.
 * 6: gbp_ba_loop (the body of the reference's loop, prior weakening included, in one call); graphs of up to 256 workgroups run in the
 *    persistent kernel, which hands over through tagged records instead of device-wide barriers.
 * Callers compare gbp_abi_version() with the header they were built against. */
#define GBP_ABI_VERSION 6

typedef enum {
  GBP_OK = 0,
  GBP_ERR_INVALID = -1,    /* bad argument / inconsistent sizes            */
  GBP_ERR_NO_DEVICE = -2,  /* no gfx950 device (ref: ba.cpp:652-655 exit(-1)) */
  GBP_ERR_HIP = -3,        /* a HIP runtime call failed                    */
  GBP_ERR_STATE = -4,      /* call order violated (e.g. iterate before upload) */
  GBP_ERR_IO = -5,         /* file could not be read (ref: ba.cpp:484-487)  */
  GBP_ERR_NOMEM = -6,      /* host allocation failed (no C++ exception ever crosses this ABI) */
  GBP_ERR_COMM = -7        /* the exchange between ranks failed (RCCL error, a rank died / timed out)  */
} gbp_status;

typedef struct gbp_ctx gbp_ctx;

/* Graph structure: what ba.cpp:503-512 extracts from the BAL file and wires into vertices
 * (ba.cpp:71-97,243-366).  Factor e joins camera cam_id[e] and landmark lmk_id[e]; the message
 * slot order of a variable is the FILE order of its incident factors (ba.cpp:267-279). */
typedef struct {
  uint32_t n_cams;          /* C  */
  uint32_t n_lmks;          /* L  */
  uint32_t n_edges;         /* E  */
  const uint32_t* cam_id;   /* [E] */
  const uint32_t* lmk_id;   /* [E] */
  float K[9];               /* row-major pin-hole {fx,0,cx,0,fy,cy,0,0,1}; the reference replicates
                               this one matrix per edge (ba.cpp:494-501), we keep one copy        */
} gbp_problem;

/* Hyper-parameters: compile-time globals of the reference (gbp_codelets.cpp:11-16). */
typedef struct {
  float   maxeta_damping;      /* 0.4   */
  int32_t num_undamped_iters;  /* 8     */
  float   dmu_threshold;       /* 3e-3  */
  int32_t min_linear_iters;    /* 10    */
  float   nstds;               /* 2.5   */
  int32_t relin_mode;          /* 0 = accumulate onto the old potential (faithful: matMul is `+=`
                                  and PrepMessageVertex does not zero, gbp_codelets.cpp:285-336);
                                  1 = reset (zero first)                                          */
  int32_t graph_unroll;        /* GBP iterations captured per hipGraph (>=1); 0 = library default (10 on a single-GPU ctx;
                                  direct launches on a sharded ctx, where the graph is slower); < 0 = never capture.
                                  A replay costs 10-20 us of launch work: on a 1M-factor graph 20 per graph are 1 % faster
                                  than 10 (bench.py passes 20), calls of fewer than graph_unroll iterations launch directly */
  int32_t per_factor_mu;       /* 0 (default): belief means are computed once per variable (bit-identical
                                  to the per-factor recomputation of gbp_codelets.cpp:264-277, requires the
                                  uploaded mu/oldmu to be zero as in ba.cpp:582-583); 1: keep the literal
                                  per-factor mu/oldmu tensors                                            */
  int32_t tile_order;          /* XCD-aware execution order (results are identical in every mode):
                                  0 (default) = the landmark blocks of the belief kernel that share an XCD take one contiguous
                                      landmark range (both 64-B halves of a 128-B message line then meet in one L2); on graphs of
                                      2 048 tiles (131 072 factor positions) or more the sweep runs in order 3, below in device order;
                                  1 = everything sequential;
                                  2 = as 0, and each XCD sweeps the tiles of one landmark range (global permutation: every XCD
                                      gets a stream front of its own; kept for measurements);
                                  3 = as 0, and the sweep's tiles are permuted LOCALLY (within a few cameras) so that workgroup w —
                                      which lands on XCD w mod 8 — holds factors of landmark octile w mod 8: each private L2 then
                                      serves 1/8 of the gathered landmark-belief table while every stream keeps one compact front
                                      (S1: 689 -> 614 MB per sweep, 7 920 -> 8 200 iterations/s, profiles/r04_tile_order.md).
                                      Where order 3 applies and cameras are small (fewer than 512 factors per camera and rank on
                                      average: BASELINE config 5) the 16-factor ROWS of 32 neighbouring cameras are laid out by
                                      landmark octile as well (a row stays whole, a camera's rows are added in the camera's own
                                      order: same sums; +2.4 % on the config-5 shard shape)                                                */
  int32_t persistent;          /* gbp_iterate(n >= 2) on a graph small enough that all of its workgroups are resident at once
                                  (BASELINE configs 1-3) runs the n iterations inside ONE kernel launch (k_persist_flow: per-factor
                                  state in registers, hand-offs through tagged records instead of kernel boundaries; identical results):
                                  0 (default) = automatically up to 256 workgroups (65 536 factor positions; all shipped sequences need <= 61), 1 = whenever the graph is
                                  co-resident, -1 = never.  Single-GPU ctx with hoisted means only.
                                  The waits inside that kernel need all of its workgroups resident at once; two things
                                  stand behind that (persist_coop below) and a third behind both: a wait that lasts longer than
                                  1.5 s gives up, every later launch of the ctx returns at once, and the library — at the next call
                                  that synchronises anyway — restores the snapshot taken before the failed launch, replays the
                                  affected bursts on the two-kernel path (identical results), returns GBP_OK and leaves a warning in
                                  gbp_last_error; the ctx then stays on the two-kernel path until the next gbp_upload. */
  int32_t reserved[1];
  int32_t persist_coop;        /* how the persistent kernel is launched: 0 (default) = plain launch, vouched for by a creation-time probe
                                  of the placement (the launches of one process are serialised by the library; a second process that
                                  takes CUs away is caught by the time-out + recovery above);  1 = hipLaunchCooperativeKernel, or the
                                  two-kernel path where the device refuses it: the runtime rejects a grid that cannot be co-resident
                                  and the driver never runs two cooperative grids side by side, also not those of two PROCESSES — a
                                  guarantee instead of a way back, for 30-60 us more per launch (measured, profiles/r04_persist_launch.md).
                                  Not capturable either way: while the ctx's stream is being captured gbp_iterate uses the two-kernel path. */
} gbp_params;

/* Landmark shard of a multi-GPU run (one process per GPU; gbp_mi355x_multi.h).  The global problem is passed to
 * gbp_create on every rank; a rank owns landmarks [lmk_begin, lmk_end) and every factor incident
 * to them.  Cameras are replicated.  NULL shard == {0,1,0,L}.  Replaces `--ipus N`
 * (ba.cpp:414-417,617-623). */
typedef struct {
  int32_t  rank, world;
  uint32_t lmk_begin, lmk_end;
} gbp_shard;

/* WRITE_PROG streams (ba.cpp:868-886).  cams_dofs/lmk_dofs (always 6/3, ba.cpp:577-578) are not
 * passed.  Any pointer may be NULL = "all zeros" (what the reference uploads for damping, mu,
 * oldmu: ba.cpp:580-584), except priors/measurements/meas_variances/active_flag. */
typedef struct {
  const float*    damping;            /* [E]   */
  const int32_t*  damping_count;      /* [E]   */
  const float*    mu;                 /* [9E]  */
  const float*    oldmu;              /* [9E]  */
  const uint32_t* active_flag;        /* [E]   */
  const float*    cam_scaling;        /* [C]   */
  const float*    lmk_scaling;        /* [L]   */
  const uint32_t* cam_weaken_flag;    /* [C]   */
  const uint32_t* lmk_weaken_flag;    /* [L]   */
  const float*    cam_priors_eta;     /* [6C]  */
  const float*    cam_priors_lambda;  /* [36C] */
  const float*    lmk_priors_eta;     /* [3L]  */
  const float*    lmk_priors_lambda;  /* [9L]  */
  const float*    measurements;       /* [2E]  */
  const float*    meas_variances;     /* [E]   */
} gbp_state_in;

/* READ_PROG streams (ba.cpp:908-916).  NULL members are skipped.  In a sharded ctx only owned
 * landmarks / factors are written, camera arrays are complete on every rank. */
typedef struct {
  float*    cam_beliefs_eta;     /* [6C]  */
  float*    cam_beliefs_lambda;  /* [36C] */
  float*    lmk_beliefs_eta;     /* [3L]  */
  float*    lmk_beliefs_lambda;  /* [9L]  */
  float*    damping;             /* [E]   */
  int32_t*  damping_count;       /* [E]   */
  uint32_t* robust_flag;         /* [E]   */
} gbp_state_out;

/* READ_PRIORS streams (slam.cpp:913-917): message slot 0 of every variable. */
typedef struct {
  float* cam_priors_eta;     /* [6C]  */
  float* cam_priors_lambda;  /* [36C] */
  float* lmk_priors_eta;     /* [3L]  */
  float* lmk_priors_lambda;  /* [9L]  */
} gbp_priors_out;

/* NEW_KEYFRAME streams (slam.cpp:919-928). */
typedef struct {
  const int32_t*  damping_count;      /* [E]   */
  const float*    cam_priors_eta;     /* [6C]  */
  const float*    cam_priors_lambda;  /* [36C] */
  const float*    lmk_priors_eta;     /* [3L]  */
  const float*    lmk_priors_lambda;  /* [9L]  */
  const uint32_t* active_flag;        /* [E]   */
  const uint32_t* cam_weaken_flag;    /* [C]   */
  const uint32_t* lmk_weaken_flag;    /* [L]   */
} gbp_kf_update;

/* Result of gbp_eval: util.cpp:74-144 + the two counters of ba.cpp:1011-1020.
 * Sums are raw (not yet divided) so that shards can be added. */
typedef struct {
  double   sum_norm;     /* sum_e ||z - pi(x)||_2  over active edges  (reproj[0] * n_active)   */
  double   sum_half_sq;  /* sum_e 0.5*||r||^2                         (reproj[1], "Cost")      */
  uint64_t n_active;
  uint64_t n_relin;      /* #(damping_count == -num_undamped_iters)   ba.cpp:1016-1020         */
  uint64_t n_robust;     /* sum robust_flag                            ba.cpp:1013-1015        */
  uint64_t n_nonfinite;  /* variables whose belief mean is non-finite (replaces Poplar FP traps, ba.cpp:888-891) */
  uint64_t n_nonpd;      /* variables whose belief Lambda has a non-positive un-pivoted LDL^T pivot: what
                            inv6x6 / inv3x3 (matlib.cpp:143-222) silently accept; early warning of divergence */
} gbp_eval_out;

typedef struct {
  double   sweep_ms, belief_ms, total_ms;   /* accumulated hipEvent times of timed iterations */
  uint64_t iterations;
  uint64_t algorithmic_bytes_per_iter;      /* 1112*E + 336*C + 96*L               */
  uint64_t device_bytes_allocated;
  double   exchange_ms;                     /* sharded ctx with a communicator, profiling on: accumulated time of the
                                               camera side of the exchange (local partial sums + all-gather)        */
} gbp_timing_out;

/* ---- life cycle: graph build + Engine ctor/load (ba.cpp:659-937) -------------------------- */
GBP_API int  gbp_abi_version(void);
GBP_API void gbp_default_params(gbp_params* p);
GBP_API int  gbp_create(const gbp_problem* problem, const gbp_params* params /*NULL=defaults*/,
                const gbp_shard* shard /*NULL=single GPU*/, gbp_ctx** out);
GBP_API void gbp_destroy(gbp_ctx* ctx);
GBP_API const char* gbp_last_error(const gbp_ctx* ctx /*NULL = last create error*/);
/* (a call that returned GBP_OK may leave a line that starts with "warning:" — a recovered incident — or, after gbp_create, with "info:":
 *  where the call spent its time: device order, runtime, allocation, layout upload, persistent-kernel set-up; the CLIs' --profile quotes it) */

/* ---- the program list ------------------------------------------------------------------- */
GBP_API int gbp_upload(gbp_ctx* ctx, const gbp_state_in* in);          /* WRITE_PROG      ba.cpp:868-886  */
GBP_API int gbp_linearise(gbp_ctx* ctx);                               /* LINEARISE_PROG  ba.cpp:890-893  */
GBP_API int gbp_iterate(gbp_ctx* ctx, int n_iters);                    /* GBP_PROG x n    ba.cpp:895-905  */
GBP_API int gbp_prepare(gbp_ctx* ctx);                                 /* optional: pay the one-off costs of gbp_iterate(n >= graph_unroll)
                                                                  now (hipGraph capture + instantiation + upload; runs nothing) —
                                                                  what Engine::load does for a Poplar program, ba.cpp:936-937 */
GBP_API int gbp_weaken_priors(gbp_ctx* ctx);                           /* WEAKEN_PRIORS   ba.cpp:863-865  */
GBP_API int gbp_read(gbp_ctx* ctx, gbp_state_out* out);                /* READ_PROG       ba.cpp:908-916  */
GBP_API int gbp_read_priors(gbp_ctx* ctx, gbp_priors_out* out);        /* READ_PRIORS     slam.cpp:913-917 */
GBP_API int gbp_new_keyframe(gbp_ctx* ctx, const gbp_kf_update* upd);  /* NEW_KEYFRAME    slam.cpp:919-928 */
GBP_API int gbp_eval(gbp_ctx* ctx, gbp_eval_out* out);                 /* util.cpp:74-144 on device (local shard) */
/* n passes of the BODY of the reference's iteration loop (ba.cpp:1001-1028, slam.cpp:1048-1103), blocking, from loop index iter0:
 *   if ((i + 1) % 2 == 0 && i < 2 * steps) WEAKEN_PRIORS;  GBP_PROG;  out[i - iter0] = the metric      for i = iter0 .. iter0 + n - 1
 * (steps = the reference's --steps; steps = 0: n iterations with the metric after every one).  Exactly the calls it stands for
 * (gbp_weaken_priors, gbp_iterate(1), gbp_eval: identical results);
 * on a graph that runs in the persistent kernel the passes are ONE launch however many weakenings lie between them (the kernel applies
 * WeakenPriorVertex itself in front of the iterations the loop weakens before) — the ten short launches of a run's, or a SLAM
 * keyframe's, weakening phase become one.  No evaluation may be in flight.
 * out == NULL: the passes WITHOUT the metric, not blocking (like gbp_iterate): a weakening then rides in the launch of the persistent
 * kernel or, on the two-kernel path, in the belief update of the iteration in front of it (one launch instead of two). */
GBP_API int gbp_ba_loop(gbp_ctx* ctx, int n_passes, unsigned iter0, unsigned steps, gbp_eval_out* out /* [n_passes] */);
GBP_API int gbp_sync(gbp_ctx* ctx);                                    /* wait for queued device work     */
GBP_API int gbp_timing(gbp_ctx* ctx, gbp_timing_out* out, int reset);  /* ba.cpp:980,1056-1058            */
/* which path gbp_iterate / gbp_ba_loop take on this ctx: 2 = bursts run inside the persistent kernel (small graph), 1 = a captured hipGraph is
 * replayed, 0 = not captured (yet), -1 = capture failed: direct launches */
GBP_API int gbp_graph_state(const gbp_ctx* ctx);

/* ---- measurement ---------------------------------------------------------------------------- */
/* per_stage_events != 0: gbp_iterate launches kernels directly with a hipEvent pair around the
 * sweep and the belief kernels of every iteration (feeds gbp_timing.sweep_ms / belief_ms); in the
 * split-phase path gbp_iterate_begin brackets its sweep launch the same way (sweep_ms and iterations
 * only; the pairs are read by the next gbp_timing call). */
GBP_API int gbp_set_profiling(gbp_ctx* ctx, int per_stage_events);

/* ---- host-side helpers of the path's callers (pure CPU, no device needed) ---------------- */
/* BALProblem::LoadFile (dataio.cpp:17-57).  Two-call pattern: pass NULL arrays to get sizes.
 * Files above 1 MB are converted by several host threads (at most 32; GBP_HOST_THREADS=n overrides) with the conversions fscanf makes:
 * the values do not depend on the thread count.  A file that is not a plain stream of whole numeric tokens is read by the fscanf chain itself. */
typedef struct {
  uint32_t n_cams, n_lmks, n_edges;
  double   fx, fy, cx, cy;
  uint32_t* cam_id;        /* [E]  caller-allocated */
  uint32_t* lmk_id;        /* [E]  */
  double*   observations;  /* [2E] */
  double*   cameras;       /* [6C] */
  double*   points;        /* [3L] */
} gbp_bal;
GBP_API int gbp_bal_read_header(const char* path, gbp_bal* hdr);
GBP_API int gbp_bal_read(const char* path, gbp_bal* bal);
GBP_API int gbp_bal_write(const char* path, const gbp_bal* bal);
/* Import of a standard "Bundle Adjustment in the Large" text file (C L E; E x "cam point x y"; 9 values per
 * camera: Rodrigues R, t, f, k1, k2; 3 per point; camera looks down -z, image origin at the centre, y up) into
 * the reference's conventions (sequences/README.md:5-16: one shared pin-hole K, cameras [t_cw, w_cw] looking
 * down +z): frame flipped by diag(1,-1,-1), observations undistorted and rescaled to the mean focal length
 * (fx = fy = mean f, cx = cy = 0), edges sorted by (camera, landmark) as util.cpp:95-99 / dataio.cpp:483-486
 * assume.  Same two-call pattern as gbp_bal_read; the result can be written with gbp_bal_write and fed to
 * ./ba or ./slam.  Not in the reference. */
GBP_API int gbp_bal_import_standard_header(const char* path, gbp_bal* hdr);
GBP_API int gbp_bal_import_standard(const char* path, gbp_bal* bal);

/* set_prior_lambda (dataio.cpp:67-117 + util.cpp:48-72), O(E).  cam_file/lmk_file are the FILE
 * values cast to float (the linearisation point of the prior strength), *_mean the (possibly
 * noised) prior means. */
GBP_API int gbp_set_prior_lambda(const gbp_problem* problem, float reproj_meas_var,
                         const float* cam_file /*[6C]*/, const float* lmk_file /*[3L]*/,
                         const float* cam_mean /*[6C]*/, const float* lmk_mean /*[3L]*/,
                         float* cam_priors_eta, float* cam_priors_lambda,
                         float* lmk_priors_eta, float* lmk_priors_lambda);
/* Prior-weakening scale factors (ba.cpp:561-572). */
GBP_API int gbp_prior_scalings(uint32_t n_cams, uint32_t n_lmks, const float* cam_priors_lambda,
                       float steps, float prior_std_weaker_factor, float first_cam_prior_std,
                       float* cam_scaling, float* lmk_scaling);
/* Initialisation options of ba.cpp:536-548.  gbp_init_add_noise = add_cam_trans_noise (--tn, metres), add_cam_rot_noise
 * (--rn, degrees about a random axis, camera centre kept) and add_lmk_noise (--ltn), dataio.cpp:330-415, on the prior
 * MEANS in place; cameras 0 and 1 stay exact (dataio.h:114-119).  The reference seeds from the clock; here the seed is
 * explicit (`--seed`), zero stds draw nothing.  gbp_init_av_depth = av_depth_init (--avdepth_on,
 * dataio.cpp:417-453): every landmark is placed one unit in front of the lowest-indexed camera observing it. */
GBP_API int gbp_init_add_noise(uint32_t n_cams, uint32_t n_lmks, float trans_std, float rot_std_deg, float lmk_std,
                       uint64_t seed, float* cam_mean /*[6C] in/out*/, float* lmk_mean /*[3L] in/out*/);
GBP_API int gbp_init_av_depth(const gbp_problem* problem, const float* cam_mean /*[6C]*/, float* lmk_mean /*[3L] out*/);
/* SLAM flag bookkeeping (dataio.cpp:455-475, 477-508).  update returns n_new_lmks via out. */
GBP_API int gbp_slam_create_flags(const gbp_problem* problem, uint32_t steps, uint32_t* active_flag,
                          uint32_t* cam_weaken_flag, uint32_t* lmk_weaken_flag,
                          uint32_t* lmk_active_flag);
GBP_API int gbp_slam_update_flags(const gbp_problem* problem, uint32_t steps, uint32_t data_counter,
                          uint32_t* active_flag, uint32_t* lmk_weaken_flag,
                          uint32_t* cam_weaken_flag, uint32_t* lmk_active_flag,
                          int32_t* n_new_lmks);
/* initialise_new_kf (util.cpp:183-223): prior eta of camera data_counter+1 from the belief mean of
 * camera data_counter.  The new-landmark branch is dead in the reference (out-of-bounds index,
 * util.cpp:215) and is not reproduced. */
GBP_API int gbp_slam_initialise_new_kf(uint32_t data_counter, const float* cam_beliefs_eta,
                               const float* cam_beliefs_lambda, const float* cam_priors_lambda,
                               float* cam_priors_eta);
/* Host metric (util.cpp:74-144) on read-back beliefs; same arithmetic as gbp_eval. */
GBP_API int gbp_eval_host(const gbp_problem* problem, const uint32_t* active_flag, const float* measurements,
                  const float* cam_beliefs_eta, const float* cam_beliefs_lambda,
                  const float* lmk_beliefs_eta, const float* lmk_beliefs_lambda,
                  double* sum_norm, double* sum_half_sq, uint64_t* n_active);

/* The solution: belief means mu = Lambda^-1 eta of every camera ([6C]: t_cw, axis-angle) and landmark ([3L]) from
 * read-back beliefs, same solve as the metric (util.cpp:103-108).  With gbp_bal_write this gives `--out_file`: the
 * refined problem in the input's own format (the reference only prints metrics and keeps the result on the device). */
GBP_API int gbp_belief_means(uint32_t n_cams, uint32_t n_lmks, const float* cam_beliefs_eta, const float* cam_beliefs_lambda,
                     const float* lmk_beliefs_eta, const float* lmk_beliefs_lambda, double* cameras, double* points);

/* Synthetic BAL generator (SURVEY 8d spec; the reference has none).  Fills a caller-allocated
 * gbp_bal with n_edges = n_lmks * obs_per_lmk, edges sorted by (camera, landmark). */
GBP_API int gbp_synth_generate(uint32_t n_cams, uint32_t n_lmks, uint32_t obs_per_lmk, uint64_t seed,
                       gbp_bal* out, double* gt_cameras /*[6C] or NULL*/, double* gt_points /*[3L] or NULL*/);

#ifdef __cplusplus
}
#endif
#endif /* GBP_MI355X_H */
