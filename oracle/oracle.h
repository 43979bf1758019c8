/* oracle.h — TEST INFRASTRUCTURE ONLY.  CPU oracle for the GBP sweep of joeaortiz/gbp-poplar.
 *
 * Who may use this: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — as the
 * checker / reported baseline, never as the product path.  gbp_poplar_amd/ does not import it.
 *
 * What it is: a plain-C restatement of the reference's seven Poplar vertex classes
 * (ba/gbp_codelets.cpp) driven in the program order of ba/ba.cpp:890-905 and
 * ba/slam.cpp:1018-1055, plus the Eigen-free restatement of the host functions that feed /
 * consume the path (dataio.cpp:17-117,455-508; util.cpp:48-144,183-223; ba.cpp:561-572).
 * It shares the C structs of include/gbp_mi355x.h so that parity tests drive both sides alike.
 *
 * Pin status (details in DESIGN.md "Oracle"):
 *   - dense-math layer (matlib.cpp, bafuncs.cpp): pinned bit-for-bit against the reference's own
 *     code compiled here (ref_adapter.cpp, built out of tree by `make ref`);
 *   - vertex classes + schedule: gbp_codelets.cpp / ba.cpp need the Poplar SDK (absent) and are
 *     NOT built; pinned against the reference-run known answers recorded in BASELINE.md
 *     (metric trajectory of the reference's vertex code on fr1xyz / fr2robot2, incl. the chaotic
 *     1500-iteration finals) — tests/test_oracle_known_answers.py;
 *   - popops::reduceWithOutput summation order and Eigen's inverse: unpinned by the reference
 *     (third-party, unspecified) — fixed here as ascending slot order / fp64 partial-pivot solve.
 */
#ifndef GBP_ORACLE_H
#define GBP_ORACLE_H
#include "../include/gbp_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

orc_ctx* orc_create(const gbp_problem* problem, const gbp_params* params);
void     orc_destroy(orc_ctx* o);
void     orc_set_threads(int n);      /* OpenMP threads for the factor / variable loops */
int      orc_get_max_threads(void);

/* Camera-belief summation order.  mode 0: ascending slot order, prior first (default; what the
 * BASELINE.md trajectories were produced with).  mode 1: the device's order — per landmark shard
 * r, the camera's local factors in file order in rows of 16, each row a balanced binary tree,
 * rows added left to right; belief = prior + local_0 + ... + local_{n-1}.  Landmark beliefs are
 * slot-ordered in both modes.  bounds = n_shards+1 landmark indices (NULL for one shard). */
int orc_set_sum_order(orc_ctx* o, int mode, int n_shards, const uint32_t* bounds);

int orc_upload(orc_ctx* o, const gbp_state_in* in);
int orc_linearise(orc_ctx* o);
int orc_iterate(orc_ctx* o, int n_iters);
int orc_weaken_priors(orc_ctx* o);
int orc_read(orc_ctx* o, gbp_state_out* out);
int orc_read_priors(orc_ctx* o, gbp_priors_out* out);
int orc_new_keyframe(orc_ctx* o, const gbp_kf_update* upd);
int orc_eval(orc_ctx* o, gbp_eval_out* out);

/* Rank-local split-phase view, mirroring gbp_iterate_begin/_end, gbp_refresh_begin/_end and
 * gbp_linearise_factors of the C-ABI (camera records of 44 floats: eta 6, pad 2, Lambda 36), so the
 * multi-process host logic can be tested on CPU (gloo).  Only factors whose landmark lies in
 * [lmk_begin, lmk_end) are processed. */
int orc_set_shard(orc_ctx* o, int rank, int world, uint32_t lmk_begin, uint32_t lmk_end);
int orc_iterate_begin(orc_ctx* o, float* send);
int orc_iterate_end(orc_ctx* o, const float* recv);
int orc_refresh_begin(orc_ctx* o, float* send);
int orc_refresh_end(orc_ctx* o, const float* recv);
int orc_linearise_factors(orc_ctx* o);
int orc_weaken_priors_sharded(orc_ctx* o, const float* recv);
int orc_new_keyframe_sharded(orc_ctx* o, const gbp_kf_update* upd, const float* recv);

/* raw internal state for stage-level parity (reference tensor names, ba.cpp:665-687,759-775) */
int orc_get_factor_potentials(orc_ctx* o, float* eta9E, float* lambda81E);
int orc_get_messages(orc_ctx* o, float* cam_eta6E, float* cam_lam36E, float* lmk_eta3E, float* lmk_lam9E);
int orc_get_mu(orc_ctx* o, float* mu9E, float* dmuE);

/* host-side restatements */
int orc_bal_read_header(const char* path, gbp_bal* hdr);
int orc_bal_read(const char* path, gbp_bal* bal);
int orc_set_prior_lambda(const gbp_problem* p, float reproj_meas_var, const float* cam_file,
                         const float* lmk_file, const float* cam_mean, const float* lmk_mean,
                         float* cam_eta, float* cam_lam, float* lmk_eta, float* lmk_lam);
int orc_prior_scalings(uint32_t C, uint32_t L, const float* cam_priors_lambda, float steps,
                       float weaker, float first_std, float* cam_scaling, float* lmk_scaling);
int orc_slam_create_flags(const gbp_problem* p, uint32_t steps, uint32_t* active, uint32_t* cam_wf,
                          uint32_t* lmk_wf, uint32_t* lmk_af);
int orc_slam_update_flags(const gbp_problem* p, uint32_t steps, uint32_t data_counter,
                          uint32_t* active, uint32_t* lmk_wf, uint32_t* cam_wf, uint32_t* lmk_af,
                          int32_t* n_new);
int orc_slam_initialise_new_kf(uint32_t data_counter, const float* cam_bel_eta,
                               const float* cam_bel_lam, const float* cam_prior_lam,
                               float* cam_prior_eta);
int orc_eval_host(const gbp_problem* p, const uint32_t* active, const float* meas,
                  const float* cbe, const float* cbl, const float* lbe, const float* lbl,
                  double* sum_norm, double* sum_half_sq, uint64_t* n_active);
/* the reference's own accumulation (fp32, sequential: util.cpp:136-143) for documentation */
int orc_eval_host_f32(const gbp_problem* p, const uint32_t* active, const float* meas,
                      const float* cbe, const float* cbl, const float* lbe, const float* lbl,
                      float* reproj2);
const char* orc_math_impl(void);

#ifdef __cplusplus
}
#endif
#endif
