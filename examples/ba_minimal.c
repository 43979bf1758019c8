/* ba_minimal.c — `./ba --bal_file F [--n_iters N]` of the reference (ba/ba.cpp:479-1085) in plain C over the C-ABI alone:
 * include/gbp_mi355x.h, no C++, no Python.  What a foreign host (cgo, JNI, a C application) has to do, in the reference's order:
 *
 *   BALProblem::LoadFile            gbp_bal_read_header + gbp_bal_read                 dataio.cpp:17-57
 *   priors, scalings                gbp_set_prior_lambda, gbp_prior_scalings           dataio.cpp:67-117, ba.cpp:561-572
 *   graph build + Engine            gbp_create                                          ba.cpp:659-937
 *   WRITE_PROG, LINEARISE_PROG      gbp_upload, gbp_linearise                           ba.cpp:868-893
 *   the loop (weaken, GBP_PROG, metric, print)   gbp_ba_loop                            ba.cpp:1001-1028
 *   READ_PROG                       gbp_read                                            ba.cpp:908-916
 *
 *   gcc -std=c11 -O2 -Iinclude examples/ba_minimal.c -Lgbp_poplar_amd -lgbp_mi355x -Wl,-rpath,$PWD/gbp_poplar_amd -o /tmp/ba_minimal
 *
 * Prints the reference's lines (ba.cpp:996,1004,1026-1028); tests/test_cli.py holds its output against bin/ba's. */
#include "gbp_mi355x.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call)                                                                             \
  do {                                                                                          \
    const int rc_ = (call);                                                                     \
    if (rc_ != GBP_OK) {                                                                        \
      const char* why_ = gbp_last_error(ctx);                                                   \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, why_ ? why_ : "");                  \
      return 1;                                                                                 \
    }                                                                                           \
  } while (0)

static void* zalloc(size_t n, size_t size) {
  void* p = calloc(n ? n : 1, size);
  if (!p) { fprintf(stderr, "out of memory\n"); exit(1); }
  return p;
}

int main(int argc, char** argv) {
  const char* file = NULL;
  int n_iters = 1500;                                      /* ba.cpp:406-409 */
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--bal_file")) file = argv[i + 1];
    else if (!strcmp(argv[i], "--n_iters")) n_iters = atoi(argv[i + 1]);
  }
  if (!file) { fprintf(stderr, "usage: %s --bal_file F [--n_iters N]\n", argv[0]); return 2; }
  gbp_ctx* ctx = NULL;

  /* ---- the file ---- */
  gbp_bal bal;
  memset(&bal, 0, sizeof(bal));
  if (gbp_bal_read_header(file, &bal) != GBP_OK) { fprintf(stderr, "ERROR: unable to open file %s\n", file); return 1; }   /* ba.cpp:484-487 */
  const size_t C = bal.n_cams, L = bal.n_lmks, E = bal.n_edges;
  bal.cam_id = zalloc(E, 4); bal.lmk_id = zalloc(E, 4); bal.observations = zalloc(2 * E, 8);
  bal.cameras = zalloc(6 * C, 8); bal.points = zalloc(3 * L, 8);
  if (gbp_bal_read(file, &bal) != GBP_OK) { fprintf(stderr, "Invalid UW data file.\n"); return 1; }

  /* ---- what ba.cpp:489-604 builds on the host ---- */
  gbp_problem prob;
  memset(&prob, 0, sizeof(prob));
  prob.n_cams = bal.n_cams; prob.n_lmks = bal.n_lmks; prob.n_edges = bal.n_edges; prob.cam_id = bal.cam_id; prob.lmk_id = bal.lmk_id;
  const float K[9] = {(float)bal.fx, 0.f, (float)bal.cx, 0.f, (float)bal.fy, (float)bal.cy, 0.f, 0.f, 1.f};
  memcpy(prob.K, K, sizeof(K));
  float* meas = zalloc(2 * E, 4); float* var = zalloc(E, 4);
  float* cam = zalloc(6 * C, 4);  float* lmk = zalloc(3 * L, 4);
  for (size_t i = 0; i < 2 * E; ++i) meas[i] = (float)bal.observations[i];
  for (size_t i = 0; i < E; ++i) var[i] = 4.0f;                                          /* --reproj_meas_var, ba.cpp:442-445 */
  for (size_t i = 0; i < 6 * C; ++i) cam[i] = (float)bal.cameras[i];
  for (size_t i = 0; i < 3 * L; ++i) lmk[i] = (float)bal.points[i];
  float* cpe = zalloc(6 * C, 4); float* cpl = zalloc(36 * C, 4); float* lpe = zalloc(3 * L, 4); float* lpl = zalloc(9 * L, 4);
  CHECK(gbp_set_prior_lambda(&prob, 4.0f, cam, lmk, cam, lmk, cpe, cpl, lpe, lpl));
  const unsigned steps = 5;                                                                /* --steps, ba.cpp:454-457 */
  float* cs = zalloc(C, 4); float* ls = zalloc(L, 4);
  CHECK(gbp_prior_scalings((uint32_t)C, (uint32_t)L, cpl, (float)steps, 100.0f, 0.01f, cs, ls));   /* ba.cpp:561-572 */
  int32_t* count = zalloc(E, 4);
  uint32_t* active = zalloc(E, 4); uint32_t* cwf = zalloc(C, 4); uint32_t* lwf = zalloc(L, 4);
  for (size_t i = 0; i < E; ++i) { count[i] = -15; active[i] = 1u; }                      /* --undamped_start, ba.cpp:581; all factors active */
  for (size_t i = 0; i < C; ++i) cwf[i] = steps;                                           /* ba.cpp:574-576 */
  for (size_t i = 0; i < L; ++i) lwf[i] = steps;
  printf("Completed loading data!\n");

  /* ---- the engine and its program list ---- */
  if (gbp_create(&prob, NULL, NULL, &ctx) != GBP_OK) {
    const char* why = gbp_last_error(NULL);
    printf("Could not find a device\n%s\n", why ? why : "");                              /* ba.cpp:652-655 */
    return 255;
  }
  gbp_state_in in;
  memset(&in, 0, sizeof(in));                                                             /* damping, mu, oldmu: NULL = the zeros the reference uploads */
  in.damping_count = count; in.active_flag = active; in.cam_scaling = cs; in.lmk_scaling = ls; in.cam_weaken_flag = cwf; in.lmk_weaken_flag = lwf;
  in.cam_priors_eta = cpe; in.cam_priors_lambda = cpl; in.lmk_priors_eta = lpe; in.lmk_priors_lambda = lpl;
  in.measurements = meas; in.meas_variances = var;
  CHECK(gbp_upload(ctx, &in));
  CHECK(gbp_linearise(ctx));
  gbp_eval_out ev;
  CHECK(gbp_eval(ctx, &ev));
  /* (the reference prints floats through iostream: six significant digits = %g) */
  printf("Initial Reprojection error: %g Cost %g\n", (double)(float)(ev.sum_norm / (double)ev.n_active), (double)(float)ev.sum_half_sq);     /* ba.cpp:996 */

  /* ---- the loop: up to 256 passes per call, the metric of every pass comes back ---- */
  gbp_eval_out* out = zalloc(256, sizeof(gbp_eval_out));
  for (int it = 0; it < n_iters;) {
    const int n = n_iters - it < 256 ? n_iters - it : 256;
    CHECK(gbp_ba_loop(ctx, n, (unsigned)it, steps, out));
    for (int k = 0; k < n; ++k) {
      const unsigned i = (unsigned)(it + k);
      if ((i + 1) % 2 == 0 && i < 2 * steps) printf("Weakening priors \n");               /* ba.cpp:1003-1006 */
      printf("Iter %u // Reprojection error %g // Cost %g // n relins: %llu // n robust edges %llu\n", i,
             (double)(float)(out[k].sum_norm / (double)out[k].n_active), (double)(float)out[k].sum_half_sq, (unsigned long long)out[k].n_relin,
             (unsigned long long)out[k].n_robust);
    }
    it += n;
  }

  /* ---- READ_PROG ---- */
  gbp_state_out back;
  memset(&back, 0, sizeof(back));
  back.cam_beliefs_eta = zalloc(6 * C, 4); back.cam_beliefs_lambda = zalloc(36 * C, 4);
  CHECK(gbp_read(ctx, &back));
  printf("camera 1 belief eta: %.6f %.6f %.6f %.6f %.6f %.6f\n", back.cam_beliefs_eta[6], back.cam_beliefs_eta[7], back.cam_beliefs_eta[8],
         back.cam_beliefs_eta[9], back.cam_beliefs_eta[10], back.cam_beliefs_eta[11]);
  gbp_destroy(ctx);
  return 0;
}
