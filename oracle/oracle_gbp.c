/* oracle_gbp.c — TEST INFRASTRUCTURE ONLY (see oracle.h for who may use it and the pin status).
 *
 * Plain-C restatement of the reference's vertex programs and their schedule:
 *   RelineariseFactorVertex        gbp_codelets.cpp:20-172    -> relinearise_zero()
 *   WeakenPriorVertex              gbp_codelets.cpp:176-197   -> weaken_var()
 *   PrepMessageVertex              gbp_codelets.cpp:215-379   -> prep_factor()
 *   ComputeCamMessageEtaVertex     gbp_codelets.cpp:382-472   -> msg_cam_eta()
 *   ComputeLmkMessageEtaVertex     gbp_codelets.cpp:475-563   -> msg_lmk_eta()
 *   ComputeCamMessageLambdaVertex  gbp_codelets.cpp:567-638   -> msg_cam_lambda()
 *   ComputeLmkMessageLambdaVertex  gbp_codelets.cpp:641-710   -> msg_lmk_lambda()
 *   popops::reduceWithOutput x4    ba.cpp:104-139             -> update_beliefs()
 *   program order                  ba.cpp:890-905, slam.cpp:919-928
 * Tensors that the reference leaves uninitialised (ba.cpp:668-687,759-775) start at zero.
 * Messages are kept per factor with CSR slot lists instead of the reference's max-degree padded
 * slots (ba.cpp:680-687); padded slots hold +0 and do not change an fp32 sum.
 */
#include "oracle.h"
#include "oracle_math.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct orc_ctx {
  uint32_t C, L, E;
  uint32_t *cam_id, *lmk_id;
  float K[9];
  gbp_params prm;
  uint32_t *cam_ptr, *cam_f, *lmk_ptr, *lmk_f; /* incident factors in slot (= file) order */
  float *fac_eta, *fac_lam;                    /* [9E], [81E] = [cc36|cl18|lc18|ll9] (ba.cpp:93-96) */
  float *mce, *mcl, *mle, *mll;                /* current messages  [6E] [36E] [3E] [9E] */
  float *pce, *pcl, *ple, *pll;                /* previous messages */
  float *damping, *mu, *oldmu, *dmu, *meas, *var;
  int32_t* count;
  uint32_t *active, *robust;
  float *cpe, *cpl, *lpe, *lpl;                /* priors = message slot 0 */
  float *cbe, *cbl, *lbe, *lbl;                /* beliefs */
  float *cscale, *lscale;
  uint32_t *cwf, *lwf;
  int sum_mode, n_shards;
  uint32_t* bounds;
  int uploaded;
  /* rank-local view for the multi-process tests (orc_set_shard): only factors whose landmark lies in
   * [lb, le) are processed; camera beliefs come from gathered per-rank partials */
  int sh_rank, sh_world;
  uint32_t lb, le;
};

static int is_local(const orc_ctx* o, uint32_t e) { return o->lmk_id[e] >= o->lb && o->lmk_id[e] < o->le; }

const char* orc_math_impl(void) { return om_impl_name(); }

void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static void* zalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

orc_ctx* orc_create(const gbp_problem* p, const gbp_params* prm) {
  orc_ctx* o = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  uint32_t C = p->n_cams, L = p->n_lmks, E = p->n_edges, e;
  uint32_t *cc, *lc;
  o->C = C; o->L = L; o->E = E;
  memcpy(o->K, p->K, sizeof(o->K));
  if (prm) o->prm = *prm;
  else {
    o->prm.maxeta_damping = 0.4f; o->prm.num_undamped_iters = 8; o->prm.dmu_threshold = 3e-3f;
    o->prm.min_linear_iters = 10; o->prm.nstds = 2.5f; o->prm.relin_mode = 0;
  }
  o->cam_id = (uint32_t*)zalloc(E, 4); o->lmk_id = (uint32_t*)zalloc(E, 4);
  memcpy(o->cam_id, p->cam_id, (size_t)E * 4); memcpy(o->lmk_id, p->lmk_id, (size_t)E * 4);
  /* slot lists: O(E) counting pass replacing the O(E^2) search of ba.cpp:267-279 */
  o->cam_ptr = (uint32_t*)zalloc(C + 1, 4); o->lmk_ptr = (uint32_t*)zalloc(L + 1, 4);
  o->cam_f = (uint32_t*)zalloc(E, 4); o->lmk_f = (uint32_t*)zalloc(E, 4);
  for (e = 0; e < E; ++e) { o->cam_ptr[o->cam_id[e] + 1]++; o->lmk_ptr[o->lmk_id[e] + 1]++; }
  for (e = 0; e < C; ++e) o->cam_ptr[e + 1] += o->cam_ptr[e];
  for (e = 0; e < L; ++e) o->lmk_ptr[e + 1] += o->lmk_ptr[e];
  cc = (uint32_t*)zalloc(C, 4); lc = (uint32_t*)zalloc(L, 4);
  for (e = 0; e < E; ++e) {
    uint32_t c = o->cam_id[e], l = o->lmk_id[e];
    o->cam_f[o->cam_ptr[c] + cc[c]++] = e;
    o->lmk_f[o->lmk_ptr[l] + lc[l]++] = e;
  }
  free(cc); free(lc);
  o->fac_eta = (float*)zalloc((size_t)E * 9, 4); o->fac_lam = (float*)zalloc((size_t)E * 81, 4);
  o->mce = (float*)zalloc((size_t)E * 6, 4);  o->mcl = (float*)zalloc((size_t)E * 36, 4);
  o->mle = (float*)zalloc((size_t)E * 3, 4);  o->mll = (float*)zalloc((size_t)E * 9, 4);
  o->pce = (float*)zalloc((size_t)E * 6, 4);  o->pcl = (float*)zalloc((size_t)E * 36, 4);
  o->ple = (float*)zalloc((size_t)E * 3, 4);  o->pll = (float*)zalloc((size_t)E * 9, 4);
  o->damping = (float*)zalloc(E, 4); o->count = (int32_t*)zalloc(E, 4);
  o->mu = (float*)zalloc((size_t)E * 9, 4); o->oldmu = (float*)zalloc((size_t)E * 9, 4);
  o->dmu = (float*)zalloc(E, 4); o->meas = (float*)zalloc((size_t)E * 2, 4); o->var = (float*)zalloc(E, 4);
  o->active = (uint32_t*)zalloc(E, 4); o->robust = (uint32_t*)zalloc(E, 4);
  o->cpe = (float*)zalloc((size_t)C * 6, 4); o->cpl = (float*)zalloc((size_t)C * 36, 4);
  o->lpe = (float*)zalloc((size_t)L * 3, 4); o->lpl = (float*)zalloc((size_t)L * 9, 4);
  o->cbe = (float*)zalloc((size_t)C * 6, 4); o->cbl = (float*)zalloc((size_t)C * 36, 4);
  o->lbe = (float*)zalloc((size_t)L * 3, 4); o->lbl = (float*)zalloc((size_t)L * 9, 4);
  o->cscale = (float*)zalloc(C, 4); o->lscale = (float*)zalloc(L, 4);
  o->cwf = (uint32_t*)zalloc(C, 4); o->lwf = (uint32_t*)zalloc(L, 4);
  o->sum_mode = 0; o->n_shards = 1; o->bounds = NULL;
  o->sh_rank = 0; o->sh_world = 1; o->lb = 0; o->le = L;
  return o;
}

void orc_destroy(orc_ctx* o) {
  if (!o) return;
  free(o->cam_id); free(o->lmk_id); free(o->cam_ptr); free(o->cam_f); free(o->lmk_ptr); free(o->lmk_f);
  free(o->fac_eta); free(o->fac_lam); free(o->mce); free(o->mcl); free(o->mle); free(o->mll);
  free(o->pce); free(o->pcl); free(o->ple); free(o->pll); free(o->damping); free(o->count);
  free(o->mu); free(o->oldmu); free(o->dmu); free(o->meas); free(o->var); free(o->active);
  free(o->robust); free(o->cpe); free(o->cpl); free(o->lpe); free(o->lpl); free(o->cbe);
  free(o->cbl); free(o->lbe); free(o->lbl); free(o->cscale); free(o->lscale); free(o->cwf);
  free(o->lwf); free(o->bounds); free(o);
}

int orc_set_sum_order(orc_ctx* o, int mode, int n_shards, const uint32_t* bounds) {
  o->sum_mode = mode;
  o->n_shards = n_shards > 0 ? n_shards : 1;
  free(o->bounds); o->bounds = NULL;
  if (bounds && n_shards > 0) {
    o->bounds = (uint32_t*)malloc((size_t)(n_shards + 1) * 4);
    memcpy(o->bounds, bounds, (size_t)(n_shards + 1) * 4);
  }
  return 0;
}

#define CPY(dst, src, n) do { if (src) memcpy(dst, src, (size_t)(n) * 4); else memset(dst, 0, (size_t)(n) * 4); } while (0)

/* WRITE_PROG, ba.cpp:868-886 */
int orc_upload(orc_ctx* o, const gbp_state_in* in) {
  uint32_t C = o->C, L = o->L, E = o->E;
  if (!in->cam_priors_eta || !in->cam_priors_lambda || !in->lmk_priors_eta || !in->lmk_priors_lambda ||
      !in->measurements || !in->meas_variances || !in->active_flag) return GBP_ERR_INVALID;
  CPY(o->damping, in->damping, E); CPY(o->count, in->damping_count, E);
  CPY(o->mu, in->mu, (size_t)E * 9); CPY(o->oldmu, in->oldmu, (size_t)E * 9);
  CPY(o->active, in->active_flag, E);
  CPY(o->cscale, in->cam_scaling, C); CPY(o->lscale, in->lmk_scaling, L);
  CPY(o->cwf, in->cam_weaken_flag, C); CPY(o->lwf, in->lmk_weaken_flag, L);
  CPY(o->cpe, in->cam_priors_eta, (size_t)C * 6); CPY(o->cpl, in->cam_priors_lambda, (size_t)C * 36);
  CPY(o->lpe, in->lmk_priors_eta, (size_t)L * 3); CPY(o->lpl, in->lmk_priors_lambda, (size_t)L * 9);
  CPY(o->meas, in->measurements, (size_t)E * 2); CPY(o->var, in->meas_variances, E);
  o->uploaded = 1;
  return 0;
}

/* balanced binary tree over 16 values in natural order (the device's row reduction) */
static float tree16(const float* x) {
  float a[8], b[4], c[2];
  int i;
  for (i = 0; i < 8; ++i) a[i] = x[2 * i] + x[2 * i + 1];
  for (i = 0; i < 4; ++i) b[i] = a[2 * i] + a[2 * i + 1];
  for (i = 0; i < 2; ++i) c[i] = b[2 * i] + b[2 * i + 1];
  return c[0] + c[1];
}

/* element k (0..5 eta, 6..41 Lambda) of camera c's partial sum over the factors whose landmark lies
 * in [lo, hi): rows of 16 in file order, each a balanced tree, rows added left to right */
static float cam_partial(const orc_ctx* o, uint32_t c, int k, uint32_t lo, uint32_t hi) {
  float local = 0.f, row[16];
  int n = 0, have = 0;
  uint32_t s;
  for (s = o->cam_ptr[c]; s < o->cam_ptr[c + 1]; ++s) {
    uint32_t e = o->cam_f[s];
    if (o->lmk_id[e] < lo || o->lmk_id[e] >= hi) continue;
    row[n++] = k < 6 ? o->mce[(size_t)e * 6 + k] : o->mcl[(size_t)e * 36 + (k - 6)];
    if (n == 16) { float t = tree16(row); local = have ? local + t : t; have = 1; n = 0; }
  }
  if (n) {
    float t;
    while (n < 16) row[n++] = 0.f;
    t = tree16(row); local = have ? local + t : t; have = 1;
  }
  return local;
}

static void cam_belief_device_order(const orc_ctx* o, uint32_t c) {
  int k, r;
  float* be = o->cbe + (size_t)c * 6;
  float* bl = o->cbl + (size_t)c * 36;
  for (k = 0; k < 42; ++k) {
    float acc = k < 6 ? o->cpe[(size_t)c * 6 + k] : o->cpl[(size_t)c * 36 + (k - 6)];
    for (r = 0; r < o->n_shards; ++r) {
      uint32_t lo = o->bounds ? o->bounds[r] : 0, hi = o->bounds ? o->bounds[r + 1] : o->L;
      acc = acc + cam_partial(o, c, k, lo, hi);
    }
    if (k < 6) be[k] = acc; else bl[k - 6] = acc;
  }
}

/* prog_ub: four popops::reduceWithOutput over {var, slot, dofs}, ba.cpp:104-139; slot 0 = prior. */
static void update_beliefs(orc_ctx* o) {
  long c, l;
#pragma omp parallel for schedule(static)
  for (c = 0; c < (long)o->C; ++c) {
    if (o->sum_mode == 1) { cam_belief_device_order(o, (uint32_t)c); continue; }
    {
      float* be = o->cbe + c * 6;
      float* bl = o->cbl + c * 36;
      uint32_t s;
      int k;
      for (k = 0; k < 6; ++k) be[k] = o->cpe[c * 6 + k];
      for (k = 0; k < 36; ++k) bl[k] = o->cpl[c * 36 + k];
      for (s = o->cam_ptr[c]; s < o->cam_ptr[c + 1]; ++s) {
        uint32_t e = o->cam_f[s];
        for (k = 0; k < 6; ++k) be[k] += o->mce[(size_t)e * 6 + k];
        for (k = 0; k < 36; ++k) bl[k] += o->mcl[(size_t)e * 36 + k];
      }
    }
  }
#pragma omp parallel for schedule(static)
  for (l = 0; l < (long)o->L; ++l) {
    float* be = o->lbe + l * 3;
    float* bl = o->lbl + l * 9;
    uint32_t s;
    int k;
    for (k = 0; k < 3; ++k) be[k] = o->lpe[l * 3 + k];
    for (k = 0; k < 9; ++k) bl[k] = o->lpl[l * 9 + k];
    for (s = o->lmk_ptr[l]; s < o->lmk_ptr[l + 1]; ++s) {
      uint32_t e = o->lmk_f[s];
      for (k = 0; k < 3; ++k) be[k] += o->mle[(size_t)e * 3 + k];
      for (k = 0; k < 9; ++k) bl[k] += o->mll[(size_t)e * 9 + k];
    }
  }
}

/* linearisation point = belief means: inf2mean6x6 / inf2mean3x3, bafuncs.cpp:2-15 */
static void belief_means(const orc_ctx* o, uint32_t e, float* x0c, float* x0l) {
  uint32_t c = o->cam_id[e], l = o->lmk_id[e];
  int i;
  for (i = 0; i < 6; ++i) x0c[i] = 0.f;
  for (i = 0; i < 3; ++i) x0l[i] = 0.f;
  om_inf2mean6x6(o->cbe + (size_t)c * 6, o->cbl + (size_t)c * 36, x0c);
  om_inf2mean3x3(o->lbe + (size_t)l * 3, o->lbl + (size_t)l * 9, x0l);
}

/* Shared body of gbp_codelets.cpp:90-168 and :294-373: accumulate J^T J and J^T(Jx0+z-h(x0))
 * onto the factor potential, Huber-rescale, Lambda_lc = Lambda_cl^T. */
static void relin_core(orc_ctx* o, uint32_t e, const float* x0c, const float* x0l) {
  float* eta = o->fac_eta + (size_t)e * 9;
  float* cc = o->fac_lam + (size_t)e * 81;
  float *cl = cc + 36, *lc = cc + 54, *ll = cc + 72;
  const float* z = o->meas + (size_t)e * 2;
  float var = o->var[e];
  float Jkf[12] = {0}, Jl[6] = {0}, hx[2] = {0}, buf[2] = {0}, x0[9], J[18] = {0};
  float err, mvar, nstds = o->prm.nstds;
  int i, j;
  om_jac(x0c, x0l, o->K, Jkf, Jl);
  om_matmul(Jkf, 2, 6, Jkf, 2, 6, cc, 6, 1, 0);
  om_matmul(Jl, 2, 3, Jl, 2, 3, ll, 3, 1, 0);
  om_matmul(Jkf, 2, 6, Jl, 2, 3, cl, 3, 1, 0);
  om_hfunc(x0c, x0l, o->K, hx);
  for (i = 0; i < 6; ++i) x0[i] = x0c[i];
  for (i = 0; i < 3; ++i) x0[i + 6] = x0l[i];
  for (i = 0; i < 2; ++i) {
    for (j = 0; j < 6; ++j) J[i * 9 + j] = Jkf[i * 6 + j];
    for (j = 0; j < 3; ++j) J[i * 9 + j + 6] = Jl[i * 3 + j];
  }
  om_matmul(J, 2, 9, x0, 9, 1, buf, 1, 0, 0);
  for (i = 0; i < 2; ++i) buf[i] = buf[i] + z[i];
  for (i = 0; i < 2; ++i) buf[i] = buf[i] - hx[i];
  om_matmul(J, 2, 9, buf, 2, 1, eta, 1, 1, 0);

  /* Huber: gbp_codelets.cpp:135-141 (the 0.5 literal makes the denominator a double expression) */
  err = sqrtf((hx[0] - z[0]) * (hx[0] - z[0]) + (hx[1] - z[1]) * (hx[1] - z[1]));
  mvar = var;
  if (err > nstds * sqrtf(var)) {
    o->robust[e] = 1;
    mvar = var * err * err / (2 * (nstds * sqrtf(var) * err - 0.5 * nstds * nstds * var));
  } else {
    o->robust[e] = 0;
  }
  for (i = 0; i < 36; ++i) cc[i] /= mvar;
  for (i = 0; i < 9; ++i) ll[i] /= mvar;
  for (i = 0; i < 18; ++i) cl[i] /= mvar;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 6; ++j) lc[i * 6 + j] = cl[j * 3 + i];
  for (i = 0; i < 9; ++i) eta[i] /= mvar;
}

/* RelineariseFactorVertex, gbp_codelets.cpp:38-171 (no active_flag test: runs on every factor) */
static void relinearise_zero(orc_ctx* o, uint32_t e) {
  float x0c[6], x0l[3];
  memset(o->fac_eta + (size_t)e * 9, 0, 9 * 4);
  memset(o->fac_lam + (size_t)e * 81, 0, 81 * 4);
  belief_means(o, e, x0c, x0l);
  relin_core(o, e, x0c, x0l);
}

/* PrepMessageVertex, gbp_codelets.cpp:241-378 */
static void prep_factor(orc_ctx* o, uint32_t e) {
  float x0c[6], x0l[3], d;
  int i;
  if (o->active[e] != 1) return;
  if (0 == o->count[e]) o->damping[e] = o->prm.maxeta_damping;
  o->count[e] += 1;
  belief_means(o, e, x0c, x0l);
  d = 0.0;
  for (i = 0; i < 6; ++i) {
    d += (o->oldmu[(size_t)e * 9 + i] - x0c[i]) * (o->oldmu[(size_t)e * 9 + i] - x0c[i]);
    o->mu[(size_t)e * 9 + i] = x0c[i];
  }
  for (i = 0; i < 3; ++i) {
    d += (o->oldmu[(size_t)e * 9 + i + 6] - x0l[i]) * (o->oldmu[(size_t)e * 9 + i + 6] - x0l[i]);
    o->mu[(size_t)e * 9 + i + 6] = x0l[i];
  }
  d = sqrtf(d);
  o->dmu[e] = d;
  if ((d < o->prm.dmu_threshold) && (o->count[e] > o->prm.min_linear_iters - o->prm.num_undamped_iters)) {
    o->damping[e] = 0.0;
    o->count[e] = -o->prm.num_undamped_iters;
    if (o->prm.relin_mode == 1) { /* opt-in: zero first (quirk C-1 "fixed") */
      memset(o->fac_eta + (size_t)e * 9, 0, 9 * 4);
      memset(o->fac_lam + (size_t)e * 81, 0, 81 * 4);
    }
    relin_core(o, e, x0c, x0l); /* accumulates onto the old potential: matMul is += */
  }
}

/* ComputeCamMessageEtaVertex, gbp_codelets.cpp:411-471 */
static void msg_cam_eta(orc_ctx* o, uint32_t e) {
  uint32_t l = o->lmk_id[e];
  const float* Ef = o->fac_eta + (size_t)e * 9;
  const float* Lf = o->fac_lam + (size_t)e * 81;
  float* out = o->mce + (size_t)e * 6;
  int i;
  if (o->active[e] == 1) {
    float b1[9], b2[9] = {0}, b3[18] = {0}, b4[3], b5[6] = {0}, b6[6];
    float d = o->damping[e];
    for (i = 0; i < 9; ++i) b1[i] = Lf[72 + i] + o->lbl[(size_t)l * 9 + i];
    for (i = 0; i < 9; ++i) b1[i] = b1[i] - o->pll[(size_t)e * 9 + i];
    om_inv3x3(b1, b2);
    om_matmul(Lf + 36, 6, 3, b2, 3, 3, b3, 3, 0, 0);
    for (i = 0; i < 3; ++i) b4[i] = Ef[6 + i] + o->lbe[(size_t)l * 3 + i];
    for (i = 0; i < 3; ++i) b4[i] = b4[i] - o->ple[(size_t)e * 3 + i];
    om_matmul(b3, 6, 3, b4, 3, 1, b5, 1, 0, 0);
    for (i = 0; i < 6; ++i) b6[i] = Ef[i] - b5[i];
    for (i = 0; i < 6; ++i) out[i] = b6[i] * (1 - d) + o->pce[(size_t)e * 6 + i] * d;
  } else {
    for (i = 0; i < 6; ++i) out[i] = 0.0;
  }
}

/* ComputeLmkMessageEtaVertex, gbp_codelets.cpp:503-562 */
static void msg_lmk_eta(orc_ctx* o, uint32_t e) {
  uint32_t c = o->cam_id[e];
  const float* Ef = o->fac_eta + (size_t)e * 9;
  const float* Lf = o->fac_lam + (size_t)e * 81;
  float* out = o->mle + (size_t)e * 3;
  int i;
  if (o->active[e] == 1) {
    float b1[36], b2[36] = {0}, b3[18] = {0}, b4[6], b5[3] = {0}, b6[3];
    float d = o->damping[e];
    for (i = 0; i < 36; ++i) b1[i] = Lf[i] + o->cbl[(size_t)c * 36 + i];
    for (i = 0; i < 36; ++i) b1[i] = b1[i] - o->pcl[(size_t)e * 36 + i];
    om_inv6x6(b1, b2);
    om_matmul(Lf + 54, 3, 6, b2, 6, 6, b3, 6, 0, 0);
    for (i = 0; i < 6; ++i) b4[i] = Ef[i] + o->cbe[(size_t)c * 6 + i];
    for (i = 0; i < 6; ++i) b4[i] = b4[i] - o->pce[(size_t)e * 6 + i];
    om_matmul(b3, 3, 6, b4, 6, 1, b5, 1, 0, 0);
    for (i = 0; i < 3; ++i) b6[i] = Ef[6 + i] - b5[i];
    for (i = 0; i < 3; ++i) out[i] = b6[i] * (1 - d) + o->ple[(size_t)e * 3 + i] * d;
  } else {
    for (i = 0; i < 3; ++i) out[i] = 0.0;
  }
}

/* ComputeCamMessageLambdaVertex, gbp_codelets.cpp:592-637 */
static void msg_cam_lambda(orc_ctx* o, uint32_t e) {
  uint32_t l = o->lmk_id[e];
  const float* Lf = o->fac_lam + (size_t)e * 81;
  float* out = o->mcl + (size_t)e * 36;
  int i;
  if (o->active[e] == 1) {
    float b1[9], b2[9] = {0}, b3[18] = {0}, b4[36] = {0};
    for (i = 0; i < 9; ++i) b1[i] = Lf[72 + i] + o->lbl[(size_t)l * 9 + i];
    for (i = 0; i < 9; ++i) b1[i] = b1[i] - o->pll[(size_t)e * 9 + i];
    om_inv3x3(b1, b2);
    om_matmul(Lf + 36, 6, 3, b2, 3, 3, b3, 3, 0, 0);
    om_matmul(b3, 6, 3, Lf + 54, 3, 6, b4, 6, 0, 0);
    for (i = 0; i < 36; ++i) out[i] = Lf[i] - b4[i];
  } else {
    for (i = 0; i < 36; ++i) out[i] = 0.0;
  }
}

/* ComputeLmkMessageLambdaVertex, gbp_codelets.cpp:664-709 */
static void msg_lmk_lambda(orc_ctx* o, uint32_t e) {
  uint32_t c = o->cam_id[e];
  const float* Lf = o->fac_lam + (size_t)e * 81;
  float* out = o->mll + (size_t)e * 9;
  int i;
  if (o->active[e] == 1) {
    float b1[36], b2[36] = {0}, b3[18] = {0}, b4[9] = {0};
    for (i = 0; i < 36; ++i) b1[i] = Lf[i] + o->cbl[(size_t)c * 36 + i];
    for (i = 0; i < 36; ++i) b1[i] = b1[i] - o->pcl[(size_t)e * 36 + i];
    om_inv6x6(b1, b2);
    om_matmul(Lf + 54, 3, 6, b2, 6, 6, b3, 6, 0, 0);
    om_matmul(b3, 3, 6, Lf + 36, 6, 3, b4, 3, 0, 0);
    for (i = 0; i < 9; ++i) out[i] = Lf[72 + i] - b4[i];
  } else {
    for (i = 0; i < 9; ++i) out[i] = 0.0;
  }
}

/* LINEARISE_PROG, ba.cpp:890-893: prog_ub then cs_relinearise */
int orc_linearise(orc_ctx* o) {
  long e;
  if (!o->uploaded) return GBP_ERR_STATE;
  update_beliefs(o);
#pragma omp parallel for schedule(static)
  for (e = 0; e < (long)o->E; ++e) relinearise_zero(o, (uint32_t)e);
  return 0;
}

/* GBP_PROG, ba.cpp:895-905 */
int orc_iterate(orc_ctx* o, int n) {
  int it;
  long e;
  size_t E = o->E;
  if (!o->uploaded) return GBP_ERR_STATE;
  for (it = 0; it < n; ++it) {
#pragma omp parallel for schedule(static)
    for (e = 0; e < (long)E; ++e) prep_factor(o, (uint32_t)e);
    memcpy(o->oldmu, o->mu, E * 9 * 4);
#pragma omp parallel for schedule(static)
    for (e = 0; e < (long)E; ++e) {
      msg_cam_eta(o, (uint32_t)e);
      msg_lmk_eta(o, (uint32_t)e);
      msg_cam_lambda(o, (uint32_t)e);
      msg_lmk_lambda(o, (uint32_t)e);
    }
    update_beliefs(o);
    memcpy(o->pce, o->mce, E * 6 * 4);  memcpy(o->pcl, o->mcl, E * 36 * 4);
    memcpy(o->ple, o->mle, E * 3 * 4);  memcpy(o->pll, o->mll, E * 9 * 4);
  }
  return 0;
}

/* WeakenPriorVertex, gbp_codelets.cpp:184-196 */
static void weaken_var(float scaling, uint32_t* flag, float* eta, int ne, float* lam, int nl) {
  int i;
  if ((*flag == 5) || (*flag == 4) || (*flag == 3) || (*flag == 2) || (*flag == 1)) {
    *flag -= 1;
    for (i = 0; i < ne; ++i) eta[i] *= scaling;
    for (i = 0; i < nl; ++i) lam[i] *= scaling;
  }
}

/* WEAKEN_PRIORS, ba.cpp:863-865 */
int orc_weaken_priors(orc_ctx* o) {
  uint32_t v;
  for (v = 0; v < o->C; ++v) weaken_var(o->cscale[v], &o->cwf[v], o->cpe + (size_t)v * 6, 6, o->cpl + (size_t)v * 36, 36);
  for (v = 0; v < o->L; ++v) weaken_var(o->lscale[v], &o->lwf[v], o->lpe + (size_t)v * 3, 3, o->lpl + (size_t)v * 9, 9);
  update_beliefs(o);
  return 0;
}

#define OUT(dst, src, n) do { if (dst) memcpy(dst, src, (size_t)(n) * 4); } while (0)

/* READ_PROG, ba.cpp:908-916 */
int orc_read(orc_ctx* o, gbp_state_out* out) {
  OUT(out->cam_beliefs_eta, o->cbe, (size_t)o->C * 6); OUT(out->cam_beliefs_lambda, o->cbl, (size_t)o->C * 36);
  OUT(out->lmk_beliefs_eta, o->lbe, (size_t)o->L * 3); OUT(out->lmk_beliefs_lambda, o->lbl, (size_t)o->L * 9);
  OUT(out->damping, o->damping, o->E); OUT(out->damping_count, o->count, o->E);
  OUT(out->robust_flag, o->robust, o->E);
  return 0;
}

/* READ_PRIORS, slam.cpp:913-917 */
int orc_read_priors(orc_ctx* o, gbp_priors_out* out) {
  OUT(out->cam_priors_eta, o->cpe, (size_t)o->C * 6); OUT(out->cam_priors_lambda, o->cpl, (size_t)o->C * 36);
  OUT(out->lmk_priors_eta, o->lpe, (size_t)o->L * 3); OUT(out->lmk_priors_lambda, o->lpl, (size_t)o->L * 9);
  return 0;
}

/* NEW_KEYFRAME, slam.cpp:919-928 */
int orc_new_keyframe(orc_ctx* o, const gbp_kf_update* u) {
  if (u->damping_count) memcpy(o->count, u->damping_count, (size_t)o->E * 4);
  if (u->cam_priors_eta) memcpy(o->cpe, u->cam_priors_eta, (size_t)o->C * 6 * 4);
  if (u->cam_priors_lambda) memcpy(o->cpl, u->cam_priors_lambda, (size_t)o->C * 36 * 4);
  if (u->lmk_priors_eta) memcpy(o->lpe, u->lmk_priors_eta, (size_t)o->L * 3 * 4);
  if (u->lmk_priors_lambda) memcpy(o->lpl, u->lmk_priors_lambda, (size_t)o->L * 9 * 4);
  if (u->active_flag) memcpy(o->active, u->active_flag, (size_t)o->E * 4);
  if (u->cam_weaken_flag) memcpy(o->cwf, u->cam_weaken_flag, (size_t)o->C * 4);
  if (u->lmk_weaken_flag) memcpy(o->lwf, u->lmk_weaken_flag, (size_t)o->L * 4);
  update_beliefs(o);
  return 0;
}

int orc_get_factor_potentials(orc_ctx* o, float* eta, float* lam) {
  OUT(eta, o->fac_eta, (size_t)o->E * 9); OUT(lam, o->fac_lam, (size_t)o->E * 81);
  return 0;
}
int orc_get_messages(orc_ctx* o, float* ce, float* cl, float* le, float* ll) {
  OUT(ce, o->mce, (size_t)o->E * 6); OUT(cl, o->mcl, (size_t)o->E * 36);
  OUT(le, o->mle, (size_t)o->E * 3); OUT(ll, o->mll, (size_t)o->E * 9);
  return 0;
}
int orc_get_mu(orc_ctx* o, float* mu, float* dmu) {
  OUT(mu, o->mu, (size_t)o->E * 9); OUT(dmu, o->dmu, o->E);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Rank-local split-phase view (mirrors gbp_iterate_begin/_end etc. of the C-ABI) so that the
 * multi-process host logic (gbp_poplar_amd/distributed.py) can be exercised on CPU with gloo.
 * Camera record layout of the exchange buffers: 44 floats = eta 6, pad 2, Lambda 36.
 * ------------------------------------------------------------------------------------------ */
int orc_set_shard(orc_ctx* o, int rank, int world, uint32_t lb, uint32_t le) {
  if (rank < 0 || rank >= world || lb > le || le > o->L) return GBP_ERR_INVALID;
  o->sh_rank = rank; o->sh_world = world; o->lb = lb; o->le = le;
  return 0;
}

static void local_cam_partials(const orc_ctx* o, float* send) {
  long c;
#pragma omp parallel for schedule(static)
  for (c = 0; c < (long)o->C; ++c) {
    int k;
    float* rec = send + c * 44;
    rec[6] = 0.f; rec[7] = 0.f;
    for (k = 0; k < 42; ++k) rec[k < 6 ? k : k + 2] = cam_partial(o, (uint32_t)c, k, o->lb, o->le);
  }
}

static void beliefs_from_gathered(orc_ctx* o, const float* recv) {
  long c, l;
#pragma omp parallel for schedule(static)
  for (c = 0; c < (long)o->C; ++c) {
    int k, r;
    for (k = 0; k < 42; ++k) {
      float acc = k < 6 ? o->cpe[c * 6 + k] : o->cpl[c * 36 + (k - 6)];
      for (r = 0; r < o->sh_world; ++r) acc = acc + recv[((size_t)r * o->C + c) * 44 + (k < 6 ? k : k + 2)];
      if (k < 6) o->cbe[c * 6 + k] = acc; else o->cbl[c * 36 + (k - 6)] = acc;
    }
  }
#pragma omp parallel for schedule(static)
  for (l = (long)o->lb; l < (long)o->le; ++l) {
    float* be = o->lbe + l * 3;
    float* bl = o->lbl + l * 9;
    uint32_t s;
    int k;
    for (k = 0; k < 3; ++k) be[k] = o->lpe[l * 3 + k];
    for (k = 0; k < 9; ++k) bl[k] = o->lpl[l * 9 + k];
    for (s = o->lmk_ptr[l]; s < o->lmk_ptr[l + 1]; ++s) {
      uint32_t e = o->lmk_f[s];
      for (k = 0; k < 3; ++k) be[k] += o->mle[(size_t)e * 3 + k];
      for (k = 0; k < 9; ++k) bl[k] += o->mll[(size_t)e * 9 + k];
    }
  }
}

int orc_refresh_begin(orc_ctx* o, float* send) { local_cam_partials(o, send); return 0; }
int orc_refresh_end(orc_ctx* o, const float* recv) { beliefs_from_gathered(o, recv); return 0; }

int orc_linearise_factors(orc_ctx* o) {
  long e;
#pragma omp parallel for schedule(static)
  for (e = 0; e < (long)o->E; ++e) if (is_local(o, (uint32_t)e)) relinearise_zero(o, (uint32_t)e);
  return 0;
}

int orc_iterate_begin(orc_ctx* o, float* send) {
  long e;
  size_t E = o->E;
#pragma omp parallel for schedule(static)
  for (e = 0; e < (long)E; ++e) if (is_local(o, (uint32_t)e)) prep_factor(o, (uint32_t)e);
  memcpy(o->oldmu, o->mu, E * 9 * 4);
#pragma omp parallel for schedule(static)
  for (e = 0; e < (long)E; ++e) {
    if (!is_local(o, (uint32_t)e)) continue;
    msg_cam_eta(o, (uint32_t)e); msg_lmk_eta(o, (uint32_t)e);
    msg_cam_lambda(o, (uint32_t)e); msg_lmk_lambda(o, (uint32_t)e);
  }
  local_cam_partials(o, send);
  return 0;
}

int orc_iterate_end(orc_ctx* o, const float* recv) {
  size_t E = o->E;
  beliefs_from_gathered(o, recv);
  memcpy(o->pce, o->mce, E * 6 * 4);  memcpy(o->pcl, o->mcl, E * 36 * 4);
  memcpy(o->ple, o->mle, E * 3 * 4);  memcpy(o->pll, o->mll, E * 9 * 4);
  return 0;
}

/* weaken priors without a fresh exchange: partials are unchanged, only priors move (as the C-ABI does) */
int orc_weaken_priors_sharded(orc_ctx* o, const float* recv) {
  uint32_t v;
  for (v = 0; v < o->C; ++v) weaken_var(o->cscale[v], &o->cwf[v], o->cpe + (size_t)v * 6, 6, o->cpl + (size_t)v * 36, 36);
  for (v = 0; v < o->L; ++v) weaken_var(o->lscale[v], &o->lwf[v], o->lpe + (size_t)v * 3, 3, o->lpl + (size_t)v * 9, 9);
  beliefs_from_gathered(o, recv);
  return 0;
}

/* NEW_KEYFRAME on a shard: the same re-upload as orc_new_keyframe, then beliefs from the partials of the last
 * exchange (factor messages are untouched by the program, slam.cpp:919-928, so the partials are still valid) */
int orc_new_keyframe_sharded(orc_ctx* o, const gbp_kf_update* u, const float* recv) {
  if (u->damping_count) memcpy(o->count, u->damping_count, (size_t)o->E * 4);
  if (u->cam_priors_eta) memcpy(o->cpe, u->cam_priors_eta, (size_t)o->C * 6 * 4);
  if (u->cam_priors_lambda) memcpy(o->cpl, u->cam_priors_lambda, (size_t)o->C * 36 * 4);
  if (u->lmk_priors_eta) memcpy(o->lpe + (size_t)o->lb * 3, u->lmk_priors_eta + (size_t)o->lb * 3, (size_t)(o->le - o->lb) * 3 * 4);
  if (u->lmk_priors_lambda) memcpy(o->lpl + (size_t)o->lb * 9, u->lmk_priors_lambda + (size_t)o->lb * 9, (size_t)(o->le - o->lb) * 9 * 4);
  if (u->active_flag) memcpy(o->active, u->active_flag, (size_t)o->E * 4);
  if (u->cam_weaken_flag) memcpy(o->cwf, u->cam_weaken_flag, (size_t)o->C * 4);
  if (u->lmk_weaken_flag) memcpy(o->lwf, u->lmk_weaken_flag, (size_t)o->L * 4);
  beliefs_from_gathered(o, recv);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Host-side restatements (Eigen-free).
 * ------------------------------------------------------------------------------------------ */

/* BALProblem::LoadFile, dataio.cpp:17-57 */
int orc_bal_read_header(const char* path, gbp_bal* h) {
  FILE* f = fopen(path, "r");
  int c, l, e;
  if (!f) return GBP_ERR_IO;
  if (fscanf(f, "%d %d %d", &c, &l, &e) != 3) { fclose(f); return GBP_ERR_IO; }
  if (fscanf(f, "%lf %lf %lf %lf", &h->fx, &h->fy, &h->cx, &h->cy) != 4) { fclose(f); return GBP_ERR_IO; }
  h->n_cams = (uint32_t)c; h->n_lmks = (uint32_t)l; h->n_edges = (uint32_t)e;
  fclose(f);
  return 0;
}
int orc_bal_read(const char* path, gbp_bal* b) {
  FILE* f = fopen(path, "r");
  int c, l, e, i, ok = 1;
  if (!f) return GBP_ERR_IO;
  if (fscanf(f, "%d %d %d", &c, &l, &e) != 3) ok = 0;
  if (ok && fscanf(f, "%lf %lf %lf %lf", &b->fx, &b->fy, &b->cx, &b->cy) != 4) ok = 0;
  if (ok && ((uint32_t)c != b->n_cams || (uint32_t)l != b->n_lmks || (uint32_t)e != b->n_edges)) ok = 0;
  for (i = 0; ok && i < e; ++i) {
    int ci, li;
    if (fscanf(f, "%d %d %lf %lf", &ci, &li, &b->observations[2 * i], &b->observations[2 * i + 1]) != 4) ok = 0;
    b->cam_id[i] = (uint32_t)ci; b->lmk_id[i] = (uint32_t)li;
  }
  for (i = 0; ok && i < 6 * c; ++i) if (fscanf(f, "%lf", &b->cameras[i]) != 1) ok = 0;
  for (i = 0; ok && i < 3 * l; ++i) if (fscanf(f, "%lf", &b->points[i]) != 1) ok = 0;
  fclose(f);
  return ok ? 0 : GBP_ERR_IO;
}

/* eigenso3exp, util.cpp:20-32: R = I + (sin t / t) W + ((1 - cos t)/t^2) W W, one expression */
static void eig_so3exp(const float* w, float* R) {
  float th = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  float W[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
  float a, b;
  int i, j, k;
  for (i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.f : 0.f;
  if (th < 1e-6) return;
  a = sinf(th) / th;
  b = (1 - cosf(th)) / (th * th);
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      float ww = 0.f;
      for (k = 0; k < 3; ++k) ww += W[i * 3 + k] * W[k * 3 + j];
      R[i * 3 + j] = R[i * 3 + j] + (a * W[i * 3 + j] + b * ww);
    }
}

/* max |J_ij| of reprojectionJacFn, util.cpp:48-72 */
static float reproj_jac_maxabs(const float* cam, const float* lmk, const float* K) {
  float R[9], Ry[3], pc[3], p[3], jp[6], jK[6], D[9], J2[6], J3[6], m = 0.f;
  int i, j, k;
  eig_so3exp(cam + 3, R);
  for (i = 0; i < 3; ++i) Ry[i] = (R[i * 3] * lmk[0] + R[i * 3 + 1] * lmk[1]) + R[i * 3 + 2] * lmk[2];
  for (i = 0; i < 3; ++i) pc[i] = Ry[i] + cam[i];
  for (i = 0; i < 3; ++i) p[i] = (K[i * 3] * pc[0] + K[i * 3 + 1] * pc[1]) + K[i * 3 + 2] * pc[2];
  jp[0] = 1 / p[2]; jp[1] = 0; jp[2] = (float)(-p[0] / ((double)p[2] * (double)p[2]));
  jp[3] = 0; jp[4] = 1 / p[2]; jp[5] = (float)(-p[1] / ((double)p[2] * (double)p[2]));
  for (i = 0; i < 2; ++i)
    for (j = 0; j < 3; ++j) {
      float s = 0.f;
      for (k = 0; k < 3; ++k) s += jp[i * 3 + k] * K[k * 3 + j];
      jK[i * 3 + j] = s;
    }
  D[0] = -0.f; D[1] = Ry[2]; D[2] = -Ry[1];
  D[3] = -Ry[2]; D[4] = -0.f; D[5] = Ry[0];
  D[6] = Ry[1]; D[7] = -Ry[0]; D[8] = -0.f;
  for (i = 0; i < 2; ++i)
    for (j = 0; j < 3; ++j) {
      float s = 0.f, t = 0.f;
      for (k = 0; k < 3; ++k) { s += jK[i * 3 + k] * D[k * 3 + j]; t += jK[i * 3 + k] * R[k * 3 + j]; }
      J2[i * 3 + j] = s; J3[i * 3 + j] = t;
    }
  for (i = 0; i < 6; ++i) {
    if (fabsf(jK[i]) > m) m = fabsf(jK[i]);
    if (fabsf(J2[i]) > m) m = fabsf(J2[i]);
    if (fabsf(J3[i]) > m) m = fabsf(J3[i]);
  }
  return m;
}

/* set_prior_lambda, dataio.cpp:67-117, as one O(E) max-reduction */
int orc_set_prior_lambda(const gbp_problem* p, float var, const float* cam_file, const float* lmk_file,
                         const float* cam_mean, const float* lmk_mean, float* ce, float* cl,
                         float* le, float* ll) {
  uint32_t C = p->n_cams, L = p->n_lmks, E = p->n_edges, e, v;
  float* mc = (float*)zalloc(C, 4);
  float* ml = (float*)zalloc(L, 4);
  int i;
  for (e = 0; e < E; ++e) {
    uint32_t c = p->cam_id[e], l = p->lmk_id[e];
    float m = reproj_jac_maxabs(cam_file + (size_t)c * 6, lmk_file + (size_t)l * 3, p->K);
    if (m > mc[c]) mc[c] = m;
    if (m > ml[l]) ml[l] = m;
  }
  memset(cl, 0, (size_t)C * 36 * 4); memset(ll, 0, (size_t)L * 9 * 4);
  for (v = 0; v < C; ++v) {
    float lam = (float)(((double)mc[v] * (double)mc[v]) / (double)var);
    for (i = 0; i < 6; ++i) { ce[(size_t)v * 6 + i] = cam_mean[(size_t)v * 6 + i] * lam; cl[(size_t)v * 36 + i * 7] = lam; }
  }
  for (v = 0; v < L; ++v) {
    float lam = (float)(((double)ml[v] * (double)ml[v]) / (double)var);
    for (i = 0; i < 3; ++i) { le[(size_t)v * 3 + i] = lmk_mean[(size_t)v * 3 + i] * lam; ll[(size_t)v * 9 + i * 4] = lam; }
  }
  free(mc); free(ml);
  return 0;
}

/* ba.cpp:561-572 — note the mixed float/double arithmetic: pow(float,int) is double */
int orc_prior_scalings(uint32_t C, uint32_t L, const float* cpl, float steps, float weaker,
                       float first_std, float* cs, float* ls) {
  uint32_t v;
  for (v = 0; v < C; ++v) {
    if (v == 0 || v == 1) cs[v] = (float)exp(-1 / steps * log(cpl[(size_t)v * 36] * pow((double)first_std, 2)));
    else cs[v] = expf(-2 / steps * logf(weaker));
  }
  for (v = 0; v < L; ++v) ls[v] = expf(-2 / steps * logf(weaker));
  return 0;
}

/* create_flags, dataio.cpp:455-475 */
int orc_slam_create_flags(const gbp_problem* p, uint32_t steps, uint32_t* active, uint32_t* cwf,
                          uint32_t* lwf, uint32_t* laf) {
  uint32_t e;
  cwf[0] = steps; cwf[1] = steps;
  for (e = 0; e < p->n_edges; ++e)
    if (p->cam_id[e] == 0 || p->cam_id[e] == 1) { active[e] = 1; lwf[p->lmk_id[e]] = steps; }
  for (e = 0; e < p->n_lmks; ++e) laf[e] = lwf[e];
  return 0;
}

/* update_flags, dataio.cpp:477-508 */
int orc_slam_update_flags(const gbp_problem* p, uint32_t steps, uint32_t dc, uint32_t* active,
                          uint32_t* lwf, uint32_t* cwf, uint32_t* laf, int32_t* n_new) {
  uint32_t e, v;
  int n = 0;
  for (e = 0; e < p->n_edges; ++e) {
    if (p->cam_id[e] == dc + 1) active[e] = 1;
    if (p->cam_id[e] <= dc + 1) lwf[p->lmk_id[e]] = steps;
  }
  for (v = 0; v < p->n_cams; ++v) cwf[v] = 0;
  cwf[dc + 1] = steps;
  for (v = 0; v < p->n_lmks; ++v) { lwf[v] -= laf[v]; laf[v] += lwf[v]; }
  for (v = 0; v < p->n_lmks; ++v) n += (int)lwf[v];
  n /= (int)steps;
  if (n_new) *n_new = n;
  return 0;
}

/* x = A^-1 b in fp64 with partial pivoting on fp32 inputs (stands in for Eigen's .inverse(),
 * util.cpp:104,108,191 — Eigen is absent and its version unpinned) */
static void solve_f64(const float* A, const float* b, int n, float* x) {
  double M[6][7];
  int i, j, k;
  for (i = 0; i < n; ++i) { for (j = 0; j < n; ++j) M[i][j] = A[i * n + j]; M[i][n] = b[i]; }
  for (k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(M[k][k]);
    for (i = k + 1; i < n; ++i) if (fabs(M[i][k]) > best) { best = fabs(M[i][k]); piv = i; }
    if (piv != k) for (j = 0; j <= n; ++j) { double t = M[k][j]; M[k][j] = M[piv][j]; M[piv][j] = t; }
    for (i = k + 1; i < n; ++i) {
      double f = M[i][k] / M[k][k];
      for (j = k; j <= n; ++j) M[i][j] -= f * M[k][j];
    }
  }
  for (i = n - 1; i >= 0; --i) {
    double s = M[i][n];
    for (j = i + 1; j < n; ++j) s -= M[i][j] * (double)x[j];
    x[i] = (float)(s / M[i][i]);
  }
}

/* initialise_new_kf, util.cpp:183-197 (camera part; the landmark part is dead code, util.cpp:215) */
int orc_slam_initialise_new_kf(uint32_t dc, const float* cbe, const float* cbl, const float* cpl, float* cpe) {
  float mu[6];
  int i, k;
  solve_f64(cbl + (size_t)dc * 36, cbe + (size_t)dc * 6, 6, mu);
  for (i = 0; i < 6; ++i) {
    float s = 0.f;
    for (k = 0; k < 6; ++k) s += cpl[(size_t)(dc + 1) * 36 + i * 6 + k] * mu[k];
    cpe[(size_t)(dc + 1) * 6 + i] = s;
  }
  return 0;
}

/* residual of one edge given the variable means: util.cpp:110-127 */
static void edge_residual(const float* cmu, const float* lmu, const float* K, const float* z,
                          float* norm, float* half_sq) {
  float R[9], pcf[3], pr[3], r0, r1;
  int i;
  eig_so3exp(cmu + 3, R);
  for (i = 0; i < 3; ++i) pcf[i] = (R[i * 3] * lmu[0] + R[i * 3 + 1] * lmu[1]) + R[i * 3 + 2] * lmu[2];
  for (i = 0; i < 3; ++i) pcf[i] += cmu[i];
  for (i = 0; i < 3; ++i) pr[i] = ((K[i * 3] * pcf[0] + K[i * 3 + 1] * pcf[1]) + K[i * 3 + 2] * pcf[2]) / pcf[2];
  r0 = z[0] - pr[0]; r1 = z[1] - pr[1];
  *norm = sqrtf(r0 * r0 + r1 * r1);
  *half_sq = (float)(0.5 * (r0 * r0 + r1 * r1));
}

static void var_means(uint32_t C, uint32_t L, const float* cbe, const float* cbl, const float* lbe,
                      const float* lbl, float* cmu, float* lmu) {
  long v;
#pragma omp parallel for schedule(static)
  for (v = 0; v < (long)C; ++v) solve_f64(cbl + v * 36, cbe + v * 6, 6, cmu + v * 6);
#pragma omp parallel for schedule(static)
  for (v = 0; v < (long)L; ++v) solve_f64(lbl + v * 9, lbe + v * 3, 3, lmu + v * 3);
}

/* eval_reprojection_error, util.cpp:74-144, with fp64 accumulation; "active edges" instead of the
 * reference's "first n_active edges" (identical for camera-sorted files, util.cpp:95-99) */
static int eval_range(const gbp_problem* p, const uint32_t* active, const float* meas, const float* cbe,
                      const float* cbl, const float* lbe, const float* lbl, uint32_t lmk_lo, uint32_t lmk_hi,
                      double* sn, double* sh, uint64_t* na) {
  float* cmu = (float*)zalloc((size_t)p->n_cams * 6, 4);
  float* lmu = (float*)zalloc((size_t)p->n_lmks * 3, 4);
  double a = 0, b = 0;
  uint64_t n = 0;
  uint32_t e;
  var_means(p->n_cams, p->n_lmks, cbe, cbl, lbe, lbl, cmu, lmu);
  for (e = 0; e < p->n_edges; ++e) {
    float nr, hs;
    if (active[e] != 1) continue;
    if (lmk_lo != lmk_hi && (p->lmk_id[e] < lmk_lo || p->lmk_id[e] >= lmk_hi)) continue;
    edge_residual(cmu + (size_t)p->cam_id[e] * 6, lmu + (size_t)p->lmk_id[e] * 3, p->K, meas + (size_t)e * 2, &nr, &hs);
    a += nr; b += hs; ++n;
  }
  *sn = a; *sh = b; *na = n;
  free(cmu); free(lmu);
  return 0;
}

int orc_eval_host(const gbp_problem* p, const uint32_t* active, const float* meas, const float* cbe,
                  const float* cbl, const float* lbe, const float* lbl, double* sn, double* sh, uint64_t* na) {
  return eval_range(p, active, meas, cbe, cbl, lbe, lbl, 0, 0, sn, sh, na);
}

int orc_eval_host_f32(const gbp_problem* p, const uint32_t* active, const float* meas, const float* cbe,
                      const float* cbl, const float* lbe, const float* lbl, float* reproj) {
  float* cmu = (float*)zalloc((size_t)p->n_cams * 6, 4);
  float* lmu = (float*)zalloc((size_t)p->n_lmks * 3, 4);
  uint32_t e, n = 0;
  var_means(p->n_cams, p->n_lmks, cbe, cbl, lbe, lbl, cmu, lmu);
  reproj[0] = 0.f; reproj[1] = 0.f;
  for (e = 0; e < p->n_edges; ++e) n += active[e];
  for (e = 0; e < n && e < p->n_edges; ++e) { /* first n_active edges, util.cpp:99 */
    float nr, hs;
    edge_residual(cmu + (size_t)p->cam_id[e] * 6, lmu + (size_t)p->lmk_id[e] * 3, p->K, meas + (size_t)e * 2, &nr, &hs);
    reproj[0] += nr; reproj[1] += hs;
  }
  reproj[0] /= n;
  free(cmu); free(lmu);
  return 0;
}

/* un-pivoted LDL^T pivots of the lower triangle (the test inv6x6/inv3x3 never make) in fp64 */
static int ldl_positive(const float* A, int n) {
  double Lm[6][6], D[6];
  int i, j, k, ok = 1;
  for (j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (k = 0; k < j; ++k) d -= Lm[j][k] * Lm[j][k] * D[k];
    D[j] = d;
    if (!(d > 0.0)) ok = 0;
    for (i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (k = 0; k < j; ++k) v -= Lm[i][k] * Lm[j][k] * D[k];
      Lm[i][j] = v / d;
    }
  }
  return ok;
}

int orc_eval(orc_ctx* o, gbp_eval_out* out) {
  gbp_problem p;
  uint32_t e;
  size_t i;
  memset(&p, 0, sizeof(p));
  p.n_cams = o->C; p.n_lmks = o->L; p.n_edges = o->E; p.cam_id = o->cam_id; p.lmk_id = o->lmk_id;
  memcpy(p.K, o->K, sizeof(p.K));
  memset(out, 0, sizeof(*out));
  eval_range(&p, o->active, o->meas, o->cbe, o->cbl, o->lbe, o->lbl, o->sh_world > 1 ? o->lb : 0,
             o->sh_world > 1 ? o->le : 0, &out->sum_norm, &out->sum_half_sq, &out->n_active);
  for (e = 0; e < o->E; ++e) {
    if (!is_local(o, e)) continue;
    out->n_robust += o->robust[e];
    if (o->count[e] == -o->prm.num_undamped_iters) out->n_relin++;
  }
  for (i = 0; i < (size_t)o->C; ++i) {
    int k, bad = 0;
    for (k = 0; k < 6; ++k) if (!isfinite(o->cbe[i * 6 + k])) bad = 1;
    for (k = 0; k < 36; ++k) if (!isfinite(o->cbl[i * 36 + k])) bad = 1;
    out->n_nonfinite += bad;
    out->n_nonpd += !ldl_positive(o->cbl + i * 36, 6);
  }
  for (i = 0; i < (size_t)o->L; ++i) {
    int k, bad = 0;
    for (k = 0; k < 3; ++k) if (!isfinite(o->lbe[i * 3 + k])) bad = 1;
    for (k = 0; k < 9; ++k) if (!isfinite(o->lbl[i * 9 + k])) bad = 1;
    out->n_nonfinite += bad;
    if (o->sh_world == 1 || (i >= o->lb && i < o->le)) out->n_nonpd += !ldl_positive(o->lbl + i * 9, 3);
  }
  return 0;
}
