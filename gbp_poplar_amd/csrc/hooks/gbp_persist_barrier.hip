// gbp_persist_barrier.hip — k_persist<EV>: the persistent kernel of rounds 3-4, two counter barriers per iteration.
//
// Not part of the product since round 5 (k_persist_flow replaced it: hand-offs through tagged records).  Kept in the TEST-HOOKS build
// (included by gbp_kernels.hip under GBP_BUILD_TEST_HOOKS, at the place it used to stand: it uses the helpers above it) as the
// reference the parity tests compare the new kernel with and for A/B measurements (gbp_debug_persist_flow(ctx, 0),
// profiles/time_bursts.py flow=0).

// EV: does the launch carry the metric (A.ev.on)?  A launch without it runs an instantiation that holds none of the metric's
// code or registers (plain bursts 14.3 -> 14.0 us per iteration on fr1xyz).
template <bool EV>
__global__ __launch_bounds__(256) void k_persist(const PersistArgs A) {
  const bool ev_on = EV && A.ev.on != 0;
  const SweepArgs& a = A.s;
  const BeliefArgs& b = A.b;
  const uint32_t wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // an earlier launch of this ctx gave up at a barrier: the state is not what this launch expects — touch nothing (the host
  // restores the snapshot and replays; every workgroup reads the same word before anyone could write it in THIS launch)
  if (__hip_atomic_load(A.sync + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  // placement (profiles/r03_small_graphs.md): the grid is `spread` times larger than the work; filler workgroups leave at once.
  // spread > 0: workgroup b works iff b % spread == 0;  spread < 0 (s = -spread): iff (b / 8) % s == 0 (every XCD keeps working,
  // every s-th dispatch slot inside an XCD)
  uint32_t bid = blockIdx.x, nblk = gridDim.x;
  if ((int)A.spread > 1) {
    if (blockIdx.x % A.spread) return;
    bid = blockIdx.x / A.spread; nblk = gridDim.x / A.spread;
  } else if ((int)A.spread < -1) {
    const uint32_t sp = (uint32_t)(-(int)A.spread), slot = blockIdx.x >> 3;
    if (slot % sp) return;
    bid = (slot / sp) * 8 + (blockIdx.x & 7u); nblk = A.n_work_blocks;
    if (bid >= nblk) return;
  }
  const uint32_t w = bid * 4 + wib;                           // wave of the grid
  __shared__ float4 lm_stage[4][64 * 4];
  __shared__ float sh[4][48];
  float4* stage = lm_stage[wib];
  const XwBuf X_lmsg(a.lmsg), X_rowp(a.rowp), X_camb(a.camb), X_lmkb(a.lmkb), X_cmu(a.cam_mu), X_lmu(a.lmk_mu), X_clin(a.cam_lin);
  const XwBuf X_emc(A.ev.cam_mu), X_eml(A.ev.lmk_mu);     // metric means (only with A.ev.on)

  // ---- phase-A role: sweep tile w.  State that only this lane ever touches lives in registers for the whole launch.
  const bool has_tile = w < A.n_tiles;
  const uint32_t tile = has_tile ? w : 0u, p = tile * 64 + lane;
  const uint32_t rec_t = lane >> 2, swz_own = ((lane >> 2) & 3u) ^ (lane & 2u);
  const uint32_t lm_tile4 = tile * 256u;                        // first float4 of the wave's 64 landmark-message records
  float fac[56], cm[28], lm[16];
  uint32_t cam_i = 0, lmk_i = 0;
  bool fac_dirty = false;
  if (has_tile) {
    cam_i = a.row_cam[p >> 4];
    lmk_i = a.lmk_idx[p];
    load_tile<kFacG, false>(a.fac, tile, lane, fac);
    load_tile<kCmsgG, false>(a.cmsg, tile, lane, cm);
    GBP_UNROLL
    for (int k = 0; k < 4; ++k) {   // the wave's 64 landmark-message records: coalesced, transposed through LDS (see k_sweep)
      const uint32_t r = k * 16 + rec_t;
      stage[r * 4 + ((lane & 3u) ^ (((r >> 2) & 3u) ^ (r & 2u)))] = X_lmsg.ld4(lm_tile4 + (uint32_t)k * 64u + lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    GBP_UNROLL
    for (int q = 0; q < 4; ++q) {
      const float4 v = stage[lane * 4 + ((uint32_t)q ^ swz_own)];
      lm[4 * q] = v.x; lm[4 * q + 1] = v.y; lm[4 * q + 2] = v.z; lm[4 * q + 3] = v.w;
    }
  }
  float K[9];
  GBP_UNROLL
  for (int i = 0; i < 9; ++i) K[i] = a.K[i];
  const uint32_t cb_rec4 = cam_i * (uint32_t)kCamRec4, lb_rec4 = lmk_i * (uint32_t)kLmkRec4;   // loop-invariant: phase A is ONE round of loads
  const uint32_t cmu_rec4 = cam_i * 4u, lmu_rec4 = lmk_i * 2u, clin_rec4 = cam_i * (uint32_t)kCamLin4;

  // ---- phase-B role: camera v (lanes 0..43 = the record), or landmarks 16 (v - C) .. + 15 (4 lanes each).  Roles are
  // numbered ACROSS the workgroups (v = wave-in-workgroup * workgroups + workgroup): the camera waves, whose lane 0 runs long
  // serial fp64 chains when the metric rides along, land one per CU instead of four
  // Third kind of role, used only while the metric rides in the launch: waves [C + G, 2C + G) — where the grid has them
  // (persist_blocks) — take the METRIC mean of camera v - (C + G): they sum the camera's rows themselves (same loads, same
  // order: same belief) and run the fp64 pivoted solve + the fp64 LDL check that would otherwise sit behind the hoisted mean
  // and CAM_LIN on the camera wave's single working lane (belief phase of a camera wave 6.1 us, of everyone else <= 4.0 us:
  // profiles/r04_small_graphs.md).
  const uint32_t v = wib * nblk + bid;
  const bool cam_wave = v < b.n_cams;
  const bool lmk_wave = !cam_wave && (v - b.n_cams) < A.n_lmk_groups;
  const uint32_t v_met0 = b.n_cams + A.n_lmk_groups;
  const bool met_wave = ev_on && v >= v_met0 && v - v_met0 < b.n_cams;
  const bool cam_has_met_wave = cam_wave && v_met0 + v < nblk * 4u;       // this camera's metric mean is solved by wave v_met0 + v
  const uint32_t camv = met_wave ? v - v_met0 : v;                          // the camera of either role
  const uint32_t cj = lane;                                   // camera role: element of the 44-float record
  const bool cam_live = (cam_wave || met_wave) && cj < (uint32_t)kCamRec;
  uint32_t r0 = 0, r1 = 0;
  float cam_prior_j = 0.f;
  float4 cam_cur0 = make_float4(0.f, 0.f, 0.f, 0.f), cam_cur1 = cam_cur0;   // mean of the belief the next sweep consumes
  if (cam_wave || met_wave) {
    r0 = b.cam_row_ptr[camv]; r1 = b.cam_row_ptr[camv + 1];
    if (cam_live) cam_prior_j = b.cam_prior[(size_t)camv * kCamRec + cj];
  }
  if (cam_wave) {
    // (sc1 like EVERY access of this launch to an array that crosses waves: a plain load could leave a copy in this XCD's L2
    // that goes stale when another XCD rewrites the neighbouring half of the 128-B line)
    cam_cur0 = X_cmu.ld4(v * 4u); cam_cur1 = X_cmu.ld4(v * 4u + 1u);
  }
  const uint32_t l = lmk_wave ? (v - b.n_cams) * 16 + (lane >> 2) : 0u, q4 = lane & 3;
  const bool lmk_live = lmk_wave && l < b.n_lmks;
  uint4 ix = make_uint4(0u, 0u, 0u, 0u);
  float4 lmk_prior4 = make_float4(0.f, 0.f, 0.f, 0.f), lmk_cur = lmk_prior4;
  uint32_t lp0 = 0, lp1 = 0;
  if (lmk_live) {
    ix = reinterpret_cast<const uint4*>(b.lmk_ix)[(size_t)l * 4 + q4];
    lmk_prior4 = b.lmk_prior[(size_t)l * 4 + q4];
    lmk_cur = X_lmu.ld4(l * 2u);
    lp0 = b.lmk_ptr[l]; lp1 = b.lmk_ptr[l + 1];
  }
  const uint32_t deg = (uint32_t)__shfl((int)ix.x, 0, 4);
  uint32_t pos[15];
  GBP_UNROLL
  for (int k = 0; k < 15; ++k) {   // element k + 1 of the index record sits in lane (k + 1) / 4, component (k + 1) % 4
    const uint32_t v = ((k + 1) & 3) == 0 ? ix.x : ((k + 1) & 3) == 1 ? ix.y : ((k + 1) & 3) == 2 ? ix.z : ix.w;
    pos[k] = (uint32_t)__shfl((int)v, (k + 1) >> 2, 4);
  }

  // slots 16 .. 30 of a landmark (fr1xyz: up to 30 factors per landmark): positions fetched ONCE, so that phase B stays a
  // single round of loads; slots beyond 30 go through lmk_fpos every iteration
  uint32_t pos2[15];
  GBP_UNROLL
  for (int k = 0; k < 15; ++k) {   // unconditional loads (slot clamped, value dropped): fifteen conditional ones were fifteen round trips per launch
    const uint32_t p2 = b.lmk_fpos[lp0 + (15u + (uint32_t)k < deg ? 15u + (uint32_t)k : 0u)];
    pos2[k] = (lmk_live && 15u + (uint32_t)k < deg) ? p2 : 0u;
  }

  // ---- the metric (gbp_iterate_eval / gbp_iterate_eval_each): what k_means + k_eval compute, same bits.  The belief owners
  // write the metric means in phase B; after the next device-wide hand-off every tile wave adds its factors' residuals and the
  // workgroup reduces them in k_eval's order (a workgroup holds the same 256 positions as a block of k_eval).  `packed` is the
  // factor's state word as the sweep of the evaluated iteration left it.
  // Where the pieces of metric k live: the belief owners write the metric means of iteration k into half (k & 1) of the two
  // mean buffers during phase B of k; the tile waves evaluate their factors' residuals at the END of phase B of k + 1 — behind
  // their own belief-phase role, where all but the slowest waves have slack (in phase A the same work sat on the iteration's
  // critical path: +2.1 us on fr1xyz) — reading half (k & 1) while the owners write half ((k + 1) & 1).  The health counters of
  // an every-iteration launch alternate the same way.  Each tile wave stores its own partial sums (slot 1 + wave); the host adds
  // the four waves of a workgroup as k_eval's block reduction does, ((w0 + w1) + w2) + w3, then the workgroups in order.
  const uint32_t emc_half = b.n_cams * 6u, eml_half = b.n_lmks * 3u;
  auto health_of = [&](uint32_t k) -> unsigned long long* { return A.ev.each ? A.ev.health_each + 2u * (k & 1u) : A.ev.health; };
  auto metric = [&](uint32_t k, int packed, const float (&cmv)[6], const float (&lmu)[3]) {
    double s_norm = 0, s_half = 0;
    unsigned long long n_act = 0, n_rel = 0, n_rob = 0;
    if (has_tile) {
      const uint32_t flags = (uint32_t)packed & 7u;
      if (!(flags & kFlagPad)) {
        if (flags & kFlagRobust) ++n_rob;
        if ((packed >> 3) == -A.ev.num_undamped) ++n_rel;
        if (flags & kFlagActive) {
          eval_factor(cmv, lmu, fac[54], fac[55], a.K, s_norm, s_half);
          ++n_act;
        }
      }
    }
    DeviceEval* slots = A.ev.slots + (size_t)(A.ev.each ? k : 0u) * A.ev.stride;
    if (has_tile) {     // the lane tree of eval_wave_tree, then one record per wave
      for (int off = 32; off > 0; off >>= 1) {
        s_norm += __shfl_down(s_norm, off);
        s_half += __shfl_down(s_half, off);
        n_act += __shfl_down(n_act, off);
        n_rel += __shfl_down(n_rel, off);
        n_rob += __shfl_down(n_rob, off);
      }
      if (lane == 0) {
        DeviceEval o;
        o.sum_norm = s_norm; o.sum_half_sq = s_half; o.n_active = n_act; o.n_relin = n_rel; o.n_robust = n_rob; o.pad = 0;
        slots[1 + w] = o;
      }
    }
    if (bid == 0 && threadIdx.x == 0) {
      unsigned long long* h = health_of(k);
      unsigned long long* out = reinterpret_cast<unsigned long long*>(slots);
      out[0] = __hip_atomic_load(&h[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out[1] = __hip_atomic_load(&h[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // each: this half counts from zero again when its turn comes (two hand-offs from now); one metric: the area goes back to zero like
      // every other user of the health words leaves it (k_eval in gbp_kernels.hip)
      __hip_atomic_store(&h[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&h[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto metric_means = [&](uint32_t k, float (&cmv)[6], float (&lmu)[3]) {
    const uint32_t oc = (k & 1u) * emc_half + cam_i * 6u, ol_ = (k & 1u) * eml_half + lmk_i * 3u;
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) cmv[i] = X_emc.ld1(oc + (uint32_t)i);
    GBP_UNROLL
    for (int i = 0; i < 3; ++i) lmu[i] = X_eml.ld1(ol_ + (uint32_t)i);
  };

  unsigned epoch = 0;
  for (int it = 0; it < A.n_iters; ++it) {
    const bool ev_means = ev_on && (A.ev.each || it + 1 == A.n_iters);     // this iteration's beliefs are evaluated
    // the metric of iteration it - 1 rides in this phase A (both only READ what phase B left): its loads go out with the
    // sweep's, its arithmetic runs behind the sweep's stores
    const bool ev_prev = ev_on && A.ev.each && it > 0;
    const int ev_packed = __float_as_int(lm[13]);      // the factor's state word as the sweep of iteration it - 1 left it
    // ================= phase A: the sweep of this wave's tile =================
    if (has_tile) {
      float cb[44], lb[16], mu[12];
      // One round of loads per phase: the beliefs AND what a relinearising lane needs (hoisted means, CAM_LIN) go out together.
      // (measured and dropped, profiles/r04_small_graphs.md: fetching the camera's 18 float4 once per 16-lane row and handing
      // them round through LDS instead of 18 loads per lane — 3x fewer L1 accesses — changes nothing: 21.5 vs 21.5 ms)
      const float4 l0 = X_lmu.ld4(lmu_rec4);
      const float4 m0 = X_cmu.ld4(cmu_rec4), m1 = X_cmu.ld4(cmu_rec4 + 1u);
      float4 clq[kCamLin4];
      GBP_UNROLL
      for (int g = 0; g < kCamLin4; ++g) clq[g] = X_clin.ld4(clin_rec4 + (uint32_t)g);
      load_rec_xw<kLmkRec4>(X_lmkb, lb_rec4, lb);
      load_rec_xw<kCamRec4>(X_camb, cb_rec4, cb);
      float damping = lm[3];
      const int packed = __float_as_int(lm[13]);
      int count = packed >> 3;
      uint32_t flags = (uint32_t)packed & 7u;
      const float var = lm[14];
      const bool active = (flags & kFlagActive) != 0;
      float oc_eta[6], oc_lam[36], ol[16];
      bool relin;
      factor_update<true>(fac, cm, mu, lm, cb, lb, K, a.hp, damping, count, flags, var, active, oc_eta, oc_lam, ol, relin,
                             [&](float (&x0c)[6], float (&x0l)[3], CamLin& cl) {
                               x0c[0] = m0.x; x0c[1] = m0.y; x0c[2] = m0.z; x0c[3] = m0.w; x0c[4] = m1.x; x0c[5] = m1.y;
                               x0l[0] = l0.x; x0l[1] = l0.y; x0l[2] = l0.z;
                               cam_lin_unpack(clq, cl);
                             });
      fac_dirty = fac_dirty || (active && relin);
      ol[3] = damping;
      ol[13] = __int_as_float((int)(((uint32_t)count << 3) | flags));
      ol[14] = var;
      // the wave's landmark messages go to memory (phase B gathers them by position); this lane keeps its own copy
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      GBP_UNROLL
      for (int q = 0; q < 4; ++q)
        stage[lane * 4 + ((uint32_t)q ^ swz_own)] = make_float4(ol[4 * q], ol[4 * q + 1], ol[4 * q + 2], ol[4 * q + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      GBP_UNROLL
      for (int k = 0; k < 4; ++k) {
        const uint32_t r = k * 16 + rec_t;
        X_lmsg.st4(lm_tile4 + (uint32_t)k * 64u + lane, stage[r * 4 + ((lane & 3u) ^ (((r >> 2) & 3u) ^ (r & 2u)))]);
      }
      GBP_UNROLL
      for (int i = 0; i < 16; ++i) lm[i] = ol[i];
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) cm[i] = oc_eta[i];
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) {
        GBP_UNROLL
        for (int j = 0; j <= i; ++j) cm[6 + tri(i, j)] = oc_lam[i * 6 + j];
      }
      cm[27] = 0.f;
      {  // camera half of the belief reduction: per-row tree sums, as in k_sweep
        const uint32_t rp4 = (p >> 4) * (uint32_t)kCamRec4;
        row16_sums_store(oc_eta, oc_lam, lane, [&](uint32_t g, float4 v) { X_rowp.st4(rp4 + g, v); });
      }
    }
    grid_sync(A.sync, A.epoch_base + (++epoch) * nblk, A.status, A.seq);

    // ================= phase B: the belief update (arithmetic of k_beliefs, roll = 1) =================
    float ev_cm[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ev_lm[3] = {0.f, 0.f, 0.f};
    if (ev_prev && has_tile) metric_means((uint32_t)it - 1u, ev_cm, ev_lm);     // in flight with the role's own loads
    unsigned long long* const hw = health_of((uint32_t)it);                      // what this phase's owners count into
    const uint32_t emc_w = ((uint32_t)it & 1u) * emc_half, eml_w = ((uint32_t)it & 1u) * eml_half;
    const bool ldl_deferred = it + 1 < A.n_iters;      // a hand-off follows this belief phase
    if (cam_wave || (met_wave && ev_means)) {
      float acc = 0.f;
      if (cam_live && r1 > r0) {
        const uint32_t row = r0 * (uint32_t)kCamRec + cj;           // float index into ROWP
        const uint32_t n = r1 - r0;
        // (measured, profiles/r04_small_graphs.md: all 33 rows of a fr1xyz camera in ONE round of loads, or 17 + 16, are SLOWER than
        // this 1 + 16 + tail shape — 21.5 -> 22.1 / 23.2 ms per 1 500 iterations — so the shape stays)
        acc = X_rowp.ld1(row);
        uint32_t r = 1;
        for (; r + 16 <= n; r += 16) {
          float v[16];
          GBP_UNROLL
          for (int k = 0; k < 16; ++k) v[k] = X_rowp.ld1(row + (r + (uint32_t)k) * (uint32_t)kCamRec);
          GBP_UNROLL
          for (int k = 0; k < 16; ++k) acc = acc + v[k];
        }
        {  // tail (< 16 rows): the loads are UNCONDITIONAL (row index clamped, value dropped) — a conditional atomic load
           // becomes a branch with its own wait, i.e. one memory round trip per row
          float v[16];
          const uint32_t m = n - r;
          GBP_UNROLL
          for (int k = 0; k < 16; ++k) v[k] = X_rowp.ld1(row + ((uint32_t)k < m ? r + (uint32_t)k : n - 1u) * (uint32_t)kCamRec);
          GBP_UNROLL
          for (int k = 0; k < 16; ++k)
            if ((uint32_t)k < m) acc = acc + v[k];
        }
      }
      if (cam_live) {
        if (cam_wave) b.cam_local[(size_t)v * kCamRec + cj] = acc;
        sh[wib][cj] = cam_prior_j + acc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane == 0 && !cam_wave) {   // metric role: what k_means computes for this camera, from the belief in LDS
        float xm[6];                  // (the fp64 pivoted solve only: the fp64 LDL check stays with the camera wave, which has the slack)
        solve_pivot<6>(sh[wib] + 8, 6, sh[wib], xm);
        bool finite = true;
        GBP_UNROLL
        for (int i = 0; i < 6; ++i) { X_emc.st1(emc_w + camv * 6u + (uint32_t)i, xm[i]); finite &= (xm[i] - xm[i] == 0.f); }
        if (!finite) atomicAdd(&hw[0], 1ull);
      }
      if (lane == 0 && cam_wave) {
        const bool ev_here = ev_means && !cam_has_met_wave;     // no wave to spare for this camera's metric mean: solved here
        float cb[44], x0c[6];
        GBP_UNROLL
        for (int i = 0; i < 44; ++i) cb[i] = sh[wib][i];
        // three independent dependent-chains on ONE lane (hoisted mean; with the metric, the fp64 pivoted solve and the fp64
        // LDL pivots): computed together, before any store or branch, so that the scheduler can interleave them
        float xm[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bool pd = true;
        if (ev_here) {   // metric means of this camera (what k_means computes), from the belief in LDS
          solve_pivot<6>(sh[wib] + 8, 6, sh[wib], xm);
          pd = ldl_pivots_positive<6>(sh[wib] + 8, 6);
          cam_mean(cb, x0c);
        } else if (ev_means) {   // the solve runs on this camera's metric wave; the health check here — or, where a hand-off
          // follows, behind the arrival on the NEXT wave of the workgroup (ldl_deferred, below): it feeds nothing in this iteration
          if (!ldl_deferred) pd = ldl_pivots_positive<6>(sh[wib] + 8, 6);
          cam_mean(cb, x0c);
        } else {
          cam_mean(cb, x0c);
        }
        const uint32_t mu4 = v * 4u;            // [0,1] = means of the current belief, [2,3] = means the last sweep used
        X_cmu.st4(mu4 + 2u, cam_cur0); X_cmu.st4(mu4 + 3u, cam_cur1);
        const float used[6] = {cam_cur0.x, cam_cur0.y, cam_cur0.z, cam_cur0.w, cam_cur1.x, cam_cur1.y};
        float S = 0.f;
        GBP_UNROLL
        for (int i = 0; i < 6; ++i) S += (used[i] - x0c[i]) * (used[i] - x0c[i]);
        cam_cur0 = make_float4(x0c[0], x0c[1], x0c[2], x0c[3]);
        cam_cur1 = make_float4(x0c[4], x0c[5], 0.f, 0.f);
        X_cmu.st4(mu4, cam_cur0); X_cmu.st4(mu4 + 1u, cam_cur1);
        {  // camera-only Jacobian terms of the new mean (what k_beliefs stores): the relinearising lanes of the next sweep load them
          CamLin cl;
          const float wv[3] = {x0c[3], x0c[4], x0c[5]};
          cam_lin(wv, cl);
          float4 q[kCamLin4];
          cam_lin_pack(cl, q);
          GBP_UNROLL
          for (int g = 0; g < kCamLin4; ++g) X_clin.st4(v * (uint32_t)kCamLin4 + (uint32_t)g, q[g]);
        }
        if (ev_here) {
          bool finite = true;
          GBP_UNROLL
          for (int i = 0; i < 6; ++i) { X_emc.st1(emc_w + v * 6u + (uint32_t)i, xm[i]); finite &= (xm[i] - xm[i] == 0.f); }
          if (!finite) atomicAdd(&hw[0], 1ull);
        }
        if (ev_means && !pd) atomicAdd(&hw[1], 1ull);
        sh[wib][6] = S;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (cam_live && cam_wave) X_camb.st1(v * (uint32_t)kCamRec + cj, sh[wib][cj]);
    } else if (lmk_wave) {
      float4 acc = lmk_prior4;
      {  // both batches of loads are issued before the first add (one memory round trip for up to 30 slots: the
         // wave has 512 registers per lane to itself); the adds stay in slot order
        float4 m[15], m2[15];
        const bool second = __any(deg > 15u);
        GBP_UNROLL
        for (int k = 0; k < 15; ++k) m[k] = lmsg_piece_xw(X_lmsg, pos[k], q4);   // unconditional: unused slots hold position 0
        if (second) {
          GBP_UNROLL
          for (int k = 0; k < 15; ++k) m2[k] = lmsg_piece_xw(X_lmsg, pos2[k], q4);
        }
        GBP_UNROLL
        for (int k = 0; k < 15; ++k)     // adds in slot order
          if ((uint32_t)k < deg) { acc.x = acc.x + m[k].x; acc.y = acc.y + m[k].y; acc.z = acc.z + m[k].z; acc.w = acc.w + m[k].w; }
        if (second) {
          GBP_UNROLL
          for (int k = 0; k < 15; ++k)
            if (15u + (uint32_t)k < deg) { acc.x = acc.x + m2[k].x; acc.y = acc.y + m2[k].y; acc.z = acc.z + m2[k].z; acc.w = acc.w + m2[k].w; }
        }
      }
      if (deg > 30u) {
        for (uint32_t s = lp0 + 30u; s < lp1; s += 8) {
          uint32_t ps[8];
          float4 m[8];
          const uint32_t nleft = lp1 - s;
          GBP_UNROLL
          for (int k = 0; k < 8; ++k) ps[k] = b.lmk_fpos[(uint32_t)k < nleft ? s + k : lp1 - 1u];     // clamped, unconditional
          GBP_UNROLL
          for (int k = 0; k < 8; ++k) m[k] = lmsg_piece_xw(X_lmsg, ps[k], q4);
          GBP_UNROLL
          for (int k = 0; k < 8; ++k)
            if ((uint32_t)k < nleft) { acc.x = acc.x + m[k].x; acc.y = acc.y + m[k].y; acc.z = acc.z + m[k].z; acc.w = acc.w + m[k].w; }
        }
      }
      float rec[16];
      GBP_UNROLL
      for (int k = 0; k < 4; ++k) {
        rec[4 * k] = __shfl(acc.x, k, 4); rec[4 * k + 1] = __shfl(acc.y, k, 4);
        rec[4 * k + 2] = __shfl(acc.z, k, 4); rec[4 * k + 3] = __shfl(acc.w, k, 4);
      }
      float u[3] = {0.f, 0.f, 0.f};
      if (lmk_live && q4 == 0) {
        float B[9], S3[9], x0l[3];
        GBP_UNROLL
        for (int i = 0; i < 9; ++i) B[i] = rec[4 + i];
        inv3x3(B, S3);
        GBP_UNROLL
        for (int i = 0; i < 3; ++i) {
          float a2 = 0.f;
          GBP_UNROLL
          for (int k = 0; k < 3; ++k) a2 += S3[i * 3 + k] * rec[k];
          x0l[i] = a2;
        }
        const uint32_t mu4 = l * 2u;            // [0] = mean of the current belief, [1] = mean the last sweep used
        const float4 used = lmk_cur;
        X_lmu.st4(mu4 + 1u, used);
        u[0] = (used.x - x0l[0]) * (used.x - x0l[0]);
        u[1] = (used.y - x0l[1]) * (used.y - x0l[1]);
        u[2] = (used.z - x0l[2]) * (used.z - x0l[2]);
        lmk_cur = make_float4(x0l[0], x0l[1], x0l[2], 0.f);
        X_lmu.st4(mu4, lmk_cur);
        if (ev_means) {   // metric mean of this landmark (k_means), from the belief record in registers
          float x[3];
          solve_pivot<3>(rec + 4, 3, rec, x);
          bool finite = true;
          GBP_UNROLL
          for (int i = 0; i < 3; ++i) { X_eml.st1(eml_w + l * 3u + (uint32_t)i, x[i]); finite &= (x[i] - x[i] == 0.f); }
          if (!finite) atomicAdd(&hw[0], 1ull);
          if (!ldl_pivots_positive<3>(rec + 4, 3)) atomicAdd(&hw[1], 1ull);
        }
      }
      const float u0 = __shfl(u[0], 0, 4), u1 = __shfl(u[1], 0, 4), u2 = __shfl(u[2], 0, 4);
      if (q4 == 0) acc.w = u0;                       // record slot 3
      if (q4 == 3) { acc.y = u1; acc.z = u2; }       // record slots 13, 14
      if (lmk_live) X_lmkb.st4(l * 4u + q4, acc);
    }
    // the residuals of the PREVIOUS iteration: behind this wave's role, and — where a hand-off follows — between its arrival and
    // its wait (they read what the previous belief phase left and the factor's own registers: nothing of this phase)
    if (it + 1 < A.n_iters) {
      grid_arrive(A.sync);
      {
        // The fp64 LDL health check a camera wave left out above (with the metric every iteration it is the slowest wave of the
        // belief phase), made by the NEXT wave of the workgroup from the camera's record in LDS (Lambda at sh[.] + 8: untouched
        // until the next belief phase, visible since the arrival's workgroup barrier) — not by the camera wave itself: that is
        // wave 0, which polls the hand-off, and polling from any other wave costs 0.9 us per hand-off (measured; so does the check
        // in front of the polling).  Counted into this iteration's pair of counters, which block 0 reads one iteration from now.
        // fr1xyz, metric every iteration: 14.75 -> 14.60 us per iteration (profiles/r04_alu_diet.md section 3).
        const uint32_t pw = (wib + 3u) & 3u, pv = pw * nblk + bid;
        if (ldl_deferred && ev_means && lane == 0 && pv < b.n_cams && v_met0 + pv < nblk * 4u) {
          if (!ldl_pivots_positive<6>(sh[pw] + 8, 6)) atomicAdd(&hw[1], 1ull);
        }
      }
      if (ev_prev) metric((uint32_t)it - 1u, ev_packed, ev_cm, ev_lm);
      grid_wait(A.sync, A.epoch_base + (++epoch) * nblk, A.status, A.seq);
    } else if (ev_prev) {
      metric((uint32_t)it - 1u, ev_packed, ev_cm, ev_lm);
    }
  }

  // ---- the metric of the last iteration: one more hand-off, then as above ----
  if (ev_on) {
    if (!A.ev.each && bid == 0 && threadIdx.x == 0) { A.ev.health_next[0] = 0ull; A.ev.health_next[1] = 0ull; }   // (each: both pairs end at zero by themselves)
    grid_sync(A.sync, A.epoch_base + (++epoch) * nblk, A.status, A.seq);
    float cmv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, lmu[3] = {0.f, 0.f, 0.f};
    if (has_tile) metric_means((uint32_t)A.n_iters - 1u, cmv, lmu);
    metric((uint32_t)A.n_iters - 1u, __float_as_int(lm[13]), cmv, lmu);
  }

  // ---- what stayed in registers goes back to its arrays ----
  if (has_tile) {
    store_tile<kCmsgG, false>(a.cmsg, tile, lane, cm);
    if (fac_dirty) store_tile<kFacG, false>(a.fac, tile, lane, fac);
  }
}
