// experiments/gbp_lab_kernels.hip — the LABORATORY: mapping experiments and timing ablations that DESIGN.md's mapping decision
// and profiles/HISTORY.md quote.  Included by gbp_kernels.hip (inside namespace gbp, behind the product kernels) ONLY when the
// library is built with -DGBP_BUILD_EXPERIMENTS (`python -m gbp_poplar_amd.build --experiments` -> libgbp_mi355x_exp.so, loaded
// by profiles/*.py and tests/test_gpu_experiments.py); neither the product nor the test-hooks library contains a line of it.
//   k_sweep_lab<LAB>   a COPY of the product sweep's shell (sweep_tile) with timing ablations — results are garbage:
//                      1 = no landmark-message stream, 2 = no landmark-belief gather, 4 = no arithmetic (pass-through),
//                      16 / 32 = no landmark-message load / store; 256 = + the tail the sharded partial-sum fusion would add (write-through row sums,
//                      acknowledged stores, one agent-scope arrival per row); "no lane / every lane relinearises" needs no code: the
//                      launcher passes dmu_threshold = -1 / +inf (64 / 128)
//   k_sweep_w3         the product sweep forced to three wavefronts per SIMD (<= 168 VGPRs)
//   k_sweep_loop       the product sweep as resident waves that loop over the tiles
//   k_sweep_coop16     the north star's sub-wave mapping: 16 lanes per factor, blocks in LDS (bit-identical to k_sweep)
//   k_inv6_coop        16 lanes per 6x6 inverse (gbp_debug_math op 9)
// plus the environment overrides of the persistent kernel's placement (GBP_PERSIST_SPREAD).
#include <cstdlib>

// ---- timing ablations: the product sweep's shell with pieces switched off ----
template <int LAB>
GBP_DEV void lab_sweep_tile(const SweepArgs& a, const uint32_t wslot) {
  // (the slot is wave-uniform: as an SGPR it turns the permutation look-up into one scalar load)
  const uint32_t ws = (uint32_t)__builtin_amdgcn_readfirstlane((int)wslot);
  const uint32_t tile = a.tile_perm ? a.tile_perm[ws] : ws;
  const uint32_t lane = threadIdx.x & 63, p = tile * 64 + lane;

  const uint32_t cam_i = a.row_cam[p >> 4];
  const uint32_t lmk_i = __builtin_nontemporal_load(a.lmk_idx + p);

  float fac[56], cm[28], mu[12], lm[16], cb[44], lb[16];
  load_tile<kFacG>(a.fac, tile, lane, fac);
  // The camera messages: non-temporal like the potentials, or — SweepArgs.cmsg_cached, graphs with few cameras — with the
  // default policy like the landmark messages below (both are rewritten in place by this tile).  The potentials, which an
  // ordinary sweep only reads, keep the hint on every graph: with default-policy loads they cost 3 %.
  load_tile<kCmsgG>(a.cmsg, tile, lane, cm);
  if (!true) load_tile<kMuG>(a.mu, tile, lane, mu);
  // Landmark messages live as 64-byte records in DEVICE (camera-major) order: the wave's 64 records are one
  // contiguous 4 KiB block, moved with four coalesced 1 KiB accesses and transposed through a wave-private
  // LDS staging area.  Piece q of record r sits at float4 slot r*4 + (q ^ swz(r)), swz(r) = ((r>>2)&3) ^ (r&2):
  // a permutation inside each 64-B record, so the tile-order accesses (whole records) and the record-order
  // accesses (one piece per lane) are both bank-conflict-free for ds_read_b128 (16-lane groups, 64 banks)
  // and ds_write_b128 (8-lane groups, 32 banks).  k_beliefs gathers the records of a landmark by position
  // (random 64-B READS are ~2.3x cheaper than random 64-B writes: measured, profiles/HISTORY.md).
  __shared__ float4 lm_stage[kWpb][64 * 4];
  float4* stage = lm_stage[threadIdx.x >> 6];
  const uint32_t rec_t = lane >> 2;                                   // record handled in tile order (+16k)
  const uint32_t swz_own = ((lane >> 2) & 3u) ^ (lane & 2u);          // swizzle of the lane's own record
  float4* lm_tile = a.lmsg + (size_t)tile * 256;
  if (LAB & (1 | 16)) {  // no landmark-message load: a plausible active record
    GBP_UNROLL
    for (int i = 0; i < 16; ++i) lm[i] = 0.f;
    lm[13] = __int_as_float((int)((5u << 3) | kFlagActive));
    lm[14] = 4.f;
  } else {
  GBP_UNROLL
  for (int k = 0; k < 4; ++k) {
    // (DEFAULT policy for this one stream unless the shape says otherwise: the tile is rewritten in place ten microseconds later
    // and gathered by k_beliefs right after the sweep — measured +1.5 % iterations/s on the 1M-factor graph against the
    // non-temporal hint, with either store policy; the potentials keep the hint on every graph)
    const v4f* src = reinterpret_cast<const v4f*>(lm_tile) + k * 64 + lane;
    const v4f v = *src;
    const uint32_t r = k * 16 + rec_t;
    stage[r * 4 + ((lane & 3u) ^ (((r >> 2) & 3u) ^ (r & 2u)))] = make_float4(v.x, v.y, v.z, v.w);
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
  load_rec<kCamRec4>(a.camb + (size_t)cam_i * kCamRec4, cb);
  if (LAB & 2) {         // no landmark-belief gather: the identity
    GBP_UNROLL
    for (int i = 0; i < 16; ++i) lb[i] = (i == 4 || i == 8 || i == 12) ? 1.f : 0.f;
  } else {
    load_rec<kLmkRec4>(a.lmkb + (size_t)lmk_i * kLmkRec4, lb);
  }
  // per-factor scalar state rides in the pad slots of the landmark-message record (read and rewritten
  // every sweep anyway): [3] damping, [13] (damping_count << 3) | flags, [14] measurement variance
  float damping = lm[3];
  const int packed = __float_as_int(lm[13]);
  int count = packed >> 3;
  uint32_t flags = (uint32_t)packed & 7u;
  const float var = lm[14];
  const bool active = (flags & kFlagActive) != 0;

  float K[9];
  GBP_UNROLL
  for (int i = 0; i < 9; ++i) K[i] = a.K[i];

  float oc_eta[6], oc_lam[36], ol[16];
  bool relin;
  if (LAB & 4) {         // no arithmetic: every load kept alive, every store fed
    relin = false;
    GBP_UNROLL
    for (int i = 0; i < 16; ++i) ol[i] = lm[i] + lb[i];
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) oc_eta[i] = cm[i] + fac[i];
    GBP_UNROLL
    for (int i = 0; i < 36; ++i) oc_lam[i] = fac[9 + i] + cb[8 + i] + cm[6 + (i % 21)];
  } else
  factor_update<true>(fac, cm, mu, lm, cb, lb, K, a.hp, damping, count, flags, var, active, oc_eta, oc_lam, ol, relin,
                            [&](float (&x0c)[6], float (&x0l)[3], CamLin& cl) {   // rare path: loaded only by relinearising lanes
                              // camera side: the hoisted mean and its CAM_LIN record — per-camera tables (C x 144 B) that live in L2
                              const float4 m0 = a.cam_mu[(size_t)cam_i * 4], m1 = a.cam_mu[(size_t)cam_i * 4 + 1];
                              float4 q[kCamLin4];
                              GBP_UNROLL
                              for (int g = 0; g < kCamLin4; ++g) q[g] = a.cam_lin[(size_t)cam_i * kCamLin4 + g];
                              x0c[0] = m0.x; x0c[1] = m0.y; x0c[2] = m0.z; x0c[3] = m0.w; x0c[4] = m1.x; x0c[5] = m1.y;
                              cam_lin_unpack(q, cl);
                              // landmark side: the mean is RECOMPUTED from the belief record the lane holds anyway — inf2mean3x3
                              // (bafuncs.cpp:11-15) with the operations k_beliefs used for LMK_MU, so the same bits — instead of
                              // gathered: a second random gather (a 128-B line fill per factor for 12 useful bytes) made the
                              // lock-step relinearising sweep move 104 MB more than it has to
                              float B[9], S3[9];
                              GBP_UNROLL
                              for (int i = 0; i < 9; ++i) B[i] = lb[4 + i];
                              inv3x3(B, S3);
                              GBP_UNROLL
                              for (int i = 0; i < 3; ++i) {
                                float a2 = 0.f;
                                GBP_UNROLL
                                for (int k = 0; k < 3; ++k) a2 += S3[i * 3 + k] * lb[k];
                                x0l[i] = a2;
                              }
                            });

  // ---- outputs --------------------------------------------------------------------------------
  ol[3] = damping;
  ol[13] = __int_as_float((int)(((uint32_t)count << 3) | flags));
  ol[14] = var;
  if (!(LAB & (1 | 32))) {   // (32: no landmark-message store)
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
    const float4 f = stage[r * 4 + ((lane & 3u) ^ (((r >> 2) & 3u) ^ (r & 2u)))];
    lm_tile[k * 64 + lane] = f;
  }
  }
  {
    float cmo[28];
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) cmo[i] = oc_eta[i];
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) {
      GBP_UNROLL
      for (int j = 0; j <= i; ++j) cmo[6 + tri(i, j)] = oc_lam[i * 6 + j];
    }
    cmo[27] = 0.f;
    store_tile<kCmsgG>(a.cmsg, tile, lane, cmo);
  }
  // camera half of the belief reduction: per-row (16 factors of one camera) tree sums
  if (LAB & 256) {
    // 256: what folding the sharded iteration's partial-sum launch into the sweep would add to EVERY wave: the row sums
    // written through (sc1), the wave's stores acknowledged, one agent-scope arrival per row on its camera's counter (here: a pad word
    // of ROWP — results are garbage) — the last arriver's sum itself (one camera in ~2.4 waves) is not emulated
    const XwBuf R(a.rowp);
    row16_sums_store(oc_eta, oc_lam, lane, [&](uint32_t g, float4 v) { R.st4((p >> 4) * kCamRec4 + g, v); });
  } else {
    float4* rp = a.rowp + (size_t)(p >> 4) * kCamRec4;
    row16_sums_store(oc_eta, oc_lam, lane, [&](uint32_t g, float4 v) { rp[g] = v; });
  }
  if (active) {
    if (!true) store_tile<kMuG>(a.mu, tile, lane, mu);
    if (relin) store_tile<kFacG>(a.fac, tile, lane, fac);
  }
  if (LAB & 256) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0;
    if ((lane & 15u) == 0u)
      old = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(a.rowp) + (size_t)cam_i * kCamRec + 7, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__builtin_amdgcn_ballot_w64(old == 0xfffffff0u) != 0ull) a.rowp[0].x = 1.f;      // (the wave waits for what its arrivals return)
  }
}

template <int LAB>
__global__ __launch_bounds__(64 * kWpb) void k_sweep_lab(const SweepArgs a) {
  lab_sweep_tile<LAB>(a, blockIdx.x * kWpb + (threadIdx.x >> 6));
}

// Mapping experiment (profiles/time_mapping.py): the SAME sweep forced to three wavefronts per SIMD
// (<= 168 VGPRs): what a third wave buys against what the spills cost.
__global__ __launch_bounds__(64 * kWpb) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_sweep_w3(const SweepArgs a) {
  sweep_tile<true, 0>(a, blockIdx.x * kWpb + (threadIdx.x >> 6));
}
// Mapping experiment (gbp_params.reserved[0] = 2): the same sweep as RESIDENT waves that loop over the tiles (grid = what
// fits the chip at two waves per SIMD) instead of one wave per tile: a wave's stores overlap the next tile's loads, no
// wave slot idles between a retiring wave and its successor.
__global__ __launch_bounds__(256) void k_sweep_loop(const SweepArgs a, const uint32_t n_slots) {
  for (uint32_t ws = blockIdx.x * 4 + (threadIdx.x >> 6); ws < n_slots; ws += gridDim.x * 4) sweep_tile<true, 0>(a, ws);
}

// =================================================================================================
// k_sweep_coop16 — the sweep in the NORTH STAR's sub-wave mapping, built so that it can be measured against the product
// kernel (profiles/time_mapping.py): 16 lanes (one DPP row) cooperate on ONE factor, the factor's blocks
// (potential, both messages, both beliefs) are staged in LDS, every product / inverse is evaluated with lane = output
// element in the reference's k order, the 6x6 inverse is the cooperative LDL^T of k_inv6_coop.  One wavefront = 4
// factors, one workgroup = 16 factors = one camera row (its 44 row sums are a tree over the 16 factors, same order as
// row16_sum).  Functionally complete and bit-identical to k_sweep (a relinearising factor runs relin_core on lane 0 of
// its group); experiments build only.
// =================================================================================================
namespace coop {
constexpr int kF = 0, kCm = 56, kLm = 84, kCb = 100, kLb = 144, kWs = 160;          // LDS floats of one factor
constexpr int kAp = kWs, kU = kWs + 21, kUi = kWs + 36, kAinv = kWs + 51, kG = kWs + 87, kEd = kWs + 105, kBp = kWs + 111, kBi = kWs + 120,
              kG2 = kWs + 129, kEl = kWs + 147, kOut = kWs + 150 /* ol 16 | oc_eta 6 | oc_lam 36 */, kStride = kWs + 150 + 58 + 2;   // 370 floats
GBP_DEV void sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// cooperative un-pivoted LDL^T inverse of the packed lower triangle at w[kAp..] -> w[kAinv..] (36), lane t of 16
GBP_DEV void inv6(float* w, int t) {
  auto uidx = [](int j, int i) { return kU + j * 5 - j * (j - 1) / 2 + (i - j - 1); };   // U[j][i], j < i   (15 entries)
  float D[6], rD[6];
  GBP_UNROLL
  for (int j = 0; j < 6; ++j) {
    float d = w[kAp + tri(j, j)];
    GBP_UNROLL
    for (int k = 0; k < j; ++k) { const float ukj = w[uidx(k, j)]; d -= ukj * ukj * D[k]; }
    D[j] = d;
    rD[j] = 1 / d;
    const int i = j + 1 + t;
    if (i < 6) {
      float u = rD[j] * w[kAp + tri(i, j)];
      GBP_UNROLL
      for (int k = 0; k < j; ++k) u -= rD[j] * w[uidx(k, i)] * w[uidx(k, j)] * D[k];
      w[uidx(j, i)] = u;
    }
    sync();
  }
  {
    float ui[6];
    GBP_UNROLL
    for (int k = 0; k < 6; ++k) ui[k] = 0.f;
    GBP_UNROLL
    for (int j = 1; j < 6; ++j) {
      if (t < j) {
        float acc = 0.f;
        acc += w[uidx(t, j)];
        GBP_UNROLL
        for (int k = 1; k < j; ++k)
          if (k > t) acc += ui[k] * w[uidx(k, j)];
        ui[j] = acc / -1.f;
        w[kUi + (uidx(t, j) - kU)] = ui[j];
      }
    }
  }
  sync();
  GBP_UNROLL
  for (int r = 0; r < 3; ++r) {
    const int e = t + 16 * r;
    if (e < 36) {
      const int i = e / 6, j = e - 6 * i;
      const int k0 = i > j ? i : j;
      float acc = 0.f;
      GBP_UNROLL
      for (int k = 0; k < 6; ++k) {
        if (k >= k0) {
          const float ww = (k == i) ? rD[k] : w[kUi + (uidx(i, k) - kU)] * rD[k];
          if (k == j) acc += ww;
          else acc += ww * w[kUi + (uidx(j, k) - kU)];
        }
      }
      w[kAinv + e] = acc;
    }
  }
  sync();
}
}  // namespace coop

__global__ __launch_bounds__(256) void k_sweep_coop16(const SweepArgs a) {
  using namespace coop;
  __shared__ float lds[16][kStride];
  const uint32_t f = threadIdx.x >> 4, t = threadIdx.x & 15;        // factor of the workgroup, lane of the factor
  const uint32_t p = blockIdx.x * 16 + f, tile = p >> 6, lt = p & 63;
  float* w = lds[f];
  const uint32_t cam_i = a.row_cam[p >> 4], lmk_i = a.lmk_idx[p];
  // ---- stage the factor's 160 input floats: 40 float4 over 16 lanes ----
  GBP_UNROLL
  for (int r = 0; r < 3; ++r) {
    const int i = (int)t + 16 * r;
    float4 v;
    int dst = -1;
    if (i < 14) { v = a.fac[((size_t)tile * kFacG + i) * 64 + lt]; dst = kF + 4 * i; }
    else if (i < 21) { v = a.cmsg[((size_t)tile * kCmsgG + (i - 14)) * 64 + lt]; dst = kCm + 4 * (i - 14); }
    else if (i < 25) { v = a.lmsg[(size_t)p * 4 + (i - 21)]; dst = kLm + 4 * (i - 21); }
    else if (i < 36) { v = a.camb[(size_t)cam_i * kCamRec4 + (i - 25)]; dst = kCb + 4 * (i - 25); }
    else if (i < 40) { v = a.lmkb[(size_t)lmk_i * kLmkRec4 + (i - 36)]; dst = kLb + 4 * (i - 36); }
    if (dst >= 0) { w[dst] = v.x; w[dst + 1] = v.y; w[dst + 2] = v.z; w[dst + 3] = v.w; }
  }
  sync();
  const float* fac = w + kF; const float* cm = w + kCm; const float* lm = w + kLm; const float* cb = w + kCb; const float* lb = w + kLb;
  float damping = lm[3];
  const int packed = __float_as_int(lm[13]);
  int count = packed >> 3;
  uint32_t flags = (uint32_t)packed & 7u;
  const float var = lm[14];
  const bool active = (flags & kFlagActive) != 0;
  bool relin = false;
  float* out = w + kOut;               // ol[0..15] | oc_eta[16..21] | oc_lam[22..57]
  GBP_UNROLL
  for (int r = 0; r < 4; ++r) { const int e = (int)t + 16 * r; if (e < 58) out[e] = 0.f; }
  if (active) {    // uniform over the 16 lanes of a factor
    if (0 == count) damping = a.hp.maxeta_damping;
    count += 1;
    float d2 = cb[6];
    d2 += lb[3];
    d2 += lb[13];
    d2 += lb[14];
    const float dmu = sqrtf(d2);
    relin = (dmu < a.hp.dmu_threshold) && (count > a.hp.min_linear_iters - a.hp.num_undamped_iters);
    if (relin) {
      damping = 0.f;
      count = -a.hp.num_undamped_iters;
      if (t == 0) {                    // the rare path stays on one lane: relin_core as the product kernel runs it
        float fr[56], x0c[6], x0l[3], K[9];
        GBP_UNROLL
        for (int i = 0; i < 56; ++i) fr[i] = fac[i];
        GBP_UNROLL
        for (int i = 0; i < 9; ++i) K[i] = a.K[i];
        const float4 m0 = a.cam_mu[(size_t)cam_i * 4], m1 = a.cam_mu[(size_t)cam_i * 4 + 1];
        const float4 l0 = a.lmk_mu[(size_t)lmk_i * 2];
        x0c[0] = m0.x; x0c[1] = m0.y; x0c[2] = m0.z; x0c[3] = m0.w; x0c[4] = m1.x; x0c[5] = m1.y;
        x0l[0] = l0.x; x0l[1] = l0.y; x0l[2] = l0.z;
        if (a.hp.relin_mode == 1) {
          GBP_UNROLL
          for (int i = 0; i < 54; ++i) fr[i] = 0.f;
        }
        CamLin cl;
        const float wv[3] = {x0c[3], x0c[4], x0c[5]};
        cam_lin(wv, cl);
        const bool robust = relin_core(fr, x0c, x0l, K, var, a.hp.nstds, cl);
        GBP_UNROLL
        for (int i = 0; i < 54; ++i) w[kF + i] = fr[i];
        w[kStride - 1] = robust ? 1.f : 0.f;
      }
      sync();
      flags = w[kStride - 1] != 0.f ? (flags | kFlagRobust) : (flags & ~kFlagRobust);
    }
    const float omd = 1 - damping;
    // ---- factor -> landmark message (gbp_codelets.cpp:503-562, 664-709) ----
    GBP_UNROLL
    for (int r = 0; r < 2; ++r) {
      const int e = (int)t + 16 * r;
      if (e < 21) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        const int j = e - i * (i + 1) / 2;
        float v = fac[9 + e] + cb[8 + i * 6 + j];
        w[kAp + e] = v - cm[6 + e];
      }
    }
    if (t < 6) { const float v = fac[t] + cb[t]; w[kEd + t] = v - cm[t]; }
    sync();
    inv6(w, (int)t);
    GBP_UNROLL
    for (int r = 0; r < 2; ++r) {
      const int e = (int)t + 16 * r;
      if (e < 18) {
        const int i = e / 6, j = e - 6 * i;
        float acc = 0.f;
        GBP_UNROLL
        for (int k = 0; k < 6; ++k) acc += fac[30 + k * 3 + i] * w[kAinv + k * 6 + j];
        w[kG + e] = acc;
      }
    }
    sync();
    if (t < 3) {
      float s = 0.f;
      GBP_UNROLL
      for (int k = 0; k < 6; ++k) s += w[kG + t * 6 + k] * w[kEd + k];
      const float h = fac[6 + t] - s;
      out[t] = h * omd + lm[t] * damping;
    } else if (t < 12) {
      const int e = (int)t - 3, i = e / 3, j = e - 3 * i;
      float tt = 0.f;
      GBP_UNROLL
      for (int k = 0; k < 6; ++k) tt += w[kG + i * 6 + k] * fac[30 + k * 3 + j];
      out[4 + e] = fac[48 + trisym(i, j)] - tt;
    }
    // ---- factor -> camera message (gbp_codelets.cpp:411-471, 592-637) ----
    if (t < 9) {
      const int i = (int)t / 3, j = (int)t - 3 * i;
      const float v = fac[48 + trisym(i, j)] + lb[4 + t];
      w[kBp + t] = v - lm[4 + t];
    }
    if (t < 3) { const float v = fac[6 + t] + lb[t]; w[kEl + t] = v - lm[t]; }
    sync();
    {
      float M[9], I[9];
      GBP_UNROLL
      for (int i = 0; i < 9; ++i) M[i] = w[kBp + i];
      inv3x3(M, I);                    // 50 operations: every lane runs it, lane t < 9 keeps entry t
      if (t < 9) {
        float v = I[0];
        GBP_UNROLL
        for (int i = 1; i < 9; ++i) v = ((int)t == i) ? I[i] : v;
        w[kBi + t] = v;
      }
    }
    sync();
    GBP_UNROLL
    for (int r = 0; r < 2; ++r) {
      const int e = (int)t + 16 * r;
      if (e < 18) {
        const int i = e / 3, j = e - 3 * i;
        float acc = 0.f;
        GBP_UNROLL
        for (int k = 0; k < 3; ++k) acc += fac[30 + i * 3 + k] * w[kBi + k * 3 + j];
        w[kG2 + e] = acc;
      }
    }
    sync();
    if (t < 6) {
      float s = 0.f;
      GBP_UNROLL
      for (int k = 0; k < 3; ++k) s += w[kG2 + t * 3 + k] * w[kEl + k];
      const float h = fac[t] - s;
      out[16 + t] = h * omd + cm[t] * damping;
    }
    GBP_UNROLL
    for (int r = 0; r < 3; ++r) {
      const int e = (int)t + 16 * r;
      if (e < 36) {
        const int i = e / 6, j = e - 6 * i;
        float tt = 0.f;
        GBP_UNROLL
        for (int k = 0; k < 3; ++k) tt += w[kG2 + i * 3 + k] * fac[30 + j * 3 + k];
        out[22 + e] = fac[9 + trisym(i, j)] - tt;
      }
    }
  }
  if (t == 0) {
    out[3] = damping;
    out[13] = __int_as_float((int)(((uint32_t)count << 3) | flags));
    out[14] = var;
  }
  sync();
  // ---- stores: landmark-message record (64 B), camera message (7 groups of the tile layout), potential if relinearised ----
  if (t < 4) a.lmsg[(size_t)p * 4 + t] = make_float4(out[4 * t], out[4 * t + 1], out[4 * t + 2], out[4 * t + 3]);
  if (t < 7) {
    float c4[4];
    GBP_UNROLL
    for (int q = 0; q < 4; ++q) {
      const int e = 4 * (int)t + q;            // cmo[e]: eta 0..5, lower triangle 6..26, pad
      float v = 0.f;
      if (e < 6) v = out[16 + e];
      else if (e < 27) {
        const int m = e - 6;
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= m) ++i;
        const int j = m - i * (i + 1) / 2;
        v = out[22 + i * 6 + j];
      }
      c4[q] = v;
    }
    a.cmsg[((size_t)tile * kCmsgG + t) * 64 + lt] = make_float4(c4[0], c4[1], c4[2], c4[3]);
  }
  if (active && relin && t < 14) a.fac[((size_t)tile * kFacG + t) * 64 + lt] = make_float4(w[kF + 4 * t], w[kF + 4 * t + 1], w[kF + 4 * t + 2], w[kF + 4 * t + 3]);
  // ---- row sums over the 16 factors of the workgroup (= one camera row): the tree of row16_sum ----
  __syncthreads();
  if (threadIdx.x < 44) {
    const int j = (int)threadIdx.x;
    float r = 0.f;
    if (j != 6 && j != 7) {
      const int src = j < 6 ? kOut + 16 + j : kOut + 22 + (j - 8);
      float x[16];
      GBP_UNROLL
      for (int q = 0; q < 16; ++q) x[q] = lds[q][src];
      const float q0 = (x[0] + x[1]) + (x[2] + x[3]), q1 = (x[6] + x[7]) + (x[4] + x[5]);     // operand order of the DPP steps of lane 0 / lane 15
      const float q2 = (x[8] + x[9]) + (x[10] + x[11]), q3 = (x[15] + x[14]) + (x[13] + x[12]);
      r = (q0 + q1) + (q3 + q2);
    }
    reinterpret_cast<float*>(a.rowp)[(size_t)blockIdx.x * kCamRec + j] = r;
  }
}
// =================================================================================================
// k_inv6_coop: the SUB-WAVE mapping the north star sketches, built for the dominant routine so that it can be measured:
// 16 lanes (one DPP row) cooperate on ONE 6x6 inverse, operands staged in LDS, lane = output element, every k-loop
// in the reference's order (so the result is bit-identical to inv6x6_lower / matlib.cpp:180-222).  Four matrices per
// wavefront instead of 64.  Test + measurement hook (gbp_debug_math op 9, gbp_debug_math_timed): DESIGN.md 2 quotes
// its timing against the lane-per-matrix routine.
// =================================================================================================
__global__ __launch_bounds__(256) void k_inv6_coop(const float* __restrict__ in, float* __restrict__ out, int n) {
  __shared__ float ws_all[16][64];                          // per 16-lane group: A lower 21 | U 15 | Ui 15
  const int grp = (blockIdx.x * 256 + threadIdx.x) >> 4;    // matrix handled by this 16-lane group
  const int t = threadIdx.x & 15;
  float* ws = ws_all[threadIdx.x >> 4];
  const bool live = grp < n;
  const float* A = in + (size_t)(live ? grp : 0) * 36;
  auto uidx = [](int j, int i) { return 21 + j * 5 - j * (j - 1) / 2 + (i - j - 1); };   // U[j][i], j < i   (15 entries)
  auto sync = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  GBP_UNROLL
  for (int r = 0; r < 2; ++r) {                             // stage the lower triangle
    const int e = t + 16 * r;
    if (e < 21) {
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      const int j = e - i * (i + 1) / 2;
      ws[e] = A[i * 6 + j];
    }
  }
  sync();
  float D[6], rD[6];
  GBP_UNROLL
  for (int j = 0; j < 6; ++j) {                             // un-pivoted LDL^T, column by column
    float d = ws[tri(j, j)];
    GBP_UNROLL
    for (int k = 0; k < j; ++k) { const float ukj = ws[uidx(k, j)]; d -= ukj * ukj * D[k]; }
    D[j] = d;
    rD[j] = 1 / d;
    const int i = j + 1 + t;                                // lane t owns U[j][j+1+t]
    if (i < 6) {
      float u = rD[j] * ws[tri(i, j)];
      GBP_UNROLL
      for (int k = 0; k < j; ++k) u -= rD[j] * ws[uidx(k, i)] * ws[uidx(k, j)] * D[k];
      ws[uidx(j, i)] = u;
    }
    sync();
  }
  {                                                         // inverse of the unit upper factor: lane t < 5 owns row t
    float ui[6];
    GBP_UNROLL
    for (int k = 0; k < 6; ++k) ui[k] = 0.f;
    GBP_UNROLL
    for (int j = 1; j < 6; ++j) {
      if (t < j) {
        float acc = 0.f;
        acc += ws[uidx(t, j)];
        GBP_UNROLL
        for (int k = 1; k < j; ++k)
          if (k > t) acc += ui[k] * ws[uidx(k, j)];
        ui[j] = acc / -1.f;
        ws[36 + (uidx(t, j) - 21)] = ui[j];
      }
    }
  }
  sync();
  GBP_UNROLL
  for (int r = 0; r < 3; ++r) {                             // Ainv = (LTinv Dinv) LTinv^T, lane = output element
    const int e = t + 16 * r;
    if (e < 36) {
      const int i = e / 6, j = e - 6 * i;
      const int k0 = i > j ? i : j;
      float acc = 0.f;
      GBP_UNROLL
      for (int k = 0; k < 6; ++k) {
        if (k >= k0) {
          const float w = (k == i) ? rD[k] : ws[36 + (uidx(i, k) - 21)] * rD[k];
          if (k == j) acc += w;
          else acc += w * ws[36 + (uidx(j, k) - 21)];
        }
      }
      if (live) out[(size_t)grp * 36 + e] = acc;
    }
  }
}

void lab_launch_inv6_coop(const float* in, float* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_inv6_coop, dim3(((size_t)n * 16 + 255) / 256), dim3(256), 0, s, in, out, n);
}

// gbp_params.reserved[0] (SweepArgs.variant): 1 = the sub-wave mapping, 2 + k = resident looping waves (grid = k ? k : 512 workgroups)
bool lab_launch_sweep(const SweepArgs& a, uint32_t n_tiles, bool hoist, hipStream_t s) {
  if (!hoist || a.variant == 0) return false;
  const uint32_t n_blocks = n_tiles / 4;
  if (a.variant == 1) { hipLaunchKernelGGL(k_sweep_coop16, dim3(n_blocks * 16), dim3(256), 0, s, a); return true; }
  const uint32_t nb = a.variant > 2 ? (uint32_t)a.variant : 512u;
  hipLaunchKernelGGL(k_sweep_loop, dim3(nb < n_blocks ? nb : n_blocks), dim3(256), 0, s, a, n_blocks * 4);
  return true;
}

// gbp_debug_time_sweep: one launch of an ablated sweep (profiles/ablate_sweep.py, profiles/time_mapping.py)
bool lab_launch_sweep_ablated(const SweepArgs& a0, uint32_t n_tiles, int abl, hipStream_t s) {
  const dim3 g(n_tiles / kWpb), b(64 * kWpb);
  SweepArgs a = a0;
  if (abl & 64) { a.hp.dmu_threshold = -1.f; abl &= ~64; }                                    // no lane relinearises
  if (abl & 128) { a.hp.dmu_threshold = __builtin_inff(); a.hp.min_linear_iters = -1000000; abl &= ~128; }   // every lane does
  switch (abl) {
    case 0: hipLaunchKernelGGL((k_sweep<true, 0>), g, b, 0, s, a); break;
#define GBP_LAB_CASE(N) case N: hipLaunchKernelGGL((k_sweep_lab<N>), g, b, 0, s, a); break;
    GBP_LAB_CASE(1) GBP_LAB_CASE(2) GBP_LAB_CASE(3) GBP_LAB_CASE(4) GBP_LAB_CASE(7) GBP_LAB_CASE(16) GBP_LAB_CASE(32) GBP_LAB_CASE(256)
#undef GBP_LAB_CASE
    case 3000: hipLaunchKernelGGL(k_sweep_w3, g, b, 0, s, a); break;                                // the product sweep at 3 waves / SIMD
    case 3001: hipLaunchKernelGGL(k_sweep_coop16, dim3(n_tiles * 4), dim3(256), 0, s, a); break;    // 16 lanes per factor
    default: return false;
  }
  return true;
}

// placement studies of k_persist and the forced barrier time-out of tests/test_gpu_experiments.py
int lab_persist_spread(int spread) {
  static const int env_spread = std::getenv("GBP_PERSIST_SPREAD") ? std::atoi(std::getenv("GBP_PERSIST_SPREAD")) : 0;
  return env_spread ? env_spread : spread;
}
