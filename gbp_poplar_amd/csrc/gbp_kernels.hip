// gbp_kernels.hip — CDNA4 (gfx950) kernels of one synchronous GBP iteration.
//
// Replaces the reference's per-iteration Poplar program (ba/ba.cpp:895-905):
//   k_sweep        PrepMessageVertex + Copy(mu,oldmu) + the four Compute*Message*Vertex classes
//                  (gbp_codelets.cpp:215-710) + the camera half of popops::reduceWithOutput
//                  (ba.cpp:129-132) as per-row partial sums + Copy(msg,pmsg) (in-place messages)
//   k_beliefs      the rest of buildUpdateBeliefsProg (ba.cpp:104-139): camera rows + prior, landmark messages + prior
//   k_persist      n iterations of the two above inside ONE launch (graphs whose workgroups are all resident at once)
//   k_linearise    RelineariseFactorVertex (gbp_codelets.cpp:20-172)
//                  + WeakenPriorVertex (gbp_codelets.cpp:176-197) inside the refresh WEAKEN_PRIORS ends with
//   k_means/k_eval eval_reprojection_error (util.cpp:74-144) + counters (ba.cpp:1011-1020)
//
// Mapping: one LANE per factor, 64 factors per wavefront, all blocks in VGPRs.
// The sweep is HBM-bound: every per-factor stream is a tile-coalesced 16-B-per-lane access, the
// 6x6/3x3 algebra (no MFMA: the blocks are tiny and chains are serial) hides under the loads.
// fp32 throughout, compiled with -ffp-contract=off so results are bit-comparable with the oracle.
#include "gbp_kernels.h"
#include "gbp_device_math.hpp"

namespace gbp {
using namespace gbpdev;

namespace {

// The per-factor tiles are touched exactly once per sweep: stream them with the non-temporal hint so that
// they do not displace the gathered tables (landmark messages / beliefs, camera beliefs: ~70 MB for S1)
// from L2 and the 256 MiB Infinity Cache.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int G, bool NT = true>
GBP_DEV void load_tile(const float4* base, uint32_t tile, uint32_t lane, float (&out)[G * 4]) {
  const v4f* p = reinterpret_cast<const v4f*>(base) + (size_t)tile * G * 64 + lane;
  GBP_UNROLL
  for (int g = 0; g < G; ++g) {
    const v4f v = NT ? __builtin_nontemporal_load(p + g * 64) : p[g * 64];
    out[4 * g] = v.x; out[4 * g + 1] = v.y; out[4 * g + 2] = v.z; out[4 * g + 3] = v.w;
  }
}
template <int G, bool NT = true>
GBP_DEV void store_tile(float4* base, uint32_t tile, uint32_t lane, const float (&in)[G * 4]) {
  v4f* p = reinterpret_cast<v4f*>(base) + (size_t)tile * G * 64 + lane;
  GBP_UNROLL
  for (int g = 0; g < G; ++g) {
    const v4f v = {in[4 * g], in[4 * g + 1], in[4 * g + 2], in[4 * g + 3]};
    if (NT) __builtin_nontemporal_store(v, p + g * 64); else p[g * 64] = v;
  }
}
// SEG (k_sweep<..., SEG = true>): the same tile accesses through a buffer descriptor that covers exactly this tile's G KiB, so that
// a lane whose 64-byte segment (four lanes) holds PAD positions only can be given an offset beyond the descriptor's range: the
// hardware then returns zeros for a load and drops a store WITHOUT a memory access — no branch, no divergence, and the all-pad
// segments (the unused tail of a camera's last row: 3.7 % of the positions on the config-5 shard shape) are never streamed in or
// out.  Measured on that shape: 839 -> 813 MB per ordinary sweep (reads only drop where BOTH halves of a 128-byte line are empty,
// writes per 64 bytes), 0.1551 -> 0.1532 ms per iteration (profiles/r06_configs.md).  aux: 2 = the non-temporal hint of the gfx94x / gfx950 buffer instructions.
constexpr int kBufOob = 0x40000000;      // beyond every tile descriptor (<= 14 KiB), and + g KiB does not wrap
template <int G>
GBP_DEV __amdgpu_buffer_rsrc_t tile_rsrc(const float4* base, uint32_t tile) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(base) + (size_t)tile * G * 64, 0, G * 1024, 0x00020000);
}
template <int G, bool NT = true>
GBP_DEV void load_tile_seg(const float4* base, uint32_t tile, uint32_t lane, bool live, float (&out)[G * 4]) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t r = tile_rsrc<G>(base, tile);
  const int off = live ? (int)(lane * 16u) : kBufOob;
  GBP_UNROLL
  for (int g = 0; g < G; ++g) {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, off + g * 1024, 0, NT ? 2 : 0);
    out[4 * g] = __uint_as_float(v.x); out[4 * g + 1] = __uint_as_float(v.y); out[4 * g + 2] = __uint_as_float(v.z); out[4 * g + 3] = __uint_as_float(v.w);
  }
}
template <int G, bool NT = true>
GBP_DEV void store_tile_seg(float4* base, uint32_t tile, uint32_t lane, bool live, const float (&in)[G * 4]) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t r = tile_rsrc<G>(base, tile);
  const int off = live ? (int)(lane * 16u) : kBufOob;
  GBP_UNROLL
  for (int g = 0; g < G; ++g) {
    const v4u v = {__float_as_uint(in[4 * g]), __float_as_uint(in[4 * g + 1]), __float_as_uint(in[4 * g + 2]), __float_as_uint(in[4 * g + 3])};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off + g * 1024, 0, NT ? 2 : 0);
  }
}
template <int G>
GBP_DEV void load_rec(const float4* rec, float (&out)[G * 4]) {
  GBP_UNROLL
  for (int g = 0; g < G; ++g) {
    const float4 v = rec[g];
    out[4 * g] = v.x; out[4 * g + 1] = v.y; out[4 * g + 2] = v.z; out[4 * g + 3] = v.w;
  }
}

// The 44-float ROWP record of every 16-lane row of a tile: the sums of the camera messages (6 eta + 36 Lambda entries, two
// pad slots) over the row's 16 factors, each the balanced binary tree in lane order of row16_sum, handed to `st(g, float4)`
// (float4 group g of the row's record) by the lanes that end up holding them.
//
// A butterfly that leaves all 42 sums in all 16 lanes costs 4 x 42 tree nodes, each evaluated by sixteen lanes, as v_mov_b32_dpp +
// (packed) v_add: 252 instructions.  Here a node is ONE v_add_f32 with a DPP operand, and the two intra-quad steps HALVE the
// set a lane carries (lane%4 = q ends up with the float4 groups g = q, q+4, q+8 of the record): the partners of a step keep
// different halves, each sends the half the other keeps (two selects per pair of values), so the steps cost 60 + 36 + 12 +
// 12 = 120 instructions and the record leaves in three stores of 64 contiguous bytes per row (lanes 12..15) instead of
// eleven 16-byte ones (lane 0).  Every node adds the same two operands as the butterfly's ((x0+x1)+(x2+x3)) + ..., own +
// partner, commutative: the same bits.
//   step 1 (lane ^ 1): even lanes keep groups {0,2} mod 4, odd lanes groups {1,3} mod 4      -> t[2k], t[2k+1]
//   step 2 (lane ^ 2): lane%4 in {0,1} keep t[2k] (groups 4j, 4j+1), {2,3} keep t[2k+1]      -> u[k], k = 4j + component
//   steps 3, 4: row_shr:4 then row_shr:8 — quad 1 = Q0+Q1 and quad 3 = Q2+Q3, then quad 3 = (Q0+Q1)+(Q2+Q3)
// A DPP operand must have been written >= 2 instructions earlier and inline asm is invisible to the hazard recogniser:
// the DPP adds go in blocks of <= 6 behind one s_nop, reading only registers produced before their block (a copy or
// AGPR read-back the register allocator might place in front of a block is covered by the nop as well).  Call with every
// lane of the wave active.
#define GBP_DPP6(ctrl, o, s0, s1)                                                                                         \
  asm volatile("s_nop 1\n\t"                                                                                              \
               "v_add_f32_dpp %0, %6, %12 " ctrl "\n\tv_add_f32_dpp %1, %7, %13 " ctrl "\n\tv_add_f32_dpp %2, %8, %14 " ctrl "\n\t" \
               "v_add_f32_dpp %3, %9, %15 " ctrl "\n\tv_add_f32_dpp %4, %10, %16 " ctrl "\n\tv_add_f32_dpp %5, %11, %17 " ctrl      \
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])                              \
               : "v"(s0[0]), "v"(s0[1]), "v"(s0[2]), "v"(s0[3]), "v"(s0[4]), "v"(s0[5]),                                   \
                 "v"(s1[0]), "v"(s1[1]), "v"(s1[2]), "v"(s1[3]), "v"(s1[4]), "v"(s1[5]))
#define GBP_DPP_QP1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define GBP_DPP_QP2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define GBP_DPP_SHR4 "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define GBP_DPP_SHR8 "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
template <class Store>
GBP_DEV void row16_sums_store(const float (&oc_eta)[6], const float (&oc_lam)[36], const uint32_t lane, Store&& st) {
  float x[48];      // the record's slots as this lane's terms (6, 7: pads; 44..47: the group that does not exist)
  GBP_UNROLL
  for (int i = 0; i < 6; ++i) x[i] = oc_eta[i];
  GBP_UNROLL
  for (int i = 0; i < 36; ++i) x[8 + i] = oc_lam[i];
  x[6] = x[2]; x[7] = x[3];                      // (any defined value: what the pad / missing slots sum to is never stored)
  GBP_UNROLL
  for (int c = 0; c < 4; ++c) x[44 + c] = x[40 + c];
  const bool odd = (lane & 1u) != 0, hi = (lane & 2u) != 0;
  float t[24], u[12], w[12], r[12];
  {  // step 1: t[m], m = 2k + h, pairs the groups 4j + 2h (kept by even lanes) and 4j + 2h + 1 (kept by odd lanes)
    float keep[24], send[24];
    GBP_UNROLL
    for (int m = 0; m < 24; ++m) {
      const int k = m / 2, h = m % 2, j = k / 4, c = k % 4;
      const float a = x[4 * (4 * j + 2 * h) + c], b = x[4 * (4 * j + 2 * h + 1) + c];
      keep[m] = odd ? b : a;
      send[m] = odd ? a : b;
    }
    GBP_UNROLL
    for (int m = 0; m < 24; m += 6) {
      float o[6], s0[6], s1[6];
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) { s0[i] = send[m + i]; s1[i] = keep[m + i]; }
      GBP_DPP6(GBP_DPP_QP1, o, s0, s1);
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) t[m + i] = o[i];
    }
  }
  {  // step 2
    float keep[12], send[12];
    GBP_UNROLL
    for (int k = 0; k < 12; ++k) {
      keep[k] = hi ? t[2 * k + 1] : t[2 * k];
      send[k] = hi ? t[2 * k] : t[2 * k + 1];
    }
    GBP_UNROLL
    for (int k = 0; k < 12; k += 6) {
      float o[6], s0[6], s1[6];
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) { s0[i] = send[k + i]; s1[i] = keep[k + i]; }
      GBP_DPP6(GBP_DPP_QP2, o, s0, s1);
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) u[k + i] = o[i];
    }
  }
  GBP_UNROLL
  for (int k = 0; k < 12; k += 6) {   // step 3: lanes 4..7 <- 0..3, 12..15 <- 8..11
    float o[6], s0[6];
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) s0[i] = u[k + i];
    GBP_DPP6(GBP_DPP_SHR4, o, s0, s0);
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) w[k + i] = o[i];
  }
  GBP_UNROLL
  for (int k = 0; k < 12; k += 6) {   // step 4: lanes 12..15 <- 4..7
    float o[6], s0[6];
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) s0[i] = w[k + i];
    GBP_DPP6(GBP_DPP_SHR8, o, s0, s0);
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) r[k + i] = o[i];
  }
  if ((lane & 12u) == 12u) {
    const uint32_t q = lane & 3u;
    if (q == 1u) { r[2] = 0.f; r[3] = 0.f; }      // the pad slots of group 1
    GBP_UNROLL
    for (int j = 0; j < 3; ++j)
      if (j < 2 || q < 3u) st(q + 4u * (uint32_t)j, make_float4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]));
  }
}

// belief means: inf2mean6x6 / inf2mean3x3 (bafuncs.cpp:2-15) on CAMB / LMKB records
GBP_DEV void belief_means(const float (&cb)[44], const float (&lb)[16], float (&x0c)[6], float (&x0l)[3]) {
  float Al[21], S6[36], B[9], S3[9];
  GBP_UNROLL
  for (int i = 0; i < 6; ++i) {
    GBP_UNROLL
    for (int j = 0; j <= i; ++j) Al[tri(i, j)] = cb[8 + i * 6 + j];
  }
  inv6x6_lower(Al, S6);
  GBP_UNROLL
  for (int i = 0; i < 6; ++i) {
    float acc = 0.f;
    GBP_UNROLL
    for (int k = 0; k < 6; ++k) acc += S6[i * 6 + k] * cb[k];
    x0c[i] = acc;
  }
  GBP_UNROLL
  for (int i = 0; i < 9; ++i) B[i] = lb[4 + i];
  inv3x3(B, S3);
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) {
    float acc = 0.f;
    GBP_UNROLL
    for (int k = 0; k < 3; ++k) acc += S3[i * 3 + k] * lb[k];
    x0l[i] = acc;
  }
}

// Shared body of gbp_codelets.cpp:90-168 and :294-373 on the packed FAC record: accumulate
// J^T J / J^T (J x0 + z - h(x0)) onto the potential, Huber-rescale.  Returns the robust flag.
GBP_DEV bool relin_core(float (&fac)[56], const float (&x0c)[6], const float (&x0l)[3], const float (&K)[9],
                        float var, float nstds, const CamLin& cl /* == cam_lin(x0c[3..5]) */) {
  Lin L;
  jac_hfunc_lin(x0c, x0l, K, cl, L);
  GBP_UNROLL
  for (int i = 0; i < 6; ++i) {
    GBP_UNROLL
    for (int j = 0; j <= i; ++j) {
      float acc = fac[9 + tri(i, j)];
      GBP_UNROLL
      for (int k = 0; k < 2; ++k) acc += L.Jkf[k * 6 + i] * L.Jkf[k * 6 + j];
      fac[9 + tri(i, j)] = acc;
    }
  }
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) {
    GBP_UNROLL
    for (int j = 0; j <= i; ++j) {
      float acc = fac[48 + tri(i, j)];
      GBP_UNROLL
      for (int k = 0; k < 2; ++k) acc += L.Jl[k * 3 + i] * L.Jl[k * 3 + j];
      fac[48 + tri(i, j)] = acc;
    }
  }
  GBP_UNROLL
  for (int i = 0; i < 6; ++i) {
    GBP_UNROLL
    for (int j = 0; j < 3; ++j) {
      float acc = fac[30 + i * 3 + j];
      GBP_UNROLL
      for (int k = 0; k < 2; ++k) acc += L.Jkf[k * 6 + i] * L.Jl[k * 3 + j];
      fac[30 + i * 3 + j] = acc;
    }
  }
  const float z0 = fac[54], z1 = fac[55];
  float buf[2];
  GBP_UNROLL
  for (int r = 0; r < 2; ++r) {
    float acc = 0.f;
    GBP_UNROLL
    for (int k = 0; k < 6; ++k) acc += L.Jkf[r * 6 + k] * x0c[k];
    GBP_UNROLL
    for (int k = 0; k < 3; ++k) acc += L.Jl[r * 3 + k] * x0l[k];
    acc = acc + (r == 0 ? z0 : z1);
    acc = acc - L.hx[r];
    buf[r] = acc;
  }
  GBP_UNROLL
  for (int i = 0; i < 9; ++i) {
    float acc = fac[i];
    GBP_UNROLL
    for (int k = 0; k < 2; ++k) acc += (i < 6 ? L.Jkf[k * 6 + i] : L.Jl[k * 3 + (i - 6)]) * buf[k];
    fac[i] = acc;
  }
  // Huber (gbp_codelets.cpp:135-141): the 0.5 literal makes the denominator a double expression
  const float err = sqrtf((L.hx[0] - z0) * (L.hx[0] - z0) + (L.hx[1] - z1) * (L.hx[1] - z1));
  float mvar = var;
  const bool robust = err > nstds * sqrtf(var);
  if (robust) {
    const double den = 2 * ((double)(nstds * sqrtf(var) * err) - 0.5 * (double)nstds * (double)nstds * (double)var);
    mvar = (float)((double)(var * err * err) / den);
  }
  {  // 54 divisions by one divisor (gbp_codelets.cpp:142-168, 343-373): exact through one fp64 reciprocal
    float num[54], quo[54];
    GBP_UNROLL
    for (int i = 0; i < 54; ++i) num[i] = fac[i];
    div_shared(num, mvar, quo);
    GBP_UNROLL
    for (int i = 0; i < 54; ++i) fac[i] = quo[i];
  }
  return robust;
}

// One factor's share of a sweep on register state: PrepMessageVertex + the four Compute*Message*Vertex classes
// (gbp_codelets.cpp:215-710).  Shared by k_sweep (state streamed from HBM every launch) and k_persist (state kept in
// registers across iterations).  `means(x0c, x0l, cl)` supplies the hoisted linearisation point and the camera-only Jacobian
// terms of that point when a lane relinearises.
template <bool HOIST, class Means>
GBP_DEV void factor_update(float (&fac)[56], const float (&cm)[28], float (&mu)[12], const float (&lm)[16], const float (&cb)[44],
                           const float (&lb)[16], const float (&K)[9], const Hyper& hp, float& damping, int& count, uint32_t& flags,
                           const float var, const bool active, float (&oc_eta)[6], float (&oc_lam)[36], float (&ol)[16], bool& relin,
                           Means&& means) {
  relin = false;
  // (zero messages of an inactive factor are written in the ELSE branch at the bottom: 58 v_mov the wavefronts of a graph
  // with every factor active never execute, instead of an initialisation in front of the branch that all of them do)
  if (active) {
    ol[3] = 0.f; ol[13] = 0.f; ol[14] = 0.f; ol[15] = 0.f;      // (3, 13, 14: the caller's per-factor scalars)
    // ---- PrepMessageVertex, gbp_codelets.cpp:241-378 ----
    if (0 == count) damping = hp.maxeta_damping;
    count += 1;
    float x0c[6], x0l[3];
    CamLin cl;
    float d2;
    if (HOIST) {
      d2 = cb[6];
      d2 += lb[3];
      d2 += lb[13];
      d2 += lb[14];
    } else {
      belief_means(cb, lb, x0c, x0l);
      d2 = 0.f;
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) {
        d2 += (mu[i] - x0c[i]) * (mu[i] - x0c[i]);
        mu[i] = x0c[i];
      }
      GBP_UNROLL
      for (int i = 0; i < 3; ++i) {
        d2 += (mu[6 + i] - x0l[i]) * (mu[6 + i] - x0l[i]);
        mu[6 + i] = x0l[i];
      }
    }
    const float dmu = sqrtf(d2);
    mu[9] = dmu;
    relin = (dmu < hp.dmu_threshold) && (count > hp.min_linear_iters - hp.num_undamped_iters);
    if (relin) {
      if (HOIST) {  // linearisation point = the hoisted means + the camera's CAM_LIN record (rare path: loaded only here)
        means(x0c, x0l, cl);
      } else {
        const float w[3] = {x0c[3], x0c[4], x0c[5]};
        cam_lin(w, cl);
      }
      damping = 0.f;
      count = -hp.num_undamped_iters;
      if (hp.relin_mode == 1) {
        GBP_UNROLL
        for (int i = 0; i < 54; ++i) fac[i] = 0.f;
      }
      const bool robust = relin_core(fac, x0c, x0l, K, var, hp.nstds, cl);
      flags = robust ? (flags | kFlagRobust) : (flags & ~kFlagRobust);
    }

    const float omd = 1 - damping;

    // ---- Compute{Lmk}Message{Eta,Lambda}Vertex, gbp_codelets.cpp:503-562, 664-709 ----
    {
      float Ap[21], Ainv[36], G[18], ed[6];
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) {
        GBP_UNROLL
        for (int j = 0; j <= i; ++j) {
          float t = fac[9 + tri(i, j)] + cb[8 + i * 6 + j];
          t = t - cm[6 + tri(i, j)];
          GBP_SLP_FENCE(t);
          Ap[tri(i, j)] = t;
        }
      }
      inv6x6_lower(Ap, Ainv);
      GBP_UNROLL
      for (int i = 0; i < 3; ++i) {
        GBP_UNROLL
        for (int j = 0; j < 6; ++j) {
          float acc = 0.f;
          GBP_UNROLL
          for (int k = 0; k < 6; ++k) acc += fac[30 + k * 3 + i] * Ainv[k * 6 + j];  // Lambda_lc(i,k) = Lambda_cl(k,i)
          G[i * 6 + j] = acc;
        }
      }
      GBP_UNROLL
      for (int k = 0; k < 6; ++k) {
        float t = fac[k] + cb[k];
        t = t - cm[k];
        ed[k] = t;
      }
      GBP_UNROLL
      for (int i = 0; i < 3; ++i) {
        float s = 0.f;
        GBP_UNROLL
        for (int k = 0; k < 6; ++k) s += G[i * 6 + k] * ed[k];
        const float h = fac[6 + i] - s;
        ol[i] = h * omd + lm[i] * damping;
      }
      GBP_UNROLL
      for (int i = 0; i < 3; ++i) {
        GBP_UNROLL
        for (int j = 0; j < 3; ++j) {
          float t = 0.f;
          GBP_UNROLL
          for (int k = 0; k < 6; ++k) t += G[i * 6 + k] * fac[30 + k * 3 + j];
          ol[4 + i * 3 + j] = fac[48 + trisym(i, j)] - t;
        }
      }
    }
    // ---- Compute{Cam}Message{Eta,Lambda}Vertex, gbp_codelets.cpp:411-471, 592-637 ----
    {
      float Bp[9], Bi[9], G2[18], el[3];
      GBP_UNROLL
      for (int i = 0; i < 3; ++i) {
        GBP_UNROLL
        for (int j = 0; j < 3; ++j) {
          float t = fac[48 + trisym(i, j)] + lb[4 + i * 3 + j];
          t = t - lm[4 + i * 3 + j];
          Bp[i * 3 + j] = t;
        }
      }
      inv3x3(Bp, Bi);
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) {
        GBP_UNROLL
        for (int j = 0; j < 3; ++j) {
          float acc = 0.f;
          GBP_UNROLL
          for (int k = 0; k < 3; ++k) acc += fac[30 + i * 3 + k] * Bi[k * 3 + j];
          G2[i * 3 + j] = acc;
        }
      }
      GBP_UNROLL
      for (int k = 0; k < 3; ++k) {
        float t = fac[6 + k] + lb[k];
        t = t - lm[k];
        el[k] = t;
      }
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) {
        float s = 0.f;
        GBP_UNROLL
        for (int k = 0; k < 3; ++k) s += G2[i * 3 + k] * el[k];
        const float h = fac[i] - s;
        oc_eta[i] = h * omd + cm[i] * damping;
      }
      GBP_UNROLL
      for (int i = 0; i < 6; ++i) {
        GBP_UNROLL
        for (int j = 0; j < 6; ++j) {
          float t = 0.f;
          GBP_UNROLL
          for (int k = 0; k < 3; ++k) t += G2[i * 3 + k] * fac[30 + j * 3 + k];  // Lambda_lc(k,j) = Lambda_cl(j,k)
          oc_lam[i * 6 + j] = fac[9 + trisym(i, j)] - t;
        }
      }
    }
  } else {
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) oc_eta[i] = 0.f;
    GBP_UNROLL
    for (int i = 0; i < 36; ++i) oc_lam[i] = 0.f;
    GBP_UNROLL
    for (int i = 0; i < 16; ++i) ol[i] = 0.f;
  }
}

}  // namespace

// =================================================================================================
// Metric: util.cpp:74-144.  Variable means by an fp64 partial-pivot solve of the fp32 belief
// (stands in for Eigen's general inverse), residuals in fp32, sums in fp64.
// =================================================================================================
// Every loop has compile-time bounds and every row swap is a select, so the 6 x 7 fp64 tableau lives in registers
// (no scratch: this kernel sits on the critical path of the per-iteration metric of small graphs).
template <int N>
GBP_DEV void solve_pivot(const float* A, int lda, const float* b, float* x) {
  double M[N][N + 1];
  GBP_UNROLL
  for (int i = 0; i < N; ++i) {
    GBP_UNROLL
    for (int j = 0; j < N; ++j) M[i][j] = A[i * lda + j];
    M[i][N] = b[i];
  }
  GBP_UNROLL
  for (int k = 0; k < N; ++k) {
    int piv = k;
    double best = fabs(M[k][k]);
    GBP_UNROLL
    for (int i = k + 1; i < N; ++i)
      if (fabs(M[i][k]) > best) { best = fabs(M[i][k]); piv = i; }
    GBP_UNROLL
    for (int i = k + 1; i < N; ++i) {      // swap rows k and piv (at most one i matches)
      const bool sw = piv == i;
      GBP_UNROLL
      for (int j = 0; j <= N; ++j) {
        const double t = M[k][j];
        M[k][j] = sw ? M[i][j] : t;
        M[i][j] = sw ? t : M[i][j];
      }
    }
    GBP_UNROLL
    for (int i = k + 1; i < N; ++i) {
      const double f = M[i][k] / M[k][k];
      GBP_UNROLL
      for (int j = k; j <= N; ++j) M[i][j] -= f * M[k][j];
    }
  }
  GBP_UNROLL
  for (int i = N - 1; i >= 0; --i) {
    double s = M[i][N];
    GBP_UNROLL
    for (int j = i + 1; j < N; ++j) s -= M[i][j] * (double)x[j];
    x[i] = (float)(s / M[i][i]);
  }
}

// health check: a belief Lambda is usable by inv6x6 / inv3x3 only while the un-pivoted
// LDL^T pivots of its lower triangle (matlib.cpp:193-206) stay positive; a non-PD landmark belief is the
// early-warning sign of the blow-ups seen on fr1xyz.
template <int N>
GBP_DEV bool ldl_pivots_positive(const float* A, int lda) {
  double L[N][N], D[N];
  bool ok = true;
  GBP_UNROLL
  for (int j = 0; j < N; ++j) {
    double d = A[j * lda + j];
    GBP_UNROLL
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k] * D[k];
    D[j] = d;
    if (!(d > 0.0)) ok = false;
    GBP_UNROLL
    for (int i = j + 1; i < N; ++i) {
      double v = A[i * lda + j];
      GBP_UNROLL
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k] * D[k];
      L[i][j] = v / d;
    }
  }
  return ok;
}

// One factor's share of the metric (util.cpp:95-129): reprojection residual of the belief means, in two steps — the rotation
// of the camera's mean (eigenso3exp, util.cpp:20-32: a function of the camera alone), then the residual of one factor.  k_eval and
// the metric phase of k_persist evaluate both per factor (eval_factor), the metric that rides in the two-kernel path evaluates
// the first once per camera (k_beliefs<EV>) and the second per factor (k_sweep<EV>): the same operations on the same operands.
GBP_DEV void eval_cam_rot(const float (&cm)[6], float (&R)[9]) {
  // eigenso3exp, util.cpp:20-32 (single expression)
  const float th = sqrtf(cm[3] * cm[3] + cm[4] * cm[4] + cm[5] * cm[5]);
  GBP_UNROLL
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.f : 0.f;
  if (!(th < 1e-6)) {
    const float W[9] = {0.f, -cm[5], cm[4], cm[5], 0.f, -cm[3], -cm[4], cm[3], 0.f};
    const float sa = sinf(th) / th, sb = (1 - cosf(th)) / (th * th);
    GBP_UNROLL
    for (int r = 0; r < 3; ++r) {
      GBP_UNROLL
      for (int c = 0; c < 3; ++c) {
        float ww = 0.f;
        GBP_UNROLL
        for (int k = 0; k < 3; ++k) ww += W[r * 3 + k] * W[k * 3 + c];
        R[r * 3 + c] = R[r * 3 + c] + (sa * W[r * 3 + c] + sb * ww);
      }
    }
  }
}
template <class KF>     // Kd[i]: the pin-hole matrix from a device pointer (k_eval) or from the kernel arguments
GBP_DEV void eval_residual(const float (&R)[9], const float (&t)[3], const float (&lmu)[3], float z0, float z1, KF&& Kd, double& s_norm, double& s_half) {
  float pcf[3], pr[2];
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) pcf[i] = (R[i * 3] * lmu[0] + R[i * 3 + 1] * lmu[1]) + R[i * 3 + 2] * lmu[2];
  GBP_UNROLL
  for (int i = 0; i < 3; ++i) pcf[i] += t[i];
  GBP_UNROLL
  for (int i = 0; i < 2; ++i) pr[i] = ((Kd[i * 3] * pcf[0] + Kd[i * 3 + 1] * pcf[1]) + Kd[i * 3 + 2] * pcf[2]) / pcf[2];
  const float r0 = z0 - pr[0], r1 = z1 - pr[1];
  s_norm += (double)sqrtf(r0 * r0 + r1 * r1);
  s_half += (double)(float)(0.5 * (double)(r0 * r0 + r1 * r1));
}
GBP_DEV void eval_factor(const float (&cm)[6], const float (&lmu)[3], float z0, float z1, const float* Kd, double& s_norm, double& s_half) {
  float R[9];
  eval_cam_rot(cm, R);
  const float t[3] = {cm[0], cm[1], cm[2]};
  eval_residual(R, t, lmu, z0, z1, Kd, s_norm, s_half);
}
// THE ORDER OF THE METRIC'S SUMS (every path produces exactly these fp64 additions):
//   w_t  = the 64 lane values of tile t (0 + the factor's term) reduced by the shuffle tree below;
//   B_b  = ((w_4b + w_4b+1) + w_4b+2) + w_4b+3      the 256 factor positions of sweep workgroup b;
//   S_j  = B_j + B_j+N + B_j+2N + ...  (serially, from 0)   N = eval_blocks(n_tiles) <= 1 024 block sums travel to the host;
//   sum  = S_0 + S_1 + ... (serially, on the host: sum_eval in gbp_api_eval.cpp).
// (lane 0 ends with the sum; what the other lanes end with is not used.  The tree is s += shfl_down(s, off) for off = 32, 16, 8,
// 4, 2, 1; from 8 on the lanes that still matter read inside their own DPP row: a row shift — two v_mov_b32_dpp per double — instead
// of two ds_bpermute round trips through the LDS crossbar per double and step.  Same operands, same additions.)
GBP_DEV double row_shl_f64(double v, const int ctrl_row_shl) {
  const long long b = __double_as_longlong(v);
  int lo = (int)(unsigned)b, hi = (int)(unsigned)((unsigned long long)b >> 32);
  switch (ctrl_row_shl) {      // (the DPP control is an immediate)
    case 8: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x108, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x108, 0xF, 0xF, false); break;
    case 4: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xF, 0xF, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x102, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x102, 0xF, 0xF, false); break;
    default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x101, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x101, 0xF, 0xF, false); break;
  }
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
GBP_DEV void eval_wave_tree(double& s_norm, double& s_half) {
  s_norm += __shfl_down(s_norm, 32); s_half += __shfl_down(s_half, 32);
  s_norm += __shfl_down(s_norm, 16); s_half += __shfl_down(s_half, 16);
  s_norm += row_shl_f64(s_norm, 8); s_half += row_shl_f64(s_half, 8);
  s_norm += row_shl_f64(s_norm, 4); s_half += row_shl_f64(s_half, 4);
  s_norm += row_shl_f64(s_norm, 2); s_half += row_shl_f64(s_half, 2);
  s_norm += row_shl_f64(s_norm, 1); s_half += row_shl_f64(s_half, 1);
}

// One tile's share of the riding metric (EvalRide in gbp_kernels.h): the residual of every factor from its state word, its
// measurement and the metric records of its camera (q0..q2) and landmark (lq); the wave's partial sums go to slot counter - 1 of
// the ring, at index TILE.  Every lane computes (no branch for the callers' loads to sink into); the launch is the same for every
// iteration of a burst, the first included: the slot comes from a counter in device memory, and while that is 0 there is nothing
// to store.  Wave slot 0 also collects what the belief update counted and zeroes the counters for the next one.
// `done` = *ev.counter, read by the caller BEFORE its first store: behind stores the load would be a vector load, and waiting for
// it would wait for every store the wave has in flight (vmcnt counts both, in order: +1.5 us at the end of every sweep wave).
template <class KF>
GBP_DEV void ride_metric(const EvalRide& ev, uint32_t done, uint32_t ws, uint32_t tile, uint32_t lane, int packed, float z0, float z1,
                         const float4 q0, const float4 q1, const float4 q2, const float4 lq, KF&& K) {
  const uint32_t flags = (uint32_t)packed & 7u;
  const bool pad = (flags & kFlagPad) != 0, counts = !pad && (flags & kFlagActive) != 0;
  const float R[9] = {q0.x, q0.y, q0.z, q1.x, q1.y, q1.z, q2.x, q2.y, q2.z}, t[3] = {q0.w, q1.w, q2.w}, lmu[3] = {lq.x, lq.y, lq.z};
  double r_norm = 0, r_half = 0;
  eval_residual(R, t, lmu, z0, z1, K, r_norm, r_half);
  double s_norm = counts ? r_norm : 0.0, s_half = counts ? r_half : 0.0;
  eval_wave_tree(s_norm, s_half);
  const uint32_t n_act = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(counts));
  const uint32_t n_rel = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(!pad && (packed >> 3) == -ev.num_undamped));
  const uint32_t n_rob = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(!pad && (flags & kFlagRobust) != 0));
  if (done != 0u && lane == 0) {
    const uint32_t slot = done - 1u;
    EvalRec r;
    r.sum_norm = s_norm; r.sum_half_sq = s_half; r.n_active = n_act; r.n_relin = n_rel; r.n_robust = n_rob; r.pad = 0;
    ev.part[(size_t)slot * ev.n_tiles + tile] = r;
    if (ws == 0u) {
      ev.slot_health[2 * (size_t)slot] = ev.health[0]; ev.slot_health[2 * (size_t)slot + 1] = ev.health[1];
      ev.health[0] = 0ull; ev.health[1] = 0ull;
    }
  }
}

// =================================================================================================
// k_sweep: one lane = one factor.
// =================================================================================================
// HOIST = true: the belief means are per-VARIABLE quantities (inf2mean of the camera / landmark belief,
// gbp_codelets.cpp:264-265) that the reference recomputes in every incident factor.  The belief kernels
// compute them once per variable; dmu^2 = ((((((0+t0)+..+t5) + u0) + u1) + u2) splits into a per-camera
// prefix S_c (CAMB slot 6) and three per-landmark terms u (LMKB slots 3,13,14), evaluated with the same
// fp32 operations in the same order, so the result is bit-identical while the per-factor MU stream and
// two of the five small inverses disappear from the sweep.  HOIST = false keeps the literal per-factor
// mu/oldmu tensors (needed only if a caller uploads non-zero oldmu).
constexpr int kWpb = 4;      // wavefronts per workgroup of the sweep (the waves of a workgroup share nothing)
// POL: cache policy of the two message streams (SweepArgs.policy, chosen per graph shape by gbp_api_ctx.cpp; a template parameter,
// not a branch on the flag: with both load sequences behind a branch the non-temporal path lost 1.2 %)
// EV: the metric of the PREVIOUS iteration rides in this sweep (EvalRide in gbp_kernels.h)
// SEG: the tile's all-pad 64-byte segments are neither loaded nor stored (SweepArgs.seg_live, load_tile_seg above)
template <bool HOIST, uint32_t POL = 0, bool EV = false, bool SEG = false>
GBP_DEV void sweep_tile(const SweepArgs& a, const uint32_t wslot) {
  // (the slot is wave-uniform: as an SGPR it turns the permutation look-up into one scalar load)
  const uint32_t ws = (uint32_t)__builtin_amdgcn_readfirstlane((int)wslot);
  const uint32_t tile = a.tile_perm ? a.tile_perm[ws] : ws;
  const uint32_t lane = threadIdx.x & 63, p = tile * 64 + lane;
  const uint32_t ev_done = EV ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*a.ev.counter) : 0u;     // (a scalar load, up here)

  const uint32_t cam_i = a.row_cam[p >> 4];
  const uint32_t lmk_i = __builtin_nontemporal_load(a.lmk_idx + p);

  // SEG: bit s of the tile's mask = the 64-byte segment of lanes 4s .. 4s + 3 holds at least one factor (one scalar load)
  const uint32_t segm = SEG ? a.seg_live[tile] : 0xffffu;
  const bool live8 = !SEG || ((segm >> (lane >> 2)) & 1u) != 0u;
  float fac[56], cm[28], mu[12], lm[16], cb[44], lb[16];
  if (SEG) load_tile_seg<kFacG>(a.fac, tile, lane, live8, fac);
  else load_tile<kFacG>(a.fac, tile, lane, fac);
  // The camera messages: non-temporal like the potentials, or — SweepArgs.cmsg_cached, graphs with few cameras — with the
  // default policy like the landmark messages below (both are rewritten in place by this tile).  The potentials, which an
  // ordinary sweep only reads, keep the hint on every graph: with default-policy loads they cost 3 %.
  if (SEG) load_tile_seg<kCmsgG, !(POL & kPolCmsgLoadCached)>(a.cmsg, tile, lane, live8, cm);
  else load_tile<kCmsgG, !(POL & kPolCmsgLoadCached)>(a.cmsg, tile, lane, cm);
  if (!HOIST) load_tile<kMuG>(a.mu, tile, lane, mu);
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
  // SEG: access k moves the records 16k .. 16k + 15 (four lanes each): lanes 16j .. 16j + 15 the records of segment 4k + j
  const __amdgpu_buffer_rsrc_t lm_rsrc = tile_rsrc<4>(a.lmsg, SEG ? tile : 0u);
  GBP_UNROLL
  for (int k = 0; k < 4; ++k) {
    // (DEFAULT policy for this one stream unless the shape says otherwise: the tile is rewritten in place ten microseconds later
    // and gathered by k_beliefs right after the sweep — measured +1.5 % iterations/s on the 1M-factor graph against the
    // non-temporal hint, with either store policy; the potentials keep the hint on every graph)
    const uint32_t r = k * 16 + rec_t;
    if (SEG) {
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      const bool live_k = ((segm >> (4 * k + (int)(lane >> 4))) & 1u) != 0u;
      const v4u v = __builtin_amdgcn_raw_buffer_load_b128(lm_rsrc, (live_k ? (int)(lane * 16u) : kBufOob) + k * 1024, 0, (POL & kPolLmsgLoadNt) ? 2 : 0);
      stage[r * 4 + ((lane & 3u) ^ (((r >> 2) & 3u) ^ (r & 2u)))] =
          make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    } else {
      const v4f* src = reinterpret_cast<const v4f*>(lm_tile) + k * 64 + lane;
      const v4f v = (POL & kPolLmsgLoadNt) ? __builtin_nontemporal_load(src) : *src;
      stage[r * 4 + ((lane & 3u) ^ (((r >> 2) & 3u) ^ (r & 2u)))] = make_float4(v.x, v.y, v.z, v.w);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  GBP_UNROLL
  for (int q = 0; q < 4; ++q) {
    const float4 v = stage[lane * 4 + ((uint32_t)q ^ swz_own)];
    lm[4 * q] = v.x; lm[4 * q + 1] = v.y; lm[4 * q + 2] = v.z; lm[4 * q + 3] = v.w;
  }
  load_rec<kCamRec4>(a.camb + (size_t)cam_i * kCamRec4, cb);
  load_rec<kLmkRec4>(a.lmkb + (size_t)lmk_i * kLmkRec4, lb);
  // EV: the camera's metric record (one address per 16-lane row) and the landmark's metric mean (a 16-byte gather from a table
  // 1/4 the size of the beliefs') go out WITH the belief gathers — unconditionally: pads index 0 — and wait in a wave-private
  // LDS area for the end of the tile, where the registers are free: 16 more live registers in this prologue (50 loads in
  // flight, 236 of 256 VGPRs) made the compiler issue the belief gathers BEHIND the metric — two dependent round trips per
  // wave, +8 us per sweep (the ISA showed it, profiles/r05_default_loop.md).
  __shared__ float4 ev_stage[EV ? kWpb : 1][EV ? 64 * 4 : 1];
  float4 evq0, evq1, evq2, evq3;
  if (EV) {
    evq0 = a.ev.cam_rec[(size_t)cam_i * 3]; evq1 = a.ev.cam_rec[(size_t)cam_i * 3 + 1]; evq2 = a.ev.cam_rec[(size_t)cam_i * 3 + 2];
    evq3 = a.ev.lmk_mean[lmk_i];
    // (a compiler barrier: without it the belief gathers above — used only by active lanes — are sunk into that branch, behind
    // the wait for the metric records)
    asm volatile("" ::: "memory");
  }
  // per-factor scalar state rides in the pad slots of the landmark-message record (read and rewritten
  // every sweep anyway): [3] damping, [13] (damping_count << 3) | flags, [14] measurement variance
  if (SEG && !live8) lm[13] = __int_as_float((int)kFlagPad);      // (the record that was not loaded: a pad, as in memory)
  float damping = lm[3];
  const int packed = __float_as_int(lm[13]);
  int count = packed >> 3;
  uint32_t flags = (uint32_t)packed & 7u;
  const float var = lm[14];
  const bool active = (flags & kFlagActive) != 0;

  float K[9];
  GBP_UNROLL
  for (int i = 0; i < 9; ++i) K[i] = a.K[i];

  const int ev_packed = packed;      // EV: the factor's state word as the sweep of the evaluated iteration left it
  const float ev_z0 = fac[54], ev_z1 = fac[55];
  if (EV) {
    float4* park = ev_stage[threadIdx.x >> 6];
    park[lane] = evq0; park[64 + lane] = evq1; park[128 + lane] = evq2; park[192 + lane] = evq3;
  }

  float oc_eta[6], oc_lam[36], ol[16];
  bool relin;
  factor_update<HOIST>(fac, cm, mu, lm, cb, lb, K, a.hp, damping, count, flags, var, active, oc_eta, oc_lam, ol, relin,
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
    if (SEG) {
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      const bool live_k = ((segm >> (4 * k + (int)(lane >> 4))) & 1u) != 0u;
      const v4u v = {__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w)};
      __builtin_amdgcn_raw_buffer_store_b128(v, lm_rsrc, (live_k ? (int)(lane * 16u) : kBufOob) + k * 1024, 0, (POL & kPolLmsgStoreNt) ? 2 : 0);
    } else if (POL & kPolLmsgStoreNt) {
      const v4f v = {f.x, f.y, f.z, f.w};
      __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(lm_tile) + k * 64 + lane);
    } else {
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
    if (SEG) store_tile_seg<kCmsgG>(a.cmsg, tile, lane, live8, cmo);
    else store_tile<kCmsgG>(a.cmsg, tile, lane, cmo);
  }
  // camera half of the belief reduction: per-row (16 factors of one camera) tree sums
  {
    float4* rp = a.rowp + (size_t)(p >> 4) * kCamRec4;
    row16_sums_store(oc_eta, oc_lam, lane, [&](uint32_t g, float4 v) { rp[g] = v; });
  }
  if (active) {
    if (!HOIST) store_tile<kMuG>(a.mu, tile, lane, mu);
    if (relin) store_tile<kFacG>(a.fac, tile, lane, fac);
  }
  if (EV) {
    // The metric of the iteration that has just ended (EvalRide in gbp_kernels.h), behind this tile's own work: the factor's
    // state word as ITS sweep left it, the measurement, the metric records parked above.
    const float4* park = ev_stage[threadIdx.x >> 6];
    ride_metric(a.ev, ev_done, ws, tile, lane, ev_packed, ev_z0, ev_z1, park[lane], park[64 + lane], park[128 + lane], park[192 + lane], K);
  }
}

template <bool HOIST, uint32_t POL = 0, bool EV = false, bool SEG = false>
__global__ __launch_bounds__(64 * kWpb) void k_sweep(const SweepArgs a) {
  sweep_tile<HOIST, POL, EV, SEG>(a, blockIdx.x * kWpb + (threadIdx.x >> 6));
}

// =================================================================================================
// k_linearise: RelineariseFactorVertex on every factor (no active_flag test in the reference).
// =================================================================================================
__global__ __launch_bounds__(256) void k_linearise(const SweepArgs a) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  const uint32_t tile = p >> 6, lane = p & 63;
  const uint32_t cam_i = a.row_cam[p >> 4], lmk_i = a.lmk_idx[p];
  float4 st = a.lmsg[(size_t)p * 4 + 3];   // record slots 12..15: y = packed count/flags, z = variance
  int packed = __float_as_int(st.y);
  uint32_t flags = (uint32_t)packed & 7u;
  if (flags & kFlagPad) return;
  float fac[56], cb[44], lb[16], K[9], x0c[6], x0l[3];
  load_tile<kFacG>(a.fac, tile, lane, fac);
  load_rec<kCamRec4>(a.camb + (size_t)cam_i * kCamRec4, cb);
  load_rec<kLmkRec4>(a.lmkb + (size_t)lmk_i * kLmkRec4, lb);
  GBP_UNROLL
  for (int i = 0; i < 9; ++i) K[i] = a.K[i];
  GBP_UNROLL
  for (int i = 0; i < 54; ++i) fac[i] = 0.f;
  belief_means(cb, lb, x0c, x0l);
  CamLin cl;
  const float wv[3] = {x0c[3], x0c[4], x0c[5]};
  cam_lin(wv, cl);
  const bool robust = relin_core(fac, x0c, x0l, K, st.z, a.hp.nstds, cl);
  flags = robust ? (flags | kFlagRobust) : (flags & ~kFlagRobust);
  st.y = __int_as_float((packed & ~7) | (int)flags);
  a.lmsg[(size_t)p * 4 + 3] = st;
  store_tile<kFacG>(a.fac, tile, lane, fac);
}

// =================================================================================================
// k_beliefs: buildUpdateBeliefsProg (ba.cpp:104-139) in ONE launch.
//   blocks [0, cam_blocks): camera part, one wavefront per camera (lanes 0..43 = the 44-float record);
//   remaining blocks:       landmark part, four lanes per landmark (lane q = float4 #q of the record).
// Camera beliefs: local = rows of this camera left to right (rows were tree-summed by k_sweep), then
// belief = prior + local (single GPU) or prior + sum_r gathered[r] (after the all-gather, rank order).
// Landmark beliefs: prior + messages in slot order.  With `hoist` the belief means (inf2mean, bafuncs.cpp:2-15)
// and the dmu^2 pieces of the next sweep are computed here once per variable (see k_sweep<HOIST>).
// =================================================================================================
// float4 #q of the landmark-message record at device position pos, with the per-factor state that rides in
// the record's pad slots (3, 13, 14, 15) blanked so that it never enters a belief sum.  Two steps, and the loads of a batch are
// issued UNCONDITIONALLY (unused slots hold position 0: a valid record) before anything looks at a loaded value: written as
// `k < deg ? load : 0` every gather became a branch with its own `s_waitcnt vmcnt(0)` at the join — ten dependent round trips
// per wave instead of ten loads in flight (the ISA showed it; `k_persist` had hit the same thing with its sc1 loads).
GBP_DEV float4 lmsg_load(const float4* lmsg, uint32_t pos, uint32_t q) {
  return lmsg[(size_t)pos * 4 + q];   // default cache policy: a non-temporal hint here costs 4 us (the 128-B line's other half is a neighbour's record)
}
GBP_DEV float4 lmsg_blank(float4 m, uint32_t q) {
  if (q == 0) m.w = 0.f;
  if (q == 3) { m.y = 0.f; m.z = 0.f; m.w = 0.f; }
  return m;
}

// mean of a camera belief record (44 floats, in registers or in LDS): inf2mean6x6 (bafuncs.cpp:2-9)
template <class REC>
GBP_DEV void cam_mean(REC&& cb, float (&x0c)[6]) {
  solve6_lower([&](int i, int j) { return cb[8 + i * 6 + j]; }, [&](int k) { return cb[k]; }, x0c);
}

// EV: the belief owners also leave the METRIC RECORDS of the new beliefs (EvalRide in gbp_kernels.h): what k_means computes.
// CAM_ONLY: a launch without landmark blocks (the camera combine behind the exchange of a sharded iteration, prior-only refreshes of the
// split-phase path): the landmark half is compiled out, so the kernel needs the camera half's registers only — 8 waves per SIMD instead of
// 7, i.e. the 2 000 workgroups of an 8 000-camera graph resident in ONE generation (256 CUs x 8) instead of one and a bit
template <bool EV, bool CAM_ONLY = false>
GBP_DEV void beliefs_body(const BeliefArgs& b) {
  __shared__ float sh[4][48];
  __shared__ float lrec[EV ? 64 : 1][13];      // EV: the beliefs of the workgroup's 64 landmarks (eta 3, Lambda 9; 13: bank spread)
  if (EV && blockIdx.x == 0 && threadIdx.x == 0) *b.ev.counter = *b.ev.counter + 1u;     // one more iteration of the burst done (read by the NEXT sweep)
  // Which blocks are the cameras'?  The FIRST cam_blocks of the grid where a camera block ends in the serial chain of its means (long
  // latency: started first, hidden under the landmark blocks) — the LAST ones where it only adds up rows (partial_only, before the
  // exchange of a sharded iteration): short work that fills the slots the landmark blocks free as they drain, instead of holding the
  // first generation of the grid (config-5 shard shape: profiles/r06_sharded_timeline.md)
  const bool cams_last = !CAM_ONLY && b.partial_only != 0 && b.lmk_blocks != 0u;
  const bool cam_block = cams_last ? blockIdx.x >= b.lmk_blocks : blockIdx.x < b.cam_blocks;
  if (cam_block) {
    const uint32_t w = threadIdx.x >> 6, j = threadIdx.x & 63;
    const uint32_t cblk = cams_last ? blockIdx.x - b.lmk_blocks : blockIdx.x;
    const uint32_t c = cblk * 4 + w;
    const bool live = c < b.n_cams && j < (uint32_t)kCamRec;
    float bel = 0.f;
    if (live) {
      // WeakenPriorVertex (gbp_codelets.cpp:176-197) rides in the belief refresh WEAKEN_PRIORS ends with (ba.cpp:863-865): the
      // owner of a prior element scales it on its way into the sum and writes it back; lane 0 then counts the flag down (every
      // lane of the record sits in this wave and has read the flag by then)
      float prior = b.cam_prior[(size_t)c * kCamRec + j];
      const uint32_t wf = b.weaken ? b.cam_wflag[c] : 0u;
      if (wf >= 1u && wf <= 5u) {
        prior *= b.cam_scale[c];
        b.cam_prior_rw[(size_t)c * kCamRec + j] = prior;
      }
      if (b.gathered) {
        const float* g = b.gathered + (size_t)c * kCamRec + j;      // exchange layout: [world][C][44]
        float acc = prior;
        for (int r0 = 0; r0 < b.world; r0 += 8) {      // the partials of eight ranks in flight at once (clamped, unconditional), added in rank order
          float v[8];
          GBP_UNROLL
          for (int k = 0; k < 8; ++k) v[k] = g[(size_t)(r0 + k < b.world ? r0 + k : b.world - 1) * b.n_cams * kCamRec];
          GBP_UNROLL
          for (int k = 0; k < 8; ++k)
            if (r0 + k < b.world) acc = acc + v[k];
        }
        bel = acc;
      } else {
        const uint32_t r0 = b.cam_row_ptr[c], r1 = b.cam_row_ptr[c + 1];
        float acc = 0.f;
        if (r1 > r0 && !b.row_slot) {
          const float* row = b.rowp + (size_t)r0 * kCamRec + j;
          acc = row[0];
          uint32_t r = 1;
          const uint32_t n = r1 - r0;
          for (; r + 16 <= n; r += 16) {  // loads of 16 rows in flight, adds in row order
            float v[16];
            GBP_UNROLL
            for (int k = 0; k < 16; ++k) v[k] = row[(size_t)(r + k) * kCamRec];
            GBP_UNROLL
            for (int k = 0; k < 16; ++k) acc = acc + v[k];
          }
          {  // tail (< 16 rows) in one batch as well, the loads unconditional (row clamped, value dropped: a conditional load is a
             // branch with its own wait): this chain is pure load latency
            float v[16];
            const uint32_t m = n - r;
            GBP_UNROLL
            for (int k = 0; k < 16; ++k) v[k] = row[(size_t)((uint32_t)k < m ? r + k : n - 1u) * kCamRec];
            GBP_UNROLL
            for (int k = 0; k < 16; ++k)
              if ((uint32_t)k < m) acc = acc + v[k];
          }
        } else if (r1 > r0) {
          // the camera's rows sit where the row placement put them (gbp_layout.cpp): the same sums in the same order, each row
          // found through row_slot (wave-uniform indices: one round of scalar loads in front of the rows')
          const float* base = b.rowp + j;
          const uint32_t n = r1 - r0;
          acc = base[(size_t)b.row_slot[r0] * kCamRec];
          for (uint32_t r = 1; r < n; r += 16) {
            uint32_t sl[16];
            float v[16];
            const uint32_t m = n - r;
            GBP_UNROLL
            for (int k = 0; k < 16; ++k) sl[k] = b.row_slot[r0 + ((uint32_t)k < m ? r + k : n - 1u)];      // clamped, unconditional
            GBP_UNROLL
            for (int k = 0; k < 16; ++k) v[k] = base[(size_t)sl[k] * kCamRec];
            GBP_UNROLL
            for (int k = 0; k < 16; ++k)
              if ((uint32_t)k < m) acc = acc + v[k];
          }
        }
        b.cam_local[(size_t)c * kCamRec + j] = acc;
        bel = prior + acc;
      }
      sh[w][j] = bel;
      if (j == 0 && wf >= 1u && wf <= 5u) b.cam_wflag[c] = wf - 1u;
    }
    if (b.partial_only) return;
    __syncthreads();
    // The 6x6 mean of a camera is one serial instruction stream on ONE lane.  The four cameras of the workgroup share a
    // single stream (lanes 0..3 of wave 0) instead of issuing it from four wavefronts: with thousands of cameras the
    // camera part is bound by exactly these issue slots (8 000 cameras: 11 -> 4 us).
    if (b.hoist && w == 0 && j < 4) {
      const uint32_t cj = cblk * 4 + j;
      if (cj < b.n_cams) {
        float x0c[6];
        cam_mean(sh[j], x0c);                    // operands straight from LDS: the 44-float copy cost 14 VGPRs of occupancy
        float4* mu = b.cam_mu + (size_t)cj * 4;  // [0,1] = means of the current belief, [2,3] = means the last sweep used
        float4 u0 = mu[2], u1 = mu[3];
        if (b.roll) { u0 = mu[0]; u1 = mu[1]; mu[2] = u0; mu[3] = u1; }
        const float used[6] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y};
        float S = 0.f;
        GBP_UNROLL
        for (int i = 0; i < 6; ++i) S += (used[i] - x0c[i]) * (used[i] - x0c[i]);
        mu[0] = make_float4(x0c[0], x0c[1], x0c[2], x0c[3]);
        mu[1] = make_float4(x0c[4], x0c[5], 0.f, 0.f);
        sh[j][6] = S;
        // camera-only Jacobian terms of this mean (gbp_device_math.hpp: cam_lin), once per camera instead of once per
        // relinearising factor
        CamLin cl;
        const float wv[3] = {x0c[3], x0c[4], x0c[5]};
        cam_lin(wv, cl);
        float4 q[kCamLin4];
        cam_lin_pack(cl, q);
        GBP_UNROLL
        for (int g = 0; g < kCamLin4; ++g) b.cam_lin[(size_t)cj * kCamLin4 + g] = q[g];
      }
    }
    if (EV && (w == 1 || w == 2) && j < 4) {
      // The metric's share of a camera, on the workgroup's OTHER waves (three long serial chains on one lane would set the
      // length of the whole kernel): wave 1 solves the fp64 pivoted means of the four cameras (util.cpp:103-105 stand-in, as
      // k_means) and leaves rotation + translation for the residuals, wave 2 checks the LDL^T pivots.  Both read the belief
      // (eta at sh[.][0..5], Lambda at sh[.][8..43]); wave 0 writes sh[.][6] only.  The 6 x 7 fp64 tableau takes the kernel from
      // 68 to 96 VGPRs = from 7 to 5 waves per SIMD, which costs the landmark part 1.5 us; measured and worse: the occupancy
      // pinned at 7 (the tableau spills: 20.1 us against 18.3), the tableau in LDS (21.4 - 25.1 us: the camera chain then sets the
      // kernel's length) — profiles/r05_default_loop.md.
      const uint32_t cj = cblk * 4 + j;
      if (cj < b.n_cams) {
        if (w == 1) {
          float xm[6], R[9];
          solve_pivot<6>(sh[j] + 8, 6, sh[j], xm);
          bool finite = true;
          GBP_UNROLL
          for (int i = 0; i < 6; ++i) finite &= (xm[i] - xm[i] == 0.f);
          if (!finite) atomicAdd(&b.ev.health[0], 1ull);
          eval_cam_rot(xm, R);
          float4* rec = b.ev.cam_rec + (size_t)cj * 3;
          rec[0] = make_float4(R[0], R[1], R[2], xm[0]);
          rec[1] = make_float4(R[3], R[4], R[5], xm[1]);
          rec[2] = make_float4(R[6], R[7], R[8], xm[2]);
        } else if (!ldl_pivots_positive<6>(sh[j] + 8, 6)) {
          atomicAdd(&b.ev.health[1], 1ull);
        }
      }
    }
    __syncthreads();
    if (live) b.camb[(size_t)c * kCamRec + j] = sh[w][j];
    return;
  }
  if (CAM_ONLY) return;
  // ---- landmark part ----
  uint32_t lb = cams_last ? blockIdx.x : blockIdx.x - b.cam_blocks;
  if (b.lmk_xcd_order) {
    // XCD-aware order: the blocks that share an XCD (equal index mod 8 under round-robin placement) take one contiguous
    // run of landmarks.  The 64-B message records of two neighbouring factors share a 128-B line, and the partner's
    // landmark is a near neighbour (factors are sorted by landmark inside a camera): with both in one XCD's L2 about
    // half of the partner fetches disappear (k_beliefs: 25.6 -> 21.3 us on the 1M-factor graph).
    const uint32_t qn = b.lmk_blocks / 8, rn = b.lmk_blocks % 8, g = lb & 7u;
    lb = g * qn + (g < rn ? g : rn) + (lb >> 3);
  }
  const uint32_t t = lb * 256 + threadIdx.x;
  const uint32_t l = t >> 2, q = t & 3;
  const bool live = l < b.n_lmks;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // LMK_IX[l] = {degree, position of slot 1 .. slot 15} (64 B): ONE load tells the quad where the landmark's records
  // are, so the whole sum is two dependent round trips (index record -> up to 15 message records in flight) instead
  // of four (lmk_ptr -> fpos -> 8 records -> tail).  Slots beyond 15 (rare) go through lmk_ptr / lmk_fpos.
  uint4 ix = make_uint4(0u, 0u, 0u, 0u);
  float4 used_mu = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    ix = reinterpret_cast<const uint4*>(b.lmk_ix)[(size_t)l * 4 + q];
    acc = b.lmk_prior[(size_t)l * 4 + q];
    if (b.weaken) {        // WeakenPriorVertex on the landmark's prior (see the camera part): the quad's four lanes share a wave
      const uint32_t wf = b.lmk_wflag[l];
      if (wf >= 1u && wf <= 5u) {
        const float sc = b.lmk_scale[l];
        acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
        b.lmk_prior_rw[(size_t)l * 4 + q] = acc;
        if (q == 0) b.lmk_wflag[l] = wf - 1u;
      }
    }
    // the mean the last sweep used (roll: the current one becomes it) goes out with the first round of loads: fetched where it
    // is consumed — behind the gathers — it was a third dependent round trip of every wave
    if (b.hoist && q == 0) used_mu = b.lmk_mu[(size_t)l * 2 + (b.roll ? 0 : 1)];
  }
  const uint32_t deg = (uint32_t)__shfl((int)ix.x, 0, 4);
  {
    // element k + 1 of the index record sits in lane (k + 1) / 4, component (k + 1) % 4 of the quad
    auto slot_pos = [&](int k) -> uint32_t {
      const uint32_t v = ((k + 1) & 3) == 0 ? ix.x : ((k + 1) & 3) == 1 ? ix.y : ((k + 1) & 3) == 2 ? ix.z : ix.w;
      return (uint32_t)__shfl((int)v, (k + 1) >> 2, 4);
    };
    // Ten gathers in flight, then (only where a landmark of the wave has more than ten factors) the other five: 40
    // instead of 60 staging registers (1M-factor graph: k_beliefs 20.4 -> 18.8 us).
    float4 m[10];
    GBP_UNROLL
    for (int k = 0; k < 10; ++k) m[k] = lmsg_load(b.lmsg, slot_pos(k), q);      // all ten in flight (see lmsg_load)
    GBP_UNROLL
    for (int k = 0; k < 10; ++k) {   // adds in slot order
      const float4 v = lmsg_blank(m[k], q);
      if ((uint32_t)k < deg) { acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w; }
    }
    if (__any(deg > 10u)) {
      GBP_UNROLL
      for (int k = 10; k < 15; ++k) m[k - 10] = lmsg_load(b.lmsg, slot_pos(k), q);
      GBP_UNROLL
      for (int k = 10; k < 15; ++k) {
        const float4 v = lmsg_blank(m[k - 10], q);
        if ((uint32_t)k < deg) { acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w; }
      }
    }
  }
  if (deg > 15u) {   // slots 16.. : positions from lmk_fpos, 8 record gathers in flight per round, adds in slot order
    const uint32_t s1 = b.lmk_ptr[l + 1];
    for (uint32_t s = b.lmk_ptr[l] + 15u; s < s1; s += 8) {
      uint32_t pos[8];
      float4 m[8];
      const uint32_t nleft = s1 - s;
      GBP_UNROLL
      for (int k = 0; k < 8; ++k) pos[k] = b.lmk_fpos[(uint32_t)k < nleft ? s + k : s1 - 1u];     // clamped, unconditional
      GBP_UNROLL
      for (int k = 0; k < 8; ++k) m[k] = lmsg_load(b.lmsg, pos[k], q);
      GBP_UNROLL
      for (int k = 0; k < 8; ++k) {
        const float4 v = lmsg_blank(m[k], q);
        if ((uint32_t)k < nleft) { acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w; }
      }
    }
  }
  if (b.hoist) {
    // gather the 16-float record of the quad into its lane 0 (all lanes execute the shuffles)
    float rec[16];
    GBP_UNROLL
    for (int k = 0; k < 4; ++k) {
      rec[4 * k] = __shfl(acc.x, k, 4); rec[4 * k + 1] = __shfl(acc.y, k, 4);
      rec[4 * k + 2] = __shfl(acc.z, k, 4); rec[4 * k + 3] = __shfl(acc.w, k, 4);
    }
    float u[3] = {0.f, 0.f, 0.f};
    if (live && q == 0) {
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
      float4* mu = b.lmk_mu + (size_t)l * 2;  // [0] = mean of the current belief, [1] = mean the last sweep used
      const float4 used = used_mu;
      if (b.roll) mu[1] = used;
      u[0] = (used.x - x0l[0]) * (used.x - x0l[0]);
      u[1] = (used.y - x0l[1]) * (used.y - x0l[1]);
      u[2] = (used.z - x0l[2]) * (used.z - x0l[2]);
      mu[0] = make_float4(x0l[0], x0l[1], x0l[2], 0.f);
      if (EV) {   // the belief (eta 3, Lambda 9) for the metric below
        GBP_UNROLL
        for (int i = 0; i < 3; ++i) lrec[threadIdx.x >> 2][i] = rec[i];
        GBP_UNROLL
        for (int i = 0; i < 9; ++i) lrec[threadIdx.x >> 2][3 + i] = rec[4 + i];
      }
    }
    if (EV) {
      // The landmarks' metric means (what k_means computes: fp64 pivoted solve, LDL^T pivots) — long fp64 chains that one lane
      // in four would run at a quarter of the wave's width in each of the four waves: the 64 landmarks of the workgroup meet in
      // LDS and ONE wave solves them, all lanes busy (k_beliefs<EV> 18.6 -> measured in profiles/r05_default_loop.md).
      __syncthreads();
      if (threadIdx.x < 64) {
        const uint32_t le = lb * 64 + threadIdx.x;
        if (le < b.n_lmks) {
          float r12[12], x[3];
          GBP_UNROLL
          for (int i = 0; i < 12; ++i) r12[i] = lrec[threadIdx.x][i];
          solve_pivot<3>(r12 + 3, 3, r12, x);
          bool finite = true;
          GBP_UNROLL
          for (int i = 0; i < 3; ++i) finite &= (x[i] - x[i] == 0.f);
          b.ev.lmk_mean[le] = make_float4(x[0], x[1], x[2], 0.f);
          if (!finite) atomicAdd(&b.ev.health[0], 1ull);
          if (!ldl_pivots_positive<3>(r12 + 3, 3)) atomicAdd(&b.ev.health[1], 1ull);
        }
      }
    }
    const float u0 = __shfl(u[0], 0, 4), u1 = __shfl(u[1], 0, 4), u2 = __shfl(u[2], 0, 4);
    if (q == 0) acc.w = u0;                       // record slot 3
    if (q == 3) { acc.y = u1; acc.z = u2; }       // record slots 13, 14
  }
  if (live) b.lmkb[(size_t)l * 4 + q] = acc;
}
__global__ __launch_bounds__(256) void k_beliefs(const BeliefArgs b) { beliefs_body<false>(b); }      // (held at 8 waves per SIMD: S1 +0.1 %, config-5 shape +0.2 %: noise — profiles/HISTORY.md)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_beliefs_cam(const BeliefArgs b) { beliefs_body<false, true>(b); }
__global__ __launch_bounds__(256) void k_beliefs_ev(const BeliefArgs b) { beliefs_body<true>(b); }

// =================================================================================================
// k_persist: n iterations of {k_sweep; k_beliefs} in ONE launch, for graphs whose workgroups are all resident at once.
//
// A graph of a few thousand factors (BASELINE configs 1-3: fr1xyz = 204 wavefronts on 1 024 SIMDs) is bound by what
// happens BETWEEN its kernels: a dependent launch costs ~3.5 us on this stack and every kernel starts with cold L2s
// (profiles/r03_small_graphs.md), while the arithmetic of a whole iteration is ~10 us.  Here the iteration loop lives
// inside the kernel:
//   phase A  wave w sweeps tile w with factor_update() — the same per-lane arithmetic as k_sweep; the factor's
//            potential and both of its messages stay in REGISTERS across iterations (FAC / CMSG are written back once,
//            after the last iteration); the landmark-message tile and the row sums go to memory for phase B;
//   barrier  device-wide (grid_sync below);
//   phase B  wave w owns camera w, or 16 landmarks (4 lanes each) — the arithmetic of k_beliefs; row pointers, the
//            landmark index record and the priors stay in registers, so the phase is ONE round of loads;
//   barrier  (not after the last iteration).
// Results are bit-identical to the two-kernel path (same per-lane operations in the same order); every array the other
// kernels and the C-ABI read (FAC, CMSG, LMSG, ROWP, CAMB, LMKB, the hoisted means, cam_local) is left exactly as n
// launches of k_sweep + k_beliefs leave it.  Hoisted means only (gbp_params.per_factor_mu = 0), single-GPU ctx only.
// =================================================================================================
// Device-wide hand-off without cache maintenance.  The 8 XCDs have private, mutually incoherent L2s; the textbook grid
// barrier (release: write the L2 back, acquire: invalidate it) costs 5.5 us for 51 workgroups on this chip — more than
// the 3.7 us of a kernel boundary (profiles/r03_small_graphs.md).  Instead every access to the arrays that cross waves
// inside the launch (LMSG, ROWP, CAMB, LMKB, the hoisted means) is an agent-scope relaxed atomic: stores are written
// through to memory, loads are re-fetched (global_load/store_dwordx2 ... sc1), so the barrier itself only has to wait for
// this wave's stores (s_waitcnt vmcnt(0)), meet the workgroup, and count one arrival per workgroup: 1.4 us for 51
// workgroups.  The counter only grows (epoch e expects e * n arrivals); the host zeroes it before the launch.  The wait
// is bounded: if a workgroup were not resident (the launcher checks occupancy, so it is) the kernel raises *status and
// carries on instead of hanging the GPU.
// An array that crosses waves inside the launch ("xw"): every access is sc1 — through a buffer descriptor, because the
// buffer intrinsics give 16-byte sc1 accesses the compiler tracks (agent-scope atomics stop at 8 bytes, and 8-byte accesses
// move at 0.54-0.70x the 16-byte rate, MI355X_MICROARCH.md).  Offsets are 32-bit: arrays below 4 GiB, which the size limit of
// k_persist guarantees by a factor of a thousand.
// VER (test-hooks builds only, gbp_debug_persist_verify): every record is published TWICE — the record itself and, `mirror4` float4
// further on, a copy with the payload words complemented and the same tag — and a load hands a record to its consumer only once BOTH
// copies carry the same tag (until then the record reads as "not arrived": tag 0, the consumer keeps polling); the two payloads must
// then be bit-wise complements, or the load counts a mismatch in *verr.  A torn or stale 16-byte record, which the product's consumers
// could not tell from a good one, shows up as such a mismatch (tests/test_gpu_parity.py: test_persistent_kernel_redundant_records).
template <bool VER>
struct XwBufT {
  __amdgpu_buffer_rsrc_t r;
  uint32_t mirror4 = 0;
  unsigned long long* verr = nullptr;
  // soff: byte offset of this array inside the allocation the descriptor covers — the scalar offset operand of the buffer instructions,
  // an add the address unit does for nothing (and which its range check ignores: kNone4 stays out of range).  k_persist_flow describes
  // its nine shadows, which live in ONE allocation, with one descriptor + nine such offsets instead of nine descriptors: 23 fewer
  // live SGPRs in a kernel that spills them (profiles/r06_resources.md)
  uint32_t soff = 0;
  GBP_DEV explicit XwBufT(const void* base, uint32_t mirror4_ = 0, unsigned long long* verr_ = nullptr, uint32_t soff_ = 0)
      : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000)), mirror4(mirror4_), verr(verr_), soff(soff_) {}
  static constexpr int kSc1 = 16;        // cache-policy bit of the gfx94x / gfx950 buffer instructions
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  GBP_DEV float4 ld4(uint32_t i4) const {   // float4 #i4 of the array
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i4 * 16u), (int)soff, kSc1);
    if (VER) {
      if (mirror4 != 0u && i4 < kNone4) {
        const v4u m = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((i4 + mirror4) * 16u), (int)soff, kSc1);
        if (m.w != v.w) return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), 0.f);      // one copy has not arrived yet
        if ((m.x != ~v.x || m.y != ~v.y || m.z != ~v.z) && v.w != 0u) atomicAdd(verr, 1ull);
      }
    }
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
  // float4 #kNone4 lies beyond the descriptor's range (0x7fffffff bytes): the hardware returns zeros WITHOUT a memory access — what a
  // slot that is not needed loads (a clamped duplicate would be one more trip through the fabric; a conditional load a branch)
  static constexpr uint32_t kNone4 = 0x0fffffffu;
  GBP_DEV void st4(uint32_t i4, const float4 v) const {
    const v4u x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)(i4 * 16u), (int)soff, kSc1);
    if (VER) {
      if (mirror4 != 0u) {
        const v4u y = {~x.x, ~x.y, ~x.z, x.w};
        __builtin_amdgcn_raw_buffer_store_b128(y, r, (int)((i4 + mirror4) * 16u), (int)soff, kSc1);
      }
    }
  }
  // (the b32 intrinsics are typed unsigned: bit casts, not value conversions)
  GBP_DEV float ld1(uint32_t i) const { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(i * 4u), (int)soff, kSc1)); }
  GBP_DEV void st1(uint32_t i, const float v) const { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)(i * 4u), (int)soff, kSc1); }
};
using XwBuf = XwBufT<false>;
template <int G>
GBP_DEV void load_rec_xw(const XwBuf& buf, uint32_t i4, float (&out)[G * 4]) {
  GBP_UNROLL
  for (int g = 0; g < G; ++g) {
    const float4 v = buf.ld4(i4 + (uint32_t)g);
    out[4 * g] = v.x; out[4 * g + 1] = v.y; out[4 * g + 2] = v.z; out[4 * g + 3] = v.w;
  }
}
GBP_DEV float4 lmsg_piece_xw(const XwBuf& lmsg, uint32_t pos, uint32_t q) {
  float4 m = lmsg.ld4(pos * 4u + q);
  if (q == 0) m.w = 0.f;
  if (q == 3) { m.y = 0.f; m.z = 0.f; m.w = 0.f; }
  return m;
}

constexpr unsigned long long kBarrierTimeoutTicks = 150000000ull;   // 1.5 s of the 100 MHz wall clock
// The hand-off in two halves, so that work which needs nothing from the other workgroups can run between the arrival and the
// wait (the metric's residuals of the previous iteration: they then cost nothing while the arrivals are in flight).
GBP_DEV void grid_arrive(unsigned* sync) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's write-through stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
GBP_DEV void grid_wait(unsigned* sync, unsigned target /* arrivals to wait for */, unsigned* status, unsigned seq /* what a time-out writes to *status */) {
  if (threadIdx.x == 0) {
    unsigned spin = 0;
    unsigned long long t0 = 0;
    while ((int)(__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {   // wrap-safe
      // bounded wait (1.5 s by the wall clock), and once ONE workgroup has given up every other one leaves its barriers at once
      // (sync[32] is the abort word): a launch that can never complete ends in seconds with *status raised, it does not hang
      // the GPU.  The abort word stays set: later launches of the ctx return at once (k_persist prologue) until the host has
      // restored the state the failed launch started from (gbp_api_persist.cpp: persist_recover).
      if ((++spin & 255u) == 0u) {
        const unsigned long long now = wall_clock64();
        if (t0 == 0) t0 = now;
        if (now - t0 > kBarrierTimeoutTicks || __hip_atomic_load(sync + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          if (__hip_atomic_exchange(sync + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) *status = seq;   // the first to give up names the launch
          break;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}
GBP_DEV void grid_sync(unsigned* sync, unsigned target, unsigned* status, unsigned seq) {
  grid_arrive(sync);
  grid_wait(sync, target, status, seq);
}

// Snapshot / restore of the arrays a k_persist launch mutates (one launch for all of them): taken before every launch so that
// a launch whose barrier timed out can be undone and replayed on the two-kernel path.  `guard` != NULL: do nothing once the
// abort word is set — the snapshot then still holds the state the FIRST failed launch started from.
__global__ __launch_bounds__(256) void k_copy_segments(const CopySegs t, const unsigned* guard) {
  if (guard && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  for (int sgi = 0; sgi < t.n; ++sgi) {
    const float4* src = static_cast<const float4*>(t.src[sgi]);
    float4* dst = static_cast<float4*>(t.dst[sgi]);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < t.n4[sgi]; i += (size_t)gridDim.x * 256) dst[i] = src[i];
  }
}

#ifdef GBP_BUILD_TEST_HOOKS
#include "hooks/gbp_persist_barrier.hip"      // k_persist<EV>: the barrier kernel of rounds 3-4 (reference of the tests, A/B runs)
#endif

// =================================================================================================
// k_persist_flow: k_persist without device-wide barriers (PersistFlow in gbp_kernels.h).  Same roles, same per-lane arithmetic in
// the same order — bit-identical results — but every hand-off is a dependency on DATA: a consumer re-loads the tagged records it
// needs until they carry the iteration it waits for.  Why the two halves suffice: a tile wave overwrites its records of iteration
// it when it produces those of it + 2, for which it needed the beliefs of it + 1 of ALL of its variables, whose owners produced them
// after reading the tile's records of it + 1 — after those of it; and the other way round.  Launches without the metric only.
// A wait is bounded like k_persist's barriers (1.5 s, abort word, *status): the wave then returns and the host restores the
// snapshot and replays on the two-kernel path.
// =================================================================================================
GBP_DEV bool flow_give_up(unsigned spin, unsigned long long& t0, unsigned* sync, unsigned* status, unsigned seq) {
  if ((spin & 255u) != 255u) return false;
  const unsigned long long now = wall_clock64();
  if (t0 == 0) t0 = now;
  if (now - t0 > kBarrierTimeoutTicks || __hip_atomic_load(sync + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    if (__hip_atomic_exchange(sync + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) *status = seq;
    return true;
  }
  return false;
}
// `attempt()` issues a round of loads and says per lane whether everything it needs has arrived; true once every lane says so
// (a waiting wave re-loads through the fabric — sc1 loads miss the L2 by design: between two rounds it sleeps, and where a round is
// many records per lane it first polls ONE of them, `probe`, the one its producer writes last; 200 waves re-loading 30 records per
// lane flat out were 5 TB/s of polling and made fr1xyz 2 us per iteration SLOWER than the barriers)
template <class Attempt>
GBP_DEV bool flow_wait(Attempt&& attempt, unsigned* sync, unsigned* status, unsigned seq) {
  unsigned long long t0 = 0;
  for (unsigned spin = 0;; ++spin) {
    asm volatile("" ::: "memory");      // the loads are re-issued every time round
    if (__all(attempt())) return true;
    if (flow_give_up(spin, t0, sync, status, seq)) return false;
    __builtin_amdgcn_s_sleep(2);
  }
}
GBP_DEV float4 flow_rec(float x, float y, float z, unsigned tag) { return make_float4(x, y, z, __uint_as_float(tag)); }
GBP_DEV bool flow_is(const float4 v, unsigned tag) { return __float_as_uint(v.w) == tag; }
GBP_DEV float flow_pick(const float4 v, uint32_t c) { return c == 0u ? v.x : c == 1u ? v.y : v.z; }
// payload slot k (0 .. 29) of the tagged camera belief -> float of the 44-float CAMB record (eta 0..5, S, lower triangle row-wise)
GBP_DEV uint32_t flow_camb_src(uint32_t k) {
  if (k < 7u) return k;
  if (k >= 28u) return 0u;
  const uint32_t t = k - 7u;
  uint32_t i = 0;
  while ((i + 1u) * (i + 2u) / 2u <= t) ++i;
  return 8u + i * 6u + (t - i * (i + 1u) / 2u);
}

// EV: the launch carries the metric (A.ev, as in k_persist<true>): the metric means of iteration k are tagged records too (F.emc,
// F.eml), the tile waves evaluate their residuals one iteration later behind their belief-phase role, the health counters are
// per-iteration device words that block 0 hands to the host's slots behind the ONE barrier such a launch ends with.
template <bool EV, bool VER = false>
__global__ __launch_bounds__(256) void k_persist_flow(const PersistArgs A) {
  using Xw = XwBufT<VER>;
  const bool ev_on = EV && A.ev.on != 0;
  const SweepArgs& a = A.s;
  const BeliefArgs& b = A.b;
  const PersistFlow& F = A.f;
  const uint32_t wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (__hip_atomic_load(A.sync + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;      // (see k_persist)
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
  const uint32_t w = bid * 4 + wib;
  __shared__ float4 lm_stage[4][64 * 4];
  __shared__ float sh[4][48];
  float4* stage = lm_stage[wib];
  const uint32_t mir = VER ? F.mirror4 : 0u;
  unsigned long long* const verr = VER ? F.verify_errors : nullptr;
  // ONE descriptor (the shadows are one allocation, F.lmsg its start) + a scalar byte offset per array
  const auto off_of = [&](const float4* q) { return (uint32_t)(reinterpret_cast<const char*>(q) - reinterpret_cast<const char*>(F.lmsg)); };
  const Xw S_lmsg(F.lmsg, mir, verr), S_rowp(F.lmsg, mir, verr, off_of(F.rowp)), S_camb(F.lmsg, mir, verr, off_of(F.camb)),
      S_cmu(F.lmsg, mir, verr, off_of(F.cmu)), S_clin(F.lmsg, mir, verr, off_of(F.clin)), S_lmkb(F.lmsg, mir, verr, off_of(F.lmkb)),
      S_lmu(F.lmsg, mir, verr, off_of(F.lmu));
  const Xw S_emc(F.lmsg, mir, verr, off_of(F.emc)), S_eml(F.lmsg, mir, verr, off_of(F.eml));
  const uint32_t nC = b.n_cams, nL = b.n_lmks, Ep = A.n_tiles * 64u, n_rows = A.n_tiles * 4u;

  // ---- phase-A role: sweep tile w; the factor's potential and both of its messages stay in registers ----
  const bool has_tile = w < A.n_tiles;
  const uint32_t tile = has_tile ? w : 0u, p = tile * 64 + lane;
  const uint32_t rec_t = lane >> 2, swz_own = ((lane >> 2) & 3u) ^ (lane & 2u);
  const uint32_t lm_tile4 = tile * 256u;
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
      stage[r * 4 + ((lane & 3u) ^ (((r >> 2) & 3u) ^ (r & 2u)))] = a.lmsg[lm_tile4 + (uint32_t)k * 64u + lane];
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

  // ---- phase-B role: camera v, or landmarks 16 (v - C) .. + 15 (4 lanes each); numbered across the workgroups as in k_persist ----
  // (EV: waves [C + G, 2C + G), where the grid has them, take the METRIC mean of camera v - (C + G): k_persist's metric roles)
  // A.separate: the roles are dealt to the waves WITHOUT a tile first (whole workgroups behind the tiles', one role per workgroup
  // before a second one), cameras, then metric means, then landmark groups: a wave that owns a camera then sweeps no tile, and
  // nothing it does behind its publication (the metric's residuals, the fp64 solves of a metric mean) delays a sweep that every
  // camera waits for.  v below is the role in k_persist's numbering either way.
  const uint32_t v_met0 = nC + A.n_lmk_groups;
  const uint32_t v = persist_role(bid, wib, nblk, A.n_tiles, nC, A.n_lmk_groups, A.n_met, A.separate);
  const bool cam_wave = v < nC;
  const bool lmk_wave = !cam_wave && (v - nC) < A.n_lmk_groups;
  const bool met_wave = ev_on && v >= v_met0 && v - v_met0 < nC;
  const bool cam_has_met_wave = cam_wave && ev_on && (A.separate ? A.n_met != 0u : v_met0 + v < nblk * 4u);
  const uint32_t camv = met_wave ? v - v_met0 : v;
  const uint32_t cj = lane;
  const bool cam_live = (cam_wave || met_wave) && cj < (uint32_t)kCamRec;
  uint32_t r0 = 0, r1 = 0;
  float cam_prior_j = 0.f;
  float4 cam_cur0 = make_float4(0.f, 0.f, 0.f, 0.f), cam_cur1 = cam_cur0;
  // where element cj of a row's 44-float record sits in the row's tagged record (row16_sums_store: lane q of the row's last quad holds
  // the groups q, q + 4, q + 8 = twelve floats = four tagged records)
  const uint32_t rg = cj >> 2, rn = 4u * (rg >> 2) + (cj & 3u);
  const uint32_t row_f4 = 4u * (rg & 3u) + rn / 3u, row_c = rn % 3u;
  const uint32_t cb_s0 = flow_camb_src(3u * lane), cb_s1 = flow_camb_src(3u * lane + 1u), cb_s2 = flow_camb_src(3u * lane + 2u);   // lanes 0..9
  if (cam_wave || met_wave) {
    r0 = b.cam_row_ptr[camv]; r1 = b.cam_row_ptr[camv + 1];
    if (cam_live) cam_prior_j = b.cam_prior[(size_t)camv * kCamRec + cj];
  }
  if (cam_wave) { cam_cur0 = a.cam_mu[(size_t)v * 4]; cam_cur1 = a.cam_mu[(size_t)v * 4 + 1]; }
  // WEAKEN_PRIORS inside the launch (PersistArgs.w_first / w_steps2): every wave that holds a prior keeps its variable's flag and
  // scaling in registers and applies WeakenPriorVertex (gbp_codelets.cpp:176-197: a flag in 1 .. 5 scales the prior and counts down) in
  // front of the iterations the reference's loop weakens before; the owners write prior and flag back with the last iteration
  const bool weakens = A.w_steps2 != 0u;
  uint32_t wf = 0;              // the camera's / the landmark's weaken flag
  float wscale = 1.f;
  bool weakened = false;        // did this launch change this wave's prior?
  if (weakens && (cam_wave || met_wave)) { wf = b.cam_wflag[camv]; wscale = b.cam_scale[camv]; }
  const uint32_t l = lmk_wave ? (v - nC) * 16 + (lane >> 2) : 0u, q4 = lane & 3;
  const bool lmk_live = lmk_wave && l < nL;
  uint4 ix = make_uint4(0u, 0u, 0u, 0u);
  float pr3[3] = {0.f, 0.f, 0.f};
  float4 lmk_cur = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t lp0 = 0, lp1 = 0;
  const uint32_t l_e0 = q4 == 0u ? 0u : 4u + 3u * (q4 - 1u);      // first float of this lane's triple in the 16-float landmark record
  if (lmk_live) {
    ix = reinterpret_cast<const uint4*>(b.lmk_ix)[(size_t)l * 4 + q4];
    const float* pr = reinterpret_cast<const float*>(b.lmk_prior) + (size_t)l * 16 + l_e0;
    pr3[0] = pr[0]; pr3[1] = pr[1]; pr3[2] = pr[2];
    lmk_cur = a.lmk_mu[(size_t)l * 2];
    lp0 = b.lmk_ptr[l]; lp1 = b.lmk_ptr[l + 1];
    if (weakens) { wf = b.lmk_wflag[l]; wscale = b.lmk_scale[l]; }
  }
  const uint32_t deg = (uint32_t)__shfl((int)ix.x, 0, 4);
  uint32_t pos[32];
  GBP_UNROLL
  for (int k = 0; k < 15; ++k) {
    const uint32_t e = ((k + 1) & 3) == 0 ? ix.x : ((k + 1) & 3) == 1 ? ix.y : ((k + 1) & 3) == 2 ? ix.z : ix.w;
    pos[k] = (uint32_t)__shfl((int)e, (k + 1) >> 2, 4);
  }
  GBP_UNROLL
  for (int k = 0; k < 15; ++k) {
    const uint32_t p2 = b.lmk_fpos[lp0 + (15u + (uint32_t)k < deg ? 15u + (uint32_t)k : 0u)];
    pos[15 + k] = (lmk_live && 15u + (uint32_t)k < deg) ? p2 : 0u;
  }

  // ---- prologue: the beliefs this launch starts from, published as "iteration -1" (half 1, tag0) ----
  if (cam_wave) {
    const float* rec = b.camb + (size_t)v * kCamRec;
    if (lane < kFlowCam4) S_camb.st4((nC + v) * kFlowCam4 + lane, flow_rec(rec[cb_s0], rec[cb_s1], rec[cb_s2], F.tag0));
    if (lane == 0) {
      S_cmu.st4((nC + v) * 2u, flow_rec(cam_cur0.x, cam_cur0.y, cam_cur0.z, F.tag0));
      S_cmu.st4((nC + v) * 2u + 1u, flow_rec(cam_cur0.w, cam_cur1.x, cam_cur1.y, F.tag0));
      const float* cl = reinterpret_cast<const float*>(a.cam_lin + (size_t)v * kCamLin4);
      GBP_UNROLL
      for (int g = 0; g < (int)kFlowClin4; ++g)
        S_clin.st4((nC + v) * kFlowClin4 + (uint32_t)g, flow_rec(cl[3 * g], cl[3 * g + 1], g < 6 ? cl[3 * g + 2] : 0.f, F.tag0));
    }
  } else if (lmk_live) {
    const float* rec = reinterpret_cast<const float*>(a.lmkb) + (size_t)l * 16;
    S_lmkb.st4((nL + l) * kFlowLmk4 + q4, flow_rec(rec[l_e0], rec[l_e0 + 1u], rec[l_e0 + 2u], F.tag0));
    if (q4 == 0) {
      S_lmkb.st4((nL + l) * kFlowLmk4 + 4u, flow_rec(rec[3], rec[13], rec[14], F.tag0));
      S_lmu.st4(nL + l, flow_rec(lmk_cur.x, lmk_cur.y, lmk_cur.z, F.tag0));
    }
  }

  // ---- the metric (EV): per-iteration health words in device memory (zero on entry and on exit), the tile waves' partial sums in the
  // host's slots exactly as k_persist<true> leaves them ----
  auto health_of = [&](uint32_t kk) -> unsigned long long* { return A.ev.each ? F.health_iter + 2u * kk : A.ev.health; };
  // (the rotation of the camera's metric mean — eigenso3exp, a function of the camera alone — comes with the mean's record: computed once
  // by the mean's owner instead of once per factor, the same operations on the same operands, as in the two-kernel path's EvalRide)
  auto metric = [&](uint32_t kk, int packed, const float (&R)[9], const float (&t)[3], const float (&lmu)[3]) {
    double s_norm = 0, s_half = 0;
    const uint32_t flags = (uint32_t)packed & 7u;
    const bool pad = (flags & kFlagPad) != 0, active = !pad && (flags & kFlagActive) != 0;
    if (active) eval_residual(R, t, lmu, fac[54], fac[55], a.K, s_norm, s_half);
    DeviceEval* slots = A.ev.slots + (size_t)(A.ev.each ? kk : 0u) * A.ev.stride;
    eval_wave_tree(s_norm, s_half);      // the lane tree of every metric, then one record per wave
    const unsigned long long n_act = (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(active));
    const unsigned long long n_rel = (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(!pad && (packed >> 3) == -A.ev.num_undamped));
    const unsigned long long n_rob = (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(!pad && (flags & kFlagRobust) != 0));
    if (lane == 0) {
      DeviceEval o;
      o.sum_norm = s_norm; o.sum_half_sq = s_half; o.n_active = n_act; o.n_relin = n_rel; o.n_robust = n_rob; o.pad = 0;
      slots[1 + w] = o;
    }
  };
  // the tile wave's share of metric kk (means of iteration kk: half kk & 1, tag0 + kk + 1): c0 / c1 / l0 were loaded early, re-loaded here
  // only if they had not arrived then
  auto metric_of = [&](uint32_t kk, int packed, float4 (&c)[4], float4 l0) -> bool {
    const unsigned tg = F.tag0 + kk + 1u;
    const uint32_t ec = ((kk & 1u) * nC + cam_i) * 4u, el = (kk & 1u) * nL + lmk_i;
    if (!__all(flow_is(c[0], tg) && flow_is(c[1], tg) && flow_is(c[2], tg) && flow_is(c[3], tg) && flow_is(l0, tg))) {
      if (!flow_wait([&]() {
            GBP_UNROLL
            for (int g = 0; g < 4; ++g) c[g] = S_emc.ld4(ec + (uint32_t)g);
            l0 = S_eml.ld4(el);
            return flow_is(c[0], tg) && flow_is(c[1], tg) && flow_is(c[2], tg) && flow_is(c[3], tg) && flow_is(l0, tg);
          }, A.sync, A.status, A.seq)) return false;
    }
    const float R[9] = {c[0].x, c[0].y, c[0].z, c[1].x, c[1].y, c[1].z, c[2].x, c[2].y, c[2].z};
    const float t[3] = {c[3].x, c[3].y, c[3].z}, lmu[3] = {l0.x, l0.y, l0.z};
    metric(kk, packed, R, t, lmu);
    return true;
  };
  auto publish_metric_mean = [&](uint32_t half, uint32_t cam, const float (&xm)[6], unsigned tag) {
    float R[9];
    eval_cam_rot(xm, R);
    GBP_UNROLL
    for (int g = 0; g < 3; ++g) S_emc.st4((half * nC + cam) * 4u + (uint32_t)g, flow_rec(R[3 * g], R[3 * g + 1], R[3 * g + 2], tag));
    S_emc.st4((half * nC + cam) * 4u + 3u, flow_rec(xm[0], xm[1], xm[2], tag));
  };

  for (int it = 0; it < A.n_iters; ++it) {
    const bool ev_means = ev_on && (A.ev.each || it + 1 == A.n_iters);     // this iteration's beliefs are evaluated
    const bool ev_prev = ev_on && A.ev.each && it > 0 && has_tile;          // this tile wave owes the residuals of iteration it - 1
    const int ev_packed = __float_as_int(lm[13]);      // the factor's state word as the sweep of iteration it - 1 left it
    const uint32_t h_in = ((uint32_t)it + 1u) & 1u, h_out = (uint32_t)it & 1u;
    const unsigned t_in = F.tag0 + (unsigned)it, t_out = F.tag0 + (unsigned)it + 1u;
    const bool last = it + 1 == A.n_iters;
    // WeakenPriorVertex in front of the NEXT iteration: the owners then publish the beliefs of the weakened priors (what WEAKEN_PRIORS'
    // belief refresh leaves: same messages, same order, the means against the one the last sweep used), the metric keeps this iteration's
    const bool weak_next = weakens && !last && (((A.w_first + (uint32_t)it) & 1u) == 0u) && (A.w_first + (uint32_t)it + 1u < A.w_steps2);
    // ================= phase A: the sweep of this wave's tile =================
    if (has_tile) {
      float4 c4[kFlowCam4], m4[2], q7[kFlowClin4], l5[kFlowLmk4], u4;
      const uint32_t cb0 = (h_in * nC + cam_i), lb0 = (h_in * nL + lmk_i);
      // probe: the records the two owners store last (the camera's belief behind its mean and CAM_LIN, the landmark's squared mean changes)
      if (!flow_wait([&]() { return flow_is(S_camb.ld4(cb0 * kFlowCam4), t_in) && flow_is(S_lmkb.ld4(lb0 * kFlowLmk4 + 4u), t_in); },
                     A.sync, A.status, A.seq)) return;
      if (!flow_wait([&]() {
            bool ok = true;
            GBP_UNROLL
            for (int g = 0; g < (int)kFlowCam4; ++g) c4[g] = S_camb.ld4(cb0 * kFlowCam4 + (uint32_t)g);
            m4[0] = S_cmu.ld4(cb0 * 2u); m4[1] = S_cmu.ld4(cb0 * 2u + 1u);
            GBP_UNROLL
            for (int g = 0; g < (int)kFlowClin4; ++g) q7[g] = S_clin.ld4(cb0 * kFlowClin4 + (uint32_t)g);
            GBP_UNROLL
            for (int g = 0; g < (int)kFlowLmk4; ++g) l5[g] = S_lmkb.ld4(lb0 * kFlowLmk4 + (uint32_t)g);
            u4 = S_lmu.ld4(lb0);
            GBP_UNROLL
            for (int g = 0; g < (int)kFlowCam4; ++g) ok = ok && flow_is(c4[g], t_in);
            ok = ok && flow_is(m4[0], t_in) && flow_is(m4[1], t_in);
            GBP_UNROLL
            for (int g = 0; g < (int)kFlowClin4; ++g) ok = ok && flow_is(q7[g], t_in);
            GBP_UNROLL
            for (int g = 0; g < (int)kFlowLmk4; ++g) ok = ok && flow_is(l5[g], t_in);
            return ok && flow_is(u4, t_in);
          }, A.sync, A.status, A.seq)) return;
      float cb[44], lb[16], mu[12];
      {
        float pl[30];
        GBP_UNROLL
        for (int g = 0; g < (int)kFlowCam4; ++g) { pl[3 * g] = c4[g].x; pl[3 * g + 1] = c4[g].y; pl[3 * g + 2] = c4[g].z; }
        GBP_UNROLL
        for (int i = 0; i < 44; ++i) cb[i] = 0.f;
        GBP_UNROLL
        for (int i = 0; i < 7; ++i) cb[i] = pl[i];
        GBP_UNROLL
        for (int i = 0; i < 6; ++i) {
          GBP_UNROLL
          for (int j = 0; j <= i; ++j) cb[8 + i * 6 + j] = pl[7 + tri(i, j)];
        }
      }
      lb[0] = l5[0].x; lb[1] = l5[0].y; lb[2] = l5[0].z;
      GBP_UNROLL
      for (int g = 0; g < 3; ++g) { lb[4 + 3 * g] = l5[1 + g].x; lb[5 + 3 * g] = l5[1 + g].y; lb[6 + 3 * g] = l5[1 + g].z; }
      lb[3] = l5[4].x; lb[13] = l5[4].y; lb[14] = l5[4].z; lb[15] = 0.f;
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
                               x0c[0] = m4[0].x; x0c[1] = m4[0].y; x0c[2] = m4[0].z; x0c[3] = m4[1].x; x0c[4] = m4[1].y; x0c[5] = m4[1].z;
                               x0l[0] = u4.x; x0l[1] = u4.y; x0l[2] = u4.z;
                               float f[21];
                               GBP_UNROLL
                               for (int g = 0; g < (int)kFlowClin4; ++g) { f[3 * g] = q7[g].x; f[3 * g + 1] = q7[g].y; f[3 * g + 2] = q7[g].z; }
                               float4 clq[kCamLin4];
                               GBP_UNROLL
                               for (int g = 0; g < kCamLin4; ++g) clq[g] = make_float4(f[4 * g], f[4 * g + 1], f[4 * g + 2], f[4 * g + 3]);
                               cam_lin_unpack(clq, cl);
                             });
      fac_dirty = fac_dirty || (active && relin);
      ol[3] = damping;
      ol[13] = __int_as_float((int)(((uint32_t)count << 3) | flags));
      ol[14] = var;
      // the tagged landmark messages: eta | Lambda row 0 | row 1 | row 2, through the LDS transpose of k_sweep (coalesced stores)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      stage[lane * 4 + (0u ^ swz_own)] = flow_rec(ol[0], ol[1], ol[2], t_out);
      GBP_UNROLL
      for (int q = 1; q < 4; ++q) stage[lane * 4 + ((uint32_t)q ^ swz_own)] = flow_rec(ol[1 + 3 * q], ol[2 + 3 * q], ol[3 + 3 * q], t_out);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      GBP_UNROLL
      for (int k = 0; k < 4; ++k) {
        const uint32_t r = k * 16 + rec_t;
        S_lmsg.st4(h_out * Ep * 4u + lm_tile4 + (uint32_t)k * 64u + lane, stage[r * 4 + ((lane & 3u) ^ (((r >> 2) & 3u) ^ (r & 2u)))]);
      }
      {  // camera half of the belief reduction: per-row tree sums (row16_sums_store), the holder's twelve floats as four tagged records
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        row16_sums_store(oc_eta, oc_lam, lane, [&](uint32_t g, float4 x) {
          const uint32_t j = g >> 2;
          if (j == 0u) g0 = x; else if (j == 1u) g1 = x; else g2 = x;
          if (last) a.rowp[(size_t)(p >> 4) * kCamRec4 + g] = x;
        });
        if ((lane & 12u) == 12u) {
          const uint32_t q = lane & 3u, rp = (h_out * n_rows + (p >> 4)) * kFlowRow4 + 4u * q;
          S_rowp.st4(rp, flow_rec(g0.x, g0.y, g0.z, t_out));
          S_rowp.st4(rp + 1u, flow_rec(g0.w, g1.x, g1.y, t_out));
          S_rowp.st4(rp + 2u, flow_rec(g1.z, g1.w, g2.x, t_out));
          if (q < 3u) S_rowp.st4(rp + 3u, flow_rec(g2.y, g2.z, g2.w, t_out));
        }
      }
      if (last) {   // the ordinary LMSG tile (with the factor's scalars in its pad slots), as k_sweep leaves it
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
          a.lmsg[lm_tile4 + (uint32_t)k * 64u + lane] = stage[r * 4 + ((lane & 3u) ^ (((r >> 2) & 3u) ^ (r & 2u)))];
        }
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
    }

    // ================= phase B: the belief update (arithmetic of k_beliefs, roll = 1) =================
    float4 ev_c[4], ev_l0 = make_float4(0.f, 0.f, 0.f, 0.f);
    GBP_UNROLL
    for (int g = 0; g < 4; ++g) ev_c[g] = ev_l0;
    if (EV && ev_prev) {      // in flight with the role's own loads (measured: the residuals at the end of the phase, not behind the sweep)
      const uint32_t kk = (uint32_t)it - 1u;
      GBP_UNROLL
      for (int g = 0; g < 4; ++g) ev_c[g] = S_emc.ld4(((kk & 1u) * nC + cam_i) * 4u + (uint32_t)g);
      ev_l0 = S_eml.ld4((kk & 1u) * nL + lmk_i);
    }
    unsigned long long* const hw = EV ? health_of((uint32_t)it) : nullptr;
    if (cam_wave || (EV && met_wave && ev_means)) {
      float acc = 0.f;
      if (r1 > r0) {
        const uint32_t row4 = (h_out * n_rows + r0) * kFlowRow4 + row_f4;          // float4 index of this lane's element in row r0
        const uint32_t n = r1 - r0;
        // rows per round; rows of the first round.  (measured on the three sequences, round 5: 1 + 16 is as fast as or faster than 2 + 16,
        // 17 + 16, 9 + 8, 1 + 32 and everything in one round — k_persist's shape, for the same reason: registers)
        // (... and once the cameras had waves of their own, measured again: the instantiation with the metric is faster with two rounds for
        // up to 28 rows, 12 + 16 — fr1xyz 12.28 -> 11.94 us per iteration, fr1desk 12.47 -> 12.29, fr2robot2 11.45 -> 11.3; 4 + 24, 8 + 20,
        // 14 + 14 about the same — the plain one stays fastest with 1 + 16: 11.52 against 11.8 - 12.0)
        constexpr int RB = 16, RF = EV ? 12 : 1;
        uint32_t r = 0;
        {  // first round: rows 0 .. RF - 1 (unconditional; rows the camera does not have: out of range, no access)
          float4 x[RF];
          if (!flow_wait([&]() {
                bool ok = true;
                GBP_UNROLL
                for (int k = 0; k < RF; ++k) x[k] = S_rowp.ld4((uint32_t)k < n ? row4 + (uint32_t)k * kFlowRow4 : XwBuf::kNone4);
                GBP_UNROLL
                for (int k = 0; k < RF; ++k) ok = ok && ((uint32_t)k >= n || flow_is(x[k], t_out));
                return !cam_live || ok;
              }, A.sync, A.status, A.seq)) return;
          acc = flow_pick(x[0], row_c);
          GBP_UNROLL
          for (int k = 1; k < RF; ++k)
            if ((uint32_t)k < n) acc = acc + flow_pick(x[k], row_c);
          r = n < (uint32_t)RF ? n : (uint32_t)RF;
        }
        for (; r + RB <= n; r += RB) {
          float4 x[RB];
          if (!flow_wait([&]() {
                bool ok = true;
                GBP_UNROLL
                for (int k = 0; k < RB; ++k) x[k] = S_rowp.ld4(row4 + (r + (uint32_t)k) * kFlowRow4);
                GBP_UNROLL
                for (int k = 0; k < RB; ++k) ok = ok && flow_is(x[k], t_out);
                return !cam_live || ok;
              }, A.sync, A.status, A.seq)) return;
          GBP_UNROLL
          for (int k = 0; k < RB; ++k) acc = acc + flow_pick(x[k], row_c);
        }
        if (r < n) {  // tail (< RB rows): unconditional loads, the rows beyond the camera's out of range
          float4 x[RB];
          const uint32_t m = n - r;
          if (!flow_wait([&]() {
                bool ok = true;
                GBP_UNROLL
                for (int k = 0; k < RB; ++k) x[k] = S_rowp.ld4((uint32_t)k < m ? row4 + (r + (uint32_t)k) * kFlowRow4 : XwBuf::kNone4);
                GBP_UNROLL
                for (int k = 0; k < RB; ++k) ok = ok && ((uint32_t)k >= m || flow_is(x[k], t_out));
                return !cam_live || ok;
              }, A.sync, A.status, A.seq)) return;
          GBP_UNROLL
          for (int k = 0; k < RB; ++k)
            if ((uint32_t)k < m) acc = acc + flow_pick(x[k], row_c);
        }
      }
      if (cam_live) {
        if (last && cam_wave) b.cam_local[(size_t)v * kCamRec + cj] = acc;
        sh[wib][cj] = cam_prior_j + acc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (EV && lane == 0 && !cam_wave) {   // metric role: what k_means computes for this camera, from the belief in LDS; then — published
        float xm[6];                        // first — the health check the camera wave would otherwise run on its critical lane
        solve_pivot<6>(sh[wib] + 8, 6, sh[wib], xm);
        publish_metric_mean(h_out, camv, xm, t_out);
        bool finite = true;
        GBP_UNROLL
        for (int i = 0; i < 6; ++i) finite &= (xm[i] - xm[i] == 0.f);
        if (!finite) atomicAdd(&hw[0], 1ull);
        if (!ldl_pivots_positive<6>(sh[wib] + 8, 6)) atomicAdd(&hw[1], 1ull);
      }
      float xm[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      bool pd = true;
      const bool ev_here = EV && ev_means && cam_wave && !cam_has_met_wave;     // no wave to spare for this camera's metric mean: solved here
      if (lane == 0 && ev_here) {      // ... from this iteration's belief, before a weakening replaces it below
        solve_pivot<6>(sh[wib] + 8, 6, sh[wib], xm);
        pd = ldl_pivots_positive<6>(sh[wib] + 8, 6);
      }
      if (weak_next) {      // WeakenPriorVertex on this camera's prior (every wave that holds a copy, the metric waves too)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (wf >= 1u && wf <= 5u) {
          if (cam_live) cam_prior_j *= wscale;
          wf -= 1u;
          weakened = true;
        }
        if (cam_live && cam_wave) sh[wib][cj] = cam_prior_j + acc;      // the belief the next sweep consumes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (lane == 0 && cam_wave) {
        float cb[44], x0c[6];
        GBP_UNROLL
        for (int i = 0; i < 44; ++i) cb[i] = sh[wib][i];
        cam_mean(cb, x0c);
        const float used[6] = {cam_cur0.x, cam_cur0.y, cam_cur0.z, cam_cur0.w, cam_cur1.x, cam_cur1.y};
        float S = 0.f;
        GBP_UNROLL
        for (int i = 0; i < 6; ++i) S += (used[i] - x0c[i]) * (used[i] - x0c[i]);
        if (last) { b.cam_mu[(size_t)v * 4 + 2] = cam_cur0; b.cam_mu[(size_t)v * 4 + 3] = cam_cur1; }
        cam_cur0 = make_float4(x0c[0], x0c[1], x0c[2], x0c[3]);
        cam_cur1 = make_float4(x0c[4], x0c[5], 0.f, 0.f);
        S_cmu.st4((h_out * nC + v) * 2u, flow_rec(x0c[0], x0c[1], x0c[2], t_out));
        S_cmu.st4((h_out * nC + v) * 2u + 1u, flow_rec(x0c[3], x0c[4], x0c[5], t_out));
        if (last) { b.cam_mu[(size_t)v * 4] = cam_cur0; b.cam_mu[(size_t)v * 4 + 1] = cam_cur1; }
        {
          CamLin cl;
          const float wv[3] = {x0c[3], x0c[4], x0c[5]};
          cam_lin(wv, cl);
          float4 q[kCamLin4];
          cam_lin_pack(cl, q);
          float f[21];
          GBP_UNROLL
          for (int g = 0; g < kCamLin4; ++g) { f[4 * g] = q[g].x; f[4 * g + 1] = q[g].y; f[4 * g + 2] = q[g].z; f[4 * g + 3] = q[g].w; }
          f[20] = 0.f;
          GBP_UNROLL
          for (int g = 0; g < (int)kFlowClin4; ++g)
            S_clin.st4((h_out * nC + v) * kFlowClin4 + (uint32_t)g, flow_rec(f[3 * g], f[3 * g + 1], f[3 * g + 2], t_out));
          if (last) {
            GBP_UNROLL
            for (int g = 0; g < kCamLin4; ++g) b.cam_lin[(size_t)v * kCamLin4 + g] = q[g];
          }
        }
        if (ev_here) {
          publish_metric_mean(h_out, v, xm, t_out);
          bool finite = true;
          GBP_UNROLL
          for (int i = 0; i < 6; ++i) finite &= (xm[i] - xm[i] == 0.f);
          if (!finite) atomicAdd(&hw[0], 1ull);
          if (!pd) atomicAdd(&hw[1], 1ull);
        }
        sh[wib][6] = S;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (cam_wave && lane < kFlowCam4) S_camb.st4((h_out * nC + v) * kFlowCam4 + lane, flow_rec(sh[wib][cb_s0], sh[wib][cb_s1], sh[wib][cb_s2], t_out));
      if (last && cam_live && cam_wave) b.camb[(size_t)v * kCamRec + cj] = sh[wib][cj];
      if (last && weakened && cam_wave) {
        if (cam_live) b.cam_prior_rw[(size_t)v * kCamRec + cj] = cam_prior_j;
        if (lane == 0) b.cam_wflag[v] = wf;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // sh[] is rewritten by the next iteration
      __builtin_amdgcn_wave_barrier();
    } else if (lmk_wave) {
      float a3[3] = {pr3[0], pr3[1], pr3[2]};
      const bool lw = weak_next && wf >= 1u && wf <= 5u;      // this landmark's prior is weakened in front of the next iteration
      if (lw) { pr3[0] *= wscale; pr3[1] *= wscale; pr3[2] *= wscale; }
      if (weak_next && wf >= 1u && wf <= 5u) { wf -= 1u; weakened = true; }
      float a3w[3] = {pr3[0], pr3[1], pr3[2]};                 // the same sums from the weakened prior (== a3 without a weakening)
      {
        float4 m[15], m2[15];
        const bool second = __any(deg > 15u);
        const uint32_t base = h_out * Ep;
        if (!flow_wait([&]() {
              bool ok = true;
              GBP_UNROLL
              for (int k = 0; k < 15; ++k) m[k] = S_lmsg.ld4((uint32_t)k < deg ? (base + pos[k]) * 4u + q4 : XwBuf::kNone4);
              if (second) {
                GBP_UNROLL
                for (int k = 0; k < 15; ++k) m2[k] = S_lmsg.ld4(15u + (uint32_t)k < deg ? (base + pos[15 + k]) * 4u + q4 : XwBuf::kNone4);
              }
              GBP_UNROLL
              for (int k = 0; k < 15; ++k) ok = ok && ((uint32_t)k >= deg || flow_is(m[k], t_out));
              if (second) {
                GBP_UNROLL
                for (int k = 0; k < 15; ++k) ok = ok && (15u + (uint32_t)k >= deg || flow_is(m2[k], t_out));
              }
              return !lmk_live || ok;
            }, A.sync, A.status, A.seq)) return;
        GBP_UNROLL
        for (int k = 0; k < 15; ++k)     // adds in slot order
          if ((uint32_t)k < deg) { a3[0] = a3[0] + m[k].x; a3[1] = a3[1] + m[k].y; a3[2] = a3[2] + m[k].z; }
        if (second) {
          GBP_UNROLL
          for (int k = 0; k < 15; ++k)
            if (15u + (uint32_t)k < deg) { a3[0] = a3[0] + m2[k].x; a3[1] = a3[1] + m2[k].y; a3[2] = a3[2] + m2[k].z; }
        }
        if (weak_next) {      // (wave-uniform)
          GBP_UNROLL
          for (int k = 0; k < 15; ++k)
            if ((uint32_t)k < deg) { a3w[0] = a3w[0] + m[k].x; a3w[1] = a3w[1] + m[k].y; a3w[2] = a3w[2] + m[k].z; }
          if (second) {
            GBP_UNROLL
            for (int k = 0; k < 15; ++k)
              if (15u + (uint32_t)k < deg) { a3w[0] = a3w[0] + m2[k].x; a3w[1] = a3w[1] + m2[k].y; a3w[2] = a3w[2] + m2[k].z; }
          }
        }
      }
      if (deg > 30u) {
        for (uint32_t s = lp0 + 30u; s < lp1; s += 8) {
          uint32_t ps[8];
          float4 m[8];
          const uint32_t nleft = lp1 - s;
          GBP_UNROLL
          for (int k = 0; k < 8; ++k) ps[k] = b.lmk_fpos[(uint32_t)k < nleft ? s + k : lp1 - 1u];     // clamped, unconditional
          if (!flow_wait([&]() {
                bool ok = true;
                GBP_UNROLL
                for (int k = 0; k < 8; ++k) m[k] = S_lmsg.ld4((h_out * Ep + ps[k]) * 4u + q4);
                GBP_UNROLL
                for (int k = 0; k < 8; ++k) ok = ok && flow_is(m[k], t_out);
                return ok;
              }, A.sync, A.status, A.seq)) return;
          GBP_UNROLL
          for (int k = 0; k < 8; ++k)
            if ((uint32_t)k < nleft) {
              a3[0] = a3[0] + m[k].x; a3[1] = a3[1] + m[k].y; a3[2] = a3[2] + m[k].z;
              a3w[0] = a3w[0] + m[k].x; a3w[1] = a3w[1] + m[k].y; a3w[2] = a3w[2] + m[k].z;
            }
        }
      }
      if (!weak_next) { a3w[0] = a3[0]; a3w[1] = a3[1]; a3w[2] = a3[2]; }      // (the tail loop above adds to both)
      // a3: this iteration's belief (the metric's); a3w: what the next sweep consumes
      if (lmk_live) S_lmkb.st4((h_out * nL + l) * kFlowLmk4 + q4, flow_rec(a3w[0], a3w[1], a3w[2], t_out));
      float rec[16];
      rec[0] = __shfl(a3w[0], 0, 4); rec[1] = __shfl(a3w[1], 0, 4); rec[2] = __shfl(a3w[2], 0, 4);
      GBP_UNROLL
      for (int g = 0; g < 3; ++g) {
        rec[4 + 3 * g] = __shfl(a3w[0], 1 + g, 4); rec[5 + 3 * g] = __shfl(a3w[1], 1 + g, 4); rec[6 + 3 * g] = __shfl(a3w[2], 1 + g, 4);
      }
      float recm[16];      // the record the metric mean is solved from
      GBP_UNROLL
      for (int g = 0; g < 16; ++g) recm[g] = rec[g];
      if (EV && weak_next) {
        recm[0] = __shfl(a3[0], 0, 4); recm[1] = __shfl(a3[1], 0, 4); recm[2] = __shfl(a3[2], 0, 4);
        GBP_UNROLL
        for (int g = 0; g < 3; ++g) {
          recm[4 + 3 * g] = __shfl(a3[0], 1 + g, 4); recm[5 + 3 * g] = __shfl(a3[1], 1 + g, 4); recm[6 + 3 * g] = __shfl(a3[2], 1 + g, 4);
        }
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
        const float4 used = lmk_cur;
        u[0] = (used.x - x0l[0]) * (used.x - x0l[0]);
        u[1] = (used.y - x0l[1]) * (used.y - x0l[1]);
        u[2] = (used.z - x0l[2]) * (used.z - x0l[2]);
        lmk_cur = make_float4(x0l[0], x0l[1], x0l[2], 0.f);
        S_lmu.st4(h_out * nL + l, flow_rec(x0l[0], x0l[1], x0l[2], t_out));
        S_lmkb.st4((h_out * nL + l) * kFlowLmk4 + 4u, flow_rec(u[0], u[1], u[2], t_out));
        if (last) { b.lmk_mu[(size_t)l * 2 + 1] = used; b.lmk_mu[(size_t)l * 2] = lmk_cur; }
        if (EV && ev_means) {   // metric mean of this landmark (k_means), from the belief record in registers
          float x[3];
          solve_pivot<3>(recm + 4, 3, recm, x);
          S_eml.st4(h_out * nL + l, flow_rec(x[0], x[1], x[2], t_out));
          bool finite = true;
          GBP_UNROLL
          for (int i = 0; i < 3; ++i) finite &= (x[i] - x[i] == 0.f);
          if (!finite) atomicAdd(&hw[0], 1ull);
          if (!ldl_pivots_positive<3>(recm + 4, 3)) atomicAdd(&hw[1], 1ull);
        }
      }
      if (last) {   // the ordinary LMKB record: float4 #q4 of [eta, u0 | Lambda 0..3 | 4..7 | 8, u1, u2, 0]
        rec[3] = __shfl(u[0], 0, 4); rec[13] = __shfl(u[1], 0, 4); rec[14] = __shfl(u[2], 0, 4); rec[15] = 0.f;
        float4 o;
        o.x = q4 == 0u ? rec[0] : q4 == 1u ? rec[4] : q4 == 2u ? rec[8] : rec[12];
        o.y = q4 == 0u ? rec[1] : q4 == 1u ? rec[5] : q4 == 2u ? rec[9] : rec[13];
        o.z = q4 == 0u ? rec[2] : q4 == 1u ? rec[6] : q4 == 2u ? rec[10] : rec[14];
        o.w = q4 == 0u ? rec[3] : q4 == 1u ? rec[7] : q4 == 2u ? rec[11] : rec[15];
        if (lmk_live) b.lmkb[(size_t)l * 4 + q4] = o;
        if (lmk_live && weakened) {
          float* pw = reinterpret_cast<float*>(b.lmk_prior_rw) + (size_t)l * 16 + l_e0;
          pw[0] = pr3[0]; pw[1] = pr3[1]; pw[2] = pr3[2];
          if (q4 == 0) b.lmk_wflag[l] = wf;
        }
      }
    }
    // the residuals of the PREVIOUS iteration, behind this wave's role (they delay nobody but this wave's next sweep)
    if (EV && ev_prev)
      if (!metric_of((uint32_t)it - 1u, ev_packed, ev_c, ev_l0)) return;
  }

  // ---- what stayed in registers goes back to its arrays ----
  if (has_tile) {
    store_tile<kCmsgG, false>(a.cmsg, tile, lane, cm);
    if (fac_dirty) store_tile<kFacG, false>(a.fac, tile, lane, fac);
  }

  // ---- the metric of the last iteration; then the launch's ONE barrier: behind it every owner has counted and block 0 hands the
  // health words of every metric of the launch to the host's slots (and leaves them zero) ----
  if (EV && ev_on) {
    if (has_tile) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);      // (tag 0 is nobody's: the records are loaded in metric_of)
      float4 zc[4] = {z, z, z, z};
      if (!metric_of((uint32_t)A.n_iters - 1u, __float_as_int(lm[13]), zc, z)) return;
    }
    grid_sync(A.sync, A.epoch_base + nblk, A.status, A.seq);
    if (bid == 0 && (threadIdx.x == 0 || A.ev.each)) {
      if (A.ev.each) {
        for (int kk = (int)threadIdx.x; kk < A.n_iters; kk += 256) {
          unsigned long long* out = reinterpret_cast<unsigned long long*>(A.ev.slots + (size_t)kk * A.ev.stride);
          unsigned long long* h = F.health_iter + 2 * kk;
          out[0] = __hip_atomic_load(&h[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          out[1] = __hip_atomic_load(&h[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&h[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&h[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(A.ev.slots);
        out[0] = __hip_atomic_load(&A.ev.health[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out[1] = __hip_atomic_load(&A.ev.health[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&A.ev.health[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (both areas zero behind every metric: see k_eval)
        __hip_atomic_store(&A.ev.health[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        A.ev.health_next[0] = 0ull; A.ev.health_next[1] = 0ull;
      }
    }
  }
}

// Per-factor scalar state (damping, damping_count, flags: pad slots 3 / 13 of the LMSG record) <-> compact
// per-position arrays, so that READ_PROG's damping / damping_count / robust_flag streams (ba.cpp:912-914) and
// NEW_KEYFRAME's damping_count / active_flag streams (slam.cpp:920,926) move 8 bytes per factor over PCIe instead of the
// whole 64-byte message record.
__global__ __launch_bounds__(256) void k_state_get(const float4* __restrict__ lmsg, float* __restrict__ damping,
                                                   int* __restrict__ packed, uint32_t n) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  damping[p] = lmsg[(size_t)p * 4].w;
  packed[p] = __float_as_int(lmsg[(size_t)p * 4 + 3].y);
}
// ctl bit 0: damping_count := new_count;  bit 1: active flag := bit 2
__global__ __launch_bounds__(256) void k_state_set(float4* __restrict__ lmsg, const int* __restrict__ new_count,
                                                   const uint32_t* __restrict__ ctl, uint32_t n) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const uint32_t c = ctl[p];
  if (!(c & 3u)) return;
  float4 st = lmsg[(size_t)p * 4 + 3];
  const int packed = __float_as_int(st.y);
  int count = packed >> 3;
  uint32_t flags = (uint32_t)packed & 7u;
  if (flags & kFlagPad) return;
  if (c & 1u) count = new_count[p];
  if (c & 2u) flags = (c & 4u) ? (flags | kFlagActive) : (flags & ~kFlagActive);
  st.y = __int_as_float((int)(((uint32_t)count << 3) | flags));
  lmsg[(size_t)p * 4 + 3] = st;
}

// gbp_upload: 20 bytes per position cross PCIe (read here straight out of the pinned staging buffer when they fit it) instead of the 288 bytes
// of the records they go into
__global__ __launch_bounds__(256) void k_upload_scatter(float4* __restrict__ lmsg, float4* __restrict__ fac, const float4* __restrict__ st,
                                                        const float* __restrict__ var, uint32_t n) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const float4 s = st[p];
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  lmsg[(size_t)p * 4] = make_float4(0.f, 0.f, 0.f, s.x);
  lmsg[(size_t)p * 4 + 1] = z;
  lmsg[(size_t)p * 4 + 2] = z;
  lmsg[(size_t)p * 4 + 3] = make_float4(0.f, s.y, var[p], 0.f);
  fac[((size_t)(p >> 6) * kFacG + 13) * 64 + (p & 63)] = make_float4(0.f, 0.f, s.z, s.w);      // floats 52..55: z = .z, .w
}

__global__ __launch_bounds__(256) void k_means(const float* __restrict__ camb, const float* __restrict__ lmkb,
                                               float* __restrict__ cam_mu, float* __restrict__ lmk_mu, uint32_t n_cams,
                                               uint32_t n_lmks, unsigned long long* health, unsigned long long* health_next,
                                               int count_cams) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t == 0) { health_next[0] = 0ull; health_next[1] = 0ull; }   // the NEXT evaluation's counters (double-buffered: no memset launch)
  if (t < n_cams) {
    float x[6];
    solve_pivot<6>(camb + (size_t)t * kCamRec + 8, 6, camb + (size_t)t * kCamRec, x);
    bool finite = true;
    GBP_UNROLL
    for (int i = 0; i < 6; ++i) { cam_mu[(size_t)t * 6 + i] = x[i]; finite &= (x[i] - x[i] == 0.f); }
    if (count_cams) {
      if (!finite) atomicAdd(&health[0], 1ull);
      if (!ldl_pivots_positive<6>(camb + (size_t)t * kCamRec + 8, 6)) atomicAdd(&health[1], 1ull);
    }
  } else if (t - n_cams < n_lmks) {
    const uint32_t l = t - n_cams;
    float x[3];
    solve_pivot<3>(lmkb + (size_t)l * 16 + 4, 3, lmkb + (size_t)l * 16, x);
    bool finite = true;
    GBP_UNROLL
    for (int i = 0; i < 3; ++i) { lmk_mu[(size_t)l * 3 + i] = x[i]; finite &= (x[i] - x[i] == 0.f); }
    if (!finite) atomicAdd(&health[0], 1ull);
    if (!ldl_pivots_positive<3>(lmkb + (size_t)l * 16 + 4, 3)) atomicAdd(&health[1], 1ull);
  }
}

constexpr uint32_t kEvalBlocks = 1024;
uint32_t eval_blocks(uint32_t n_tiles) {
  const uint32_t want = (n_tiles + 3) / 4;
  return want < kEvalBlocks ? (want ? want : 1) : kEvalBlocks;
}

__global__ __launch_bounds__(256) void k_eval(const uint32_t* __restrict__ row_cam, const uint32_t* __restrict__ lmk_idx,
                                              const float4* __restrict__ lmsg, const float4* __restrict__ fac, const float* __restrict__ cam_mu,
                                              const float* __restrict__ lmk_mu, const float* __restrict__ Kd,
                                              int num_undamped, DeviceEval* partials, unsigned long long* health,
                                              unsigned long long* health_out, uint32_t n_tiles) {
  // k_means has finished (stream order).  The words go back to zero: every user of the health areas finds them zero and leaves them so — the
  // launches of k_persist_flow that carry a metric per iteration count into BOTH areas (words 2k, 2k + 1 for iteration k) without a reset.
  if (blockIdx.x == 0 && threadIdx.x == 0) { health_out[0] = health[0]; health_out[1] = health[1]; health[0] = 0ull; health[1] = 0ull; }
  // block j produces S_j (see "THE ORDER OF THE METRIC'S SUMS"): one 256-factor block of the sweep at a time, wave w = tile 4b + w
  __shared__ double sh_d[2][4];
  __shared__ unsigned sh_u[3][4];
  const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  DeviceEval o;
  o.sum_norm = 0; o.sum_half_sq = 0; o.n_active = 0; o.n_relin = 0; o.n_robust = 0; o.pad = 0;
  for (uint32_t b = blockIdx.x; b < n_tiles / 4; b += gridDim.x) {
    const uint32_t tile = b * 4 + w, p = tile * 64 + lane;
    const int packed = __float_as_int(lmsg[(size_t)p * 4 + 3].y);
    const uint32_t flags = (uint32_t)packed & 7u;
    const bool pad = (flags & kFlagPad) != 0, active = !pad && (flags & kFlagActive) != 0;
    double s_norm = 0, s_half = 0;
    if (active) {
      const uint32_t cam_i = row_cam[p >> 4], lmk_i = lmk_idx[p];
      const float4 zg = fac[((size_t)tile * kFacG + 13) * 64 + lane];  // floats 52..55: z = .z, .w
      float cm[6], lmu[3];
      for (int i = 0; i < 6; ++i) cm[i] = cam_mu[(size_t)cam_i * 6 + i];
      for (int i = 0; i < 3; ++i) lmu[i] = lmk_mu[(size_t)lmk_i * 3 + i];
      eval_factor(cm, lmu, zg.z, zg.w, Kd, s_norm, s_half);
    }
    eval_wave_tree(s_norm, s_half);
    const unsigned n_act = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(active));
    const unsigned n_rel = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(!pad && (packed >> 3) == -num_undamped));
    const unsigned n_rob = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(!pad && (flags & kFlagRobust) != 0));
    if (lane == 0) { sh_d[0][w] = s_norm; sh_d[1][w] = s_half; sh_u[0][w] = n_act; sh_u[1][w] = n_rel; sh_u[2][w] = n_rob; }
    __syncthreads();
    if (threadIdx.x == 0) {
      o.sum_norm += ((sh_d[0][0] + sh_d[0][1]) + sh_d[0][2]) + sh_d[0][3];
      o.sum_half_sq += ((sh_d[1][0] + sh_d[1][1]) + sh_d[1][2]) + sh_d[1][3];
      o.n_active += sh_u[0][0] + sh_u[0][1] + sh_u[0][2] + sh_u[0][3];
      o.n_relin += sh_u[1][0] + sh_u[1][1] + sh_u[1][2] + sh_u[1][3];
      o.n_robust += sh_u[2][0] + sh_u[2][1] + sh_u[2][2] + sh_u[2][3];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = o;
}

// The riding metric of a piece's LAST iteration, which no sweep follows: the same per-tile records from the same metric records
// (one wave per tile, ride_metric), into slot counter - 1.
__global__ __launch_bounds__(256) void k_eval_ride(const EvalRide ev, const uint32_t* __restrict__ row_cam, const uint32_t* __restrict__ lmk_idx,
                                                   const float4* __restrict__ lmsg, const float4* __restrict__ fac, const float* __restrict__ Kd) {
  const uint32_t ws = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, p = ws * 64 + lane;
  const uint32_t cam_i = row_cam[p >> 4], lmk_i = lmk_idx[p];
  const int packed = __float_as_int(lmsg[(size_t)p * 4 + 3].y);
  const float4 zg = fac[((size_t)ws * kFacG + 13) * 64 + lane];  // floats 52..55: z = .z, .w
  const float4 q0 = ev.cam_rec[(size_t)cam_i * 3], q1 = ev.cam_rec[(size_t)cam_i * 3 + 1], q2 = ev.cam_rec[(size_t)cam_i * 3 + 2];
  const float4 lq = ev.lmk_mean[lmk_i];
  ride_metric(ev, (uint32_t)__builtin_amdgcn_readfirstlane((int)*ev.counter), ws, ws, lane, packed, zg.z, zg.w, q0, q1, q2, lq, Kd);
}

// The per-tile records k_sweep<EV> left in ring slot blockIdx.x reduced to ONE result, in the order of every other metric (see
// "THE ORDER OF THE METRIC'S SUMS"): thread j forms the block sum S_j k_eval would have produced, thread 0 then adds S_0, S_1, ...
// serially — what the host does with k_eval's block sums, done here so that 56 bytes per iteration cross PCIe instead of 49 KB.
struct EvalSum {       // == gbp_eval_out (include/gbp_mi355x.h)
  double sum_norm, sum_half_sq;
  unsigned long long n_active, n_relin, n_robust, n_nonfinite, n_nonpd;
};
__global__ __launch_bounds__(1024) void k_eval_fold(const EvalRide ev, EvalSum* out, uint32_t n_sums) {
  __shared__ double sh_d[2][1024];
  __shared__ unsigned sh_u[3][1024];
  const uint32_t j = threadIdx.x, slot = blockIdx.x, nb = ev.n_tiles / 4;
  const EvalRec* part = ev.part + (size_t)slot * ev.n_tiles;
  double s_norm = 0, s_half = 0;
  unsigned n_act = 0, n_rel = 0, n_rob = 0;
  if (j < n_sums) {
    for (uint32_t b0 = j; b0 < nb; b0 += 4 * n_sums) {      // four blocks' records in flight at once (clamped, unconditional), added in order
      EvalRec w[4][4];
      GBP_UNROLL
      for (int u = 0; u < 4; ++u) {
        const uint32_t b = b0 + (uint32_t)u * n_sums < nb ? b0 + (uint32_t)u * n_sums : b0;
        GBP_UNROLL
        for (int k = 0; k < 4; ++k) w[u][k] = part[4 * (size_t)b + k];
      }
      GBP_UNROLL
      for (int u = 0; u < 4; ++u) {
        if (b0 + (uint32_t)u * n_sums >= nb) break;
        s_norm += ((w[u][0].sum_norm + w[u][1].sum_norm) + w[u][2].sum_norm) + w[u][3].sum_norm;
        s_half += ((w[u][0].sum_half_sq + w[u][1].sum_half_sq) + w[u][2].sum_half_sq) + w[u][3].sum_half_sq;
        n_act += w[u][0].n_active + w[u][1].n_active + w[u][2].n_active + w[u][3].n_active;
        n_rel += w[u][0].n_relin + w[u][1].n_relin + w[u][2].n_relin + w[u][3].n_relin;
        n_rob += w[u][0].n_robust + w[u][1].n_robust + w[u][2].n_robust + w[u][3].n_robust;
      }
    }
  }
  sh_d[0][j] = s_norm; sh_d[1][j] = s_half; sh_u[0][j] = n_act; sh_u[1][j] = n_rel; sh_u[2][j] = n_rob;
  __syncthreads();
  if (j == 0) {
    EvalSum o;
    o.sum_norm = 0; o.sum_half_sq = 0; o.n_active = 0; o.n_relin = 0; o.n_robust = 0;
    for (uint32_t k0 = 0; k0 < n_sums; k0 += 16) {      // sixteen LDS reads in flight, added in order (entries >= n_sums hold zeros)
      double a[16], h[16];
      GBP_UNROLL
      for (int u = 0; u < 16; ++u) { a[u] = sh_d[0][k0 + u]; h[u] = sh_d[1][k0 + u]; }
      GBP_UNROLL
      for (int u = 0; u < 16; ++u)
        if (k0 + (uint32_t)u < n_sums) { o.sum_norm += a[u]; o.sum_half_sq += h[u]; }
      GBP_UNROLL
      for (int u = 0; u < 16; ++u) { o.n_active += sh_u[0][k0 + u]; o.n_relin += sh_u[1][k0 + u]; o.n_robust += sh_u[2][k0 + u]; }
    }
    o.n_nonfinite = ev.slot_health[2 * (size_t)slot]; o.n_nonpd = ev.slot_health[2 * (size_t)slot + 1];
    out[slot] = o;
  }
}

// =================================================================================================
// launchers
// =================================================================================================
static inline uint32_t blocks_for(uint64_t threads) { return (uint32_t)((threads + 255) / 256); }

#ifdef GBP_BUILD_EXPERIMENTS
bool lab_launch_sweep(const SweepArgs& a, uint32_t n_tiles, bool hoist, hipStream_t s);   // experiments/gbp_lab_kernels.hip
#endif
void launch_sweep(const SweepArgs& a, uint32_t n_tiles, bool hoist, hipStream_t s, bool ev) {
  const dim3 g(n_tiles / kWpb), b(64 * kWpb);
  const bool seg = a.seg_live && hoist && a.policy == 0u;      // a graph of many small cameras (gbp_api_ctx.cpp decides): the all-pad segments are skipped
  if (ev && hoist) {        // the metric rides along (gbp_iterate_eval_each beyond k_persist): the policies sweep_policy_for() chooses
    if (seg) hipLaunchKernelGGL((k_sweep<true, 0, true, true>), g, b, 0, s, a);
    else if (a.policy == kPolCmsgLoadCached) hipLaunchKernelGGL((k_sweep<true, kPolCmsgLoadCached, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((k_sweep<true, 0, true>), g, b, 0, s, a);
    return;
  }
#ifdef GBP_BUILD_EXPERIMENTS
  if (lab_launch_sweep(a, n_tiles, hoist, s)) return;     // a mapping experiment / an ablated sweep was asked for (SweepArgs.variant)
#endif
  if (!hoist) { hipLaunchKernelGGL(k_sweep<false>, g, b, 0, s, a); return; }
  if (seg) { hipLaunchKernelGGL((k_sweep<true, 0, false, true>), g, b, 0, s, a); return; }
  switch (a.policy) {      // the instantiations sweep_policy_for() (gbp_api_ctx.cpp) can choose
    case kPolCmsgLoadCached: hipLaunchKernelGGL((k_sweep<true, kPolCmsgLoadCached>), g, b, 0, s, a); break;
    case kPolLmsgLoadNt | kPolLmsgStoreNt: hipLaunchKernelGGL((k_sweep<true, kPolLmsgLoadNt | kPolLmsgStoreNt>), g, b, 0, s, a); break;
#ifdef GBP_BUILD_TEST_HOOKS      // the other combinations, for measurements (gbp_debug_force_sweep_policy)
    case 2: hipLaunchKernelGGL((k_sweep<true, 2>), g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_sweep<true, 3>), g, b, 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_sweep<true, 4>), g, b, 0, s, a); break;
    case 5: hipLaunchKernelGGL((k_sweep<true, 5>), g, b, 0, s, a); break;
    case 7: hipLaunchKernelGGL((k_sweep<true, 7>), g, b, 0, s, a); break;
#endif
    default: hipLaunchKernelGGL((k_sweep<true, 0>), g, b, 0, s, a); break;
  }
}
void launch_linearise(const SweepArgs& a, uint32_t n_tiles, hipStream_t s) {
  hipLaunchKernelGGL(k_linearise, dim3(n_tiles / 4), dim3(256), 0, s, a);
}
void launch_beliefs(BeliefArgs b, bool do_cam, bool do_lmk, hipStream_t s, bool ev) {
  b.cam_blocks = do_cam ? (b.n_cams + 3) / 4 : 0;
  const uint32_t lmk_blocks = do_lmk ? blocks_for((uint64_t)b.n_lmks * 4) : 0;
  b.lmk_blocks = lmk_blocks;
  if (b.cam_blocks + lmk_blocks == 0) return;
  if (ev) hipLaunchKernelGGL(k_beliefs_ev, dim3(b.cam_blocks + lmk_blocks), dim3(256), 0, s, b);
  else if (lmk_blocks == 0) hipLaunchKernelGGL(k_beliefs_cam, dim3(b.cam_blocks), dim3(256), 0, s, b);
  else hipLaunchKernelGGL(k_beliefs, dim3(b.cam_blocks + lmk_blocks), dim3(256), 0, s, b);
}
void launch_eval_ride(const EvalRide& ev, const uint32_t* row_cam, const uint32_t* lmk_idx, const float4* lmsg, const float4* fac, const float* K9_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_eval_ride, dim3(ev.n_tiles / 4), dim3(256), 0, s, ev, row_cam, lmk_idx, lmsg, fac, K9_dev);
}
void launch_eval_fold(const EvalRide& ev, uint32_t n_slots, void* out, hipStream_t s) {
  if (n_slots == 0) return;
  hipLaunchKernelGGL(k_eval_fold, dim3(n_slots), dim3(1024), 0, s, ev, static_cast<EvalSum*>(out), eval_blocks(ev.n_tiles));
}
// The placement of k_persist without its work: the same grid, the same filler rule, three barriers.  gbp_create runs it
// once: if the working workgroups of this graph are NOT all resident at once on this device (another partition mode, another
// dispatch order than workgroup b -> XCD b mod 8) the barrier gives up, *status is raised and the ctx stays on the
// two-kernel path — found at creation, in milliseconds, not in the middle of a run.
__global__ __launch_bounds__(256) void k_persist_probe(unsigned* sync, unsigned* status, uint32_t spread, uint32_t n_work_blocks) {
  // same register footprint class as k_persist is not needed for the residency question that matters here (which CUs the
  // working workgroups may use); one workgroup per CU is enforced by asking for the LDS a CU can give only once
  extern __shared__ float probe_lds[];
  if (threadIdx.x == 0) probe_lds[0] = 0.f;
  if ((int)spread > 1 && blockIdx.x % spread) return;
  const uint32_t nblk = (int)spread > 1 ? gridDim.x / spread : gridDim.x;
  (void)n_work_blocks;
  for (unsigned e = 1; e <= 3; ++e) grid_sync(sync, e * nblk, status, 1u);
}

static int persist_spread(uint32_t nb) { return nb <= 64 ? 4 : nb <= 128 ? 2 : 1; }
#ifdef GBP_BUILD_EXPERIMENTS
int lab_persist_spread(int spread);
#endif

bool persist_probe(uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, unsigned* sync, unsigned* status_dev, volatile unsigned* status_host,
                   bool cooperative, hipStream_t s) {
  const uint32_t nb = persist_blocks(n_tiles, n_cams, n_lmks, true);     // the larger of the two grids this graph is launched with
  const int spread = persist_spread(nb);
  if (hipMemsetAsync(sync, 0, kPersistSyncWords * sizeof(unsigned), s) != hipSuccess) return false;
  // 96 KiB of dynamic LDS per workgroup: at most ONE workgroup per CU (160 KiB), like k_persist's ~450 registers per lane
  // (profiles/r04_resources.md)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return false;
  if (cooperative) {
    uint32_t sp = (uint32_t)spread, nwb = nb;
    void* args[] = {&sync, &status_dev, &sp, &nwb};
    if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_persist_probe), dim3(nb * (uint32_t)spread), dim3(256), args, 96 * 1024, s) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
  } else {
    hipLaunchKernelGGL(k_persist_probe, dim3(nb * (uint32_t)spread), dim3(256), 96 * 1024, s, sync, status_dev, (uint32_t)spread, nb);
  }
  if (hipStreamSynchronize(s) != hipSuccess) return false;
  const bool ok = *status_host == 0u;
  *status_host = 0u;
  return ok;
}

// role separation of k_persist_flow (PersistArgs.separate): the grid has a tile-less wave for every camera (and metric mean) as long
// as that still is at most one workgroup per CU (0: it is not — roles and tiles share waves as in k_persist).  Measured
// (profiles/r05_persist_flow.md): with the metric after every iteration fr1xyz 13.1 -> 12.3 us per iteration, fr2robot2 12.3 -> 11.4,
// fr1desk unchanged; plain bursts 0.1 - 0.2 us faster.
static uint32_t persist_blocks_separate(uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, bool with_metric) {
  const uint64_t cx = (uint64_t)n_cams * (with_metric ? 2u : 1u), g = ((uint64_t)n_lmks + 15) / 16;
  // with the metric after every iteration the landmark owners move off the tile waves too where that fits (their fp64 metric solves,
  // and on graphs with landmarks of > 30 factors their extra rounds of loads, then delay no sweep: fr1desk 13.6 -> 12.5 us per
  // iteration, fr1xyz 12.3 -> 12.2, fr2robot2 11.4 -> 11.5; plain bursts do not gain from it)
  const uint64_t waves_full = (uint64_t)n_tiles + cx + g;
  if (with_metric && (waves_full + 3) / 4 <= 256u) return (uint32_t)((waves_full + 3) / 4);
  const uint64_t waves = n_tiles + cx > cx + g ? n_tiles + cx : cx + g;
  const uint64_t nb = (waves + 3) / 4;
  return nb <= 256u ? (uint32_t)nb : 0u;
}
PersistGrid persist_grid(uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, bool with_metric) {
  const uint32_t nbs = persist_blocks_separate(n_tiles, n_cams, n_lmks, with_metric);
  if (nbs) return PersistGrid{nbs, 1u, with_metric ? n_cams : 0u};
  return PersistGrid{persist_blocks(n_tiles, n_cams, n_lmks, with_metric), 0u, 0u};
}
uint32_t persist_blocks(uint32_t n_tiles, uint32_t n_cams, uint32_t n_lmks, bool with_metric) {
  if (const uint32_t nbs = persist_blocks_separate(n_tiles, n_cams, n_lmks, with_metric)) return nbs;
  const uint64_t waves_b = (uint64_t)n_cams + ((uint64_t)n_lmks + 15) / 16;
  const uint64_t waves = waves_b > n_tiles ? waves_b : n_tiles;
  const uint32_t nb = (uint32_t)((waves + 3) / 4);
  if (!with_metric) return nb;
  // a launch that carries the metric after EVERY iteration gets one more wave per camera for the metric roles: fr1xyz 52 -> 56
  // workgroups, fr2robot2 19 -> 24, fr1desk 61 -> 77 (every 2nd dispatch slot instead of every 4th; without barriers more workgroups cost
  // nothing: 16.3 -> 13.9 us per iteration with the metric, round 5) — as long as the larger grid still is one workgroup per CU.
  // Launches without the metric keep the smaller grid.
  const uint64_t waves_m = waves_b + n_cams > n_tiles ? waves_b + n_cams : n_tiles;
  const uint32_t nb_m = (uint32_t)((waves_m + 3) / 4);
  return nb_m <= 256u ? nb_m : nb;
}
int persist_max_resident_blocks() {
  int dev = 0, per_cu = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (k_persist_flow<true>), 256, 0) != hipSuccess) return 0;
  return per_cu * prop.multiProcessorCount;
}
void launch_copy_segments(const CopySegs& t, const unsigned* guard, hipStream_t s) {
  size_t most = 0;
  for (int i = 0; i < t.n; ++i) most = t.n4[i] > most ? t.n4[i] : most;
  const uint32_t blocks = (uint32_t)(most / 256 > 1024 ? 1024 : (most + 255) / 256);
  hipLaunchKernelGGL(k_copy_segments, dim3(blocks ? blocks : 1), dim3(256), 0, s, t, guard);
}
hipError_t launch_persist(PersistArgs A, bool cooperative, hipStream_t s) {
  A.n_lmk_groups = (A.b.n_lmks + 15) / 16;
  const uint32_t nb = persist_blocks(A.n_tiles, A.b.n_cams, A.b.n_lmks, A.ev.on != 0 && A.ev.each != 0);   // (one final metric: not worth 4 more workgroups in every barrier)
  // Placement: the grid is 4x the work and only every 4th workgroup works (the fillers leave at once).  Measured
  // (profiles/persist_placement.py, profiles/r03_small_graphs.md): with the working workgroups in consecutive dispatch slots a
  // few of them — always all four waves of a workgroup, on a CU next to another working CU — run their fp64-heavy sections
  // 2-2.7x slower and set the barrier-to-barrier time; every 2nd or 4th slot removes those outliers (fr1xyz 22.0 -> 19.3 us
  // per iteration in the traced build), every 3rd does not.
  // Workgroup b runs on XCD b % 8 (round-robin dispatch), so every spread-th workgroup lands on 8 / spread XCDs of 32 CUs:
  // the working workgroups stay co-resident (one per CU: 256 VGPRs + 160-215 AGPRs per lane, profiles/r04_resources.md) only while nb <= 32 * 8 / spread.
  int spread = persist_spread(nb);
#ifdef GBP_BUILD_EXPERIMENTS
  spread = lab_persist_spread(spread);      // placement studies / the forced time-out of tests/test_gpu_experiments.py (GBP_PERSIST_SPREAD)
#endif
  A.n_work_blocks = nb;
  A.spread = (uint32_t)spread;
  {
    const PersistGrid pg = persist_grid(A.n_tiles, A.b.n_cams, A.b.n_lmks, A.ev.on != 0 && A.ev.each != 0);
    A.separate = pg.separate;
    A.n_met = pg.n_met;
  }
  const uint32_t grid = spread > 1 ? nb * (uint32_t)spread : spread < -1 ? ((nb + 7) / 8) * (uint32_t)(-spread) * 8 : nb;
  const bool flow = A.f.lmsg != nullptr;      // hand-offs through tagged records (PersistFlow); without: the barrier kernel of the test-hooks build
  const void* f = A.ev.on ? reinterpret_cast<const void*>(k_persist_flow<true>) : reinterpret_cast<const void*>(k_persist_flow<false>);
#ifdef GBP_BUILD_TEST_HOOKS
  if (!flow) f = A.ev.on ? reinterpret_cast<const void*>(k_persist<true>) : reinterpret_cast<const void*>(k_persist<false>);
  else if (A.f.mirror4 != 0u)      // gbp_debug_persist_verify: every record published twice and compared by its consumers
    f = A.ev.on ? reinterpret_cast<const void*>(k_persist_flow<true, true>) : reinterpret_cast<const void*>(k_persist_flow<false, true>);
#else
  if (!flow) return hipErrorInvalidValue;
#endif
  void* args[] = {&A};
  if (cooperative) return hipLaunchCooperativeKernel(f, dim3(grid), dim3(256), args, 0, s);
  (void)hipLaunchKernel(f, dim3(grid), dim3(256), args, 0, s);
  return hipGetLastError();
}
void launch_state_get(const float4* lmsg, float* damping, int* packed, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_state_get, dim3(blocks_for(n)), dim3(256), 0, s, lmsg, damping, packed, n);
}
void launch_upload_scatter(float4* lmsg, float4* fac, const float4* st, const float* var, uint32_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_upload_scatter, dim3(blocks_for(n)), dim3(256), 0, s, lmsg, fac, st, var, n);
}
void launch_state_set(float4* lmsg, const int* new_count, const uint32_t* ctl, uint32_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_state_set, dim3(blocks_for(n)), dim3(256), 0, s, lmsg, new_count, ctl, n);
}
void launch_means(const float4* camb, const float4* lmkb, float* cam_mu, float* lmk_mu, uint32_t n_cams, uint32_t n_lmks,
                  unsigned long long* health2, unsigned long long* health2_next, bool count_cams, hipStream_t s) {
  hipLaunchKernelGGL(k_means, dim3(blocks_for((uint64_t)n_cams + n_lmks)), dim3(256), 0, s, (const float*)camb,
                     (const float*)lmkb, cam_mu, lmk_mu, n_cams, n_lmks, health2, health2_next, count_cams ? 1 : 0);
}
void launch_eval(const uint32_t* row_cam, const uint32_t* lmk_idx, const float4* lmsg, const float4* fac, const float* cam_mu,
                 const float* lmk_mu, const float* K9_dev, int num_undamped_iters, DeviceEval* partials,
                 unsigned long long* health2, unsigned long long* health2_out, uint32_t n_tiles, hipStream_t s) {
  hipLaunchKernelGGL(k_eval, dim3(eval_blocks(n_tiles)), dim3(256), 0, s, row_cam, lmk_idx, lmsg, fac, cam_mu, lmk_mu, K9_dev,
                     num_undamped_iters, partials, health2, health2_out, n_tiles);
}

#ifdef GBP_BUILD_TEST_HOOKS
#include "hooks/gbp_flow_torture.hip"        // k_flow_torture: the detector under the tagged records' untorn-16-byte-store assumption
#include "hooks/gbp_debug_math.hip"          // k_debug_math: the device math layer on caller-supplied vectors (tests only)
#endif
#ifdef GBP_BUILD_EXPERIMENTS
#include "experiments/gbp_lab_kernels.hip"   // mapping experiments and timing ablations (profiles/ only)
#endif

}  // namespace gbp
