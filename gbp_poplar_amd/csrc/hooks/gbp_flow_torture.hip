// hooks/gbp_flow_torture.hip — TEST HOOK (include/gbp_mi355x_debug.h: gbp_debug_flow_torture): the detector under the one
// hardware property k_persist_flow relies on that no architecture manual promises — "a 16-byte aligned store of one lane is not
// observed torn by a wave on another XCD" (gbp_kernels.h: PersistFlow).  Included by gbp_kernels.hip inside namespace gbp when the
// library is built with -DGBP_BUILD_TEST_HOOKS; the product library does not contain it.
//
// The same instructions as the product (XwBuf::st4 / ld4: 128-bit buffer stores / loads with sc1), the same protocol (tagged records,
// two halves alternating by the parity of the round, a consumer polls until every record carries the round it waits for), nothing else:
//   round r: every lane of every wave stores K records, then reads the K records per lane of its PARTNER wave — workgroup bid ^ mask,
//            i.e. another XCD for an odd mask under round-robin placement (mask = 8: the same XCD, the control) — and checks them.
// EVERY word of a record is a function of the round: words 0..2 = mix(tag, record index, word), word 3 = the tag.  A record whose tag is
// the awaited one but whose payload belongs to an earlier round of the same slot (r - 2: the previous content of that half) is a TORN
// record — exactly the silent failure the product could not notice; any other mismatch is counted as CORRUPT.  The wait is bounded
// (time-outs are counted and end the wave's run).
// CONTROL (inject != 0): every 64th record of a lane's round is stored the way a tearing memory system would show it — its tag word
// first (a 4-byte store), the full record a little later — so that a consumer can meet the awaited tag over the previous payload:
// the detector must then REPORT torn records (tests/test_gpu_parity.py asserts > 0), which shows that it can.
GBP_DEV unsigned torture_word(unsigned tag, unsigned idx, unsigned c) {
  unsigned x = tag * 0x9E3779B1u ^ (idx + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 1u) * 0xC2B2AE35u;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
  return x;
}

template <int K>
__global__ __launch_bounds__(256) void k_flow_torture(float4* buf, unsigned long long* out /* [4]: torn, corrupt, time-outs, records checked */,
                                                      int rounds, unsigned mask, unsigned tag0, int inject) {
  const unsigned nblk = gridDim.x, bid = blockIdx.x, wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned nw = nblk * 4u, wave = bid * 4u + wib;
  const unsigned other = ((bid ^ mask) % nblk) * 4u + wib;      // (host: nblk is a power of two >= 2 * mask, so this is a pairing)
  const XwBuf X(buf);
  unsigned long long torn = 0, corrupt = 0, checked = 0, timeouts = 0;
  for (int r = 1; r <= rounds; ++r) {
    const unsigned tag = tag0 + (unsigned)r;
    const unsigned half = ((unsigned)r & 1u) * nw * 64u * (unsigned)K;
    GBP_UNROLL
    for (int k = 0; k < K; ++k) {
      const unsigned idx = (wave * (unsigned)K + (unsigned)k) * 64u + lane;
      const float4 rec = make_float4(__uint_as_float(torture_word(tag, idx, 0u)), __uint_as_float(torture_word(tag, idx, 1u)),
                                     __uint_as_float(torture_word(tag, idx, 2u)), __uint_as_float(tag));
      if (inject && ((idx + (unsigned)r) & 63u) == 0u) X.st1((half + idx) * 4u + 3u, __uint_as_float(tag));      // the control: the tag alone ...
      else X.st4(half + idx, rec);
    }
    if (inject) {      // ... and the record itself only after every other record of the round is on its way
      __builtin_amdgcn_s_sleep(64);
      GBP_UNROLL
      for (int k = 0; k < K; ++k) {
        const unsigned idx = (wave * (unsigned)K + (unsigned)k) * 64u + lane;
        if (((idx + (unsigned)r) & 63u) == 0u)
          X.st4(half + idx, make_float4(__uint_as_float(torture_word(tag, idx, 0u)), __uint_as_float(torture_word(tag, idx, 1u)),
                                        __uint_as_float(torture_word(tag, idx, 2u)), __uint_as_float(tag)));
      }
    }
    float4 v[K];
    bool arrived = false;
    unsigned long long t0 = 0;
    for (unsigned spin = 0;; ++spin) {
      asm volatile("" ::: "memory");      // the loads are re-issued every time round
      GBP_UNROLL
      for (int k = 0; k < K; ++k) v[k] = X.ld4(half + (other * (unsigned)K + (unsigned)k) * 64u + lane);
      bool ok = true;
      GBP_UNROLL
      for (int k = 0; k < K; ++k) ok = ok && __float_as_uint(v[k].w) == tag;
      if (__all(ok)) { arrived = true; break; }
      if ((spin & 255u) == 255u) {
        const unsigned long long now = wall_clock64();
        if (t0 == 0) t0 = now;
        if (now - t0 > 200000000ull) break;      // 2 s: the partner is gone (it timed out, or the grid is not co-resident)
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!arrived) { timeouts += 1; break; }
    GBP_UNROLL
    for (int k = 0; k < K; ++k) {
      const unsigned idx = (other * (unsigned)K + (unsigned)k) * 64u + lane;
      const unsigned w0 = __float_as_uint(v[k].x), w1 = __float_as_uint(v[k].y), w2 = __float_as_uint(v[k].z);
      const bool good = w0 == torture_word(tag, idx, 0u) && w1 == torture_word(tag, idx, 1u) && w2 == torture_word(tag, idx, 2u);
      if (!good) {
        // which words belong to the previous content of this slot (the same half, two rounds ago; zeros before round 3)?
        const unsigned old = tag - 2u;
        const bool o0 = r > 2 ? w0 == torture_word(old, idx, 0u) : w0 == 0u;
        const bool o1 = r > 2 ? w1 == torture_word(old, idx, 1u) : w1 == 0u;
        const bool o2 = r > 2 ? w2 == torture_word(old, idx, 2u) : w2 == 0u;
        const bool n0 = w0 == torture_word(tag, idx, 0u), n1 = w1 == torture_word(tag, idx, 1u), n2 = w2 == torture_word(tag, idx, 2u);
        if ((o0 || n0) && (o1 || n1) && (o2 || n2)) torn += 1; else corrupt += 1;
      }
      checked += 1;
    }
  }
  if (torn) atomicAdd(&out[0], torn);
  if (corrupt) atomicAdd(&out[1], corrupt);
  if (timeouts) atomicAdd(&out[2], timeouts);
  atomicAdd(&out[3], checked);
}

// blocks: a power of two in [16, 256] (one workgroup per CU: co-resident on an MI355X); K in {4, 16}
bool launch_flow_torture(float4* buf, unsigned long long* out, int blocks, int K, int rounds, unsigned mask, unsigned tag0, int inject, hipStream_t s) {
  if (K == 16) hipLaunchKernelGGL((k_flow_torture<16>), dim3(blocks), dim3(256), 0, s, buf, out, rounds, mask, tag0, inject);
  else if (K == 4) hipLaunchKernelGGL((k_flow_torture<4>), dim3(blocks), dim3(256), 0, s, buf, out, rounds, mask, tag0, inject);
  else return false;
  return true;
}
