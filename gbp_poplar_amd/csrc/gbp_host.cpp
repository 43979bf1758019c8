// gbp_host.cpp — CPU-side helpers of the GBP path's callers (no device code, no HIP calls).
//
// These are the functions the reference's main() runs around the Poplar engine; they feed or
// consume the hot path and are exported through the C-ABI (include/gbp_mi355x.h) so that the C++
// CLIs, the Python bindings and a foreign host can share one implementation:
//   gbp_bal_read / _write      BALProblem::LoadFile            reference ba/dataio.cpp:17-57
//   gbp_set_prior_lambda       set_prior_lambda + reprojectionJacFn   dataio.cpp:67-117, util.cpp:48-72
//   gbp_prior_scalings         prior-weakening scale factors   ba/ba.cpp:561-572
//   gbp_slam_create_flags / gbp_slam_update_flags              dataio.cpp:455-475, 477-508
//   gbp_slam_initialise_new_kf initialise_new_kf               util.cpp:183-197
//   gbp_eval_host              eval_reprojection_error         util.cpp:74-144
//   gbp_synth_generate         synthetic BAL graphs (the reference ships none)
//   gbp_bal_import_standard    9-parameter "Bundle Adjustment in the Large" files -> the reference's format
// The reference's O(E^2) / O((C+L)E) host loops (ba.cpp:267-279, dataio.cpp:76-116) are replaced by
// single O(E) passes with identical results.  Eigen is not used; where the reference calls
// Eigen's general inverse we solve in fp64 with partial pivoting.
#include "../../include/gbp_mi355x.h"
#include "../../include/gbp_mi355x_multi.h"      // gbp_landmark_partition
#include "gbp_export.hpp"
#include "gbp_threads.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Rot3 { float m[9]; };

// Rodrigues formula the way util.cpp:20-32 writes it (single expression, fp32).
Rot3 rodrigues_host(const float* w) {
  Rot3 R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  const float th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (th < 1e-6) return R;
  const float W[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
  const float a = std::sin(th) / th;
  const float b = (1 - std::cos(th)) / (th * th);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      float ww = 0.f;
      for (int k = 0; k < 3; ++k) ww += W[r * 3 + k] * W[k * 3 + c];
      R.m[r * 3 + c] = R.m[r * 3 + c] + (a * W[r * 3 + c] + b * ww);
    }
  return R;
}

inline float row_dot3(const float* row, const float* v) { return (row[0] * v[0] + row[1] * v[1]) + row[2] * v[2]; }

// Largest |entry| of the 2x9 reprojection Jacobian of util.cpp:48-72 at (cam, lmk).
float jacobian_peak(const float* cam, const float* lmk, const float* K) {
  const Rot3 R = rodrigues_host(cam + 3);
  float Ry[3], pc[3], p[3];
  for (int i = 0; i < 3; ++i) Ry[i] = row_dot3(R.m + 3 * i, lmk);
  for (int i = 0; i < 3; ++i) pc[i] = Ry[i] + cam[i];
  for (int i = 0; i < 3; ++i) p[i] = row_dot3(K + 3 * i, pc);
  const double z2 = static_cast<double>(p[2]) * static_cast<double>(p[2]);  // pow(p(2),2) is double
  const float jp[6] = {1 / p[2], 0, static_cast<float>(-p[0] / z2), 0, 1 / p[2], static_cast<float>(-p[1] / z2)};
  const float D[9] = {-0.f, Ry[2], -Ry[1], -Ry[2], -0.f, Ry[0], Ry[1], -Ry[0], -0.f};  // -hat(R*lmk)
  float jK[6], peak = 0.f;
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += jp[r * 3 + k] * K[k * 3 + c];
      jK[r * 3 + c] = s;
    }
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c) {
      float s = 0.f, t = 0.f;
      for (int k = 0; k < 3; ++k) {
        s += jK[r * 3 + k] * D[k * 3 + c];
        t += jK[r * 3 + k] * R.m[k * 3 + c];
      }
      peak = std::max(peak, std::max(std::fabs(jK[r * 3 + c]), std::max(std::fabs(s), std::fabs(t))));
    }
  return peak;
}

// x = A^-1 b, n <= 6: fp64 elimination with partial pivoting on fp32 inputs, result rounded to fp32.
void solve_pivot(const float* A, const float* b, int n, float* x) {
  double M[6][7];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) M[i][j] = A[i * n + j];
    M[i][n] = b[i];
  }
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = std::fabs(M[k][k]);
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(M[i][k]) > best) { best = std::fabs(M[i][k]); piv = i; }
    if (piv != k)
      for (int j = 0; j <= n; ++j) std::swap(M[k][j], M[piv][j]);
    for (int i = k + 1; i < n; ++i) {
      const double f = M[i][k] / M[k][k];
      for (int j = k; j <= n; ++j) M[i][j] -= f * M[k][j];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = M[i][n];
    for (int j = i + 1; j < n; ++j) s -= M[i][j] * static_cast<double>(x[j]);
    x[i] = static_cast<float>(s / M[i][i]);
  }
}

// ---- counter-based PRNG for the synthetic generator -------------------------------------------
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
struct Rng {
  uint64_t seed;
  uint64_t bits(uint64_t stream, uint64_t idx, uint64_t k) const {
    return splitmix64(splitmix64(seed ^ (stream * 0xD1B54A32D192ED03ull)) ^ splitmix64(idx * 0x2545F4914F6CDD1Dull + k));
  }
  double uni(uint64_t stream, uint64_t idx, uint64_t k) const {  // (0,1)
    return (static_cast<double>(bits(stream, idx, k) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  }
  double normal(uint64_t stream, uint64_t idx, uint64_t k) const {  // Box-Muller on draws 2k, 2k+1
    const double u1 = uni(stream, idx, 2 * k), u2 = uni(stream, idx, 2 * k + 1);
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2);
  }
};

void rodrigues_f64(const double* w, double* R) {
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  const double a = th > 1e-12 ? std::sin(th) / th : 1.0, b = th > 1e-12 ? (1 - std::cos(th)) / (th * th) : 0.5;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double ww = 0;
      for (int k = 0; k < 3; ++k) ww += W[r * 3 + k] * W[k * 3 + c];
      R[r * 3 + c] = (r == c ? 1.0 : 0.0) + a * W[r * 3 + c] + b * ww;
    }
}


// counter-based generator for the optional initialisation noise (splitmix64 -> Box-Muller): the reference seeds
// std::default_random_engine from the clock (dataio.cpp:334,349,406), which no test can pin
struct NoiseGen {
  unsigned long long s;
  explicit NoiseGen(unsigned long long seed) : s(seed) {}
  unsigned long long next() {
    s += 0x9E3779B97F4A7C15ull;
    unsigned long long x = s;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
  }
  double uni() { return ((double)(next() >> 11) + 0.5) / 9007199254740992.0; }
  float normal(float sd) { return sd * (float)(std::sqrt(-2.0 * std::log(uni())) * std::cos(6.283185307179586 * uni())); }
};

}  // namespace

// ---- text files of numbers, read by every host core ------------------------------------------------------
// Both file formats are one stream of whitespace-separated tokens whose meaning follows from the position alone:
//   n_hdr_int integers, n_hdr_dbl doubles, E records (int int double double), n_tail doubles; whatever follows is ignored
// (what a chain of fscanf("%d") / fscanf("%lf") calls reads: dataio.cpp:17-57).  The whole file is read into memory, cut into
// one piece per thread at whitespace, the tokens of every piece are counted, and with the counts summed every thread knows
// the position of its first token and converts its own — with strtol / strtod, the conversions fscanf itself makes, so the values
// are the ones the serial reader produced.  A token that strtol / strtod does not consume whole ("12abc", "1.5.3") is where fscanf's
// behaviour depends on what follows: such files are reported as `irregular` and the caller falls back to the fscanf chain.
struct NumberFile {
  int hdr_int[3] = {0, 0, 0};
  double hdr_dbl[4] = {0, 0, 0, 0};
  enum Status { kOk, kNoFile, kIrregular };
};

using gbp::host::host_threads;
using gbp::host::on_threads;

inline bool is_space(unsigned char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }      // isspace(), "C" locale

NumberFile::Status read_number_file(const char* path, int n_hdr_dbl, uint64_t E, uint64_t n_tail, NumberFile& h,
                                    int* rec_a, int* rec_b, double* rec_xy /* [2E] */, double* tail) {
  const bool trace = std::getenv("GBP_HOST_TRACE") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "read_number_file: %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  };
  struct Fd { int v; ~Fd() { if (v >= 0) ::close(v); } } file{::open(path, O_RDONLY)};      // (closed on every way out, an allocation failure included)
  const int fd = file.v;
  if (fd < 0) return NumberFile::kNoFile;
  std::unique_ptr<char[]> buf;
  size_t n = 0;
  struct stat sb;
  if (::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
    n = (size_t)sb.st_size;
    buf.reset(new char[n + 2]);
    const unsigned R = host_threads(n, 4u << 20);      // every thread copies (and first touches) its own part of the buffer
    std::vector<int> short_read(R, 0);
    on_threads(R, [&](unsigned t) {
      size_t lo = n / R * t;
      const size_t hi = t + 1 == R ? n : n / R * (t + 1);
      while (lo < hi) {
        const ssize_t got = ::pread(fd, buf.get() + lo, hi - lo, (off_t)lo);
        if (got <= 0) { short_read[t] = 1; return; }
        lo += (size_t)got;
      }
    });
    for (unsigned t = 0; t < R; ++t) if (short_read[t]) n = 0;
  }
  if (n == 0) return NumberFile::kIrregular;           // (not a regular file, empty, or it shrank under us: the fscanf chain decides)
  lap("file into memory");
  buf[n++] = ' ';                                      // every token ends at a whitespace character ...
  buf[n] = 0;                                          // ... and strtod never runs off the end
  const char* base = buf.get();
  const uint64_t n_hdr = 3 + (uint64_t)n_hdr_dbl, want = n_hdr + 4 * E + n_tail;

  const unsigned T = host_threads(n, 1u << 20);        // one thread per MB at most (the shipped sequences: 100 - 400 KB, one thread)
  std::vector<size_t> cut(T + 1);
  cut[0] = 0; cut[T] = n;
  for (unsigned t = 1; t < T; ++t) {                   // a piece starts behind a whitespace character: no token straddles two pieces
    size_t p = std::max(cut[t - 1], n / T * t);
    while (p < n && !is_space((unsigned char)base[p])) ++p;
    cut[t] = p;
  }
  std::vector<uint64_t> first(T + 1, 0);
  std::vector<int> bad(T, 0);
  auto run = [&](auto&& fn) { on_threads(T, fn); };
  run([&](unsigned t) {                                // pass 1: tokens per piece
    uint64_t k = 0;                                    // a token starts where a non-space follows a space (no loop-carried state: the loop vectorises)
    const size_t lo = cut[t], hi = cut[t + 1];
    if (lo < hi && lo == 0) k += !is_space((unsigned char)base[0]);
    for (size_t p = std::max<size_t>(lo, 1); p < hi; ++p) {
      const unsigned char c = (unsigned char)base[p], d = (unsigned char)base[p - 1];
      const unsigned sc = (c == ' ') | ((unsigned char)(c - 9) < 5), sd = (d == ' ') | ((unsigned char)(d - 9) < 5);
      k += (sc ^ 1u) & sd;
    }
    first[t + 1] = k;
  });
  for (unsigned t = 0; t < T; ++t) first[t + 1] += first[t];
  lap("tokens counted");
  if (first[T] < want) return NumberFile::kIrregular;      // fewer tokens than numbers: a truncated file, or glued tokens ("1.5.25") that only fscanf tells apart
  run([&](unsigned t) {                                // pass 2: the conversions
    uint64_t g = first[t];
    size_t p = cut[t];
    const size_t end = cut[t + 1];
    while (g < want) {
      while (p < end && is_space((unsigned char)base[p])) ++p;
      if (p >= end) break;
      char* q = nullptr;
      bool is_int;
      if (g < 3) is_int = true;
      else if (g < n_hdr) is_int = false;
      else if (g < n_hdr + 4 * E) is_int = ((g - n_hdr) & 3) < 2;
      else is_int = false;
      if (is_int) {
        const long v = std::strtol(base + p, &q, 10);
        const int iv = (v < -2147483647L - 1 || v > 2147483647L) ? -1 : (int)v;      // (no index of a valid file is out of int's range)
        if (g < 3) h.hdr_int[g] = iv;
        else { const uint64_t r = g - n_hdr; ((r & 3) == 0 ? rec_a : rec_b)[r >> 2] = iv; }
      } else {
        const double v = std::strtod(base + p, &q);
        if (g < n_hdr) h.hdr_dbl[g - 3] = v;
        else if (g < n_hdr + 4 * E) { const uint64_t r = g - n_hdr; rec_xy[2 * (r >> 2) + ((r & 3) - 2)] = v; }
        else tail[g - n_hdr - 4 * E] = v;
      }
      if (q == base + p || !is_space((unsigned char)*q)) { bad[t] = 1; return; }
      p = (size_t)(q - base);
      ++g;
    }
  });
  lap("tokens converted");
  for (unsigned t = 0; t < T; ++t) if (bad[t]) return NumberFile::kIrregular;
  return NumberFile::kOk;
}

GBP_EXPORT(gbp_bal_read_header, nullptr, (const char* path, gbp_bal* h),
           (path, h)) {
  if (!path || !h) return GBP_ERR_INVALID;
  FILE* f = std::fopen(path, "r");
  if (!f) return GBP_ERR_IO;
  int c = 0, l = 0, e = 0;
  bool ok = std::fscanf(f, "%d %d %d", &c, &l, &e) == 3 &&
            std::fscanf(f, "%lf %lf %lf %lf", &h->fx, &h->fy, &h->cx, &h->cy) == 4 && c > 0 && l > 0 && e > 0;
  std::fclose(f);
  if (!ok) return GBP_ERR_IO;
  h->n_cams = static_cast<uint32_t>(c);
  h->n_lmks = static_cast<uint32_t>(l);
  h->n_edges = static_cast<uint32_t>(e);
  return GBP_OK;
}

// Unlike the reference (FscanfOrDie only prints, dataio.cpp:59-65) a malformed file is an error.
GBP_EXPORT(gbp_bal_read, nullptr, (const char* path, gbp_bal* b),
           (path, b)) {
  if (!path || !b || !b->cam_id || !b->lmk_id || !b->observations || !b->cameras || !b->points) return GBP_ERR_INVALID;
  {      // every host core reads its piece of the file; only a file with a token no conversion takes whole goes to the fscanf chain below
    const uint64_t C = b->n_cams, L = b->n_lmks, E = b->n_edges;
    std::vector<double> tail(6 * C + 3 * L);
    static_assert(sizeof(int) == sizeof(uint32_t), "the indices are converted in place");
    NumberFile h;
    const NumberFile::Status st = read_number_file(path, 4, E, tail.size(), h, reinterpret_cast<int*>(b->cam_id), reinterpret_cast<int*>(b->lmk_id),
                                                   b->observations, tail.data());
    if (st == NumberFile::kNoFile) return GBP_ERR_IO;
    if (st == NumberFile::kOk) {
      if (h.hdr_int[0] < 0 || (uint64_t)h.hdr_int[0] != C || h.hdr_int[1] < 0 || (uint64_t)h.hdr_int[1] != L || h.hdr_int[2] < 0 || (uint64_t)h.hdr_int[2] != E) return GBP_ERR_IO;
      for (uint64_t i = 0; i < E; ++i)
        if ((int)b->cam_id[i] < 0 || b->cam_id[i] >= C || (int)b->lmk_id[i] < 0 || b->lmk_id[i] >= L) return GBP_ERR_IO;
      b->fx = h.hdr_dbl[0]; b->fy = h.hdr_dbl[1]; b->cx = h.hdr_dbl[2]; b->cy = h.hdr_dbl[3];
      std::copy(tail.begin(), tail.begin() + 6 * C, b->cameras);
      std::copy(tail.begin() + 6 * C, tail.end(), b->points);
      return GBP_OK;
    }
  }
  FILE* f = std::fopen(path, "r");
  if (!f) return GBP_ERR_IO;
  int c = 0, l = 0, e = 0;
  bool ok = std::fscanf(f, "%d %d %d", &c, &l, &e) == 3 &&
            std::fscanf(f, "%lf %lf %lf %lf", &b->fx, &b->fy, &b->cx, &b->cy) == 4;
  ok = ok && static_cast<uint32_t>(c) == b->n_cams && static_cast<uint32_t>(l) == b->n_lmks &&
       static_cast<uint32_t>(e) == b->n_edges;
  for (int i = 0; ok && i < e; ++i) {
    int ci, li;
    ok = std::fscanf(f, "%d %d %lf %lf", &ci, &li, &b->observations[2 * i], &b->observations[2 * i + 1]) == 4 &&
         ci >= 0 && ci < c && li >= 0 && li < l;
    if (ok) { b->cam_id[i] = static_cast<uint32_t>(ci); b->lmk_id[i] = static_cast<uint32_t>(li); }
  }
  for (int i = 0; ok && i < 6 * c; ++i) ok = std::fscanf(f, "%lf", &b->cameras[i]) == 1;
  for (int i = 0; ok && i < 3 * l; ++i) ok = std::fscanf(f, "%lf", &b->points[i]) == 1;
  std::fclose(f);
  return ok ? GBP_OK : GBP_ERR_IO;
}

GBP_EXPORT(gbp_bal_write, nullptr, (const char* path, const gbp_bal* b),
           (path, b)) {
  if (!path || !b) return GBP_ERR_INVALID;
  FILE* f = std::fopen(path, "w");
  if (!f) return GBP_ERR_IO;
  std::fprintf(f, "%u %u %u\n%.16e %.16e %.16e %.16e\n", b->n_cams, b->n_lmks, b->n_edges, b->fx, b->fy, b->cx, b->cy);
  for (uint32_t i = 0; i < b->n_edges; ++i)
    std::fprintf(f, "%u %u %.16e %.16e\n", b->cam_id[i], b->lmk_id[i], b->observations[2 * i], b->observations[2 * i + 1]);
  for (uint32_t i = 0; i < 6 * b->n_cams; ++i) std::fprintf(f, "%.16e\n", b->cameras[i]);
  for (uint32_t i = 0; i < 3 * b->n_lmks; ++i) std::fprintf(f, "%.16e\n", b->points[i]);
  return std::fclose(f) == 0 ? GBP_OK : GBP_ERR_IO;
}

// ---- standard "Bundle Adjustment in the Large" files -> the reference's format --------------
// Published BAL model (9 parameters per camera: Rodrigues R, t, f, k1, k2): P = R X + t, p = -P / P.z,
// pixel = f (1 + k1 |p|^2 + k2 |p|^4) p, image origin at the centre, y up, camera looking down -z.
// The reference wants one shared pin-hole K, no distortion, +z cameras stored as [t, w] (sequences/README.md:5-16):
//   * frame: P' = S P with S = diag(1,-1,-1) (a rotation by pi about x): R' = S R, t' = S t, (u', v') = (u, -v);
//   * lens: every observation is undistorted (Newton on the radial polynomial) and rescaled from its camera's f
//     to the shared focal length f_bar = mean f, principal point (0, 0);
//   * edges are re-sorted by (camera, landmark): the reference's SLAM mode and metric rely on camera-sorted files
//     (util.cpp:95-99, dataio.cpp:483-486), standard BAL files are sorted by point.
GBP_EXPORT(gbp_bal_import_standard_header, nullptr, (const char* path, gbp_bal* h),
           (path, h)) {
  if (!path || !h) return GBP_ERR_INVALID;
  FILE* f = std::fopen(path, "r");
  if (!f) return GBP_ERR_IO;
  int c = 0, l = 0, e = 0;
  const bool ok = std::fscanf(f, "%d %d %d", &c, &l, &e) == 3 && c > 0 && l > 0 && e > 0;
  std::fclose(f);
  if (!ok) return GBP_ERR_IO;
  h->n_cams = static_cast<uint32_t>(c);
  h->n_lmks = static_cast<uint32_t>(l);
  h->n_edges = static_cast<uint32_t>(e);
  h->fx = h->fy = h->cx = h->cy = 0.0;
  return GBP_OK;
}

GBP_EXPORT(gbp_bal_import_standard, nullptr, (const char* path, gbp_bal* b),
           (path, b)) {
  if (!path || !b || !b->cam_id || !b->lmk_id || !b->observations || !b->cameras || !b->points) return GBP_ERR_INVALID;
  FILE* f = std::fopen(path, "r");
  if (!f) return GBP_ERR_IO;
  int c = 0, l = 0, e = 0;
  bool ok = std::fscanf(f, "%d %d %d", &c, &l, &e) == 3 && static_cast<uint32_t>(c) == b->n_cams &&
            static_cast<uint32_t>(l) == b->n_lmks && static_cast<uint32_t>(e) == b->n_edges;
  struct Obs { uint32_t cam, lmk; double x, y; };
  std::vector<Obs> obs(ok ? e : 0);
  std::vector<double> cam9(ok ? 9ull * c : 0);
  bool parsed = false;
  if (ok) {      // every host core reads its piece (read_number_file); an irregular file goes through the fscanf chain below
    std::vector<int> ca(e), la(e);
    std::vector<double> xy(2ull * e), tail(9ull * c + 3ull * l);
    NumberFile h;
    const NumberFile::Status st = read_number_file(path, 0, (uint64_t)e, tail.size(), h, ca.data(), la.data(), xy.data(), tail.data());
    if (st == NumberFile::kOk) {
      for (int i = 0; ok && i < e; ++i) {
        ok = ca[i] >= 0 && ca[i] < c && la[i] >= 0 && la[i] < l;
        obs[i] = Obs{(uint32_t)ca[i], (uint32_t)la[i], xy[2ull * i], xy[2ull * i + 1]};
      }
      std::copy(tail.begin(), tail.begin() + 9ull * c, cam9.begin());
      std::copy(tail.begin() + 9ull * c, tail.end(), b->points);
      parsed = true;
    }
  }
  for (int i = 0; !parsed && ok && i < e; ++i) {
    int ci, li;
    ok = std::fscanf(f, "%d %d %lf %lf", &ci, &li, &obs[i].x, &obs[i].y) == 4 && ci >= 0 && ci < c && li >= 0 && li < l;
    if (ok) { obs[i].cam = static_cast<uint32_t>(ci); obs[i].lmk = static_cast<uint32_t>(li); }
  }
  for (size_t i = 0; !parsed && ok && i < cam9.size(); ++i) ok = std::fscanf(f, "%lf", &cam9[i]) == 1;
  for (int i = 0; !parsed && ok && i < 3 * l; ++i) ok = std::fscanf(f, "%lf", &b->points[i]) == 1;
  std::fclose(f);
  if (!ok) return GBP_ERR_IO;

  double fbar = 0.0;
  for (int i = 0; i < c; ++i) fbar += cam9[9ull * i + 6];
  fbar /= c;
  b->fx = b->fy = fbar;
  b->cx = b->cy = 0.0;

  for (int i = 0; i < c; ++i) {
    const double* w = &cam9[9ull * i];
    // R = exp([w]x) as a unit quaternion, then q' = q_S * q with q_S = (0; 1,0,0) (rotation by pi about x)
    const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double k = th < 1e-12 ? 0.5 : std::sin(0.5 * th) / th;
    const double q[4] = {std::cos(0.5 * th), k * w[0], k * w[1], k * w[2]};
    double r[4] = {-q[1], q[0], -q[3], q[2]};   // (0,1,0,0) * (q0,q1,q2,q3)
    if (r[0] < 0) for (double& v : r) v = -v;   // angle in [0, pi]
    const double vn = std::sqrt(r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    const double ang = 2.0 * std::atan2(vn, r[0]);
    const double sc = vn < 1e-300 ? 0.0 : ang / vn;
    double* out = &b->cameras[6ull * i];
    out[0] = w[3]; out[1] = -w[4]; out[2] = -w[5];                  // t' = S t
    out[3] = sc * r[1]; out[4] = sc * r[2]; out[5] = sc * r[3];      // w' = log(S R)
  }

  // undistort + rescale + flip v, then order by (camera, landmark)
  for (Obs& o : obs) {
    const double* p9 = &cam9[9ull * o.cam];
    const double fi = p9[6], k1 = p9[7], k2 = p9[8];
    if (!(fi > 0)) return GBP_ERR_INVALID;
    const double dx = o.x / fi, dy = o.y / fi, rd = std::sqrt(dx * dx + dy * dy);
    double ru = rd;                                // solve ru (1 + k1 ru^2 + k2 ru^4) = rd
    for (int it = 0; it < 50; ++it) {
      const double r2 = ru * ru, g = ru * (1 + k1 * r2 + k2 * r2 * r2) - rd, dg = 1 + 3 * k1 * r2 + 5 * k2 * r2 * r2;
      if (!(std::fabs(dg) > 1e-12)) break;
      const double step = g / dg;
      ru -= step;
      if (std::fabs(step) <= 1e-16 * std::fabs(ru)) break;
    }
    const double s = rd > 0 ? ru / rd : 1.0;
    o.x = fbar * s * dx;
    o.y = -fbar * s * dy;
  }
  std::stable_sort(obs.begin(), obs.end(), [](const Obs& a, const Obs& c2) {
    return a.cam != c2.cam ? a.cam < c2.cam : a.lmk < c2.lmk;
  });
  for (int i = 0; i < e; ++i) {
    b->cam_id[i] = obs[i].cam; b->lmk_id[i] = obs[i].lmk;
    b->observations[2ull * i] = obs[i].x; b->observations[2ull * i + 1] = obs[i].y;
  }
  return GBP_OK;
}

GBP_EXPORT(gbp_set_prior_lambda, nullptr, (const gbp_problem* p, float var, const float* cam_file, const float* lmk_file, const float* cam_mean, const float* lmk_mean, float* ce, float* cl, float* le, float* ll),
           (p, var, cam_file, lmk_file, cam_mean, lmk_mean, ce, cl, le, ll)) {
  if (!p || !cam_file || !lmk_file || !cam_mean || !lmk_mean || !ce || !cl || !le || !ll) return GBP_ERR_INVALID;
  const uint32_t C = p->n_cams, L = p->n_lmks, E = p->n_edges;
  // one pass: max over incident observations (dataio.cpp:78-87, 99-108).  A maximum does not depend on the order it is taken in: large
  // graphs are cut into one range of factors per host thread, each with its own peaks, and the peaks are merged.
  const unsigned T = host_threads(E, 1u << 16);
  std::vector<std::vector<float>> pc(T, std::vector<float>(C, 0.f)), pl(T, std::vector<float>(L, 0.f));      // (allocated here: nothing throws inside a thread)
  std::vector<int> bad(T, 0);
  on_threads(T, [&](unsigned t) {
    const uint64_t lo = (uint64_t)E * t / T, hi = (uint64_t)E * (t + 1) / T;
    for (uint64_t e = lo; e < hi; ++e) {
      const uint32_t c = p->cam_id[e], l = p->lmk_id[e];
      if (c >= C || l >= L) { bad[t] = 1; return; }
      const float m = jacobian_peak(cam_file + 6ull * c, lmk_file + 3ull * l, p->K);
      pc[t][c] = std::max(pc[t][c], m);
      pl[t][l] = std::max(pl[t][l], m);
    }
  });
  for (unsigned t = 0; t < T; ++t) if (bad[t]) return GBP_ERR_INVALID;
  std::vector<float>& peak_c = pc[0];
  std::vector<float>& peak_l = pl[0];
  for (unsigned t = 1; t < T; ++t) {
    for (uint32_t c = 0; c < C; ++c) peak_c[c] = std::max(peak_c[c], pc[t][c]);
    for (uint32_t l = 0; l < L; ++l) peak_l[l] = std::max(peak_l[l], pl[t][l]);
  }
  std::fill(cl, cl + 36ull * C, 0.f);
  std::fill(ll, ll + 9ull * L, 0.f);
  for (uint32_t c = 0; c < C; ++c) {  // lam = pow(max_jac,2)/var in double (dataio.cpp:88), eta = mean*lam
    const float lam = static_cast<float>((static_cast<double>(peak_c[c]) * static_cast<double>(peak_c[c])) / static_cast<double>(var));
    for (int i = 0; i < 6; ++i) { ce[6ull * c + i] = cam_mean[6ull * c + i] * lam; cl[36ull * c + 7 * i] = lam; }
  }
  for (uint32_t l = 0; l < L; ++l) {
    const float lam = static_cast<float>((static_cast<double>(peak_l[l]) * static_cast<double>(peak_l[l])) / static_cast<double>(var));
    for (int i = 0; i < 3; ++i) { le[3ull * l + i] = lmk_mean[3ull * l + i] * lam; ll[9ull * l + 4 * i] = lam; }
  }
  return GBP_OK;
}

GBP_EXPORT(gbp_prior_scalings, nullptr, (uint32_t C, uint32_t L, const float* cpl, float steps, float weaker, float first_std, float* cs, float* ls),
           (C, L, cpl, steps, weaker, first_std, cs, ls)) {
  if (!cpl || !cs || !ls) return GBP_ERR_INVALID;
  // ba.cpp:564: exp(-1/steps * log(lambda00 * pow(std,2))) — pow(float,int) promotes to double;
  // ba.cpp:566,571: exp(-2/steps * log(weaker)) stays in fp32.
  const float weak = std::exp(-2 / steps * std::log(weaker));
  for (uint32_t c = 0; c < C; ++c)
    cs[c] = (c < 2) ? static_cast<float>(std::exp(-1 / steps * std::log(cpl[36ull * c] * std::pow(static_cast<double>(first_std), 2))))
                    : weak;
  for (uint32_t l = 0; l < L; ++l) ls[l] = weak;
  return GBP_OK;
}

GBP_EXPORT(gbp_slam_create_flags, nullptr, (const gbp_problem* p, uint32_t steps, uint32_t* active, uint32_t* cwf, uint32_t* lwf, uint32_t* laf),
           (p, steps, active, cwf, lwf, laf)) {
  if (!p || !active || !cwf || !lwf || !laf || p->n_cams < 2) return GBP_ERR_INVALID;
  cwf[0] = cwf[1] = steps;
  for (uint32_t e = 0; e < p->n_edges; ++e)
    if (p->cam_id[e] <= 1) { active[e] = 1; lwf[p->lmk_id[e]] = steps; }
  std::copy(lwf, lwf + p->n_lmks, laf);
  return GBP_OK;
}

GBP_EXPORT(gbp_slam_update_flags, nullptr, (const gbp_problem* p, uint32_t steps, uint32_t dc, uint32_t* active, uint32_t* lwf, uint32_t* cwf, uint32_t* laf, int32_t* n_new),
           (p, steps, dc, active, lwf, cwf, laf, n_new)) {
  if (!p || !active || !cwf || !lwf || !laf || dc + 1 >= p->n_cams || steps == 0) return GBP_ERR_INVALID;
  for (uint32_t e = 0; e < p->n_edges; ++e) {
    if (p->cam_id[e] == dc + 1) active[e] = 1;
    if (p->cam_id[e] <= dc + 1) lwf[p->lmk_id[e]] = steps;
  }
  std::fill(cwf, cwf + p->n_cams, 0u);
  cwf[dc + 1] = steps;
  int total = 0;
  for (uint32_t l = 0; l < p->n_lmks; ++l) {
    lwf[l] -= laf[l];
    laf[l] += lwf[l];
    total += static_cast<int>(lwf[l]);
  }
  if (n_new) *n_new = total / static_cast<int>(steps);
  return GBP_OK;
}

GBP_EXPORT(gbp_slam_initialise_new_kf, nullptr, (uint32_t dc, const float* cbe, const float* cbl, const float* cpl, float* cpe),
           (dc, cbe, cbl, cpl, cpe)) {
  if (!cbe || !cbl || !cpl || !cpe) return GBP_ERR_INVALID;
  float mu[6];
  solve_pivot(cbl + 36ull * dc, cbe + 6ull * dc, 6, mu);
  for (int i = 0; i < 6; ++i) {
    float s = 0.f;
    for (int k = 0; k < 6; ++k) s += cpl[36ull * (dc + 1) + 6 * i + k] * mu[k];
    cpe[6ull * (dc + 1) + i] = s;
  }
  return GBP_OK;
}

GBP_EXPORT(gbp_eval_host, nullptr, (const gbp_problem* p, const uint32_t* active, const float* meas, const float* cbe, const float* cbl, const float* lbe, const float* lbl, double* sum_norm, double* sum_half_sq, uint64_t* n_active),
           (p, active, meas, cbe, cbl, lbe, lbl, sum_norm, sum_half_sq, n_active)) {
  if (!p || !active || !meas || !cbe || !cbl || !lbe || !lbl || !sum_norm || !sum_half_sq || !n_active) return GBP_ERR_INVALID;
  std::vector<float> cmu(6ull * p->n_cams), lmu(3ull * p->n_lmks);
  for (uint32_t c = 0; c < p->n_cams; ++c) solve_pivot(cbl + 36ull * c, cbe + 6ull * c, 6, &cmu[6ull * c]);
  for (uint32_t l = 0; l < p->n_lmks; ++l) solve_pivot(lbl + 9ull * l, lbe + 3ull * l, 3, &lmu[3ull * l]);
  double a = 0, b = 0;
  uint64_t n = 0;
  for (uint32_t e = 0; e < p->n_edges; ++e) {
    if (active[e] != 1) continue;
    const float* cm = &cmu[6ull * p->cam_id[e]];
    const float* lm = &lmu[3ull * p->lmk_id[e]];
    const Rot3 R = rodrigues_host(cm + 3);
    float pcf[3], pr[2];
    for (int i = 0; i < 3; ++i) pcf[i] = row_dot3(R.m + 3 * i, lm);
    for (int i = 0; i < 3; ++i) pcf[i] += cm[i];
    for (int i = 0; i < 2; ++i) pr[i] = row_dot3(p->K + 3 * i, pcf) / pcf[2];
    const float r0 = meas[2ull * e] - pr[0], r1 = meas[2ull * e + 1] - pr[1];
    a += std::sqrt(r0 * r0 + r1 * r1);
    b += static_cast<float>(0.5 * (r0 * r0 + r1 * r1));
    ++n;
  }
  *sum_norm = a; *sum_half_sq = b; *n_active = n;
  return GBP_OK;
}

// Belief means mu = Lambda^-1 eta of every variable (the solution a caller takes away), with the solve of the metric
// (util.cpp:103-108 uses Eigen's general inverse; here fp64 partial pivoting).  Variables without information
// (Lambda = 0: never observed) come out non-finite, like in the reference.
GBP_EXPORT(gbp_belief_means, nullptr, (uint32_t C, uint32_t L, const float* cbe, const float* cbl, const float* lbe, const float* lbl, double* cameras, double* points),
           (C, L, cbe, cbl, lbe, lbl, cameras, points)) {
  if (!cbe || !cbl || !lbe || !lbl || !cameras || !points) return GBP_ERR_INVALID;
  float mu[6];
  for (uint32_t c = 0; c < C; ++c) {
    solve_pivot(cbl + 36ull * c, cbe + 6ull * c, 6, mu);
    for (int i = 0; i < 6; ++i) cameras[6ull * c + i] = mu[i];
  }
  for (uint32_t l = 0; l < L; ++l) {
    solve_pivot(lbl + 9ull * l, lbe + 3ull * l, 3, mu);
    for (int i = 0; i < 3; ++i) points[3ull * l + i] = mu[i];
  }
  return GBP_OK;
}

// Contiguous landmark ranges balanced by incident-factor count: bounds[r] = first landmark whose cumulative
// degree reaches r/world of the factors.  A factor lives with its landmark, so this balances the sweep's work.
GBP_EXPORT(gbp_landmark_partition, nullptr, (const gbp_problem* p, int world, uint32_t* bounds),
           (p, world, bounds)) {
  if (!p || !bounds || world < 1) return GBP_ERR_INVALID;
  std::vector<uint64_t> csum((size_t)p->n_lmks + 1, 0);
  for (uint32_t e = 0; e < p->n_edges; ++e) {
    if (p->lmk_id[e] >= p->n_lmks) return GBP_ERR_INVALID;
    csum[(size_t)p->lmk_id[e] + 1]++;
  }
  for (size_t l = 0; l < p->n_lmks; ++l) csum[l + 1] += csum[l];
  const uint64_t total = csum[p->n_lmks];
  bounds[0] = 0;
  for (int r = 1; r < world; ++r) {
    const uint64_t target = total * (uint64_t)r / (uint64_t)world;
    const uint32_t b = (uint32_t)(std::lower_bound(csum.begin(), csum.end(), target) - csum.begin());
    bounds[r] = std::min(std::max(b, bounds[r - 1]), p->n_lmks);
  }
  bounds[world] = p->n_lmks;
  return GBP_OK;
}

// add_cam_trans_noise / add_cam_rot_noise / add_lmk_noise (dataio.cpp:330-415) with an explicit seed.  The first two
// cameras anchor the gauge and stay exact (dataio.h:114-119, k = 2).  Draw order: translations, rotations, landmarks.
GBP_EXPORT(gbp_init_add_noise, nullptr, (uint32_t C, uint32_t L, float tn, float rn_deg, float ltn, uint64_t seed, float* cam, float* lmk),
           (C, L, tn, rn_deg, ltn, seed, cam, lmk)) {
  if (!cam || !lmk) return GBP_ERR_INVALID;
  NoiseGen rng(seed);
  if (tn != 0.f)
    for (uint32_t c = 2; c < C; ++c)
      for (int i = 0; i < 3; ++i) cam[6 * (size_t)c + i] += rng.normal(tn);
  if (rn_deg != 0.f) {
    for (uint32_t c = 2; c < C; ++c) {   // dataio.cpp:345-400: rotate the camera-to-world orientation about a random axis
      const float ang = rng.normal(rn_deg) * (float)M_PI / 180.f;
      const int axis = (int)(rng.next() % 3);
      float Rn[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      const float cs = std::cos(ang), sn = std::sin(ang);
      if (axis == 0) { Rn[4] = cs; Rn[5] = -sn; Rn[7] = sn; Rn[8] = cs; }
      else if (axis == 1) { Rn[0] = cs; Rn[2] = sn; Rn[6] = -sn; Rn[8] = cs; }
      else { Rn[0] = cs; Rn[1] = -sn; Rn[3] = sn; Rn[4] = cs; }
      float* x = cam + 6 * (size_t)c;
      const Rot3 Rw2c = rodrigues_host(x + 3);
      // Tc2w = [R^T, -R^T t]; its rotation block becomes Rn R^T, its translation (the camera centre) is kept
      float Rc2w[9], ctr[3], Rp[9];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rc2w[i * 3 + j] = Rw2c.m[j * 3 + i];
      for (int i = 0; i < 3; ++i) ctr[i] = -(Rc2w[i * 3] * x[0] + Rc2w[i * 3 + 1] * x[1] + Rc2w[i * 3 + 2] * x[2]);
      for (int i = 0; i < 3; ++i)      // back to world->camera: R' = (Rn Rc2w)^T
        for (int j = 0; j < 3; ++j) {
          float acc = 0.f;
          for (int k = 0; k < 3; ++k) acc += Rn[j * 3 + k] * Rc2w[k * 3 + i];
          Rp[i * 3 + j] = acc;
        }
      for (int i = 0; i < 3; ++i) x[i] = -(Rp[i * 3] * ctr[0] + Rp[i * 3 + 1] * ctr[1] + Rp[i * 3 + 2] * ctr[2]);
      const float d = 0.5f * (Rp[0] + Rp[4] + Rp[8] - 1);                 // so3log, util.cpp:34-46
      const float f = std::acos(d) / (2 * std::sqrt(1 - d * d));
      x[3] = f * (Rp[7] - Rp[5]);
      x[4] = f * (Rp[2] - Rp[6]);
      x[5] = f * (Rp[3] - Rp[1]);
    }
  }
  if (ltn != 0.f)
    for (size_t i = 0; i < 3 * (size_t)L; ++i) lmk[i] += rng.normal(ltn);
  return GBP_OK;
}

// av_depth_init (dataio.cpp:417-453): every landmark starts at the point (0,0,1) of the camera frame of the LOWEST-indexed
// keyframe observing it (the reference passes av_depth but uses the literal depth 1.0, dataio.cpp:437).  One O(E) pass:
// the reference's camera-major visiting order picks, for each landmark, its observer with the smallest camera index.
GBP_EXPORT(gbp_init_av_depth, nullptr, (const gbp_problem* p, const float* cam_mean, float* lmk_mean),
           (p, cam_mean, lmk_mean)) {
  if (!p || !cam_mean || !lmk_mean) return GBP_ERR_INVALID;
  std::vector<uint32_t> first(p->n_lmks, ~0u);
  for (uint32_t e = 0; e < p->n_edges; ++e) {
    if (p->cam_id[e] >= p->n_cams || p->lmk_id[e] >= p->n_lmks) return GBP_ERR_INVALID;
    first[p->lmk_id[e]] = std::min(first[p->lmk_id[e]], p->cam_id[e]);
  }
  std::vector<float> spot(3 * (size_t)p->n_cams);
  for (uint32_t c = 0; c < p->n_cams; ++c) {
    const Rot3 R = rodrigues_host(cam_mean + 6 * (size_t)c + 3);
    const float v[3] = {0.f - cam_mean[6 * (size_t)c], 0.f - cam_mean[6 * (size_t)c + 1], 1.f - cam_mean[6 * (size_t)c + 2]};
    for (int i = 0; i < 3; ++i) spot[3 * (size_t)c + i] = R.m[i] * v[0] + R.m[3 + i] * v[1] + R.m[6 + i] * v[2];  // R^T (p - t)
  }
  for (uint32_t l = 0; l < p->n_lmks; ++l)
    if (first[l] != ~0u)
      for (int i = 0; i < 3; ++i) lmk_mean[3 * (size_t)l + i] = spot[3 * (size_t)first[l] + i];
  return GBP_OK;
}

GBP_EXPORT(gbp_synth_generate, nullptr, (uint32_t C, uint32_t L, uint32_t obs_per_lmk, uint64_t seed, gbp_bal* out, double* gt_cams, double* gt_pts),
           (C, L, obs_per_lmk, seed, out, gt_cams, gt_pts)) {
  if (!out || !out->cam_id || !out->lmk_id || !out->observations || !out->cameras || !out->points || C < 2 || L < 1)
    return GBP_ERR_INVALID;
  const uint32_t per = std::min(obs_per_lmk, C);
  if (per == 0 || static_cast<uint64_t>(L) * per > 0xFFFFFFF0ull) return GBP_ERR_INVALID;
  const Rng rng{seed};
  enum { S_LMK = 1, S_CAMC = 2, S_CAMJ = 3, S_PICK = 4, S_PIX = 5, S_INITC = 6, S_INITL = 7 };
  const double fx = 500, fy = 500, cx = 320, cy = 240;
  out->n_cams = C; out->n_lmks = L; out->n_edges = L * per;
  out->fx = fx; out->fy = fy; out->cx = cx; out->cy = cy;
  std::vector<double> cam(6ull * C), pts(3ull * L);
  for (uint32_t l = 0; l < L; ++l)
    for (int k = 0; k < 3; ++k) pts[3ull * l + k] = -2.0 + 4.0 * rng.uni(S_LMK, l, k);
  for (uint32_t c = 0; c < C; ++c) {
    for (uint64_t attempt = 0;; ++attempt) {  // redraw until the axis-angle is well away from 0 and pi
      const uint64_t id = c + attempt * 0x100000000ull;
      double d[3];  // direction: normalised gaussian triple
      for (int k = 0; k < 3; ++k) d[k] = rng.normal(S_CAMC, id, k);
      const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
      if (n2 < 1e-12) continue;
      const double r = 8.0 + 4.0 * rng.uni(S_CAMC, id, 16), inv = 1.0 / std::sqrt(n2);
      const double ctr[3] = {d[0] * inv * r, d[1] * inv * r, d[2] * inv * r};
      double z[3] = {-ctr[0] / r, -ctr[1] / r, -ctr[2] / r};          // optical axis looks at the origin
      double up[3] = {0, 0, 1};
      if (std::fabs(z[2]) > 0.9) { up[0] = 1; up[2] = 0; }
      double x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
      const double xn = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      for (double& v : x) v /= xn;
      const double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
      const double R0[9] = {x[0], x[1], x[2], y[0], y[1], y[2], z[0], z[1], z[2]};  // rows = camera axes in world
      const double tr = R0[0] + R0[4] + R0[8];
      const double th = std::acos(std::max(-1.0, std::min(1.0, 0.5 * (tr - 1.0))));
      if (th < 0.05 || th > 3.0) continue;
      const double s = th / (2.0 * std::sin(th));
      double w[3] = {s * (R0[7] - R0[5]), s * (R0[2] - R0[6]), s * (R0[3] - R0[1])};
      for (int k = 0; k < 3; ++k) w[k] += 0.05 * rng.normal(S_CAMJ, id, k);
      if (w[0] * w[0] + w[1] * w[1] + w[2] * w[2] < 1e-3) continue;
      double R[9];
      rodrigues_f64(w, R);
      for (int k = 0; k < 3; ++k) {
        cam[6ull * c + k] = -(R[3 * k] * ctr[0] + R[3 * k + 1] * ctr[1] + R[3 * k + 2] * ctr[2]);
        cam[6ull * c + 3 + k] = w[k];
      }
      break;
    }
  }
  if (gt_cams) std::copy(cam.begin(), cam.end(), gt_cams);
  if (gt_pts) std::copy(pts.begin(), pts.end(), gt_pts);

  // choose `per` distinct cameras per landmark, bucket the edges by camera (landmarks ascend inside a bucket)
  std::vector<uint32_t> pick(static_cast<size_t>(L) * per), deg(C + 1, 0);
  for (uint32_t l = 0; l < L; ++l) {
    uint32_t* mine = &pick[static_cast<size_t>(l) * per];
    uint32_t got = 0;
    for (uint64_t k = 0; got < per; ++k) {
      const uint32_t c = static_cast<uint32_t>(rng.bits(S_PICK, l, k) % C);
      if (std::find(mine, mine + got, c) == mine + got) mine[got++] = c;
    }
    for (uint32_t j = 0; j < per; ++j) deg[mine[j] + 1]++;
  }
  for (uint32_t c = 0; c < C; ++c) deg[c + 1] += deg[c];
  std::vector<uint32_t> fill(deg.begin(), deg.end() - 1);
  std::vector<double> Rc(9ull * C);
  for (uint32_t c = 0; c < C; ++c) rodrigues_f64(&cam[6ull * c + 3], &Rc[9ull * c]);
  for (uint32_t l = 0; l < L; ++l)
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t c = pick[static_cast<size_t>(l) * per + j];
      const uint32_t e = fill[c]++;
      const double* R = &Rc[9ull * c];
      const double* t = &cam[6ull * c];
      const double* y = &pts[3ull * l];
      const double X = R[0] * y[0] + R[1] * y[1] + R[2] * y[2] + t[0];
      const double Y = R[3] * y[0] + R[4] * y[1] + R[5] * y[2] + t[1];
      const double Z = R[6] * y[0] + R[7] * y[1] + R[8] * y[2] + t[2];
      out->cam_id[e] = c;
      out->lmk_id[e] = l;
      const uint64_t key = static_cast<uint64_t>(l) * per + j;
      out->observations[2ull * e] = fx * X / Z + cx + rng.normal(S_PIX, key, 0);
      out->observations[2ull * e + 1] = fy * Y / Z + cy + rng.normal(S_PIX, key, 1);
    }
  // initial values = ground truth + noise; cameras 0 and 1 are the gauge anchors and stay exact (ba.cpp:563-564)
  for (uint32_t c = 0; c < C; ++c)
    for (int k = 0; k < 6; ++k) {
      const double sd = (c < 2) ? 0.0 : (k < 3 ? 0.05 : 0.01);
      out->cameras[6ull * c + k] = cam[6ull * c + k] + sd * rng.normal(S_INITC, c, k);
    }
  for (uint32_t l = 0; l < L; ++l)
    for (int k = 0; k < 3; ++k) out->points[3ull * l + k] = pts[3ull * l + k] + 0.1 * rng.normal(S_INITL, l, k);
  return GBP_OK;
}

