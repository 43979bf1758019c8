// cli_common.hpp — shared host driver code of the `ba` and `slam` executables.
//
// Keeps the command-line contract of the reference's two programs (flag names and defaults of
// ba/ba.cpp:394-476 and ba/slam.cpp:394-476, stdout lines of ba.cpp:996,1004,1026-1028 and
// slam.cpp:1073-1076) on top of the C-ABI instead of a Poplar engine.  Boost.Program_options is
// replaced by a small `--name value` / `--name=value` parser.  Differences, all deliberate:
//   * `--help` prints the options and exits 0 (the reference throws an uncaught exception, ba.cpp:469-472);
//   * a malformed data file is an error (the reference only prints "Invalid UW data file.");
//   * `--ipus N` is kept as an alias of `--gpus N`: N > 1 starts one process per GPU (forked before anything touches
//     HIP), landmark-sharded, camera partial sums all-gathered once per iteration by RCCL over xGMI
//     (ba.cpp:414-417,617-649 spread one Poplar graph over N x 1216 tiles); `--camspertile` is accepted and ignored;
//   * `--seed S` makes the initialisation-noise flags reproducible (the reference seeds from the clock,
//     dataio.cpp:334,349,406); `--eval_every K` thins the per-iteration read-back + metric (default 1).
#pragma once
#include "../../include/gbp_mi355x.h"
#include "../../include/gbp_mi355x_multi.h"       // --ipus N: one forked rank per GPU
#include "../../include/gbp_mi355x_compat.h"      // MetricPipe: the metric in two halves (gbp_iterate_eval / gbp_eval_end), so that printing overlaps the next iterations

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <sched.h>
#include <chrono>
#include <cmath>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

namespace cli {

// Where a run spends its wall time, from exec to exit (--profile: the `startup` object of gbp_profile.json; the "Total time" line):
// for the shipped sequences the iteration loop is a few per cent of what a user of ./ba waits for (profiles/r06_configs.md).
struct Phases {
  using clk = std::chrono::steady_clock;
  clk::time_point t_main = clk::now(), last = t_main;
  double exec_to_main_s = 0;                                  // exec -> main(): the dynamic loader (libamdhip64 and what it pulls in)
  std::vector<std::pair<std::string, double>> v;              // phase, seconds — in order
  std::string create_info;                                    // gbp_create's own breakdown (gbp_last_error(ctx) right after it)
  Phases() {
    // start time of the process (clock ticks since boot, /proc/self/stat field 22) against the time since boot now: 10 ms resolution
    double up = 0;
    if (FILE* f = std::fopen("/proc/uptime", "r")) { if (std::fscanf(f, "%lf", &up) != 1) up = 0; std::fclose(f); }
    if (FILE* f = std::fopen("/proc/self/stat", "r")) {
      char buf[1024];
      const size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
      std::fclose(f);
      buf[n] = 0;
      if (const char* p = std::strrchr(buf, ')')) {          // the fields behind "(comm)": state is field 3
        unsigned long long start = 0;
        int field = 2;
        for (const char* q = p + 1; *q && field < 22; ++q)
          if (*q == ' ') { ++field; if (field == 22) start = std::strtoull(q + 1, nullptr, 10); }
        const long hz = sysconf(_SC_CLK_TCK);
        if (start && hz > 0 && up > 0) exec_to_main_s = std::max(0.0, up - (double)start / (double)hz);
      }
    }
  }
  void mark(const char* name) {
    const auto n = clk::now();
    v.emplace_back(name, std::chrono::duration<double>(n - last).count());
    last = n;
  }
  double since_exec() const { return exec_to_main_s + std::chrono::duration<double>(clk::now() - t_main).count(); }
  double get(const char* name) const { for (const auto& p : v) if (p.first == name) return p.second; return 0; }
};
inline Phases& phases() { static Phases p; return p; }

struct Options {
  std::string bal_file;
  int n_iters = 1500;           // ba only
  int iters_between_kfs = 700;  // slam only
  bool profile = false;
  int gpus = 1;
  int cams_per_tile = 1;
  float transnoise = 0.f, rotnoise = 0.f, lmktrans_noise = 0.f;
  bool av_depth_on = false;
  float av_depth = 1.f;
  float reproj_meas_var = 4.f;
  float prior_std_weaker_factor = 100.f;
  float first_cam_prior_std = 0.01f;
  float steps = 5.f;
  int iters_before_damping = 15;
  bool verbose = false;
  unsigned long long seed = 0;
  int eval_every = 1;
  int transport = 0;            // --transport: 0 auto, 1 RCCL, 2 host-staged (ranks sharing a GPU)
  std::string out_file;         // --out_file: write the refined problem (belief means) in the input's format
  bool force_sharded = false;   // --force_sharded: run the multi-rank code path (fork, shard ctx, communicator) with one rank
};

inline void usage(bool slam) {
  std::cout << "Options:\n"
               "  --help                         Show command help\n"
               "  --bal_file arg                 Set the bal file\n";
  if (slam) std::cout << "  --iters_between_kfs arg (=700) Number of iterations of GBP between new keyframes\n";
  else std::cout << "  --n_iters arg (=1500)          Number of iterations of synchronous GBP\n";
  std::cout << "  --profile arg (=0)             Save profile report after execution\n"
               "  --gpus arg (=1)                Number of GPUs (alias: --ipus)\n"
               "  --camspertile arg (=1)         accepted for compatibility, ignored\n"
               "  --tn arg (=0)                  Set keyframe translation noise value\n"
               "  --rn arg (=0)                  Set keyframe rotation noise value\n"
               "  --ltn arg (=0)                 Set landmark translation noise noise value\n"
               "  --avdepth_on arg (=0)          initialise landmarks at a depth in front of the first observing keyframe\n"
               "  --avdepth arg (=1)             Average depth\n"
               "  --reproj_meas_var arg (=4)     Variance of the reprojection measurement model\n"
               "  --prior_std_weaker_factor arg (=100)\n"
               "  --first_cam_prior_std arg (=0.01)\n"
               "  --steps arg (=5)               The priors are gradually weakened over this many steps\n"
               "  --undamped_start arg (=15)     Number of undamped iterations before damping GBP\n"
               "  --v arg (=0)                   Verbose: print beliefs\n"
               "  --seed arg (=0)                seed of the initialisation noise (0 = from the clock)\n"
               "  --eval_every arg (=1)          read back + evaluate every K iterations\n"
               "  --transport arg (=auto)        exchange between ranks: auto | rccl | host (ranks sharing one GPU)\n"
               "  --out_file arg                 write the refined cameras / landmarks (belief means) in the input's format\n";
}

// returns 0 = run, 1 = exit with code 0 (help), 2 = exit with code 1 (error)
inline int parse(int argc, char** argv, bool slam, Options& o) {
  std::map<std::string, std::string> kv;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.rfind("--", 0) != 0) { std::cerr << "unrecognised argument '" << a << "'\n"; return 2; }
    a = a.substr(2);
    if (a == "help") { usage(slam); return 1; }
    std::string val;
    const size_t eq = a.find('=');
    if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); }
    else if (i + 1 < argc) val = argv[++i];
    else { std::cerr << "the required argument for option '--" << a << "' is missing\n"; return 2; }
    kv[a] = val;
  }
  auto B = [](const std::string& s) { return s == "1" || s == "true" || s == "on" || s == "yes"; };
  try {
    for (auto& p : kv) {
      const std::string &k = p.first, &v = p.second;
      if (k == "bal_file") o.bal_file = v;
      else if (k == "n_iters" && !slam) o.n_iters = std::stoi(v);
      else if (k == "iters_between_kfs" && slam) o.iters_between_kfs = std::stoi(v);
      else if (k == "profile") o.profile = B(v);
      else if (k == "ipus" || k == "gpus") o.gpus = std::stoi(v);
      else if (k == "camspertile") o.cams_per_tile = std::stoi(v);
      else if (k == "tn") o.transnoise = std::stof(v);
      else if (k == "rn") o.rotnoise = std::stof(v);
      else if (k == "ltn") o.lmktrans_noise = std::stof(v);
      else if (k == "avdepth_on") o.av_depth_on = B(v);
      else if (k == "avdepth") o.av_depth = std::stof(v);
      else if (k == "reproj_meas_var") o.reproj_meas_var = std::stof(v);
      else if (k == "prior_std_weaker_factor") o.prior_std_weaker_factor = std::stof(v);
      else if (k == "first_cam_prior_std") o.first_cam_prior_std = std::stof(v);
      else if (k == "steps") o.steps = std::stof(v);
      else if (k == "undamped_start") o.iters_before_damping = std::stoi(v);
      else if (k == "v") o.verbose = B(v);
      else if (k == "seed") o.seed = std::stoull(v);
      else if (k == "eval_every") o.eval_every = std::max(1, std::stoi(v));
      else if (k == "force_sharded") o.force_sharded = B(v);
      else if (k == "out_file") o.out_file = v;
      else if (k == "transport") o.transport = v == "rccl" ? 1 : (v == "host" ? 2 : (v == "auto" ? 0 : std::stoi(v)));
      else { std::cerr << "unrecognised option '--" << k << "'\n"; return 2; }
    }
  } catch (const std::exception&) {
    std::cerr << "invalid option value\n";
    return 2;
  }
  if (o.bal_file.empty()) { std::cerr << "the option '--bal_file' is required but missing\n"; return 2; }
  return 0;
}

// ---- problem data (what ba.cpp:489-604 builds on the host) -----------------------------------------
struct Problem {
  gbp_bal bal{};
  std::vector<uint32_t> cam_id, lmk_id;
  std::vector<double> obs, cams, pts;
  std::vector<float> K, meas, var, cam_file, lmk_file, cam_mean, lmk_mean;
  std::vector<float> cpe, cpl, lpe, lpl, cscale, lscale, damping;
  std::vector<int32_t> count;
  std::vector<uint32_t> active, cwf, lwf;
  gbp_problem prob{};
};

inline int load_problem(const Options& o, Problem& P) {
  const bool trace = std::getenv("GBP_HOST_TRACE") != nullptr;      // milliseconds of every step on stderr
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "load_problem: %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  };
  if (gbp_bal_read_header(o.bal_file.c_str(), &P.bal) != GBP_OK) {
    std::cerr << "ERROR: unable to open file " << o.bal_file << "\n";  // ba.cpp:484-487
    return 1;
  }
  const uint32_t C = P.bal.n_cams, L = P.bal.n_lmks, E = P.bal.n_edges;
  P.cam_id.resize(E); P.lmk_id.resize(E); P.obs.resize(2 * (size_t)E); P.cams.resize(6 * (size_t)C); P.pts.resize(3 * (size_t)L);
  P.bal.cam_id = P.cam_id.data(); P.bal.lmk_id = P.lmk_id.data(); P.bal.observations = P.obs.data();
  P.bal.cameras = P.cams.data(); P.bal.points = P.pts.data();
  lap("header + arrays");
  if (gbp_bal_read(o.bal_file.c_str(), &P.bal) != GBP_OK) {
    std::cerr << "Invalid UW data file.\nERROR: unable to read file " << o.bal_file << "\n";
    return 1;
  }
  lap("gbp_bal_read");
  P.K = {(float)P.bal.fx, 0.f, (float)P.bal.cx, 0.f, (float)P.bal.fy, (float)P.bal.cy, 0.f, 0.f, 1.f};  // ba.cpp:494-495
  P.meas.resize(2 * (size_t)E);
  for (size_t i = 0; i < 2 * (size_t)E; ++i) P.meas[i] = (float)P.obs[i];
  P.var.assign(E, o.reproj_meas_var);
  P.cam_file.resize(6 * (size_t)C); P.lmk_file.resize(3 * (size_t)L);
  for (size_t i = 0; i < P.cam_file.size(); ++i) P.cam_file[i] = (float)P.cams[i];
  for (size_t i = 0; i < P.lmk_file.size(); ++i) P.lmk_file[i] = (float)P.pts[i];
  P.cam_mean = P.cam_file; P.lmk_mean = P.lmk_file;
  P.prob.n_cams = C; P.prob.n_lmks = L; P.prob.n_edges = E; P.prob.cam_id = P.cam_id.data(); P.prob.lmk_id = P.lmk_id.data();
  std::memcpy(P.prob.K, P.K.data(), 9 * sizeof(float));

  lap("measurements, variances, means as float");
  // ba.cpp:536-548; the reference seeds from the clock (dataio.cpp:334,349,406): --seed 0 does the same
  const unsigned long long seed = o.seed ? o.seed : (unsigned long long)std::chrono::system_clock::now().time_since_epoch().count();
  if (o.transnoise != 0.f)
    std::cout << "\nAdding Gaussian noise with std: " << o.transnoise << "m to the keyframe translaton intialisations\n";
  if (o.rotnoise != 0.f)
    std::cout << "Adding Gaussian noise with std: " << o.rotnoise << " to the keyframe rotation intialisations\n";
  const bool lmk_noise = o.lmktrans_noise != 0.f && !o.av_depth_on;
  if (lmk_noise)
    std::cout << "Adding Gaussian noise with std: " << o.lmktrans_noise << "m to the landmark intialisations\n";
  gbp_init_add_noise(C, L, o.transnoise, o.rotnoise, lmk_noise ? o.lmktrans_noise : 0.f, seed, P.cam_mean.data(), P.lmk_mean.data());
  if (o.av_depth_on) {
    std::cout << "Initialising all landmarks at an average depth of: " << o.av_depth << "\n";
    gbp_init_av_depth(&P.prob, P.cam_mean.data(), P.lmk_mean.data());
  }

  lap("initialisation options");
  P.cpe.resize(6 * (size_t)C); P.cpl.resize(36 * (size_t)C); P.lpe.resize(3 * (size_t)L); P.lpl.resize(9 * (size_t)L);
  gbp_set_prior_lambda(&P.prob, o.reproj_meas_var, P.cam_file.data(), P.lmk_file.data(), P.cam_mean.data(), P.lmk_mean.data(),
                       P.cpe.data(), P.cpl.data(), P.lpe.data(), P.lpl.data());
  lap("gbp_set_prior_lambda");
  P.cscale.resize(C); P.lscale.resize(L);
  gbp_prior_scalings(C, L, P.cpl.data(), o.steps, o.prior_std_weaker_factor, o.first_cam_prior_std, P.cscale.data(), P.lscale.data());
  std::cout << "Completed loading data!\n";
  P.damping.assign(E, 0.f);
  P.count.assign(E, -o.iters_before_damping);  // ba.cpp:581
  lap("scalings, damping state");
  // mu / oldmu: the reference uploads 9 E zeros for each (ba.cpp:582-583); gbp_upload takes NULL for "zeros" — 72 MB per million factors
  // that are neither allocated nor compared here
  return 0;
}

inline gbp_state_in state_in(const Problem& P) {
  gbp_state_in s{};
  s.damping = P.damping.data(); s.damping_count = P.count.data(); s.mu = nullptr; s.oldmu = nullptr;
  s.active_flag = P.active.data(); s.cam_scaling = P.cscale.data(); s.lmk_scaling = P.lscale.data();
  s.cam_weaken_flag = P.cwf.data(); s.lmk_weaken_flag = P.lwf.data();
  s.cam_priors_eta = P.cpe.data(); s.cam_priors_lambda = P.cpl.data(); s.lmk_priors_eta = P.lpe.data();
  s.lmk_priors_lambda = P.lpl.data(); s.measurements = P.meas.data(); s.meas_variances = P.var.data();
  return s;
}

struct Readback {
  std::vector<float> cbe, cbl, lbe, lbl;
  gbp_state_out out{};
  Readback(uint32_t C, uint32_t L) : cbe(6 * (size_t)C), cbl(36 * (size_t)C), lbe(3 * (size_t)L), lbl(9 * (size_t)L) {
    out.cam_beliefs_eta = cbe.data(); out.cam_beliefs_lambda = cbl.data();
    out.lmk_beliefs_eta = lbe.data(); out.lmk_beliefs_lambda = lbl.data();
  }
};

inline void print_verbose(const Readback& r) {  // ba.cpp:1030-1051
  std::cout << "\nKeyframe Eta beliefs: \n";
  for (unsigned i = 0; i < 6; ++i) std::printf("%.12f  ", r.cbe[6 + i]);
  std::cout << "\nKeyframe Lambda beliefs: \n";
  for (unsigned i = 0; i < 36; ++i) std::printf("%.12f  ", r.cbl[36 + i]);
  std::cout << '\n';
  std::cout << "\nLandmark Eta beliefs: \n";
  for (unsigned i = 0; i < 12; ++i) std::printf("%.12f  ", r.lbe[i]);
  std::cout << "\nLandmark Lambda beliefs: \n";
  for (unsigned i = 0; i < 18; ++i) std::printf("%.12f  ", r.lbl[i]);
  std::cout << '\n';
  std::fflush(stdout);
}

// --out_file: the solution (belief means) in the format of the input, observations unchanged.  Single-process runs only:
// a rank of a sharded run holds the landmarks of its own range.
inline int write_solution(const Options& o, const Problem& P, gbp_ctx* ctx, bool sharded) {
  if (o.out_file.empty()) return 0;
  if (sharded) { std::cout << "--out_file is ignored with --ipus N > 1 (each rank holds its own landmark range)\n"; return 0; }
  const uint32_t C = P.bal.n_cams, L = P.bal.n_lmks;
  std::vector<float> cbe(6 * (size_t)C), cbl(36 * (size_t)C), lbe(3 * (size_t)L), lbl(9 * (size_t)L);
  gbp_state_out out{};
  out.cam_beliefs_eta = cbe.data(); out.cam_beliefs_lambda = cbl.data(); out.lmk_beliefs_eta = lbe.data(); out.lmk_beliefs_lambda = lbl.data();
  if (gbp_read(ctx, &out) != GBP_OK) { std::cerr << "gbp_read failed: " << gbp_last_error(ctx) << "\n"; return 1; }
  std::vector<double> cams(6 * (size_t)C), pts(3 * (size_t)L);
  gbp_belief_means(C, L, cbe.data(), cbl.data(), lbe.data(), lbl.data(), cams.data(), pts.data());
  gbp_bal res = P.bal;
  res.cameras = cams.data(); res.points = pts.data();
  if (gbp_bal_write(o.out_file.c_str(), &res) != GBP_OK) { std::cerr << "ERROR: unable to write " << o.out_file << "\n"; return 1; }
  std::cout << "Refined problem written to " << o.out_file << "\n";
  return 0;
}

// ---- one process per GPU (`--ipus N` / `--gpus N`) -------------------------------------------------------------------
struct RankCtx {
  int rank = 0, world = 1;
  void* region = nullptr;       // shared rendezvous / staging region (gbp_comm_region_*), world > 1 only
  gbp_shard shard{0, 1, 0, 0};
};

// A single-process run brings the HIP runtime up (120 - 240 ms: the first call that needs it) on a second thread WHILE the main thread
// reads the file and builds the priors — nothing for the shipped sequences (4 ms of host work), the runtime's whole start-up for a
// file of a million factors (1 - 2 s of parsing).  Never with --ipus N > 1: the ranks are forked before anything touches HIP.
struct RuntimeWarmup {
  std::thread t;
  void start(bool on) {      // (the switch: profiles/big_file_cli.py's A/B; a system that refuses the thread: the first HIP call stays where it was)
    if (!on || std::getenv("GBP_CLI_NO_WARMUP")) return;
    try { t = std::thread([] { (void)gbp_device_count(); }); } catch (...) {}
  }
  void wait() { if (t.joinable()) t.join(); }
  ~RuntimeWarmup() { wait(); }
};
inline RuntimeWarmup& runtime_warmup() { static RuntimeWarmup w; return w; }

// EMPIRICAL, measured and not explained (profiles/r06_exit_probe.txt, profiles/exit_probe.py): on this driver stack (ROCm 7.0 user space, the pool's
// amdgpu / KFD) a process whose address space has taken page faults from SIX OR MORE threads before the HIP runtime comes up is gone 40 - 60 ms sooner
// after its main() has ended — `ba fr1xyz --n_iters 200`: 0.15 - 0.19 s -> 0.105 - 0.12 s of process wall — than one whose memory was only ever touched by
// one to four threads.  What it takes: threads that TOUCH fresh memory (eight threads that allocate and free without touching, or thirty-two that do
// nothing: no effect; one thread touching 128 MB: no effect; malloc's arenas and thresholds, CPU affinity: no effect).  It showed first as the multi-threaded
// file reader making the 10^6-factor run exit faster than the 12 908-factor one.  Eight threads write 1 MB each: < 1 ms.  GBP_CLI_NO_PRIME=1 turns it off.
inline void prime_address_space() {
  if (std::getenv("GBP_CLI_NO_PRIME")) return;
  constexpr int kThreads = 8;
  constexpr size_t kBytes = (size_t)1 << 20;
  std::unique_ptr<char[]> buf(new (std::nothrow) char[kThreads * kBytes]);
  if (!buf) return;
  std::thread th[kThreads];
  try {      // (a system that refuses threads: the run goes on without them)
    const int pin = std::getenv("GBP_CLI_PRIME_PIN") ? sched_getcpu() : -1;      // (experiment: all eight on the caller's CPU — r06_exit_probe.txt)
    for (int i = 0; i < kThreads; ++i)
      th[i] = std::thread([p = buf.get() + (size_t)i * kBytes, i, pin] {
        if (pin >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(pin, &set); (void)sched_setaffinity(0, sizeof(set), &set); }
        std::memset(p, i + 1, kBytes);
      });
  } catch (...) {
  }
  for (auto& t : th)
    if (t.joinable()) t.join();
  asm volatile("" ::"r"(buf.get()) : "memory");
}

inline int round_up_pow2(int n) {   // ba.cpp:617-621: nIPUs is rounded up to a power of two
  int p = 1;
  while (p < n) p *= 2;
  return p;
}

// Creates the ctx of this rank on its GPU, landmark-sharded for world > 1, with the communicator attached.
inline int create_rank_ctx(const Options& o, const Problem& P, RankCtx& rk, gbp_ctx** ctx) {
  phases().mark("host_setup_s");                                  // (flags, fork, the lines printed so far)
  runtime_warmup().wait();
  const int ndev = gbp_device_count();
  phases().mark("runtime_init_s");                                // the HIP runtime comes up with the first call that needs it — what of it the file's parsing did not hide
  if (ndev <= 0) {
    std::cout << "Could not find a device\n";                      // ba.cpp:652-655
    return 255;
  }
  const bool sharded = rk.region != nullptr;
  if (sharded) {
    if (gbp_set_device(rk.rank % ndev) != GBP_OK) { std::cout << "Could not find a device\n" << gbp_last_error(nullptr) << "\n"; return 255; }
    std::vector<uint32_t> bounds((size_t)rk.world + 1);
    gbp_landmark_partition(&P.prob, rk.world, bounds.data());
    rk.shard = gbp_shard{rk.rank, rk.world, bounds[rk.rank], bounds[rk.rank + 1]};
  } else {
    rk.shard = gbp_shard{0, 1, 0, P.bal.n_lmks};
  }
  if (gbp_create(&P.prob, nullptr, sharded ? &rk.shard : nullptr, ctx) != GBP_OK) {
    std::cout << "Could not find a device\n" << gbp_last_error(nullptr) << "\n";
    return 255;
  }
  if (sharded) {
    if (gbp_comm_init(*ctx, rk.region, o.transport) != GBP_OK) {
      std::cerr << "rank " << rk.rank << ": " << gbp_last_error(*ctx) << "\n";
      return 1;
    }
    std::cout << "Exchange between the " << rk.world << " ranks: " << gbp_comm_transport(*ctx)
              << (ndev < rk.world ? " (fewer GPUs than ranks: ranks share a GPU)" : "") << "\n";
  }
  if (const char* info = gbp_last_error(*ctx)) phases().create_info = info;
  phases().mark("create_s");
  return 0;
}

// Runs body(rank) in `world` processes forked from this one — which must not have touched HIP yet — and supervises them:
// the exit status is the first failing rank's, and a rank that dies wakes the others out of their barriers.
template <class F> int run_ranks(int world, uint32_t n_cams, bool force, F&& body) {
  RankCtx rk;
  if (world <= 1 && !force) return body(rk);
  if (!std::getenv("HSA_ENABLE_IPC_MODE_LEGACY")) setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 1);   // dmabuf IPC for RCCL
  const size_t bytes = gbp_comm_region_bytes(n_cams, world);
  void* region = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (region == MAP_FAILED || gbp_comm_region_init(region, bytes, n_cams, world) != GBP_OK) {
    std::cerr << "could not create the shared region of the ranks\n";
    return 1;
  }
  std::cout.flush();
  std::fflush(nullptr);
  std::vector<pid_t> pid((size_t)world, -1);
  bool fork_failed = false;
  for (int r = 0; r < world; ++r) {
    pid[r] = fork();
    if (pid[r] < 0) { std::cerr << "fork failed\n"; gbp_comm_region_abort(region); fork_failed = true; break; }
    if (pid[r] == 0) {
      rk.rank = r; rk.world = world; rk.region = region;
      if (r != 0 && !std::freopen("/dev/null", "w", stdout)) std::_Exit(1);     // rank 0 prints (ba.cpp:996,1026-1028)
      int rc = 1;
      try { rc = body(rk); } catch (const std::exception& e) { std::cerr << "rank " << r << ": " << e.what() << "\n"; }
      std::cout.flush();
      std::fflush(nullptr);
      std::_Exit(rc);
    }
  }
  int status = 0, left = 0, first_bad = fork_failed ? 1 : 0;   // a run that could not start all its ranks never exits 0
  for (pid_t p : pid) left += p > 0;
  auto code_of = [](int st) { return WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0); };
  auto reap = [&](pid_t p, int st) {
    for (pid_t& q : pid) if (q == p) q = -1;
    --left;
    if (code_of(st) != 0 && first_bad == 0) {
      first_bad = code_of(st);
      gbp_comm_region_abort(region);            // the others leave their barriers with an error instead of waiting
    }
  };
  while (left > 0 && first_bad == 0) {
    const pid_t p = wait(&status);
    if (p < 0) {
      if (errno == EINTR) continue;            // a signal is not a failed rank
      break;
    }
    reap(p, status);
  }
  // a rank failed: the others normally notice at their next barrier; one that is stuck inside a collective whose peer
  // is gone never will, so after a grace period the remaining ranks are killed
  for (int waited_ms = 0; left > 0; waited_ms += 20) {
    const pid_t p = waitpid(-1, &status, WNOHANG);
    if (p > 0) { reap(p, status); continue; }
    if (p < 0) break;
    if (waited_ms >= 5000) {
      for (pid_t q : pid) if (q > 0) kill(q, SIGKILL);
      while (left > 0 && (wait(&status)) > 0) --left;
      break;
    }
    usleep(20000);
  }
  munmap(region, bytes);
  return first_bad;
}

// The reference's loop prints the metric after every iteration (ba.cpp:1009-1028).  To keep that output without making
// the GPU wait for the host every iteration, the metric of iteration i is queued (gbp_eval_begin) and printed only after
// iteration i+1 has been queued too; lines that belong after it ("Weakening priors") are deferred with it, so stdout is
// byte-identical to the unpipelined loop.  Off for multi-rank runs (the metric needs a gather) and for --v.
struct MetricPipe {
  gbp_ctx* ctx = nullptr;
  bool on = false, pending = false;
  std::function<void(const gbp_eval_out&)> printer;
  std::string deferred;
  void line(const std::string& s) { if (pending) deferred += s; else std::cout << s; }
  int flush() {
    if (!pending) return GBP_OK;
    gbp_eval_out ev{};
    const int rc = gbp_eval_end(ctx, &ev);
    pending = false;
    if (rc != GBP_OK) return rc;
    printer(ev);
    std::cout << deferred;
    deferred.clear();
    return GBP_OK;
  }
  // metric of the current beliefs, printed by `p` — now (pipe off) or once the next submit / flush comes.
  // iterate_first > 0: that many GBP iterations are issued first, in the SAME call as the metric (gbp_iterate_eval: one
  // launch for both on graphs that run in the persistent kernel).
  int submit(std::function<void(const gbp_eval_out&)> p, int iterate_first = 0) {
    if (!on) {
      if (iterate_first > 0) { const int rc = gbp_iterate(ctx, iterate_first); if (rc != GBP_OK) return rc; }
      gbp_eval_out ev{};
      const int rc = gbp_eval_global(ctx, &ev);
      if (rc == GBP_OK) p(ev);
      return rc;
    }
    int rc = iterate_first > 0 ? gbp_iterate_eval(ctx, iterate_first) : gbp_eval_begin(ctx);   // the metric is queued behind the iterations
    if (rc != GBP_OK) return rc;
    if (pending) {                     // the previous metric: its kernels finished long ago
      gbp_eval_out ev{};
      rc = gbp_eval_end(ctx, &ev);
      if (rc != GBP_OK) return rc;
      printer(ev);
      std::cout << deferred;
      deferred.clear();
    }
    printer = std::move(p);
    pending = true;
    return GBP_OK;
  }
};

// The metric lines of the reference's default loop (one per iteration: 1 500 - 13 299 of them) are formatted and written by ONE
// writer thread, in order, while the main thread has already issued the next burst: iostream formatting costs ~1.3 us a line,
// time the GPU would otherwise sit idle between two launches (ba fr1xyz: 2 ms of a 22 ms loop).  Everything else the loop prints
// (rare: "Weakening priors", keyframe banners, --v dumps) first waits for the writer (drain), so stdout stays byte-identical.
struct AsyncLines {
  std::mutex mu;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool done = false, busy = false;
  std::thread th;
  AsyncLines() : th([this] { run(); }) {}
  ~AsyncLines() { finish(); }
  void post(std::function<void()> f) {
    { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); }
    cv.notify_one();
  }
  void drain() {
    std::unique_lock<std::mutex> lk(mu);
    idle.wait(lk, [&] { return q.empty() && !busy; });
  }
  void finish() {
    if (!th.joinable()) return;
    { std::lock_guard<std::mutex> lk(mu); done = true; }
    cv.notify_one();
    th.join();
  }
  void run() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done || !q.empty(); });
        if (q.empty()) return;
        f = std::move(q.front());
        q.pop_front();
        busy = true;
      }
      f();
      { std::lock_guard<std::mutex> lk(mu); busy = false; }
      idle.notify_all();
    }
  }
};

#define CLI_CHECK(ctx, call)                                                              \
  do {                                                                                    \
    const int rc_ = (call);                                                               \
    if (rc_ != GBP_OK) {                                                                  \
      std::cerr << #call << " failed (" << rc_ << "): " << gbp_last_error(ctx) << "\n";  \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

// a call that returned GBP_OK can leave a `warning: ...` in gbp_last_error (a recovered time-out of the persistent kernel)
inline void print_warning(gbp_ctx* ctx) {
  const char* w = gbp_last_error(ctx);
  if (w && std::strncmp(w, "warning:", 8) == 0) std::cerr << w << "\n";
}

// --profile (ba.cpp:410-413,1060-1082 dump Poplar's execution / graph profile): here a one-line JSON report with the times of
// the run, which iteration path the library chose and the LAST printed metric at full precision (stdout carries six digits).
// bench.py reads it for the fr1xyz / fr2robot2 halves of BASELINE.json's metric.
struct RunReport {
  double wall_s = 0, setup_s = 0, loop_s = 0;
  long iters = 0;
  bool have_metric = false;
  gbp_eval_out last{};
};

// (called after the ctx is gone — its teardown is one of the phases — with what was read from it before)
inline void write_profile(const gbp_timing_out& t, int graph_state, const char* tool, const RunReport& r) {
  const char* dir = std::getenv("GC_PROFILE_LOG_DIR");
  const std::string path = std::string(dir ? dir : ".") + "/gbp_profile.json";
  if (FILE* f = std::fopen(path.c_str(), "w")) {
    std::fprintf(f, "{\"tool\": \"%s\", \"iterations\": %ld, \"wall_s\": %.6f, \"setup_s\": %.6f, \"loop_s\": %.6f, \"device_ms\": %.3f, "
                    "\"device_iterations\": %llu, \"iters_per_s_device\": %.3f, \"graph_state\": %d, "
                    "\"algorithmic_bytes_per_iter\": %llu, \"device_bytes_allocated\": %llu",
                 tool, r.iters, r.wall_s, r.setup_s, r.loop_s, t.total_ms, (unsigned long long)t.iterations,
                 t.total_ms > 0 ? 1e3 * (double)t.iterations / t.total_ms : 0.0, graph_state,
                 (unsigned long long)t.algorithmic_bytes_per_iter, (unsigned long long)t.device_bytes_allocated);
    if (r.have_metric && r.last.n_active) {
      const double n = (double)r.last.n_active;
      std::fprintf(f, ", \"final_mean_reproj_px\": %.9g, \"final_cost\": %.9g, \"final_rmse_px\": %.9g, \"n_active\": %llu, "
                      "\"n_relin\": %llu, \"n_robust\": %llu, \"n_nonfinite\": %llu",
                   r.last.sum_norm / n, r.last.sum_half_sq, std::sqrt(2.0 * r.last.sum_half_sq / n), (unsigned long long)r.last.n_active,
                   (unsigned long long)r.last.n_relin, (unsigned long long)r.last.n_robust, (unsigned long long)r.last.n_nonfinite);
    }
    {  // where the wall time of the PROCESS went
      const Phases& ph = phases();
      std::fprintf(f, ", \"startup\": {\"process_s\": %.6f, \"exec_to_main_s\": %.3f", ph.since_exec(), ph.exec_to_main_s);
      for (const auto& p : ph.v) std::fprintf(f, ", \"%s\": %.6f", p.first.c_str(), p.second);
      std::string info = ph.create_info;
      for (char& ch : info) if (ch == '"' || ch == '\\') ch = '\'';
      std::fprintf(f, ", \"create\": \"%s\"}", info.c_str());
    }
    std::fprintf(f, "}\n");
    std::fclose(f);
    std::cout << "Profile written to " << path << "\n";
  }
}

// The end of a run, shared by ./ba and ./slam: the ctx is destroyed FIRST (its teardown is part of what the user waits for), then the
// "Total time" line — which covers the process from exec to here — and the --profile report.
inline int finish_run(const Options& o, const Problem& P, gbp_ctx* ctx, const RankCtx& rk, const char* tool, RunReport& rep, long iters) {
  gbp_timing_out tm{};
  gbp_timing(ctx, &tm, 0);
  const int gs = gbp_graph_state(ctx);
  print_warning(ctx);
  const int wrc = write_solution(o, P, ctx, rk.region != nullptr);
  phases().mark("loop_s");
  gbp_destroy(ctx);
  phases().mark("teardown_s");
  const Phases& ph = phases();
  rep.loop_s = ph.get("loop_s");
  rep.setup_s = ph.get("upload_s") + ph.get("linearise_s");
  rep.wall_s = rep.setup_s + rep.loop_s;
  rep.iters = iters;
  std::cout << "Total time: " << ph.since_exec() << " s from exec (loader " << ph.exec_to_main_s << " s, file + priors " << ph.get("file_parse_s")
            << " s, runtime " << ph.get("runtime_init_s") << " s, gbp_create " << ph.get("create_s") << " s, upload " << ph.get("upload_s")
            << " s, linearise + first metric " << ph.get("linearise_s") << " s, iteration loop " << rep.loop_s << " s, teardown " << ph.get("teardown_s")
            << " s); device time in GBP iterations: " << tm.total_ms << " ms over " << tm.iterations
            << " iterations (" << (tm.total_ms > 0 ? 1e3 * (double)tm.iterations / tm.total_ms : 0.0) << " iters/s)\n";
  if (o.profile) write_profile(tm, gs, tool, rep);
  return wrc;
}

// --steps as gbp_ba_loop takes it: a whole, non-negative number below 2^30 (the loop's test is `iter < 2 * steps`: the product must not
// wrap, and a float beyond UINT_MAX must not be cast at all) — anything else keeps the loop's own float test, call by call
inline bool whole_steps(float steps) { return steps >= 0.f && steps < 1073741824.f && steps == (float)(unsigned)steps; }

// The last step of main(): everything is written, the ctx is destroyed.  Normally the process then leaves WITHOUT running exit handlers
// (std::_Exit) — the HIP runtime's take 15 - 30 ms in which the user's prompt does not come back, and the driver reclaims the process's
// resources either way.  Under an injected library (LD_PRELOAD: rocprofv3, sanitizers — their reports are written by exit handlers) or
// with GBP_CLI_FULL_EXIT set, the ordinary return.
inline int leave(int rc) {
  if (std::getenv("GBP_HOST_TRACE")) {      // what the kernel will have to tear down: the address space as it stands (profiles/r06_exit_probe.txt)
    size_t vmas = 0;
    if (FILE* f = std::fopen("/proc/self/maps", "r")) { int ch; while ((ch = std::fgetc(f)) != EOF) vmas += ch == '\n'; std::fclose(f); }
    std::fprintf(stderr, "leave: %zu mappings;", vmas);
    for (const char* path : {"/proc/self/status", "/proc/self/smaps_rollup"})
      if (FILE* f = std::fopen(path, "r")) {
        char line[256];
        while (std::fgets(line, sizeof(line), f))
          for (const char* key : {"VmRSS:", "VmPTE:", "RssAnon:", "RssFile:", "RssShmem:", "Threads:", "AnonHugePages:", "ShmemPmdMapped:", "FilePmdMapped:", "Locked:", "VmLck:", "VmPin:"})
            if (!std::strncmp(line, key, std::strlen(key))) { line[std::strcspn(line, "\n")] = 0; std::fprintf(stderr, " %s", line); }
        std::fclose(f);
      }
    std::fprintf(stderr, "\n");
  }
  std::cout.flush();
  std::cerr.flush();
  std::fflush(nullptr);
  if (std::getenv("LD_PRELOAD") || std::getenv("GBP_CLI_FULL_EXIT") || std::getenv("ROCP_TOOL_LIBRARIES")) return rc;
  std::_Exit(rc);
}

}  // namespace cli
