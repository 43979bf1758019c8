// slam_main.cpp — `./slam --bal_file F`: incremental SLAM, one new keyframe every --iters_between_kfs
// GBP iterations.  Same flow as the reference's ba/slam.cpp main() (479-1135): factors of cameras 0,1
// start active, then per keyframe update_flags -> READ_PRIORS -> initialise_new_kf -> re-arm
// damping_count to -15 -> NEW_KEYFRAME, with the usual {WEAKEN_PRIORS?, GBP, READ, eval} body.
#include "cli_common.hpp"

static int run(const cli::Options& o, cli::Problem& P, cli::RankCtx& rk) {
  const uint32_t C = P.bal.n_cams, L = P.bal.n_lmks, E = P.bal.n_edges;

  std::cout << "SLAM\n";                                        // slam.cpp:586-595
  const unsigned steps = static_cast<unsigned>(o.steps);
  P.active.assign(E, 0u); P.cwf.assign(C, 0u); P.lwf.assign(L, 0u);
  std::vector<uint32_t> lmk_active(L, 0u);
  gbp_slam_create_flags(&P.prob, steps, P.active.data(), P.cwf.data(), P.lwf.data(), lmk_active.data());
  std::cout << "\nNumber of keyframe nodes in the graph: " << C << '\n';
  std::cout << "Number of landmark nodes in the graph: " << L << '\n';
  std::cout << "Number of edges in the graph: " << E << '\n';
  std::cout << "\nNumber of GPUs: " << rk.world << "\n\nAttaching to GPU device..." << std::endl;

  gbp_ctx* ctx = nullptr;
  if (const int rc = cli::create_rank_ctx(o, P, rk, &ctx)) return rc;
  const gbp_state_in in = cli::state_in(P);
  CLI_CHECK(ctx, gbp_upload(ctx, &in));
  cli::phases().mark("upload_s");
  std::cout << "Sending priors and computing factor potentials.\n";
  CLI_CHECK(ctx, gbp_linearise(ctx));

  cli::Readback rb(C, L);
  gbp_priors_out po{};
  po.cam_priors_eta = P.cpe.data(); po.cam_priors_lambda = P.cpl.data();
  po.lmk_priors_eta = P.lpe.data(); po.lmk_priors_lambda = P.lpl.data();
  gbp_eval_out ev{};
  CLI_CHECK(ctx, gbp_eval_global(ctx, &ev));
  std::cout << "Initial Reprojection error: " << (float)(ev.sum_norm / (double)ev.n_active) << " Cost " << (float)ev.sum_half_sq << "\n";

  const unsigned niters = (C - 1) * (unsigned)o.iters_between_kfs - 1;   // slam.cpp:1013
  unsigned iter = 0, data_counter = 0;
  std::cout << "Total number of GBP iterations: " << niters << "\n";
  std::cout << "GBP iterations between sucessive keyframes: " << o.iters_between_kfs << "\n";
  cli::phases().mark("linearise_s");      // LINEARISE_PROG + the first metric
  cli::MetricPipe pipe;
  pipe.ctx = ctx;
  pipe.on = !rk.region && !o.verbose;
  cli::RunReport rep;
  const auto write_iter = [](unsigned total, unsigned since, const gbp_eval_out& e) {
    std::cout << "Iters " << total;
    std::cout << " (since last kf " << since << ") // Reprojection error " << (float)(e.sum_norm / (double)e.n_active);
    std::cout << " // Cost " << (float)e.sum_half_sq << " // n relins: " << e.n_relin;
    std::cout << " // n robust edges " << e.n_robust << "\n";
  };
  const auto print_iter = [&rep, &write_iter](unsigned total, unsigned since, const gbp_eval_out& e) {
    rep.last = e; rep.have_metric = true;
    write_iter(total, since, e);
  };
  cli::AsyncLines lines;      // cli_common.hpp: the per-iteration lines are written while the next burst runs
  std::vector<gbp_eval_out> series;
  for (unsigned i = 0; i < niters; ++i) {
    if ((i + 1) % (unsigned)o.iters_between_kfs == 0) {         // slam.cpp:1020-1046
      CLI_CHECK(ctx, pipe.flush());                             // the keyframe logic reads beliefs back: no metric in flight
      iter = 0;
      data_counter += 1;
      int32_t n_new = 0;
      gbp_slam_update_flags(&P.prob, steps, data_counter, P.active.data(), P.lwf.data(), P.cwf.data(), lmk_active.data(), &n_new);
      const auto banner = [kf = data_counter + 1, n_new] {
        std::cout << "\n**********************************************************";
        std::cout << "\n Adding keyframe " << kf;
        std::cout << "\n Adding " << n_new << " new landmarks";
        std::cout << "\n**********************************************************\n\n";
      };
      if (pipe.on && o.eval_every == 1) lines.post(banner);     // through the writer: in order behind the lines of the burst before it
      else banner();
      CLI_CHECK(ctx, gbp_read_priors(ctx, &po));
      CLI_CHECK(ctx, gbp_read(ctx, &rb.out));
      gbp_slam_initialise_new_kf(data_counter, rb.cbe.data(), rb.cbl.data(), P.cpl.data(), P.cpe.data());
      std::fill(P.count.begin(), P.count.end(), -15);           // literal, slam.cpp:1039-1041
      gbp_kf_update up{};
      up.damping_count = P.count.data();
      up.cam_priors_eta = P.cpe.data(); up.cam_priors_lambda = P.cpl.data();
      up.lmk_priors_eta = P.lpe.data(); up.lmk_priors_lambda = P.lpl.data();
      up.active_flag = P.active.data(); up.cam_weaken_flag = P.cwf.data(); up.lmk_weaken_flag = P.lwf.data();
      CLI_CHECK(ctx, gbp_new_keyframe(ctx, &up));
    }
    if (pipe.on && o.eval_every == 1 && cli::whole_steps(o.steps)) {
      // the loop's body up to the next keyframe as gbp_ba_loop (see ba_main.cpp): prior weakening, iteration, metric — one call
      const unsigned cap = gbp_graph_state(ctx) == 2 ? 512u : 128u;
      unsigned nb = 1;
      while (nb < cap && i + nb < niters && (i + nb + 1) % (unsigned)o.iters_between_kfs != 0) ++nb;
      if (cap == 512u && i + nb == niters && nb > 96u) nb -= 64u;
      series.resize(nb);
      CLI_CHECK(ctx, gbp_ba_loop(ctx, (int)nb, iter, (unsigned)o.steps, series.data()));
      rep.last = series.back(); rep.have_metric = true;
      lines.post([total0 = (unsigned)o.iters_between_kfs * data_counter + iter, since0 = iter, steps2 = 2u * (unsigned)o.steps, batch = series, &write_iter] {
        for (size_t k = 0; k < batch.size(); ++k) {
          const unsigned since = since0 + (unsigned)k;
          if ((since + 1) % 2 == 0 && since < steps2) std::cout << "Weakening priors \n";
          write_iter(total0 + (unsigned)k, since, batch[k]);
        }
      });
      i += nb - 1;
      iter += nb;
      continue;
    }
    if (o.eval_every > 1 && !o.verbose && cli::whole_steps(o.steps)) {
      // --eval_every N: everything up to the next metric / keyframe in one call, the prior weakenings inside included (see ba_main.cpp)
      const auto weak = [&](unsigned it_) { return (it_ + 1) % 2 == 0 && it_ < 2u * (unsigned)o.steps; };
      unsigned burst = 1;
      while (i + burst < niters && (i + burst + 1) % (unsigned)o.iters_between_kfs != 0 && (i + burst) % (unsigned)o.eval_every != 0) ++burst;
      const bool eval_now = (i + burst) % (unsigned)o.eval_every == 0 || i + burst == niters;
      for (unsigned k = 0; k < burst; ++k)
        if (weak(iter + k)) pipe.line("Weakening priors \n");
      const unsigned head = eval_now ? burst - 1 : burst;
      if (head > 0) CLI_CHECK(ctx, gbp_ba_loop(ctx, (int)head, iter, (unsigned)o.steps, nullptr));
      if (eval_now) {
        const unsigned since = iter + burst - 1u, total = (unsigned)o.iters_between_kfs * data_counter + since;
        if (weak(since)) CLI_CHECK(ctx, gbp_weaken_priors(ctx));
        CLI_CHECK(ctx, pipe.submit([total, since, &print_iter](const gbp_eval_out& e) { print_iter(total, since, e); }, 1));
      }
      i += burst - 1;
      iter += burst;
      continue;
    }
    if (((iter + 1) % 2 == 0) && (iter < o.steps * 2)) {
      if (pipe.on && o.eval_every == 1) lines.post([] { std::cout << "Weakening priors \n"; });
      else { lines.drain(); pipe.line("Weakening priors \n"); }
      CLI_CHECK(ctx, gbp_weaken_priors(ctx));
    }
    if (pipe.on && o.eval_every == 1) {
      // the reference's default, the metric after EVERY iteration: everything up to the next keyframe / prior weakening in
      // one call (gbp_ba_loop, see ba_main.cpp)
      const unsigned cap = gbp_graph_state(ctx) == 2 ? 512u : 128u;      // (see ba_main.cpp)
      unsigned nb = 1;
      while (nb < cap && i + nb < niters && (i + nb + 1) % (unsigned)o.iters_between_kfs != 0 &&
             !(((iter + nb + 1) % 2 == 0) && (iter + nb < o.steps * 2)))
        ++nb;
      if (cap == 512u && i + nb == niters && nb > 96u) nb -= 64u;      // the run ends with a burst of 64 (its lines are the ones no launch overlaps)
      series.resize(nb);
      CLI_CHECK(ctx, gbp_ba_loop(ctx, (int)nb, iter, 0u, series.data()));      // steps = 0: the weakenings are this loop's own calls
      rep.last = series.back(); rep.have_metric = true;
      lines.post([total0 = (unsigned)o.iters_between_kfs * data_counter + iter, since0 = iter, batch = series, &write_iter] {
        for (size_t k = 0; k < batch.size(); ++k) write_iter(total0 + (unsigned)k, since0 + (unsigned)k, batch[k]);
      });
      i += nb - 1;
      iter += nb;
      continue;
    }
    // up to the next host event (keyframe, prior weakening, metric read-back) in one call, like ba_main.cpp
    unsigned burst = 1;
    while (i + burst < niters && (i + burst + 1) % (unsigned)o.iters_between_kfs != 0 && (i + burst) % (unsigned)o.eval_every != 0 &&
           !(((iter + burst + 1) % 2 == 0) && (iter + burst < o.steps * 2)))
      ++burst;
    const bool eval_now = (i + burst) % (unsigned)o.eval_every == 0 || i + burst == niters;
    if (!eval_now) CLI_CHECK(ctx, gbp_iterate(ctx, (int)burst));
    i += burst - 1;
    iter += burst - 1;
    if (eval_now) {
      const unsigned total = (unsigned)o.iters_between_kfs * data_counter + iter, since = iter;
      CLI_CHECK(ctx, pipe.submit([total, since, &print_iter](const gbp_eval_out& e) { print_iter(total, since, e); }, (int)burst));
      if (o.verbose) {
        CLI_CHECK(ctx, gbp_read(ctx, &rb.out));
        cli::print_verbose(rb);
      }
    }
    iter += 1;
  }
  lines.finish();
  CLI_CHECK(ctx, pipe.flush());
  std::cout << "\n Finished GBP.\n";
  return cli::finish_run(o, P, ctx, rk, "slam", rep, (long)niters);
}

int main(int argc, char** argv) {
  cli::Options o;
  const int pr = cli::parse(argc, argv, /*slam=*/true, o);
  if (pr) return pr == 1 ? 0 : 1;
  cli::Problem P;
  const bool one_process = cli::round_up_pow2(std::max(1, o.gpus)) == 1 && !o.force_sharded;
  if (one_process) cli::prime_address_space();      // (an empirical 40 - 60 ms off the process's exit: see its comment)
  cli::runtime_warmup().start(one_process);          // the HIP runtime comes up beside the file's parsing
  cli::phases().mark("parse_args_s");
  if (cli::load_problem(o, P)) return 1;          // host only: the ranks are forked before anything touches HIP
  cli::phases().mark("file_parse_s");             // the file, the priors, the scalings
  if (P.bal.n_cams < 2) { std::cerr << "slam needs at least two keyframes\n"; return 1; }
  const int world = cli::round_up_pow2(std::max(1, o.gpus));   // slam.cpp:422-425 / ba.cpp:617-621
  const int rc = cli::run_ranks(world, P.bal.n_cams, o.force_sharded, [&](cli::RankCtx& rk) { return run(o, P, rk); });
  // Everything is written and flushed, the ctx is destroyed: leave WITHOUT running the HIP runtime's exit handlers (70 - 90 ms during
  // which the user's prompt does not come back: a fifth of a `ba fr1xyz` run; the driver reclaims the process's resources either way)
  return cli::leave(rc);
}
