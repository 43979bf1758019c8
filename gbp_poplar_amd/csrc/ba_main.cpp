// ba_main.cpp — `./ba --bal_file F`: batch bundle adjustment by synchronous GBP on one MI355X.
// Same flow, flags and stdout lines as the reference's ba/ba.cpp main() (479-1085), with the Poplar
// engine replaced by the C-ABI: WRITE -> LINEARISE -> READ -> loop {WEAKEN_PRIORS?, GBP, READ, eval}.
#include "cli_common.hpp"

static int run(const cli::Options& o, cli::Problem& P, cli::RankCtx& rk) {
  const uint32_t C = P.bal.n_cams, L = P.bal.n_lmks, E = P.bal.n_edges;

  std::cout << "\nBundle Adjustment\n";                       // ba.cpp:587-590
  P.active.assign(E, 1u);
  P.cwf.assign(C, (uint32_t)o.steps);
  P.lwf.assign(L, (uint32_t)o.steps);
  std::cout << "\nNumber of keyframe nodes in the graph: " << C << '\n';
  std::cout << "Number of landmark nodes in the graph: " << L << '\n';
  std::cout << "Number of edges in the graph: " << E << '\n';
  std::cout << "\nNumber of GPUs: " << rk.world << "\n\nAttaching to GPU device..." << std::endl;

  gbp_ctx* ctx = nullptr;
  if (const int rc = cli::create_rank_ctx(o, P, rk, &ctx)) return rc;
  std::cout << "Running program to stream initial data to GPU\n";
  const gbp_state_in in = cli::state_in(P);
  CLI_CHECK(ctx, gbp_upload(ctx, &in));
  cli::phases().mark("upload_s");
  std::cout << "Initial data streaming complete\n\n";
  std::cout << "Sending priors and computing factor potentials.\n";
  CLI_CHECK(ctx, gbp_linearise(ctx));

  cli::Readback rb(C, L);
  gbp_eval_out ev{};
  CLI_CHECK(ctx, gbp_eval_global(ctx, &ev));
  std::cout << "Initial Reprojection error: " << (float)(ev.sum_norm / (double)ev.n_active) << " Cost " << (float)ev.sum_half_sq << "\n";
  std::cout << "Number of iterations: " << o.n_iters << "\n";

  cli::phases().mark("linearise_s");      // LINEARISE_PROG + the first metric
  cli::MetricPipe pipe;
  pipe.ctx = ctx;
  pipe.on = !rk.region && !o.verbose;
  unsigned iter = 0;
  cli::RunReport rep;
  const auto write_iter = [](unsigned it_now, const gbp_eval_out& e) {   // ba.cpp:1020-1024
    std::cout << "Iter " << it_now << " // Reprojection error " << (float)(e.sum_norm / (double)e.n_active);
    std::cout << " // Cost " << (float)e.sum_half_sq << " // n relins: " << e.n_relin;
    std::cout << " // n robust edges " << e.n_robust << "\n";
    if (e.n_nonfinite) std::cout << "warning: " << e.n_nonfinite << " beliefs are non-finite\n";
  };
  const auto print_iter = [&rep, &write_iter](unsigned it_now, const gbp_eval_out& e) {
    rep.last = e; rep.have_metric = true;
    write_iter(it_now, e);
  };
  cli::AsyncLines lines;      // cli_common.hpp: the per-iteration lines are written while the next burst runs
  std::vector<gbp_eval_out> series;
  // the reference's default (the metric after EVERY iteration) with a whole number of --steps: the loop's body goes down as
  // gbp_ba_loop — prior weakening, iteration and metric of many passes in one call (ONE launch on a graph that runs in the persistent
  // kernel, which weakens the priors itself); the lines are written from the results, "Weakening priors" where the loop weakens
  const bool whole_loop = pipe.on && o.eval_every == 1 && cli::whole_steps(o.steps);
  for (int i = 0; i < o.n_iters; ++i) {
    if (whole_loop) {
      const int cap = gbp_graph_state(ctx) == 2 ? 512 : 128;      // (so that the lines keep coming on a large graph)
      int burst = std::min(cap, o.n_iters - i);
      if (cap == 512 && i + burst == o.n_iters && burst > 96) burst -= 64;      // the run ends with a short burst: its lines are the ones no launch overlaps
      series.resize((size_t)burst);
      CLI_CHECK(ctx, gbp_ba_loop(ctx, burst, iter, (unsigned)o.steps, series.data()));
      rep.last = series.back(); rep.have_metric = true;
      lines.post([first = iter, steps2 = 2u * (unsigned)o.steps, batch = series, &write_iter] {
        for (size_t k = 0; k < batch.size(); ++k) {
          const unsigned it_now = first + (unsigned)k;
          if ((it_now + 1) % 2 == 0 && it_now < steps2) std::cout << "Weakening priors \n";       // ba.cpp:1003-1006
          write_iter(it_now, batch[k]);
        }
      });
      i += burst - 1;
      iter += (unsigned)burst;
      continue;
    }
    if (o.eval_every > 1 && !o.verbose && cli::whole_steps(o.steps)) {
      // --eval_every N: everything up to the next metric in one call, the prior weakenings inside included (gbp_ba_loop without the
      // metric); the pass the metric follows goes down with it (pipe.submit: gbp_iterate_eval)
      const auto weak = [&](unsigned it_) { return (it_ + 1) % 2 == 0 && it_ < 2u * (unsigned)o.steps; };
      int burst = 1;
      while (i + burst < o.n_iters && (i + burst) % o.eval_every != 0) ++burst;
      const bool eval_now = (i + burst) % o.eval_every == 0 || i + burst == o.n_iters;
      for (int k = 0; k < burst; ++k)
        if (weak(iter + (unsigned)k)) pipe.line("Weakening priors \n");
      const int head = eval_now ? burst - 1 : burst;
      if (head > 0) CLI_CHECK(ctx, gbp_ba_loop(ctx, head, iter, (unsigned)o.steps, nullptr));
      if (eval_now) {
        const unsigned it_now = iter + (unsigned)burst - 1u;
        if (weak(it_now)) CLI_CHECK(ctx, gbp_weaken_priors(ctx));
        CLI_CHECK(ctx, pipe.submit([it_now, &print_iter](const gbp_eval_out& e) { print_iter(it_now, e); }, 1));
      }
      i += burst - 1;
      iter += (unsigned)burst;
      continue;
    }
    if (((iter + 1) % 2 == 0) && (iter < o.steps * 2)) {       // ba.cpp:1003-1006
      if (pipe.on && o.eval_every == 1) lines.post([] { std::cout << "Weakening priors \n"; });      // in order, behind the lines of the burst before it
      else pipe.line("Weakening priors \n");
      CLI_CHECK(ctx, gbp_weaken_priors(ctx));
    }
    if (pipe.on && o.eval_every == 1) {
      // the reference's default: the metric after EVERY iteration.  All iterations up to the next prior weakening go down in
      // one call (gbp_ba_loop: one launch on a graph that runs in the persistent kernel), at most 128 at a time so
      // that the lines keep coming on a large graph — 512 where an iteration takes microseconds (the persistent kernel: every
      // launch boundary is ~60 us of idle GPU), but the LAST burst of the run short: its lines are the ones no launch overlaps
      const int cap = gbp_graph_state(ctx) == 2 ? 512 : 128;
      int burst = 1;
      while (burst < cap && i + burst < o.n_iters && !(((iter + burst + 1) % 2 == 0) && (iter + burst < o.steps * 2))) ++burst;
      if (cap == 512 && i + burst == o.n_iters && burst > 96) burst -= 64;      // ... so the run ends with a burst of 64
      series.resize((size_t)burst);
      CLI_CHECK(ctx, gbp_ba_loop(ctx, burst, iter, 0u, series.data()));      // steps = 0: the weakenings are this loop's own calls
      rep.last = series.back(); rep.have_metric = true;
      lines.post([first = iter, batch = series, &write_iter] {
        for (size_t k = 0; k < batch.size(); ++k) write_iter(first + (unsigned)k, batch[k]);
      });
      i += burst - 1;
      iter += (unsigned)burst;
      continue;
    }
    // run up to the next host event (prior weakening or read-back) in one call: gbp_iterate(k) replays
    // the captured hipGraph, so with --eval_every > 1 the loop never leaves the device in between
    int burst = 1;
    while (i + burst < o.n_iters && (i + burst) % o.eval_every != 0 &&
           !(((iter + burst + 1) % 2 == 0) && (iter + burst < o.steps * 2)))
      ++burst;
    const bool eval_now = (i + burst) % o.eval_every == 0 || i + burst == o.n_iters;
    if (!eval_now) CLI_CHECK(ctx, gbp_iterate(ctx, burst));
    i += burst - 1;
    iter += burst - 1;
    if (eval_now) {
      const unsigned it_now = iter;
      CLI_CHECK(ctx, pipe.submit([it_now, &print_iter](const gbp_eval_out& e) { print_iter(it_now, e); }, burst));   // the burst and its metric in one call
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
  return cli::finish_run(o, P, ctx, rk, "ba", rep, o.n_iters);
}

int main(int argc, char** argv) {
  cli::Options o;
  const int pr = cli::parse(argc, argv, /*slam=*/false, o);
  if (pr) return pr == 1 ? 0 : 1;
  cli::Problem P;
  const bool one_process = cli::round_up_pow2(std::max(1, o.gpus)) == 1 && !o.force_sharded;
  if (one_process) cli::prime_address_space();      // (an empirical 40 - 60 ms off the process's exit: see its comment)
  cli::runtime_warmup().start(one_process);          // the HIP runtime comes up beside the file's parsing
  cli::phases().mark("parse_args_s");
  if (cli::load_problem(o, P)) return 1;          // host only: the ranks are forked before anything touches HIP
  cli::phases().mark("file_parse_s");             // the file, the priors, the scalings
  const int world = cli::round_up_pow2(std::max(1, o.gpus));   // ba.cpp:617-621
  const int rc = cli::run_ranks(world, P.bal.n_cams, o.force_sharded, [&](cli::RankCtx& rk) { return run(o, P, rk); });
  // Everything is written and flushed, the ctx is destroyed: leave WITHOUT running the HIP runtime's exit handlers (70 - 90 ms during
  // which the user's prompt does not come back: a fifth of a `ba fr1xyz` run; the driver reclaims the process's resources either way)
  return cli::leave(rc);
}
