/* gbp_mi355x_compat.h — earlier forms of the iteration loop with the metric, kept so that hosts written against ABI 2-5 keep
 * linking.  New code does not need them: gbp_mi355x.h has the two calls they reduce to.
 *
 *   gbp_iterate_eval_each(ctx, n, out)              ==  gbp_ba_loop(ctx, n, 0, 0, out)        (steps = 0: no pass weakens priors)
 *   gbp_iterate_eval(ctx, n); gbp_eval_end(ctx, o)  ==  gbp_iterate(ctx, n); gbp_eval(ctx, o) (non-blocking until gbp_eval_end;
 *                                                       on a graph that runs in the persistent kernel the metric rides in the launch)
 *   gbp_eval_begin(ctx); gbp_eval_end(ctx, o)       ==  gbp_eval(ctx, o)                      (in two halves, up to two in flight)
 * Identical results in every case (tests/test_gpu_parity.py compares them bit for bit).
 */
#ifndef GBP_MI355X_COMPAT_H
#define GBP_MI355X_COMPAT_H

#include "gbp_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* gbp_eval in two halves: begin queues the metric of the CURRENT beliefs, end waits for the oldest queued one.  Up to two
 * may be in flight, so the loop of ba.cpp:1001-1028 can queue iteration i+1 before it prints the metric of iteration i. */
GBP_API int gbp_eval_begin(gbp_ctx* ctx);
GBP_API int gbp_eval_end(gbp_ctx* ctx, gbp_eval_out* out);
/* gbp_iterate(n) + gbp_eval_begin() in ONE call (collect with gbp_eval_end): the loop of ba.cpp:1001-1028 prints the metric
 * after every iteration; on a graph that runs in the persistent kernel the metric then rides in the same launch (identical
 * results), elsewhere it is exactly the two calls. */
GBP_API int gbp_iterate_eval(gbp_ctx* ctx, int n_iters);
/* n iterations with the metric after EVERY one (what the loops of ba.cpp:1001-1028 and slam.cpp print), blocking: out[k] is
 * what gbp_iterate(ctx, 1) followed by gbp_eval would have returned for the k-th of them.  A burst between two host events
 * (prior weakening, a new keyframe) is ONE launch on a graph that runs in the persistent kernel — the metric of iteration k
 * is computed inside the sweep phase of iteration k + 1 — and the plain loop elsewhere.  No evaluation may be in flight. */
GBP_API int gbp_iterate_eval_each(gbp_ctx* ctx, int n_iters, gbp_eval_out* out /* [n_iters] */);

#ifdef __cplusplus
}
#endif
#endif
