"""Helpers shared by tests and tests/golden/make_golden.py: which iterations of a run carry a recorded metric
(the first 30, every 50th / every keyframe, the last 50) and an engine wrapper that records them."""
import numpy as np

from gbp_poplar_amd import driver


def wanted(n_total, head=30, every=50, tail=50):
    w = set(range(min(head, n_total))) | set(range(every - 1, n_total, every)) | set(range(max(0, n_total - tail), n_total))
    return sorted(w)


class EvalAt:
    """Wraps anything with the program-list verbs (GbpEngine, ShardedGbp, the oracle) for driver.run_ba / run_slam
    with eval_every=0: the metric is evaluated only after the wanted iterations.
    rows: (iteration, mean reproj, cost, RMSE, n_relins, n_robust, n_active)."""

    def __init__(self, engine, wanted_iters):
        self.e, self.wanted, self.i, self.rows = engine, set(wanted_iters), 0, []
        for name in ("upload", "linearise", "weaken_priors", "read", "read_priors", "new_keyframe", "eval"):
            setattr(self, name, getattr(engine, name))

    def iterate(self, n=1):
        """A burst of n iterations, split only where a metric is wanted (so that the engine sees real bursts:
        hipGraph replay / k_persist on the GPU)."""
        left = int(n)
        while left > 0:
            k = 1
            while k < left and (self.i + k - 1) not in self.wanted:
                k += 1
            self.e.iterate(k)
            self.i += k
            left -= k
            if (self.i - 1) in self.wanted:
                ev = self.e.eval()
                m = driver.metric(ev)
                self.rows.append((self.i - 1, m[0], m[1], m[2], ev["n_relin"], ev["n_robust"], ev["n_active"]))

    def array(self):
        return np.array(self.rows, dtype=np.float64)
