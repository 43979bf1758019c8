"""The split-phase C-ABI (gbp_iterate_begin / _local / _end) driven by a CALLER-OWNED collective — test harness and
example, not the product's multi-GPU path.

The product path is the library-owned exchange: `bin/ba --ipus N` / `bin/slam --ipus N` and bench.py fork / launch one
process per GPU, csrc/gbp_comm.cpp issues the RCCL all-gather inside gbp_iterate (ONE exchange
path).  This module stays because (a) the CPU suite exercises the sharding logic on 2 gloo ranks through it with an
oracle-backed engine (tests/test_distributed_gloo.py), (b) the GPU suite emulates N shards on one GPU with it
(dist=None: the "exchange" is a copy), (c) it documents what a caller with its own collective (torch.distributed, MPI)
has to do — examples/ba_torchrun.py.

Landmark-sharded multi-GPU GBP: one process per GPU, torch.distributed (RCCL over xGMI) for the
single exchange step of an iteration.

Replaces the reference's `--ipus N` (ba.cpp:414-417,617-623: one Poplar graph spread over N x 1216
tiles, exchange compiled by Poplar).  Here a rank owns a contiguous landmark range and every factor
incident to it; factor potentials, per-factor state, both message directions and landmark beliefs are
rank-local.  Cameras are replicated: per iteration each rank reduces its factor->camera messages to a
[C x 44] partial, ONE all-gather moves the partials, and every rank adds prior + partials in rank order
(deterministic, bit-identical camera beliefs on all ranks; no float atomics, no ring all-reduce whose
association order would differ between ranks).

`engine` is a rank-local object with the split-phase verbs of the C-ABI (gbp_iterate_begin/_end,
gbp_refresh_begin/_end, gbp_linearise_factors, ...): the product passes GbpEngine (HIP); the CPU gloo
tests pass an oracle-backed stand-in to exercise this host logic without a GPU.
"""
import numpy as np

CAM_REC = 44  # GBP_CAM_REC: eta 6, pad 2, Lambda 36


def landmark_partition(lmk_id, n_lmks, world):
    """Contiguous landmark ranges balanced by incident-factor count -> bounds[world+1]."""
    deg = np.bincount(np.asarray(lmk_id, dtype=np.int64), minlength=n_lmks).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(deg)])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r // world
        b = int(np.searchsorted(csum, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), n_lmks))
    bounds.append(n_lmks)
    return np.asarray(bounds, dtype=np.uint32)


class ShardedGbp:
    """The Poplar program list over `world` ranks.  Same verbs as GbpEngine / the oracle."""

    def __init__(self, engine, n_cams, rank, world, dist=None, device="cpu", always_collective=False, use_graph=False):
        import torch
        self.torch = torch
        self.e = engine
        self.C, self.rank, self.world = int(n_cams), int(rank), int(world)
        self.dist = dist
        self.always_collective = always_collective   # run the all-gather even for world == 1 (exercises RCCL in tests)
        self.use_graph = use_graph                   # capture kernels + collective of `graph_unroll` iterations
        self.graph_unroll = 10
        self.graph = None
        self.graph_error = None
        self.send = torch.zeros(self.C * CAM_REC, dtype=torch.float32, device=device)
        self.recv = torch.zeros(self.world * self.C * CAM_REC, dtype=torch.float32, device=device)
        self.stream = None
        if device != "cpu":
            # One dedicated (non-default) stream carries BOTH the engine's kernels and the collective, so
            # ordering is plain stream order: kernels -> all_gather (RCCL waits on / signals this stream) -> kernels.
            self.stream = torch.cuda.Stream()
            engine.set_stream(self.stream.cuda_stream)
            torch.cuda.synchronize()       # the zero-fills above ran on the default stream
        engine.set_exchange_buffers(self.send.data_ptr(), self.recv.data_ptr())

    def _on_stream(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _exchange(self):
        with self._on_stream():
            if self.dist is None or (self.world == 1 and not self.always_collective):
                self.recv.copy_(self.send)
            else:
                self.dist.all_gather_into_tensor(self.recv, self.send)

    def upload(self, state):
        self.e.upload(state)

    def linearise(self):
        """LINEARISE_PROG (ba.cpp:890-893): prog_ub needs the exchange, then relinearise local factors."""
        self.e.refresh_begin()
        self._exchange()
        self.e.refresh_end()
        self.e.linearise_factors()

    def _one_iteration(self, overlap):
        if overlap:
            # The collective runs on RCCL's stream, ordered after the sweep that filled `send`; the rank-local landmark half of
            # the belief update fills the GPU meanwhile; the camera combine waits for it.
            self.e.iterate_begin()
            with self._on_stream():
                work = self.dist.all_gather_into_tensor(self.recv, self.send, async_op=True)
            self.e.iterate_local()
            with self._on_stream():
                work.wait()
        else:
            self.e.iterate_begin()                      # sweep + local camera partials -> send
            self._exchange()
        self.e.iterate_end()

    def _capture(self, overlap):
        """Capture `graph_unroll` iterations (kernels of this engine AND the RCCL all-gather) into one hipGraph on
        the shared stream.  Returns False (and stays on direct launches) if the runtime refuses the capture."""
        torch = self.torch
        try:
            for _ in range(3):                       # RCCL must have run on this stream before a capture
                self._one_iteration(overlap)
            self.stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                for _ in range(self.graph_unroll):
                    self._one_iteration(overlap)
            self.graph = g
            return 3
        except Exception as exc:                     # noqa: BLE001 - any capture failure means "no graph"
            self.graph = None
            self.graph_error = repr(exc)
            self.use_graph = False
            try:
                self.stream.synchronize()
            except Exception:
                pass
            return None

    def iterate(self, n=1):
        n = int(n)
        overlap = (self.dist is not None and (self.world > 1 or self.always_collective) and self.stream is not None
                   and hasattr(self.e, "iterate_local"))
        if self.use_graph and self.stream is not None and self.dist is not None and n >= self.graph_unroll + 3:
            if self.graph is None:
                done = self._capture(overlap)
                if done is not None:
                    n -= done
            while self.graph is not None and n >= self.graph_unroll:
                with self._on_stream():
                    self.graph.replay()
                n -= self.graph_unroll
        for _ in range(n):
            self._one_iteration(overlap)

    def weaken_priors(self):
        self.e.weaken_priors()          # priors are replicated / local: no exchange (partials unchanged)

    def new_keyframe(self, upd):
        self.e.new_keyframe(upd)

    def read(self):
        return self.e.read()

    def read_priors(self):
        return self.e.read_priors()

    def eval(self):
        """Local shard sums, added over ranks (integers and fp64 partials; rank order fixed by all_gather)."""
        ev = self.e.eval()
        if self.dist is None or (self.world == 1 and not self.always_collective):
            return ev
        torch = self.torch
        keys = ["sum_norm", "sum_half_sq", "n_active", "n_relin", "n_robust", "n_nonfinite", "n_nonpd"]
        with self._on_stream():
            mine = torch.tensor([float(ev[k]) for k in keys], dtype=torch.float64, device=self.send.device)
            allv = torch.zeros(self.world * len(keys), dtype=torch.float64, device=self.send.device)
            self.dist.all_gather_into_tensor(allv, mine)
        if self.stream is not None:
            self.stream.synchronize()
        allv = allv.view(self.world, len(keys)).cpu().numpy()
        out = {}
        for j, k in enumerate(keys):
            acc = 0.0
            for r in range(self.world):
                acc = acc + float(allv[r, j])
            out[k] = acc if j < 2 else int(round(acc))
        return out

    def sync(self):
        if hasattr(self.e, "sync"):
            self.e.sync()
        if self.stream is not None:
            self.stream.synchronize()
