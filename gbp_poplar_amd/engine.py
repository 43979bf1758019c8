"""GbpEngine — Python face of the device half of the C-ABI: the reference's Poplar program list
(ba.cpp:925-934, slam.cpp:937-948) as methods.  Every call goes through libgbp_mi355x.so (HIP);
construction raises if the library or a GPU is missing — there is no CPU path."""
import ctypes as C

import numpy as np

from . import _cabi as cabi
from ._lib import load


class GbpError(RuntimeError):
    pass


class GbpEngine:
    def __init__(self, cam_id, lmk_id, n_cams, n_lmks, K9, params=None, shard=None, hooks=False):
        """hooks=True loads libgbp_mi355x_test.so — the product sources + the gbp_debug_* test hooks
        (include/gbp_mi355x_debug.h) — instead of the product library; only tests and profiles/ ask for it."""
        self.hooks = hooks if hooks == "exp" else bool(hooks)     # "exp": the experiments build (has the hooks too)
        self.lib = load(hooks=self.hooks)
        self._keep = []
        self.problem = cabi.make_problem(cam_id, lmk_id, n_cams, n_lmks, K9, self._keep)
        self.C, self.L, self.E = int(n_cams), int(n_lmks), int(self.problem.n_edges)
        self.params = params if params is not None else cabi.GbpParams.defaults()
        self.shard = None
        if shard is not None:
            self.shard = cabi.GbpShard(int(shard[0]), int(shard[1]), int(shard[2]), int(shard[3]))
        h = C.c_void_p()
        rc = self.lib.gbp_create(C.byref(self.problem), C.byref(self.params),
                                 C.byref(self.shard) if self.shard is not None else None, C.byref(h))
        if rc != 0:
            raise GbpError("gbp_create: %s (status %d)" % (self.lib.gbp_last_error(None).decode(), rc))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.gbp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise GbpError("%s: %s (status %d)" % (what, self.lib.gbp_last_error(self.h).decode(), rc))

    def last_error(self):
        """Text of the last error — or of the last recovered incident (a call that returned 0 can leave a `warning: ...` here)."""
        return self.lib.gbp_last_error(self.h).decode()

    # ---- program list ----
    def upload(self, state):
        keep = []
        s = cabi.fill_struct(cabi.GbpStateIn(), state, keep)
        self._chk(self.lib.gbp_upload(self.h, C.byref(s)), "gbp_upload")

    def linearise(self):
        self._chk(self.lib.gbp_linearise(self.h), "gbp_linearise")

    def iterate(self, n=1):
        self._chk(self.lib.gbp_iterate(self.h, int(n)), "gbp_iterate")

    def prepare(self):
        """Pay the one-off costs of the multi-iteration path now (graph capture, instantiation, upload); runs nothing."""
        self._chk(self.lib.gbp_prepare(self.h), "gbp_prepare")

    def weaken_priors(self):
        self._chk(self.lib.gbp_weaken_priors(self.h), "gbp_weaken_priors")

    def read(self):
        out = {"cam_beliefs_eta": np.zeros(6 * self.C, np.float32),
               "cam_beliefs_lambda": np.zeros(36 * self.C, np.float32),
               "lmk_beliefs_eta": np.zeros(3 * self.L, np.float32),
               "lmk_beliefs_lambda": np.zeros(9 * self.L, np.float32),
               "damping": np.zeros(self.E, np.float32),
               "damping_count": np.zeros(self.E, np.int32),
               "robust_flag": np.zeros(self.E, np.uint32)}
        keep = []
        s = cabi.fill_struct(cabi.GbpStateOut(), out, keep)
        self._chk(self.lib.gbp_read(self.h, C.byref(s)), "gbp_read")
        return out

    def read_priors(self):
        out = {"cam_priors_eta": np.zeros(6 * self.C, np.float32),
               "cam_priors_lambda": np.zeros(36 * self.C, np.float32),
               "lmk_priors_eta": np.zeros(3 * self.L, np.float32),
               "lmk_priors_lambda": np.zeros(9 * self.L, np.float32)}
        keep = []
        s = cabi.fill_struct(cabi.GbpPriorsOut(), out, keep)
        self._chk(self.lib.gbp_read_priors(self.h, C.byref(s)), "gbp_read_priors")
        return out

    def new_keyframe(self, upd):
        keep = []
        s = cabi.fill_struct(cabi.GbpKfUpdate(), upd, keep)
        self._chk(self.lib.gbp_new_keyframe(self.h, C.byref(s)), "gbp_new_keyframe")

    def eval(self):
        o = cabi.GbpEvalOut()
        self._chk(self.lib.gbp_eval(self.h, C.byref(o)), "gbp_eval")
        return {k: getattr(o, k) for k, _ in o._fields_}

    def eval_begin(self):
        self._chk(self.lib.gbp_eval_begin(self.h), "gbp_eval_begin")

    def iterate_eval(self, n=1):
        """iterate(n) + eval_begin() in one call (fused into one launch on graphs that run in the persistent kernel)."""
        self._chk(self.lib.gbp_iterate_eval(self.h, int(n)), "gbp_iterate_eval")

    def iterate_eval_each(self, n):
        """n iterations with the metric after every one (blocking); one launch per burst on graphs that run in the
        persistent kernel.  Returns a list of n dicts like eval()."""
        n = int(n)
        arr = (cabi.GbpEvalOut * max(n, 1))()
        self._chk(self.lib.gbp_iterate_eval_each(self.h, n, arr), "gbp_iterate_eval_each")
        return [{k: getattr(arr[i], k) for k, _ in arr[i]._fields_} for i in range(n)]

    def ba_loop(self, n, iter0, steps, metrics=True):
        """n passes of the body of the reference's loop from loop index iter0 (prior weakening where the loop weakens, the iteration,
        the metric): gbp_ba_loop.  Returns a list of n dicts like eval()."""
        n = int(n)
        if not metrics:      # the passes without the metric: not blocking, returns nothing
            self._chk(self.lib.gbp_ba_loop(self.h, n, int(iter0), int(steps), None), "gbp_ba_loop")
            return None
        arr = (cabi.GbpEvalOut * max(n, 1))()
        self._chk(self.lib.gbp_ba_loop(self.h, n, int(iter0), int(steps), arr), "gbp_ba_loop")
        return [{k: getattr(arr[i], k) for k, _ in arr[i]._fields_} for i in range(n)]

    def eval_end(self):
        o = cabi.GbpEvalOut()
        self._chk(self.lib.gbp_eval_end(self.h, C.byref(o)), "gbp_eval_end")
        return {k: getattr(o, k) for k, _ in o._fields_}

    def sync(self):
        self._chk(self.lib.gbp_sync(self.h), "gbp_sync")

    def timing(self, reset=False):
        t = cabi.GbpTimingOut()
        self._chk(self.lib.gbp_timing(self.h, C.byref(t), int(reset)), "gbp_timing")
        return {k: getattr(t, k) for k, _ in t._fields_}

    def set_profiling(self, on):
        self._chk(self.lib.gbp_set_profiling(self.h, int(bool(on))), "gbp_set_profiling")

    # ---- split-phase (sharded) ----
    def set_stream(self, stream_handle):
        self._chk(self.lib.gbp_set_stream(self.h, C.c_void_p(stream_handle)), "gbp_set_stream")

    def set_exchange_buffers(self, send_ptr, recv_ptr):
        self._chk(self.lib.gbp_set_exchange_buffers(self.h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr)),
                  "gbp_set_exchange_buffers")

    def iterate_begin(self):
        self._chk(self.lib.gbp_iterate_begin(self.h), "gbp_iterate_begin")

    def iterate_local(self):
        self._chk(self.lib.gbp_iterate_local(self.h), "gbp_iterate_local")

    def iterate_end(self):
        self._chk(self.lib.gbp_iterate_end(self.h), "gbp_iterate_end")

    def refresh_begin(self):
        self._chk(self.lib.gbp_refresh_begin(self.h), "gbp_refresh_begin")

    def refresh_end(self):
        self._chk(self.lib.gbp_refresh_end(self.h), "gbp_refresh_end")

    def linearise_factors(self):
        self._chk(self.lib.gbp_linearise_factors(self.h), "gbp_linearise_factors")

    # ---- library-owned exchange (RCCL from the C++ host) ----
    def comm_unique_id(self):
        """Rank 0: the 128-byte RCCL id every rank passes to comm_init_rccl (distribute it with your launcher's means)."""
        buf = C.create_string_buffer(128)
        rc = self.lib.gbp_comm_unique_id(buf)
        if rc != 0:
            raise GbpError("gbp_comm_unique_id: %s (status %d)" % (self.lib.gbp_last_error(None).decode(), rc))
        return buf.raw

    def comm_init_rccl(self, id128):
        buf = C.create_string_buffer(bytes(id128), 128)
        self._chk(self.lib.gbp_comm_init_rccl(self.h, buf), "gbp_comm_init_rccl")

    def comm_describe(self):
        """dict: rank, world, device, pci_bus_id, transport, library (resolved path of librccl), library_version, two_streams"""
        import json
        buf = C.create_string_buffer(1024)
        self._chk(self.lib.gbp_comm_describe(self.h, buf, 1024), "gbp_comm_describe")
        return json.loads(buf.value.decode())

    def comm_set_schedule(self, two_streams):
        self._chk(self.lib.gbp_comm_set_schedule(self.h, int(bool(two_streams))), "gbp_comm_set_schedule")

    def comm_probe(self, reps=50):
        """mean duration (us) of one all-gather of the camera partial buffers, `reps` back to back (collective)"""
        us = C.c_double(0.0)
        self._chk(self.lib.gbp_comm_probe(self.h, int(reps), C.byref(us)), "gbp_comm_probe")
        return us.value

    def comm_transport(self):
        return self.lib.gbp_comm_transport(self.h).decode()

    def graph_state(self):
        """2: bursts run inside the persistent kernel (small graph); 1: gbp_iterate replays a captured hipGraph;
        0: nothing captured yet; -1: capture failed, direct launches."""
        return int(self.lib.gbp_graph_state(self.h))

    def comm_barrier(self):
        self._chk(self.lib.gbp_comm_barrier(self.h), "gbp_comm_barrier")

    def eval_global(self):
        o = cabi.GbpEvalOut()
        self._chk(self.lib.gbp_eval_global(self.h, C.byref(o)), "gbp_eval_global")
        return {k: getattr(o, k) for k, _ in o._fields_}

    # ---- raw state for parity tests ----
    def _need_hooks(self):
        if not self.hooks:
            raise GbpError("internal state is reachable only through the test-hooks build: GbpEngine(..., hooks=True)")

    def _debug(self, what, na, nb):
        self._need_hooks()
        a, b = np.zeros(na, np.float32), np.zeros(nb, np.float32)
        self._chk(self.lib.gbp_debug_get(self.h, what, cabi.ptr(a, cabi.c_f32p), cabi.ptr(b, cabi.c_f32p)),
                  "gbp_debug_get")
        return a, b

    def persist_flow(self, on):
        """bursts without the metric in the persistent kernel: tagged-record hand-offs (default) or counter barriers (A/B, tests)"""
        self._need_hooks()
        self._chk(self.lib.gbp_debug_persist_flow(self.h, int(bool(on))), "gbp_debug_persist_flow")

    def persist_verify(self, on):
        """redundant records in the persistent kernel (test hook): every tagged record published twice and compared by its consumers;
        returns the mismatches counted since the last call"""
        self._need_hooks()
        n = C.c_uint64(0)
        self._chk(self.lib.gbp_debug_persist_verify(self.h, int(bool(on)), C.byref(n)), "gbp_debug_persist_verify")
        return int(n.value)

    def factor_potentials(self):
        return self._debug(0, 9 * self.E, 81 * self.E)

    def set_factor_potentials(self, eta, lam):
        self._need_hooks()
        eta, lam = np.ascontiguousarray(eta, np.float32), np.ascontiguousarray(lam, np.float32)
        self._chk(self.lib.gbp_debug_set_factor_potentials(self.h, cabi.ptr(eta, cabi.c_f32p), cabi.ptr(lam, cabi.c_f32p)),
                  "gbp_debug_set_factor_potentials")

    def messages(self):
        ce, cl = self._debug(1, 6 * self.E, 36 * self.E)
        le, ll = self._debug(2, 3 * self.E, 9 * self.E)
        return {"cam_eta": ce, "cam_lambda": cl, "lmk_eta": le, "lmk_lambda": ll}

    def mu(self):
        return self._debug(3, 9 * self.E, self.E)
