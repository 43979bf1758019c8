"""Rank-local stand-in for GbpEngine built on the CPU oracle's split-phase view (tests only).

It exposes the same split-phase verbs the C-ABI offers for a sharded ctx, operating on CPU tensors,
so gbp_poplar_amd.distributed.ShardedGbp (the product's multi-process host logic) can run under gloo."""
import ctypes as C

import numpy as np

from oracle import oracle as orc
from gbp_poplar_amd import _cabi as cabi


class OracleShardEngine(orc.Oracle):
    def __init__(self, cam_id, lmk_id, n_cams, n_lmks, K9, shard):
        super().__init__(cam_id, lmk_id, n_cams, n_lmks, K9)
        rank, world, lb, le = shard
        lib = self.lib
        lib.orc_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
        for n in ("orc_iterate_begin", "orc_iterate_end", "orc_refresh_begin", "orc_refresh_end",
                  "orc_weaken_priors_sharded"):
            getattr(lib, n).argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_linearise_factors.argtypes = [C.c_void_p]
        self._chk(lib.orc_set_shard(self.h, rank, world, lb, le), "set_shard")
        self.shard = shard
        self.send = self.recv = None
        self.world = world

    def set_stream(self, s):
        pass

    def set_exchange_buffers(self, send_ptr, recv_ptr):
        self.send, self.recv = C.c_void_p(send_ptr), C.c_void_p(recv_ptr)

    def _recv_ptr(self):
        return self.recv

    def iterate_begin(self):
        self._chk(self.lib.orc_iterate_begin(self.h, self.send), "iterate_begin")

    def iterate_end(self):
        self._chk(self.lib.orc_iterate_end(self.h, self._recv_ptr()), "iterate_end")

    def refresh_begin(self):
        self._chk(self.lib.orc_refresh_begin(self.h, self.send), "refresh_begin")

    def refresh_end(self):
        self._chk(self.lib.orc_refresh_end(self.h, self._recv_ptr()), "refresh_end")

    def linearise_factors(self):
        self._chk(self.lib.orc_linearise_factors(self.h), "linearise_factors")

    def weaken_priors(self):
        self._chk(self.lib.orc_weaken_priors_sharded(self.h, self._recv_ptr()), "weaken_priors")

    def new_keyframe(self, upd):
        keep = []
        s = cabi.fill_struct(cabi.GbpKfUpdate(), upd, keep)
        self.lib.orc_new_keyframe_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._chk(self.lib.orc_new_keyframe_sharded(self.h, C.byref(s), self._recv_ptr()), "new_keyframe")

    def read_priors(self):
        """Like the sharded C-ABI ctx: camera priors complete, landmark priors of the owned range only."""
        p = super().read_priors()
        lo, hi = self.shard[2], self.shard[3]
        for k, w in (("lmk_priors_eta", 3), ("lmk_priors_lambda", 9)):
            v = np.zeros_like(p[k])
            v[w * lo:w * hi] = p[k][w * lo:w * hi]
            p[k] = v
        return p

    def iterate_local(self):
        pass                                  # the oracle does the landmark half inside iterate_end

    def sync(self):
        pass
