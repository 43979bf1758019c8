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
        self.world, self.chunks = world, 1
        self._flat = None

    def set_stream(self, s):
        pass

    def set_exchange_buffers(self, send_ptr, recv_ptr):
        self.send, self.recv = C.c_void_p(send_ptr), C.c_void_p(recv_ptr)

    # ---- pipelined exchange: the oracle computes all partials in piece 0; the chunked receive layout
    #      ([chunk][world][n_i][44]) is re-flattened to the oracle's [world][C][44] before combining ----
    def set_exchange_chunks(self, n):
        self.chunks = int(n)

    def iterate_begin_chunk(self, chunk):
        if chunk == 0:
            self.iterate_begin()

    def _recv_ptr(self):
        if self.chunks == 1:
            return self.recv
        n = self.world * self.C * 44
        raw = np.ctypeslib.as_array((C.c_float * n).from_address(self.recv.value))
        flat = np.empty((self.world, self.C, 44), np.float32)
        st = [self.C * i // self.chunks for i in range(self.chunks + 1)]
        for i in range(self.chunks):
            n_i = st[i + 1] - st[i]
            blk = raw[self.world * st[i] * 44:self.world * st[i + 1] * 44].reshape(self.world, n_i, 44)
            flat[:, st[i]:st[i + 1], :] = blk
        self._flat = np.ascontiguousarray(flat)
        return C.c_void_p(self._flat.ctypes.data)

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
