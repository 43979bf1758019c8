"""The rendezvous / staging region of the C++ launcher (csrc/gbp_comm.cpp) across REAL processes on the CPU: forked
ranks share a MAP_SHARED mapping, gather two doubles per rank and meet at a barrier, many rounds; a supervisor's abort
wakes ranks that wait for one that never arrives.  (The device half of the transports is covered by the -m gpu tests of
bin/ba --ipus N.)"""
import ctypes
import mmap
import os
import time

import pytest


def _lib():
    from gbp_poplar_amd._lib import load
    return load()


def _fork_ranks(world, body):
    pids = []
    for rank in range(world):
        pid = os.fork()
        if pid == 0:
            rc = 1
            try:
                rc = body(rank)
            finally:
                os._exit(rc)
        pids.append(pid)
    return pids


@pytest.mark.parametrize("world", [2, 3, 5])
def test_forked_ranks_gather_and_barrier(world):
    lib = _lib()
    n = lib.gbp_comm_region_bytes(100, world)
    mm = mmap.mmap(-1, n)                                   # MAP_SHARED | MAP_ANONYMOUS: inherited by the forked ranks
    region = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    assert lib.gbp_comm_region_init(region, n, 100, world) == 0
    pids = _fork_ranks(world, lambda rank: 0 if lib.gbp_comm_region_selftest(region, rank, world, 2000) == 0 else 3)
    codes = [os.waitpid(p, 0)[1] for p in pids]
    assert all(os.WIFEXITED(c) and os.WEXITSTATUS(c) == 0 for c in codes), codes


def test_abort_wakes_ranks_waiting_for_a_missing_one():
    lib = _lib()
    world = 3
    n = lib.gbp_comm_region_bytes(10, world)
    mm = mmap.mmap(-1, n)
    region = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    assert lib.gbp_comm_region_init(region, n, 10, world) == 0
    # ranks 0 and 1 start, rank 2 never does: they sit in the first barrier until the supervisor aborts the region
    pids = _fork_ranks(2, lambda rank: 0 if lib.gbp_comm_region_selftest(region, rank, world, 10) == 0 else 3)
    time.sleep(0.3)
    assert all(os.waitpid(p, os.WNOHANG)[0] == 0 for p in pids)          # still waiting
    lib.gbp_comm_region_abort(region)
    t0 = time.time()
    codes = [os.waitpid(p, 0)[1] for p in pids]
    assert time.time() - t0 < 5.0
    assert all(os.WIFEXITED(c) and os.WEXITSTATUS(c) == 3 for c in codes), codes   # error return, not a hang


def test_wrong_world_is_refused():
    lib = _lib()
    n = lib.gbp_comm_region_bytes(10, 2)
    buf = ctypes.create_string_buffer(n)
    assert lib.gbp_comm_region_init(buf, n, 10, 2) == 0
    assert lib.gbp_comm_region_selftest(buf, 0, 3, 1) != 0               # region was made for 2 ranks
    assert lib.gbp_comm_region_selftest(buf, 2, 2, 1) != 0               # rank out of range


def test_missing_rccl_library_is_reported_not_crashed():
    """with librccl unloadable, gbp_comm_unique_id (needs no device) returns GBP_ERR_COMM and a message naming
    the path and the loader's reason — it used to dereference a NULL dlerror() and crash.  GBP_RCCL_LIB is a strict
    override: only that path is tried."""
    import subprocess
    import sys
    code = ("import ctypes, os, sys\n"
            "sys.path.insert(0, %r)\n"
            "from gbp_poplar_amd._lib import load\n"
            "lib = load()\n"
            "buf = ctypes.create_string_buffer(128)\n"
            "rc = lib.gbp_comm_unique_id(buf)\n"
            "rc2 = lib.gbp_comm_unique_id(buf)\n"          # the cached failure path
            "print(rc, rc2, lib.gbp_last_error(None).decode())\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GBP_NO_TORCH="1", GBP_RCCL_LIB="/nonexistent/dir/librccl-missing.so")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    rc, rc2, msg = p.stdout.strip().split(" ", 2)
    assert (rc, rc2) == ("-7", "-7"), p.stdout
    assert "librccl not found" in msg and "/nonexistent/dir/librccl-missing.so" in msg and "No such file" in msg, msg
