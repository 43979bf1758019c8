"""CPU property tests of the DEVICE ORDER (csrc/gbp_layout.cpp) — what gbp_create builds before it touches the GPU: which
factor sits at which device position, where a camera's rows and a landmark's message records are, in which order the sweep's
wavefronts take the tiles.  It replaces the vertex-to-tile mapping and tensor-slice wiring of the reference (ba/ba.cpp:71-97,
243-366); the one thing the reference fixes is kept and checked here: the message-slot order of a variable is the FILE order of
its incident factors (ba/ba.cpp:267-279).

Every check below is an independent numpy restatement of a property, not of the construction; the same families of graphs run
through the builder under ASan + UBSan in tests/sanitize/ (tests/test_host_sanitizers.py).  No GPU is needed: the layout
comes out of the test-hooks library through gbp_debug_layout_build.
"""
import numpy as np
import pytest

from gbp_poplar_amd import hostlib

PAD = 0xFFFFFFFF


def check_layout(cam_id, lmk_id, C, L, y, shard=None, opt=None, tile_order=0):
    cam_id, lmk_id = np.asarray(cam_id, np.int64), np.asarray(lmk_id, np.int64)
    lo, hi = (0, L) if shard is None else (shard[2], shard[3])
    local = np.flatnonzero((lmk_id >= lo) & (lmk_id < hi))          # local factors in FILE order
    E_loc = local.size
    assert (y["C"], y["L"], y["E"], y["lmk_begin"], y["lmk_end"], y["L_loc"], y["E_loc"]) == (C, L, cam_id.size, lo, hi, hi - lo, E_loc)
    deg = np.bincount(cam_id[local], minlength=C)
    rows = (deg + 15) // 16
    assert y["n_rows"] == rows.sum()
    assert y["Ep"] % 256 == 0 and y["Ep"] >= max(16 * y["n_rows"], 256) and y["Ep"] - 16 * y["n_rows"] < 256 + (256 if y["n_rows"] == 0 else 0)
    assert y["n_tiles"] * 64 == y["Ep"]
    assert np.array_equal(y["cam_row_ptr"], np.concatenate([[0], np.cumsum(rows)]))
    pe = y["pos_edge"]
    assert pe.size == y["Ep"]
    used = pe != PAD
    # every local factor placed exactly once, nothing else placed
    assert np.array_equal(np.sort(pe[used]), local)
    # pads are flagged everywhere: landmark slot "behind the last", landmark / camera index valid (they are read)
    assert np.all(y["pos_lpos"][~used] == E_loc)
    assert np.all(y["pos_lmk_loc"] < max(hi - lo, 1)) and np.all(y["pos_cam"] < C)
    # a camera's factors sit in FILE order along its rows; rows are whole (16 consecutive factors) and found through row_slot
    row_slot = y["row_slot"] if y["row_slot"].size else np.arange(y["n_rows"])
    assert row_slot.size == y["n_rows"]
    order = local[np.argsort(cam_id[local], kind="stable")]                     # camera-major, file order inside a camera
    i_in_cam = np.arange(E_loc) - np.repeat(np.concatenate([[0], np.cumsum(deg)[:-1]]), deg)
    cam_of = cam_id[order]
    want_pos = row_slot[y["cam_row_ptr"][cam_of] + i_in_cam // 16].astype(np.int64) * 16 + i_in_cam % 16
    assert np.array_equal(pe[want_pos], order)
    assert np.array_equal(y["pos_cam"][want_pos], cam_of)
    assert np.array_equal(y["pos_lmk_loc"][want_pos], lmk_id[order] - lo)
    # row -> camera table of the sweep: every USED device row names its camera (all 16 lanes of a used row, pads included)
    dev_rows = row_slot[np.repeat(y["cam_row_ptr"][:-1], rows) + (np.arange(y["n_rows"]) - np.repeat(y["cam_row_ptr"][:-1], rows))] if y["n_rows"] else np.zeros(0, np.int64)
    assert np.array_equal(np.sort(dev_rows), np.arange(y["n_rows"]))            # row_slot is a bijection onto the used rows
    assert np.array_equal(y["row_cam"][dev_rows], np.repeat(np.arange(C), rows))
    assert np.array_equal(y["pos_cam"].reshape(-1, 16)[dev_rows], np.repeat(np.arange(C), rows)[:, None] * np.ones((1, 16), np.int64))
    # landmark slots: file order of the incident factors (ba.cpp:267-279), slot list -> device position, 64-B index record
    ldeg = np.bincount(lmk_id[local] - lo, minlength=hi - lo)
    assert np.array_equal(y["lmk_ptr"], np.concatenate([[0], np.cumsum(ldeg)]))
    lorder = local[np.argsort(lmk_id[local], kind="stable")]                    # landmark-major, file order inside a landmark
    inv = np.empty(cam_id.size, np.int64)
    inv[pe[used]] = np.flatnonzero(used)                                        # file edge -> device position
    assert np.array_equal(y["lmk_fpos"], inv[lorder])
    assert np.array_equal(y["pos_lpos"][inv[lorder]], np.arange(E_loc))
    ix = y["lmk_ix"].reshape(-1, 16)
    assert ix.shape[0] == hi - lo and np.array_equal(ix[:, 0], ldeg)
    for k in range(15):
        has = ldeg > k
        assert np.array_equal(ix[has, 1 + k], y["lmk_fpos"][y["lmk_ptr"][:-1][has] + k])
        assert np.all(ix[~has, 1 + k] == 0)                                     # unused slots: position 0 (a valid record, loaded unconditionally)
    # row placement: a bijection INSIDE each window of cameras, stable by the landmark class of the row's key factor
    if y["row_slot"].size:
        W, K = y["row_window"], (opt.classes if opt is not None else 8)
        assert W > 0
        key_lane = opt.row_key_lane if opt is not None else 8
        first = np.full(y["n_rows"], -1, np.int64)
        r_of = y["cam_row_ptr"][cam_of] + i_in_cam // 16
        for lane in sorted({0, key_lane}):                                      # the key lane overrides the first factor where the row has one
            m = i_in_cam % 16 == lane
            first[r_of[m]] = lmk_id[order][m] - lo
        cls = np.minimum(first * K // max(hi - lo, 1), K - 1)
        for c0 in range(0, C, W):
            c1 = min(C, c0 + W)
            R0, R1 = y["cam_row_ptr"][c0], y["cam_row_ptr"][c1]
            sl = y["row_slot"][R0:R1].astype(np.int64)
            assert np.array_equal(np.sort(sl), np.arange(R0, R1))
            if opt is not None and opt.row_sort_in_class:      # (measurement option: by class, then by the key landmark, then camera-major)
                want = np.argsort(cls[R0:R1] * (hi - lo + 1) + first[R0:R1], kind="stable")
            else:
                want = np.argsort(cls[R0:R1], kind="stable")
            assert np.array_equal(np.argsort(sl), want)                 # device order = stable sort by class
    else:
        assert y["row_window"] == 0
    # execution order of the tiles: a bijection; the local order keeps every tile near its sequential place
    if y["tile_perm"].size:
        perm = y["tile_perm"].astype(np.int64)
        assert np.array_equal(np.sort(perm), np.arange(y["n_tiles"]))
        if tile_order != 2:
            window = opt.tile_window if opt is not None else 96
            assert np.max(np.abs(perm - np.arange(y["n_tiles"]))) <= window + 32


def random_graph(rng, C, L, E, sort=False, dup=False):
    cam = rng.integers(0, C, E)
    lmk = rng.integers(0, L, E)
    if dup:
        cam[1::3], lmk[1::3] = cam[0:-1:3][: cam[1::3].size], lmk[0:-1:3][: lmk[1::3].size]   # every third edge repeats its predecessor
    if sort:
        o = np.lexsort((lmk, cam))
        cam, lmk = cam[o], lmk[o]
    return cam.astype(np.uint32), lmk.astype(np.uint32)


SMALL = dict(tile_min_tiles=4, row_window=3, row_place_max_deg=10 ** 6)      # the placement machinery on graphs of a few tiles


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("tile_order", [0, 1, 2, 3])
def test_random_graphs(seed, tile_order):
    """unsorted files, duplicate edges, cameras / landmarks without factors, ragged degrees; default and shrunken thresholds"""
    rng = np.random.default_rng(100 + seed)
    C, L = int(rng.integers(1, 40)), int(rng.integers(1, 300))
    E = int(rng.integers(1, 4000))
    cam, lmk = random_graph(rng, C, L, E, sort=seed % 2 == 0, dup=seed % 3 == 0)
    for kw in ({}, SMALL, dict(SMALL, classes=16, row_key_lane=3, tile_window=7), dict(SMALL, row_sort_in_class=1, row_key_lane=0)):
        opt = hostlib.layout_options(**kw)
        y = hostlib.layout_build(cam, lmk, C, L, tile_order=tile_order, options=opt)
        check_layout(cam, lmk, C, L, y, opt=opt, tile_order=tile_order)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_cover_the_graph(world):
    """every rank of a landmark-sharded run lays out exactly the factors of its landmark range; ranges may be empty"""
    rng = np.random.default_rng(7)
    cam, lmk = random_graph(rng, 24, 200, 3000, sort=True)
    bounds = sorted(rng.integers(0, 201, world - 1).tolist())
    bounds = [0] + bounds + [200]
    bounds[1] = bounds[0] if world == 8 else bounds[1]              # an EMPTY shard among eight
    seen = []
    for r in range(world):
        sh = (r, world, bounds[r], bounds[r + 1])
        opt = hostlib.layout_options(**SMALL)
        y = hostlib.layout_build(cam, lmk, 24, 200, shard=sh, options=opt)
        check_layout(cam, lmk, 24, 200, y, shard=sh, opt=opt)
        seen.append(y["pos_edge"][y["pos_edge"] != PAD])
        if bounds[r] == bounds[r + 1]:
            assert y["E_loc"] == 0 and y["Ep"] == 256 and np.all(y["pos_edge"] == PAD)
    assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(3000))


def test_degenerate_shapes():
    for cam, lmk, C, L in (([0], [0], 1, 1),                                      # one factor
                           ([0] * 700, list(range(700)), 1, 700),                  # ONE camera, 44 rows
                           (list(range(50)), [0] * 50, 50, 1),                     # one landmark seen by every camera (degree > 15: the slot list beyond the index record)
                           ([3, 3, 3], [1, 1, 1], 6, 4)):                          # three duplicates, cameras without factors
        for kw in ({}, SMALL):
            opt = hostlib.layout_options(**kw)
            y = hostlib.layout_build(np.array(cam, np.uint32), np.array(lmk, np.uint32), C, L, tile_order=3, options=opt)
            check_layout(cam, lmk, C, L, y, opt=opt, tile_order=3)


def test_config5_shard_shape_places_rows_by_landmark_class():
    """The shape every rank of BASELINE config 5 gets — many cameras with few factors each, >= 2 048 tiles — with the PRODUCT's
    options: rows are placed (windows of 32 cameras), tiles are permuted, every property holds; and the point of it: a tile covers
    a third of the landmark range it covers in camera-major order."""
    bal = hostlib.synth_generate(4096, 40000, 10, 7)          # 400 000 factors, ~98 per camera
    cam, lmk = bal["cam_id"], bal["lmk_id"]
    y = hostlib.layout_build(cam, lmk, 4096, 40000)
    assert y["row_window"] == 32 and y["row_slot"].size == y["n_rows"] and y["tile_perm"].size == y["n_tiles"] >= 2048
    check_layout(cam, lmk, 4096, 40000, y)
    y1 = hostlib.layout_build(cam, lmk, 4096, 40000, tile_order=1)
    assert y1["row_slot"].size == 0 and y1["tile_perm"].size == 0
    check_layout(cam, lmk, 4096, 40000, y1, tile_order=1)

    def mean_tile_span(lay):          # share of the landmark range a tile's 64 factors cover, averaged over the tiles
        l = lay["pos_lmk_loc"].astype(np.float64) / 40000
        pad = lay["pos_edge"] == PAD
        lo, hi = np.where(pad, 9, l).reshape(-1, 64).min(axis=1), np.where(pad, -9, l).reshape(-1, 64).max(axis=1)
        return float(np.mean((hi - lo)[hi >= 0]))
    assert mean_tile_span(y) < 0.3 < 0.6 < mean_tile_span(y1)          # measured 0.24 against 0.78


def test_s1_shape_keeps_camera_major_rows():
    """1 000 factors per camera (the 1M-factor graph's shape, scaled down): tiles permuted, rows NOT placed"""
    bal = hostlib.synth_generate(200, 20000, 10, 3)
    y = hostlib.layout_build(bal["cam_id"], bal["lmk_id"], 200, 20000)
    assert y["row_slot"].size == 0 and y["tile_perm"].size == y["n_tiles"] >= 2048
    check_layout(bal["cam_id"], bal["lmk_id"], 200, 20000, y)


def test_bad_inputs_are_refused_without_a_device():
    cam, lmk = np.array([0, 1], np.uint32), np.array([0, 5], np.uint32)
    with pytest.raises(RuntimeError, match="index out of range"):
        hostlib.layout_build(cam, lmk, 2, 5)
    with pytest.raises(RuntimeError, match="bad shard"):
        hostlib.layout_build(cam, np.array([0, 1], np.uint32), 2, 5, shard=(0, 2, 3, 9))
    with pytest.raises(RuntimeError, match="bad shard"):
        hostlib.layout_build(cam, np.array([0, 1], np.uint32), 2, 5, shard=(2, 2, 0, 5))
    with pytest.raises(RuntimeError, match="bad layout options"):
        hostlib.layout_build(cam, np.array([0, 1], np.uint32), 2, 5, options=hostlib.layout_options(classes=0))


def test_local_tile_order_with_finer_classes():
    rng = np.random.default_rng(2)
    for K in (8, 16, 32):
        cls = rng.integers(0, K + 1, 3000).astype(np.uint8)        # class K: a tile of pads only
        for window in (1, 16, 96):
            perm = hostlib.tile_order_local(cls, window, K).astype(np.int64)
            assert np.array_equal(np.sort(perm), np.arange(3000))
            assert np.max(np.abs(perm - np.arange(3000))) <= window + 32
            if window == 96 and K == 8:        # most workgroups get the class they ask for (random classes: the worst case)
                want = (np.arange(3000) // 4) % K
                assert np.mean(cls[perm] == want) > 0.75


@pytest.mark.parametrize("shape", ["s1_like", "config5_like", "unsorted"])
def test_the_device_order_does_not_depend_on_the_host_thread_count(shape, monkeypatch):
    """Above 2^19 factors the builder cuts the file into chunks of consecutive factors, one host thread each (at most 8): every chunk counts
    its factors per camera and per landmark, the running sums over the chunks are where a chunk continues a camera's rows and a landmark's
    slots.  The result must be the one-thread result — every array — and keep every property above (file order of the slots among them)."""
    rng = np.random.default_rng(5)
    if shape == "s1_like":
        C, L, E = 300, 220000, 2200000
        cam, lmk = random_graph(rng, C, L, E, sort=True)
    elif shape == "config5_like":      # many small cameras: rows placed by landmark class, tiles permuted
        C, L, E = 12000, 150000, 2200000
        cam, lmk = random_graph(rng, C, L, E, sort=True)
    else:
        C, L, E = 700, 90000, 2200000
        cam, lmk = random_graph(rng, C, L, E, sort=False, dup=True)
    out = {}
    for threads in ("1", "8", "3"):
        monkeypatch.setenv("GBP_HOST_THREADS", threads)
        out[threads] = hostlib.layout_build(cam, lmk, C, L)
    for k in out["1"]:
        for threads in ("8", "3"):
            assert np.array_equal(out["1"][k], out[threads][k]), (shape, threads, k)
    check_layout(cam, lmk, C, L, out["8"])
    shard = (1, 3, L // 3, 2 * L // 3)
    a = hostlib.layout_build(cam, lmk, C, L, shard=shard)
    monkeypatch.setenv("GBP_HOST_THREADS", "1")
    b = hostlib.layout_build(cam, lmk, C, L, shard=shard)
    for k in a:
        assert np.array_equal(a[k], b[k]), (shape, "shard", k)
