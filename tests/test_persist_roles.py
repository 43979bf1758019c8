"""Who owns what inside a launch of the persistent kernel (csrc/gbp_kernels.h: persist_role, persist_grid) — evaluated on the host
through the test-hooks library, no GPU: every camera, every group of 16 landmarks and (with the metric after every iteration) every
metric mean has exactly one wave; where the grid is sized for it the cameras and metric means sit on waves WITHOUT a sweep tile; the
grid never exceeds one workgroup per CU when it separates roles."""
import ctypes as C

import numpy as np
import pytest


def roles(n_tiles, n_cams, n_lmks, with_metric):
    from gbp_poplar_amd import _lib
    lib = _lib.load(hooks=True)
    dims = (C.c_uint32 * 4)()
    assert lib.gbp_debug_persist_roles(n_tiles, n_cams, n_lmks, int(with_metric), dims, None, 0) == 0
    nb = dims[0]
    role = (C.c_uint32 * (4 * nb))()
    assert lib.gbp_debug_persist_roles(n_tiles, n_cams, n_lmks, int(with_metric), dims, role, 4 * nb) == 0
    return list(dims), np.array(role, dtype=np.uint64)


SHAPES = [(208, 42, 2194), (244, 63, 2869), (60, 20, 862),          # fr1xyz, fr1desk, fr2robot2
          (4, 1, 1), (4, 5, 24), (8, 2, 500), (1000, 100, 6400), (1024, 3, 17), (512, 300, 9000), (12, 40, 100), (400, 200, 30000)]


@pytest.mark.parametrize("with_metric", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_every_role_has_exactly_one_wave(shape, with_metric):
    n_tiles, C_, L = shape
    (nb, separate, n_met, G), role = roles(n_tiles, C_, L, with_metric)
    assert G == (L + 15) // 16 and role.size == 4 * nb and nb * 4 >= n_tiles
    none = (role == 0xFFFFFFFF) | (role >= 2 * C_ + G)      # (shared waves: a number past the last role is no role)
    r = role[~none]
    cams = r[r < C_]
    lmks = r[(r >= C_) & (r < C_ + G)] - C_
    mets = r[r >= C_ + G] - (C_ + G)
    assert sorted(cams.tolist()) == list(range(C_)), "every camera exactly once"
    assert sorted(lmks.tolist()) == list(range(G)), "every landmark group exactly once"
    if separate:
        assert nb <= 256 and n_met == (C_ if with_metric else 0)
        assert sorted(mets.tolist()) == list(range(n_met))
        wave = np.arange(4 * nb)
        tile_less = wave >= n_tiles                       # wave 4 * workgroup + w sweeps tile of the same number
        is_cam_or_met = ~none & ((role < C_) | (role >= C_ + G))
        assert np.all(tile_less[is_cam_or_met]), "cameras and metric means on waves without a tile"
        if with_metric and (n_tiles + 2 * C_ + G + 3) // 4 <= 256:
            assert np.all(tile_less[~none]), "with the metric the landmark owners move off the tile waves too where that fits"
    else:
        # roles and tiles share waves (k_persist's numbering): metric roles only where the grid has waves for them
        assert mets.size <= C_ and len(set(mets.tolist())) == mets.size


def test_the_shipped_sequences_get_separated_roles():
    for shape, want in (((208, 42, 2194), (63, 108)), ((244, 63, 2869), (77, 138)), ((60, 20, 862), (20, 39))):
        (nb0, sep0, _, _), _ = roles(*shape, False)
        (nb1, sep1, nm, _), _ = roles(*shape, True)
        assert (sep0, sep1, nm) == (1, 1, shape[1]) and (nb0, nb1) == want
    (nb, sep, _, _), _ = roles(1000, 100, 6400, False)      # 64 000 factors: one workgroup per CU already — no room to separate
    assert sep == 0 and nb == 250
