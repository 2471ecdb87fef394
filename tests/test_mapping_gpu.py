"""Stage C parity: scan-to-map on the HIP path vs the oracle, fed the same stage-A/B outputs scan by scan.

Bar: poses within 1e-5 (north_star).  Observed differences are ~1e-12 per step because every f32 quantity that
feeds a gate (kNN distances, voxel centroids, associated points) is bit-identical and only the order of the f64
normal-equation reduction differs from Ceres' sequential sum; the test asserts 1e-7 to leave room for that.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sorted_rows(a):
    a = np.ascontiguousarray(a, np.float32)
    v = a.view(np.uint32).reshape(-1, 4)
    order = np.lexsort((v[:, 3], v[:, 2], v[:, 1], v[:, 0]))
    return v[order]


@pytest.fixture(scope="module")
def stage_ab(O, hdl64_stream):
    """Oracle stage A + B outputs for the first scans (inputs of stage C for both implementations)."""
    od = O.Odometry()
    out = []
    for k in range(10):
        f = O.features(hdl64_stream(k), O.HDL64, 5.0)
        c = f["cloud"]
        qlc, tlc, qw, tw, st = od.step(c[f["sharp"]], c[f["less_sharp"]], c[f["flat"]], f["less_flat"])
        out.append(dict(corner=c[f["less_sharp"]].copy(), surf=f["less_flat"].copy(), full=c.copy(), q=qw.copy(), t=tw.copy()))
    return out


def test_stream_parity(O, S, stage_ab):
    om = O.Mapper(0.4, 0.8, voxel_order=1, knn_mode=0)
    gm = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=2000000)
    worst = 0.0
    for k, fr in enumerate(stage_ab):
        qo, to, so, rego = om.step(fr["corner"], fr["surf"], fr["full"], fr["q"], fr["t"], want_registered=True)
        qg, tg, sg, regg = gm.process(fr["corner"], fr["surf"], fr["full"], fr["q"], fr["t"], want_registered=True)
        assert sg.n_corner_stack == so.n_corner_stack and sg.n_surf_stack == so.n_surf_stack, k
        assert sg.n_corner_map == so.n_corner_map and sg.n_surf_map == so.n_surf_map, k
        assert sg.solved == so.solved
        assert list(sg.n_edge) == list(so.n_edge) and list(sg.n_plane) == list(so.n_plane), (k, list(sg.n_edge), list(so.n_edge), list(sg.n_plane), list(so.n_plane))
        assert list(sg.lm_iters) == list(so.lm_iters) and list(sg.lm_success) == list(so.lm_success), k
        for o in range(2):
            assert abs(sg.cost_init[o] - so.cost_init[o]) <= 1e-9 * max(1.0, so.cost_init[o])
            assert abs(sg.cost_final[o] - so.cost_final[o]) <= 1e-9 * max(1.0, so.cost_final[o])
        d = max(np.abs(qg - qo).max(), np.abs(tg - to).max())
        worst = max(worst, d)
        assert d <= 1e-7, (k, d)
        # registered full-resolution cloud (:845-849): f32 results of an f64 transform
        nbad = (regg.view(np.uint32) != rego.view(np.uint32)).any(axis=1).sum()
        assert nbad <= max(5, regg.shape[0] // 10000), (k, nbad)
        # map content of the 5x5x3 window (laserCloudCornerFromMap / SurfFromMap for the next scan): same point set
        for which in (0, 1):
            mo = _sorted_rows(om.export(which))
            mg = _sorted_rows(gm.export(which))
            assert mo.shape == mg.shape, (k, which, mo.shape, mg.shape)
            nb = (mo != mg).any(axis=1).sum()
            assert nb <= max(3, mo.shape[0] // 5000), (k, which, nb, mo.shape[0])
        qa, ta = om.wmap_wodom()
        qb, tb = gm.wmap_wodom()
        assert np.abs(qa - qb).max() <= 1e-7 and np.abs(ta - tb).max() <= 1e-7
    print("worst pose difference over the stream:", worst)
    gm.close()


def test_first_scan_and_small_map(O, S, stage_ab):
    """Map too small (:555, :731-734): no solve, the prior pose is returned and the scan is still inserted."""
    om = O.Mapper(0.4, 0.8)
    gm = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=1000000)
    fr = stage_ab[0]
    qo, to, so, _ = om.step(fr["corner"], fr["surf"], None, fr["q"], fr["t"])
    qg, tg, sg, _ = gm.process(fr["corner"], fr["surf"], None, fr["q"], fr["t"])
    assert so.solved == 0 and sg.solved == 0
    assert np.array_equal(qo, qg) and np.array_equal(to, tg)
    assert sg.n_map_corner_total > 0 and sg.n_map_surf_total > 0
    # empty inputs are legal
    qg, tg, sg, _ = gm.process(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), None, fr["q"], fr["t"])
    assert sg.n_corner_stack == 0 and sg.n_surf_stack == 0
    gm.close()


def test_window_roll(O, S, stage_ab):
    """Drive the pose across cube boundaries so the 21x21x11 window rolls (:324-508) and slabs are dropped."""
    om = O.Mapper(0.4, 0.8)
    gm = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=2000000)
    fr = stage_ab[1]
    q = np.array([0.0, 0.0, 0.0, 1.0])
    for step, tx in enumerate([0.0, 180.0, 420.0, 480.0, -300.0, -620.0, 0.0]):
        t = np.array([tx, 0.3 * tx, -0.1 * tx])
        qo, to, so, _ = om.step(fr["corner"], fr["surf"], None, q, t)
        qg, tg, sg, _ = gm.process(fr["corner"], fr["surf"], None, q, t)
        assert sg.n_corner_map == so.n_corner_map and sg.n_surf_map == so.n_surf_map, step
        assert max(np.abs(qg - qo).max(), np.abs(tg - to).max()) <= 1e-7, step
        for which in (0, 1):
            mo, mg = _sorted_rows(om.export(which)), _sorted_rows(gm.export(which))
            assert mo.shape == mg.shape, (step, which)
    gm.close()


def test_merge_insert_equals_full_sort(S, stage_ab):
    """The map insertion (:738-802) has two device paths: a full sort of the map pool and, while the cube window stays put, a
    merge of the scan's sorted points into the already sorted map.  Both must give the same map, bit for bit, hence the same poses."""
    a = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=2000000)
    b = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=2000000)
    b.set_merge_insert(False)
    used = 0
    frames = list(stage_ab) + [stage_ab[-1], stage_ab[-1]]  # a repeated scan: every new voxel run joins an old point
    for k, fr in enumerate(frames):
        qa, ta, sa, _ = a.process(fr["corner"], fr["surf"], None, fr["q"], fr["t"])
        qb, tb, sb, _ = b.process(fr["corner"], fr["surf"], None, fr["q"], fr["t"])
        assert sb.insert_path == 0
        used += sa.insert_path
        assert np.array_equal(qa, qb) and np.array_equal(ta, tb), k
        assert sa.n_map_corner_total == sb.n_map_corner_total and sa.n_map_surf_total == sb.n_map_surf_total, k
        for which in (0, 1):
            ma, mb = _sorted_rows(a.export(which)), _sorted_rows(b.export(which))  # export order is not defined
            assert ma.shape == mb.shape and np.array_equal(ma, mb), (k, which)
    assert used >= len(frames) - 3, used  # first scan and window moves take the full sort
    a.close(), b.close()
