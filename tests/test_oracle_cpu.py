"""CPU tests of the oracle (the checker): what pins it to the reference, cross-checks of the restated third-party
arithmetic against numpy/scipy, and regression vectors.  No GPU needed."""
import os

import numpy as np
import pytest


# ------------------------------------------------------------------ pinned by the reference's own data / code
@pytest.mark.parametrize("name", ["KAIST03_000000.npy", "KAIST03_000007.npy", "KAIST03_000020.npy"])
@pytest.mark.parametrize("float_math", [0, 1])
def test_kaist03_ring_ids_and_order(O, golden, name, float_math):
    """utils/sample_data/KAIST03 keyframes are stage-A outputs of the reference (ring-major, intensity = scanID +
    0.1*relTime): recomputing the OS1-64 scanID from xyz must reproduce round(intensity) for every
    point, the ring-major order must be the identity, and relTime must stay in the observed range."""
    a = golden(name)
    f = O.features(a[:, :3], O.OS1_64, 0.5, float_math=float_math)
    assert f["rc"] == 0 and f["n_kept"] == a.shape[0]
    assert np.array_equal(f["src_index"], np.arange(a.shape[0]))  # already ring-major => stable reorder is the identity
    assert np.array_equal(np.round(a[:, 3]).astype(int), np.round(f["cloud"][:, 3]).astype(int))
    ring = np.round(f["cloud"][:, 3]).astype(int)
    assert (np.diff(ring) >= 0).all() and ring.min() >= 3 and ring.max() <= 20  # only pseudo-rings 3..20 are populated
    # relTime as the reference stored it (the recomputed one is not comparable: start/end azimuth come from the first and
    # last point of the cloud, and this file is already ring-major): observed range -3.8e-5 .. 1.003
    rel = (a[:, 3].astype(np.float64) - np.round(a[:, 3])) / 0.1
    assert rel.min() > -0.01 and rel.max() < 1.01
    # feature counts the survey measured on these scans: 216 sharp, ~2.1k lessSharp, 80-85 flat, ~22.9k lessFlat
    assert len(f["sharp"]) == 216 and 2080 <= len(f["less_sharp"]) <= 2120 and 55 <= len(f["flat"]) <= 110
    assert 22000 <= f["less_flat"].shape[0] <= 24000


def test_all_kaist03_sample_scans_pin_ring_ids_and_order(O):
    """The wide form of the pin above, over EVERY keyframe the reference ships under utils/sample_data/KAIST03/Scans (21 files,
    765,919 points - the count SURVEY.md section 8c measured): OS1-64 scanID of every point reproduced from xyz, ring-major
    order = identity.  Reads the reference's data files where they lie (build container only; skipped on the GPU box, where
    /root/reference does not exist); nothing from the reference is copied or shipped.  Seosan01's 21 keyframes cannot serve:
    their fourth channel is the sensor's raw intensity (integers 3..7156), not scanID + 0.1 * relTime."""
    import glob
    from scaloam import formats
    files = sorted(glob.glob("/root/reference/utils/sample_data/KAIST03/Scans/*.pcd"))
    if not files:
        pytest.skip("/root/reference/utils/sample_data is not on this host")
    assert len(files) == 21
    total = matched = 0
    for f in files:
        a = formats.read_pcd(f)
        r = O.features(a[:, :3], O.OS1_64, 0.5)
        assert r["rc"] == 0 and r["n_kept"] == a.shape[0], f
        assert np.array_equal(r["src_index"], np.arange(a.shape[0])), f
        matched += int((np.round(a[:, 3]).astype(int) == np.round(r["cloud"][:, 3]).astype(int)).sum())
        total += a.shape[0]
    assert total == 765919 and matched == total


def test_ringkey_knn_matches_vendored_nanoflann(O):
    """D5: the oracle's brute-force f32 ring-key KNN vs the reference's own nanoflann (compiled into oracle/_ref)."""
    rng = np.random.default_rng(401)
    keys = rng.uniform(0.03, 4.4, (5000, 20)).astype(np.float32)
    q = rng.uniform(0.03, 4.4, (64, 20)).astype(np.float32)
    ref = O.ref_ringkey_knn(keys, q, 3)
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this host)")
    idx, d = ref
    for i in range(q.shape[0]):
        # nanoflann L2_Adaptor accumulation order (nanoflann.hpp:383-408)
        diff = (q[i][None, :] - keys).astype(np.float32)
        sq = (diff * diff).astype(np.float32)
        acc = np.zeros(keys.shape[0], np.float32)
        for g in range(0, 20, 4):
            acc = (acc + (((sq[:, g] + sq[:, g + 1]) + sq[:, g + 2]) + sq[:, g + 3])).astype(np.float32)
        order = np.lexsort((np.arange(keys.shape[0]), acc))[:3]
        assert np.array_equal(order, idx[i]), i
        assert np.array_equal(acc[order].view(np.uint32), d[i].view(np.uint32)), i


def test_sc_detect_candidates_match_nanoflann(O):
    """The oracle's detectLoopClosureID candidates (stale tree, newest 30 excluded) vs nanoflann on the same key set."""
    rng = np.random.default_rng(9)
    m = O.SCManager()
    descs = [rng.uniform(-2, 18, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5) for _ in range(95)]
    keys = []
    size_at_rebuild, counter = 0, 0
    for i, d in enumerate(descs):
        m.saveScancontextAndKeys(d)
        keys.append(m.get(i)[1])
        r = m.detectLoopClosureID()
        if i + 1 < 31:
            assert r["loop_id"] == -1
            continue
        if counter % 30 == 0:
            size_at_rebuild = i + 1
        counter += 1
        ref = O.ref_ringkey_knn(np.stack(keys[: size_at_rebuild - 30]), keys[i][None, :], 3)
        if ref is None:
            pytest.skip("oracle/_ref not built")
        n_tree = size_at_rebuild - 30
        assert np.array_equal(r["cand"][: min(3, n_tree)], ref[0][0][: min(3, n_tree)]), i


# ------------------------------------------------------------------ restated third-party arithmetic vs numpy / scipy
def test_voxel_grid_vs_numpy(O, golden):
    a = golden("Seosan01_000011.npy")
    for leaf in (0.2, 0.4, 0.8):
        out, guard = O.voxel_grid(a, leaf, order_mode=1)
        inv = np.float32(1.0) / np.float32(leaf)
        ijk = np.floor(a[:, :3] * inv).astype(np.int64)
        ijk -= ijk.min(0)
        dims = ijk.max(0) + 1
        idx = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
        uniq, invx, cnt = np.unique(idx, return_inverse=True, return_counts=True)
        assert guard == 0 and out.shape[0] == uniq.shape[0]
        mean = np.zeros((uniq.shape[0], 4))
        np.add.at(mean, invx, a.astype(np.float64))
        mean /= cnt[:, None]
        assert np.abs(out[:, :3] - mean[:, :3]).max() < 2e-4  # f32 sequential sums vs f64 means
        assert np.abs(out[:, 3] - mean[:, 3]).max() < 2e-6 * max(1.0, np.abs(mean[:, 3]).max()) * 50
    # literal std::sort order vs pinned (voxel, arrival) order: same voxels, centroids equal to f32 rounding
    o0, _ = O.voxel_grid(a, 0.4, order_mode=0)
    o1, _ = O.voxel_grid(a, 0.4, order_mode=1)
    assert o0.shape == o1.shape and np.abs(o0[:, :3] - o1[:, :3]).max() < 1e-4


def test_eigen_and_plane_fit_vs_numpy(O):
    rng = np.random.default_rng(2)
    for _ in range(200):
        P = rng.normal(size=(5, 3)) * rng.uniform(0.01, 1.0, 3) + rng.uniform(-100, 100, 3)
        c = P.mean(0)
        M = (P - c).T @ (P - c)
        w, V = O.eig3_sym(M)
        w0, V0 = np.linalg.eigh(M)
        assert np.abs(w - w0).max() <= 1e-12 * max(1.0, np.abs(w0).max())
        assert abs(abs(V[:, 2] @ V0[:, 2]) - 1) < 1e-9
        x = O.plane_fit(P, -np.ones(5))
        x0 = np.linalg.lstsq(P, -np.ones(5), rcond=None)[0]
        assert np.abs(x - x0).max() <= 1e-6 * max(1e-3, np.abs(x0).max())


def test_kdtree_vs_scipy(O, golden):
    """The mapper with kd-tree kNN and with brute-force kNN must produce identical steps (exact search)."""
    from scipy.spatial import cKDTree  # noqa: F401  (scipy present: cross-check of the 5-NN sets below)
    a = golden("KAIST03_000000.npy")
    b = golden("KAIST03_000007.npy")
    res = []
    for mode in (0, 1):
        od, mp = O.Odometry(), O.Mapper(0.4, 0.8, knn_mode=mode)
        for s in (a, b):
            f = O.features(s[:, :3], O.OS1_64, 0.5)
            c = f["cloud"]
            x = od.step(c[f["sharp"]], c[f["less_sharp"]], c[f["flat"]], f["less_flat"])
            q, t, st, _ = mp.step(c[f["less_sharp"]], f["less_flat"], c, x[2], x[3])
        res.append((q, t, list(st.n_edge), list(st.n_plane)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]


def test_autodiff_jacobians_vs_finite_differences(O):
    rng = np.random.default_rng(5)
    x = np.array([0.03, -0.02, 0.05, 0.998, 0.4, -0.1, 0.2])
    for kind in (0, 1, 2):
        cp = rng.uniform(-20, 20, 3)
        pa = cp + rng.normal(0, 0.5, 3)
        pb = pa + rng.normal(0, 0.5, 3)
        if kind == 1:
            pb = pb / np.linalg.norm(pb)
        if kind == 2:
            pa = pa / np.linalg.norm(pa)
        r, J = O.factor_eval(kind, cp, np.concatenate([pa, pb]), x)
        for k in range(7):
            h = 1e-6
            xp, xm = x.copy(), x.copy()
            xp[k] += h
            xm[k] -= h
            rp, _ = O.factor_eval(kind, cp, np.concatenate([pa, pb]), xp)
            rm, _ = O.factor_eval(kind, cp, np.concatenate([pa, pb]), xm)
            assert np.abs((rp - rm) / (2 * h) - J[:, k]).max() < 1e-6


def test_lm_recovers_known_pose(O):
    """ceres_solve stand-in: plane-norm + edge blocks generated from a known pose, started 5 cm / 1 deg away."""
    rng = np.random.default_rng(8)
    ang = np.radians(3.0)
    q_true = np.array([0, 0, np.sin(ang / 2), np.cos(ang / 2)])
    t_true = np.array([0.5, -0.3, 0.1])

    def rot(q, v):
        u = q[:3]
        uv = 2 * np.cross(u, v)
        return v + q[3] * uv + np.cross(u, uv)
    n = 400
    kind = np.zeros(n, np.int32)
    cp = rng.uniform(-30, 30, (n, 3))
    pa = np.zeros((n, 3))
    pb = np.zeros((n, 3))
    for i in range(n):
        w = rot(q_true, cp[i]) + t_true
        if i % 2 == 0:
            kind[i] = 2
            nv = rng.normal(size=3)
            nv /= np.linalg.norm(nv)
            pa[i] = nv
            pb[i, 0] = -(nv @ w)
        else:
            kind[i] = 0
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            pa[i] = w + 0.1 * d
            pb[i] = w - 0.1 * d
    x0 = np.array([0, 0, 0, 1.0, 0.45, -0.25, 0.05])
    x, it, trace, term = O.ceres_solve(kind, cp, pa, pb, x0)
    assert it <= 4 and len(trace) >= 2 and (np.diff(trace) <= 1e-12).all()  # monotone cost
    x2, _, tr2, _ = O.ceres_solve(kind, cp, pa, pb, x)
    assert tr2[-1] < 1e-9
    assert np.abs(x2[:4] - q_true).max() < 1e-5 and np.abs(x2[4:] - t_true).max() < 1e-4
    # no residual blocks: Ceres leaves the parameters untouched
    x3, it3, _, term3 = O.ceres_solve(np.zeros(0, np.int32), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), x0)
    assert np.array_equal(x3, x0) and it3 == 0 and term3 == 4


def test_scancontext_vs_numpy(O, golden):
    """makeScancontext / keys / distanceBtnScanContext against an independent ~30-line numpy ScanContext."""
    a = golden("Seosan01_000000.npy")
    ds, _ = O.voxel_grid(a, 0.4)
    m = O.SCManager(max_radius=80.0)
    d = m.makeScancontext(ds)
    x, y, z = ds[:, 0].astype(np.float64), ds[:, 1].astype(np.float64), ds[:, 2].astype(np.float64) + 2.0
    r = np.hypot(x, y)
    th = (np.degrees(np.arctan2(y, x)) + 360.0) % 360.0
    keep = r <= 80.0
    ring = np.clip(np.ceil(r / 80.0 * 20).astype(int), 1, 20) - 1
    sec = np.clip(np.ceil(th / 360.0 * 60).astype(int), 1, 60) - 1
    ref = np.full((20, 60), -1000.0)
    np.maximum.at(ref, (ring[keep], sec[keep]), z[keep].astype(np.float32).astype(np.float64))
    ref[ref == -1000.0] = 0
    # identical except for points within float rounding of a bin edge
    assert (d != ref).sum() <= 3
    rk, sk = O.sc_keys(d)
    assert np.abs(rk - d.mean(1)).max() < 1e-12 and np.abs(sk - d.mean(0)).max() < 1e-12

    def np_dist(a_, b_):
        va, vb = a_.mean(0), b_.mean(0)
        shift0 = int(np.argmin([np.linalg.norm(va - np.roll(vb, s)) for s in range(60)]))
        best = (1e7, 0)
        for s in sorted((shift0 + k) % 60 for k in range(-3, 4)):
            bs = np.roll(b_, s, axis=1)
            na, nb = np.linalg.norm(a_, axis=0), np.linalg.norm(bs, axis=0)
            ok = (na > 0) & (nb > 0)
            dd = 1 - ((a_ * bs).sum(0)[ok] / (na[ok] * nb[ok])).mean()
            if dd < best[0]:
                best = (dd, s)
        return best
    b = golden("Seosan01_000011.npy")
    d2 = m.makeScancontext(O.voxel_grid(b, 0.4)[0])
    for p, q in ((d, d2), (d2, d), (d, np.roll(d, 17, axis=1))):
        do, so = O.sc_distance(p, q)
        dn, sn = np_dist(p, q)
        assert abs(do - dn) < 1e-12 and so == sn
    do, so = O.sc_distance(d, np.roll(d, 17, axis=1))
    assert do < 1e-12 and so == 43  # column shift that undoes the roll: (60 - 17)


def test_libm_and_sort_variants_are_quantified(O, golden, worlds):
    """cr_libm (correctly rounded atan2f) vs this host's libm, std::sort vs the pinned order: differences counted."""
    a = golden("KAIST03_000020.npy")
    f1 = O.features(a[:, :3], O.OS1_64, 0.5, cr_libm=1, sort_mode=1)
    f0 = O.features(a[:, :3], O.OS1_64, 0.5, cr_libm=0, sort_mode=0, voxel_order=0)
    assert f0["n_ties"] == f1["n_ties"]
    for k in ("sharp", "less_sharp", "flat"):
        assert np.array_equal(f0[k], f1[k])  # no curvature ties on this scan: std::sort cannot reorder anything
    assert np.array_equal(f0["cloud"][:, :3], f1["cloud"][:, :3])
    assert np.abs(f0["cloud"][:, 3] - f1["cloud"][:, 3]).max() <= 1e-5  # libm atan2f within 1 ulp of the rounded value
    assert np.array_equal(f0["cloud"][:, 3].astype(int), f1["cloud"][:, 3].astype(int))


# ------------------------------------------------------------------ regression vectors of the oracle itself
def test_oracle_golden_vectors(O, golden):
    G = golden("oracle_golden.npz")
    od, mp = O.Odometry(), O.Mapper(0.4, 0.8)
    descs = []
    for nm in ["KAIST03_000000", "KAIST03_000007", "KAIST03_000020"]:
        a = golden(nm + ".npy")
        f = O.features(a[:, :3], O.OS1_64, 0.5)
        c = f["cloud"]
        assert f["n_kept"] == int(G[nm + "_n_kept"])
        for k in ("sharp", "less_sharp", "flat"):
            assert np.array_equal(f[k], G[nm + "_" + k])
        assert f["less_flat"].shape[0] == int(G[nm + "_less_flat_n"])
        assert np.allclose(f["less_flat"].astype(np.float64).sum(0), G[nm + "_less_flat_sum"], rtol=0, atol=1e-6)
        x = od.step(c[f["sharp"]], c[f["less_sharp"]], c[f["flat"]], f["less_flat"])
        q, t, st, _ = mp.step(c[f["less_sharp"]], f["less_flat"], c, x[2], x[3])
        assert np.abs(np.concatenate([x[2], x[3]]) - G[nm + "_odom_pose"]).max() <= 1e-9
        assert np.abs(np.concatenate([q, t]) - G[nm + "_map_pose"]).max() <= 1e-9
        assert list(st.n_edge) + list(st.n_plane) == list(G[nm + "_map_blocks"])
        ds, _ = O.voxel_grid(c, 0.4)
        assert ds.shape[0] == int(G[nm + "_ds04_n"])
        descs.append(O.SCManager().makeScancontext(ds))
    for nm in ["Seosan01_000000", "Seosan01_000011"]:
        descs.append(O.SCManager().makeScancontext(O.voxel_grid(golden(nm + ".npy"), 0.4)[0]))
    assert np.array_equal(np.stack(descs), G["sc_descs"])
    for i in range(5):
        for j in range(5):
            d, s = O.sc_distance(descs[i], descs[j])
            assert abs(d - G["sc_dist"][i, j]) <= 1e-12 and s == G["sc_shift"][i, j]


def test_mapmerge_oracle_properties(O, golden):
    """Offline map merge (makeMergedMap.py:83-133) restatement: equals a plain numpy f64 evaluation, keeps the order, drops
    exactly the points within 2 m of the sensor, and a rigid transform preserves point-to-point distances."""
    names = ["Seosan01_000000.npy", "Seosan01_000011.npy"]
    frames = [golden(n) for n in names]
    poses = golden("Seosan01_poses21.npy")[[0, 11]]
    m = O.mapmerge(frames, poses, 2.0)
    ref = []
    for f, p in zip(frames, poses):
        x = f[:, :3].astype(np.float64)
        keep = np.sqrt((x * x).sum(1)) > 2.0
        T = p.reshape(3, 4)
        ref.append(np.hstack([(x[keep] @ T[:, :3].T + T[:, 3]).astype(np.float32), f[keep, 3:4]]))
    ref = np.concatenate(ref)
    assert m.shape == ref.shape
    assert np.abs(m - ref).max() <= 1e-5  # summation order of the 4x4 product differs from numpy's matmul by <= 1 f32 ulp
    assert np.array_equal(m[:, 3], ref[:, 3])
    n0 = int((np.sqrt((frames[0][:, :3].astype(np.float64) ** 2).sum(1)) > 2.0).sum())
    a, b = m[:n0][::997], frames[0][np.sqrt((frames[0][:, :3].astype(np.float64) ** 2).sum(1)) > 2.0][::997]
    da = np.linalg.norm(a[1:, :3].astype(np.float64) - a[:-1, :3], axis=1)
    db = np.linalg.norm(b[1:, :3].astype(np.float64) - b[:-1, :3], axis=1)
    assert np.abs(da - db).max() <= 2e-3  # optimized_poses.txt rotations are orthonormal to ~1e-6
    assert O.mapmerge([], np.zeros((0, 12))).shape == (0, 4)


def test_icp_oracle_recovers_motion_and_gates(O, golden):
    """pcl::IterativeClosestPoint restatement (laserPosegraphOptimization.cpp:518-535): a known rigid motion of a real keyframe
    is recovered, the closed-form increment is a proper rotation, and the gates (iteration cap, too few correspondences,
    correspondence distance) behave as PCL's do."""
    a, _ = O.voxel_grid(golden("Seosan01_000000.npy"), 0.4)
    th = 0.03
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    t = np.array([-0.3, 0.2, 0.1])
    src = a.copy()
    src[:, :3] = (a[:, :3].astype(np.float64) @ R.T + t).astype(np.float32)
    r = O.icp_align(src, a)
    Ti = np.eye(4)
    Ti[:3, :3], Ti[:3, 3] = R.T, -R.T @ t
    assert r["converged"] and r["state"] in (2, 3, 4) and r["iterations"] < 40
    assert np.abs(r["T"] - Ti).max() <= 1e-5 and r["fitness"] <= 1e-9
    Rf = r["T"][:3, :3]
    assert abs(np.linalg.det(Rf) - 1.0) <= 1e-5 and np.abs(Rf @ Rf.T - np.eye(3)).max() <= 1e-5
    capped = O.icp_align(src, a, max_iter=1)
    assert capped["converged"] and capped["state"] == 1 and capped["iterations"] == 1
    few = O.icp_align(src[:2], a)
    assert not few["converged"] and few["state"] == 5
    far = O.icp_align(src + np.array([1000, 0, 0, 0], np.float32), a, max_corr=1.0)
    assert not far["converged"] and far["state"] == 5


def test_literal_vs_pinned_variants_over_a_pose_stream(O, hdl64_stream):
    """The GPU parity tests compare with the oracle in its PINNED variant (correctly rounded atan2, (key, index) sort order, stable
    voxel order: the only choices that do not depend on the C library / libstdc++ of the build).  This test quantifies what each
    LITERAL choice does to a whole pose stream: 20 HDL-64 scans of the BASELINE config #2 sequence through A -> B -> C.
      * glibc atan2f, std::sort on the bare curvature: identical feature index lists and poses on this stream;
      * PCL's unstable sort inside stage A's per-ring VoxelGrid: poses within 1e-7;
      * PCL's unstable sort inside the MAP's per-cube VoxelGrid (laserMapping.cpp:793-801): centroids of voxels whose members
        arrive in a different order move by up to ~1e-4 m, stay harmless for ~10 scans (1e-7) and then flip one of stage C's
        gates (d^2 < 1, lambda2 > 3 lambda1, 0.2 m): the mapping pose moves by 1e-4 .. 6e-4 m.  That is ABOVE north_star's 1e-5:
        the reference's own trajectory depends on libstdc++'s introsort tie order at that level, and 'parity' can only mean parity
        with one fixed order.  The bound asserted here documents the size of that effect."""
    n = 20

    def chain(kw, vo):
        od, mp = O.Odometry(), O.Mapper(0.4, 0.8, voxel_order=vo)
        poses, feats = [], []
        for k in range(n):
            f = O.features(hdl64_stream(k), O.HDL64, 5.0, **kw)
            c = f["cloud"]
            qlc, tlc, qw, tw, _ = od.step(c[f["sharp"]], c[f["less_sharp"]], c[f["flat"]], f["less_flat"])
            qm, tm, st, _ = mp.step(c[f["less_sharp"]], f["less_flat"], c, qw, tw)
            poses.append(np.concatenate([qw, tw, qm, tm]))
            feats.append((f["sharp"].copy(), f["less_sharp"].copy(), f["flat"].copy()))
        return np.array(poses), feats

    base, fbase = chain(dict(cr_libm=1, sort_mode=1, voxel_order=1), 1)
    report = {}
    for name, kw, vo in (("glibc atan2f", dict(cr_libm=0, sort_mode=1, voxel_order=1), 1),
                         ("std::sort on curvature", dict(cr_libm=1, sort_mode=0, voxel_order=1), 1),
                         ("stage-A voxel order", dict(cr_libm=1, sort_mode=1, voxel_order=0), 1),
                         ("map voxel order", dict(cr_libm=1, sort_mode=1, voxel_order=1), 0)):
        p, f = chain(kw, vo)
        same = sum(all(np.array_equal(x, y) for x, y in zip(a, b)) for a, b in zip(f, fbase))
        report[name] = (same, np.abs(p[:, :7] - base[:, :7]).max(), np.abs(p[:, 7:] - base[:, 7:]).max())
    for k, v in report.items():
        print(f"literal '{k}': feature lists equal on {v[0]}/{n} scans, max |d pose| odometry {v[1]:.2e}, mapping {v[2]:.2e}")
    for k in ("glibc atan2f", "std::sort on curvature"):
        assert report[k][0] == n and report[k][1] == 0.0 and report[k][2] == 0.0
    assert report["stage-A voxel order"][0] == n and max(report["stage-A voxel order"][1:]) <= 1e-6
    assert report["map voxel order"][1] == 0.0 and report["map voxel order"][2] <= 5e-3
