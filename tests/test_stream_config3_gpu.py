"""BASELINE config #3: MulRan-like OS1-64 stream through the WHOLE path (A -> B -> C, ScanContext on keyframes), launch values of
launch/aloam_mulran.launch (scan_line 64, lidar_type OS1-64, minimum_range 0.5, mapping resolutions 0.4 / 0.8, keyframe gap 1 m /
10 deg, sc_dist_thres 0.2, sc_max_radius 80), HIP path against the oracle chain scan by scan.

The OS1-64 decoder of the reference (scanRegistration.cpp:205-213, 2 deg ring spacing) folds the 64 beams into ~18 pseudo-rings
with ~80 flat points per scan: a very different walk through stage B's ring windows and stage C's gates than HDL-64.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_stream(O, S, scans, n_db=40):
    reg = S.ScanRegistration(S.OS1_64, 0.5, max_points=200000)
    od = S.LaserOdometry(max_points=200000)
    mp = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=2000000)
    sc = S.SCManager(max_radius=80.0, dist_thres=0.2)
    oo, om, osc = O.Odometry(), O.Mapper(0.4, 0.8), O.SCManager(max_radius=80.0, dist_thres=0.2)
    rng = np.random.default_rng(31)
    for _ in range(n_db):  # so that detectLoopClosureID really searches (>= 31 keyframes, Scancontext.cpp:346-350)
        d = rng.uniform(-2, 18, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5)
        sc.saveScancontextAndKeys(d)
        osc.saveScancontextAndKeys(d)
    from scaloam.pgo import KeyframeGate
    gate_g, gate_o = KeyframeGate(1.0, 10.0), KeyframeGate(1.0, 10.0)  # keyframe_meter_gap 1.0 (aloam_mulran.launch:15), 10 deg
    worst = dict(odom=0.0, map=0.0, sc=0.0)
    n_key = 0
    for k, xyz in enumerate(scans):
        g = reg.laserCloudHandler(xyz)
        o = O.features(xyz, O.OS1_64, 0.5)
        for key in ("sharp", "less_sharp", "flat"):
            assert np.array_equal(g[key], o[key]), (k, key)
        assert np.array_equal(g["less_flat"].view(np.uint32), o["less_flat"].view(np.uint32)), k
        qlc, tlc, qw, tw, sg = od.step_features(reg)
        c = o["cloud"]
        a = oo.step(c[o["sharp"]], c[o["less_sharp"]], c[o["flat"]], o["less_flat"])
        assert list(sg.n_edge) == list(a[4].n_edge) and list(sg.n_plane) == list(a[4].n_plane), (k, list(sg.n_edge), list(a[4].n_edge))
        worst["odom"] = max(worst["odom"], np.abs(qw - a[2]).max(), np.abs(tw - a[3]).max())
        qm, tm, ms = mp.process_features(reg, qw, tw)
        qo, to, so, _ = om.step(c[o["less_sharp"]], o["less_flat"], c, a[2], a[3])
        assert ms.solved == so.solved and list(ms.n_edge) == list(so.n_edge) and list(ms.n_plane) == list(so.n_plane), k
        assert list(ms.lm_iters) == list(so.lm_iters), k
        worst["map"] = max(worst["map"], np.abs(qm - qo).max(), np.abs(tm - to).max())
        kg, ko = gate_g(qm, tm), gate_o(qo, to)
        assert kg == ko, k
        if kg:  # process_pg: VoxelGrid 0.4 + makeAndSaveScancontextAndKeys (:629-639), then detectLoopClosureID (:718)
            n_key += 1
            sc.insert_features(reg)
            ds, _ = O.voxel_grid(c, 0.4)
            osc.makeAndSaveScancontextAndKeys(ds)
            rg, ro = sc.detectLoopClosureID(), osc.detectLoopClosureID()
            assert rg["loop_id"] == ro["loop_id"] and rg["nn_idx"] == ro["nn_idx"] and rg["yaw"] == ro["yaw"], k
            worst["sc"] = max(worst["sc"], abs(rg["min_dist"] - ro["min_dist"]))
    for x in (reg, od, mp, sc):
        x.close()
    return worst, n_key


def test_os1_64_stream_matches_oracle(O, S, worlds):
    """14 scans of the seeded OS1-64 sequence (seed 301): feature indices bit-exact, residual-block counts and
    LM iterations equal, poses within 1e-5 (observed ~1e-12), keyframe decisions equal, loop answers equal, SC distance within 1e-5."""
    w = worlds(O.OS1_64, 301)
    worst, n_key = _run_stream(O, S, [w.scan(k) for k in range(14)])
    print("OS1-64 stream, worst differences:", worst, "keyframes:", n_key)
    assert n_key >= 2
    assert worst["odom"] <= 1e-5 and worst["map"] <= 1e-5 and worst["sc"] <= 1e-5


def test_kaist03_keyframes_stream_matches_oracle(O, S, golden):
    """The reference's own KAIST03 keyframe scans (real OS1-64 data, utils/sample_data/KAIST03; xyz of keyframes 0, 5, 7, 20) as a short
    stream through the same chain.  The keyframes are metres apart, so stage B starts far from its solution: still the same numbers."""
    import os
    import scaloam.formats as F
    from conftest import GOLDEN
    scans = [golden("KAIST03_000000.npy")[:, :3], F.read_pcd(os.path.join(GOLDEN, "KAIST03_000005.pcd"))[:, :3],
             golden("KAIST03_000007.npy")[:, :3], golden("KAIST03_000020.npy")[:, :3]]
    worst, n_key = _run_stream(O, S, [np.ascontiguousarray(s, np.float32) for s in scans])
    print("KAIST03 keyframes, worst differences:", worst, "keyframes:", n_key)
    assert worst["odom"] <= 1e-5 and worst["map"] <= 1e-5 and worst["sc"] <= 1e-5
