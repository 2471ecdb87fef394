"""On-disk formats against files the reference itself wrote: a keyframe PCD saved by
pcl::io::savePCDFileBinary, the first lines of optimized_poses.txt and times.txt (utils/sample_data/KAIST03)."""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


def _fmt():
    import importlib.util
    p = os.path.join(HERE, "..", "sc-a-loam_amd", "python", "scaloam", "formats.py")
    spec = importlib.util.spec_from_file_location("scaloam_formats", p)  # no GPU library needed for this module
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pcd_binary_roundtrip_is_byte_exact(tmp_path):
    F = _fmt()
    src = os.path.join(G, "KAIST03_000005.pcd")
    a = F.read_pcd(src)
    assert a.shape == (35976, 4) and a.dtype == np.float32
    assert np.isfinite(a).all() and np.abs(a[:, :3]).max() < 300
    raw = open(src, "rb").read()
    assert F.pcd_binary_bytes(a) == raw  # header, packed records and PCL's zero tail
    out = tmp_path / "k.pcd"
    F.write_pcd_binary(out, a)
    assert np.array_equal(F.read_pcd(out).view(np.uint32), a.view(np.uint32))
    # empty cloud and ascii variant
    F.write_pcd_binary(out, np.zeros((0, 4), np.float32))
    assert F.read_pcd(out).shape == (0, 4)
    asc = tmp_path / "a.pcd"
    asc.write_text("VERSION 0.7\nFIELDS intensity x y z\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n"
                   "7 1 2 3\n9 4 5 6\n")
    assert np.array_equal(F.read_pcd(asc), np.array([[1, 2, 3, 7], [4, 5, 6, 9]], np.float32))


def test_pose_and_time_text_roundtrip():
    F = _fmt()
    src = os.path.join(G, "KAIST03_optimized_poses_head5.txt")
    lines = open(src).read().splitlines()
    P = F.read_poses(src)
    assert P.shape == (5, 12)
    assert [F.format_pose_line(p) for p in P] == lines  # C++ default formatting reproduced
    R = P[1].reshape(3, 4)[:, :3]
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5
    ts = open(os.path.join(G, "KAIST03_times_head5.txt")).read().splitlines()
    t = F.read_times(os.path.join(G, "KAIST03_times_head5.txt"))
    assert [F.format_time(v) for v in t] == ts


def test_scd_and_kitti_bin(tmp_path):
    # PARITY UNPINNED: the reference ships no .scd and no KITTI .bin file (utils/sample_data holds PCDs, poses and times only), so
    # these two formats are checked against their definition in the source text (Eigen IOFormat(3), saveSCD :178-191; four
    # float32 per point, kittiHelper.cpp:140-150) and for a round trip - not against a file the reference wrote.
    F = _fmt()
    rng = np.random.default_rng(0)
    d = rng.uniform(-2, 18, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5)
    txt = F.scd_text(d)
    assert txt.count("\n") == 19 and not txt.endswith("\n")
    back = F.read_scd(txt)
    assert back.shape == (20, 60) and np.abs(back - d).max() <= 0.05 + 1e-9  # three significant digits
    assert F.scd_text(np.array([[0.0, 12.3456, -0.001234, 100000.0]])) == "0 12.3 -0.00123 1e+05"
    p = tmp_path / "v.bin"
    a = rng.normal(size=(1000, 4)).astype(np.float32)
    F.write_kitti_bin(p, a)
    assert np.array_equal(F.read_kitti_bin(p), a)


def test_keyframe_gate_and_g2o(tmp_path):
    """Host glue of the pose-graph node (scaloam/pgo.py): the keyframe gate of laserPosegraphOptimization.cpp:598-617 and the g2o
    text of :147-216.  The reference ships no pose-graph file, so the text format is pinned by its std::to_string construction only."""
    from scaloam.pgo import KeyframeGate, write_g2o, read_g2o, g2o_vertex, quat_from_rpy, rpy_from_quat
    g = KeyframeGate(meter_gap=1.0, deg_gap=10.0)
    q0 = np.array([0.0, 0.0, 0.0, 1.0])
    assert g(q0, np.zeros(3))                      # accumulators start at 1e6: the first pose is a keyframe
    assert not g(q0, np.array([0.4, 0.0, 0.0]))
    assert not g(q0, np.array([0.8, 0.0, 0.0]))
    assert g(q0, np.array([1.3, 0.0, 0.0]))        # 0.4 + 0.4 + 0.5 > 1.0, accumulators reset
    assert not g(q0, np.array([1.4, 0.0, 0.0]))
    yaw = np.deg2rad(11.0)
    assert g(np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)]), np.array([1.4, 0.0, 0.0]))  # rotation alone passes 10 deg
    r, p, y = rpy_from_quat(quat_from_rpy(0.1, -0.2, 0.3))
    assert abs(r - 0.1) < 1e-12 and abs(p + 0.2) < 1e-12 and abs(y - 0.3) < 1e-12
    assert g2o_vertex(3, (1.0, 2.5, -0.25), (0.0, 0.0, 0.0, 1.0)) == "VERTEX_SE3:QUAT 3 1.000000 2.500000 -0.250000 0.000000 0.000000 0.000000 1.000000"
    poses = [(0, 0, 0, 0, 0, 0), (1.0, 0.5, 0.0, 0.0, 0.0, 0.2)]
    edges = [(0, 1, (1.0, 0.5, 0.0), quat_from_rpy(0, 0, 0.2))]
    f = tmp_path / "singlesession_posegraph.g2o"
    write_g2o(str(f), poses, edges)
    V, E = read_g2o(str(f))
    assert len(V) == 2 and len(E) == 1 and E[0][:2] == (0, 1)
    assert np.abs(V[1][0] - [1.0, 0.5, 0.0]).max() < 1e-6 and np.abs(V[1][1] - quat_from_rpy(0, 0, 0.2)).max() < 1e-6
