"""scal_pipeline (the four stages scheduled inside the library) and the C++ replay host that drives the same C-ABI.

Bar: the pipelined schedule changes WHEN work is queued, never what is computed - poses, odometry poses, residual-block counts and
loop answers must be bit-identical to one scan at a time through the per-stage calls (which the other GPU tests hold to the oracle)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAY = os.path.join(ROOT, "sc-a-loam_amd", "bin", "replay_main")


def _descs(n, seed=7):
    rng = np.random.default_rng(seed)
    return [rng.uniform(-2.0, 18.0, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5) for _ in range(n)]


def _serial(S, scans, descs, sc_thres=0.4):
    """one scan at a time through the per-stage entry points"""
    cap = max(s.shape[0] for s in scans) + 1024
    reg = S.ScanRegistration(S.HDL64, 5.0, max_points=cap)
    od, mp = S.LaserOdometry(max_points=cap), S.LaserMapping(0.4, 0.8, max_scan_points=cap, max_map_points=3000000)
    sc = S.SCManager(max_radius=80.0, dist_thres=sc_thres, max_keyframes=len(descs) + len(scans) + 8)
    for d in descs:
        sc.saveScancontextAndKeys(d)
    out = []
    for xyz in scans:
        reg.laserCloudHandler(xyz)
        _, _, qo, to, ost = od.step_features(reg)
        q, t, mst = mp.process_features(reg, qo, to)
        sc.insert_features(reg)
        r = sc.detectLoopClosureID()
        out.append(dict(q=q, t=t, q_odom=qo, t_odom=to, n_edge=list(mst.n_edge), n_plane=list(mst.n_plane), iters=list(mst.lm_iters), loop=r))
    maps = [mp.export(0), mp.export(1)]
    for x in (reg, od, mp, sc):
        x.close()
    return out, maps


def _sorted_rows(a):
    v = np.ascontiguousarray(a, np.float32).view(np.uint32).reshape(-1, 4)
    return v[np.lexsort((v[:, 3], v[:, 2], v[:, 1], v[:, 0]))]


@pytest.mark.parametrize("ahead", [0, 4])
def test_pipeline_equals_stage_calls(S, hdl64_stream, ahead):
    """ahead = 0: push, pop, push, pop (nothing overlaps).  ahead = 4: four scans pushed beyond the awaited one, so stage A runs
    scans ahead of stage C, stage-C steps queue behind each other on the device and ScanContext runs beside both."""
    n = 18
    scans = [hdl64_stream(k) for k in range(n)]
    descs = _descs(40)
    ref, ref_maps = _serial(S, scans, descs)
    cap = max(s.shape[0] for s in scans) + 1024
    p = S.Pipeline(S.HDL64, 5.0, max_points=cap, max_map_points=3000000, sc_mode=S.SC_EVERY_SCAN, sc_dist_thres=0.4, sc_max_keyframes=len(descs) + n + 8)
    for d in descs:
        p.sc.saveScancontextAndKeys(d)
    got = []
    for k in range(n):
        p.push(scans[k])
        while p.in_flight() > ahead:
            got.append(p.pop())
    p.drain()
    while p.in_flight():
        got.append(p.pop())
    assert [g["seq"] for g in got] == list(range(n))
    for k in range(n):
        g, r = got[k], ref[k]
        assert np.array_equal(g["q_odom"], r["q_odom"]) and np.array_equal(g["t_odom"], r["t_odom"]), k
        assert np.array_equal(g["q"], r["q"]) and np.array_equal(g["t"], r["t"]), (k, np.abs(g["t"] - r["t"]).max())
        assert list(g["map"].n_edge) == r["n_edge"] and list(g["map"].n_plane) == r["n_plane"] and list(g["map"].lm_iters) == r["iters"], k
        for key in ("loop_id", "nn_idx", "nn_shift", "min_dist"):
            assert g["loop"][key] == r["loop"][key] or (np.isnan(g["loop"][key]) and np.isnan(r["loop"][key])), (k, key)
        assert np.array_equal(g["loop"]["cand"], r["loop"]["cand"]), k
    for which in (0, 1):
        assert np.array_equal(_sorted_rows(p.map.export(which)), _sorted_rows(ref_maps[which])), which
    p.close()


def test_pipeline_device_input_and_errors(S, hdl64_stream):
    import torch
    scans = [hdl64_stream(k) for k in range(8)]
    cap = max(s.shape[0] for s in scans) + 1024
    ref, _ = _serial(S, scans, [])
    p = S.Pipeline(S.HDL64, 5.0, max_points=cap, max_map_points=3000000, sc_mode=S.SC_OFF)
    with pytest.raises(S.ScalError) as e:
        p.pop()
    assert e.value.code == S.E_STATE
    with pytest.raises(S.ScalError) as e:
        p.push(np.zeros((cap + 1, 3), np.float32))
    assert e.value.code == S.E_TOO_MANY
    d = [torch.from_numpy(s).cuda() for s in scans]
    torch.cuda.synchronize()
    for k in range(8):
        p.push_device(d[k].data_ptr(), scans[k].shape[0], 3)
    p.drain()
    for k in range(8):
        g = p.pop()
        assert g["loop"] is None and np.array_equal(g["q"], ref[k]["q"]) and np.array_equal(g["t"], ref[k]["t"]), k
    p.close()


@pytest.mark.parametrize("mode", ["serial", "pipeline", "integrated"])
def test_cpp_replay_host_matches_python_path(S, hdl64_stream, tmp_path, mode):
    """host/replay_main.cpp - C++, includes only include/scaloam_hip.h, links libscaloam_hip.so - as a fresh child process.
    `integrated` issues the synchronous host-array calls of INTEGRATION.md sections 1-4 from one thread per stage; `pipeline` drives
    scal_pipeline.  Both must print the poses and loop ids the Python path gets for the same scans."""
    from scaloam import formats
    assert os.path.exists(REPLAY), "sc-a-loam_amd/bin/replay_main is not built (make -C sc-a-loam_amd)"
    n = 14
    scans = [hdl64_stream(k) for k in range(n)]
    ref, _ = _serial(S, scans, [])
    f = str(tmp_path / "scans.bin")
    formats.write_scan_stream(f, scans)
    poses = str(tmp_path / "poses.txt")
    r = subprocess.run([REPLAY, "--scans", f, "--mode", mode, "--warmup", "3", "--poses", poses, "--sc-db", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["mode"] == mode and line["scans"] == n - 3 and line["scans_per_s"] > 0
    rows = np.loadtxt(poses)
    assert rows.shape == (n, 9)
    for k in range(n):
        got = rows[k, 1:8]
        want = np.concatenate([ref[k]["q"], ref[k]["t"]])
        assert np.array_equal(got, want), (mode, k, np.abs(got - want).max())
        assert int(rows[k, 8]) == ref[k]["loop"]["loop_id"], (mode, k)


@pytest.mark.parametrize("n_seqs", [2, 4])
def test_multi_sequence_batched_launches_equal_solo_runs(S, worlds, n_seqs):
    """scal_pipeline_create_multi: S independent sequences (different seeded worlds) stepping together, their kernels sharing launches
    (every hot-path kernel takes up to four argument sets, blockIdx.z selects one; the per-stage calls are recorded per sequence and
    zipped, csrc/batch.hpp).  Every sequence must get, bit for bit, the poses, odometry poses, block counts, loop answers and map it
    gets when it runs alone."""
    import torch
    n = 12
    seqs = [[worlds(S.HDL64, 205 + 1000 * q).scan(k) for k in range(n)] for q in range(n_seqs)]
    cap = max(s.shape[0] for sq in seqs for s in sq) + 1024
    descs = _descs(40)
    solo, solo_maps = [], []
    for q in range(n_seqs):
        p = S.Pipeline(S.HDL64, 5.0, max_points=cap, max_map_points=3000000, sc_mode=S.SC_EVERY_SCAN, sc_dist_thres=0.4, sc_max_keyframes=len(descs) + n + 8)
        for d in descs:
            p.sc.saveScancontextAndKeys(d)
        got = []
        for k in range(n):
            p.push(seqs[q][k])
            while p.in_flight() > 3:
                got.append(p.pop())
        p.drain()
        while p.in_flight():
            got.append(p.pop())
        solo.append(got)
        solo_maps.append([_sorted_rows(p.map.export(w)) for w in (0, 1)])
        p.close()
    d_scans = [[torch.from_numpy(s).cuda() for s in sq] for sq in seqs]
    torch.cuda.synchronize()
    m = S.Pipeline(S.HDL64, 5.0, max_points=cap, max_map_points=3000000, sc_mode=S.SC_EVERY_SCAN, sc_dist_thres=0.4, sc_max_keyframes=len(descs) + n + 8,
                   n_seqs=n_seqs)
    for q in range(n_seqs):
        for d in descs:
            m.scs[q].saveScancontextAndKeys(d)
    got = []
    for k in range(n):
        m.push_device_multi([d_scans[q][k].data_ptr() for q in range(n_seqs)], [seqs[q][k].shape[0] for q in range(n_seqs)])
        while m.in_flight() > 3:
            got.append(m.pop_multi())
    m.drain()
    while m.in_flight():
        got.append(m.pop_multi())
    assert len(got) == n
    for k in range(n):
        for q in range(n_seqs):
            g, r = got[k][q], solo[q][k]
            assert np.array_equal(g["q_odom"], r["q_odom"]) and np.array_equal(g["t_odom"], r["t_odom"]), (k, q)
            assert np.array_equal(g["q"], r["q"]) and np.array_equal(g["t"], r["t"]), (k, q, np.abs(g["t"] - r["t"]).max())
            assert list(g["map"].n_edge) == list(r["map"].n_edge) and list(g["map"].n_plane) == list(r["map"].n_plane), (k, q)
            for key in ("loop_id", "nn_idx", "nn_shift"):
                assert g["loop"][key] == r["loop"][key], (k, q, key)
            assert g["loop"]["min_dist"] == r["loop"]["min_dist"] or (np.isnan(g["loop"]["min_dist"]) and np.isnan(r["loop"]["min_dist"])), (k, q)
    for q in range(n_seqs):
        for w in (0, 1):
            assert np.array_equal(_sorted_rows(m.maps[q].export(w)), solo_maps[q][w]), (q, w)
    m.close()
