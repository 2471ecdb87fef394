"""Stage B parity, the factor kernel against the oracle's autodiff, and the device-resident A->B->C->D pipeline."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_factor_kernel_vs_autodiff(O, S):
    """k_lm_eval (analytic Jacobians, Huber, J^T J reduction) vs the oracle's forward-mode Jets + Ceres corrector."""
    rng = np.random.default_rng(3)
    n = 700
    kind = rng.integers(0, 3, n).astype(np.int32)
    cp = rng.uniform(-40, 40, (n, 3))
    pa = rng.uniform(-40, 40, (n, 3))
    pb = rng.uniform(-40, 40, (n, 3))
    for i in range(n):
        if kind[i] == 0:
            pa[i] = cp[i] + rng.normal(0, 0.3, 3)
            pb[i] = pa[i] + rng.normal(0, 0.2, 3)
        elif kind[i] == 1:
            pa[i] = cp[i] + rng.normal(0, 0.3, 3)
            v = rng.normal(size=3)
            pb[i] = v / np.linalg.norm(v)
        else:
            v = rng.normal(size=3)
            pa[i] = v / np.linalg.norm(v)
            pb[i] = [-(pa[i] @ cp[i]) + rng.normal(0, 0.2), 0, 0]
    q = np.array([0.02, -0.01, 0.03, 0.999])  # deliberately not unit: the reference never normalises either
    x = np.concatenate([q, [0.3, -0.2, 0.1]])
    cost, g, H = S.factors_eval(kind, cp, pa, pb, x)
    # oracle: per-block autodiff, plus-Jacobian, Huber corrector, sequential accumulation
    P = np.array([[x[3], x[2], -x[1]], [-x[2], x[3], x[0]], [x[1], -x[0], x[3]], [-x[0], -x[1], -x[2]]])
    c0, g0, H0 = 0.0, np.zeros(6), np.zeros((6, 6))
    a = 0.1
    for i in range(n):
        r, J = O.factor_eval(int(kind[i]), cp[i], np.concatenate([pa[i], pb[i]]), x)
        Jl = np.concatenate([J[:, :4] @ P, J[:, 4:]], axis=1)
        s = float(r @ r)
        if s > a * a:
            rho0, rho1 = 2 * a * np.sqrt(s) - a * a, a / np.sqrt(s)
        else:
            rho0, rho1 = s, 1.0
        c0 += 0.5 * rho0
        g0 += rho1 * (Jl.T @ r)
        H0 += rho1 * (Jl.T @ Jl)
    assert abs(cost - c0) <= 1e-10 * c0
    assert np.abs(g - g0).max() <= 1e-9 * np.abs(g0).max()
    assert np.abs(H - H0).max() <= 1e-9 * np.abs(H0).max()


def _stage_a(O, scans):
    out = []
    for xyz in scans:
        f = O.features(xyz, O.HDL64, 5.0)
        c = f["cloud"]
        out.append(dict(sharp=c[f["sharp"]], less_sharp=c[f["less_sharp"]], flat=c[f["flat"]], less_flat=f["less_flat"], cloud=c))
    return out


def test_odometry_stream(O, S, hdl64_stream):
    fa = _stage_a(O, [hdl64_stream(k) for k in range(8)])
    oo = O.Odometry()
    go = S.LaserOdometry(max_points=200000)
    worst = 0.0
    for k, f in enumerate(fa):
        a = oo.step(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
        b = go.step(f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
        so, sg = a[4], b[4]
        assert list(sg.n_edge) == list(so.n_edge) and list(sg.n_plane) == list(so.n_plane), (k, list(sg.n_edge), list(so.n_edge), list(sg.n_plane), list(so.n_plane))
        assert list(sg.lm_iters) == list(so.lm_iters), k
        for o in range(2):
            assert abs(sg.cost_init[o] - so.cost_init[o]) <= 1e-9 * max(1.0, so.cost_init[o]), k
        d = max(np.abs(a[i] - b[i]).max() for i in range(4))
        worst = max(worst, d)
        assert d <= 1e-7, (k, d)
    print("worst odometry pose difference:", worst)
    go.close()


def test_odometry_sparse_targets_reach_every_level_of_the_cell_index(O, S, hdl64_stream):
    """Stage B's NN(1) goes through a two-level hashed cell index (1 m cells, then 5 m cells, then a sweep of the cloud when the nearest
    candidate is not certified within 5 m).  Consecutive dense scans almost always answer at the first level; here every third scan is
    used (3 m of motion against the constant-velocity prior) and the target clouds are thinned to every 9th / 25th point, so that many
    queries find their neighbour between 1 and 5 m, at the 5 m edge or not at all.  Block counts of both outer iterations, LM
    iterations, costs and poses must still follow the oracle's kd-tree."""
    ks = [0, 3, 6, 9, 12, 15]
    fa = _stage_a(O, [hdl64_stream(k) for k in ks])
    for step_c, step_s in ((9, 25), (2, 3)):
        oo = O.Odometry()
        go = S.LaserOdometry(max_points=200000)
        for k, f in enumerate(fa):
            ls, lf = f["less_sharp"][::step_c].copy(), f["less_flat"][::step_s].copy()
            a = oo.step(f["sharp"], ls, f["flat"], lf)
            b = go.step(f["sharp"], ls, f["flat"], lf)
            so, sg = a[4], b[4]
            assert list(sg.n_edge) == list(so.n_edge) and list(sg.n_plane) == list(so.n_plane), (step_c, k, list(sg.n_edge), list(so.n_edge), list(sg.n_plane), list(so.n_plane))
            assert list(sg.lm_iters) == list(so.lm_iters), (step_c, k)
            for o in range(2):
                assert abs(sg.cost_init[o] - so.cost_init[o]) <= 1e-9 * max(1.0, so.cost_init[o]), (step_c, k)
            assert max(np.abs(a[i] - b[i]).max() for i in range(4)) <= 1e-7, (step_c, k)
        print("thinning", step_c, step_s, "blocks of the last scan:", list(sg.n_edge), list(sg.n_plane))
        go.close()


def test_device_resident_pipeline(O, S, hdl64_stream):
    """A -> B -> C with every intermediate left in HBM (scal_*_step_features) must equal the host-array path."""
    reg = S.ScanRegistration(S.HDL64, 5.0, max_points=200000)
    od_dev = S.LaserOdometry(max_points=200000)
    mp_dev = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=2000000)
    od_host = S.LaserOdometry(max_points=200000)
    mp_host = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=2000000)
    oo, om = O.Odometry(), O.Mapper(0.4, 0.8)
    for k in range(6):
        xyz = hdl64_stream(k)
        f = reg.laserCloudHandler(xyz)  # leaves the device-resident results in the context too
        if k % 2 == 1:  # split entry points: B queued first, the pose-independent part of stage C on the side stream meanwhile
            od_dev.enqueue_features(reg)
            mp_dev.prefetch_features(reg)
            qlc, tlc, qw, tw, st = od_dev.collect()
        else:
            qlc, tlc, qw, tw, st = od_dev.step_features(reg)
        qm, tm, sm = mp_dev.process_features(reg, qw, tw)
        c = f["cloud"]
        h = od_host.step(c[f["sharp"]], c[f["less_sharp"]], c[f["flat"]], f["less_flat"])
        qh, th, sh, _ = mp_host.process(c[f["less_sharp"]], f["less_flat"], c, h[2], h[3])
        assert np.array_equal(qw, h[2]) and np.array_equal(tw, h[3]), k
        assert np.array_equal(qm, qh) and np.array_equal(tm, th), k
        # and both equal the oracle chain on the same scan
        fo = O.features(xyz, O.HDL64, 5.0)
        co = fo["cloud"]
        a = oo.step(co[fo["sharp"]], co[fo["less_sharp"]], co[fo["flat"]], fo["less_flat"])
        qo, to, so, _ = om.step(co[fo["less_sharp"]], fo["less_flat"], co, a[2], a[3])
        assert max(np.abs(qm - qo).max(), np.abs(tm - to).max()) <= 1e-6, k
    for x in (reg, od_dev, mp_dev, od_host, mp_host):
        x.close()


def test_odometry_capacity_error(S):
    go = S.LaserOdometry(max_points=1000)
    z = np.zeros((5, 4), np.float32)
    with pytest.raises(S.ScalError) as e:
        go.step(z, z, z, np.zeros((2000, 4), np.float32))
    assert e.value.code == S.E_TOO_MANY
    go.close()


def test_sc_insert_features_matches_host_path(O, S, hdl64_stream):
    """scal_sc_insert_features (ordered cloud -> VoxelGrid 0.4 -> descriptor, all on the GPU) vs the oracle chain, and the
    detector on top of it."""
    reg = S.ScanRegistration(S.HDL64, 5.0, max_points=200000)
    sc = S.SCManager(dist_thres=0.4)
    osc = O.SCManager(dist_thres=0.4)
    rng = np.random.default_rng(3)
    for i in range(40):
        d = rng.uniform(-2, 18, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5)
        sc.saveScancontextAndKeys(d)
        osc.saveScancontextAndKeys(d)
    for k in range(4):
        f = reg.laserCloudHandler(hdl64_stream(k))
        sc.insert_features(reg)
        ds, _ = O.voxel_grid(f["cloud"], 0.4)
        osc.makeAndSaveScancontextAndKeys(ds)
        dg, kg = sc.get(40 + k)
        do, ko = osc.get(40 + k)
        assert np.array_equal(dg, do) and np.array_equal(kg.view(np.uint32), ko.view(np.uint32)), k
        rg, ro = sc.detectLoopClosureID(), osc.detectLoopClosureID()
        assert rg["loop_id"] == ro["loop_id"] and rg["nn_idx"] == ro["nn_idx"] and abs(rg["min_dist"] - ro["min_dist"]) <= 1e-12
    assert sc.size() == 44
    reg.close()
    sc.close()


def test_sc_side_stream_overlap_is_ordered(O, S, hdl64_stream):
    """ScanContext on the device's side stream (stage D overlapping stages B and C): the same records and detections as
    the oracle while the features context is re-run back to back, i.e. the events keep reader and writer apart."""
    reg = S.ScanRegistration(S.HDL64, 5.0, max_points=200000)
    od = S.LaserOdometry(max_points=200000)
    sc = S.SCManager(dist_thres=0.4, side_stream=1)
    osc = O.SCManager(dist_thres=0.4)
    rng = np.random.default_rng(5)
    for i in range(40):
        d = rng.uniform(-2, 18, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5)
        sc.saveScancontextAndKeys(d)
        osc.saveScancontextAndKeys(d)
    for k in range(6):
        reg.laserCloudHandler(hdl64_stream(k))
        sc.insert_features(reg)   # side stream, no host synchronisation
        if k % 2 == 0:            # detect on some scans only: the next run must wait for the reader by itself
            sc.detect_enqueue()
        od.step_features(reg)     # main stream work of the same scan
        if k % 2 == 0:
            rg = sc.detect_collect()
        f = O.features(hdl64_stream(k), O.HDL64, 5.0)
        ds, _ = O.voxel_grid(f["cloud"], 0.4)
        osc.makeAndSaveScancontextAndKeys(ds)
        if k % 2 == 0:
            ro = osc.detectLoopClosureID()
            assert rg["loop_id"] == ro["loop_id"] and rg["nn_idx"] == ro["nn_idx"] and abs(rg["min_dist"] - ro["min_dist"]) <= 1e-12, k
    for k in range(6):
        dg, kg = sc.get(40 + k)
        do, ko = osc.get(40 + k)
        assert np.array_equal(dg, do) and np.array_equal(kg.view(np.uint32), ko.view(np.uint32)), k
    for x in (reg, od, sc):
        x.close()


def test_pipelined_stages_match_oracle(O, S, hdl64_stream):
    """scal_set_stream_mode(1): every stage on its own stream, consecutive scans overlapping like the reference's four nodes
    (A and B of scan k+1, and its stage-C prefetch, while C of scan k still runs; map insertion behind the pose).  Software-
    pipelined exactly as bench.py does it: a ring of features contexts, stage A two scans ahead, stage B one scan ahead of the
    pose it hands over, stage-C steps queued behind each other (two uncollected).  Every pose and every loop-closure answer must
    equal the oracle chain."""
    n = 8
    S.set_stream_mode(1)
    try:
        regs = [S.ScanRegistration(S.HDL64, 5.0, max_points=200000) for _ in range(2)]
        od = S.LaserOdometry(max_points=200000)
        mp = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=2000000)
        sc = S.SCManager(dist_thres=0.4)
        osc = O.SCManager(dist_thres=0.4)
        rng = np.random.default_rng(11)
        for i in range(40):
            d = rng.uniform(-2, 18, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5)
            sc.saveScancontextAndKeys(d)
            osc.saveScancontextAndKeys(d)
        poses, loops = [], {}
        regs += [S.ScanRegistration(S.HDL64, 5.0, max_points=200000) for _ in range(4)]  # ring of six, as bench.py uses
        R = len(regs)

        def side(j):  # bench.py's side job j: loop answer of scan j, stage A of scan j+2, prefetch + ScanContext of scan j+1
            if j >= 1:
                loops[j] = sc.detect_collect()
            if j + 2 < n:
                regs[(j + 2) % R].laserCloudHandler(hdl64_stream(j + 2))
            if j + 1 < n:
                mp.prefetch_features(regs[(j + 1) % R])
                sc.insert_features(regs[(j + 1) % R])
                sc.detect_enqueue()

        regs[0].laserCloudHandler(hdl64_stream(0))
        od.enqueue_features(regs[0])
        side(-1)
        loops[0] = sc.detect_collect()
        inflight = 0
        for k in range(n):
            if k > 0:
                side(k)
            else:  # job 0 without the collect (done above, nothing was queued in between)
                regs[2 % R].laserCloudHandler(hdl64_stream(2))
                mp.prefetch_features(regs[1])
                sc.insert_features(regs[1])
                sc.detect_enqueue()
            if k + 1 < n:
                od.enqueue_features(regs[(k + 1) % R])   # stage B of scan k+1 queued before scan k's pose is collected
            qlc, tlc, qw, tw, ost = od.collect()
            mp.enqueue_features(regs[k % R], qw, tw)     # queues behind the stage-C steps still running
            inflight += 1
            if inflight > 2:
                poses.append(mp.collect()[:2])
                inflight -= 1
        while inflight:
            poses.append(mp.collect()[:2])
            inflight -= 1
        mp.finish()
        print("stage C paths (speculative, general, redone after window move, insertion redone):", mp.path_counters())
        loops = [loops[k] for k in range(n)]
        oo, om = O.Odometry(), O.Mapper(0.4, 0.8)
        for k in range(n):
            fo = O.features(hdl64_stream(k), O.HDL64, 5.0)
            co = fo["cloud"]
            a = oo.step(co[fo["sharp"]], co[fo["less_sharp"]], co[fo["flat"]], fo["less_flat"])
            qo, to, so, _ = om.step(co[fo["less_sharp"]], fo["less_flat"], co, a[2], a[3])
            assert max(np.abs(poses[k][0] - qo).max(), np.abs(poses[k][1] - to).max()) <= 1e-6, k
            ds, _ = O.voxel_grid(co, 0.4)
            osc.makeAndSaveScancontextAndKeys(ds)
            ro = osc.detectLoopClosureID()
            assert loops[k]["loop_id"] == ro["loop_id"] and loops[k]["nn_idx"] == ro["nn_idx"], k
            assert abs(loops[k]["min_dist"] - ro["min_dist"]) <= 1e-12, k
        for which in (0, 1):
            assert mp.export(which).shape == om.export(which).shape
        for x in regs + [od, mp, sc]:
            x.close()
    finally:
        S.set_stream_mode(0)


def test_sharded_stage_d_flow_equals_single_context(S, hdl64_stream):
    """The stage-D flow bench.py uses for N > 1, in one process: a builder context turns the keyframe cloud into a descriptor
    (queued, waited for by ticket order while the next one is already building), two database shards insert / answer in batches.
    Must equal ONE context doing insert_features + detectLoopClosureID on the same scans."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    G = 2
    regs = [S.ScanRegistration(S.HDL64, 5.0, max_points=200000) for _ in range(2)]
    single = S.SCManager(dist_thres=0.4)
    builder = S.SCManager(dist_thres=0.4, max_keyframes=4, side_stream=1)
    shards = [S.SCManager(dist_thres=0.4, n_shards=G, shard=s, side_stream=5) for s in range(G)]
    rng = np.random.default_rng(23)
    for i in range(40):
        d = rng.uniform(-2, 18, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5)
        single.saveScancontextAndKeys(d)
        for sh in shards:
            sh.saveScancontextAndKeys(d)
    d_q = [ctypes.c_void_p(), ctypes.c_void_p()]
    d_out = ctypes.c_void_p()
    for p in d_q:
        assert hip.hipMalloc(ctypes.byref(p), 1200 * 8) == 0
    assert hip.hipMalloc(ctypes.byref(d_out), 3 * 24) == 0
    counter, size_at_rebuild, n_global = 0, 0, 40
    refs, got = [], []

    def exchange(k):
        nonlocal counter, size_at_rebuild, n_global
        builder.wait_descriptor()
        n_global += 1
        if counter % 30 == 0:
            size_at_rebuild = n_global
        counter += 1
        cands = []
        for sh in shards:
            sh.insert_descriptors_device(d_q[k % 2], 1)
            sh.shard_query_batch_device(d_q[k % 2], [size_at_rebuild], d_out)
            sh.sync()
            buf = np.zeros(3 * 24, np.uint8)
            assert hip.hipMemcpy(buf.ctypes.data, d_out, buf.nbytes, 2) == 0
            cands += [S.SCCand.from_buffer_copy(buf[24 * j:24 * j + 24].tobytes()) for j in range(3)]
        got.append(S.merge_candidates(cands, 0.4))

    n = 6
    for k in range(n):
        reg = regs[k % 2]
        reg.laserCloudHandler(hdl64_stream(k))
        single.insert_features(reg)
        refs.append(single.detectLoopClosureID())
        builder.make_features_enqueue(reg, d_q[k % 2])   # scan k builds ...
        if k > 0:
            exchange(k - 1)                              # ... while scan k-1 is exchanged
    exchange(n - 1)
    for k in range(n):
        assert got[k]["loop_id"] == refs[k]["loop_id"] and got[k]["nn_idx"] == refs[k]["nn_idx"], k
        assert abs(got[k]["min_dist"] - refs[k]["min_dist"]) <= 1e-12, k
    for p in d_q + [d_out]:
        hip.hipFree(p)
    for x in regs + [single, builder] + shards:
        x.close()


def test_async_entry_points_reject_misuse(S, hdl64_stream):
    """The split entry points keep a bounded number of steps in flight per context (odometry 4, mapping 4, prefetches 3): collecting
    nothing, or queueing more, is an error (SCAL_E_STATE), and the context stays usable."""
    reg = S.ScanRegistration(S.HDL64, 5.0, max_points=200000)
    od = S.LaserOdometry(max_points=200000)
    mp = S.LaserMapping(0.4, 0.8, max_scan_points=200000, max_map_points=1000000)
    sc = S.SCManager()
    with pytest.raises(S.ScalError) as e:
        od.collect()
    assert e.value.code == S.E_STATE
    with pytest.raises(S.ScalError) as e:
        mp.collect()
    assert e.value.code == S.E_STATE
    with pytest.raises(S.ScalError) as e:
        sc.detect_collect()
    assert e.value.code == S.E_STATE
    with pytest.raises(S.ScalError) as e:
        sc.wait_descriptor()
    assert e.value.code == S.E_STATE
    reg.laserCloudHandler(hdl64_stream(0))
    for _ in range(4):
        od.enqueue_features(reg)
    with pytest.raises(S.ScalError) as e:
        od.enqueue_features(reg)
    assert e.value.code == S.E_STATE
    for _ in range(4):
        qlc, tlc, qw, tw, st = od.collect()
    with pytest.raises(S.ScalError) as e:
        od.collect()
    assert e.value.code == S.E_STATE
    for _ in range(4):
        mp.enqueue_features(reg, qw, tw)
    with pytest.raises(S.ScalError) as e:
        mp.enqueue_features(reg, qw, tw)   # a fifth step before any pose has been collected
    assert e.value.code == S.E_STATE
    for _ in range(4):
        q, t, ms = mp.collect()
        assert ms.insert_path == -1          # the insertion may still be behind the pose
    with pytest.raises(S.ScalError) as e:
        mp.collect()
    assert e.value.code == S.E_STATE
    mp.finish()
    assert mp.export(0).shape[0] > 0
    # four prefetches without a step in between: the fourth is refused
    for _ in range(3):
        mp.prefetch_features(reg)
    with pytest.raises(S.ScalError) as e:
        mp.prefetch_features(reg)
    assert e.value.code == S.E_STATE
    q2, t2, ms2 = mp.process_features(reg, qw, tw)   # consumes the oldest prefetch
    assert ms2.insert_path in (0, 1)
    mp.finish()
    # a features context that is run again between a prefetch and its step: the prefetched inputs belong to the previous scan,
    # the step is refused (generation check) and the next step works again
    reg.laserCloudHandler(hdl64_stream(1))
    mp.prefetch_features(reg)
    reg.laserCloudHandler(hdl64_stream(2))
    with pytest.raises(S.ScalError) as e:
        mp.enqueue_features(reg, qw, tw)
    assert e.value.code == S.E_STATE and "run again" in str(e.value)
    q3, t3, ms3 = mp.process_features(reg, qw, tw)
    assert ms3.insert_path in (0, 1)
    for x in (reg, od, mp, sc):
        x.close()


def test_odometry_ceres_adapter_mode(O, S, hdl64_stream):
    """Stage B with the solver on the host (scal_odom_adapter_*): blocks and their batched residual / Jacobian evaluation against the
    oracle's autodiff, and - with the oracle's restatement of ceres::Solve as the host solver - the same poses as the all-device step."""
    a, b = S.LaserOdometry(max_points=200000), S.LaserOdometry(max_points=200000)
    checked = 0
    for k in range(5):
        f = O.features(hdl64_stream(k), O.HDL64, 5.0)
        c = f["cloud"]
        clouds = (c[f["sharp"]], c[f["less_sharp"]], c[f["flat"]], f["less_flat"])
        qlc_a, tlc_a, qw_a, tw_a, st_a = a.step(*clouds)
        q, t, need = b.adapter_begin(*clouds)
        assert need == (k > 0)
        if need:
            for outer in range(2):   # laserOdometry.cpp:278
                nb, nr = b.associate(q, t)
                assert nb == st_a.n_edge[outer] + st_a.n_plane[outer], (k, outer, nb)
                kind, cp, pa, pb = b.blocks()
                assert (kind == 0).sum() == st_a.n_edge[outer] and (kind == 1).sum() == st_a.n_plane[outer]
                x7 = np.concatenate([q, t])
                r, J = b.eval_blocks(x7)
                for i in range(0, nb, max(1, nb // 60)):
                    row = int(3 * (kind[:i] == 0).sum() + (kind[:i] != 0).sum())
                    ro, Jo = O.factor_eval(int(kind[i]), cp[i], np.concatenate([pa[i], pb[i]]), x7)
                    n_r = 3 if kind[i] == 0 else 1
                    assert np.abs(r[row:row + n_r] - ro).max() <= 1e-9 * max(1.0, np.abs(ro).max())
                    assert np.abs(J[row:row + n_r] - Jo).max() <= 1e-9 * max(1.0, np.abs(Jo).max())
                    checked += 1
                x, _, _, _ = O.ceres_solve(kind, cp, pa, pb, x7)
                q, t = x[:4].copy(), x[4:].copy()
        qw_b, tw_b = b.adapter_finish(q, t)
        assert max(np.abs(qw_a - qw_b).max(), np.abs(tw_a - tw_b).max()) <= 1e-9, k
        assert max(np.abs(qlc_a - q).max(), np.abs(tlc_a - t).max()) <= 1e-9, k
    assert checked > 200
    a.close(), b.close()
