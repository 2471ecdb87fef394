"""Voxel grid and ScanContext parity (HIP path through the C-ABI vs the oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("leaf", [0.2, 0.4, 0.8])
def test_voxel_real_scan(O, S, golden, leaf):
    a = golden("Seosan01_000000.npy")
    vg = S.VoxelGrid()
    g = vg.filter(a, leaf)
    o, guard = O.voxel_grid(a, leaf, order_mode=1)
    assert guard == 0
    assert g.shape == o.shape
    assert np.array_equal(_bits(g), _bits(o))
    vg.close()


def test_voxel_edge_cases(O, S):
    vg = S.VoxelGrid(max_points=100000)
    assert vg.filter(np.zeros((0, 4), np.float32), 0.4).shape == (0, 4)
    one = np.array([[1.0, 2.0, 3.0, 4.0]], np.float32)
    assert np.array_equal(vg.filter(one, 0.4), one)
    # duplicates and negative coordinates
    rng = np.random.default_rng(1)
    p = rng.uniform(-30, 30, (5000, 4)).astype(np.float32)
    p = np.concatenate([p, p[:500]])
    g = vg.filter(p, 0.4)
    o, _ = O.voxel_grid(p, 0.4, order_mode=1)
    assert np.array_equal(_bits(g), _bits(o))
    # PCL's overflow guard: leaf far too small for the extent => output is the input, unchanged
    q = rng.uniform(-500, 500, (3000, 4)).astype(np.float32)
    g = vg.filter(q, 0.001)
    o, guard = O.voxel_grid(q, 0.001, order_mode=1)
    assert guard == 1
    assert np.array_equal(_bits(g), _bits(o)) and np.array_equal(_bits(g), _bits(q))
    vg.close()


def test_voxel_idempotent_and_order(O, S, golden):
    """Size-independent properties: output sorted by voxel index, one point per voxel, re-filtering is stable."""
    a = golden("KAIST03_000007.npy")
    vg = S.VoxelGrid()
    g = vg.filter(a, 0.4)
    inv = np.float32(1.0) / np.float32(0.4)
    ijk = np.floor(g[:, :3] * inv).astype(np.int64)
    ijk -= np.floor(a[:, :3].min(0) * inv).astype(np.int64)
    dims = ijk.max(0) + 2
    idx = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    assert (np.diff(idx) > 0).all()
    g2 = vg.filter(g, 0.4)
    assert g2.shape[0] <= g.shape[0]
    vg.close()


def _sc_pair(O, S, **kw):
    return O.SCManager(**kw), S.SCManager(**kw)


def test_sc_descriptor_and_keys(O, S, golden):
    vg = S.VoxelGrid()
    for name in ["KAIST03_000000.npy", "Seosan01_000000.npy", "Seosan01_000011.npy"]:
        a = golden(name)
        ds = vg.filter(a, 0.4)  # the detector's input is the keyframe downsampled at 0.4 m (PGO :629-631)
        for fm in (0, 1):
            om, gm = _sc_pair(O, S, max_radius=80.0, float_math=fm)
            do = om.makeScancontext(ds)
            dg = gm.makeScancontext(ds)
            assert np.array_equal(do, dg)
            om.makeAndSaveScancontextAndKeys(ds)
            gm.makeAndSaveScancontextAndKeys(ds)
            d1, k1 = om.get(0)
            d2, k2 = gm.get(0)
            assert np.array_equal(d1, d2)
            assert np.array_equal(k1.view(np.uint32), k2.view(np.uint32))
            gm.close()
    vg.close()


def _random_descs(rng, n, revisit_every=7):
    """Descriptors with the statistics of real ones (occupancy ~0.5, heights -2.8..18) plus shifted revisits."""
    out = []
    for i in range(n):
        if i >= 40 and i % revisit_every == 0:
            j = int(rng.integers(0, i - 35))
            d = np.roll(out[j], int(rng.integers(0, 60)), axis=1) + rng.normal(0, 0.05, (20, 60)) * (out[j] != 0)
        else:
            d = rng.uniform(-2.0, 18.0, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5)
            d[:, rng.integers(0, 60, 3)] = 0.0  # empty sectors
        out.append(d)
    return out


def test_sc_detect_stream(O, S):
    """Insertion-order replay: the >=31 gate, the 30-query tree period and the exclusion of the newest 30 keys
    (Scancontext.cpp:346-365) must give the same candidates, distances, shift and loop id at every step."""
    rng = np.random.default_rng(42)
    descs = _random_descs(rng, 150)
    om, gm = _sc_pair(O, S, max_radius=80.0, dist_thres=0.2)
    loops = 0
    for i, d in enumerate(descs):
        om.saveScancontextAndKeys(d)
        gm.saveScancontextAndKeys(d)
        ro = om.detectLoopClosureID()
        rg = gm.detectLoopClosureID()
        assert rg["loop_id"] == ro["loop_id"], i
        if i >= 30:
            assert np.array_equal(rg["cand"], ro["cand"]), (i, rg["cand"], ro["cand"])
            assert rg["nn_idx"] == ro["nn_idx"]
            assert abs(rg["min_dist"] - ro["min_dist"]) <= 1e-12
            assert rg["yaw"] == ro["yaw"]
        loops += ro["loop_id"] >= 0
    assert loops > 5
    assert gm.size() == 150
    gm.close()


def test_sc_pair_and_matrix(O, S):
    rng = np.random.default_rng(7)
    descs = _random_descs(rng, 48)
    gm = S.SCManager()
    for d in descs:
        gm.saveScancontextAndKeys(d)
    ia = rng.integers(0, 48, 200).astype(np.int32)
    ib = rng.integers(0, 48, 200).astype(np.int32)
    dist, shift = gm.distance_pairs(ia, ib)
    for k in range(200):
        do, so = O.sc_distance(descs[ia[k]], descs[ib[k]])
        assert abs(dist[k] - do) <= 1e-12 and shift[k] == so
    D, Sh = gm.distance_matrix(0, 12, 0, 48, mode=1)
    for q in range(12):
        for j in range(48):
            full = O.sc_distance_full(descs[q], descs[j])
            full = np.where(np.isnan(full), 1e300, full)
            assert abs(D[q, j] - full.min()) <= 1e-12
            assert Sh[q, j] == int(np.argmin(full))
    D0, S0 = gm.distance_matrix(3, 9, 10, 30, mode=0)
    for q in range(3, 9):
        for j in range(10, 30):
            do, so = O.sc_distance(descs[q], descs[j])
            assert abs(D0[q - 3, j - 10] - do) <= 1e-12 and S0[q - 3, j - 10] == so
    gm.close()


def test_sc_matrix_mfma(O, S):
    """Dense 60-shift matrix on the matrix cores (mode 2, unit columns through v_mfma_f64_16x16x4_f64) against the oracle's
    distDirectSC over every shift (Scancontext.cpp:83-110) and against mode 1, which sums in the reference's order: same shift,
    distances within 1e-12 (SURVEY.md 8d asks 1e-5).  Ragged ranges (not multiples of the 64-query x 4-entry workgroup tile),
    descriptors with empty sectors, one all-zero descriptor (no effective column: the 10000000 start value survives, :133)."""
    rng = np.random.default_rng(23)
    descs = _random_descs(rng, 140)
    descs[17] = np.zeros((20, 60))
    descs[90][:, 5:50] = 0.0
    gm = S.SCManager()
    for d in descs:
        gm.saveScancontextAndKeys(d)
    D1, S1 = gm.distance_matrix(0, 140, 0, 140, mode=1)
    D2, S2 = gm.distance_matrix(0, 140, 0, 140, mode=2)
    assert np.abs(D2 - D1).max() <= 1e-12
    assert np.array_equal(S2, S1)
    assert np.all(D2[17, :] == 10000000) and np.all(D2[:, 17] == 10000000) and np.all(S2[17, :] == 0)
    for q, j in [(0, 1), (3, 77), (90, 91), (139, 0), (70, 70), (42, 139)]:
        full = O.sc_distance_full(descs[q], descs[j])
        full = np.where(np.isnan(full), 1e300, full)
        assert abs(D2[q, j] - full.min()) <= 1e-12 and S2[q, j] == int(np.argmin(full))
    # every descriptor matches itself at shift 0
    keep = np.arange(140) != 17
    assert np.abs(D2[np.arange(140), np.arange(140)][keep]).max() <= 1e-12 and np.all(S2[np.arange(140), np.arange(140)] == 0)
    # ragged sub-block with offsets
    Db, Sb = gm.distance_matrix(5, 78, 9, 138, mode=2)
    assert np.array_equal(Db, D2[5:78, 9:138]) and np.array_equal(Sb, S2[5:78, 9:138])
    Dc, Sc = gm.distance_matrix(100, 101, 0, 3, mode=2)
    assert np.array_equal(Dc, D2[100:101, 0:3]) and np.array_equal(Sc, S2[100:101, 0:3])
    # mode 3: f32 matrix cores, held to the path's 1e-5 (BASELINE.json north_star); the shift may differ only between
    # shifts whose f64 distances are closer than the f32 error
    D3, S3 = gm.distance_matrix(0, 140, 0, 140, mode=3)
    assert np.abs(D3 - D2).max() <= 1e-5
    for q, j in zip(*np.nonzero(S3 != S2)):
        full = O.sc_distance_full(descs[q], descs[j])
        assert abs(full[S3[q, j]] - full[S2[q, j]]) <= 1e-5
    assert (S3 != S2).mean() < 0.01
    assert np.all(D3[17, :] == 10000000) and np.all(S3[17, :] == 0)
    Db3, Sb3 = gm.distance_matrix(5, 78, 9, 138, mode=3)
    assert np.array_equal(Db3, D3[5:78, 9:138]) and np.array_equal(Sb3, S3[5:78, 9:138])
    gm.close()


def test_sc_batch_loop_search(O, S):
    """Exhaustive loop mining over a stored session (scal_sc_batch_loop_search): per query keyframe the k best keyframes older than
    q - exclude, against the oracle's distDirectSC over all 60 shifts (Scancontext.cpp:83-110,:113-147) on every eligible pair.
    Ragged: 700 keyframes = one full 512-query tile + a partial one; exclusion window larger than the first queries' history;
    an all-zero descriptor (distance 10000000, never a NaN winner); revisits (rolled copies) must come back first with their shift."""
    rng = np.random.default_rng(29)
    n = 700
    descs = _random_descs(rng, n)
    descs[40] = np.zeros((20, 60))
    for q, j, sh in [(300, 100, 7), (650, 5, 59), (699, 600, 1), (530, 20, 33)]:
        descs[q] = np.roll(descs[j], sh, axis=1)
    gm = S.SCManager()
    for d in descs:
        gm.saveScancontextAndKeys(d)
    K, EXCL = 4, 30
    idx, dist, shift = gm.batch_loop_search(0, n, EXCL, K, 2)
    assert idx.shape == (n, K)
    assert np.all(idx[:EXCL + 1] == -1) and np.all(dist[:EXCL + 1] == 10000000)
    assert np.all(idx[EXCL + 2, 2:] == -1) and np.all(idx[EXCL + 2, :2] >= 0)  # two eligible entries only
    for q, j, sh in [(300, 100, 7), (650, 5, 59), (699, 600, 1), (530, 20, 33)]:
        assert idx[q, 0] == j and dist[q, 0] <= 1e-12 and shift[q, 0] == sh
    for q in [31, 33, 100, 300, 511, 512, 513, 650, 699]:
        lim = q - EXCL
        best = np.empty(lim)
        arg = np.empty(lim, int)
        for j in range(lim):
            full = O.sc_distance_full(descs[q], descs[j])
            full = np.where(np.isnan(full), 1e300, full)
            arg[j] = int(np.argmin(full))
            best[j] = full[arg[j]] if full[arg[j]] < 1e300 else 10000000.0
        order = np.lexsort((np.arange(lim), best))[:K]
        m = len(order)
        assert np.all(np.abs(dist[q, :m] - best[order]) <= 1e-12), q
        # the picks are the oracle's unless two candidates are closer than the matrix-core rounding
        for r in range(m):
            assert idx[q, r] == order[r] or abs(best[idx[q, r]] - best[order[r]]) <= 1e-12, (q, r)
            assert shift[q, r] == arg[idx[q, r]]
        assert np.all(idx[q, m:] == -1)
    # a sub-range equals the same rows of the full answer; the matrix mode only changes the rounding
    i2, d2, s2 = gm.batch_loop_search(500, 640, EXCL, K, 2)
    assert np.array_equal(i2, idx[500:640]) and np.array_equal(d2, dist[500:640]) and np.array_equal(s2, shift[500:640])
    i1, d1, s1 = gm.batch_loop_search(0, n, EXCL, K, 1)
    assert np.abs(d1 - dist).max() <= 1e-12 and (i1 != idx).mean() < 0.01
    D, Sh = gm.distance_matrix(0, n, 0, n, mode=2)
    for q in range(n):
        lim = max(0, q - EXCL)
        order = np.lexsort((np.arange(lim), D[q, :lim]))[:K]
        assert np.array_equal(idx[q, :len(order)], order) and np.array_equal(dist[q, :len(order)], D[q, order])
    with pytest.raises(S.ScalError):
        gm.batch_loop_search(0, n + 1, EXCL, K, 2)
    with pytest.raises(S.ScalError):
        gm.batch_loop_search(0, n, EXCL, 17, 2)
    assert gm.batch_loop_search(5, 5, EXCL, K, 2)[0].shape == (0, K)
    gm.close()


def test_sc_matrix_mfma_real_scans(O, S, golden):
    """Modes 1-3 of the dense matrix on descriptors of the reference's real sample scans (two sessions, keyframes downsampled at
    0.4 m as the detector sees them, PGO :629-631), each also inserted rolled by a few sectors (a revisit with another heading):
    every pair against the oracle's distDirectSC over the 60 shifts."""
    vg = S.VoxelGrid()
    gm = S.SCManager()
    descs = []
    for name in ["KAIST03_000000.npy", "KAIST03_000007.npy", "KAIST03_000020.npy", "Seosan01_000000.npy", "Seosan01_000011.npy"]:
        d = gm.makeScancontext(vg.filter(golden(name), 0.4))
        descs += [d, np.roll(d, 7 + len(descs), axis=1)]
    for d in descs:
        gm.saveScancontextAndKeys(d)
    n = len(descs)
    want_d, want_s = np.zeros((n, n)), np.zeros((n, n), np.int32)
    for q in range(n):
        for j in range(n):
            full = O.sc_distance_full(descs[q], descs[j])
            full = np.where(np.isnan(full), 1e300, full)
            want_d[q, j], want_s[q, j] = full.min(), int(np.argmin(full))
    for mode, tol in ((1, 1e-12), (2, 1e-12), (3, 1e-5)):
        D, Sh = gm.distance_matrix(0, n, 0, n, mode=mode)
        assert np.abs(D - want_d).max() <= tol, mode
        assert np.array_equal(Sh, want_s), mode
    # the rolled copies are found at their shift with zero distance
    for k in range(0, n, 2):
        assert want_d[k + 1, k] <= 1e-12 and want_s[k, k + 1] == (60 - (7 + k)) % 60
    gm.close()
    vg.close()


def test_sc_sharded_equals_single(O, S):
    """Keyframe i lives on shard i % G; per-shard top-3 + merge must equal the single-context answer."""
    rng = np.random.default_rng(11)
    descs = _random_descs(rng, 100)
    single = S.SCManager(dist_thres=0.3)
    G = 4
    shards = [S.SCManager(dist_thres=0.3, n_shards=G, shard=s) for s in range(G)]
    counter = 0
    size_at_rebuild = 0
    for i, d in enumerate(descs):
        single.saveScancontextAndKeys(d)
        for sh in shards:
            sh.saveScancontextAndKeys(d)
        ref = single.detectLoopClosureID()
        if i + 1 < 31:
            continue
        if counter % 30 == 0:
            size_at_rebuild = i + 1
        counter += 1
        cands = []
        for sh in shards:
            cands += list(sh.shard_query(d, size_at_rebuild))
        got = S.merge_candidates(cands, 0.3)
        assert got["loop_id"] == ref["loop_id"], i
        assert got["nn_idx"] == ref["nn_idx"] and abs(got["min_dist"] - ref["min_dist"]) <= 1e-12
        assert np.array_equal(got["cand"], ref["cand"])
    for sh in shards:
        sh.close()
    single.close()


def test_sc_sharded_batch_equals_single(O, S):
    """The batched multi-GPU form (bench.py --gpus N): per step, N descriptors arrive in global order, every shard inserts the
    ones it owns in ONE launch and answers all N queries in three launches, each query against its own tree size.  Query q of a
    step must get the single-database answer the reference would give right after inserting descriptor q."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    G = 4
    rng = np.random.default_rng(13)
    steps = 14
    descs = _random_descs(rng, G * steps)
    single = S.SCManager(dist_thres=0.3)
    shards = [S.SCManager(dist_thres=0.3, n_shards=G, shard=s) for s in range(G)]
    d_q, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_q), G * 1200 * 8) == 0 and hip.hipMalloc(ctypes.byref(d_out), G * 3 * 24) == 0
    counter, size_at_rebuild, n_global = 0, 0, 0
    for st in range(steps):
        batch = descs[st * G:(st + 1) * G]
        # reference semantics: insert, detect, insert, detect ... in global order
        refs, limits = [], []
        for d in batch:
            single.saveScancontextAndKeys(d)
            n_global += 1
            refs.append(single.detectLoopClosureID())
            if n_global >= 31:
                if counter % 30 == 0:
                    size_at_rebuild = n_global
                counter += 1
            limits.append(size_at_rebuild)
        host = np.ascontiguousarray(np.stack([d.T.reshape(-1) for d in batch]), np.float64)  # column-major 20x60 each
        assert hip.hipMemcpy(d_q, host.ctypes.data, host.nbytes, 1) == 0
        recs = []
        for sh in shards:
            sh.insert_descriptors_device(d_q, G)
            sh.shard_query_batch_device(d_q, limits, d_out)
            sh.sync()
            buf = np.zeros(G * 3 * 24, np.uint8)
            assert hip.hipMemcpy(buf.ctypes.data, d_out, buf.nbytes, 2) == 0
            recs.append(buf.reshape(G, 3, 24))
        for q in range(G):
            if n_global - (G - 1 - q) < 31:
                continue
            cands = [S.SCCand.from_buffer_copy(recs[s][q, j].tobytes()) for s in range(G) for j in range(3)]
            got = S.merge_candidates(cands, 0.3)
            # descriptors q+1.. of this step are already in the shards but newer than every tree: they cannot be candidates
            assert got["loop_id"] == refs[q]["loop_id"] and got["nn_idx"] == refs[q]["nn_idx"], (st, q)
            assert abs(got["min_dist"] - refs[q]["min_dist"]) <= 1e-12, (st, q)
    hip.hipFree(d_q), hip.hipFree(d_out)
    for sh in shards:
        sh.close()
    single.close()


def test_mapmerge_matches_oracle(O, S, golden):
    """Offline dense map merge: frame-by-frame adds and the batched device path against the oracle, bit for bit
    (same f64 operation order, compiled without FMA contraction), including an empty frame and a frame that is dropped whole."""
    import ctypes
    names = ["KAIST03_000000.npy", "KAIST03_000007.npy", "KAIST03_000020.npy"]
    frames = [golden(n) for n in names] + [np.zeros((0, 4), np.float32), np.full((300, 4), 0.5, np.float32)]
    poses = np.concatenate([golden("KAIST03_poses21.npy")[[0, 7, 20]], golden("KAIST03_poses21.npy")[[1, 2]]])
    want = O.mapmerge(frames, poses, 2.0)
    mm = S.MapMerge(max_points=1000000, max_frame_points=200000)
    for f, p in zip(frames, poses):
        mm.add(f, p, 2.0)
    got = mm.download()
    assert got.shape == want.shape and np.array_equal(_bits(got), _bits(want))
    # batched, device-resident input
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    cat = np.ascontiguousarray(np.concatenate(frames), np.float32)
    offs = np.concatenate([[0], np.cumsum([f.shape[0] for f in frames])]).astype(int)
    d = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d), cat.nbytes) == 0 and hip.hipMemcpy(d, cat.ctypes.data, cat.nbytes, 1) == 0
    mm.reset()
    mm.add_batch_device(d, offs, poses, 2.0)
    got2 = mm.download()
    assert np.array_equal(_bits(got2), _bits(want))
    # pubMap's VoxelGrid over the merged map = the oracle's voxel grid over the oracle's merge
    ds = mm.downsample(0.4)
    dso, guard = O.voxel_grid(want, 0.4, order_mode=1)
    assert guard == 0 and ds.shape == dso.shape and np.array_equal(_bits(ds), _bits(dso))
    assert mm.size() == want.shape[0]  # the merged map is kept
    # appending continues behind what is there
    mm.add(frames[0], poses[0], 2.0)
    assert mm.size() == want.shape[0] + int((np.sqrt((frames[0][:, :3].astype(np.float64) ** 2).sum(1)) > 2.0).sum())
    hip.hipFree(d)
    small = S.MapMerge(max_points=1000, max_frame_points=200000)
    small.add(frames[0], poses[0], 2.0)
    with pytest.raises(S.ScalError) as e:
        small.size()
    assert e.value.code == S.E_CAPACITY
    small.close()
    mm.close()


def test_voxel_device_in_out_equals_host(S, golden):
    """scal_voxel_downsample_device (records already in HBM, centroids left in HBM) = scal_voxel_downsample, and the chain
    device voxel -> scal_icp_align_device = the host chain: loopFindNearKeyframesCloud's VoxelGrid + doICPVirtualRelative
    (laserPosegraphOptimization.cpp:491-492, :518-535) without the clouds leaving the GPU."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    vg = S.VoxelGrid()
    clouds = {}
    for name in ("KAIST03_000000.npy", "KAIST03_000007.npy"):
        a = np.ascontiguousarray(golden(name), np.float32)
        want = vg.filter(a, 0.4)
        d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(d_in), a.nbytes) == 0 and hip.hipMalloc(ctypes.byref(d_out), a.nbytes) == 0
        assert hip.hipMemcpy(d_in, a.ctypes.data, a.nbytes, 1) == 0
        m = vg.filter_device(d_in, a.shape[0], 0.4, d_out)
        got = np.zeros((m, 4), np.float32)
        assert hip.hipMemcpy(got.ctypes.data, d_out, got.nbytes, 2) == 0
        assert m == want.shape[0] and np.array_equal(_bits(got), _bits(want))
        clouds[name] = (want, d_out, m, d_in)
    (src_h, d_src, ns, _), (tgt_h, d_tgt, nt, _) = clouds["KAIST03_000007.npy"], clouds["KAIST03_000000.npy"]
    icp = S.LoopICP(max_source=100000, max_target=400000)
    rh, rd = icp.align(src_h, tgt_h), icp.align_device(d_src, ns, d_tgt, nt)
    assert np.array_equal(rh["T"], rd["T"]) and rh["fitness"] == rd["fitness"] and rh["iterations"] == rd["iterations"]
    for _, d_out, _, d_in in clouds.values():
        hip.hipFree(d_out), hip.hipFree(d_in)
    icp.close()
    vg.close()


def test_new_entry_points_reject_bad_arguments(S):
    """Error behaviour of the round's added entry points: status codes, nothing thrown across the ABI, contexts stay usable."""
    gm = S.SCManager()
    rng = np.random.default_rng(1)
    for d in _random_descs(rng, 6):
        gm.saveScancontextAndKeys(d)
    for args in ((0, 6, 0, 6, 4), (0, 6, 0, 6, -1), (0, 7, 0, 6, 2), (3, 2, 0, 6, 2), (0, 6, -1, 6, 2)):
        with pytest.raises(S.ScalError) as e:
            gm.distance_matrix(*args[:4], mode=args[4])
        assert e.value.code == S.E_ARG
    D, _ = gm.distance_matrix(0, 6, 0, 6, mode=2)  # still works
    assert D.shape == (6, 6) and np.abs(np.diag(D)).max() <= 1e-12
    sharded = S.SCManager(n_shards=2, shard=0)
    with pytest.raises(S.ScalError) as e:
        sharded.distance_matrix(0, 0, 0, 0, mode=2)
    assert e.value.code == S.E_ARG
    icp = S.LoopICP(max_source=1000, max_target=1000)
    with pytest.raises(S.ScalError) as e:
        icp.set_search(2)
    assert e.value.code == S.E_ARG
    with pytest.raises(S.ScalError) as e:
        icp.align(np.zeros((1001, 4), np.float32), np.zeros((10, 4), np.float32))
    assert e.value.code == S.E_TOO_MANY
    vg = S.VoxelGrid(max_points=1000)
    with pytest.raises(S.ScalError) as e:
        vg.filter_device(1, 1001, 0.4, 1)  # over capacity: refused before any pointer is touched
    assert e.value.code == S.E_TOO_MANY
    with pytest.raises(S.ScalError) as e:
        vg.filter_device(None, 10, 0.4, None)
    assert e.value.code == S.E_ARG
    for x in (gm, sharded, icp, vg):
        x.close()


def test_loop_icp_matches_oracle(O, S, golden):
    """Loop-closure verification ICP: the HIP path against the oracle's restatement of pcl::IterativeClosestPoint
    on real keyframes downsampled at 0.4 m as doICPVirtualRelative prepares them.  Nearest neighbours are exact on both sides and
    the increment is stored as the f32 Matrix4f PCL keeps, which absorbs the different order of the f64 correspondence sums: on the
    real keyframe pair every one of the 99 iterates equals the oracle's bit for bit (tools/diag_icp_iterates.py), so the bar here
    is 1e-9, not the 2e-3 an earlier version of this test allowed."""
    vg = S.VoxelGrid()
    icp = S.LoopICP(max_source=100000, max_target=400000)
    a = vg.filter(golden("KAIST03_000000.npy"), 0.4)
    # (1) a known rigid motion is recovered
    th = 0.04
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    t = np.array([0.4, -0.25, 0.05])
    src = a.copy()
    src[:, :3] = (a[:, :3].astype(np.float64) @ R.T + t).astype(np.float32)
    rg, ro = icp.align(src, a), O.icp_align(src, a)
    Ti = np.eye(4)
    Ti[:3, :3], Ti[:3, 3] = R.T, -R.T @ t
    assert rg["converged"] and ro["converged"] and rg["state"] == ro["state"]
    assert np.abs(rg["T"] - Ti).max() <= 1e-5 and np.abs(rg["T"] - ro["T"]).max() <= 1e-9
    assert rg["fitness"] <= 1e-9 and rg["iterations"] == ro["iterations"]
    # (2) two different keyframes of the sample session (what the verification really sees: partial overlap)
    b = vg.filter(golden("KAIST03_000007.npy"), 0.4)
    tgt = np.concatenate([a, vg.filter(golden("KAIST03_000020.npy"), 0.4)])
    rg, ro = icp.align(b, tgt), O.icp_align(b, tgt)
    assert rg["converged"] == ro["converged"] and rg["iterations"] == ro["iterations"] and rg["state"] == ro["state"]
    assert np.abs(rg["T"] - ro["T"]).max() <= 1e-9, (rg["T"], ro["T"])
    assert abs(rg["fitness"] - ro["fitness"]) <= 1e-9 * max(1.0, ro["fitness"]), (rg["fitness"], ro["fitness"])
    # (3) iteration cap, too few points, empty clouds
    capped = S.LoopICP(max_source=100000, max_target=400000, max_iterations=2)
    rc, rco = capped.align(b, tgt), O.icp_align(b, tgt, max_iter=2)
    assert rc["iterations"] == 2 and rc["state"] == 1 and rc["converged"] and rco["state"] == 1
    assert np.abs(rc["T"] - rco["T"]).max() <= 1e-9
    r2 = icp.align(b[:2], tgt)
    assert not r2["converged"] and r2["state"] == 5
    r0 = icp.align(np.zeros((0, 4), np.float32), tgt)
    assert not r0["converged"]
    for x in (vg, icp, capped):
        x.close()


def test_loop_icp_cell_grid_equals_dense_sweep(S, golden):
    """The cell grid (default) and the dense sweep find the same nearest neighbours - same squared distances, lowest index on
    ties - so every iterate, the final transform and the fitness are bit-identical.  Cases: partial overlap; a 30 m initial offset
    (most queries unresolved by the grid in the first iterations: the dense-sweep list path); a target with a far outlier, a
    non-finite point and duplicated points (equal distances); queries far outside the grid."""
    vg = S.VoxelGrid()
    a = vg.filter(golden("KAIST03_000000.npy"), 0.4)
    b = vg.filter(golden("KAIST03_000007.npy"), 0.4)
    tgt = np.concatenate([a, vg.filter(golden("KAIST03_000020.npy"), 0.4)])
    assert tgt.shape[0] >= 8192  # large enough for the grid to be built
    far = b.copy()
    far[:, 0] += 30.0
    odd = np.concatenate([tgt, tgt[:500], np.array([[5000.0, -3000.0, 40.0, 0.0], [np.nan, 1.0, 2.0, 0.0]], np.float32)]).astype(np.float32)
    out = b.copy()
    out[:50, :3] += np.array([800.0, 650.0, -30.0], np.float32)
    grid = S.LoopICP(max_source=100000, max_target=400000, max_iterations=30)
    dense = S.LoopICP(max_source=100000, max_target=400000, max_iterations=30)
    dense.set_search(0)
    for src, t in ((b, tgt), (far, tgt), (b, odd), (out, tgt)):
        rg, rd = grid.align(src, t), dense.align(src, t)
        assert rg["iterations"] == rd["iterations"] and rg["state"] == rd["state"] and rg["n_correspondences"] == rd["n_correspondences"]
        assert np.array_equal(rg["T"], rd["T"]) and rg["fitness"] == rd["fitness"]
    # clouds already in HBM: same result as from host memory
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    bs, ts = np.ascontiguousarray(b, np.float32), np.ascontiguousarray(tgt, np.float32)
    d_s, d_t = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_s), bs.nbytes) == 0 and hip.hipMemcpy(d_s, bs.ctypes.data, bs.nbytes, 1) == 0
    assert hip.hipMalloc(ctypes.byref(d_t), ts.nbytes) == 0 and hip.hipMemcpy(d_t, ts.ctypes.data, ts.nbytes, 1) == 0
    rh, rdv = grid.align(b, tgt), grid.align_device(d_s, bs.shape[0], d_t, ts.shape[0])
    assert np.array_equal(rh["T"], rdv["T"]) and rh["fitness"] == rdv["fitness"] and rh["iterations"] == rdv["iterations"]
    hip.hipFree(d_s), hip.hipFree(d_t)
    for x in (vg, grid, dense):
        x.close()


def test_sc_5k_database_8_shards(O, S):
    """BASELINE config #4 shape: a 5000-keyframe database sharded 8 ways (keyframe i on shard i % 8), batched insert + query per
    step exactly as bench.py --gpus 8 issues them, against ONE context holding the whole database AND against the oracle's
    detectLoopClosureID fed the same 5000 descriptors (same tree period, same exclusion of the newest 30).  Every 25th step is
    compared: loop id, best candidate and its distance."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    G, steps = 8, 625
    rng = np.random.default_rng(17)
    base = rng.uniform(-2.0, 18.0, (64, 20, 60)) * (rng.uniform(size=(64, 20, 60)) < 0.5)
    single = S.SCManager(dist_thres=0.3, max_keyframes=5100)
    osc = O.SCManager(dist_thres=0.3)
    orefs_all = []
    shards = [S.SCManager(dist_thres=0.3, max_keyframes=700, n_shards=G, shard=s) for s in range(G)]
    d_q, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_q), G * 1200 * 8) == 0 and hip.hipMalloc(ctypes.byref(d_out), G * 3 * 24) == 0
    counter, size_at_rebuild, n_global, hits, checked = 0, 0, 0, 0, 0
    for st in range(steps):
        batch = []
        for q in range(G):
            i = st * G + q
            if i >= 100 and i % 9 == 0:   # a revisit: an old place seen again, rotated
                d = np.roll(batch_hist[int(rng.integers(0, len(batch_hist) - 60))], int(rng.integers(0, 60)), axis=1)
                d = d + rng.normal(0, 0.03, d.shape) * (d != 0)
            else:
                d = base[i % 64] * (rng.uniform(size=(20, 60)) < 0.97) + rng.normal(0, 0.4, (20, 60)) * (base[i % 64] != 0)
                d = np.roll(d, int(rng.integers(0, 60)), axis=1) * rng.uniform(0.6, 1.4)
            batch.append(d)
        if st == 0:
            batch_hist = []
        batch_hist.extend(batch)
        refs, orefs, limits = [], [], []
        check = st % 25 == 24
        for d in batch:
            single.saveScancontextAndKeys(d)
            osc.saveScancontextAndKeys(d)
            n_global += 1
            if check or n_global >= 31:  # (also between the checked steps: keeps the tree period in step with the reference sequence)
                refs.append(single.detectLoopClosureID())
                orefs.append(osc.detectLoopClosureID())
            else:
                refs.append(None)
                orefs.append(None)
            if n_global >= 31:
                if counter % 30 == 0:
                    size_at_rebuild = n_global
                counter += 1
            limits.append(size_at_rebuild)
        host = np.ascontiguousarray(np.stack([d.T.reshape(-1) for d in batch]), np.float64)
        assert hip.hipMemcpy(d_q, host.ctypes.data, host.nbytes, 1) == 0
        recs = []
        for sh in shards:
            sh.insert_descriptors_device(d_q, G)
            if check:
                sh.shard_query_batch_device(d_q, limits, d_out)
                sh.sync()
                buf = np.zeros(G * 3 * 24, np.uint8)
                assert hip.hipMemcpy(buf.ctypes.data, d_out, buf.nbytes, 2) == 0
                recs.append(buf.reshape(G, 3, 24))
        if check:
            for q in range(G):
                cands = [S.SCCand.from_buffer_copy(recs[s][q, j].tobytes()) for s in range(G) for j in range(3)]
                got = S.merge_candidates(cands, 0.3)
                assert got["loop_id"] == refs[q]["loop_id"] and got["nn_idx"] == refs[q]["nn_idx"], (st, q)
                assert abs(got["min_dist"] - refs[q]["min_dist"]) <= 1e-12, (st, q)
                assert got["loop_id"] == orefs[q]["loop_id"] and got["nn_idx"] == orefs[q]["nn_idx"], (st, q, "oracle")
                assert abs(got["min_dist"] - orefs[q]["min_dist"]) <= 1e-12, (st, q, "oracle")
                hits += got["loop_id"] >= 0
                checked += 1
    assert single.size() == 5000 and sum(sh.size() for sh in shards) == 5000 * G  # every shard counts every global insert
    assert checked == 200 and hits > 5
    hip.hipFree(d_q), hip.hipFree(d_out)
    for sh in shards:
        sh.close()
    single.close()


def test_voxel_large_cloud(O, S):
    """A cloud large enough (1.5 M points) for the radix sort's hierarchical histogram scan: still the PCL order and the ordered
    f32 centroids, bit for bit."""
    rng = np.random.default_rng(21)
    n = 1500000
    p = np.empty((n, 4), np.float32)
    p[:, :3] = rng.uniform(-150, 150, (n, 3)).astype(np.float32)
    p[:, 2] *= 0.1
    p[:, 3] = rng.uniform(0, 64, n).astype(np.float32)
    p[: n // 4] = p[n // 2: n // 2 + n // 4] + np.float32(0.01)  # dense voxels too
    vg = S.VoxelGrid(max_points=1600000)
    g = vg.filter(p, 0.5)
    o, guard = O.voxel_grid(p, 0.5, order_mode=1)
    assert guard == 0 and g.shape == o.shape
    assert np.array_equal(_bits(g), _bits(o))
    vg.close()


def test_device_input_behind_a_producer_stream(S, golden):
    """`_device` entry points read the caller's memory on the context's own stream (include/scaloam_hip.h "ordering against the
    caller's own streams").  The sharded map filter orders that stream behind the producer of its input on the device
    (scal_voxel_stream + torch ExternalStream): here the input is written by a copy that sits behind a ~20 ms busy-wait on a torch
    side stream - as an RCCL all-to-all would still be in flight when torch returns - and the result must equal the filter of the
    finished data."""
    torch = pytest.importorskip("torch")
    from scaloam.sharded import gpu_voxel_filter
    src = torch.from_numpy(np.ascontiguousarray(np.concatenate([golden("Seosan01_000000.npy"), golden("Seosan01_000011.npy")]), np.float32)).cuda()
    run = gpu_voxel_filter(0)
    want = run(src, 0.4)                      # producer long finished
    side = torch.cuda.Stream()
    recv = torch.zeros_like(src)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        torch.cuda._sleep(40_000_000)         # the "collective" is still running when the host moves on
        recv.copy_(src, non_blocking=True)
        got = run(recv, 0.4)                  # must wait for the copy on the device, not read the zeros
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
