"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes driver for oracle/liboracle.so (the CPU restatement of the reference path) and, when present,
oracle/_ref/libref_nanoflann.so (the reference's own vendored nanoflann).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
VLP16, HDL32, HDL64, OS1_64 = 0, 1, 2, 3
SCAN_LINES = {VLP16: 16, HDL32: 32, HDL64: 64, OS1_64: 64}

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class FeatureConfig(C.Structure):
    _fields_ = [("lidar_type", C.c_int), ("n_scans", C.c_int), ("minimum_range", C.c_double), ("float_math", C.c_int),
                ("cr_libm", C.c_int), ("sort_mode", C.c_int), ("voxel_order", C.c_int), ("check_finite", C.c_int)]


class FeatureOut(C.Structure):
    _fields_ = [("cloud", _f32p), ("src_index", _i32p), ("curvature", _f32p), ("label", _i32p), ("ring_start", _i32p),
                ("ring_end", _i32p), ("sharp", _i32p), ("less_sharp", _i32p), ("flat", _i32p), ("less_flat", _f32p),
                ("n_kept", C.c_int), ("n_sharp", C.c_int), ("n_less_sharp", C.c_int), ("n_flat", C.c_int),
                ("n_less_flat", C.c_int), ("n_ties", C.c_int)]


class SCConfig(C.Structure):
    _fields_ = [("max_radius", C.c_double), ("dist_thres", C.c_double), ("float_math", C.c_int), ("cr_libm", C.c_int)]


class MapConfig(C.Structure):
    _fields_ = [("line_res", C.c_float), ("plane_res", C.c_float), ("voxel_order", C.c_int), ("knn_mode", C.c_int)]


class MapStats(C.Structure):
    _fields_ = [("n_corner_stack", C.c_int), ("n_surf_stack", C.c_int), ("n_corner_map", C.c_int), ("n_surf_map", C.c_int),
                ("n_edge", C.c_int * 2), ("n_plane", C.c_int * 2), ("lm_iters", C.c_int * 2), ("lm_success", C.c_int * 2),
                ("cost_init", C.c_double * 2), ("cost_final", C.c_double * 2), ("solved", C.c_int), ("t_ms", C.c_double * 8)]


class OdomStats(C.Structure):
    _fields_ = [("n_edge", C.c_int * 2), ("n_plane", C.c_int * 2), ("lm_iters", C.c_int * 2), ("cost_init", C.c_double * 2),
                ("cost_final", C.c_double * 2), ("t_ms", C.c_double * 4)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_features_run.argtypes = [C.POINTER(FeatureConfig), _f32p, C.c_int, C.c_int, C.POINTER(FeatureOut)]
        L.orc_voxel_grid.argtypes = [_f32p, C.c_int, C.c_float, C.c_int, _f32p, _i32p, _i32p]
        L.orc_sc_create.restype = C.c_void_p
        L.orc_sc_create.argtypes = [C.POINTER(SCConfig)]
        L.orc_sc_destroy.argtypes = [C.c_void_p]
        L.orc_sc_size.argtypes = [C.c_void_p]
        L.orc_sc_make.argtypes = [C.c_void_p, _f32p, C.c_int, _f64p]
        L.orc_sc_keys.argtypes = [_f64p, _f64p, _f64p]
        L.orc_sc_insert_cloud.argtypes = [C.c_void_p, _f32p, C.c_int]
        L.orc_sc_insert_desc.argtypes = [C.c_void_p, _f64p]
        L.orc_sc_get.argtypes = [C.c_void_p, C.c_int, _f64p, _f32p]
        L.orc_sc_distance.argtypes = [_f64p, _f64p, _f64p, _i32p]
        L.orc_sc_distance_full.argtypes = [_f64p, _f64p, _f64p]
        L.orc_sc_detect.argtypes = [C.c_void_p, _f32p, _f64p, _i32p, _i32p, _f32p]
        L.orc_map_create.restype = C.c_void_p
        L.orc_map_create.argtypes = [C.POINTER(MapConfig)]
        L.orc_map_destroy.argtypes = [C.c_void_p]
        L.orc_map_step.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f64p, _f64p, _f64p, _f64p, _f32p,
                                   C.POINTER(MapStats)]
        L.orc_map_export.argtypes = [C.c_void_p, C.c_int, _f32p, C.c_int]
        L.orc_map_export_all.argtypes = [C.c_void_p, C.c_int, _f32p, C.c_int]
        L.orc_map_get_wmap_wodom.argtypes = [C.c_void_p, _f64p, _f64p]
        L.orc_odom_create.restype = C.c_void_p
        L.orc_odom_destroy.argtypes = [C.c_void_p]
        L.orc_odom_step.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f64p, _f64p, _f64p,
                                    _f64p, C.POINTER(OdomStats)]
        L.orc_factor_eval.argtypes = [C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p]
        L.orc_eig3_sym.argtypes = [_f64p, _f64p, _f64p]
        L.orc_plane_fit_5x3.argtypes = [_f64p, _f64p, _f64p]
        L.orc_ceres_solve.argtypes = [C.c_int, _i32p, _f64p, _f64p, _f64p, _f64p, _f64p, _i32p, _i32p]
        _lib = L
    return _lib


def ref_lib():
    """The reference's vendored nanoflann compiled by oracle/Makefile (None if it was never built)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libref_nanoflann.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_ringkey_knn.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _i32p, _f32p]
        _ref = R
    return _ref


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------------------------------------- stage A
def features(xyz, lidar_type, minimum_range, float_math=0, cr_libm=1, sort_mode=1, voxel_order=1, check_finite=1):
    xyz = _f32(xyz)
    n, stride = xyz.shape
    cfg = FeatureConfig(lidar_type, SCAN_LINES[lidar_type], float(minimum_range), float_math, cr_libm, sort_mode, voxel_order, check_finite)
    ns = cfg.n_scans
    m = max(n, 1)
    bufs = dict(cloud=np.zeros((m, 4), np.float32), src_index=np.zeros(m, np.int32), curvature=np.zeros(m, np.float32),
                label=np.zeros(m, np.int32), ring_start=np.zeros(ns, np.int32), ring_end=np.zeros(ns, np.int32),
                sharp=np.zeros(m, np.int32), less_sharp=np.zeros(m, np.int32), flat=np.zeros(m, np.int32),
                less_flat=np.zeros((m, 4), np.float32))
    out = FeatureOut(_p(bufs["cloud"], _f32p), _p(bufs["src_index"], _i32p), _p(bufs["curvature"], _f32p), _p(bufs["label"], _i32p),
                     _p(bufs["ring_start"], _i32p), _p(bufs["ring_end"], _i32p), _p(bufs["sharp"], _i32p),
                     _p(bufs["less_sharp"], _i32p), _p(bufs["flat"], _i32p), _p(bufs["less_flat"], _f32p))
    rc = lib().orc_features_run(C.byref(cfg), _p(xyz, _f32p), n, stride, C.byref(out))
    k = out.n_kept
    return dict(rc=rc, n_kept=k, cloud=bufs["cloud"][:k], src_index=bufs["src_index"][:k], curvature=bufs["curvature"][:k],
                label=bufs["label"][:k], ring_start=bufs["ring_start"], ring_end=bufs["ring_end"],
                sharp=bufs["sharp"][:out.n_sharp], less_sharp=bufs["less_sharp"][:out.n_less_sharp], flat=bufs["flat"][:out.n_flat],
                less_flat=bufs["less_flat"][:out.n_less_flat], n_ties=out.n_ties)


def voxel_grid(xyzi, leaf, order_mode=1):
    xyzi = _f32(xyzi)
    n = xyzi.shape[0]
    out = np.zeros((max(n, 1), 4), np.float32)
    n_out = C.c_int(0)
    guard = C.c_int(0)
    lib().orc_voxel_grid(_p(xyzi, _f32p), n, C.c_float(leaf), order_mode, _p(out, _f32p), C.byref(n_out), C.byref(guard))
    return out[:n_out.value].copy(), guard.value


def icp_align(src, tgt, max_corr=150.0, max_iter=100, trans_eps=1e-6, fit_eps=1e-6):
    """doICPVirtualRelative's pcl::IterativeClosestPoint call (:518-531) -> dict(converged, T, fitness, iterations, state)"""
    L = lib()
    L.orc_icp_align.argtypes = [_f32p, C.c_int, _f32p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, _f64p, _f64p,
                                C.POINTER(C.c_int), C.POINTER(C.c_int)]
    s, t = _f32(src), _f32(tgt)
    T = np.zeros(16)
    fit = np.zeros(1)
    it, st = C.c_int(0), C.c_int(0)
    conv = L.orc_icp_align(_p(s, _f32p), s.shape[0], _p(t, _f32p), t.shape[0], max_corr, max_iter, trans_eps, fit_eps, _p(T, _f64p),
                           _p(fit, _f64p), C.byref(it), C.byref(st))
    return dict(converged=bool(conv), T=T.reshape(4, 4), fitness=float(fit[0]), iterations=it.value, state=st.value)


def mapmerge(frames, poses12, near_thres=2.0):
    """makeMergedMap.py: concatenation of the transformed, near-range-filtered keyframes (list of [n,4] f32, [k,12] f64)"""
    L = lib()
    L.orc_mapmerge_frame.argtypes = [_f32p, C.c_int, _f64p, C.c_double, _f32p]
    out = []
    for f, p in zip(frames, poses12):
        f = _f32(f)
        p = np.ascontiguousarray(p, np.float64)
        o = np.zeros((max(1, f.shape[0]), 4), np.float32)
        m = L.orc_mapmerge_frame(_p(f, _f32p), f.shape[0], _p(p, _f64p), float(near_thres), _p(o, _f32p))
        out.append(o[:m].copy())
    return np.concatenate(out) if out else np.zeros((0, 4), np.float32)


# ---------------------------------------------------------------------------------------------- stage D
class SCManager:
    def __init__(self, max_radius=80.0, dist_thres=0.2, float_math=0, cr_libm=1):
        cfg = SCConfig(max_radius, dist_thres, float_math, cr_libm)
        self.h = lib().orc_sc_create(C.byref(cfg))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sc_destroy(self.h)
            self.h = None

    def size(self):
        return lib().orc_sc_size(self.h)

    def makeScancontext(self, xyzi):
        xyzi = _f32(xyzi)
        desc = np.zeros(1200)
        lib().orc_sc_make(self.h, _p(xyzi, _f32p), xyzi.shape[0], _p(desc, _f64p))
        return desc.reshape(60, 20).T.copy()  # [ring, sector]

    def makeAndSaveScancontextAndKeys(self, xyzi):
        xyzi = _f32(xyzi)
        lib().orc_sc_insert_cloud(self.h, _p(xyzi, _f32p), xyzi.shape[0])

    def saveScancontextAndKeys(self, desc_ring_sector):
        d = _f64(np.asarray(desc_ring_sector).T.reshape(-1))
        lib().orc_sc_insert_desc(self.h, _p(d, _f64p))

    def get(self, idx):
        desc = np.zeros(1200)
        key = np.zeros(20, np.float32)
        lib().orc_sc_get(self.h, idx, _p(desc, _f64p), _p(key, _f32p))
        return desc.reshape(60, 20).T.copy(), key

    def detectLoopClosureID(self):
        yaw = C.c_float(0)
        md = C.c_double(0)
        nn = C.c_int(0)
        cand = np.zeros(3, np.int32)
        cd = np.zeros(3, np.float32)
        loop = lib().orc_sc_detect(self.h, C.byref(yaw), C.byref(md), C.byref(nn), _p(cand, _i32p), _p(cd, _f32p))
        return dict(loop_id=loop, yaw=yaw.value, min_dist=md.value, nn_idx=nn.value, cand=cand, cand_d=cd)


def sc_keys(desc_ring_sector):
    d = _f64(np.asarray(desc_ring_sector).T.reshape(-1))
    rk = np.zeros(20)
    sk = np.zeros(60)
    lib().orc_sc_keys(_p(d, _f64p), _p(rk, _f64p), _p(sk, _f64p))
    return rk, sk


def sc_distance(sc1, sc2):
    a = _f64(np.asarray(sc1).T.reshape(-1))
    b = _f64(np.asarray(sc2).T.reshape(-1))
    d = C.c_double(0)
    s = C.c_int(0)
    lib().orc_sc_distance(_p(a, _f64p), _p(b, _f64p), C.byref(d), C.byref(s))
    return d.value, s.value


def sc_distance_full(sc1, sc2):
    a = _f64(np.asarray(sc1).T.reshape(-1))
    b = _f64(np.asarray(sc2).T.reshape(-1))
    out = np.zeros(60)
    lib().orc_sc_distance_full(_p(a, _f64p), _p(b, _f64p), _p(out, _f64p))
    return out


def ref_ringkey_knn(keys, queries, k=3):
    R = ref_lib()
    if R is None:
        return None
    keys = _f32(keys)
    queries = _f32(queries)
    idx = np.zeros((queries.shape[0], k), np.int32)
    d = np.zeros((queries.shape[0], k), np.float32)
    R.ref_ringkey_knn(_p(keys, _f32p), keys.shape[0], keys.shape[1], _p(queries, _f32p), queries.shape[0], k, _p(idx, _i32p), _p(d, _f32p))
    return idx, d


# ---------------------------------------------------------------------------------------------- stage C
class Mapper:
    def __init__(self, line_res=0.4, plane_res=0.8, voxel_order=1, knn_mode=0):
        cfg = MapConfig(line_res, plane_res, voxel_order, knn_mode)
        self.h = lib().orc_map_create(C.byref(cfg))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_map_destroy(self.h)
            self.h = None

    def step(self, corner_last, surf_last, full_res, q_wodom, t_wodom, want_registered=False):
        c = _f32(corner_last).reshape(-1, 4)
        s = _f32(surf_last).reshape(-1, 4)
        f = _f32(full_res).reshape(-1, 4) if full_res is not None else None
        q = _f64(q_wodom)
        t = _f64(t_wodom)
        qo = np.zeros(4)
        to = np.zeros(3)
        reg = np.zeros_like(f) if (want_registered and f is not None) else None
        st = MapStats()
        lib().orc_map_step(self.h, _p(c, _f32p), c.shape[0], _p(s, _f32p), s.shape[0], _p(f, _f32p), 0 if f is None else f.shape[0],
                           _p(q, _f64p), _p(t, _f64p), _p(qo, _f64p), _p(to, _f64p), _p(reg, _f32p), C.byref(st))
        return qo, to, st, reg

    def export_all(self, which):
        """every cube of the 21x21x11 grid (what /laser_cloud_map carries, laserMapping.cpp:824-837), one feature class"""
        n = lib().orc_map_export_all(self.h, which, None, 0)
        out = np.zeros((max(n, 1), 4), np.float32)
        lib().orc_map_export_all(self.h, which, _p(out, _f32p), n)
        return out[:n]

    def export(self, which):
        n = lib().orc_map_export(self.h, which, None, 0)
        out = np.zeros((max(n, 1), 4), np.float32)
        lib().orc_map_export(self.h, which, _p(out, _f32p), n)
        return out[:n]

    def wmap_wodom(self):
        q = np.zeros(4)
        t = np.zeros(3)
        lib().orc_map_get_wmap_wodom(self.h, _p(q, _f64p), _p(t, _f64p))
        return q, t


# ---------------------------------------------------------------------------------------------- stage B
class Odometry:
    def __init__(self):
        self.h = lib().orc_odom_create()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_odom_destroy(self.h)
            self.h = None

    def step(self, sharp, less_sharp, flat, less_flat):
        a = [_f32(x).reshape(-1, 4) for x in (sharp, less_sharp, flat, less_flat)]
        qlc = np.zeros(4)
        tlc = np.zeros(3)
        qw = np.zeros(4)
        tw = np.zeros(3)
        st = OdomStats()
        lib().orc_odom_step(self.h, _p(a[0], _f32p), a[0].shape[0], _p(a[1], _f32p), a[1].shape[0], _p(a[2], _f32p), a[2].shape[0],
                            _p(a[3], _f32p), a[3].shape[0], _p(qlc, _f64p), _p(tlc, _f64p), _p(qw, _f64p), _p(tw, _f64p), C.byref(st))
        return qlc, tlc, qw, tw, st


def factor_eval(kind, cp, params6, x7):
    cp = _f64(cp)
    pr = _f64(params6)
    x = _f64(x7)
    r = np.zeros(3)
    J = np.zeros((3, 7))
    lib().orc_factor_eval(kind, _p(cp, _f64p), _p(pr, _f64p), _p(x, _f64p), _p(r, _f64p), _p(J, _f64p))
    nr = 3 if kind == 0 else 1
    return r[:nr], J[:nr]


def eig3_sym(A):
    A = _f64(A).reshape(9)
    w = np.zeros(3)
    V = np.zeros(9)
    lib().orc_eig3_sym(_p(A, _f64p), _p(w, _f64p), _p(V, _f64p))
    return w, V.reshape(3, 3)


def plane_fit(A5x3, b5):
    A = _f64(A5x3).reshape(15)
    b = _f64(b5)
    x = np.zeros(3)
    lib().orc_plane_fit_5x3(_p(A, _f64p), _p(b, _f64p), _p(x, _f64p))
    return x


def ceres_solve(kind, cp, pa, pb, x7):
    kind = np.ascontiguousarray(kind, np.int32)
    cp, pa, pb = _f64(cp), _f64(pa), _f64(pb)
    x = _f64(x7).copy()
    trace = np.zeros(6)
    nt = C.c_int(0)
    term = C.c_int(0)
    it = lib().orc_ceres_solve(kind.shape[0], _p(kind, _i32p), _p(cp, _f64p), _p(pa, _f64p), _p(pb, _f64p), _p(x, _f64p), _p(trace, _f64p),
                               C.byref(nt), C.byref(term))
    return x, it, trace[:nt.value], term.value
