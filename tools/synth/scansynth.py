"""Seeded synthetic LiDAR scans (tools/synth/libscansynth.so): input data for tests and bench.py."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
VLP16, HDL32, HDL64, OS1_64 = 0, 1, 2, 3
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


class SynthConfig(C.Structure):
    _fields_ = [("sensor", C.c_int), ("seed", C.c_uint64), ("n_boxes", C.c_int), ("n_cyl", C.c_int),
                ("region", C.c_double * 4), ("noise_sigma", C.c_double), ("threads", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libscansynth.so")
        if not os.path.exists(path):
            raise RuntimeError("tools/synth/libscansynth.so missing: run `make -C tools/synth` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.syn_world_create.restype = C.c_void_p
        L.syn_world_create.argtypes = [C.POINTER(SynthConfig)]
        L.syn_world_destroy.argtypes = [C.c_void_p]
        L.syn_world_pose.argtypes = [C.c_void_p, C.c_int, _f64p, _f64p]
        L.syn_world_max_points.argtypes = [C.c_void_p]
        L.syn_world_scan.argtypes = [C.c_void_p, C.c_int, _f32p]
        L.syn_world_scan_pose.argtypes = [C.c_void_p, _f64p, _f64p, C.c_uint64, _f32p]
        _lib = L
    return _lib


class World:
    def __init__(self, sensor, seed, n_boxes=200, n_cyl=400, region=(-120.0, 220.0, -120.0, 260.0), noise_sigma=0.02, threads=0):
        cfg = SynthConfig(sensor, seed, n_boxes, n_cyl, (C.c_double * 4)(*region), noise_sigma, threads)
        self.h = lib().syn_world_create(C.byref(cfg))
        self.sensor = sensor
        self.cap = lib().syn_world_max_points(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().syn_world_destroy(self.h)
            self.h = None

    def pose(self, k):
        q = np.zeros(4)
        t = np.zeros(3)
        lib().syn_world_pose(self.h, k, q.ctypes.data_as(_f64p), t.ctypes.data_as(_f64p))
        return q, t

    def scan(self, k):
        buf = np.zeros((self.cap, 3), np.float32)
        n = lib().syn_world_scan(self.h, k, buf.ctypes.data_as(_f32p))
        return buf[:n].copy()

    def scan_pose(self, q, t, noise_seed):
        buf = np.zeros((self.cap, 3), np.float32)
        q = np.ascontiguousarray(q, np.float64)
        t = np.ascontiguousarray(t, np.float64)
        n = lib().syn_world_scan_pose(self.h, q.ctypes.data_as(_f64p), t.ctypes.data_as(_f64p), noise_seed, buf.ctypes.data_as(_f32p))
        return buf[:n].copy()
