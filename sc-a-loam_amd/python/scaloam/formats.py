"""On-disk formats either side of the hot path, host-side and dependency-free: what the reference's
savers write and its tools read.  No GPU code here.

  PCD binary      pcl::io::savePCDFileBinary of PointXYZI keyframes (laserPosegraphOptimization.cpp:695); read back by
                  utils/python/makeMergedMap.py:95-99.  PCL maps a file of round_up(header, 4096) + data bytes and writes the
                  data right behind the header, which leaves 4096 - len(header) zero bytes at the end: reproduced byte for byte.
  pose text       one keyframe per line, the top 3x4 of the SE(3) matrix row by row, C++ default ostream formatting = %g with
                  6 significant digits (:218-259); read by makeMergedMap.py:48-56.
  times.txt       one stamp per line at max_digits10 = 17 significant digits (:700, :862-863).
  .scd            ScanContext descriptor as text, Eigen::IOFormat(3, DontAlignCols, " ", "\\n") (saveSCD, :178-191).
  KITTI .bin      float32 x, y, z, reflectance records (kittiHelper.cpp:140-150).
"""
import os
import re
import numpy as np

_FIELDS = ("x", "y", "z", "intensity")


def read_pcd(path):
    """-> [n, 4] float32 (x, y, z, intensity); DATA ascii or binary, any field order, missing intensity -> 0."""
    with open(path, "rb") as f:
        raw = f.read()
    m = re.search(rb"DATA\s+(\w+)\s*\n", raw)
    if not m:
        raise ValueError(f"{path}: no DATA line")
    head = raw[:m.end()].decode("ascii", "replace")
    kv = {}
    for line in head.splitlines():
        if line and not line.startswith("#"):
            k, _, v = line.partition(" ")
            kv[k.upper()] = v.split()
    fields, sizes, types = kv["FIELDS"], [int(v) for v in kv["SIZE"]], kv["TYPE"]
    counts = [int(v) for v in kv.get("COUNT", ["1"] * len(fields))]
    n = int(kv["POINTS"][0]) if "POINTS" in kv else int(kv["WIDTH"][0]) * int(kv["HEIGHT"][0])
    out = np.zeros((n, 4), np.float32)
    if m.group(1) == b"ascii":
        cols = np.loadtxt(raw[m.end():].decode().splitlines(), dtype=np.float64, ndmin=2) if n else np.zeros((0, sum(counts)))
        col = 0
        for name, c in zip(fields, counts):
            if name in _FIELDS:
                out[:, _FIELDS.index(name)] = cols[:, col]
            col += c
        return out
    if m.group(1) != b"binary":
        raise ValueError(f"{path}: DATA {m.group(1).decode()} is not supported (ascii and binary are)")
    kinds = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4", ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}
    dt = np.dtype([(name, kinds[(t, s)]) if c == 1 else (name, kinds[(t, s)], (c,)) for name, s, t, c in zip(fields, sizes, types, counts)])
    if n == 0:
        return out
    rec = np.frombuffer(raw, dtype=dt, count=n, offset=m.end())
    for name in fields:
        if name in _FIELDS:
            col = np.asarray(rec[name], np.float32)
            out[:, _FIELDS.index(name)] = col if col.ndim == 1 else col[:, 0]
    return out


def pcd_binary_bytes(xyzi):
    """The bytes pcl::io::savePCDFileBinary writes for a PointXYZI cloud (header, packed x y z intensity, PCL's page tail)."""
    a = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
    n = a.shape[0]
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
            f"COUNT 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n").encode("ascii")
    page = 4096
    total = (len(head) + page - 1) // page * page + a.nbytes  # the mapping PCL creates; the data sits right behind the header
    return head + a.tobytes() + b"\0" * (total - len(head) - a.nbytes)


def write_pcd_binary(path, xyzi):
    with open(path, "wb") as f:
        f.write(pcd_binary_bytes(xyzi))


def _g6(v):
    return "%g" % v  # C++ ostream default: %g, precision 6


def format_pose_line(T):
    """T: 3x4 / 4x4 / 12 numbers -> the line laserPosegraphOptimization.cpp:228-230 / :253-255 writes (without the newline)."""
    T = np.asarray(T, np.float64).reshape(-1)[:12]
    return " ".join(_g6(v) for v in T)


def read_poses(path):
    """-> [k, 12] float64, as makeMergedMap.py:48-56 parses optimized_poses.txt / odom_poses.txt"""
    rows = [[float(v) for v in line.split()] for line in open(path) if line.strip()]
    return np.asarray(rows, np.float64).reshape(-1, 12)


def write_poses(path, poses):
    with open(path, "w") as f:
        for T in np.asarray(poses, np.float64).reshape(-1, 12):
            f.write(format_pose_line(T) + "\n")


def format_time(t):
    return "%.17g" % t  # precision(max_digits10)


def read_times(path):
    return np.asarray([float(line) for line in open(path) if line.strip()], np.float64)


def scd_text(desc):
    """saveSCD: rows = rings, %.3g coefficients separated by one blank, rows by newline, no trailing newline (Eigen)."""
    d = np.asarray(desc, np.float64)
    return "\n".join(" ".join("%.3g" % v for v in row) for row in d)


def read_scd(path_or_text):
    txt = open(path_or_text).read() if os.path.exists(str(path_or_text)) else str(path_or_text)
    return np.asarray([[float(v) for v in line.split()] for line in txt.splitlines() if line.strip()], np.float64)


def read_kitti_bin(path):
    """KITTI velodyne .bin -> [n, 4] float32 (x, y, z, reflectance), kittiHelper.cpp:140-150"""
    a = np.fromfile(path, dtype="<f4")
    return a[: a.size // 4 * 4].reshape(-1, 4).copy()


def write_kitti_bin(path, xyzi):
    np.ascontiguousarray(xyzi, "<f4").reshape(-1, 4).tofile(path)


def write_scan_stream(path, scans):
    """A recorded sequence of raw scans for the C++ replay host (sc-a-loam_amd/host/replay_main.cpp): "SCALSCN1", int32 count, then
    per scan int32 n + n x 3 float32 xyz.  No counterpart in the reference (its input is a rosbag / KITTI .bin directory)."""
    with open(path, "wb") as f:
        f.write(b"SCALSCN1")
        f.write(np.int32(len(scans)).tobytes())
        for s in scans:
            a = np.ascontiguousarray(np.asarray(s, np.float32)[:, :3])
            f.write(np.int32(a.shape[0]).tobytes())
            f.write(a.tobytes())


def read_scan_stream(path):
    with open(path, "rb") as f:
        if f.read(8) != b"SCALSCN1":
            raise ValueError(f"{path} is not a scan stream file")
        count = int(np.frombuffer(f.read(4), np.int32)[0])
        out = []
        for _ in range(count):
            n = int(np.frombuffer(f.read(4), np.int32)[0])
            out.append(np.frombuffer(f.read(12 * n), np.float32).reshape(n, 3).copy())
        return out
