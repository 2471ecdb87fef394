#!/usr/bin/env python3
"""Convert a few of the reference's sample keyframe scans (data files, not source) into
small .npy fixtures under tests/golden/.

Source data: /root/reference/utils/sample_data/{KAIST03,Seosan01}/Scans/*.pcd
(binary PCD, fields x y z intensity, float32; parse by POINTS, the files carry trailing pad bytes;
SURVEY.md section 4).  Run in the build container only; the GPU box has no /root/reference.
"""
import os, sys
import numpy as np

REF = "/root/reference/utils/sample_data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def read_pcd(path):
    b = open(path, "rb").read()
    k = b.index(b"DATA binary\n") + len(b"DATA binary\n")
    hdr = b[:k].decode()
    n = int([l for l in hdr.splitlines() if l.startswith("POINTS")][0].split()[1])
    fields = [l for l in hdr.splitlines() if l.startswith("FIELDS")][0].split()[1:]
    assert fields == ["x", "y", "z", "intensity"], fields
    return np.frombuffer(b, dtype=np.float32, count=n * 4, offset=k).reshape(n, 4).copy()


def main():
    os.makedirs(OUT, exist_ok=True)
    picks = [("KAIST03", 0), ("KAIST03", 7), ("KAIST03", 20), ("Seosan01", 0), ("Seosan01", 11)]
    for ds, i in picks:
        a = read_pcd(f"{REF}/{ds}/Scans/{i:06d}.pcd")
        np.save(f"{OUT}/{ds}_{i:06d}.npy", a)
        print(ds, i, a.shape, a[:, :3].min(0), a[:, :3].max(0))
    # poses of the first 21 keyframes (KITTI 3x4 row-major, one line each)
    for ds in ("KAIST03", "Seosan01"):
        P = np.loadtxt(f"{REF}/{ds}/optimized_poses.txt")[:21]
        np.save(f"{OUT}/{ds}_poses21.npy", P)


if __name__ == "__main__":
    main()
