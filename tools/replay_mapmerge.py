#!/usr/bin/env python3
"""Replay of utils/python/makeMergedMap.py on the GPU: reads a session directory as the reference saves it (Scans/*.pcd +
optimized_poses.txt), merges the keyframes into the global frame (near-range removal 2 m) with scal_mapmerge_*, writes the map as
a binary PCD.  `--check` also runs the CPU oracle and compares bit for bit."""
import argparse
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", os.path.join("sc-a-loam_amd", "python")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("data_dir", help="session directory: Scans/*.pcd and optimized_poses.txt")
    ap.add_argument("--range", type=int, nargs=2, default=None, help="keyframe index range [a, b) like scan_idx_range_to_stack")
    ap.add_argument("--near", type=float, default=2.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    import scaloam as S
    from scaloam import formats as F
    files = sorted(glob.glob(os.path.join(a.data_dir, "Scans", "*.pcd")))
    poses = F.read_poses(os.path.join(a.data_dir, "optimized_poses.txt"))
    lo, hi = a.range if a.range else (0, min(len(files), len(poses)))
    frames = [F.read_pcd(f) for f in files[lo:hi]]
    mm = S.MapMerge(max_points=sum(f.shape[0] for f in frames) + 1, max_frame_points=max(f.shape[0] for f in frames))
    for f, p in zip(frames, poses[lo:hi]):
        mm.add(f, p, a.near)
    out = mm.download()
    print(f"{len(frames)} keyframes, {sum(f.shape[0] for f in frames)} points in, {out.shape[0]} points in the map")
    if a.check:
        import oracle_py as O
        ref = O.mapmerge(frames, poses[lo:hi], a.near)
        print("bit-exact vs oracle:", bool(ref.shape == out.shape and np.array_equal(ref.view(np.uint32), out.view(np.uint32))))
    path = a.out or os.path.join(a.data_dir, f"map_{lo}_to_{hi}_with_intensity.pcd")
    F.write_pcd_binary(path, out)
    print("written", path)


if __name__ == "__main__":
    main()
