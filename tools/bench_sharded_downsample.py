#!/usr/bin/env python3
"""Offline map merge over N GPUs with one exchange: every rank merges its own contiguous range of keyframes
(scal_mapmerge_add_batch_device), then scaloam.sharded.sharded_downsample runs pubMap's VoxelGrid over the union - slab all-to-all
(RCCL under --backend nccl) + the HIP filter on each slab.  Rank 0 also filters the whole map on its own GPU and checks that the
ranks' parts, concatenated in rank order, are bit-identical.  Launch: python -m torch.distributed.run --nproc-per-node N ...
(--backend gloo rehearses it with ranks sharing one GPU)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join("sc-a-loam_amd", "python"), os.path.join("tools", "synth")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64, help="keyframes in the whole map")
    ap.add_argument("--leaf", type=float, default=0.4)
    ap.add_argument("--backend", default="nccl")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    torch.zeros(1, device="cuda")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    dist.init_process_group(a.backend, rank=rank, world_size=world)
    import scaloam as S
    from scaloam import sharded
    import scansynth
    world_gen = scansynth.World(scansynth.HDL64, 77)
    base = [np.hstack([world_gen.scan(k), np.full((world_gen.scan(k).shape[0], 1), float(k), np.float32)]) for k in range(4)]

    def pose(f):
        q, t = world_gen.pose(f * 5)
        x, y, z, w = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        return np.hstack([R, t.reshape(3, 1)]).reshape(-1)

    def merged(f0, f1):  # keyframes [f0, f1) merged on this rank's GPU -> device tensor [n, 4]
        frames = [base[f % 4] for f in range(f0, f1)]
        offs = np.concatenate([[0], np.cumsum([x.shape[0] for x in frames])]).astype(np.int64)
        d_in = torch.from_numpy(np.ascontiguousarray(np.concatenate(frames), np.float32)).cuda()
        mm = S.MapMerge(max_points=int(offs[-1]) + 1024, max_frame_points=16, device=local)
        mm.add_batch_device(d_in.data_ptr(), offs, np.array([pose(f) for f in range(f0, f1)]), 2.0)
        n = mm.size()
        return mm, n

    f0, f1 = rank * a.frames // world, (rank + 1) * a.frames // world
    mm, n = merged(f0, f1)
    pts = torch.from_numpy(mm.download()[:n]).cuda()  # the rank's part of the map as a tensor the exchange can take
    sharded.sharded_downsample(pts, a.leaf)  # warm-up (allocations, RCCL rings)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    part = sharded.sharded_downsample(pts, a.leaf)
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    sizes = [None] * world
    dist.all_gather_object(sizes, (int(part.shape[0]), int(n), float(dt.item())))
    parts = [None] * world
    dist.all_gather_object(parts, part)
    out = None
    if rank == 0:
        whole_mm, whole_n = merged(0, a.frames)
        want = whole_mm.downsample(a.leaf)
        got = np.concatenate(parts)
        same = bool(got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)))
        out = {"metric": "points/sec through the sharded VoxelGrid (exchange + local filters)", "n_gpus": world, "backend": a.backend,
               "frames": a.frames, "points_in": int(sum(s[1] for s in sizes)), "voxels_out": int(got.shape[0]), "per_rank_voxels": [s[0] for s in sizes],
               "ms": max(s[2] for s in sizes) * 1e3, "value": sum(s[1] for s in sizes) / max(s[2] for s in sizes), "unit": "points/s",
               "equals_single_gpu_filter_bitwise": same,
               "note": "host-side orchestration in torch (sort by destination, all_to_all_single) + the HIP filter; includes the centroid download"}
    dist.barrier()
    dist.destroy_process_group()
    if out:
        print(json.dumps(out))
        if not out["equals_single_gpu_filter_bitwise"]:
            sys.exit(1)


if __name__ == "__main__":
    main()
