#!/usr/bin/env python3
"""Loop-closure verification ICP on one MI355X vs the CPU oracle (kd-tree): a keyframe against a submap of
2*25+1 keyframes moved by one root pose and downsampled at 0.4 m, as doICPVirtualRelative builds them.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", os.path.join("sc-a-loam_amd", "python"), os.path.join("tools", "synth")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402


def main():
    import torch
    torch.cuda.set_device(0)  # torch's HIP context first, as in bench.py
    torch.zeros(1, device="cuda")
    import scaloam as S
    import scansynth
    import oracle_py as O
    world = scansynth.World(scansynth.HDL64, 99)
    vg = S.VoxelGrid(max_points=8000000)
    frames = [np.hstack([world.scan(k), np.zeros((world.scan(k).shape[0], 1), np.float32)]) for k in range(51)]
    tgt = vg.filter(np.concatenate(frames), 0.4)   # loopFindNearKeyframesCloud(.., 25, root): one pose for all -> identity here
    src = vg.filter(frames[60 % 51], 0.4).copy()
    src[:, 0] += 0.3
    src[:, 1] -= 0.2
    icp = S.LoopICP(max_source=src.shape[0] + 16, max_target=tgt.shape[0] + 16)
    out = {}
    for name, mode in (("cell_grid", 1), ("dense_sweep", 0)):
        icp.set_search(mode)
        icp.align(src, tgt)  # warm-up
        S.prof_reset()
        S.prof_enable(True)
        t0 = time.perf_counter()
        r = icp.align(src, tgt)
        dt = time.perf_counter() - t0
        S.prof_enable(False)
        prof = S.prof_read_all()
        out[name] = {"ms_per_alignment": dt * 1e3, "iterations": r["iterations"], "T": r["T"], "fitness": r["fitness"], "converged": r["converged"],
                     "kernels": {k: {"launches": v[1], "avg_ms": v[0] / v[1]} for k, v in prof.items() if k.startswith("k_icp") and v[1]}}
    # both clouds already in HBM (as they are when the submap comes from scal_mapmerge_* + the device voxel filter)
    icp.set_search(1)
    d_src, d_tgt = torch.from_numpy(np.ascontiguousarray(src, np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(tgt, np.float32)).cuda()
    icp.align_device(d_src.data_ptr(), src.shape[0], d_tgt.data_ptr(), tgt.shape[0])
    t0 = time.perf_counter()
    rdv = icp.align_device(d_src.data_ptr(), src.shape[0], d_tgt.data_ptr(), tgt.shape[0])
    dev_ms = (time.perf_counter() - t0) * 1e3
    dev_same = bool(np.array_equal(rdv["T"], out["cell_grid"]["T"]) and rdv["fitness"] == out["cell_grid"]["fitness"])
    # the whole verification on the GPU: 51 keyframes (resident) -> one root pose, concatenated (scal_mapmerge_add_batch_device)
    # -> VoxelGrid 0.4 (scal_voxel_downsample_device) -> scal_icp_align_device; what loopFindNearKeyframesCloud +
    # doICPVirtualRelative do per loop candidate (laserPosegraphOptimization.cpp:472-548)
    cat = np.ascontiguousarray(np.concatenate(frames), np.float32)
    offs = np.concatenate([[0], np.cumsum([f.shape[0] for f in frames])]).astype(np.int64)
    d_frames = torch.from_numpy(cat).cuda()
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64), (len(frames), 1))
    mm = S.MapMerge(max_points=cat.shape[0] + 1024, max_frame_points=16)
    d_sub = torch.empty((cat.shape[0], 4), dtype=torch.float32, device="cuda")
    chain_ms, chain = [], None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mm.reset()
        mm.add_batch_device(d_frames.data_ptr(), offs, ident, -1.0)
        n_all = mm.size()
        n_sub = vg.filter_device(mm.device_points(), n_all, 0.4, d_sub.data_ptr())
        chain = icp.align_device(d_src.data_ptr(), src.shape[0], d_sub.data_ptr(), n_sub)
        chain_ms.append((time.perf_counter() - t0) * 1e3)
    chain_same = bool(n_sub == tgt.shape[0] and np.array_equal(chain["T"], out["cell_grid"]["T"]) and chain["fitness"] == out["cell_grid"]["fitness"])
    same = bool(np.array_equal(out["cell_grid"]["T"], out["dense_sweep"]["T"]) and out["cell_grid"]["fitness"] == out["dense_sweep"]["fitness"])
    t1 = time.perf_counter()
    ro = O.icp_align(src, tgt)
    cpu = time.perf_counter() - t1
    g, d = out["cell_grid"], out["dense_sweep"]
    pairs = float(src.shape[0]) * tgt.shape[0]
    print(json.dumps({"metric": "loop-closure ICP alignments/sec", "value": 1e3 / g["ms_per_alignment"], "unit": "alignments/s",
                      "n_source": int(src.shape[0]), "n_target": int(tgt.shape[0]), "iterations": g["iterations"], "converged": g["converged"],
                      "fitness": g["fitness"], "ms_per_alignment": g["ms_per_alignment"], "kernels": g["kernels"],
                      "note": "host clouds in, result out: includes the H2D copy of both clouds and the cell-grid build",
                      "dense_sweep": {"ms_per_alignment": d["ms_per_alignment"], "kernels": d["kernels"],
                                      "pair_evaluations_per_s": pairs / (d["kernels"]["k_icp_nn"]["avg_ms"] * 1e-3)},
                      "device_resident_clouds": {"ms_per_alignment": dev_ms, "alignments_per_s": 1e3 / dev_ms, "equals_host_path_bitwise": dev_same},
                      "whole_verification_on_gpu": {"ms": min(chain_ms), "keyframes": len(frames), "points_merged": int(cat.shape[0]),
                                                    "submap_points": int(n_sub), "equals_host_chain_bitwise": chain_same,
                                                    "note": "merge of the resident keyframes + VoxelGrid 0.4 + ICP, nothing leaves HBM but the result"},
                      "cell_grid_equals_dense_sweep_bitwise": same,
                      "cpu_baseline": {"value": 1.0 / cpu, "unit": "alignments/s", "cores": 1, "kind": "port", "iterations": ro["iterations"]},
                      "max_abs_T_difference_vs_oracle": float(np.abs(g["T"] - ro["T"]).max())}))


if __name__ == "__main__":
    main()
