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
    icp.align(src, tgt)  # warm-up
    S.prof_reset()
    S.prof_enable(True)
    t0 = time.perf_counter()
    r = icp.align(src, tgt)
    dt = time.perf_counter() - t0
    S.prof_enable(False)
    ms, cnt = S.prof_read_all()["k_icp_nn"]
    t1 = time.perf_counter()
    ro = O.icp_align(src, tgt)
    cpu = time.perf_counter() - t1
    pairs = float(src.shape[0]) * tgt.shape[0]
    print(json.dumps({"metric": "loop-closure ICP alignments/sec", "value": 1.0 / dt, "unit": "alignments/s", "n_source": int(src.shape[0]),
                      "n_target": int(tgt.shape[0]), "iterations": r["iterations"], "converged": r["converged"], "fitness": r["fitness"],
                      "ms_per_alignment": dt * 1e3, "k_icp_nn": {"launches": cnt, "avg_ms": ms / cnt, "pair_evaluations_per_s": pairs / (ms / cnt * 1e-3)},
                      "cpu_baseline": {"value": 1.0 / cpu, "unit": "alignments/s", "cores": 1, "kind": "port", "iterations": ro["iterations"]},
                      "max_abs_T_difference_vs_oracle": float(np.abs(r["T"] - ro["T"]).max())}))


if __name__ == "__main__":
    main()
