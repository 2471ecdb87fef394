#!/usr/bin/env python3
"""Where do the HIP ICP and the oracle part ways on a real keyframe pair?  Both are run with an iteration cap of 1, 2, 3, ... and
their transforms compared after every iterate (tests/test_voxel_sc_gpu.py::test_loop_icp_matches_oracle, case 2)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", os.path.join("sc-a-loam_amd", "python")):
    sys.path.insert(0, os.path.join(ROOT, p))
import scaloam as S, oracle_py as O
G = os.path.join(ROOT, "tests", "golden")
vg = S.VoxelGrid()
a = vg.filter(np.load(os.path.join(G, "KAIST03_000000.npy")), 0.4)
b = vg.filter(np.load(os.path.join(G, "KAIST03_000007.npy")), 0.4)
tgt = np.concatenate([a, vg.filter(np.load(os.path.join(G, "KAIST03_000020.npy")), 0.4)])
full_g = S.LoopICP(max_source=100000, max_target=400000).align(b, tgt)
full_o = O.icp_align(b, tgt)
rows = []
for it in range(1, max(full_g["iterations"], full_o["iterations"]) + 2):
    g = S.LoopICP(max_source=100000, max_target=400000, max_iterations=it).align(b, tgt)
    o = O.icp_align(b, tgt, max_iter=it)
    rows.append(dict(cap=it, d_T=float(np.abs(g["T"] - o["T"]).max()), it_g=g["iterations"], it_o=o["iterations"], st_g=g["state"], st_o=o["state"],
                     ncorr_g=g.get("n_correspondences"), fit_g=g["fitness"], fit_o=o["fitness"]))
print(json.dumps(dict(full=dict(it_g=full_g["iterations"], it_o=full_o["iterations"], st_g=full_g["state"], st_o=full_o["state"],
                                d_T=float(np.abs(full_g["T"] - full_o["T"]).max())), per_cap=rows), indent=1))
