#!/usr/bin/env python3
"""Generate tests/golden/oracle_golden.npz: outputs of the oracle (CPU restatement) on the committed real scans.

The reference ships no tests and cannot be built here, so these vectors pin the ORACLE against
accidental change; what pins the oracle against the reference is checked separately in tests/test_oracle_cpu.py
(KAIST03 ring ids / ordering, the reference's own vendored nanoflann).  Inputs are only the committed .npy scans, so
the file regenerates bit-identically on any x86-64 glibc host:  python tools/make_golden.py
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O

G = os.path.join(ROOT, "tests", "golden")
out = {}
names = ["KAIST03_000000", "KAIST03_000007", "KAIST03_000020"]
od, mp = O.Odometry(), O.Mapper(0.4, 0.8)
descs = []
for nm in names:
    a = np.load(os.path.join(G, nm + ".npy"))
    f = O.features(a[:, :3], O.OS1_64, 0.5)
    c = f["cloud"]
    out[nm + "_n_kept"] = np.int64(f["n_kept"])
    for k in ("sharp", "less_sharp", "flat"):
        out[nm + "_" + k] = f[k].astype(np.int32)
    out[nm + "_less_flat_n"] = np.int64(f["less_flat"].shape[0])
    out[nm + "_less_flat_sum"] = f["less_flat"].astype(np.float64).sum(0)
    out[nm + "_curv_sum"] = np.float64(f["curvature"].astype(np.float64).sum())
    x = od.step(c[f["sharp"]], c[f["less_sharp"]], c[f["flat"]], f["less_flat"])
    q, t, st, _ = mp.step(c[f["less_sharp"]], f["less_flat"], c, x[2], x[3])
    out[nm + "_odom_pose"] = np.concatenate([x[2], x[3]])
    out[nm + "_map_pose"] = np.concatenate([q, t])
    out[nm + "_map_blocks"] = np.array(list(st.n_edge) + list(st.n_plane), np.int64)
    ds, _ = O.voxel_grid(c, 0.4)
    out[nm + "_ds04_n"] = np.int64(ds.shape[0])
    sc = O.SCManager()
    descs.append(sc.makeScancontext(ds))
for nm in ["Seosan01_000000", "Seosan01_000011"]:
    a = np.load(os.path.join(G, nm + ".npy"))
    ds, _ = O.voxel_grid(a, 0.4)
    descs.append(O.SCManager().makeScancontext(ds))
out["sc_descs"] = np.stack(descs)
D = np.zeros((5, 5)); S = np.zeros((5, 5), np.int64)
for i in range(5):
    for j in range(5):
        D[i, j], S[i, j] = O.sc_distance(descs[i], descs[j])
out["sc_dist"] = D
out["sc_shift"] = S
np.savez_compressed(os.path.join(G, "oracle_golden.npz"), **out)
print("wrote", len(out), "arrays;", "sc_dist diag", np.diag(D))
