#!/usr/bin/env python3
"""Timeline of bench.py --timeline FILE (HIP events attached to the dispatches, unprofiled run): kernels grouped by stage stream."""
import sys, collections
rows = []
for line in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/timeline.csv"):
    n, a, b = line.strip().split(",")
    rows.append((float(a) * 1e3, float(b) * 1e3, n))
rows.sort()
stage = {}
for n in "k_pre k_classify k_ringscan k_scatter k_curv k_ring k_compact".split(): stage[n] = "A"
for n in "k_odom_gather k_odom_assoc k_odom_handover k_odom_cellscan k_odom_cellfill".split(): stage[n] = "B"
for n in "k_map_begin k_grid_count k_grid_alloc k_grid_fill k_assoc_knn k_assoc_fit k_map_pose_done k_merge_keys k_merge_lookup k_merge_write k_transform_cloud k_map_end k_grid_clear k_insert_keys k_map_heads k_map_reduce".split(): stage[n] = "C"
for n in "k_sc_bin k_sc_finish k_sc_topk k_sc_detect k_sc_keys k_sc_store".split(): stage[n] = "D"
for n in "k_map_gather k_vox_small k_keep_error k_after_stack".split(): stage[n] = "P"
t0 = rows[0][0]
# ambiguous kernels (lm, publish, voxel/radix): assign to the stage whose previous kernel ended closest before this one starts
last_end = {}
out = collections.defaultdict(list)
for a, b, n in rows:
    st = stage.get(n)
    if st is None:
        cands = ["B", "C"] if n in ("k_lm_solve", "k_publish") else ["D", "P"]
        st = min(cands, key=lambda s: abs(a - last_end.get(s, -1e9)))
    last_end[st] = b
    out[st].append((a - t0, b - a, n))
for st in "ABCDP":
    rs = out[st]
    print("--- stage", st, "busy", round(sum(r[1] for r in rs)), "us of", round(rows[-1][1] - t0))
    prev = None
    for a, d, n in rs:
        print(f"  {a:9.1f} +{d:7.1f} gap {0 if prev is None else a - prev:7.1f}  {n}")
        prev = a + d
