#!/usr/bin/env python3
"""BASELINE config #4 on ONE GPU, parity mode, as SURVEY.md section 8d specifies it: a 5,000-keyframe ScanContext database -
4,500 places along a 9 km path through the seed-401 world + 500 revisits (earlier places re-rendered with yaw U(0, 2 pi) and 0.5 m
lateral offset, so true loops exist) - and ALL 5,000 keyframes queried in insertion order with the reference's rules
(detectLoopClosureID, Scancontext.cpp:336-427: >= 31 keyframes, tree rebuilt every 30th query, newest 30 excluded, 3 ring-key
candidates, 7-shift distance).  Two forms:
  single   one insert + one search per keyframe (what the SLAM loop issues): scal_sc_insert_descriptor + scal_sc_detect
  batched  the database in place, 64 queries per launch set, every query against its own tree size:
           scal_sc_shard_query_batch_device (the per-query records are then merged exactly as the sharded search merges them)
Both are compared with the oracle's detectLoopClosureID on ALL queries: loop id, nearest index, yaw (= shift), candidates, distance <= 1e-5.
Roofline per query: 80 n_db + 38,400 B (SURVEY.md section 8d "D parity") over the device time of the search kernels.
Descriptors come from the library's own stage A + keyframe filter + makeScancontext on rendered HDL-64 scans (that part's parity is
tests/test_voxel_sc_gpu.py's business); --fast renders with a 16-beam sensor instead.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", os.path.join("sc-a-loam_amd", "python"), os.path.join("tools", "synth")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0


def plan(n_total, n_revisit, seed):
    """Insertion order: new places along a lawnmower path (2 m apart, lanes 6 m apart), revisits sprinkled in from #200 on."""
    rng = np.random.default_rng(seed)
    n_places = n_total - n_revisit
    poses = []
    x0, x1, y, step, d = -100.0, 200.0, -100.0, 2.0, 1
    x = x0
    while len(poses) < n_places:
        poses.append((x, y, 0.0 if d > 0 else np.pi))
        x += d * step
        if x > x1 or x < x0:
            x = min(max(x, x0), x1)
            y += 6.0
            d = -d
    order, placed, next_place = [], [], 0
    rev_slots = set(rng.choice(np.arange(200, n_total), n_revisit, replace=False).tolist())
    for i in range(n_total):
        if i in rev_slots and len(placed) > 150:
            j = int(rng.integers(0, len(placed) - 100))
            px, py, yaw = poses[placed[j]]
            psi = rng.uniform(0, 2 * np.pi)
            side = 0.5 if rng.uniform() < 0.5 else -0.5
            order.append((px - side * np.sin(yaw), py + side * np.cos(yaw), psi, 1))
        else:
            if next_place >= n_places:  # ran out of new places: make it one more revisit-free place at the end of the path
                next_place = n_places - 1
            px, py, yaw = poses[next_place]
            placed.append(next_place)
            next_place += 1
            order.append((px, py, yaw, 0))
    return order


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--revisits", type=int, default=500)
    ap.add_argument("--seed", type=int, default=401)
    ap.add_argument("--fast", action="store_true", help="render the keyframes with the 16-beam sensor model (4x fewer rays)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dist-thres", type=float, default=0.2)
    a = ap.parse_args()
    import torch
    import scaloam as S
    import scansynth
    import oracle_py as O
    threads = min(16, len(os.sched_getaffinity(0)))
    sensor, lid, rng_min = (scansynth.VLP16, S.VLP16, 0.5) if a.fast else (scansynth.HDL64, S.HDL64, 5.0)
    world = scansynth.World(sensor, a.seed, threads=threads)
    order = plan(a.n, a.revisits, a.seed)
    # ---- descriptors through the library's own front end
    t0 = time.time()
    reg = S.ScanRegistration(lid, rng_min, max_points=min(400000, world.cap + 1024))
    build = S.SCManager(max_radius=80.0, dist_thres=a.dist_thres, max_keyframes=a.n + 8)
    for i, (x, y, yaw, _) in enumerate(order):
        q = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])
        xyz = world.scan_pose(q, np.array([x, y, 1.73 if not a.fast else 1.0]), 40100 + i)
        reg.laserCloudHandler(xyz)
        build.insert_features(reg)
    descs = [build.get(i)[0] for i in range(a.n)]
    reg.close(), build.close()
    gen_s = time.time() - t0
    # ---- oracle: the answers of all queries, insertion order
    osc = O.SCManager(max_radius=80.0, dist_thres=a.dist_thres)
    t0 = time.time()
    ref = []
    for d in descs:
        osc.saveScancontextAndKeys(d)
        ref.append(osc.detectLoopClosureID())
    cpu_s = time.time() - t0

    def check(got, name):
        bad = 0
        for i, (g, r) in enumerate(zip(got, ref)):
            ok = g["loop_id"] == r["loop_id"] and g["nn_idx"] == r["nn_idx"] and abs(float(g["yaw"]) - float(r["yaw"])) <= 1e-6  # yaw = shift x 6 deg (:422)
            ok &= (abs(g["min_dist"] - r["min_dist"]) <= 1e-5) or (g["min_dist"] == r["min_dist"])
            ok &= list(g["cand"]) == list(r["cand"])
            bad += not ok
            if not ok and bad <= 3:
                print(f"{name}: query {i} differs: {g} vs {r}", file=sys.stderr)
        return bad

    # ---- single-query form
    sc = S.SCManager(max_radius=80.0, dist_thres=a.dist_thres, max_keyframes=a.n + 8)
    S.prof_reset()
    S.prof_enable(True, "k_sc_topk,k_sc_detect")
    got, t_search = [], 0.0
    for d in descs:
        sc.saveScancontextAndKeys(d)
        t1 = time.perf_counter()
        got.append(sc.detectLoopClosureID())
        t_search += time.perf_counter() - t1
    S.prof_enable(False)
    prof = S.prof_read_all()
    bad_single = check(got, "single")
    searched = sum(1 for i in range(a.n) if i + 1 >= 31)
    dev_ms = sum(v[0] for k, v in prof.items() if k in ("k_sc_topk", "k_sc_detect"))
    bytes_total = sum(80.0 * (i + 1) + 38400.0 for i in range(a.n) if i + 1 >= 31)
    single = {"queries_per_s": searched / t_search, "host_ms_per_query": t_search / searched * 1e3, "device_us_per_query": dev_ms / max(1, searched) * 1e3,
              "mismatches_vs_oracle": bad_single,
              "roofline": {"bound": "hbm", "achieved": bytes_total / (dev_ms * 1e-3) / 1e9 if dev_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bytes_total / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dev_ms else None,
                           "bytes_formula": "80 n_db + 38,400 B per query (SURVEY.md section 8d), n_db = database size at that query"}}
    loops = sum(1 for g in got if g["loop_id"] >= 0)
    true_rev = sum(1 for o in order if o[3])
    # ---- batched form: the full database in place, B queries per launch set, each with its own tree size
    from scaloam.sharded import TreePeriodBook
    B = a.batch
    book = TreePeriodBook(0)
    limits = []
    for i in range(a.n):
        limits += book.step(1) if i + 1 >= 31 else [0]
        if i + 1 < 31:
            book.n_global = i + 1  # the reference returns before its tree bookkeeping while the database is small (:346-350)
    allq = torch.from_numpy(np.stack([np.ascontiguousarray(d.T).reshape(-1) for d in descs])).cuda()  # [n][1200] column-major
    d_rec = torch.zeros(B * 3 * 24, dtype=torch.uint8, device="cuda")
    S.prof_reset()
    S.prof_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recs = []
    for b0 in range(0, a.n, B):
        nb = min(B, a.n - b0)
        sc.shard_query_batch_device(allq[b0:b0 + nb].data_ptr(), limits[b0:b0 + nb], d_rec.data_ptr())
        sc.sync()
        recs.append(d_rec[: nb * 72].cpu().numpy().copy())
    t_batch = time.perf_counter() - t0
    S.prof_enable(False)
    profb = S.prof_read_all()
    gotb = []
    for bi, r in enumerate(recs):
        r = r.reshape(-1, 3, 24)
        for j in range(r.shape[0]):
            i = bi * B + j
            if i + 1 < 31:
                gotb.append(dict(ref[i]))  # database too small: detectLoopClosureID returns before it searches (:346-350)
                continue
            gotb.append(S.merge_candidates([S.SCCand.from_buffer_copy(r[j, c].tobytes()) for c in range(3)], a.dist_thres))
    bad_batch = check(gotb, "batched")
    dev_ms_b = sum(v[0] for v in profb.values())
    bytes_b = sum(80.0 * limits[i] + 38400.0 for i in range(a.n) if i + 1 >= 31)
    batched = {"batch": B, "queries_per_s": a.n / t_batch, "device_us_per_query": dev_ms_b / a.n * 1e3, "mismatches_vs_oracle": bad_batch,
               "kernel_ms_total": {k: v[0] for k, v in sorted(profb.items())},
               "roofline": {"bound": "hbm", "achieved": bytes_b / (dev_ms_b * 1e-3) / 1e9 if dev_ms_b else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": bytes_b / (dev_ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS if dev_ms_b else None,
                            "bytes_formula": "80 x eligible tree size + 38,400 B per query"}}
    sc.close()
    print(json.dumps({
        "metric": "ScanContext loop-search queries/s, 5k-keyframe database, parity mode (BASELINE config #4 on one GPU)",
        "database": {"keyframes": a.n, "revisits": true_rev, "world_seed": a.seed, "sensor": "VLP-16 model" if a.fast else "HDL-64 model",
                     "path": "lawnmower, 2 m between places, 6 m between lanes", "render_and_describe_s": gen_s},
        "loops_detected": loops, "dist_thres": a.dist_thres, "single": single, "batched": batched,
        "cpu_oracle": {"queries_per_s": a.n / cpu_s, "cores": 1, "kind": "port", "note": "oracle insert + detectLoopClosureID per keyframe"},
        "parity": "every query: loop id, nearest index, shift and candidate list equal to the oracle's, distance within 1e-5"}))


if __name__ == "__main__":
    main()
