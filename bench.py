#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config #2: scans/sec through feature extraction + scan-to-map ICP +
ScanContext loop search on KITTI-like HDL-64 scans (~120k points), one MI355X per process.

A "step" is one scan through the whole hot path on one GPU, inputs already resident in HBM:
  stage A  feature extraction                 (scanRegistration.cpp:134-421)
  stage B  scan-to-scan odometry              (laserOdometry.cpp:267-568)   - provides the prior for stage C
  stage C  scan-to-map, 2 outer x <=4 LM its  (laserMapping.cpp:310-802, :845-849)
  stage D  ScanContext insert + loop search   (Scancontext.cpp:151-260, :336-427) over a pre-filled keyframe database
Default schedule: scal_pipeline (include/scaloam_hip.h) - the four stages on their own streams with consecutive scans overlapping,
the way the reference's four ROS nodes (scanRegistration, laserOdometry, laserMapping, laserPosegraphOptimization) work on different
scans at the same time.  The schedule lives INSIDE the library (five C++ host threads); this file only pushes scans and pops poses,
exactly what the C++ host sc-a-loam_amd/host/replay_main.cpp does.  Every scan still goes through A -> B -> C and A -> D with the
reference's data dependencies, and the K timed steps end only when the last scan's map insertion is done.  --no-overlap runs one
scan at a time through the per-stage calls on one stream; both schedules give bit-identical poses (`final_map_pose`).
N > 1 (one process per GPU, torch.distributed/RCCL): stages A-C do not shard (pose k+1 depends on pose k and on the map), so every
rank replays its own seeded sequence ("replicas only", weak scaling); the ScanContext keyframe database IS sharded (keyframe i on
rank i % N): descriptors and per-shard top-3 records are exchanged with two RCCL all-gathers per --sc-exchange-every scans
(SURVEY.md section 8e).

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the launching streams (dominant kernel over the
timed region, per-stage lines from extra untimed steps); `cpu_baseline` is the oracle (CPU restatement of the reference path) timed
on this host's cores on a bounded sample of the same scans (rank 0, N=1 only); `as_integrated` is the C++ host replaying the first
scans through the synchronous host-array entry points, one thread per stage, as INTEGRATION.md has a maintainer wire them.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import tempfile
import time

# four busy HIP streams next to torch's; the runtime multiplexes streams onto 4 hardware queues by default, which would serialise
# stages that share a queue.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in ("oracle", os.path.join("sc-a-loam_amd", "python"), os.path.join("tools", "synth")):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
REPLAY = os.path.join(ROOT, "sc-a-loam_amd", "bin", "replay_main")
METRIC = "scans/sec (feat-extract + scan-to-map ICP + SC loop search), KITTI HDL-64"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, help="2: KITTI-like HDL-64 (BASELINE.json's metric); 3: MulRan-like OS1-64 stream harness")
    ap.add_argument("--replay", default="lockstep", help="config 3: lockstep (every scan through every stage, as fast as possible) or realtime")
    ap.add_argument("--rate", type=float, default=10.0, help="config 3, realtime: scans per second of the replay (10 Hz sensor)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3, help="the timed region of K steps is run at least this many times on consecutive scans; value = median")
    ap.add_argument("--min-timed-s", type=float, default=0.3, help="repetitions are added until the timed regions together last about this long "
                    "(a 20-step region is ~6 ms: three of those are three noisy samples)")
    ap.add_argument("--h2d", type=int, default=1, help="N=1: one more repetition with every scan uploaded from host memory inside the timed region (reported beside value)")
    ap.add_argument("--sc-db", type=int, default=5000, help="keyframes pre-filled into the ScanContext database")
    ap.add_argument("--sc-revisits", type=int, default=-1, help="of those, keyframes that are earlier visits of the places the timed scans see, re-rendered "
                    "with a random heading and 0.5 m lateral offset (true loops exist); -1 = one per three scans of the sequence")
    ap.add_argument("--cpu-sample", type=int, default=60, help="scans timed through the CPU oracle (0 = skip)")
    ap.add_argument("--cpp-sample", type=int, default=60, help="scans the C++ host (replay_main) replays as integrated / through scal_pipeline (0 = skip)")
    ap.add_argument("--seed", type=int, default=205)
    ap.add_argument("--ring", type=int, default=6, help="features contexts used in turn by the pipeline")
    ap.add_argument("--depth", type=int, default=2, help="stage-C steps queued on the device and not collected")
    ap.add_argument("--ahead", type=int, default=4, help="scans pushed beyond the one whose pose is awaited")
    ap.add_argument("--no-overlap", action="store_true", help="one scan at a time on one stream through the per-stage calls (no pipeline)")
    ap.add_argument("--sc-exchange-every", type=int, default=1, help="N > 1: scans whose descriptors / candidate records travel in one pair of all-gathers "
                    "(the reference searches at 1 Hz, laserPosegraphOptimization.cpp:732-741; answers are identical for every value)")
    ap.add_argument("--prof-every", type=int, default=8,
                    help="0 = no kernel timing at all; otherwise the two LM solve kernels (the roofline line) carry start/stop timestamps on every "
                         "N-th step of the timed region, and every instrumented kernel on --prof-steps extra steps behind it (outside the timing)")
    ap.add_argument("--prof-steps", type=int, default=24, help="extra, untimed steps with per-kernel timestamps on every launch (kernel_ms_per_step)")
    ap.add_argument("--timeline-kernels", default="", help="with --timeline: only these kernels (comma separated), over the whole timed region")
    ap.add_argument("--timeline", default="", help="development aid: write (kernel, start ms, stop ms) of every dispatch of six timed steps to this CSV")
    ap.add_argument("--seqs", type=int, default=4, help="N=1: after the single-sequence measurement, this many independent sequences are run through ONE "
                    "pipeline whose kernels share launches (scal_pipeline_create_multi); reported as `batched`, never as `value` (0/1 = skip)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real runs; gloo only to rehearse N>1 on fewer GPUs")
    return ap.parse_args()


def synth_descs(rng, n):
    """ScanContext-like filler descriptors (occupancy ~0.5, heights -2..18 m, a few empty sectors)."""
    d = rng.uniform(-2.0, 18.0, (n, 60, 20)) * (rng.uniform(size=(n, 60, 20)) < 0.5)
    for i in range(n):
        d[i, rng.integers(0, 60, 3), :] = 0.0
    return d  # [n][sector][ring] == column-major 20x60


def revisit_descriptors(S, world_gen, ks, cap, device, seed):
    """Descriptors of EARLIER VISITS of places the sequence passes (SURVEY.md section 8d #4 "revisits = earlier places re-rendered
    with yaw U(0, 2 pi) and 0.5 m lateral offset"): pose of scan k turned by a random heading and moved 0.5 m sideways, rendered by
    the same sensor model with its own noise, through stage A and makeScancontext on the GPU.  Returned as [ring, sector] arrays."""
    if not ks:
        return []
    rng = np.random.default_rng(seed)
    reg = S.ScanRegistration(S.HDL64, 5.0, max_points=cap, device=device)
    sc = S.SCManager(max_radius=80.0, dist_thres=0.4, max_keyframes=len(ks) + 8, device=device)
    out = []
    for i, k in enumerate(ks):
        q, t = world_gen.pose(k)
        psi = rng.uniform(0.0, 2.0 * np.pi)
        qz = np.array([0.0, 0.0, np.sin(psi / 2), np.cos(psi / 2)])
        # q' = q * Rz(psi), storage (x, y, z, w)
        a, b = q, qz
        q2 = np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2],
                       a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])
        yaw = 2.0 * np.arctan2(q[2], q[3])
        side = 0.5 if (i & 1) else -0.5
        t2 = t + side * np.array([-np.sin(yaw), np.cos(yaw), 0.0])
        xyz = world_gen.scan_pose(q2, t2, 700001 + 13 * k)
        reg.laserCloudHandler(xyz)
        sc.insert_features(reg)
        out.append(sc.get(i)[0])
    reg.close(), sc.close()
    return out


def build_database(S, world_gen, a, n_scans, cap, device):
    """The pre-filled keyframe database: revisits of the sequence's own places first (oldest keyframes: inside every query's tree and
    outside its 30 newest), filler descriptors behind them."""
    n_rev = a.sc_revisits if a.sc_revisits >= 0 else n_scans // 3
    n_rev = max(0, min(n_rev, a.sc_db))
    ks = [int(round(j * (n_scans - 1) / max(1, n_rev - 1))) for j in range(n_rev)] if n_rev else []
    rev = revisit_descriptors(S, world_gen, ks, cap, device, 401)
    fill = synth_descs(np.random.default_rng(4242), a.sc_db - len(rev))
    return rev + [d.T for d in fill], len(rev)


def cpp_host(mode, stream_file, n_warm, sc_db, resident=0, stream_mode=0, timeout=600):
    """The C++ host (sc-a-loam_amd/host/replay_main.cpp: includes include/scaloam_hip.h, links libscaloam_hip.so) as a child process."""
    if not os.path.exists(REPLAY):
        return {"error": "sc-a-loam_amd/bin/replay_main is not built"}
    try:
        r = subprocess.run([REPLAY, "--scans", stream_file, "--mode", mode, "--warmup", str(n_warm), "--sc-db", str(sc_db), "--resident", str(resident), "--stream-mode", str(stream_mode)],
                           capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            return {"error": f"replay_main rc {r.returncode}: {r.stderr[-300:]}"}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, subprocess.TimeoutExpired) as e:
        return {"error": str(e)}


def run_stream(a):
    """BASELINE config #3: MulRan-like OS1-64 stream, full odom + mapping (+ ScanContext on keyframes), launch values of
    launch/aloam_mulran.launch.  lockstep: every scan through every stage, one scan at a time (the parity schedule).  realtime:
    scans arrive at --rate Hz; stages A and B take every scan, stage C takes the NEWEST finished one and drops what queued up
    behind it (the reference's rule to stay real-time, laserMapping.cpp:300-304); latency = arrival -> mapping pose on the host."""
    import torch
    import scaloam as S
    import scansynth
    from scaloam.pgo import KeyframeGate
    K, W = a.steps, a.warmup
    world_gen = scansynth.World(scansynth.OS1_64, 301, threads=min(16, len(os.sched_getaffinity(0))))
    scans = [world_gen.scan(k) for k in range(W + K)]
    cap = max(s.shape[0] for s in scans) + 1024
    d_scans = [torch.from_numpy(s).cuda() for s in scans]
    reg = S.ScanRegistration(S.OS1_64, 0.5, max_points=min(400000, cap))
    od = S.LaserOdometry(max_points=cap)
    mp = S.LaserMapping(0.4, 0.8, max_scan_points=cap, max_map_points=4000000)
    sc = S.SCManager(max_radius=80.0, dist_thres=0.2, max_keyframes=a.sc_db + W + K + 64)
    rng = np.random.default_rng(4242)
    for d in synth_descs(rng, a.sc_db):
        sc.saveScancontextAndKeys(d.T)
    gate = KeyframeGate(1.0, 10.0)
    out = dict(keyframes=0, loops=0, dropped=0, lat=[])
    lm = dict(map=0.0, map_launches=0, odom=0.0, odom_launches=0, map_blocks=0, map_evals=0)

    def one(k, t_arrival=None):
        reg.run_device(d_scans[k].data_ptr(), scans[k].shape[0], 3)
        qlc, tlc, qw, tw, ost = od.step_features(reg)
        qm, tm, mst = mp.process_features(reg, qw, tw)
        lm_account(lm, "odom", ost), lm_account(lm, "map", mst)
        if t_arrival is not None:
            out["lat"].append(time.perf_counter() - t_arrival)
        if gate(qm, tm):
            out["keyframes"] += 1
            sc.insert_features(reg)
            out["loops"] += sc.detectLoopClosureID()["loop_id"] >= 0
        return qm, tm

    for k in range(W):
        one(k)
    torch.cuda.synchronize()
    for key in lm:
        lm[key] = 0
    S.prof_reset()
    S.prof_enable(a.prof_every > 0, "k_lm_solve_map,k_lm_solve_odom")
    t0 = time.perf_counter()
    if a.replay == "lockstep":
        for k in range(W, W + K):
            pose = one(k)
    else:
        period = 1.0 / a.rate
        k = W
        while k < W + K:
            now = time.perf_counter() - t0
            due = int(now / period)  # scans that have arrived so far: W .. W + due
            newest = min(W + K - 1, W + due)
            if newest < k:
                time.sleep(max(0.0, (k - W) * period - now))
                continue
            # stages A and B see every scan (laserOdometry has no drop rule); stage C only the newest (:300-304)
            for j in range(k, newest):
                reg.run_device(d_scans[j].data_ptr(), scans[j].shape[0], 3)
                od.step_features(reg)
                out["dropped"] += 1
            pose = one(newest, t0 + (newest - W) * period)
            k = newest + 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    S.prof_enable(False)
    prof = S.prof_read_all()
    lat = np.array(out["lat"]) * 1e3 if out["lat"] else None
    roofline = roofline_of(prof, dict(lm=lm), False)
    cpu = None
    if a.cpu_sample > 0:
        cpu = cpu_baseline(scans[: min(len(scans), a.cpu_sample)], a.sc_db, sensor="OS1_64", min_range=0.5, sc_thres=0.2, gate=(1.0, 10.0))
    print(json.dumps({
        "metric": "scans/sec, MulRan-like OS1-64 stream: full odom + mapping, ScanContext on keyframes (BASELINE config #3)",
        "value": K / dt, "unit": "scans/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 points / f64 pose algebra", "data": "synthetic",
        "config": {"workload": "OS1-64 (64 beams x 1024 columns, seed 301, minimum_range 0.5, line/plane 0.4/0.8, keyframe gap 1 m / 10 deg, "
                               "sc_dist_thres 0.2), replay " + a.replay + (f" at {a.rate:g} Hz" if a.replay != "lockstep" else ""),
                   "points_per_scan_in": int(np.mean([s.shape[0] for s in scans])), "sc_db_keyframes": a.sc_db},
        "roofline": roofline, "cpu_baseline": cpu,
        "keyframes": out["keyframes"], "loops_detected": int(out["loops"]), "scans_dropped_by_mapping": out["dropped"],
        "latency_ms": None if lat is None else {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
        "final_map_pose": {"q": pose[0].tolist(), "t": pose[1].tolist()}}))


def lm_account(lm, which, st_):
    """algorithmic bytes of the LM solves (72 B per edge block, 56 B per plane block, read once per evaluation;
    evaluations of a solve = 1 + its iterations), stage C and stage B separately"""
    for o in range(2):
        ne, npl, it = st_.n_edge[o], st_.n_plane[o], st_.lm_iters[o]
        if ne + npl == 0:
            continue
        lm[which] += (72.0 * ne + 56.0 * npl) * (1 + it)
        lm[which + "_E"] = lm.get(which + "_E", 0.0) + 72.0 * ne + 56.0 * npl
        lm[which + "_launches"] += 1
        if which == "map":
            lm["map_blocks"] += ne + npl
            lm["map_evals"] += 1 + it


def run_batched(a, S, n_seqs, single_rate, device, threads):
    """SURVEY.md section 8d "batched / streamed numbers (several independent scans or sequences in flight)": n_seqs independent
    sequences - different seeded worlds, own map / poses / ScanContext database each - step together through ONE pipeline whose
    kernels share launches (blockIdx.z = sequence, csrc/batch.hpp).  Same K-step timed region, same per-step work as the headline
    run for every sequence; scans/s counts the scans of all sequences.  Poses are bit-identical to solo runs
    (tests/test_pipeline_gpu.py::test_multi_sequence_batched_launches_equal_solo_runs)."""
    import torch
    import scansynth
    K, W = a.steps, a.warmup
    R = max(1, min(a.reps, 3), min(int(math.ceil(0.1 / (K * 0.1e-3 * n_seqs))), max(1, 400 // K)))
    PS = 12 if a.prof_every > 0 else 0  # extra, untimed steps with every launch timed (kernel table of the batched launches)
    total = W + K * R + PS
    gens = [scansynth.World(scansynth.HDL64, a.seed + 7919 * (q + 1), threads=threads) for q in range(n_seqs)]
    scans = [[g.scan(k) for k in range(total)] for g in gens]
    npts = [[s.shape[0] for s in sq] for sq in scans]
    d_scans = [[torch.from_numpy(s).cuda(device) for s in sq] for sq in scans]
    cap = min(400000, max(max(n) for n in npts) + 1024)
    P = S.Pipeline(S.HDL64, 5.0, max_points=cap, line_res=0.4, plane_res=0.8, max_map_points=4000000, sc_mode=S.SC_EVERY_SCAN, sc_max_radius=80.0,
                   sc_dist_thres=0.4, sc_max_keyframes=a.sc_db + total + 64, device=device, ring=a.ring, depth=a.depth, n_seqs=n_seqs)
    fill = synth_descs(np.random.default_rng(4242), a.sc_db)
    for q in range(n_seqs):
        for d in fill:
            P.scs[q].saveScancontextAndKeys(d.T)
    lm = dict(map=0.0, map_launches=0, odom=0.0, odom_launches=0, map_blocks=0, map_evals=0)
    poses = {}

    def region(k0, n, prof):
        in_flight = 0
        for k in range(k0, k0 + n):
            if prof:
                S.prof_enable((k - W) % max(1, a.prof_every) == 0, "k_lm_solve_map,k_lm_solve_odom")
            P.push_device_multi([d_scans[q][k].data_ptr() for q in range(n_seqs)], [npts[q][k] for q in range(n_seqs)])
            in_flight += 1
            while in_flight > a.ahead:
                take(P.pop_multi())
                in_flight -= 1
        P.drain()
        while in_flight:
            take(P.pop_multi())
            in_flight -= 1

    def take(results):
        for q, r in enumerate(results):
            lm_account(lm, "map", r["map"])
            lm_account(lm, "odom", r["odom"])
            poses[q] = r["t"].tolist()

    region(0, W, False)
    for key in list(lm):
        lm[key] = 0
    S.prof_reset()
    torch.cuda.synchronize()
    rep_dt = []
    for rep in range(R):
        t0 = time.perf_counter()
        region(W + rep * K, K, a.prof_every > 0)
        torch.cuda.synchronize()
        rep_dt.append(time.perf_counter() - t0)
    S.prof_enable(False)
    prof = S.prof_read_all()
    table = None
    if PS:
        S.prof_reset()
        S.prof_enable(True)
        region(W + R * K, PS, False)
        torch.cuda.synchronize()
        S.prof_enable(False)
        table = {k: v[0] / PS for k, v in sorted(S.prof_read_all().items())}
    P.close()
    dt = float(np.median(rep_dt))
    rate = n_seqs * K / dt
    out = {"seqs": n_seqs, "scans_per_s": rate, "ms_per_step_all_seqs": dt / K * 1e3, "ms_per_scan": dt / K / n_seqs * 1e3,
           "speedup_vs_single_sequence": rate / single_rate if single_rate else None, "repetitions": R,
           "rep_ms_per_step": [x / K * 1e3 for x in rep_dt], "final_map_t": poses, "kernel_ms_per_step_all_seqs": table,
           "note": "independent sequences through ONE pipeline, their kernels sharing launches (gridDim.z = sequences); not the metric's value"}
    if "k_lm_solve_map" in prof and prof["k_lm_solve_map"][1] and lm["map_launches"]:
        ms, cnt = prof["k_lm_solve_map"]
        per_launch = lm["map"] / lm["map_launches"] * n_seqs  # a batched launch carries one solve of every sequence
        gbs = per_launch / (ms / cnt * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_lm_solve (stage C), batched launch", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": gbs / HBM_PEAK_GBS, "avg_launch_us": ms / cnt * 1e3, "timed_launches": cnt, "algorithmic_bytes_per_launch": per_launch}
    return out


def main():
    a = parse()
    if a.config == 3:
        return run_stream(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if a.backend != "nccl":
        local = local % torch.cuda.device_count()  # rehearsal: several ranks may share a card
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    def all_gather(out_t, in_t):
        """RCCL all-gather on the device tensors; the gloo rehearsal path stages through the host."""
        if a.backend == "nccl":
            dist.all_gather_into_tensor(out_t, in_t)
        else:
            parts = [torch.zeros_like(in_t, device="cpu") for _ in range(world)]
            dist.all_gather(parts, in_t.cpu())
            out_t.copy_(torch.stack(parts).reshape(out_t.shape))
    import scaloam as S
    import scansynth
    from scaloam.sharded import TreePeriodBook

    K, W = a.steps, a.warmup
    # enough repetitions that the timed regions together last ~min_timed_s (estimated at 0.3 ms per step), at most 1600 timed steps
    R = max(1, a.reps, min(int(math.ceil(a.min_timed_s / (K * 0.3e-3))), max(1, 1600 // K)))
    do_h2d = bool(a.h2d) and world == 1
    P_STEPS = a.prof_steps if (a.prof_every > 0 and not a.timeline) else 0
    total = W + K * (R + (1 if do_h2d else 0)) + P_STEPS
    # ---- synthetic HDL-64 sequence for this rank (weak scaling: one independent sequence per GPU).  The trajectory is a 100 m
    # circle at 1 m per scan: after 628 scans the vehicle drives through places it has mapped before.
    threads = max(1, min(16, len(os.sched_getaffinity(0))) // max(1, world))
    world_gen = scansynth.World(scansynth.HDL64, a.seed + 1000 * rank, threads=threads)
    t0 = time.time()
    scans = [world_gen.scan(k) for k in range(total)]
    gen_s = time.time() - t0
    npts = [s.shape[0] for s in scans]
    d_scans = [torch.from_numpy(s).cuda(local) for s in scans]  # inputs resident in HBM before the timed region
    cap = min(400000, max(npts) + 1024)
    t0 = time.time()
    descs, n_rev = build_database(S, world_gen, a, total, cap, local)
    db_s = time.time() - t0

    pipelined = not a.no_overlap
    Q = max(1, min(a.sc_exchange_every, 64 // max(1, world)))
    ring = max(a.ring, 8) if world > 1 else a.ring
    stats = dict(loops=0, blocks=0, stack_pts=0, solved=0, map_pts=0, win_pts=0, edge=0, plane=0, evals=0, odom_blocks=0, odom_evals=0)
    lm_bytes = dict(map=0.0, map_launches=0, odom=0.0, odom_launches=0, map_blocks=0, map_evals=0)
    last_pose = {}
    sc_max = a.sc_db // world + total * world + 64

    P = reg = od = mp = None
    if pipelined:
        desc_ring = torch.zeros(ring, 1200, dtype=torch.float64, device="cuda") if world > 1 else None
        P = S.Pipeline(S.HDL64, 5.0, max_points=cap, line_res=0.4, plane_res=0.8, max_map_points=4000000,
                       sc_mode=S.SC_EVERY_SCAN if world == 1 else S.SC_DESCRIPTOR, sc_max_radius=80.0, sc_dist_thres=0.4,
                       sc_max_keyframes=sc_max if world == 1 else 8, device=local, ring=ring, depth=a.depth,
                       d_desc_ring=desc_ring.data_ptr() if world > 1 else None)
        sc = P.sc if world == 1 else S.SCManager(max_radius=80.0, dist_thres=0.4, max_keyframes=sc_max, device=local, n_shards=world, shard=rank, side_stream=5)
    else:
        S.set_stream_mode(0)
        reg = S.ScanRegistration(S.HDL64, 5.0, max_points=cap, device=local)
        od = S.LaserOdometry(max_points=cap, device=local)
        mp = S.LaserMapping(0.4, 0.8, max_scan_points=cap, max_map_points=4000000, device=local)
        sc = S.SCManager(max_radius=80.0, dist_thres=0.4, max_keyframes=sc_max, device=local, n_shards=world, shard=rank)
    for d in descs:
        sc.saveScancontextAndKeys(d)  # every shard sees every insert and keeps the ones it owns

    # ---- N > 1: the exchange of the sharded ScanContext search (two all-gathers per Q scans), on a host thread of its own so that
    # neither the collectives nor their waits hold up pushing and popping
    book = TreePeriodBook(a.sc_db)
    if world > 1:
        import queue
        import threading
        q_buf = [torch.zeros(Q, 1200, dtype=torch.float64, device="cuda") for _ in range(2)]
        all_q = torch.zeros(world, Q, 1200, dtype=torch.float64, device="cuda")
        d_rec = torch.zeros(Q * world * 3 * 24, dtype=torch.uint8, device="cuda")
        all_rec = torch.zeros(world, Q * world * 3 * 24, dtype=torch.uint8, device="cuda")
        sc_ext = torch.cuda.ExternalStream(sc.stream_ptr()) if a.backend == "nccl" else None
        xq, loop_out, slot_free = queue.Queue(), queue.Queue(), [threading.Event(), threading.Event()]
        for e in slot_free:
            e.set()

        def exchange(slot, nq):
            """nq scans of every rank at once.  Global order of the descriptors: scan-major, rank-minor - the order in which a
            one-exchange-per-scan run would have inserted and queried them (TreePeriodBook), so the answers are the same."""
            if nq < Q:
                q_buf[slot][nq:].zero_()
            all_gather(all_q, q_buf[slot])
            perm = all_q.permute(1, 0, 2)[:nq].contiguous().view(nq * world, 1200)
            if sc_ext is not None:
                sc_ext.wait_stream(torch.cuda.current_stream())
            else:
                torch.cuda.current_stream().synchronize()
            sc.insert_descriptors_device(perm.data_ptr(), nq * world)
            sc.shard_query_batch_device(perm.data_ptr(), book.batch(nq, world), d_rec.data_ptr())
            if sc_ext is not None:
                torch.cuda.current_stream().wait_stream(sc_ext)
            else:
                sc.sync()
            all_gather(all_rec, d_rec)
            rec = all_rec.cpu().numpy().reshape(world, Q * world, 3, 24)  # [shard][query][candidate]: the one host wait
            slot_free[slot].set()
            for j in range(nq):
                cands = [S.SCCand.from_buffer_copy(rec[s, j * world + rank, c].tobytes()) for s in range(world) for c in range(3)]
                loop_out.put(S.merge_candidates(cands, 0.4))

        def xchg_worker():
            torch.cuda.set_device(local)
            while True:
                job = xq.get()
                if job is None:
                    return
                try:
                    exchange(*job)
                except Exception as e:  # surfaced in the main thread
                    loop_out.put(e)

        xthread = threading.Thread(target=xchg_worker, daemon=True)
        xthread.start()
        xstate = dict(slot=0, n=0, pending=0)

        def feed_exchange(res, flush=False):
            if res is not None:
                s_ = xstate["slot"]
                if xstate["n"] == 0:
                    slot_free[s_].wait()
                    slot_free[s_].clear()
                q_buf[s_][xstate["n"]].copy_(desc_ring[res["seq"] % ring])  # scan k's descriptor, valid since it was popped
                xstate["n"] += 1
            if xstate["n"] == Q or (flush and xstate["n"] > 0):
                xq.put((xstate["slot"], xstate["n"]))
                xstate["pending"] += xstate["n"]
                xstate["slot"] ^= 1
                xstate["n"] = 0

        def take_loops(block):
            while xstate["pending"] and (block or not loop_out.empty()):
                r = loop_out.get()
                if isinstance(r, Exception):
                    raise r
                stats["loops"] += r["loop_id"] >= 0
                xstate["pending"] -= 1

    def account_step(mst, ost, loop):
        lm_account(lm_bytes, "map", mst)
        lm_account(lm_bytes, "odom", ost)
        if loop is not None:
            stats["loops"] += loop["loop_id"] >= 0
        stats["blocks"] += mst.n_edge[0] + mst.n_plane[0] + mst.n_edge[1] + mst.n_plane[1]
        stats["edge"] += mst.n_edge[0] + mst.n_edge[1]
        stats["plane"] += mst.n_plane[0] + mst.n_plane[1]
        stats["evals"] += (1 + mst.lm_iters[0]) * (mst.n_edge[0] + mst.n_plane[0] > 0) + (1 + mst.lm_iters[1]) * (mst.n_edge[1] + mst.n_plane[1] > 0)
        stats["odom_blocks"] += ost.n_edge[0] + ost.n_plane[0] + ost.n_edge[1] + ost.n_plane[1]
        stats["odom_evals"] += 2 + ost.lm_iters[0] + ost.lm_iters[1]
        stats["stack_pts"] += mst.n_corner_stack + mst.n_surf_stack
        stats["solved"] += mst.solved
        stats["map_pts"] += mst.n_map_corner_total + mst.n_map_surf_total
        stats["win_pts"] += mst.n_corner_map + mst.n_surf_map

    def on_result(res):
        last_pose["q"], last_pose["t"] = res["q"].tolist(), res["t"].tolist()
        account_step(res["map"], res["odom"], res["loop"])
        if world > 1:
            feed_exchange(res)
            take_loops(False)

    prof_plan = dict(fn=None)

    def run_region(k0, n, h2d=False):
        """n consecutive scans through the hot path; returns when the last scan's pose, loop answer AND map insertion are done"""
        if pipelined:
            in_flight = 0
            for k in range(k0, k0 + n):
                if prof_plan["fn"]:
                    prof_plan["fn"](k)
                if h2d:
                    P.push(scans[k])
                else:
                    P.push_device(d_scans[k].data_ptr(), npts[k], 3)
                in_flight += 1
                while in_flight > a.ahead:
                    on_result(P.pop())
                    in_flight -= 1
            P.drain()
            while in_flight:
                on_result(P.pop())
                in_flight -= 1
            if world > 1:
                feed_exchange(None, flush=True)
                take_loops(True)
            return
        for k in range(k0, k0 + n):  # serial: A -> B -> C -> D, each stage finished before the next starts
            if prof_plan["fn"]:
                prof_plan["fn"](k)
            if h2d:
                reg.enqueue_host(scans[k])
            else:
                reg.run_device(d_scans[k].data_ptr(), npts[k], 3)
            qlc, tlc, qw, tw, ost = od.step_features(reg)
            qm, tm, mst = mp.process_features(reg, qw, tw)
            last_pose["q"], last_pose["t"] = qm.tolist(), tm.tolist()
            if world == 1:
                sc.insert_features(reg)
                r = sc.detectLoopClosureID()
            else:
                dq = q_buf[0][0]
                sc.make_features(reg, dq.data_ptr())
                gathered = torch.zeros(world, 1200, dtype=torch.float64, device="cuda")
                all_gather(gathered, dq)
                torch.cuda.current_stream().synchronize()
                sc.insert_descriptors_device(gathered.data_ptr(), world)
                rec1 = torch.zeros(world * 3 * 24, dtype=torch.uint8, device="cuda")
                sc.shard_query_batch_device(gathered.data_ptr(), book.step(world), rec1.data_ptr())
                sc.sync()
                allr = torch.zeros(world, world * 3 * 24, dtype=torch.uint8, device="cuda")
                all_gather(allr, rec1)
                rec = allr.cpu().numpy().reshape(world, world, 3, 24)[:, rank]
                r = S.merge_candidates([S.SCCand.from_buffer_copy(rec[s, c].tobytes()) for s in range(world) for c in range(3)], 0.4)
            account_step(mst, ost, r)
        mp.finish()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_region(0, W)
    S.prof_reset()
    for key in stats:
        stats[key] = 0
    for key in lm_bytes:
        lm_bytes[key] = 0
    import gc
    gc.collect()
    gc.disable()  # a generation-2 collection inside a ~6-40 ms timed region would be a visible fraction of it
    fence()
    t_wall0 = time.time()
    n_prof_steps = [0]
    rep_dt = []
    # Timestamps on the dispatches cost throughput (tools/gpu_prof_overhead.sh), so inside the timed region only the two solve kernels
    # of the roofline line carry them, on every N-th step; the per-kernel table comes from extra steps behind the timed ones.
    LM_FILTER = "k_lm_solve_map,k_lm_solve_odom"
    lm_only = a.prof_every > 0 and not a.timeline
    for rep in range(R):  # the same K-step region on consecutive scans of the sequence; the state (map, poses, database) carries on
        k0 = W + rep * K

        def plan(k, rep=rep, k0=k0):
            if a.timeline:
                on = True if a.timeline_kernels else (rep == R - 1 and K // 2 <= k - k0 < K // 2 + 6)
                S.prof_timeline(on)
                S.prof_enable(on, a.timeline_kernels or None)
            else:
                on = lm_only and (k - W) % a.prof_every == 0
                S.prof_enable(on, LM_FILTER)
            n_prof_steps[0] += on
        prof_plan["fn"] = plan
        t0 = time.perf_counter()
        run_region(k0, K)
        fence()
        rep_dt.append(time.perf_counter() - t0)
    prof_plan["fn"] = None
    S.prof_enable(False)
    t_wall1 = time.time()
    final_pose_resident = dict(last_pose)
    loops_timed = int(stats["loops"])
    dt_h2d = None
    if do_h2d:  # PCIe-inclusive leg: every scan starts in (pageable) host memory; never reported as `value`
        k0 = W + R * K
        t0 = time.perf_counter()
        run_region(k0, K, h2d=True)
        fence()
        dt_h2d = time.perf_counter() - t0
    dt = float(np.median(rep_dt))
    gc.enable()
    prof = S.prof_read_all()
    prof_all, n_all_steps = prof, n_prof_steps[0]
    if P_STEPS > 0:  # per-kernel table: every instrumented launch timed, outside the timed region; the run's statistics are put back
        import copy
        keep_stats, keep_lm = copy.deepcopy(stats), copy.deepcopy(lm_bytes)
        S.prof_reset()
        S.prof_enable(True)
        run_region(W + K * (R + (1 if do_h2d else 0)), P_STEPS)
        fence()
        S.prof_enable(False)
        prof_all, n_all_steps = S.prof_read_all(), P_STEPS
        stats.update(keep_stats)
        lm_bytes.update(keep_lm)
    if a.timeline:
        S.prof_timeline_dump(a.timeline)
    # sizes of one representative scan (outside the timed region) for the algorithmic-byte formulas
    kr = W + R * K - 1
    rz = S.ScanRegistration(S.HDL64, 5.0, max_points=cap, device=local)
    rz.run_device(d_scans[kr].data_ptr(), npts[kr], 3)
    fz = rz.fetch()
    kf = S.VoxelGrid(cap, device=local)
    n_kf = kf.filter(fz["cloud"], 0.4).shape[0]  # the keyframe cloud ScanContext builds its descriptor from (:629-631)
    rz.close(), kf.close()
    nsteps = K * (R + (1 if do_h2d else 0))
    per = lambda key: stats[key] / max(1, nsteps)  # noqa: E731
    counts = dict(n_in=npts[kr], n_kept=fz["n_kept"], n_sharp=len(fz["sharp"]), n_less_sharp=len(fz["less_sharp"]), n_flat=len(fz["flat"]),
                  n_less_flat=fz["less_flat"].shape[0], n_keyframe_ds=n_kf, stack_pts=per("stack_pts"), blocks=stats["blocks"] / max(1, 2 * nsteps),
                  map_pts=per("map_pts"), win_pts=per("win_pts"), edge=per("edge"), plane=per("plane"), evals=per("evals"),
                  odom_blocks=per("odom_blocks"), odom_evals=per("odom_evals"), n_db=a.sc_db + W + nsteps // 2, n_scans_sensor=64, lm=lm_bytes)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        value = world * K / dt
        roofline = roofline_of(prof, counts, pipelined, prof_all, n_all_steps)
        cpu = cpp = None
        if world == 1 and a.cpu_sample > 0:
            cpu = cpu_baseline(scans[: min(total, a.cpu_sample)], a.sc_db)
        batched = None
        if world == 1 and a.seqs > 1 and pipelined:
            if P is not None:
                P.close()
                P = None
            batched = run_batched(a, S, min(a.seqs, 4), value, local, threads)
        as_integrated = None
        if world == 1 and a.cpp_sample > 0:
            from scaloam import formats
            n_cpp = min(total, a.cpp_sample + 5)
            with tempfile.TemporaryDirectory() as td:
                f = os.path.join(td, "scans.bin")
                formats.write_scan_stream(f, scans[:n_cpp])
                if P is not None:
                    P.close()  # the child builds its own contexts (5.3 GB of grid pools each): free ours first
                    P = None
                integ = cpp_host("integrated", f, 5, a.sc_db)
                integ1 = cpp_host("integrated", f, 5, a.sc_db, stream_mode=1)
                cpp = {"integrated": integ, "integrated_own_streams": integ1, "pipeline_resident": cpp_host("pipeline", f, 5, a.sc_db, resident=1),
                       "pipeline_host_scans": cpp_host("pipeline", f, 5, a.sc_db, resident=0)}
            if "error" not in integ:
                as_integrated = {"value": integ["scans_per_s"], "unit": "scans/s", "ms_per_scan": integ["ms_per_scan"],
                                 "latency_ms_p50": integ["latency_ms"]["p50"], "latency_ms_p99": integ["latency_ms"]["p99"], "scans": integ["scans"],
                                 "note": "sc-a-loam_amd/host/replay_main.cpp --mode integrated (C++, a child process): the synchronous host-array entry "
                                         "points scal_features_run / scal_odom_step / scal_map_step / scal_voxel_downsample + scal_sc_insert_cloud + "
                                         "scal_sc_detect exactly as INTEGRATION.md sections 1-4 place them in the reference's four nodes, one thread per "
                                         "stage, clouds handed over as host arrays; latency = scan in -> mapping pose on the host; not the metric's value",
                                 "own_streams": None if "error" in integ1 else {
                                     "value": integ1["scans_per_s"], "ms_per_scan": integ1["ms_per_scan"], "latency_ms_p50": integ1["latency_ms"]["p50"],
                                     "note": "the same with scal_set_stream_mode(1): every node's context on its own stream, as four separate node "
                                             "processes have it (in one process and the default mode all contexts share one in-order stream)"}}
        out = {
            "metric": METRIC, "value": value, "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "repetitions": R, "timed_total_s": float(sum(rep_dt)), "rep_ms_per_step": [x / K * 1e3 for x in rep_dt],
            "h2d_inclusive": None if dt_h2d is None else {
                "value": world * K / dt_h2d, "unit": "scans/s", "ms_per_step": dt_h2d / K * 1e3,
                "note": "one more repetition of the K steps with every scan starting in pageable host memory: copy into pinned staging + "
                        "asynchronous upload on stage A's stream inside the timed region (scal_pipeline_push_host); not the metric's value"},
            "as_integrated": as_integrated, "batched": batched,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 points / f64 pose algebra",
            "data": "synthetic",
            "config": {"workload": "KITTI-like HDL-64 (64 beams x 1900 az, seeded procedural world, ~95k pts after the reference's ring filter) "
                                   "scan-to-map: 2 outer x <=4 LM iterations edge+surf correspondence + JtJ, with stage A features, stage B "
                                   "odometry prior and ScanContext insert+detect per scan",
                       "points_per_scan_in": int(np.mean(npts)), "sc_db_keyframes": a.sc_db, "sc_db_revisits": n_rev, "line_res": 0.4, "plane_res": 0.8,
                       "parallelism": (f"replicas for A-C, SC database sharded i % N with two RCCL all-gathers per {Q} scan(s)" if world > 1 else "single GPU"),
                       "schedule": "scal_pipeline (in-library, five C++ host threads): one stream per stage, consecutive scans overlap as the "
                                   "reference's four nodes do" if pipelined else "serial: one scan at a time through the per-stage calls"},
            "roofline": roofline, "cpu_baseline": cpu, "cpp_host": cpp,
            "kernel_ms_per_step": {k: v[0] / max(1, n_all_steps) for k, v in sorted(prof_all.items())}, "profiled_steps": n_all_steps,
            "roofline_timed_steps": n_prof_steps[0],
            "loops_detected": loops_timed, "input_gen_s": gen_s, "database_gen_s": db_s, "final_map_pose": final_pose_resident,
            "timed_window_unix": [t_wall0, t_wall1],
        }
        print(json.dumps(out))
    if world > 1:
        xq.put(None)
        dist.barrier()
        dist.destroy_process_group()
    return out


# SURVEY.md section 8d's algorithmic bytes per scan, evaluated on the run's own counts
def stage_bytes(c):
    A = 12.0 * c["n_in"] + 16.0 * c["n_kept"] + 16.0 * (c["n_sharp"] + c["n_less_sharp"] + c["n_flat"] + c["n_less_flat"])
    E = 72.0 * c["edge"] + 56.0 * c["plane"]  # both outer iterations together
    Pw, M, N = c["win_pts"], c["stack_pts"], c["n_kept"]
    C = 16.0 * (2 * Pw + 2 * Pw + 2 * N) + 2 * 16.0 * M * 6 + E + c["lm"]["map"] / max(1, c["lm"]["map_launches"]) * 2
    Tc, Ts = c["n_less_sharp"], c["n_less_flat"]
    w = 5.0 / max(1, c["n_scans_sensor"] * 0.8) * Ts  # points of the target cloud inside the +-2 ring window a plane query walks (:402-455)
    ob = (c["lm"]["odom"] + c["lm"].get("odom_E", 0.0)) / max(1, c["lm"]["odom_launches"]) * 2  # blocks written once, read once per evaluation
    B = 16.0 * (Tc + Ts) + 2 * 16.0 * (c["n_sharp"] + c["n_flat"]) * (1 + w) + ob
    D = 80.0 * c["n_db"] + 4 * 9600.0 + 16.0 * c["n_kept"] + 16.0 * c["n_keyframe_ds"]
    return {"A": A, "B": B, "C": C, "D": D}


STAGE_KERNELS = {
    "A": ("k_pre", "k_classify", "k_ringscan", "k_scatter", "k_curv", "k_ring", "k_compact"),
    "B": ("k_odom_gather", "k_odom_assoc", "k_lm_solve_odom", "k_odom_handover", "k_odom_cellscan", "k_odom_cellfill"),
    "C": ("k_map_gather", "k_map_begin", "k_grid_build", "k_grid_count", "k_grid_alloc", "k_grid_fill", "k_grid_clear", "k_assoc_knn", "k_assoc_knn.0", "k_assoc_fit",
          "k_lm_solve_map", "k_merge_keys", "k_merge_lookup", "k_merge_write", "k_map_end", "k_insert_keys", "k_map_heads", "k_map_reduce",
          "k_transform_cloud", "k_scan"),
    "D": ("k_sc_bin", "k_sc_finish", "k_sc_keys", "k_sc_store", "k_sc_topk", "k_sc_detect"),
}


def roofline_of(prof, c, pipelined=True, prof_all=None, n_all_steps=0):
    """Roofline line of the dominant kernel: the stage-C LM solve (k_lm_solve, the largest single kernel of the pose chains).
    achieved = ALGORITHMIC bytes per launch / average launch duration, both measured in THIS run:
      bytes  = SURVEY.md section 8d's per-unit figures - 72 B per edge block, 56 B per plane block, read once per evaluation - times
               the residual blocks and evaluations (1 + LM iterations) the solves of the timed steps really had (scal_map_stats);
      time   = HIP events attached to the dispatches on stage C's stream (sampled steps).
    Stage B's solves run the same kernel on ~10x fewer blocks and are priced separately (`stage_b`).  `stages`: section 8d's per-scan
    byte formulas over the summed kernel time of each stage (extra, untimed steps with every launch timed)."""
    if not prof:
        return None
    lm = c["lm"]

    def line(key, nbytes, launches):
        if key not in prof or not prof[key][1] or not launches:
            return None
        ms, cnt = prof[key]
        avg_s = ms / cnt * 1e-3
        per_launch = nbytes / launches
        return {"kernel": key, "achieved": per_launch / avg_s / 1e9, "avg_launch_us": avg_s * 1e6, "timed_launches": cnt,
                "algorithmic_bytes_per_launch": per_launch}

    lc = line("k_lm_solve_map", lm["map"], lm["map_launches"])
    lb = line("k_lm_solve_odom", lm["odom"], lm["odom_launches"])
    if lc is None:
        return None
    # HBM traffic of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of this
    # benchmark: bench.py cannot collect counters on itself); null when the file is missing
    traffic, src = None, None
    for name in ("r03_pmc_fetch_write.json", "r02_pmc_fetch_write.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            k = pmc["kernels"].get("k_lm_solve_map")
            if k:
                traffic = (k["fetch_kb_per_dispatch"] + k["write_kb_per_dispatch"]) * 1024.0
                src = f"profiles/{name}: " + pmc.get("config", "")
                break
        except (OSError, KeyError, ValueError):
            pass
    stages = None
    if prof_all and n_all_steps and "n_in" in c:
        sb = stage_bytes(c)
        stages = {}
        for st, names in STAGE_KERNELS.items():
            # "k_rs_scatter.C+D": one launch carrying stage C's surf filter and ScanContext's keyframe filter - charged half and half
            ms = sum(v[0] / (k.count("+") + 1) for k, v in prof_all.items() if k in names or st in k.rsplit(".", 1)[-1].split("+") and "." in k) / n_all_steps
            if ms > 0:
                gbs = sb[st] / (ms * 1e-3) / 1e9
                stages[st] = {"algorithmic_bytes_per_scan": sb[st], "kernel_ms_per_scan": ms, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    out = {"bound": "hbm", "kernel": "k_lm_solve (stage C)", "achieved": lc["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": lc["achieved"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
           "avg_launch_us": lc["avg_launch_us"], "timed_launches": lc["timed_launches"],
           "algorithmic_bytes_per_launch": lc["algorithmic_bytes_per_launch"],
           "blocks_per_launch": lm["map_blocks"] / max(1, lm["map_launches"]), "evaluations_per_launch": lm["map_evals"] / max(1, lm["map_launches"]),
           "bytes_formula": "(72 B x edge blocks + 56 B x plane blocks) x (1 + LM iterations), SURVEY.md section 8d",
           "stage_b": lb, "stages": stages,
           "stages_note": "per scan: SURVEY.md section 8d's byte formulas (A: 12 N_in + 16 N_kept + 16 features; B: targets + 2 x queries x (1 + ring "
                          "window) + evaluations; C: 16 (4 P + 2 N) + 2 x 16 M x 6 + (1 + evaluations) x E; D: 80 n_db + 38,400 + the keyframe "
                          "VoxelGrid) over the summed device time of the stage's kernels",
           "note": "latency-bound by design: <= 5 dependent evaluation rounds of ~9 us on <= 48 workgroups (DESIGN.md section 6)"}
    return out


def cpu_baseline(scans, sc_db, sensor="HDL64", min_range=5.0, sc_thres=0.4, gate=None):
    """The oracle (dependency-free CPU restatement of the reference path, g++ -O3, one thread per stage) on the same scans:
    (iii) the serial sum on one core and (ii) the pipelined figure 1 / max(stage) - the reference's four ROS nodes run as four
    processes, so its throughput on >= 4 cores is bounded by its slowest stage, not by the sum."""
    import oracle_py as O
    oo, om, osc = O.Odometry(), O.Mapper(0.4, 0.8), O.SCManager(max_radius=80.0, dist_thres=sc_thres)
    rng = np.random.default_rng(4242)
    for d in synth_descs(rng, sc_db):
        osc.saveScancontextAndKeys(d.T)
    kg = None
    if gate:
        from scaloam.pgo import KeyframeGate
        kg = KeyframeGate(*gate)
    t_stage = np.zeros(4)
    t0 = time.perf_counter()
    for xyz in scans:
        ta = time.perf_counter()
        f = O.features(xyz, getattr(O, sensor), min_range)
        c = f["cloud"]
        tb = time.perf_counter()
        x = oo.step(c[f["sharp"]], c[f["less_sharp"]], c[f["flat"]], f["less_flat"])
        tc = time.perf_counter()
        qm, tm, _, _ = om.step(c[f["less_sharp"]], f["less_flat"], c, x[2], x[3], want_registered=True)
        td = time.perf_counter()
        if kg is None or kg(qm, tm):
            ds, _ = O.voxel_grid(c, 0.4)
            osc.makeAndSaveScancontextAndKeys(ds)
            osc.detectLoopClosureID()
        te = time.perf_counter()
        t_stage += [tb - ta, tc - tb, td - tc, te - td]
    dt = time.perf_counter() - t0
    n = len(scans)
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    per = t_stage / n
    return {"value": n / dt, "unit": "scans/s", "cores": 1, "kind": "port",
            "sample": f"first {n} scans of the same sequence through the oracle (A+B+C+D serial on one core, kd-tree kNN)",
            "ms_per_scan": {"features": per[0] * 1e3, "odometry": per[1] * 1e3, "mapping": per[2] * 1e3, "scancontext": per[3] * 1e3},
            "pipelined_scans_per_s": 1.0 / per.max(), "pipelined_cores": 4,
            "pipelined_note": "1 / max(stage): what four processes (the reference's four ROS nodes), one core each, would sustain; "
                              "computed from the per-stage times of the one-core run above",
            "cpu_model": model, "host": os.uname().nodename, "nproc": os.cpu_count()}


if __name__ == "__main__":
    main()
