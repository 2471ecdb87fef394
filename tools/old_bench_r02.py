#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config #2: scans/sec through feature extraction + scan-to-map ICP +
ScanContext loop search on KITTI-like HDL-64 scans (~120k points), one MI355X per process.

A "step" is one scan through the whole hot path on one GPU, inputs already resident in HBM:
  stage A  scal_features_run_device                      (scanRegistration.cpp:134-421)
  stage B  scal_odom_enqueue_features / scal_odom_collect  (laserOdometry.cpp:267-568)   - provides the prior for stage C
  stage C  scal_map_prefetch_features / _enqueue_features / _collect (laserMapping.cpp:310-802,:845-849)   2 outer x <=4 LM iterations
  stage D  scal_sc_insert_features + scal_sc_detect_*    (Scancontext.cpp:151-260, :336-427) over a pre-filled keyframe DB
Default schedule: stage-pipelined - every stage on its own stream and consecutive scans overlapping, the way the reference's four
ROS nodes (scanRegistration, laserOdometry, laserMapping, laserPosegraphOptimization) process different scans at the same time;
every scan still goes through A -> B -> C and A -> D with the reference's data dependencies (C(k) registers against the map that
contains scan k-1), and the K timed steps end only when the last scan's map insertion is done.  --no-overlap runs one scan
at a time on one stream; both schedules give bit-identical poses (tools/gpu_sched_check.sh).
N > 1 (one process per GPU, torch.distributed/RCCL): stages A-C do not shard (pose k+1 depends on pose k and on the
map), so every rank replays its own seeded sequence ("replicas only", weak scaling); the ScanContext keyframe database
IS sharded (keyframe i on rank i % N) and every step exchanges descriptors and per-shard top-3 records with two RCCL
all-gathers.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the launching stream for the
dominant kernel over the timed region; `cpu_baseline` is the oracle (CPU restatement of the reference path) timed on
this host's cores on a bounded sample of the same scans (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# The stage pipeline keeps five HIP streams busy (A, B, C, C's prefetch, D) next to torch's; the runtime multiplexes streams
# onto 4 hardware queues by default, which would serialise stages that share a queue.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in ("oracle", os.path.join("sc-a-loam_amd", "python"), os.path.join("tools", "synth")):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, help="2: KITTI-like HDL-64 (BASELINE.json's metric); 3: MulRan-like OS1-64 stream harness")
    ap.add_argument("--replay", default="lockstep", help="config 3: lockstep (every scan through every stage, as fast as possible) or realtime")
    ap.add_argument("--rate", type=float, default=10.0, help="config 3, realtime: scans per second of the replay (10 Hz sensor)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3, help="the timed region of K steps is run this many times on consecutive scans; value = median")
    ap.add_argument("--h2d", type=int, default=1, help="N=1: one more repetition with every scan uploaded from host memory inside the timed region (reported beside value)")
    ap.add_argument("--sc-db", type=int, default=5000, help="keyframes pre-filled into the ScanContext database")
    ap.add_argument("--cpu-sample", type=int, default=60, help="scans timed through the CPU oracle (0 = skip)")
    ap.add_argument("--seed", type=int, default=205)
    ap.add_argument("--side-thread", type=int, default=2, help="host threads besides the main one that queue work: 0 = none, 1 = one for stage A + "
                    "stage C's prefetch + ScanContext, 2 = ScanContext on its own thread, 3 = stage B too: four host threads, as the "
                    "reference runs four processes (the main thread keeps stage C)")
    ap.add_argument("--ring", type=int, default=6, help="features contexts used in turn by the stage pipeline")
    ap.add_argument("--host-timing", action="store_true", help="report the host time spent inside each library call (us per step)")
    ap.add_argument("--no-overlap", action="store_true", help="one scan at a time on one stream (no stage pipelining)")
    ap.add_argument("--prof-every", type=int, default=8,
                    help="0 = no kernel timing at all; otherwise the two LM solve kernels (the roofline line) carry start/stop timestamps on every "
                         "N-th step of the timed region, and every instrumented kernel on --prof-steps extra steps behind it (outside the timing)")
    ap.add_argument("--prof-steps", type=int, default=24, help="extra, untimed steps with per-kernel timestamps on every launch (kernel_ms_per_step)")
    ap.add_argument("--timeline-kernels", default="", help="with --timeline: only these kernels (comma separated), over the whole timed region")
    ap.add_argument("--timeline", default="", help="development aid: write (kernel, start ms, stop ms) of every dispatch of six timed steps to this CSV")
    ap.add_argument("--sync-dir", default="", help="start the timed region together with --sync-n other bench.py processes (ready files in this directory)")
    ap.add_argument("--sync-n", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) for real runs; gloo only to rehearse N>1 on fewer GPUs")
    return ap.parse_args()


def synth_descs(rng, n):
    """ScanContext-like descriptors (occupancy ~0.5, heights -2..18 m, a few empty sectors)."""
    d = rng.uniform(-2.0, 18.0, (n, 60, 20)) * (rng.uniform(size=(n, 60, 20)) < 0.5)
    for i in range(n):
        d[i, rng.integers(0, 60, 3), :] = 0.0
    return d  # [n][sector][ring] == column-major 20x60


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
    world_gen = scansynth.World(scansynth.OS1_64, 301, threads=os.cpu_count() or 8)
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

    def one(k, t_arrival=None):
        reg.run_device(d_scans[k].data_ptr(), scans[k].shape[0], 3)
        qlc, tlc, qw, tw, _ = od.step_features(reg)
        qm, tm, _ = mp.process_features(reg, qw, tw)
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
    lat = np.array(out["lat"]) * 1e3 if out["lat"] else None
    print(json.dumps({
        "metric": "scans/sec, MulRan-like OS1-64 stream: full odom + mapping, ScanContext on keyframes (BASELINE config #3)",
        "value": K / dt, "unit": "scans/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32 points / f64 pose algebra", "data": "synthetic",
        "config": {"workload": "OS1-64 (64 beams x 1024 columns, seed 301, minimum_range 0.5, line/plane 0.4/0.8, keyframe gap 1 m / 10 deg, "
                               "sc_dist_thres 0.2), replay " + a.replay + (f" at {a.rate:g} Hz" if a.replay != "lockstep" else ""),
                   "points_per_scan_in": int(np.mean([s.shape[0] for s in scans])), "sc_db_keyframes": a.sc_db},
        "keyframes": out["keyframes"], "loops_detected": int(out["loops"]), "scans_dropped_by_mapping": out["dropped"],
        "latency_ms": None if lat is None else {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
        "final_map_pose": {"q": pose[0].tolist(), "t": pose[1].tolist()}}))


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

    K, W = a.steps, a.warmup
    R = max(1, a.reps)
    do_h2d = bool(a.h2d) and world == 1
    P_STEPS = a.prof_steps if (a.prof_every > 0 and not a.timeline) else 0
    total = W + K * (R + (1 if do_h2d else 0)) + P_STEPS
    # ---- synthetic HDL-64 sequence for this rank (weak scaling: one independent sequence per GPU)
    threads = max(1, (os.cpu_count() or 8) // max(1, world))
    world_gen = scansynth.World(scansynth.HDL64, a.seed + 1000 * rank, threads=threads)
    t0 = time.time()
    scans = [world_gen.scan(k) for k in range(total)]
    gen_s = time.time() - t0
    npts = [s.shape[0] for s in scans]
    d_scans = [torch.from_numpy(s).cuda(local) for s in scans]  # inputs resident in HBM before the timed region
    cap = max(npts) + 1024

    pipelined = not a.no_overlap
    # One stream per stage, consecutive scans overlapping like the reference's four ROS nodes (scanRegistration, laserOdometry,
    # laserMapping, laserPosegraphOptimization run concurrently on different scans); two features contexts used alternately.
    S.set_stream_mode(1 if pipelined else 0)
    # pipelined: a features context is not run again before the stage-C step that used it has been collected (scal_map_enqueue_features)
    regs = [S.ScanRegistration(S.HDL64, 5.0, max_points=min(400000, cap), device=local) for _ in range((max(a.ring, 8) if world > 1 else a.ring) if pipelined else 1)]  # N > 1: the exchange thread may lag 3 scans
    reg = regs[0]
    od = S.LaserOdometry(max_points=cap, device=local)
    mp = S.LaserMapping(0.4, 0.8, max_scan_points=cap, max_map_points=4000000, device=local)
    # N > 1, pipelined: the database shard gets its own stream (lane 5) so that its small insert/query kernels never queue behind
    # the next scan's descriptor build, which runs in a separate builder context on the stage-D stream
    sc = S.SCManager(max_radius=80.0, dist_thres=0.4, max_keyframes=a.sc_db // world + total * world + 64, device=local,
                     n_shards=world, shard=rank, side_stream=5 if (pipelined and world > 1) else 0)
    sc_build = S.SCManager(max_radius=80.0, dist_thres=0.4, max_keyframes=8, device=local) if (pipelined and world > 1) else None
    rng = np.random.default_rng(4242)
    for d in synth_descs(rng, a.sc_db):
        sc.saveScancontextAndKeys(d.T)  # every shard sees every insert and keeps the ones it owns

    if world > 1:
        d_q = [torch.zeros(1200, dtype=torch.float64, device="cuda") for _ in range(2)]  # scan k+1's descriptor is queued early
        all_q = torch.zeros(world, 1200, dtype=torch.float64, device="cuda")
        d_rec = torch.zeros(world * 3 * 24, dtype=torch.uint8, device="cuda")
        all_rec = torch.zeros(world, world * 3 * 24, dtype=torch.uint8, device="cuda")
    sc_state = dict(counter=0, size_at_rebuild=0, n_global=a.sc_db)
    stats = dict(loops=0, blocks=0, stack_pts=0, solved=0, map_pts=0)
    # algorithmic bytes of the LM solves (72 B per edge block, 56 B per plane block, read once per evaluation;
    # evaluations of a solve = 1 + its iterations), stage C and stage B separately
    lm_bytes = dict(map=0.0, map_launches=0, odom=0.0, odom_launches=0, map_blocks=0, map_evals=0)

    def lm_account(which, st_):
        for o in range(2):
            ne, npl, it = st_.n_edge[o], st_.n_plane[o], st_.lm_iters[o]
            if ne + npl == 0:
                continue
            lm_bytes[which] += (72.0 * ne + 56.0 * npl) * (1 + it)
            lm_bytes[which + "_launches"] += 1
            if which == "map":
                lm_bytes["map_blocks"] += ne + npl
                lm_bytes["map_evals"] += 1 + it

    host_t = {}

    def timed(name, fn, *args):
        if not a.host_timing:
            return fn(*args)
        t = time.perf_counter()
        r = fn(*args)
        host_t[name] = host_t.get(name, 0.0) + time.perf_counter() - t
        return r

    sc_ext = torch.cuda.ExternalStream(sc.stream_ptr()) if (world > 1 and a.backend == "nccl") else None

    def sc_sharded(k, queued=False):
        """stage D with the database sharded over the ranks: two all-gathers (descriptors, candidate records).  Under RCCL the
        collectives (torch's stream) and the shard's kernels (the library's stream, scal_sc_stream) are ordered against each other
        on the device; the host only waits for the final records.  The gloo rehearsal stages through the host and synchronises."""
        if queued:
            sc_build.wait_descriptor()  # the oldest queued descriptor (scan k); younger builds keep running
        else:
            sc.make_features(reg, d_q[k % 2].data_ptr())
        all_gather(all_q, d_q[k % 2])
        if sc_ext is not None:
            sc_ext.wait_stream(torch.cuda.current_stream())
        else:
            torch.cuda.current_stream().synchronize()
        sc.insert_descriptors_device(all_q.data_ptr(), world)  # global insertion order: rank 0..N-1 of this step, one launch
        sc_state["n_global"] += world
        # detectLoopClosureID's tree period (Scancontext.cpp:353-365), one query per rank in global order
        limits = []
        for rr in range(world):
            if sc_state["counter"] % 30 == 0:
                sc_state["size_at_rebuild"] = sc_state["n_global"]
            sc_state["counter"] += 1
            limits.append(sc_state["size_at_rebuild"])
        sc.shard_query_batch_device(all_q.data_ptr(), limits, d_rec.data_ptr())  # every query against its own tree size
        if sc_ext is not None:
            torch.cuda.current_stream().wait_stream(sc_ext)
        else:
            sc.sync()
        all_gather(all_rec, d_rec)
        rec = all_rec.cpu().numpy().reshape(world, world, 3, 24)[:, rank]  # shard s's three records for my query (the one host wait)
        cands = [S.SCCand.from_buffer_copy(rec[s, j].tobytes()) for s in range(world) for j in range(3)]
        return S.merge_candidates(cands, 0.4)

    last_pose = {}

    def account(mst, r):
        lm_account("map", mst)
        stats["loops"] += r["loop_id"] >= 0
        stats["blocks"] += mst.n_edge[0] + mst.n_plane[0] + mst.n_edge[1] + mst.n_plane[1]
        stats["stack_pts"] += mst.n_corner_stack + mst.n_surf_stack
        stats["solved"] += mst.solved
        stats["map_pts"] += mst.n_map_corner_total + mst.n_map_surf_total

    mode = dict(h2d=False)

    def stage_a(r_, k):
        """stage A of scan k: from the copy resident in HBM, or (h2d leg) from host memory through pinned staging + async upload"""
        if mode["h2d"]:
            r_.enqueue_host(scans[k])
        else:
            r_.run_device(d_scans[k].data_ptr(), npts[k], 3)

    def step_serial(k):
        """one scan at a time: A -> B -> C -> D, each stage finished before the next starts"""
        timed("A.run_device", stage_a, reg, k)
        qlc, tlc, qw, tw, ost = od.step_features(reg)
        lm_account("odom", ost)
        qm, tm, mst = timed("C.process", mp.process_features, reg, qw, tw)
        last_pose["q"], last_pose["t"] = qm.tolist(), tm.tolist()
        if world == 1:
            sc.insert_features(reg)
            r = sc.detectLoopClosureID()
        else:
            r = sc_sharded(k)
        account(mst, r)

    # ---- stage-pipelined schedule.  Nothing on the pose chains waits for the host any more: stage C composes its prior, decides
    # the rolling window and tracks the map sizes on the device, so scan k's stage C is queued while scan k-1's (and k-2's) still
    # runs; stage B is kept one scan ahead of the pose it hands to stage C, stage A two scans ahead.
    pipe = dict(c_inflight=[], loops={}, side_inflight=False, b_queued=-1, started=False)
    PENDING = object()
    C_DEPTH = 2   # stage-C steps queued and not collected (the library allows 4)

    # A second host thread queues the side-stream work and stage A (the library calls release the GIL).
    import queue
    import threading
    side_q, side_done = queue.Queue(), queue.Queue()
    side2_q, side2_done = queue.Queue(), queue.Queue()

    def side_worker(q_in, q_out):
        torch.cuda.set_device(local)
        while True:
            job = q_in.get()
            if job is None:
                return
            try:
                job()
                q_out.put(None)
            except Exception as e:  # surfaced in the main thread
                q_out.put(e)

    side_thread = threading.Thread(target=side_worker, args=(side_q, side_done), daemon=True) if (pipelined and a.side_thread) else None
    if side_thread:
        side_thread.start()
    # ScanContext's calls (keyframe filter, descriptor, insert, search, collect) from a thread of their own: ~20 launches per scan
    side2_thread = threading.Thread(target=side_worker, args=(side2_q, side2_done), daemon=True) if (pipelined and a.side_thread >= 2 and world == 1) else None
    if side2_thread:
        side2_thread.start()

    # N > 1: the sharded ScanContext step blocks on two collectives and a few synchronisations per scan; a third host thread
    # runs it (same order on every rank) so that stages A-C of the following scans keep being queued meanwhile.
    xchg_q, loop_q = queue.Queue(), queue.Queue()

    def xchg_worker():
        torch.cuda.set_device(local)
        prev = None
        while True:
            job = xchg_q.get()
            if job is None:
                return
            try:
                if job != "flush":
                    k, r_ = job
                    sc_build.make_features_enqueue(r_, d_q[k % 2].data_ptr())  # scan k's descriptor starts building ...
                if prev is not None:
                    loop_q.put(sc_sharded(prev, True))                         # ... while scan k-1's is exchanged and searched
                prev = None if job == "flush" else k
            except Exception as e:
                loop_q.put(e)

    xchg_thread = threading.Thread(target=xchg_worker, daemon=True) if (pipelined and world > 1) else None
    if xchg_thread:
        xchg_thread.start()

    def loop_result(k):
        if world == 1:
            return pipe["loops"].pop(k)
        r = loop_q.get()
        if isinstance(r, Exception):
            raise r
        return r

    def side_job(j, last):
        """Everything that only needs stage A, queued one iteration ahead of its use: for scan j+1 the stage-C prefetch, stage A of
        scan j+2, the loop answer of scan j (its search was queued by the previous job), then scan j+1's ScanContext insert + search."""
        r_ = regs[(j + 1) % len(regs)] if j + 1 < last else None

        def run_a():
            if r_ is not None:
                timed("side.prefetch", mp.prefetch_features, r_)   # first: its gather + corner filter ride on stage A's stream
            if j + 2 < last:
                timed("A.run_device", stage_a, regs[(j + 2) % len(regs)], j + 2)
            if r_ is not None and world > 1:
                xchg_q.put((j + 1, r_))

        def run_d():
            if world != 1:
                return
            if r_ is not None:
                timed("D.insert", sc.insert_features, r_)
                timed("D.detect_enqueue", sc.detect_enqueue)
                pipe.setdefault("d_queued", set()).add(j + 1)
            if j in pipe.get("d_queued", ()):      # queued by the previous job: a whole period to finish
                pipe["loops"][j] = timed("D.detect_collect", sc.detect_collect)
                pipe["d_queued"].discard(j)

        def run():
            run_a()
            run_d()
        return run, run_a, run_d

    def run_side(jobs):
        both, job_a, job_d = jobs
        if side_thread and side2_thread:
            side_q.put(job_a)
            side2_q.put(job_d)
            pipe["side_inflight"] = 2
        elif side_thread:
            side_q.put(both)
            pipe["side_inflight"] = 1
        else:
            both()

    def join_side():
        if pipe["side_inflight"]:
            errs = [timed("side.join", side_done.get)]
            if pipe["side_inflight"] == 2:
                errs.append(timed("side2.join", side2_done.get))
            pipe["side_inflight"] = False
            for err in errs:
                if err is not None:
                    raise err

    def collect_c():
        k = pipe["c_inflight"].pop(0)
        qm, tm, mst = timed("C.collect", mp.collect)
        last_pose["q"], last_pose["t"] = qm.tolist(), tm.tolist()
        return k, mst

    # stage B from a thread of its own (--side-thread 3): "go k" = queue B(k+1), collect B(k), hand the pose to the main thread
    b_q, b_out = queue.Queue(), queue.Queue()

    def b_worker():
        torch.cuda.set_device(local)
        while True:
            job = b_q.get()
            if job is None:
                return
            try:
                k, nxt = job
                if nxt is not None:
                    timed("B.enqueue", od.enqueue_features, nxt)
                b_out.put(timed("B.collect", od.collect))
            except Exception as e:
                b_out.put(e)

    b_thread = threading.Thread(target=b_worker, daemon=True) if (pipelined and a.side_thread >= 3) else None
    if b_thread:
        b_thread.start()

    def step_pipelined(k, last):
        """Iteration k of the software pipeline: stage B of scan k+1 is queued before the pose of scan k's stage B is collected,
        that pose goes straight into scan k's stage C, which queues behind the stage-C steps still running; the oldest of those
        is collected when more than C_DEPTH are outstanding."""
        if not pipe["started"]:  # first scan of a run: what the previous iterations would have queued
            stage_a(regs[k % len(regs)], k)
            od.enqueue_features(regs[k % len(regs)])
            pipe["b_queued"] = k
            pipe["started"] = True
            side_job(k - 1, last)[0]()   # A(k+1), prefetch(k), D(k)
        join_side()                   # job k-1: A(k+1), prefetch(k), D(k) are queued
        run_side(side_job(k, last))
        nxt = None
        if k + 1 < last and pipe["b_queued"] < k + 1:
            nxt = regs[(k + 1) % len(regs)]
            pipe["b_queued"] = k + 1
        if b_thread:
            b_q.put((k, nxt))
            res = timed("B.wait_pose", b_out.get)
            if isinstance(res, Exception):
                raise res
            qlc, tlc, qw, tw, ost = res
        else:
            if nxt is not None:
                timed("B.enqueue", od.enqueue_features, nxt)
            qlc, tlc, qw, tw, ost = timed("B.collect", od.collect)
        lm_account("odom", ost)
        timed("C.enqueue", mp.enqueue_features, regs[k % len(regs)], qw, tw)
        pipe["c_inflight"].append(k)
        while len(pipe["c_inflight"]) > C_DEPTH:
            kk, mst = collect_c()
            pipe.setdefault("accounts", []).append((kk, mst))
        # N > 1: the exchange thread may lag two scans behind (their features contexts are still intact)
        while pipe.get("accounts") and (len(pipe["accounts"]) > 2 if world > 1 else pipe["accounts"][0][0] in pipe["loops"]):
            kk, mst = pipe["accounts"].pop(0)
            account(mst, timed("D.loop_result", loop_result, kk))

    def drain():
        join_side()
        while pipe["c_inflight"]:
            pipe.setdefault("accounts", []).append(collect_c())
        if world == 1:
            for j in sorted(pipe.get("d_queued", ())):
                pipe["loops"][j] = sc.detect_collect()
            pipe["d_queued"] = set()
        elif xchg_thread and pipe.get("accounts"):
            xchg_q.put("flush")  # the exchange of the last scan runs one job late
        for kk, mst in pipe.get("accounts", []):
            account(mst, loop_result(kk))
        pipe["accounts"] = []
        pipe["started"] = False
        mp.finish()

    step = step_pipelined if pipelined else (lambda k, last: step_serial(k))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(W):
        step(k, W)
    if pipelined:
        drain()
    S.prof_reset()
    host_t.clear()
    for key in stats:
        stats[key] = 0
    import gc
    gc.collect()
    gc.disable()  # a generation-2 collection inside a ~40 ms timed region would be a visible fraction of it
    fence()
    if a.sync_dir and a.sync_n > 1:  # several independent sequences on one GPU (tools/gpu_multi_seq.sh): common start
        open(os.path.join(a.sync_dir, f"ready_{os.getpid()}"), "w").close()
        while len([f for f in os.listdir(a.sync_dir) if f.startswith("ready_")]) < a.sync_n:
            time.sleep(0.0005)
    t_wall0 = time.time()
    n_prof_steps = 0
    rep_dt = []
    # Timestamps on the dispatches cost throughput (tools/gpu_prof_overhead.sh: -8 % and jitter even with only the two solve
    # kernels timed on every step), so inside the timed region only the two solve kernels of the roofline line carry them, on every
    # N-th step; the per-kernel table comes from extra steps behind the timed ones.
    LM_FILTER = "k_lm_solve_map,k_lm_solve_odom"
    lm_only = a.prof_every > 0 and not a.timeline
    for rep in range(R):  # the same K-step region on consecutive scans of the sequence; the state (map, poses, database) carries on
        k0 = W + rep * K
        t0 = time.perf_counter()
        for k in range(k0, k0 + K):
            if a.timeline:
                on = True if a.timeline_kernels else (rep == R - 1 and K // 2 <= k - k0 < K // 2 + 6)
                S.prof_timeline(on)
                S.prof_enable(on, a.timeline_kernels or None)
                n_prof_steps += on
            else:
                on = lm_only and (k - W) % a.prof_every == 0
                S.prof_enable(on, LM_FILTER)
                n_prof_steps += on
            step(k, k0 + K)
        if pipelined:
            drain()  # the last scan's pose and map insertion belong to the timed region
        fence()
        rep_dt.append(time.perf_counter() - t0)
    S.prof_enable(False)
    t_wall1 = time.time()
    final_pose_resident = dict(last_pose)
    dt_h2d = None
    if do_h2d:  # PCIe-inclusive leg: every scan starts in (pageable) host memory; never reported as `value`
        mode["h2d"] = True
        k0 = W + R * K
        t0 = time.perf_counter()
        for k in range(k0, k0 + K):
            step(k, k0 + K)
        if pipelined:
            drain()
        fence()
        dt_h2d = time.perf_counter() - t0
        mode["h2d"] = False
    dt = float(np.median(rep_dt))
    gc.enable()
    S.prof_enable(False)
    prof = S.prof_read_all()
    prof_all, n_all_steps = prof, n_prof_steps
    if P_STEPS > 0:  # per-kernel table: every instrumented launch timed, outside the timed region; the run's statistics are put back
        import copy
        keep_stats, keep_lm = copy.deepcopy(stats), copy.deepcopy(lm_bytes)
        S.prof_reset()
        S.prof_enable(True)
        k0 = W + K * (R + (1 if do_h2d else 0))
        for k in range(k0, k0 + P_STEPS):
            step(k, k0 + P_STEPS)
        if pipelined:
            drain()
        fence()
        S.prof_enable(False)
        prof_all, n_all_steps = S.prof_read_all(), P_STEPS
        stats.update(keep_stats)
        lm_bytes.update(keep_lm)
    if a.timeline:
        S.prof_timeline_dump(a.timeline)
    # sizes of one representative scan (outside the timed region) for the algorithmic-byte formulas
    kr = W + R * K - 1
    reg.run_device(d_scans[kr].data_ptr(), npts[kr], 3)
    fz = reg.fetch()
    nsteps = K * (R + (1 if do_h2d else 0))
    counts = dict(n_in=npts[kr], n_kept=fz["n_kept"], n_sharp=len(fz["sharp"]), n_less_sharp=len(fz["less_sharp"]), n_flat=len(fz["flat"]),
                  n_less_flat=fz["less_flat"].shape[0], stack_pts=stats["stack_pts"] / max(1, nsteps), blocks=stats["blocks"] / max(1, 2 * nsteps),
                  map_pts=stats["map_pts"] / max(1, nsteps), lm=lm_bytes)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        value = world * K / dt
        # ---- roofline of the dominant kernel (HBM bound): algorithmic bytes per launch / measured average launch time
        roofline = roofline_of(prof, max(1, n_prof_steps), counts, pipelined)
        cpu = None
        if world == 1 and a.cpu_sample > 0:
            cpu = cpu_baseline(scans[: min(total, a.cpu_sample)], a.sc_db)
        out = {
            "metric": "scans/sec (feat-extract + scan-to-map ICP + SC loop search), KITTI HDL-64",
            "value": value, "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "repetitions": R, "rep_ms_per_step": [x / K * 1e3 for x in rep_dt],
            "h2d_inclusive": None if dt_h2d is None else {
                "value": world * K / dt_h2d, "unit": "scans/s", "ms_per_step": dt_h2d / K * 1e3,
                "note": "one more repetition of the K steps with every scan starting in pageable host memory: copy into pinned staging + "
                        "asynchronous upload on stage A's stream inside the timed region (scal_features_enqueue_host); not the metric's value"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 points / f64 pose algebra",
            "data": "synthetic",
            "config": {"workload": "KITTI-like HDL-64 (64 beams x 1900 az, seeded procedural world, ~95k pts after the reference's ring filter) "
                                   "scan-to-map: 2 outer x <=4 LM iterations edge+surf correspondence + JtJ, with stage A features, stage B "
                                   "odometry prior and ScanContext insert+detect per scan",
                       "points_per_scan_in": int(np.mean(npts)), "sc_db_keyframes": a.sc_db, "line_res": 0.4, "plane_res": 0.8,
                       "parallelism": "replicas for A-C, SC database sharded i % N with RCCL all-gather" if world > 1 else "single GPU",
                       "schedule": "stage-pipelined: one stream per stage, consecutive scans overlap as the reference's four nodes do" if pipelined
                       else "serial: one scan at a time"},
            "roofline": roofline, "cpu_baseline": cpu,
            "kernel_ms_per_step": {k: v[0] / max(1, n_all_steps) for k, v in sorted(prof_all.items())}, "profiled_steps": n_all_steps,
            "roofline_timed_steps": n_prof_steps,
            "loops_detected": int(stats["loops"]), "input_gen_s": gen_s, "final_map_pose": final_pose_resident,
            "host_us_per_step": {k: v / K * 1e6 for k, v in host_t.items()} if a.host_timing else None,
            "timed_window_unix": [t_wall0, t_wall1],
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def roofline_of(prof, K, c, pipelined=True):
    """Roofline line of the dominant kernel: the stage-C LM solve (k_lm_solve, the largest single kernel of the pose chains).
    achieved = ALGORITHMIC bytes per launch / average launch duration, both measured in THIS run:
      bytes  = SURVEY.md section 8d's per-unit figures - 72 B per edge block, 56 B per plane block, read once per evaluation - times
               the residual blocks and evaluations (1 + LM iterations) the solves of the timed steps really had (scal_map_stats);
      time   = HIP events attached to the dispatches on stage C's stream (sampled steps).
    Stage B's solves run the same kernel on ~10x fewer blocks and are priced separately (`stage_b`)."""
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
    # HBM traffic of that kernel from the committed PMC passes of THIS build (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate
    # runs of this benchmark: bench.py cannot collect counters on itself); null when the file is missing
    traffic, src = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_fetch_write.json")))
        k = pmc["kernels"].get("k_lm_solve_map")
        if k:
            traffic = (k["fetch_kb_per_dispatch"] + k["write_kb_per_dispatch"]) * 1024.0
            src = "profiles/r02_pmc_fetch_write.json: " + pmc.get("config", "")
    except (OSError, KeyError, ValueError):
        pass
    out = {"bound": "hbm", "kernel": "k_lm_solve (stage C)", "achieved": lc["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": lc["achieved"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
           "avg_launch_us": lc["avg_launch_us"], "timed_launches": lc["timed_launches"],
           "algorithmic_bytes_per_launch": lc["algorithmic_bytes_per_launch"],
           "blocks_per_launch": lm["map_blocks"] / max(1, lm["map_launches"]), "evaluations_per_launch": lm["map_evals"] / max(1, lm["map_launches"]),
           "bytes_formula": "(72 B x edge blocks + 56 B x plane blocks) x (1 + LM iterations), SURVEY.md section 8d",
           "stage_b": lb,
           "share_of_step_ms": prof["k_lm_solve_map"][0] / K,
           "note": "latency-bound by design: <= 5 dependent evaluation rounds of ~9 us on <= 48 workgroups (DESIGN.md section 6)"}
    return out


def cpu_baseline(scans, sc_db):
    """The oracle (dependency-free CPU restatement of the reference path, g++ -O3, one thread per stage) on the same scans:
    (iii) the serial sum on one core and (ii) the pipelined figure 1 / max(stage) - the reference's four ROS nodes run as four
    processes, so its throughput on >= 4 cores is bounded by its slowest stage, not by the sum."""
    import oracle_py as O
    oo, om, osc = O.Odometry(), O.Mapper(0.4, 0.8), O.SCManager(max_radius=80.0, dist_thres=0.4)
    rng = np.random.default_rng(4242)
    for d in synth_descs(rng, sc_db):
        osc.saveScancontextAndKeys(d.T)
    t_stage = np.zeros(4)
    t0 = time.perf_counter()
    for xyz in scans:
        ta = time.perf_counter()
        f = O.features(xyz, O.HDL64, 5.0)
        c = f["cloud"]
        tb = time.perf_counter()
        x = oo.step(c[f["sharp"]], c[f["less_sharp"]], c[f["flat"]], f["less_flat"])
        tc = time.perf_counter()
        om.step(c[f["less_sharp"]], f["less_flat"], c, x[2], x[3], want_registered=True)
        td = time.perf_counter()
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
