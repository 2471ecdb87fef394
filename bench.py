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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--sc-db", type=int, default=5000, help="keyframes pre-filled into the ScanContext database")
    ap.add_argument("--cpu-sample", type=int, default=60, help="scans timed through the CPU oracle (0 = skip)")
    ap.add_argument("--seed", type=int, default=205)
    ap.add_argument("--side-thread", type=int, default=1, help="queue the side-stream work from a second host thread (0/1)")
    ap.add_argument("--ring", type=int, default=6, help="features contexts used in turn by the stage pipeline")
    ap.add_argument("--host-timing", action="store_true", help="report the host time spent inside each library call (us per step)")
    ap.add_argument("--no-overlap", action="store_true", help="one scan at a time on one stream (no stage pipelining)")
    ap.add_argument("--prof-every", type=int, default=8,
                    help="attach start/stop timestamps to the instrumented kernel launches on every N-th timed step (0 = never)")
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


def main():
    a = parse()
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
    total = K + W
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

    host_t = {}

    def timed(name, fn, *args):
        if not a.host_timing:
            return fn(*args)
        t = time.perf_counter()
        r = fn(*args)
        host_t[name] = host_t.get(name, 0.0) + time.perf_counter() - t
        return r

    def sc_sharded(k, queued=False):
        """stage D with the database sharded over the ranks: two all-gathers (descriptors, candidate records)"""
        if queued:
            sc_build.wait_descriptor()  # the oldest queued descriptor (scan k); younger builds keep running
        else:
            sc.make_features(reg, d_q[k % 2].data_ptr())
        all_gather(all_q, d_q[k % 2])
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
        sc.sync()
        all_gather(all_rec, d_rec)
        rec = all_rec.cpu().numpy().reshape(world, world, 3, 24)[:, rank]  # shard s's three records for my query
        cands = [S.SCCand.from_buffer_copy(rec[s, j].tobytes()) for s in range(world) for j in range(3)]
        return S.merge_candidates(cands, 0.4)

    last_pose = {}

    def account(mst, r):
        stats["loops"] += r["loop_id"] >= 0
        stats["blocks"] += mst.n_edge[0] + mst.n_plane[0] + mst.n_edge[1] + mst.n_plane[1]
        stats["stack_pts"] += mst.n_corner_stack + mst.n_surf_stack
        stats["solved"] += mst.solved
        stats["map_pts"] += mst.n_map_corner_total + mst.n_map_surf_total

    def step_serial(k):
        """one scan at a time: A -> B -> C -> D, each stage finished before the next starts"""
        timed("A.run_device", reg.run_device, d_scans[k].data_ptr(), npts[k], 3)
        qlc, tlc, qw, tw, ost = od.step_features(reg)
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

    def side_worker():
        while True:
            job = side_q.get()
            if job is None:
                return
            try:
                job()
                side_done.put(None)
            except Exception as e:  # surfaced in the main thread
                side_done.put(e)

    side_thread = threading.Thread(target=side_worker, daemon=True) if (pipelined and a.side_thread) else None
    if side_thread:
        side_thread.start()

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
        def run():
            r_ = regs[(j + 1) % len(regs)] if j + 1 < last else None
            if r_ is not None:
                timed("side.prefetch", mp.prefetch_features, r_)   # first: its gather + corner filter ride on stage A's stream
            if j + 2 < last:
                timed("A.run_device", regs[(j + 2) % len(regs)].run_device, d_scans[j + 2].data_ptr(), npts[j + 2], 3)
            if r_ is not None:
                if world == 1:
                    timed("D.insert", sc.insert_features, r_)
                    timed("D.detect_enqueue", sc.detect_enqueue)
                    pipe.setdefault("d_queued", set()).add(j + 1)
                else:
                    xchg_q.put((j + 1, r_))
            if world == 1 and j in pipe.get("d_queued", ()):      # queued by the previous job: a whole period to finish
                pipe["loops"][j] = timed("D.detect_collect", sc.detect_collect)
                pipe["d_queued"].discard(j)
        return run

    def run_side(job):
        if side_thread:
            side_q.put(job)
            pipe["side_inflight"] = True
        else:
            job()

    def join_side():
        if pipe["side_inflight"]:
            err = timed("side.join", side_done.get)
            pipe["side_inflight"] = False
            if err is not None:
                raise err

    def collect_c():
        k = pipe["c_inflight"].pop(0)
        qm, tm, mst = timed("C.collect", mp.collect)
        last_pose["q"], last_pose["t"] = qm.tolist(), tm.tolist()
        return k, mst

    def step_pipelined(k, last):
        """Iteration k of the software pipeline: stage B of scan k+1 is queued before the pose of scan k's stage B is collected,
        that pose goes straight into scan k's stage C, which queues behind the stage-C steps still running; the oldest of those
        is collected when more than C_DEPTH are outstanding."""
        if not pipe["started"]:  # first scan of a run: what the previous iterations would have queued
            regs[k % len(regs)].run_device(d_scans[k].data_ptr(), npts[k], 3)
            od.enqueue_features(regs[k % len(regs)])
            pipe["b_queued"] = k
            pipe["started"] = True
            side_job(k - 1, last)()   # A(k+1), prefetch(k), D(k)
        join_side()                   # job k-1: A(k+1), prefetch(k), D(k) are queued
        run_side(side_job(k, last))
        if k + 1 < last and pipe["b_queued"] < k + 1:
            timed("B.enqueue", od.enqueue_features, regs[(k + 1) % len(regs)])
            pipe["b_queued"] = k + 1
        qlc, tlc, qw, tw, ost = timed("B.collect", od.collect)
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
    t0 = time.perf_counter()
    n_prof_steps = 0
    for k in range(W, W + K):
        on = a.prof_every > 0 and (k - W) % a.prof_every == 0
        if a.timeline:
            on = True if a.timeline_kernels else K // 2 <= k - W < K // 2 + 6
            S.prof_timeline(on)
        S.prof_enable(on, a.timeline_kernels or None)  # per-kernel timestamps on the sampled steps of the timed region
        n_prof_steps += on
        step(k, W + K)
    if pipelined:
        drain()  # the last scan's pose and map insertion belong to the timed region
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    S.prof_enable(False)
    prof = S.prof_read_all()
    if a.timeline:
        S.prof_timeline_dump(a.timeline)
    # sizes of one representative scan (outside the timed region) for the algorithmic-byte formulas
    reg.run_device(d_scans[W + K - 1].data_ptr(), npts[W + K - 1], 3)
    fz = reg.fetch()
    counts = dict(n_in=npts[W + K - 1], n_kept=fz["n_kept"], n_sharp=len(fz["sharp"]), n_less_sharp=len(fz["less_sharp"]), n_flat=len(fz["flat"]),
                  n_less_flat=fz["less_flat"].shape[0], stack_pts=stats["stack_pts"] / max(1, K), blocks=stats["blocks"] / max(1, 2 * K),
                  map_pts=stats["map_pts"] / max(1, K))
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
            "kernel_ms_per_step": {k: v[0] / max(1, n_prof_steps) for k, v in sorted(prof.items())}, "profiled_steps": n_prof_steps,
            "loops_detected": int(stats["loops"]), "input_gen_s": gen_s, "final_map_pose": last_pose,
            "host_us_per_step": {k: v / K * 1e6 for k, v in host_t.items()} if a.host_timing else None,
            "timed_window_unix": [t_wall0, t_wall0 + dt],
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def roofline_of(prof, K, c, pipelined=True):
    """Pick the kernel with the largest total time in the timed region (HIP events on its stream) and price it against
    the HBM roofline with ALGORITHMIC bytes per launch (SURVEY.md section 8d per-unit figures; DESIGN.md "Kernels")."""
    if not prof:
        return None
    M = c["stack_pts"]
    per_launch_bytes = {
        # stage A selection: ordered cloud xyzi + curvature read once, picked / lessFlat points written once
        "k_ring": 20.0 * c["n_kept"] + 16.0 * (c["n_sharp"] + c["n_less_sharp"] + c["n_flat"] + c["n_less_flat"]),
        # one radix pass moves every (key, value) pair once: 12 B read + 12 B written; mean pair count over the sorts of a scan
        "k_rs_scatter": 24.0 * (3 * c["n_kept"] + 3 * c["n_less_flat"] + 4 * (c["map_pts"] + M)) / 10.0,
        # one evaluation reads every residual block once (72 B edge / 56 B plane-norm parameters + 8 B kind/valid)
        "k_lm_solve": 80.0 * M * 5.0,  # up to five evaluations (initial point + <=4 candidates) per launch
        # association: each stack point (16 B) and its 5 neighbours (5 x 16 B); fit: 5 neighbours in, one block out
        "k_assoc_knn": 16.0 * M * 6.0,
        "k_assoc_fit": 16.0 * M * 5.0 + 64.0 * c["blocks"],
        # odometry NN: target clouds read once, queries once, one 8 B partial per (query, chunk)
        "k_odom_nn": 12.0 * (c["n_less_sharp"] + c["n_less_flat"]) + 16.0 * (c["n_sharp"] + c["n_flat"]),
        "k_odom_assoc": 16.0 * (c["n_sharp"] + c["n_flat"]) * (1.0 + 5.0 * c["n_less_flat"] / 51.0 / 16.0),
        "k_vox_small": 16.0 * c["n_less_sharp"] * 2.0,
    }
    # Dominant kernel = largest event-timed total among the kernels of the pose chains (streams A, B, C).  The kernels of the two
    # side streams (radix passes and the one-workgroup voxel filter of stage C's prefetch and of stage D) are left out of the
    # choice when the stages are pipelined: their queues are deep, and a dispatch's start..stop events then include its wait
    # for the command processor, which rocprofv3's kernel durations do not (about 2x for k_rs_scatter in the same
    # traced run, profiles/README.md) - by rocprofv3's own totals k_lm_solve leads either way.
    side = {"k_rs_scatter", "k_vox_small"} if pipelined else set()
    cands = {k: v for k, v in prof.items() if k not in side and k in per_launch_bytes} or prof
    name = max(cands, key=lambda k: cands[k][0])
    ms, cnt = prof[name]
    avg_s = ms / cnt * 1e-3
    ach = per_launch_bytes[name] / avg_s / 1e9
    # HBM traffic of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs of
    # this benchmark; bench.py cannot collect counters on itself).  FETCH_SIZE is reported raw, see the file's note.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_v3.json")))["kernels"].get(name)
        if pmc:
            traffic = (pmc["fetch_kb_per_dispatch_raw"] + pmc["write_kb_per_dispatch"]) * 1024.0
    except (OSError, KeyError, ValueError):
        pass
    return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": "profiles/r01_pmc_fetch_write_v3.json (bytes per launch, FETCH_SIZE raw + WRITE_SIZE)" if traffic else None,
            "avg_launch_us": avg_s * 1e6, "launches": cnt, "algorithmic_bytes_per_launch": per_launch_bytes[name],
            "share_of_step": ms / K, "all_kernels_ms_per_step": {k: v[0] / K for k, v in sorted(prof.items())}}


def cpu_baseline(scans, sc_db):
    """The oracle (dependency-free CPU restatement of the reference path, g++ -O3, one thread) on the same scans."""
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
    return {"value": n / dt, "unit": "scans/s", "cores": 1, "kind": "port",
            "sample": f"first {n} scans of the same sequence through the oracle (A+B+C+D serial on one core, kd-tree kNN)",
            "ms_per_scan": {"features": t_stage[0] / n * 1e3, "odometry": t_stage[1] / n * 1e3, "mapping": t_stage[2] / n * 1e3,
                            "scancontext": t_stage[3] / n * 1e3},
            "host": os.uname().nodename, "nproc": os.cpu_count()}


if __name__ == "__main__":
    main()
