#!/usr/bin/env python3
"""Dense ScanContext distance matrix (SURVEY.md 8d "D dense", BASELINE.json config #4's 5,000 x 5,000 all-pairs mode): every
query against every database descriptor over all 60 column shifts, on the matrix cores (scal_sc_distance_matrix_device mode 2,
k_sc_gram).  Prints one JSON line with the MFMA roofline of k_sc_gram (algorithmic flops = 2*20*60*60 per pair), the vector-ALU
mode 1 (reference summation order) on a slice, the oracle's CPU rate on a sample, and two checks that hold at any size:
mode 2 == mode 1 on the slice, and the shift symmetry D[a][b] == D[b][a], shift[a][b] == (60 - shift[b][a]) % 60.
Not the headline benchmark (that is bench.py, config #2).
N > 1 (torch.distributed, one process per GPU): every rank holds the whole database (48 MB at 5,000) and computes its own row
block of the pair grid - no exchange on the data path (dense mode), strong scaling, time = max over ranks."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", os.path.join("sc-a-loam_amd", "python")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402

FLOP_PER_PAIR = 2.0 * 20 * 60 * 60
PEAK_F64_MFMA = 78.6  # TFLOP/s, AMD's MI355X FP64 matrix figure (dense; the guide's table has no f64 row)


def descriptors(n, seed):
    """Descriptors with the statistics SURVEY.md 8d quotes for real ones (occupancy ~0.5, cell values -2.8..18.3 m, a few empty
    sectors); every tenth from #500 on is a revisit: an earlier place rolled by a random yaw plus 5 cm of noise."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 20, 60))
    for i in range(n):
        if i >= 500 and i % 10 == 0:
            j = int(rng.integers(0, i - 100))
            out[i] = np.roll(out[j], int(rng.integers(0, 60)), axis=1) + rng.normal(0, 0.05, (20, 60)) * (out[j] != 0)
        else:
            d = rng.uniform(-2.8, 18.3, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5)
            d[:, rng.integers(0, 60, 3)] = 0.0
            out[i] = d
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--db-size", "--n", dest="n", type=int, default=5000, help="descriptors in the database (= queries)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slice", type=int, default=256, help="queries of the mode-1 comparison slice")
    ap.add_argument("--cpu-pairs", type=int, default=20000)
    ap.add_argument("--probe", default="", help="path of the built tools/mfma_f64_peak.hip probe (optional)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--backend", default="nccl")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    import scaloam as S
    n = a.n
    descs = descriptors(n, 401)
    sc = S.SCManager(max_keyframes=max(8192, n + 8), device=local)
    for d in descs:
        sc.saveScancontextAndKeys(d)
    r0, r1 = rank * n // world, (rank + 1) * n // world  # this rank's row block of the pair grid
    d_dist = torch.empty((r1 - r0) * n, dtype=torch.float64, device="cuda")
    d_shift = torch.empty((r1 - r0) * n, dtype=torch.int32, device="cuda")
    for _ in range(a.warmup):
        sc.distance_matrix_device(r0, r1, 0, n, 2, d_dist.data_ptr(), d_shift.data_ptr())
        sc.sync()
    S.prof_reset()
    S.prof_enable(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sc.distance_matrix_device(r0, r1, 0, n, 2, d_dist.data_ptr(), d_shift.data_ptr())
    sc.sync()
    dt = (time.perf_counter() - t0) / a.steps
    S.prof_enable(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # every rank checks its own block against the oracle on a few pairs; rank 0 reports
        import oracle_py as O
        Dr = d_dist.cpu().numpy().reshape(r1 - r0, n)
        Sr = d_shift.cpu().numpy().reshape(r1 - r0, n)
        rng = np.random.default_rng(5 + rank)
        bad = 0
        for x, y in zip(rng.integers(r0, r1, 200), rng.integers(0, n, 200)):
            full = O.sc_distance_full(descs[x], descs[y])
            full = np.where(np.isnan(full), 1e300, full)
            k = int(np.argmin(full))
            bad += int(abs(float(full[k]) - Dr[x - r0, y]) > 1e-12 or k != Sr[x - r0, y])
        tb = torch.tensor([bad], dtype=torch.int64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(tb)
        ms, cnt = S.prof_read_all()["k_sc_gram"]
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            ach = FLOP_PER_PAIR * (r1 - r0) * n / (ms / cnt * 1e-3) / 1e12
            print(json.dumps({
                "metric": "pairs/sec, dense ScanContext distance matrix over all 60 shifts", "value": n * n / dt, "unit": "pairs/s",
                "n_gpus": world, "scaling": "strong", "descriptors": n, "ms_per_matrix": dt * 1e3, "dtype": "f64",
                "roofline": {"bound": "mfma", "kernel": "k_sc_gram", "achieved": ach, "peak": PEAK_F64_MFMA, "unit": "TFLOP/s",
                             "frac": ach / PEAK_F64_MFMA, "note": "rank 0's row block"},
                "checks": {"oracle_mismatches_over_all_ranks": int(tb.item()), "pairs_checked": 200 * world}}))
        return
    prof = S.prof_read_all()
    ms, cnt = prof["k_sc_gram"]
    gram_s = ms / cnt * 1e-3
    prep_ms = prof["k_sc_gram_prep"][0] / max(1, prof["k_sc_gram_prep"][1])
    D2 = d_dist.cpu().numpy().reshape(n, n)
    S2 = d_shift.cpu().numpy().reshape(n, n)
    # shift symmetry (size-independent property): rolling b by s onto a is rolling a by -s onto b
    finite = D2 < 1e6
    sym_d = float(np.abs(D2 - D2.T)[finite & finite.T].max())
    sym_s = int((S2[finite] != ((60 - S2.T) % 60)[finite]).sum())
    diag_ok = bool(np.abs(np.diag(D2)).max() <= 1e-12 and np.all(np.diag(S2) == 0))
    # vector-ALU mode 1 on a slice of queries: timing and equality
    ns = min(a.slice, n)
    d1 = torch.empty(ns * n, dtype=torch.float64, device="cuda")
    s1 = torch.empty(ns * n, dtype=torch.int32, device="cuda")
    sc.distance_matrix_device(0, ns, 0, n, 1, d1.data_ptr(), s1.data_ptr())
    sc.sync()
    S.prof_reset()
    S.prof_enable(True)
    sc.distance_matrix_device(0, ns, 0, n, 1, d1.data_ptr(), s1.data_ptr())
    sc.sync()
    S.prof_enable(False)
    m1_ms, m1_cnt = S.prof_read_all()["k_sc_matrix"]
    D1 = d1.cpu().numpy().reshape(ns, n)
    S1 = s1.cpu().numpy().reshape(ns, n)
    # mode 3: the same product on the f32 matrix cores
    d3 = torch.empty(n * n, dtype=torch.float64, device="cuda")
    s3 = torch.empty(n * n, dtype=torch.int32, device="cuda")
    sc.distance_matrix_device(0, n, 0, n, 3, d3.data_ptr(), s3.data_ptr())
    sc.sync()
    S.prof_reset()
    S.prof_enable(True)
    for _ in range(a.steps):
        sc.distance_matrix_device(0, n, 0, n, 3, d3.data_ptr(), s3.data_ptr())
    sc.sync()
    S.prof_enable(False)
    f32_ms, f32_cnt = S.prof_read_all()["k_sc_gram_f32"]
    D3 = d3.cpu().numpy().reshape(n, n)
    S3 = s3.cpu().numpy().reshape(n, n)
    f32_diff = float(np.abs(D3 - D2).max())
    f32_shift = int((S3 != S2).sum())
    # where the shift differs, the two shifts' f64 distances are closer than the f32 error (checked through the oracle below)
    f32_shift_pairs = [(int(x), int(y)) for x, y in zip(*np.nonzero(S3 != S2))][:50]
    del d3, s3
    max_diff = float(np.abs(D1 - D2[:ns]).max())
    shift_mismatch = int((S1 != S2[:ns]).sum())
    # oracle (CPU restatement of distDirectSC over the 60 shifts) on a sample of pairs
    import oracle_py as O
    rng = np.random.default_rng(5)
    pa, pb = rng.integers(0, n, a.cpu_pairs), rng.integers(0, n, a.cpu_pairs)
    worst = 0.0
    bad_shift = 0
    t1 = time.perf_counter()
    for x, y in zip(pa, pb):
        full = O.sc_distance_full(descs[x], descs[y])
        full = np.where(np.isnan(full), 1e300, full)
        k = int(np.argmin(full))
        worst = max(worst, abs(float(full[k]) - D2[x, y]))
        bad_shift += int(k != S2[x, y])
    cpu_dt = time.perf_counter() - t1
    f32_gap = 0.0
    for x, y in f32_shift_pairs:
        full = O.sc_distance_full(descs[x], descs[y])
        f32_gap = max(f32_gap, abs(float(full[S3[x, y]]) - float(full[S2[x, y]])))
    # batch loop search: the dense block in row tiles + top-3 per row on the device, only the records come back
    K, EXCL = 3, 30
    sc.batch_loop_search(0, n, EXCL, K, 2)
    S.prof_reset()
    S.prof_enable(True)
    t1 = time.perf_counter()
    for _ in range(a.steps):
        bi, bd, bs = sc.batch_loop_search(0, n, EXCL, K, 2)
    bls_dt = (time.perf_counter() - t1) / a.steps
    S.prof_enable(False)
    pr = S.prof_read_all()
    topk_ms = pr["k_sc_row_topk"][0] / max(1, pr["k_sc_row_topk"][1])
    bls_bad = 0
    for q in range(n):
        lim = max(0, q - EXCL)
        row = D2[q, :lim]
        order = np.lexsort((np.arange(lim), row))[:K]
        want = np.full(K, -1)
        want[:len(order)] = order
        ok = np.array_equal(bi[q], want) and np.array_equal(bd[q][:len(order)], row[order]) and np.array_equal(bs[q][:len(order)], S2[q, order])
        bls_bad += int(not ok)
    n_pairs_bls = sum(max(0, min(n, t0 + 512) - 1 - EXCL) * (min(n, t0 + 512) - t0) for t0 in range(0, n, 512))
    probe = None
    if a.probe and os.path.exists(a.probe):
        try:
            probe = json.loads(subprocess.run([a.probe], capture_output=True, text=True, timeout=120).stdout.strip())
        except (OSError, ValueError, subprocess.TimeoutExpired):
            probe = None
    ach = FLOP_PER_PAIR * n * n / gram_s / 1e12
    print(json.dumps({
        "metric": "pairs/sec, dense ScanContext distance matrix over all 60 shifts", "value": n * n / dt, "unit": "pairs/s", "n_gpus": 1,
        "descriptors": n, "ms_per_matrix": dt * 1e3, "dtype": "f64",
        "roofline": {"bound": "mfma", "kernel": "k_sc_gram", "achieved": ach, "peak": PEAK_F64_MFMA, "unit": "TFLOP/s", "frac": ach / PEAK_F64_MFMA,
                     "avg_launch_ms": gram_s * 1e3, "algorithmic_flop_per_launch": FLOP_PER_PAIR * n * n, "issued_flop_per_launch":
                     2048.0 * 15 * 300 * 4 * ((n + 63) // 64) * ((n + 3) // 4), "measured_issue_peak": probe, "traffic": None},
        "prep_ms": prep_ms,
        "mode3_f32_mfma": {"kernel": "k_sc_gram_f32", "avg_launch_ms": f32_ms / f32_cnt, "pairs_per_s": n * n / (f32_ms / f32_cnt * 1e-3),
                           "achieved": FLOP_PER_PAIR * n * n / (f32_ms / f32_cnt * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                           "frac": FLOP_PER_PAIR * n * n / (f32_ms / f32_cnt * 1e-3) / 1e12 / 157.3, "max_abs_vs_mode2": f32_diff,
                           "shift_differs_from_mode2": f32_shift, "max_f64_gap_between_differing_shifts": f32_gap},
        "batch_loop_search": {"queries": n, "k": K, "exclude_recent": EXCL, "ms": bls_dt * 1e3, "queries_per_s": n / bls_dt,
                              "pairs_evaluated": n_pairs_bls, "k_sc_row_topk_avg_ms": topk_ms, "rows_differing_from_numpy_topk_of_the_matrix": bls_bad},
        "mode1_vector_alu": {"queries": ns, "ms": m1_ms / m1_cnt, "pairs_per_s": ns * n / (m1_ms / m1_cnt * 1e-3)},
        "cpu_baseline": {"value": a.cpu_pairs / cpu_dt, "unit": "pairs/s", "cores": 1, "kind": "port",
                         "sample": f"{a.cpu_pairs} random pairs through the oracle's 60-shift distDirectSC (incl. the ctypes call)"},
        "checks": {"mode2_vs_mode1_max_abs": max_diff, "mode2_vs_mode1_shift_mismatches": shift_mismatch, "vs_oracle_max_abs": worst,
                   "vs_oracle_shift_mismatches": bad_shift, "symmetry_max_abs": sym_d, "symmetry_shift_mismatches": sym_s, "self_match": diag_ok}}))


if __name__ == "__main__":
    main()
