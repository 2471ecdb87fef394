"""N > 1 path on CPU: world_size-2 `gloo` run of the sharded ScanContext search exchange.

On a GPU box each rank's shard lives in a scal_sc context; here the shard-local search is done by the oracle's pieces
(no GPU in this container) while the exchange (all_gather of per-shard top-3 records) and the merge
(scal_sc_merge_candidates, host code of the product library) are the real ones.  The merged answer must equal the
single-database detectLoopClosureID of the oracle at every step."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, every=1):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "sc-a-loam_amd", "python"))
    import torch
    import torch.distributed as dist
    import oracle_py as O
    import scaloam as S
    from scaloam.sharded import TreePeriodBook
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(77)  # every rank draws the same stream of descriptors
    descs = []
    for i in range(90):
        if i >= 40 and i % 5 == 0:
            j = int(rng.integers(0, i - 35))
            d = np.roll(descs[j], int(rng.integers(0, 60)), axis=1)
        else:
            d = rng.uniform(-2, 18, (20, 60)) * (rng.uniform(size=(20, 60)) < 0.5)
        descs.append(d)
    single = O.SCManager(dist_thres=0.3)
    local_keys, local_idx = [], []
    book = TreePeriodBook(0)  # the bookkeeping bench.py uses: a query's tree size depends on the ORDER of the queries only
    ok = True
    pending = []              # `every` queries travel in one all_gather (bench.py --sc-exchange-every)

    def exchange(batch):
        """one all_gather carries the shard-local records of all queries of the batch; every query is merged on its own"""
        good = True
        t = torch.from_numpy(np.stack([b[0] for b in batch]))
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        for j, (_, ref) in enumerate(batch):
            cands = []
            for g in gathered:
                for row in g[j].numpy():
                    if row[4] > 0:
                        cands.append(S.SCCand(np.float32(row[0]), int(row[1]), float(row[2]), int(row[3]), 0))
                    else:
                        cands.append(S.SCCand(3.4e38, -1, 1e7, 0, 0))
            got = S.merge_candidates(cands, 0.3)
            good &= got["loop_id"] == ref["loop_id"] and got["nn_idx"] == ref["nn_idx"] and abs(got["min_dist"] - ref["min_dist"]) < 1e-12
            good &= list(got["cand"]) == list(ref["cand"])
        return good

    for i, d in enumerate(descs):
        single.saveScancontextAndKeys(d)
        if i % world == rank:  # shard ownership: keyframe i on rank i % N
            local_keys.append(single.get(i)[1])
            local_idx.append(i)
        ref = single.detectLoopClosureID()
        if i + 1 < 31:
            continue
        book.n_global = i
        size_at_rebuild = book.step(1)[0]
        # shard-local top-3 by nanoflann's f32 key distance over eligible keys, then SC distance of those three
        qk = single.get(i)[1]
        rec = np.zeros((3, 6), np.float64)  # key_dist, idx, sc_dist, shift, valid, pad
        elig = [(k, g) for k, g in zip(local_keys, local_idx) if g < size_at_rebuild - 30]
        if elig:
            K = np.stack([k for k, _ in elig])
            diff = (qk[None, :] - K).astype(np.float32)
            sq = (diff * diff).astype(np.float32)
            acc = np.zeros(K.shape[0], np.float32)
            for g in range(0, 20, 4):
                acc = (acc + (((sq[:, g] + sq[:, g + 1]) + sq[:, g + 2]) + sq[:, g + 3])).astype(np.float32)
            gi = np.array([g for _, g in elig])
            order = np.lexsort((gi, acc))[:3]
            for s, o in enumerate(order):
                dd, sh = O.sc_distance(d, descs[gi[o]])
                rec[s] = [acc[o], gi[o], dd, sh, 1, 0]
        pending.append((rec, ref))
        if len(pending) == every:
            ok &= exchange(pending)
            pending = []
    if pending:  # the last, partial batch (every rank holds the same number of queries)
        ok &= exchange(pending)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world,every", [(2, 1), (2, 4), (3, 1), (3, 7)])
def test_sharded_sc_search_gloo(world, every):
    """every = 1: one exchange per query, as the reference's cadence allows at most; every > 1: the batched exchange of
    bench.py --sc-exchange-every (one pair of all-gathers per `every` scans).  The answers must not depend on it."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world + every
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, every)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)], res


def _voxel_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "sc-a-loam_amd", "python"))
    import torch
    import torch.distributed as dist
    import oracle_py as O
    from scaloam import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for case, (n, zspan) in enumerate([(60000, 12.0), (5000, 0.3), (0, 1.0)]):
        rng = np.random.default_rng(900 + case)  # the same whole map on every rank; rank r owns the r-th contiguous part
        pts = np.concatenate([rng.uniform(-40, 40, (n, 2)), rng.uniform(-zspan, zspan, (n, 1)), rng.uniform(0, 1, (n, 1))], axis=1).astype(np.float32)
        pts[: n // 3, :3] = np.round(pts[: n // 3, :3] * 2) / 2  # many points on voxel borders and in shared voxels
        cut = [0, n // 3 + 7 if n else 0, n] if world == 2 else [0, n // 4, n // 2, n]
        mine = pts[cut[rank]:cut[rank + 1]]

        def oracle_filter(t, leaf):  # the checker stands in for the HIP filter (no GPU in this container)
            a = t.numpy()
            return O.voxel_grid(a, leaf)[0] if a.shape[0] else np.zeros((0, 4), np.float32)

        part = sharded.sharded_downsample(torch.from_numpy(mine.copy()), 0.4, local_filter=oracle_filter)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([part.shape[0]], dtype=torch.int64))
        mx = max(1, max(int(s.item()) for s in sizes))
        pad = np.zeros((mx, 4), np.float32)
        pad[: part.shape[0]] = part
        parts = [torch.zeros((mx, 4)) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(pad))
        got = np.concatenate([p.numpy()[: int(s.item())] for p, s in zip(parts, sizes)])
        want = O.voxel_grid(pts, 0.4)[0] if n else np.zeros((0, 4), np.float32)
        ok &= got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # the exchange keeps the global order: a rank's received points are a subsequence of the whole map
        recv, (lo, hi) = sharded.exchange_by_layer(torch.from_numpy(mine.copy()), 0.4)
        r = recv.numpy()
        if n:
            k_all = np.floor(pts[:, 2] * (np.float32(1.0) / np.float32(0.4))).astype(np.int64)
            sel = pts[(k_all >= lo) & (k_all < hi)]
            ok &= sel.shape == r.shape and np.array_equal(sel.view(np.uint32), r.view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_voxel_downsample_gloo(world):
    """SURVEY 8e, offline map merge: the slab exchange of scaloam.sharded (all-to-all over gloo here, RCCL on a GPU node) with
    the oracle's VoxelGrid as the local filter equals the oracle's VoxelGrid over the whole map bit for bit - balanced slabs,
    a map thinner than one voxel layer per rank (empty slabs), an empty map."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_voxel_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)], res


def test_balanced_cuts_properties():
    """Slab boundaries of the sharded VoxelGrid: non-decreasing, inside [0, layers], every layer owned by exactly one rank, and
    no rank's share above the ideal by more than the heaviest single layer (a layer is never split)."""
    sys.path.insert(0, os.path.join(ROOT, "sc-a-loam_amd", "python"))
    from scaloam.sharded import balanced_cuts
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        for hist in (rng.integers(0, 1000, 75), np.array([0, 0, 5, 0, 0]), np.array([7]), np.zeros(4, np.int64), rng.integers(0, 3, 200)):
            cuts = balanced_cuts(hist, world)
            assert len(cuts) == world - 1 and all(0 <= c <= len(hist) for c in cuts) and cuts == sorted(cuts)
            edges = [0] + cuts + [len(hist)]
            shares = [int(hist[edges[r]:edges[r + 1]].sum()) for r in range(world)]
            assert sum(shares) == int(hist.sum())
            if hist.sum():
                assert max(shares) <= hist.sum() / world + hist.max()
