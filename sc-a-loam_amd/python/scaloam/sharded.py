"""VoxelGrid over a map whose points are spread over several GPUs (offline map merge: "by keyframe, then one
exchange"; the filter is pubMap's, laserPosegraphOptimization.cpp:810-834, and makeMergedMap.py's voxel_down_sample).

Each rank holds the merged points of a contiguous, ascending range of keyframes (scal_mapmerge_* output).  A voxel's centroid is
an f32 sum in arrival order, so a voxel must be reduced by ONE rank that sees its points in the global order.  The space is cut
into slabs along z at voxel-layer boundaries (balanced by an all-reduced histogram of the layers), every point goes to the owner
of its layer with one all-to-all that keeps (source rank, local order) - the global order - and the owner runs the ordinary
filter on what it received.  VoxelGrid emits voxels in ascending (k, j, i) order whatever the bounding box is, so the ranks'
outputs concatenated in rank order are bit-identical to filtering the whole map on one GPU.

The exchange is torch.distributed (backend "nccl" = RCCL over xGMI on a GPU node; "gloo" in the CPU tests); the local filter
is the HIP voxel filter through the C-ABI.  There is no CPU fallback: without a `local_filter` the tensors must be on a GPU."""
import numpy as np

_filters = {}


def layer_of(points, leaf):
    """Voxel layer index along z exactly as the filter computes it: floor(z * (1.0f / leaf)) in f32 (voxel.hip, k_vox_keys;
    PCL voxel_grid.hpp: ijk2 = floor(p.z * inverse_leaf_size_[2]))."""
    import torch
    inv = float(np.float32(1.0) / np.float32(leaf))
    return torch.floor(points[:, 2] * inv).to(torch.int64)


def balanced_cuts(hist, world):
    """Split layers [0, len(hist)) into `world` consecutive ranges of roughly equal point count: returns the first layer of
    ranges 1..world-1 (non-decreasing)."""
    total = int(hist.sum())
    cum = np.cumsum(hist)
    cuts = []
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(np.searchsorted(cum, target, side="left")) + 1 if total else 0)
    return [min(c, len(hist)) for c in cuts]


def exchange_by_layer(points, leaf, group=None):
    """points: [n, 4] f32 tensor of this rank (global order = rank order, then local order).  Returns the points of this
    rank's slab, still in global order, and the slab's layer range (lo, hi) in absolute layer indices."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = points.device
    # gloo has no all-to-all on device tensors: when N > 1 is rehearsed with gloo ranks sharing a GPU the collectives go through
    # host copies; under nccl (= RCCL) the device tensors are exchanged directly
    via_host = points.is_cuda and dist.get_backend(group) == "gloo"
    cdev = torch.device("cpu") if via_host else dev
    k = layer_of(points, leaf)
    big = torch.iinfo(torch.int64).max
    lo = torch.tensor([int(k.min()) if k.numel() else big], dtype=torch.int64, device=cdev)
    hi = torch.tensor([int(k.max()) if k.numel() else -big], dtype=torch.int64, device=cdev)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    k0, k1 = int(lo.item()), int(hi.item())
    if k1 < k0:  # no points anywhere
        return points[:0], (0, 0)
    layers = k1 - k0 + 1
    hist = (torch.bincount(k - k0, minlength=layers) if k.numel() else torch.zeros(layers, dtype=torch.int64, device=dev)).to(cdev)
    dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    cuts = balanced_cuts(hist.cpu().numpy(), world)  # identical on every rank
    bounds = torch.tensor(cuts, dtype=torch.int64, device=dev)
    dest = torch.bucketize(k - k0, bounds, right=True) if world > 1 else torch.zeros_like(k)  # layer >= cuts[r-1] -> rank r
    order = torch.sort(dest, stable=True).indices  # groups by destination, keeps the local order inside each group
    send = points[order].contiguous().to(cdev)
    send_counts = torch.bincount(dest, minlength=world).to(cdev)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = [int(v) for v in send_counts.tolist()], [int(v) for v in recv_counts.tolist()]
    recv = torch.empty((sum(rc), 4), dtype=points.dtype, device=cdev)
    dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=group)
    edges = [0] + cuts + [layers]
    return recv.to(dev), (k0 + edges[rank], k0 + edges[rank + 1])


def gpu_voxel_filter(device):
    """The product's local filter: scal_voxel_downsample_device on the received device tensor."""
    import torch
    from . import VoxelGrid
    state = {}

    def run(recv, leaf):
        if not recv.is_cuda:
            raise RuntimeError("sharded_downsample: points are not on a GPU and no local_filter was given (there is no CPU path)")
        n = int(recv.shape[0])
        if n == 0:
            return np.zeros((0, 4), np.float32)
        vg = state.get("vg")
        if vg is None or state["cap"] < n:
            if vg is not None:
                vg.close()
            vg = state["vg"] = VoxelGrid(max_points=n + 1024, device=device)
            state["cap"] = n + 1024
        out = torch.empty_like(recv)
        # The filter runs on the library's own stream: order it behind whatever produced `recv` (under nccl the all-to-all has
        # only been ENQUEUED on torch's stream when it returns) and torch's stream behind the filter, on the device.
        ext = torch.cuda.ExternalStream(vg.stream_ptr(), device=recv.device)
        ext.wait_stream(torch.cuda.current_stream(recv.device))
        m = vg.filter_device(recv.data_ptr(), n, leaf, out.data_ptr())   # waits for its own stream before returning
        torch.cuda.current_stream(recv.device).wait_stream(ext)
        return out[:m].cpu().numpy()

    return run


def sharded_downsample(points, leaf, group=None, local_filter=None):
    """VoxelGrid(leaf) over the union of every rank's `points` ([n, 4] f32 tensor; ranks hold ascending keyframe ranges).
    Returns this rank's part of the result ([m, 4] f32 numpy): the parts concatenated in rank order equal the single-GPU filter
    of the concatenated points bit for bit.  `local_filter(tensor, leaf) -> numpy` defaults to the HIP filter."""
    recv, _ = exchange_by_layer(points, leaf, group)
    if local_filter is None:
        dev = points.device.index or 0
        if dev not in _filters:
            _filters[dev] = gpu_voxel_filter(dev)  # keeps its map-merge context (and its buffers) between calls
        local_filter = _filters[dev]
    return local_filter(recv, leaf)


class TreePeriodBook:
    """detectLoopClosureID's kd-tree bookkeeping (Scancontext.cpp:353-365: the ring-key tree is rebuilt when `tree_making_period_conter
    % 30 == 0`, then the counter is incremented, so the tree is up to 29 queries stale) for a database that every scan step extends by
    `world` descriptors - rank 0's, rank 1's, ... in that global order - and queries once per descriptor.  limits[q] = database size
    the tree of query q was built from; a shard then considers keyframes with index < limits[q] - 30 (scal_sc_shard_query_batch_device).
    The bookkeeping depends on the ORDER of the queries only, so an exchange that carries Q scan steps at once gives every query the
    limit it would have had with one exchange per scan."""

    def __init__(self, n_global=0, period=30):
        self.n_global, self.period = int(n_global), int(period)
        self.counter, self.size_at_rebuild = 0, 0

    def step(self, world=1):
        self.n_global += world  # the step's descriptors are all in the database before its first query is answered
        limits = []
        for _ in range(world):
            if self.counter % self.period == 0:
                self.size_at_rebuild = self.n_global
            self.counter += 1
            limits.append(self.size_at_rebuild)
        return limits

    def batch(self, q_steps, world=1):
        return [v for _ in range(q_steps) for v in self.step(world)]
