// Offline dense map merge on gfx950 (BASELINE.json config #5): keyframe cloud x pose -> global frame,
// near-range removal, concatenation.  Replaces the loop body of /root/reference/utils/python/makeMergedMap.py:83-133
// (open3d transform :105, range filter :109-116, concatenation :129-133, f32 save :145-147); in-process analogues in the
// reference: local2global (laserPosegraphOptimization.cpp:338-359), transformPointCloud (:446-470).
//
// The one genuinely HBM-bound piece of the system: 16 B read and (almost always) 16 B written per point, no reuse.  Two
// passes because the output must keep the reference's order (frames in order, points in order, dropped points closed up):
//   k_mm_count  keep flag per point (local range > threshold, f64 as numpy computes it), kept points per workgroup
//   k_mm_scan   exclusive prefix of the workgroup counts on top of the running total (one workgroup)
//   k_mm_write  flags again (cheaper than storing them: the point is needed anyway), rigid transform in f64 in the order of
//               a column-major 4x4 product, ballot-ranked stable write of xyzi records
// A workgroup handles 1024 consecutive points of ONE frame; a batch of frames is one launch of each kernel.
#include "common.hpp"
#include "device_utils.hpp"
#include "voxel_dev.hpp"
#include <vector>
#include <cstring>

namespace scal {

struct MMBlock {
    int first;  // first point of the workgroup in the concatenated input
    int end;    // end of its frame
    int frame;
};

__device__ __forceinline__ bool mm_keep(const float4& p, double thres) {
    const double x = p.x, y = p.y, z = p.z;
    return sqrt((x * x + y * y) + z * z) > thres;  // LA.norm(local, axis=1) > thres_near_removal (makeMergedMap.py:109, :112)
}

constexpr int MM_ITEMS = 4;                 // points per thread
constexpr int MM_TILE = 256 * MM_ITEMS;    // points per workgroup; point (j, t) of a workgroup = first + j*256 + t

__global__ void __launch_bounds__(256) k_mm_count(const float4* __restrict__ in, const MMBlock* __restrict__ blocks, double thres,
                                                  int* __restrict__ blkcnt) {
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const MMBlock b = blocks[blockIdx.x];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < MM_ITEMS; ++j) {
        const int i = b.first + j * 256 + threadIdx.x;
        const bool keep = i < b.end && mm_keep(in[i], thres);
        mine += __popcll(__ballot(keep));
    }
    if (lane_id() == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) blkcnt[blockIdx.x] = s_cnt;
}

// in-place exclusive scan of cnt[0..nb) by one workgroup (16-byte loads, every thread owns a contiguous multiple of four
// entries; cnt is padded to a multiple of 4096), offset by nothing: the running total is added by the write pass.
__global__ void __launch_bounds__(1024) k_mm_scan(int* __restrict__ cnt, int nb, long long* __restrict__ d_total, long long cap, int* __restrict__ d_error) {
    __shared__ int smem[17];
    const int per4 = ((nb + 1023) / 1024 + 3) / 4;  // int4 groups per thread
    int4* c4 = reinterpret_cast<int4*>(cnt) + static_cast<size_t>(threadIdx.x) * per4;
    const int first = threadIdx.x * per4 * 4;
    int sum = 0;
    for (int g = 0; g < per4; ++g) {
        const int4 v = c4[g];
        const int e = first + 4 * g;
        sum += (e < nb ? v.x : 0) + (e + 1 < nb ? v.y : 0) + (e + 2 < nb ? v.z : 0) + (e + 3 < nb ? v.w : 0);
    }
    int total;
    int run = block_exclusive_scan(sum, smem, &total);
    for (int g = 0; g < per4; ++g) {
        const int4 v = c4[g];
        const int e = first + 4 * g;
        int4 o;
        o.x = run, run += e < nb ? v.x : 0;
        o.y = run, run += e + 1 < nb ? v.y : 0;
        o.z = run, run += e + 2 < nb ? v.z : 0;
        o.w = run, run += e + 3 < nb ? v.w : 0;
        c4[g] = o;
    }
    if (threadIdx.x == 0) {
        const long long base = *d_total;
        if (base + total > cap) *d_error = SCAL_E_CAPACITY;
        *d_total = base + total;
    }
}

__global__ void __launch_bounds__(256) k_mm_write(const float4* __restrict__ in, const MMBlock* __restrict__ blocks, const double* __restrict__ poses,
                                                  double thres, const int* __restrict__ blkoff, long long base, long long cap,
                                                  float4* __restrict__ out) {
    __shared__ int s_cnt[MM_ITEMS][4];
    const MMBlock b = blocks[blockIdx.x];
    const int w = wave_id();
    float4 p[MM_ITEMS];
    bool keep[MM_ITEMS];
    int rank[MM_ITEMS];
#pragma unroll
    for (int j = 0; j < MM_ITEMS; ++j) {
        const int i = b.first + j * 256 + threadIdx.x;
        keep[j] = false;
        if (i < b.end) {
            p[j] = in[i];
            keep[j] = mm_keep(p[j], thres);
        }
        const uint64_t m = __ballot(keep[j]);
        rank[j] = __popcll(m & lanemask_lt());
        if (lane_id() == 0) s_cnt[j][w] = __popcll(m);
    }
    __syncthreads();
    const double* T = poses + 12 * b.frame;
    int before = 0;  // kept points of the workgroup in front of (j, wave w)
#pragma unroll
    for (int j = 0; j < MM_ITEMS; ++j) {
        int mine = before;
        for (int q = 0; q < w; ++q) mine += s_cnt[j][q];
        before += s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
        if (!keep[j]) continue;
        const long long o = base + blkoff[blockIdx.x] + mine + rank[j];
        if (o >= cap) continue;  // k_mm_scan has flagged the overflow
        const double x = p[j].x, y = p[j].y, z = p[j].z;
        float4 g;  // T * [x y z 1]^T, column-major product order; the 4th row is (0 0 0 1), so the division by w is by 1
        g.x = static_cast<float>(((T[0] * x + T[1] * y) + T[2] * z) + T[3]);
        g.y = static_cast<float>(((T[4] * x + T[5] * y) + T[6] * z) + T[7]);
        g.z = static_cast<float>(((T[8] * x + T[9] * y) + T[10] * z) + T[11]);
        g.w = p[j].w;
        out[o] = g;
    }
}

}  // namespace scal

using namespace scal;

struct scal_mapmerge {
    scal_mapmerge_config cfg;
    hipStream_t stream = nullptr;
    long long cap = 0;
    DevBuf<float4> out, stage;
    PinBuf<float4> h_stage;     // pinned staging of host frames, block descriptors and poses (stream-ordered uploads)
    PinBuf<MMBlock> h_blocks;
    PinBuf<double> h_poses;
    DevBuf<MMBlock> blocks;
    DevBuf<int> blkcnt;
    DevBuf<double> poses;
    DevBuf<long long> d_total;
    DevBuf<int> d_error;
    int nb_cap = 0, pose_cap = 0;
    long long n_host = 0;  // upper bound of the running total known to the host (sum of the frame sizes added)
    // downsampling of the merged map (lazily allocated)
    VoxelFilter vf;
    DevBuf<float> sx, sy, sz, sw, ox, oy, oz, ow;
    DevBuf<int> d_n;
    bool vf_ready = false;
};

extern "C" int scal_mapmerge_create(const scal_mapmerge_config* cfg, scal_mapmerge_t** out) {
    if (!cfg || !out || cfg->max_points <= 0 || cfg->max_frame_points <= 0) {
        set_error("scal_mapmerge_create: bad argument");
        return SCAL_E_ARG;
    }
    *out = nullptr;
    SCAL_TRY(select_device(cfg->device));
    auto* c = new scal_mapmerge();
    c->cfg = *cfg;
    c->cap = cfg->max_points;
    int rc = SCAL_OK;
    auto A = [&](int r) { if (rc == SCAL_OK) rc = r; };
    A(c->out.alloc(static_cast<size_t>(c->cap)));
    A(c->stage.alloc(static_cast<size_t>(cfg->max_frame_points)));
    A(c->h_stage.alloc(static_cast<size_t>(cfg->max_frame_points)));
    A(c->d_total.alloc(1));
    A(c->d_error.alloc(1));
    if (rc == SCAL_OK && acquire_stream(cfg->device, &c->stream) != SCAL_OK) rc = SCAL_E_HIP;
    if (rc == SCAL_OK && (hipMemsetAsync(c->d_total.p, 0, sizeof(long long), c->stream) != hipSuccess ||
                          hipMemsetAsync(c->d_error.p, 0, sizeof(int), c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess))
        rc = SCAL_E_HIP;
    if (rc != SCAL_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return SCAL_OK;
}

extern "C" void scal_mapmerge_destroy(scal_mapmerge_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        release_stream(c->cfg.device);
    }
    delete c;
}

extern "C" int scal_mapmerge_reset(scal_mapmerge_t* c) {
    if (!c) return SCAL_E_ARG;
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_HIP(hipMemsetAsync(c->d_total.p, 0, sizeof(long long), c->stream));
    SCAL_HIP(hipMemsetAsync(c->d_error.p, 0, sizeof(int), c->stream));
    c->n_host = 0;
    return SCAL_OK;
}

// frames [0, n_frames) of a concatenated device-resident xyzi array; frame f = points [offsets[f], offsets[f+1])
static int merge_batch(scal_mapmerge* c, const float4* d_in, const int* offsets, const double* poses12, int n_frames, double near_thres) {
    hipStream_t s = c->stream;
    std::vector<MMBlock> hb;
    for (int f = 0; f < n_frames; ++f)
        for (int i = offsets[f]; i < offsets[f + 1]; i += MM_TILE) hb.push_back(MMBlock{i, offsets[f + 1], f});
    const int nb = static_cast<int>(hb.size());
    if (nb == 0) return SCAL_OK;
    if (nb > c->nb_cap) {
        SCAL_HIP(hipStreamSynchronize(s));
        SCAL_TRY(c->blocks.alloc(nb));
        SCAL_TRY(c->h_blocks.alloc(nb));
        SCAL_TRY(c->blkcnt.alloc(static_cast<size_t>(nb) + 8192));  // k_mm_scan reads whole 16-byte groups per thread
        c->nb_cap = nb;
    }
    if (n_frames > c->pose_cap) {
        SCAL_HIP(hipStreamSynchronize(s));
        SCAL_TRY(c->poses.alloc(static_cast<size_t>(12) * n_frames));
        SCAL_TRY(c->h_poses.alloc(static_cast<size_t>(12) * n_frames));
        c->pose_cap = n_frames;
    }
    SCAL_HIP(hipStreamSynchronize(s));  // the pinned staging buffers may still feed the previous batch
    std::memcpy(c->h_blocks.p, hb.data(), sizeof(MMBlock) * nb);
    std::memcpy(c->h_poses.p, poses12, sizeof(double) * 12 * n_frames);
    SCAL_HIP(hipMemcpyAsync(c->blocks.p, c->h_blocks.p, sizeof(MMBlock) * nb, hipMemcpyHostToDevice, s));
    SCAL_HIP(hipMemcpyAsync(c->poses.p, c->h_poses.p, sizeof(double) * 12 * n_frames, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_mm_count, dim3(nb), dim3(256), 0, s, d_in, c->blocks.p, near_thres, c->blkcnt.p);
    long long base = 0;
    // the running total is read by the write pass as a kernel argument: fetch it (tiny, and add() is not the measured path)
    SCAL_HIP(hipMemcpyAsync(&base, c->d_total.p, sizeof(long long), hipMemcpyDeviceToHost, s));
    SCAL_HIP(hipStreamSynchronize(s));
    hipLaunchKernelGGL(k_mm_scan, dim3(1), dim3(1024), 0, s, c->blkcnt.p, nb, c->d_total.p, c->cap, c->d_error.p);
    SCAL_LAUNCH_PROF("k_mm_write", k_mm_write, dim3(nb), dim3(256), 0, s, d_in, c->blocks.p, c->poses.p, near_thres, c->blkcnt.p, base, c->cap,
                     c->out.p);
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

extern "C" int scal_mapmerge_add(scal_mapmerge_t* c, const float* xyzi, int n, const double* pose12, double near_thres) {
    if (!c || n < 0 || (n > 0 && !xyzi) || !pose12) {
        set_error("scal_mapmerge_add: bad argument");
        return SCAL_E_ARG;
    }
    if (n > c->cfg.max_frame_points) {
        set_error("scal_mapmerge_add: frame of %d points exceeds max_frame_points %d", n, c->cfg.max_frame_points);
        return SCAL_E_TOO_MANY;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    if (n == 0) return SCAL_OK;
    SCAL_HIP(hipStreamSynchronize(c->stream));  // the previous frame's write pass reads the staging buffers
    std::memcpy(c->h_stage.p, xyzi, sizeof(float) * 4 * n);
    SCAL_HIP(hipMemcpyAsync(c->stage.p, c->h_stage.p, sizeof(float) * 4 * n, hipMemcpyHostToDevice, c->stream));
    const int offsets[2] = {0, n};
    return merge_batch(c, c->stage.p, offsets, pose12, 1, near_thres);
}

extern "C" int scal_mapmerge_add_batch_device(scal_mapmerge_t* c, const float* d_xyzi, const int* offsets, const double* poses12, int n_frames,
                                              double near_thres) {
    if (!c || !d_xyzi || !offsets || !poses12 || n_frames < 0) {
        set_error("scal_mapmerge_add_batch_device: bad argument");
        return SCAL_E_ARG;
    }
    for (int f = 0; f < n_frames; ++f)
        if (offsets[f + 1] < offsets[f]) {
            set_error("scal_mapmerge_add_batch_device: offsets must not decrease");
            return SCAL_E_ARG;
        }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    return merge_batch(c, reinterpret_cast<const float4*>(d_xyzi), offsets, poses12, n_frames, near_thres);
}

extern "C" long long scal_mapmerge_size(scal_mapmerge_t* c) {
    if (!c) return SCAL_E_ARG;
    if (hipSetDevice(c->cfg.device) != hipSuccess) return SCAL_E_HIP;
    long long n = 0;
    int err = 0;
    if (hipMemcpyAsync(&n, c->d_total.p, sizeof n, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(&err, c->d_error.p, sizeof err, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return SCAL_E_HIP;
    if (err) {
        set_error("scal_mapmerge: the merged map exceeds max_points (%lld)", c->cap);
        return err;
    }
    return n;
}

extern "C" int scal_mapmerge_download(scal_mapmerge_t* c, float* out_xyzi, long long cap_points) {
    if (!c || !out_xyzi || cap_points < 0) {
        set_error("scal_mapmerge_download: bad argument");
        return SCAL_E_ARG;
    }
    const long long n = scal_mapmerge_size(c);
    if (n < 0) return static_cast<int>(n);
    const long long m = n < cap_points ? n : cap_points;
    if (m > 0) {
        SCAL_HIP(hipMemcpyAsync(out_xyzi, c->out.p, sizeof(float) * 4 * static_cast<size_t>(m), hipMemcpyDeviceToHost, c->stream));
        SCAL_HIP(hipStreamSynchronize(c->stream));
    }
    return SCAL_OK;
}

extern "C" const float* scal_mapmerge_device_points(scal_mapmerge_t* c) { return c ? reinterpret_cast<const float*>(c->out.p) : nullptr; }

// pubMap's VoxelGrid over the merged map (laserPosegraphOptimization.cpp:810-834, leaf = mapviz_filter_size): PCL order, ordered
// f32 centroids, PCL's overflow guard (output = input when the voxel count overflows int32).  Result to host memory.
extern "C" int scal_mapmerge_downsample(scal_mapmerge_t* c, float leaf, float* out_xyzi, long long cap_points, long long* n_out) {
    if (!c || !(leaf > 0) || !n_out || cap_points < 0 || (cap_points > 0 && !out_xyzi)) {
        set_error("scal_mapmerge_downsample: bad argument");
        return SCAL_E_ARG;
    }
    const long long n = scal_mapmerge_size(c);
    if (n < 0) return static_cast<int>(n);
    if (n > 2000000000ll) {
        set_error("scal_mapmerge_downsample: %lld points exceed the 32-bit indices of the voxel filter", n);
        return SCAL_E_TOO_MANY;
    }
    *n_out = 0;
    if (n == 0) return SCAL_OK;
    hipStream_t s = c->stream;
    if (!c->vf_ready) {
        const int cap = static_cast<int>(std::min<long long>(c->cap, 2000000000ll));
        SCAL_TRY(c->vf.init(cap));
        SCAL_TRY(c->sx.alloc(cap)); SCAL_TRY(c->sy.alloc(cap)); SCAL_TRY(c->sz.alloc(cap)); SCAL_TRY(c->sw.alloc(cap));
        SCAL_TRY(c->ox.alloc(cap)); SCAL_TRY(c->oy.alloc(cap)); SCAL_TRY(c->oz.alloc(cap)); SCAL_TRY(c->ow.alloc(cap));
        SCAL_TRY(c->d_n.alloc(2));
        c->vf_ready = true;
    }
    const int ni = static_cast<int>(n);
    launch_deinterleave(s, reinterpret_cast<const float*>(c->out.p), ni, SoA4{c->sx.p, c->sy.p, c->sz.p, c->sw.p});
    SCAL_HIP(hipMemcpyAsync(c->d_n.p, &ni, sizeof(int), hipMemcpyHostToDevice, s));
    SCAL_HIP(hipStreamSynchronize(s));
    SCAL_TRY(c->vf.run(s, CSoA4{c->sx.p, c->sy.p, c->sz.p, c->sw.p}, c->d_n.p, ni, leaf, 45, SoA4{c->ox.p, c->oy.p, c->oz.p, c->ow.p}, c->d_n.p + 1));
    int m = 0;
    VoxMeta hm;
    SCAL_HIP(hipMemcpyAsync(&m, c->d_n.p + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    SCAL_HIP(hipMemcpyAsync(&hm, c->vf.meta.p, sizeof hm, hipMemcpyDeviceToHost, s));
    SCAL_HIP(hipStreamSynchronize(s));
    if (hm.error) {
        set_error("scal_mapmerge_downsample: the map's bounding box needs more voxel-key bits than the sort is given (leaf %g)", leaf);
        return hm.error;
    }
    *n_out = m;
    const long long w = std::min<long long>(m, cap_points);
    if (w > 0) {  // the four centroid components come back separately and are interleaved on the host (offline path)
        std::vector<float> comp(static_cast<size_t>(w));
        const float* src[4] = {c->ox.p, c->oy.p, c->oz.p, c->ow.p};
        for (int k = 0; k < 4; ++k) {
            SCAL_HIP(hipMemcpyAsync(comp.data(), src[k], sizeof(float) * static_cast<size_t>(w), hipMemcpyDeviceToHost, s));
            SCAL_HIP(hipStreamSynchronize(s));
            for (long long i = 0; i < w; ++i) out_xyzi[4 * i + k] = comp[static_cast<size_t>(i)];
        }
    }
    return SCAL_OK;
}
