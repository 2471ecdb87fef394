// pcl::VoxelGrid<pcl::PointXYZI>::filter on gfx950 (default settings: all fields averaged, no minimum count).
// Call sites it replaces: scanRegistration.cpp:414-418, laserMapping.cpp:543-551 / :793-801,
// laserPosegraphOptimization.cpp:629-631.  PCL semantics kept (PCL 1.8 voxel_grid.hpp, see oracle/voxel.cpp):
//   bounding box -> min_b = floor(min * inv_leaf); overflow guard on (max-min)*inv_leaf+1 products;
//   voxel of a point = floor(p * inv_leaf) - min_b per axis; output one centroid per occupied voxel in ascending
//   idx = i + j*dx + k*dx*dy order, which equals lexicographic (k, j, i) order, so the 64-bit sort key is
//   (k << 2b | j << b | i) and div_b is never needed on the device; centroid = f32 sum in (voxel, arrival) order
//   divided by float(count).
// Kernels: k_vox_bbox (atomic min/max on ordered uints) -> k_vox_keys -> radix sort -> k_vox_heads (run heads per
// block) -> k_scan -> k_vox_reduce (one thread per run: sequential f32 sum, the order PCL's accumulator sees).
#include "voxel_dev.hpp"
#include "device_utils.hpp"

namespace scal {

// bounding box as per-block parts (order-preserving uint images: min xyz, max xyz); k_vox_keys reduces them.  No atomics: 768
// atomic min/max on six words of one cache line cost more than the rest of this kernel.
__device__ __forceinline__ void k_vox_bbox_body(const CSoA4& in, const int* __restrict__ d_n, unsigned* __restrict__ parts) {
    __shared__ unsigned s_lo[3][4], s_hi[3][4];
    const int n = *d_n;
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned a = float_to_ordered(in.x[i]), b = float_to_ordered(in.y[i]), c = float_to_ordered(in.z[i]);
        lo[0] = min(lo[0], a), hi[0] = max(hi[0], a);
        lo[1] = min(lo[1], b), hi[1] = max(hi[1], b);
        lo[2] = min(lo[2], c), hi[2] = max(hi[2], c);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], static_cast<unsigned>(__shfl_xor(static_cast<int>(lo[a]), o, 64)));
            hi[a] = max(hi[a], static_cast<unsigned>(__shfl_xor(static_cast<int>(hi[a]), o, 64)));
        }
        if (lane_id() == 0) s_lo[a][wave_id()] = lo[a], s_hi[a][wave_id()] = hi[a];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        parts[blockIdx.x * 6 + a] = min(min(s_lo[a][0], s_lo[a][1]), min(s_lo[a][2], s_lo[a][3]));
        parts[blockIdx.x * 6 + 3 + a] = max(max(s_hi[a][0], s_hi[a][1]), max(s_hi[a][2], s_hi[a][3]));
    }
}
SCAL_KERNEL(256, k_vox_bbox)

__device__ __forceinline__ int bits_for(int cells) {  // smallest b with (1 << b) >= cells, at least 1
    int b = 1;
    while ((1 << b) < cells) ++b;
    return b;
}

__device__ __forceinline__ void k_vox_keys_body(const CSoA4& in, const int* __restrict__ d_n, float inv, int max_bits, VoxMeta* m,
                                                  unsigned long long* __restrict__ keys, int* __restrict__ vals, const unsigned* __restrict__ ext_parts, int n_parts) {
    const int n = *d_n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (static_cast<int>(blockIdx.x * blockDim.x) >= n && blockIdx.x != 0) return;  // uniform over the block
    __shared__ unsigned s_box[6];
    {   // the box arrives as per-block parts: every block reduces them (L2-resident, a few KB)
        __shared__ unsigned s_w[4][6];
        unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
        for (int b = threadIdx.x; b < n_parts; b += 256)
#pragma unroll
            for (int a = 0; a < 3; ++a) lo[a] = min(lo[a], ext_parts[b * 6 + a]), hi[a] = max(hi[a], ext_parts[b * 6 + 3 + a]);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo[a] = min(lo[a], static_cast<unsigned>(__shfl_xor(static_cast<int>(lo[a]), o, 64)));
                hi[a] = max(hi[a], static_cast<unsigned>(__shfl_xor(static_cast<int>(hi[a]), o, 64)));
            }
            if (lane_id() == 0) s_w[wave_id()][a] = lo[a], s_w[wave_id()][3 + a] = hi[a];
        }
        __syncthreads();
        if (threadIdx.x < 3) s_box[threadIdx.x] = min(min(s_w[0][threadIdx.x], s_w[1][threadIdx.x]), min(s_w[2][threadIdx.x], s_w[3][threadIdx.x]));
        else if (threadIdx.x < 6) s_box[threadIdx.x] = max(max(s_w[0][threadIdx.x], s_w[1][threadIdx.x]), max(s_w[2][threadIdx.x], s_w[3][threadIdx.x]));
    }
    __syncthreads();
    if (i >= n && i != 0) return;
    float mn[3], mx[3];
    int mb[3], cells[3];
    long long d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = ordered_to_float(s_box[a]);
        mx[a] = ordered_to_float(s_box[3 + a]);
        d[a] = static_cast<long long>((mx[a] - mn[a]) * inv) + 1;
        mb[a] = static_cast<int>(floorf(mn[a] * inv));
        cells[a] = static_cast<int>(floorf(mx[a] * inv)) - mb[a] + 1;
    }
    const bool guard = n > 0 && d[0] * d[1] * d[2] > 2147483647ll;  // PCL: "leaf size is too small", output = input
    int b0, b1, used;
    if (guard) {
        b0 = b1 = 0;
        used = bits_for(max(n, 2));
    } else {
        b0 = bits_for(max(cells[0], 1)), b1 = bits_for(max(cells[1], 1));
        used = b0 + b1 + bits_for(max(cells[2], 1));
    }
    if (i == 0) {
        m->guard = guard ? 1 : 0;
        m->b0 = b0, m->b1 = b1;
        m->used_bits = n > 0 ? min(used, max_bits) : 0;
        m->error = (n > 0 && used > max_bits) ? SCAL_E_CAPACITY : 0;
    }
    if (i >= n) return;
    unsigned long long k;
    if (guard) {
        k = static_cast<unsigned long long>(i);  // every point is its own voxel, arrival order
    } else {
        const unsigned long long i0 = static_cast<unsigned long long>(static_cast<int>(floorf(in.x[i] * inv)) - mb[0]);
        const unsigned long long i1 = static_cast<unsigned long long>(static_cast<int>(floorf(in.y[i] * inv)) - mb[1]);
        const unsigned long long i2 = static_cast<unsigned long long>(static_cast<int>(floorf(in.z[i] * inv)) - mb[2]);
        k = (i2 << (b0 + b1)) | (i1 << b0) | i0;  // idx = i0 + i1*dx + i2*dx*dy orders like (i2, i1, i0)
    }
    keys[i] = k;
    vals[i] = i;
}
SCAL_KERNEL(256, k_vox_keys)

__device__ __forceinline__ void k_vox_heads_body(const SortedPairs& sp, const int* __restrict__ d_n, int* __restrict__ blockcnt) {
    const int n = *d_n;
    const int nb = (n + 255) / 256;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    __shared__ int s[17];
    const unsigned long long* keys = sp.keys[sorted_sel(sp)];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int head = (i < n) && (i == 0 || keys[i] != keys[i - 1]);
    int total;
    block_exclusive_scan(head, s, &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}
SCAL_KERNEL(256, k_vox_heads)

// One thread per voxel (run head) sums its members in sorted = arrival order, as CentroidPoint<PointXYZI> does in f32.  The
// additions of a run are inherently serial, the loads need not be: the block stages its 256 sorted positions plus a
// 256-position look-ahead (keys + gathered points) in LDS with independent loads, the per-run loops then read LDS; only a
// run longer than the look-ahead finishes from global memory.
// The block's output offset is the sum of the earlier blocks' head counts (k_vox_heads), summed here instead of in a scan
// launch of its own; the last block publishes the total, runs the optional epilogue and leaves the bounding box reset.
__device__ __forceinline__ void k_vox_reduce_body(const SortedPairs& sp, const int* __restrict__ d_n, const int* __restrict__ blockcnt, const CSoA4& in, const SoA4& out,
                                                    int* __restrict__ d_n_out, VoxMeta* m, const VoxTail& tail) {
    const int n = *d_n;
    const int nb = (n + 255) / 256;
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        *d_n_out = 0;
        if (tail.slots_out) *tail.slots_out = min(*tail.other_n, tail.slots_cap);
        if (tail.err_out && m->error) *tail.err_out = m->error;
    }
    if (static_cast<int>(blockIdx.x) >= nb) return;
    constexpr int SPAN = 512;
    __shared__ int s[17];
    __shared__ unsigned long long skey[SPAN];
    __shared__ float sx[SPAN], sy[SPAN], sz[SPAN], sw[SPAN];
    const int sel = sorted_sel(sp);
    const unsigned long long* keys = sp.keys[sel];
    const int* vals = sp.vals[sel];
    const int base = blockIdx.x * 256;
    int before = 0;
    for (int b = threadIdx.x; b < static_cast<int>(blockIdx.x); b += 256) before += blockcnt[b];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = h * 256 + threadIdx.x;
        if (base + t < n) {
            const int g = vals[base + t];
            skey[t] = keys[base + t];
            sx[t] = in.x[g], sy[t] = in.y[g], sz[t] = in.z[g], sw[t] = in.w[g];
        }
    }
    int block_off;
    block_exclusive_scan(before, s, &block_off);
    const int i = base + threadIdx.x;
    const int head = (i < n) && (i == 0 || keys[i] != keys[i - 1]);
    int total;
    const int rank = block_exclusive_scan(head, s, &total);  // contains the barrier that publishes the staged span
    if (static_cast<int>(blockIdx.x) == nb - 1 && threadIdx.x == 0) {
        const int n_out = block_off + total;
        *d_n_out = n_out;
        if (tail.slots_out) *tail.slots_out = min(*tail.other_n + n_out, tail.slots_cap);
        if (tail.err_out && m->error) *tail.err_out = m->error;
    }
    if (!head) return;
    const unsigned long long k = skey[threadIdx.x];
    const int lim = min(n - base, SPAN);
    float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
    int t = threadIdx.x;
    while (t < lim && skey[t] == k) {
        ax += sx[t], ay += sy[t], az += sz[t], aw += sw[t];
        ++t;
    }
    int u = base + t;
    if (t == SPAN) {
        while (u < n && keys[u] == k) {
            const int g = vals[u];
            ax += in.x[g], ay += in.y[g], az += in.z[g], aw += in.w[g];
            ++u;
        }
    }
    const float c = static_cast<float>(u - i);
    const int o = block_off + rank;
    out.x[o] = ax / c, out.y[o] = ay / c, out.z[o] = az / c, out.w[o] = aw / c;
}
SCAL_KERNEL(256, k_vox_reduce)

// Filter for a cloud of <= VOX_SMALL points in TWO launches (the general path takes ~12): the corner clouds of stage C.
//   k_vox_small_sort    one workgroup per 512-chunk of the cloud (<= 16).  Every workgroup computes the bounding box and ALL keys
//                       (14-bit voxel coordinates + arrival index) into LDS itself - 96 KB of L2-resident input, cheaper than a
//                       launch that would hand them over - sorts the 512-chunks in registers (one wave each), then ranks the keys of
//                       ITS chunk against the other chunks by binary search (the keys are distinct: they carry the arrival index)
//                       and writes them to their sorted position.  No merge network across workgroups.
//   k_vox_small_reduce  one workgroup per 256 sorted positions: run heads, ordered f32 centroids (the order PCL's accumulator sees),
//                       the output offset of a workgroup = run heads in front of it, counted from the sorted keys directly.
// Round 2 did all of this in ONE workgroup (58 us: an LDS bitonic sort of 8192 keys by 1024 threads is 30 us of it).
constexpr int VOX_SMALL = 8192;
__device__ __forceinline__ void k_vox_small_sort_body(const CSoA4& in, const int* __restrict__ d_n, int n_cap, float inv, unsigned long long* __restrict__ sorted,
                                                         VoxMeta* m, const VoxTail& tail) {
    extern __shared__ __align__(16) unsigned long long skeys[];
    __shared__ unsigned s_lo[3][16], s_hi[3][16];
    __shared__ int s_mb[4];
    const int n = min(*d_n, n_cap);
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int tile = blockIdx.x;
    const int nc = (n + 511) >> 9;
    if (tile == 0 && tid == 0) {
        m->error = (*d_n > VOX_SMALL) ? SCAL_E_CAPACITY : 0, m->guard = 0;
        if (m->error && tail.err_out) *tail.err_out = m->error;
    }
    if (tile >= nc) return;  // uniform over the workgroup (n == 0: everybody leaves, the reduce kernel publishes the empty result)
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    for (int i = tid; i < n; i += 1024) {
        const unsigned a = float_to_ordered(in.x[i]), b = float_to_ordered(in.y[i]), c = float_to_ordered(in.z[i]);
        lo[0] = min(lo[0], a), hi[0] = max(hi[0], a);
        lo[1] = min(lo[1], b), hi[1] = max(hi[1], b);
        lo[2] = min(lo[2], c), hi[2] = max(hi[2], c);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], static_cast<unsigned>(__shfl_xor(static_cast<int>(lo[a]), o, 64)));
            hi[a] = max(hi[a], static_cast<unsigned>(__shfl_xor(static_cast<int>(hi[a]), o, 64)));
        }
        if (lane == 0) s_lo[a][wv] = lo[a], s_hi[a][wv] = hi[a];
    }
    __syncthreads();
    if (tid == 0) {
        long long d[3];
        bool wide = false;
        for (int a = 0; a < 3; ++a) {
            unsigned l = s_lo[a][0], h = s_hi[a][0];
            for (int q = 1; q < 16; ++q) l = min(l, s_lo[a][q]), h = max(h, s_hi[a][q]);
            const float mn = ordered_to_float(l), mx = ordered_to_float(h);
            d[a] = static_cast<long long>((mx - mn) * inv) + 1;
            s_mb[a] = static_cast<int>(floorf(mn * inv));
            if (static_cast<int>(floorf(mx * inv)) - s_mb[a] + 1 > 16383) wide = true;
        }
        const bool guard = d[0] * d[1] * d[2] > 2147483647ll;  // PCL: "leaf size is too small", output = input
        s_mb[3] = guard ? 1 : 0;
        if (tile == 0) {
            m->guard = s_mb[3];
            if (!guard && wide) {
                m->error = SCAL_E_CAPACITY;
                if (tail.err_out) *tail.err_out = SCAL_E_CAPACITY;
            }
        }
    }
    __syncthreads();
    const bool guard = s_mb[3] != 0;
    for (int i = tid; i < nc * 512; i += 1024) {
        unsigned long long k = ~0ull;
        if (i < n) {
            if (guard) {
                k = static_cast<unsigned long long>(i) << 14;
            } else {
                const unsigned long long i0 = static_cast<unsigned long long>(static_cast<int>(floorf(in.x[i] * inv)) - s_mb[0]) & 0x3fffu;
                const unsigned long long i1 = static_cast<unsigned long long>(static_cast<int>(floorf(in.y[i] * inv)) - s_mb[1]) & 0x3fffu;
                const unsigned long long i2 = static_cast<unsigned long long>(static_cast<int>(floorf(in.z[i] * inv)) - s_mb[2]) & 0x3fffu;
                k = (i2 << 42) | (i1 << 28) | (i0 << 14);
            }
            k |= static_cast<unsigned long long>(i);  // arrival index (< 16384) keeps the order inside a voxel
        }
        skeys[i] = k;
    }
    __syncthreads();
    if (wv < nc) {
        unsigned long long v[8];
        chunk_load(skeys, wv, v);
        wave_sort512(v);
        chunk_store(skeys, wv, v);
    }
    __syncthreads();
    // two threads per key of this workgroup's chunk: each searches half of the other chunks
    const int j = tid >> 1, half = tid & 1;
    const unsigned long long key = skeys[tile * 512 + j];
    int rank = half == 0 ? j : 0;  // position inside the own chunk
    for (int c = half; c < nc; c += 2) {
        if (c == tile) continue;
        const unsigned long long* ch = skeys + c * 512;
        int l = 0, h = 512;  // lower bound: 513 possible answers, at most 10 probes
        while (l < h) {
            const int mid = (l + h) >> 1;
            if (ch[mid] < key) l = mid + 1;
            else h = mid;
        }
        rank += l;
    }
    rank += __shfl_xor(rank, 1, 64);
    if (half == 0 && key != ~0ull) sorted[rank] = key;
}
SCAL_KERNEL(1024, k_vox_small_sort)

__device__ __forceinline__ void k_vox_small_reduce_body(const unsigned long long* __restrict__ sorted, const int* __restrict__ d_n, int n_cap, const CSoA4& in,
                                                          const SoA4& out, int* __restrict__ d_n_out, const VoxTail& tail) {
    const int n = min(*d_n, n_cap);
    const int nb = (n + 255) / 256;
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        *d_n_out = 0;
        if (tail.slots_out) *tail.slots_out = min(*tail.other_n, tail.slots_cap);
    }
    if (static_cast<int>(blockIdx.x) >= nb) return;
    constexpr int SPAN = 512;
    __shared__ int s[17];
    __shared__ unsigned long long skey[SPAN];
    __shared__ float sx[SPAN], sy[SPAN], sz[SPAN], sw[SPAN];
    const int base = blockIdx.x * 256;
    // run heads in front of this workgroup's positions: counted from the keys themselves (<= 7936 of them, L2-resident)
    int before = 0;
    for (int i = threadIdx.x; i < base; i += 256) before += (i == 0) || ((sorted[i] >> 14) != (sorted[i - 1] >> 14));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = h * 256 + threadIdx.x;
        if (base + t < n) {
            const unsigned long long k = sorted[base + t];
            const int g = static_cast<int>(k & 0x3fffu);
            skey[t] = k >> 14;
            sx[t] = in.x[g], sy[t] = in.y[g], sz[t] = in.z[g], sw[t] = in.w[g];
        }
    }
    int block_off;
    {
        // block sum of `before`: the exclusive scan's total
        block_exclusive_scan(before, s, &block_off);
    }
    const int i = base + threadIdx.x;
    const int head = (i < n) && (i == 0 || (sorted[i] >> 14) != (sorted[i - 1] >> 14));
    int total;
    const int rank = block_exclusive_scan(head, s, &total);  // contains the barrier that publishes the staged span
    if (static_cast<int>(blockIdx.x) == nb - 1 && threadIdx.x == 0) {
        const int n_out = block_off + total;
        *d_n_out = n_out;
        if (tail.slots_out) *tail.slots_out = min(*tail.other_n + n_out, tail.slots_cap);
    }
    if (!head) return;
    const unsigned long long k = skey[threadIdx.x];
    const int lim = min(n - base, SPAN);
    float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
    int t = threadIdx.x;
    while (t < lim && skey[t] == k) {
        ax += sx[t], ay += sy[t], az += sz[t], aw += sw[t];
        ++t;
    }
    int u = base + t;
    if (t == SPAN) {
        while (u < n && (sorted[u] >> 14) == k) {
            const int g = static_cast<int>(sorted[u] & 0x3fffu);
            ax += in.x[g], ay += in.y[g], az += in.z[g], aw += in.w[g];
            ++u;
        }
    }
    const float c = static_cast<float>(u - i);
    const int o = block_off + rank;
    out.x[o] = ax / c, out.y[o] = ay / c, out.z[o] = az / c, out.w[o] = aw / c;
}
SCAL_KERNEL(256, k_vox_small_reduce)
SCAL_DEFINE_STAMP_READER(scal_debug_stamps_voxel)

int VoxelFilter::init(int capacity) {
    cap = capacity;
    SCAL_TRY(sorter.init(cap));
    SCAL_TRY(keys.alloc(cap));
    SCAL_TRY(vals.alloc(cap));
    SCAL_TRY(blockcnt.alloc(div_up(cap, 256) + 1));
    SCAL_TRY(meta.alloc(1));
    SCAL_TRY(box_parts.alloc(6 * 128));
    // every caller has selected its device by now: the attribute is per device, not per process
    SCAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_vox_small_sort), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 static_cast<int>(sizeof(unsigned long long) * VOX_SMALL)));
    return SCAL_OK;
}

int VoxelFilter::run(hipStream_t s, CSoA4 in, const int* d_n, int n_bound, float leaf, int max_bits, SoA4 out, int* d_n_out, const VoxTail* tail,
                     const unsigned* ext_parts, int n_parts) {
    const float inv = 1.0f / leaf;  // inverse_leaf_size_ = 1 / leaf_size_ in f32
    const VoxTail tl = tail ? *tail : VoxTail();
    if (n_bound <= VOX_SMALL) {
        const int lds = sizeof(unsigned long long) * VOX_SMALL;  // attribute set per device in VoxelFilter::init
        const int n_cap = min(VOX_SMALL, cap);
        const int chunks = max(1, div_up(min(n_bound, n_cap), 512));  // workgroups beyond the cloud's chunks leave at once
        SCAL_LAUNCH(n_small.c_str(), k_vox_small_sort, dim3(chunks), dim3(1024), lds, s, in, d_n, n_cap, inv, keys.p, meta.p, tl);
        SCAL_LAUNCH(n_small_reduce.c_str(), k_vox_small_reduce, dim3(max(1, div_up(min(n_bound, n_cap), 256))), dim3(256), 0, s, keys.p, d_n, n_cap, in, out,
                         d_n_out, tl);
        SCAL_HIP(hipGetLastError());
        return SCAL_OK;
    }
    const int nb = max(1, div_up(min(cap, n_bound), 256));
    if (!ext_parts) {
        n_parts = min(nb, 128);
        ext_parts = box_parts.p;
        SCAL_LAUNCH(n_bbox.c_str(), k_vox_bbox, dim3(n_parts), dim3(256), 0, s, in, d_n, box_parts.p);
    }
    rec_mark(true);  // keys .. reduce: the segment that filter runs of different contexts can share launches for (batch.hpp)
    SCAL_LAUNCH(n_keys.c_str(), k_vox_keys, dim3(nb), dim3(256), 0, s, in, d_n, inv, max_bits, meta.p, keys.p, vals.p, ext_parts, n_parts);
    SortedPairs sp;
    SCAL_TRY(sorter.sort(s, keys.p, vals.p, d_n, n_bound, max_bits, &meta.p->used_bits, &sp));
    SCAL_LAUNCH(n_heads.c_str(), k_vox_heads, dim3(nb), dim3(256), 0, s, sp, d_n, blockcnt.p);
    SCAL_LAUNCH(n_reduce.c_str(), k_vox_reduce, dim3(nb), dim3(256), 0, s, sp, d_n, blockcnt.p, in, out, d_n_out, meta.p, tl);
    rec_mark(false);
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

__device__ __forceinline__ void k_deinterleave_body(const float* __restrict__ aos, int n, const SoA4& o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float4 p = reinterpret_cast<const float4*>(aos)[i];
        o.x[i] = p.x, o.y[i] = p.y, o.z[i] = p.z, o.w[i] = p.w;
    }
}
SCAL_KERNEL(1024, k_deinterleave)
__device__ __forceinline__ void k_interleave4_body(const int* __restrict__ d_n, const CSoA4& in, float* __restrict__ aos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *d_n) reinterpret_cast<float4*>(aos)[i] = make_float4(in.x[i], in.y[i], in.z[i], in.w[i]);
}
SCAL_KERNEL(1024, k_interleave4)

void launch_deinterleave(hipStream_t s, const float* aos, int n, SoA4 o) {
    if (n > 0) SCAL_LAUNCH("k_deinterleave", k_deinterleave, dim3(div_up(n, 256)), dim3(256), 0, s, aos, n, o);
}
void launch_interleave(hipStream_t s, const int* d_n, int n_cap, CSoA4 in, float* aos) {
    if (n_cap > 0) SCAL_LAUNCH("k_interleave4", k_interleave4, dim3(div_up(n_cap, 256)), dim3(256), 0, s, d_n, in, aos);
}

}  // namespace scal

using namespace scal;

struct scal_voxel {
    int device = 0, cap = 0;
    hipStream_t stream = nullptr;
    VoxelFilter vf;
    DevBuf<float> aos, ix, iy, iz, iw, ox, oy, oz, ow;
    DevBuf<int> d_n;  // [0] in, [1] out
    HostStage hs;     // pinned staging of the host-array entry point (first call)
    std::mutex mu;
};

extern "C" int scal_voxel_create(int max_points, int device, scal_voxel_t** out) {
    if (!out || max_points <= 0) {
        set_error("scal_voxel_create: bad argument");
        return SCAL_E_ARG;
    }
    *out = nullptr;
    SCAL_TRY(select_device(device));
    auto* c = new scal_voxel();
    c->device = device, c->cap = max_points;
    int rc = c->vf.init(max_points);
    auto A = [&](int r) { if (rc == SCAL_OK) rc = r; };
    A(c->aos.alloc((size_t)max_points * 4));
    A(c->ix.alloc(max_points)); A(c->iy.alloc(max_points)); A(c->iz.alloc(max_points)); A(c->iw.alloc(max_points));
    A(c->ox.alloc(max_points)); A(c->oy.alloc(max_points)); A(c->oz.alloc(max_points)); A(c->ow.alloc(max_points));
    A(c->d_n.alloc(2));
    if (rc == SCAL_OK && acquire_stream(c->device, &c->stream) != SCAL_OK) {
        set_error("hipStreamCreate failed");
        rc = SCAL_E_HIP;
    }
    if (rc != SCAL_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return SCAL_OK;
}

extern "C" void scal_voxel_destroy(scal_voxel_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) {
        (void)op_stream_synchronize(c->stream);
        release_stream(c->device);
    }
    delete c;
}

// host clouds (on_device = false) or 16-byte xyzi records already in HBM; the output goes where the input came from
static int voxel_downsample(scal_voxel_t* c, const float* xyzi, int n, float leaf, float* out_xyzi, int* n_out, bool on_device) {
    if (!c || !n_out || n < 0 || (n > 0 && (!xyzi || !out_xyzi)) || !(leaf > 0.f)) {
        set_error("scal_voxel_downsample: bad argument");
        return SCAL_E_ARG;
    }
    if (n > c->cap) {
        set_error("cloud has %d points, capacity is %d", n, c->cap);
        return SCAL_E_TOO_MANY;
    }
    *n_out = 0;
    if (n == 0) return SCAL_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const float* d_in = xyzi;
    if (!on_device) {
        (void)c->hs.reserve((size_t)c->cap * 16 + 4096);
        SCAL_HIP(c->hs.h2d(c->aos.p, xyzi, sizeof(float) * 4 * n, s));
        d_in = c->aos.p;
    }
    SCAL_HIP(op_memcpy_async(c->d_n.p, &n, sizeof(int), hipMemcpyHostToDevice, s));
    SoA4 in{c->ix.p, c->iy.p, c->iz.p, c->iw.p}, o{c->ox.p, c->oy.p, c->oz.p, c->ow.p};
    launch_deinterleave(s, d_in, n, in);
    SCAL_TRY(c->vf.run(s, CSoA4{in.x, in.y, in.z, in.w}, c->d_n.p, n, leaf, 45, o, c->d_n.p + 1));
    launch_interleave(s, c->d_n.p + 1, n, CSoA4{o.x, o.y, o.z, o.w}, on_device ? out_xyzi : c->aos.p);
    VoxMeta hm;
    SCAL_HIP(op_memcpy_async(&hm, c->vf.meta.p, sizeof hm, hipMemcpyDeviceToHost, s));
    int m = 0;
    SCAL_HIP(op_memcpy_async(&m, c->d_n.p + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    c->hs.finish();  // the upload's staging is free again
    if (hm.error) {
        set_error("voxel grid bounding box needs more than 45 key bits (or 16383 cells per axis on the small-cloud path)");
        return SCAL_E_CAPACITY;
    }
    if (!on_device) {
        SCAL_HIP(c->hs.d2h(out_xyzi, c->aos.p, sizeof(float) * 4 * m, s));
        SCAL_HIP(op_stream_synchronize(s));
        c->hs.finish();
    }
    *n_out = m;
    return SCAL_OK;
}

extern "C" int scal_voxel_downsample(scal_voxel_t* c, const float* xyzi, int n, float leaf, float* out_xyzi, int* n_out) {
    return voxel_downsample(c, xyzi, n, leaf, out_xyzi, n_out, false);
}

extern "C" int scal_voxel_downsample_device(scal_voxel_t* c, const float* d_xyzi, int n, float leaf, float* d_out_xyzi, int* n_out) {
    return voxel_downsample(c, d_xyzi, n, leaf, d_out_xyzi, n_out, true);
}

extern "C" void* scal_voxel_stream(scal_voxel_t* c) { return c ? static_cast<void*>(c->stream) : nullptr; }
