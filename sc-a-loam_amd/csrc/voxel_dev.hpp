// pcl::VoxelGrid<PointXYZI> on the device (SoA clouds, device-resident counts).  See voxel.hip.
#pragma once
#include "common.hpp"
#include "radix_sort.hpp"

namespace scal {

struct VoxMeta {
    int error;                  // SCAL_E_CAPACITY when an axis needs more cells than the key layout holds
    int n_out;
    int guard;                  // PCL's "leaf size is too small" guard fired: output = input
    int used_bits;              // key bits in use (device-adaptive: the sort skips the passes above them)
    int b0, b1;                 // bit widths of the x and y voxel coordinates inside the key
};

struct SoA4 {
    float *x, *y, *z, *w;
};
struct CSoA4 {
    const float *x, *y, *z, *w;
};

// Optional epilogue of a filter run, executed by its last kernel (saves one-thread launches on dependent chains):
// *slots_out = min(*other_n + n_out, slots_cap) when slots_out is set; *err_out = error code when one was raised.
struct VoxTail {
    int* err_out = nullptr;
    int* slots_out = nullptr;
    const int* other_n = nullptr;
    int slots_cap = 0;
};

struct VoxelFilter {
    int cap = 0;
    RadixSort sorter;
    DevBuf<unsigned long long> keys;
    DevBuf<int> vals;
    DevBuf<int> blockcnt;
    DevBuf<VoxMeta> meta;
    DevBuf<unsigned> box_parts;  // [128][6] per-block bounding boxes (k_vox_bbox)
    std::string n_bbox = "k_vox_bbox", n_keys = "k_vox_keys", n_heads = "k_vox_heads", n_reduce = "k_vox_reduce", n_small = "k_vox_small",
                n_small_reduce = "k_vox_small_reduce";
    void set_tag(const char* tag) {  // stage tag of the timed launches, see RadixSort::set_tag
        n_bbox = std::string("k_vox_bbox") + tag, n_keys = std::string("k_vox_keys") + tag, n_heads = std::string("k_vox_heads") + tag;
        n_reduce = std::string("k_vox_reduce") + tag, n_small = std::string("k_vox_small") + tag;
        n_small_reduce = std::string("k_vox_small_reduce") + tag;
        sorter.set_tag(tag);
    }

    int init(int capacity);
    // out must hold `cap` points.  The key packs the three voxel coordinates as tightly as the bounding box allows
    // (computed on the device); max_bits bounds the passes the host enqueues (a multiple of 9 avoids waste).
    // n_bound: host-known upper bound of *d_n; clouds of <= 8192 points take a single-workgroup LDS path.
    // d_n_out receives the number of centroids; meta.p->error is set when the box needs more than max_bits.
    // ext_parts / n_parts: per-block boxes of the cloud written by an earlier kernel (vox_bbox_block_store): the reset and
    // bounding-box launches are skipped, the key kernel reduces the parts itself.
    int run(hipStream_t s, CSoA4 in, const int* d_n, int n_bound, float leaf, int max_bits, SoA4 out, int* d_n_out, const VoxTail* tail = nullptr,
            const unsigned* ext_parts = nullptr, int n_parts = 0);
};

}  // namespace scal

namespace scal {
#ifdef __HIPCC__
// Bounding box without a launch of its own and without contended atomics: a kernel that reads a cloud anyway (a gather, the
// curvature pass) lets every 256-thread block store the box of ITS points (order-preserving uint images: min xyz, max xyz) in
// parts[block][6]; the filter's key kernel reduces the parts (VoxelFilter::run with ext_parts).  Every thread of the block calls
// it, also in blocks without points (they store the identity); `have` = this thread holds a point.
__device__ inline void vox_bbox_block_store(unsigned* parts, int block, bool have, float x, float y, float z) {
    __shared__ unsigned s_box[4][6];
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    if (have) {
        const float v[3] = {x, y, z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const unsigned u = __float_as_uint(v[a]);
            lo[a] = hi[a] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], static_cast<unsigned>(__shfl_xor(static_cast<int>(lo[a]), o, 64)));
            hi[a] = max(hi[a], static_cast<unsigned>(__shfl_xor(static_cast<int>(hi[a]), o, 64)));
        }
        if ((threadIdx.x & 63) == 0) s_box[threadIdx.x >> 6][a] = lo[a], s_box[threadIdx.x >> 6][3 + a] = hi[a];
    }
    __syncthreads();
    if (threadIdx.x < 3)
        parts[block * 6 + threadIdx.x] = min(min(s_box[0][threadIdx.x], s_box[1][threadIdx.x]), min(s_box[2][threadIdx.x], s_box[3][threadIdx.x]));
    else if (threadIdx.x < 6)
        parts[block * 6 + threadIdx.x] = max(max(s_box[0][threadIdx.x], s_box[1][threadIdx.x]), max(s_box[2][threadIdx.x], s_box[3][threadIdx.x]));
}
#endif
void launch_deinterleave(hipStream_t s, const float* aos, int n, SoA4 o);
void launch_interleave(hipStream_t s, const int* d_n, int n_cap, CSoA4 in, float* aos);
}  // namespace scal
