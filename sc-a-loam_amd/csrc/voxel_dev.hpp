// pcl::VoxelGrid<PointXYZI> on the device (SoA clouds, device-resident counts).  See voxel.hip.
#pragma once
#include "common.hpp"
#include "radix_sort.hpp"

namespace scal {

struct VoxMeta {
    unsigned umin[3], umax[3];  // order-preserving uint images of the bounding box
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

struct VoxelFilter {
    int cap = 0;
    RadixSort sorter;
    DevBuf<unsigned long long> keys;
    DevBuf<int> vals;
    DevBuf<int> blockcnt;
    DevBuf<VoxMeta> meta;

    int init(int capacity);
    // out must hold `cap` points.  The key packs the three voxel coordinates as tightly as the bounding box allows
    // (computed on the device); max_bits bounds the passes the host enqueues (a multiple of 9 avoids waste).
    // n_bound: host-known upper bound of *d_n; clouds of <= 8192 points take a single-workgroup LDS path.
    // d_n_out receives the number of centroids; meta.p->error is set when the box needs more than max_bits.
    int run(hipStream_t s, CSoA4 in, const int* d_n, int n_bound, float leaf, int max_bits, SoA4 out, int* d_n_out);
};

}  // namespace scal

namespace scal {
void launch_deinterleave(hipStream_t s, const float* aos, int n, SoA4 o);
void launch_interleave(hipStream_t s, const int* d_n, int n_cap, CSoA4 in, float* aos);
}  // namespace scal
