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
    // out must hold `cap` points.  bits_per_axis in [8,21]: cells per axis the sort key can address.
    // d_n_out receives the number of centroids; meta.p->error is set on overflow.
    int run(hipStream_t s, CSoA4 in, const int* d_n, float leaf, int bits_per_axis, SoA4 out, int* d_n_out);
};

}  // namespace scal

namespace scal {
void launch_deinterleave(hipStream_t s, const float* aos, int n, SoA4 o);
void launch_interleave(hipStream_t s, const int* d_n, int n_cap, CSoA4 in, float* aos);
}  // namespace scal
