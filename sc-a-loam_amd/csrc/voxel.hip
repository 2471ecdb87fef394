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

__global__ void k_vox_reset(VoxMeta* m) {
    if (threadIdx.x < 3) {
        m->umin[threadIdx.x] = 0xffffffffu;
        m->umax[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) m->error = 0, m->n_out = 0, m->guard = 0;
}

__global__ void __launch_bounds__(256) k_vox_bbox(CSoA4 in, const int* __restrict__ d_n, VoxMeta* m) {
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
    if (threadIdx.x < 3) {  // one atomic pair per block and axis
        const int a = threadIdx.x;
        const unsigned l = min(min(s_lo[a][0], s_lo[a][1]), min(s_lo[a][2], s_lo[a][3]));
        const unsigned h = max(max(s_hi[a][0], s_hi[a][1]), max(s_hi[a][2], s_hi[a][3]));
        if (l != 0xffffffffu) atomicMin(&m->umin[a], l);
        if (h != 0u) atomicMax(&m->umax[a], h);
    }
}

__global__ void __launch_bounds__(256) k_vox_keys(CSoA4 in, const int* __restrict__ d_n, float inv, int bits, VoxMeta* m,
                                                  unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int n = *d_n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float mn[3], mx[3];
    int mb[3];
    long long d[3];
    bool wide = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = ordered_to_float(m->umin[a]);
        mx[a] = ordered_to_float(m->umax[a]);
        d[a] = static_cast<long long>((mx[a] - mn[a]) * inv) + 1;
        mb[a] = static_cast<int>(floorf(mn[a] * inv));
        const int xb = static_cast<int>(floorf(mx[a] * inv));
        if (xb - mb[a] + 1 > (1 << bits)) wide = true;
    }
    const bool guard = d[0] * d[1] * d[2] > 2147483647ll;
    if (i == 0) {
        m->guard = guard ? 1 : 0;
        if (!guard && wide) m->error = SCAL_E_CAPACITY;
    }
    unsigned long long k;
    if (guard) {
        k = static_cast<unsigned long long>(i);  // every point is its own voxel, arrival order
    } else {
        const unsigned long long mask = (1ull << bits) - 1ull;
        const unsigned long long i0 = static_cast<unsigned long long>(static_cast<int>(floorf(in.x[i] * inv)) - mb[0]) & mask;
        const unsigned long long i1 = static_cast<unsigned long long>(static_cast<int>(floorf(in.y[i] * inv)) - mb[1]) & mask;
        const unsigned long long i2 = static_cast<unsigned long long>(static_cast<int>(floorf(in.z[i] * inv)) - mb[2]) & mask;
        k = (i2 << (2 * bits)) | (i1 << bits) | i0;
    }
    keys[i] = k;
    vals[i] = i;
}

__global__ void __launch_bounds__(256) k_vox_heads(const unsigned long long* __restrict__ keys, const int* __restrict__ d_n,
                                                   int* __restrict__ blockcnt) {
    const int n = *d_n;
    const int nb = (n + 255) / 256;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    __shared__ int s[17];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int head = (i < n) && (i == 0 || keys[i] != keys[i - 1]);
    int total;
    block_exclusive_scan(head, s, &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_vox_reduce(const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                                    const int* __restrict__ d_n, const int* __restrict__ blockoff, CSoA4 in, SoA4 out) {
    const int n = *d_n;
    const int nb = (n + 255) / 256;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    __shared__ int s[17];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int head = (i < n) && (i == 0 || keys[i] != keys[i - 1]);
    int total;
    const int rank = block_exclusive_scan(head, s, &total);
    if (!head) return;
    const unsigned long long k = keys[i];
    float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
    int u = i;
    while (u < n && keys[u] == k) {  // CentroidPoint<PointXYZI>: f32 sums in sorted (arrival) order
        const int g = vals[u];
        ax += in.x[g], ay += in.y[g], az += in.z[g], aw += in.w[g];
        ++u;
    }
    const float c = static_cast<float>(u - i);
    const int o = blockoff[blockIdx.x] + rank;
    out.x[o] = ax / c, out.y[o] = ay / c, out.z[o] = az / c, out.w[o] = aw / c;
}

int VoxelFilter::init(int capacity) {
    cap = capacity;
    SCAL_TRY(sorter.init(cap));
    SCAL_TRY(keys.alloc(cap));
    SCAL_TRY(vals.alloc(cap));
    SCAL_TRY(blockcnt.alloc(div_up(cap, 256) + 1));
    SCAL_TRY(meta.alloc(1));
    return SCAL_OK;
}

int VoxelFilter::run(hipStream_t s, CSoA4 in, const int* d_n, float leaf, int bits, SoA4 out, int* d_n_out) {
    const float inv = 1.0f / leaf;  // inverse_leaf_size_ = 1 / leaf_size_ in f32
    const int nb = max(1, div_up(cap, 256));
    hipLaunchKernelGGL(k_vox_reset, dim3(1), dim3(64), 0, s, meta.p);
    hipLaunchKernelGGL(k_vox_bbox, dim3(min(nb, 128)), dim3(256), 0, s, in, d_n, meta.p);
    hipLaunchKernelGGL(k_vox_keys, dim3(nb), dim3(256), 0, s, in, d_n, inv, bits, meta.p, keys.p, vals.p);
    unsigned long long* sk;
    int* sv;
    SCAL_TRY(sorter.sort(s, keys.p, vals.p, d_n, 0, 3 * bits, &sk, &sv));
    hipLaunchKernelGGL(k_vox_heads, dim3(nb), dim3(256), 0, s, sk, d_n, blockcnt.p);
    launch_scan_inplace(s, blockcnt.p, d_n, 256, 1, d_n_out);
    hipLaunchKernelGGL(k_vox_reduce, dim3(nb), dim3(256), 0, s, sk, sv, d_n, blockcnt.p, in, out);
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

__global__ void k_deinterleave(const float* __restrict__ aos, int n, SoA4 o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float4 p = reinterpret_cast<const float4*>(aos)[i];
        o.x[i] = p.x, o.y[i] = p.y, o.z[i] = p.z, o.w[i] = p.w;
    }
}
__global__ void k_interleave4(const int* __restrict__ d_n, CSoA4 in, float* __restrict__ aos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *d_n) reinterpret_cast<float4*>(aos)[i] = make_float4(in.x[i], in.y[i], in.z[i], in.w[i]);
}

void launch_deinterleave(hipStream_t s, const float* aos, int n, SoA4 o) {
    if (n > 0) hipLaunchKernelGGL(k_deinterleave, dim3(div_up(n, 256)), dim3(256), 0, s, aos, n, o);
}
void launch_interleave(hipStream_t s, const int* d_n, int n_cap, CSoA4 in, float* aos) {
    if (n_cap > 0) hipLaunchKernelGGL(k_interleave4, dim3(div_up(n_cap, 256)), dim3(256), 0, s, d_n, in, aos);
}

}  // namespace scal

using namespace scal;

struct scal_voxel {
    int device = 0, cap = 0;
    hipStream_t stream = nullptr;
    VoxelFilter vf;
    DevBuf<float> aos, ix, iy, iz, iw, ox, oy, oz, ow;
    DevBuf<int> d_n;  // [0] in, [1] out
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
        (void)hipStreamSynchronize(c->stream);
        release_stream(c->device);
    }
    delete c;
}

extern "C" int scal_voxel_downsample(scal_voxel_t* c, const float* xyzi, int n, float leaf, float* out_xyzi, int* n_out) {
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
    SCAL_HIP(hipMemcpyAsync(c->aos.p, xyzi, sizeof(float) * 4 * n, hipMemcpyHostToDevice, s));
    SCAL_HIP(hipMemcpyAsync(c->d_n.p, &n, sizeof(int), hipMemcpyHostToDevice, s));
    SoA4 in{c->ix.p, c->iy.p, c->iz.p, c->iw.p}, o{c->ox.p, c->oy.p, c->oz.p, c->ow.p};
    launch_deinterleave(s, c->aos.p, n, in);
    SCAL_TRY(c->vf.run(s, CSoA4{in.x, in.y, in.z, in.w}, c->d_n.p, leaf, 16, o, c->d_n.p + 1));
    launch_interleave(s, c->d_n.p + 1, n, CSoA4{o.x, o.y, o.z, o.w}, c->aos.p);
    VoxMeta hm;
    SCAL_HIP(hipMemcpyAsync(&hm, c->vf.meta.p, sizeof hm, hipMemcpyDeviceToHost, s));
    int m = 0;
    SCAL_HIP(hipMemcpyAsync(&m, c->d_n.p + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    SCAL_HIP(hipStreamSynchronize(s));
    if (hm.error) {
        set_error("voxel grid needs more than 65536 cells along an axis");
        return SCAL_E_CAPACITY;
    }
    SCAL_HIP(hipMemcpyAsync(out_xyzi, c->aos.p, sizeof(float) * 4 * m, hipMemcpyDeviceToHost, s));
    SCAL_HIP(hipStreamSynchronize(s));
    *n_out = m;
    return SCAL_OK;
}
