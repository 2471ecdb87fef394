// Loop-closure verification ICP on gfx950.  Replaces the pcl::IterativeClosestPoint call of
// doICPVirtualRelative, /root/reference/src/laserPosegraphOptimization.cpp:518-535 (setMaxCorrespondenceDistance(150),
// setMaximumIterations(100), setTransformationEpsilon(1e-6), setEuclideanFitnessEpsilon(1e-6), setRANSACIterations(0),
// hasConverged(), getFitnessScore()).  The submaps that feed it (:472-494: keyframes moved by ONE root pose, concatenated,
// VoxelGrid 0.4) are scal_mapmerge_* + scal_voxel_*.
//
// Per iteration (PCL icp.hpp computeTransformation, restated in oracle/icp.cpp):
//   k_icp_nn      exact nearest target point of every current source point: 1024 queries x 2048 targets per workgroup (4 queries
//                 per thread), targets staged in LDS and read as wave-wide broadcasts, (f32 squared distance, target index) keys merged across
//                 target chunks with 64-bit atomicMin (= smallest distance, lowest index on ties)
//   k_icp_reduce  correspondences within max distance -> n, sum p, sum q, sum q p^T, sum d^2 in f64: fixed-order workgroup
//                 partials, summed in workgroup order by the last kernel and written to pinned host memory
//   host          Horn's closed form of the rigid least-squares fit (what TransformationEstimationSVD / Eigen::umeyama solve),
//                 PCL's DefaultConvergenceCriteria
//   k_icp_apply   the increment applied to the source cloud in place, f32 as PCL does
// The reference pays a kd-tree query per point and iteration on the CPU.  Here the target is bucketed into a cell grid once per
// alignment and a wave per query searches it ring by ring (k_icp_grid_nn, exact); queries the grid cannot settle within three
// rings fall back to the dense sweep of k_icp_nn, which is also the whole search when scal_icp_set_search(ctx, 0) asks for it.
#include "common.hpp"
#include "device_utils.hpp"
#include <cmath>
#include <cstring>
#include <limits>

namespace scal {

constexpr int ICP_QPT = 4;     // queries per thread
constexpr int ICP_QT = 256 * ICP_QPT;  // queries per workgroup
constexpr int ICP_TC = 2048;   // targets per workgroup
constexpr int ICP_NSUM = 17;   // n, sum p[3], sum q[3], sum q p^T[9], sum d^2

// qlist == nullptr: queries are cur[0 .. n_src); else the *d_nq queries qlist[] (the cell grid's unresolved ones; surplus
// workgroups leave at once)
__global__ void __launch_bounds__(256) k_icp_nn(const float4* __restrict__ cur, int n_src, const float4* __restrict__ tgt, int n_tgt,
                                                unsigned long long* __restrict__ best, const int* __restrict__ qlist,
                                                const int* __restrict__ d_nq) {
    __shared__ float4 st[ICP_TC];
    if (qlist) {
        n_src = *d_nq;
        if (static_cast<int>(blockIdx.x) * ICP_QT >= n_src) return;
    }
    const int t0 = blockIdx.y * ICP_TC;
    const int tn = min(ICP_TC, n_tgt - t0);
    for (int i = threadIdx.x; i < tn; i += 256) st[i] = tgt[t0 + i];
    __syncthreads();
    // every thread owns ICP_QPT queries and walks the whole chunk: one LDS broadcast read feeds ICP_QPT distance evaluations
    float qx[ICP_QPT], qy[ICP_QPT], qz[ICP_QPT], bd[ICP_QPT];
    int bi[ICP_QPT], qid[ICP_QPT];
#pragma unroll
    for (int u = 0; u < ICP_QPT; ++u) {
        const int qi = blockIdx.x * ICP_QT + u * 256 + threadIdx.x;
        qid[u] = qi < n_src ? (qlist ? qlist[qi] : qi) : -1;
        const float4 q = qid[u] >= 0 ? cur[qid[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
        qx[u] = q.x, qy[u] = q.y, qz[u] = q.z, bd[u] = 3.4e38f, bi[u] = -1;
    }
#pragma unroll 2
    for (int t = 0; t < tn; ++t) {
        const float4 p = st[t];
#pragma unroll
        for (int u = 0; u < ICP_QPT; ++u) {
            const float dx = qx[u] - p.x, dy = qy[u] - p.y, dz = qz[u] - p.z;
            float d = dx * dx;  // FLANN L2_Simple<float>, the metric of pcl::KdTreeFLANN
            d += dy * dy;
            d += dz * dz;
            if (d < bd[u]) bd[u] = d, bi[u] = t;  // ascending t: the lowest index of equal distances is kept
        }
    }
#pragma unroll
    for (int u = 0; u < ICP_QPT; ++u) {
        if (qid[u] >= 0 && bi[u] >= 0)
            atomicMin(&best[qid[u]], (static_cast<unsigned long long>(__float_as_uint(bd[u])) << 32) | static_cast<unsigned>(t0 + bi[u]));
    }
}

// ---- cell grid over the target (built once per alignment) ---------------------------------------------------------------------
// Exact nearest neighbour without the dense sweep: target points bucketed into <= 2^18 cubic cells (counting sort, original
// index kept beside each point); a wave per query examines the cube of cells around the query ring by ring.  After ring k every
// unexamined point lies more than k cell widths away, so the search stops as soon as the best squared distance is below
// ((k - 0.05) c)^2 (the margin covers the f32 cell assignment); keys are (f32 squared distance, original index) exactly as in the dense
// sweep, so equal distances still resolve to the lowest index.  Queries with nothing that close within ICP_RINGS rings (far
// outliers, a poor initial guess) go to the dense sweep as a list - the result is the same either way.
constexpr int ICP_NCELL = 1 << 18;
constexpr int ICP_RINGS = 3;
struct IcpGrid {
    float ox, oy, oz, cell, inv;
    int dx, dy, dz;
};

__global__ void __launch_bounds__(256) k_icp_bbox(const float4* __restrict__ tgt, int n, unsigned* __restrict__ mm) {
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 p = tgt[i];
        if (!(fabsf(p.x) < 3.4e38f && fabsf(p.y) < 3.4e38f && fabsf(p.z) < 3.4e38f)) continue;  // never a neighbour either
        lo[0] = fminf(lo[0], p.x), lo[1] = fminf(lo[1], p.y), lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x), hi[1] = fmaxf(hi[1], p.y), hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = wave_min_i(static_cast<int>(float_to_ordered(lo[k]) ^ 0x80000000u));
        const int b = wave_max_i(static_cast<int>(float_to_ordered(hi[k]) ^ 0x80000000u));
        if (lane_id() == 0) {
            atomicMin(&mm[k], static_cast<unsigned>(a) ^ 0x80000000u);
            atomicMax(&mm[3 + k], static_cast<unsigned>(b) ^ 0x80000000u);
        }
    }
}

// one thread: origin, cell width (>= min_cell, grown until the grid fits ICP_NCELL cells), dimensions; resets the box for the next call
__global__ void k_icp_grid_setup(unsigned* __restrict__ mm, IcpGrid* __restrict__ g, float min_cell) {
    float lo[3], ext[3];
    bool ok = true;
    for (int k = 0; k < 3; ++k) {
        lo[k] = ordered_to_float(mm[k]);
        const float hi = ordered_to_float(mm[3 + k]);
        ext[k] = hi - lo[k];
        ok = ok && hi >= lo[k];
        mm[k] = 0xffffffffu, mm[3 + k] = 0u;
    }
    IcpGrid r;
    r.ox = lo[0], r.oy = lo[1], r.oz = lo[2], r.cell = min_cell, r.inv = 1.f / min_cell, r.dx = r.dy = r.dz = 0;
    if (ok) {
        float c = min_cell;
        for (int it = 0; it < 200; ++it) {
            const long long dx = static_cast<long long>(ext[0] / c) + 2, dy = static_cast<long long>(ext[1] / c) + 2,
                            dz = static_cast<long long>(ext[2] / c) + 2;
            if (dx * dy * dz <= ICP_NCELL) {
                r.cell = c, r.inv = 1.f / c, r.dx = static_cast<int>(dx), r.dy = static_cast<int>(dy), r.dz = static_cast<int>(dz);
                break;
            }
            c *= 1.1f;
        }
    }
    *g = r;
}

__device__ __forceinline__ int icp_cell_coord(float v, float o, float inv) {
    const float f = floorf((v - o) * inv);
    return static_cast<int>(fminf(fmaxf(f, -1048576.f), 1048576.f));  // NaN -> -2^20: outside every grid
}

__global__ void __launch_bounds__(256) k_icp_cell_count(const float4* __restrict__ tgt, int n, const IcpGrid* __restrict__ gp,
                                                        int* __restrict__ count, int* __restrict__ cell_of) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const IcpGrid g = *gp;
    const float4 p = tgt[i];
    const int x = icp_cell_coord(p.x, g.ox, g.inv), y = icp_cell_coord(p.y, g.oy, g.inv), z = icp_cell_coord(p.z, g.oz, g.inv);
    int cell = -1;  // non-finite points are in no cell: they can never be a nearest neighbour (d < bd fails for NaN)
    if (static_cast<unsigned>(x) < static_cast<unsigned>(g.dx) && static_cast<unsigned>(y) < static_cast<unsigned>(g.dy) &&
        static_cast<unsigned>(z) < static_cast<unsigned>(g.dz)) {
        cell = (z * g.dy + y) * g.dx + x;
        atomicAdd(&count[cell], 1);
    }
    cell_of[i] = cell;
}

// exclusive scan of count[ICP_NCELL] in two launches: workgroup sums, then every workgroup adds the sums in front of it
__global__ void __launch_bounds__(256) k_icp_scan_sums(const int* __restrict__ count, int* __restrict__ bsum) {
    const int4 v = reinterpret_cast<const int4*>(count)[blockIdx.x * 256 + threadIdx.x];
    __shared__ int red[4];
    const int s = wave_sum((v.x + v.y) + (v.z + v.w));
    if (lane_id() == 0) red[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256) k_icp_scan_apply(int* __restrict__ count, const int* __restrict__ bsum, int* __restrict__ start,
                                                        int* __restrict__ total) {
    __shared__ int smem[17];
    __shared__ int red[4];
    // sum of the workgroup sums in front of this one (gridDim.x = ICP_NCELL / 1024 = 256 = blockDim.x)
    const int mine = static_cast<int>(threadIdx.x) < static_cast<int>(blockIdx.x) ? bsum[threadIdx.x] : 0;
    const int ws = wave_sum(mine);
    if (lane_id() == 0) red[wave_id()] = ws;
    __syncthreads();
    const int base = (red[0] + red[1]) + (red[2] + red[3]);
    int4 v = reinterpret_cast<const int4*>(count)[blockIdx.x * 256 + threadIdx.x];
    int tot;
    const int ex = block_exclusive_scan((v.x + v.y) + (v.z + v.w), smem, &tot) + base;
    int4 o;
    o.x = ex, o.y = ex + v.x, o.z = o.y + v.y, o.w = o.z + v.z;
    reinterpret_cast<int4*>(start)[blockIdx.x * 256 + threadIdx.x] = o;
    reinterpret_cast<int4*>(count)[blockIdx.x * 256 + threadIdx.x] = make_int4(0, 0, 0, 0);  // becomes the fill cursor, zero again after
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) start[ICP_NCELL] = base + tot, *total = base + tot;
}

__global__ void __launch_bounds__(256) k_icp_cell_fill(const float4* __restrict__ tgt, int n, const int* __restrict__ cell_of,
                                                       const int* __restrict__ start, int* __restrict__ cursor, float4* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cell = cell_of[i];
    if (cell < 0) return;
    float4 p = tgt[i];
    p.w = __int_as_float(i);
    sorted[start[cell] + atomicAdd(&cursor[cell], 1)] = p;
}

// one wave per query
__global__ void __launch_bounds__(256) k_icp_grid_nn(const float4* __restrict__ cur, int n_src, const IcpGrid* __restrict__ gp,
                                                     const int* __restrict__ start, const float4* __restrict__ sorted,
                                                     unsigned long long* __restrict__ best, int* __restrict__ un_list, int* __restrict__ d_nun) {
    const int qi = blockIdx.x * 4 + wave_id();
    if (qi >= n_src) return;
    const IcpGrid g = *gp;
    const int lane = lane_id();
    const float4 q = cur[qi];
    const int cx = icp_cell_coord(q.x, g.ox, g.inv), cy = icp_cell_coord(q.y, g.oy, g.inv), cz = icp_cell_coord(q.z, g.oz, g.inv);
    unsigned long long bk = ~0ull;
    bool done = false;
    for (int k = 1; k <= ICP_RINGS && !done; ++k) {
        const int w = 2 * k + 1, ncube = w * w * w;
        for (int j0 = 0; j0 < ncube; j0 += 64) {  // 64 cells at a time, one per lane
            const int j = j0 + lane;
            int s0 = 0, cnt = 0;
            if (j < ncube) {
                const int jz = j / (w * w), rem = j - jz * w * w, jy = rem / w, jx = rem - jy * w;
                const int ax = jx - k, ay = jy - k, az = jz - k;
                const int x = cx + ax, y = cy + ay, z = cz + az;
                const bool fresh = k == 1 || max(max(abs(ax), abs(ay)), abs(az)) == k;  // inner cells were examined in an earlier ring
                if (fresh && static_cast<unsigned>(x) < static_cast<unsigned>(g.dx) && static_cast<unsigned>(y) < static_cast<unsigned>(g.dy) &&
                    static_cast<unsigned>(z) < static_cast<unsigned>(g.dz)) {
                    const int cell = (z * g.dy + y) * g.dx + x;
                    s0 = start[cell];
                    cnt = start[cell + 1] - s0;
                }
            }
            // the points of these cells as one list spread over the lanes: candidate t belongs to the lane whose running count
            // first exceeds t (six-step search over the inclusive scan), so a crowded cell costs no lane more than an empty one
            const int incl = wave_inclusive_scan(cnt);
            const int total = __shfl(incl, 63, 64);
            const int excl = incl - cnt;
            for (int base = 0; base < total; base += 64) {
                const int t = base + lane;
                int owner = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const int v = __shfl(incl, owner + step - 1, 64);
                    if (v <= t) owner += step;
                }
                owner = min(owner, 63);
                const int idx = __shfl(s0, owner, 64) + (t - __shfl(excl, owner, 64));
                if (t < total) {
                    const float4 p = sorted[idx];
                    const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
                    float d = dx * dx;  // the dense sweep's arithmetic
                    d += dy * dy;
                    d += dz * dz;
                    if (d < 3.4e38f) {
                        const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(d)) << 32) | static_cast<unsigned>(__float_as_int(p.w));
                        bk = key < bk ? key : bk;
                    }
                }
            }
        }
        bk = wave_min_u64(bk);
        const float bound = (static_cast<float>(k) - 0.05f) * g.cell;  // 0.05 cells: the f32 cell assignment of either point
        done = bk != ~0ull && __uint_as_float(static_cast<unsigned>(bk >> 32)) < bound * bound;
    }
    if (lane == 0) {
        if (done)
            best[qi] = bk;
        else
            un_list[atomicAdd(d_nun, 1)] = qi;
    }
}

// partial sums of one workgroup of 256 source points; resets the keys for the next sweep
__global__ void __launch_bounds__(256) k_icp_reduce(const float4* __restrict__ cur, int n_src, const float4* __restrict__ tgt,
                                                    unsigned long long* __restrict__ best, float max2, double* __restrict__ partials,
                                                    int* __restrict__ d_nun) {
    __shared__ double red[4][ICP_NSUM];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *d_nun = 0;  // the cell grid's unresolved-query list is consumed by now
    double v[ICP_NSUM];
#pragma unroll
    for (int k = 0; k < ICP_NSUM; ++k) v[k] = 0.0;
    if (i < n_src) {
        const unsigned long long key = best[i];
        best[i] = ~0ull;
        const float d = __uint_as_float(static_cast<unsigned>(key >> 32));
        if (key != ~0ull && !(d > max2)) {
            const float4 p = cur[i];
            const float4 q = tgt[static_cast<unsigned>(key & 0xffffffffu)];
            v[0] = 1.0;
            v[1] = p.x, v[2] = p.y, v[3] = p.z;
            v[4] = q.x, v[5] = q.y, v[6] = q.z;
            const double qq[3] = {q.x, q.y, q.z}, pp[3] = {p.x, p.y, p.z};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[7 + 3 * r + c] = qq[r] * pp[c];
            v[16] = d;
        }
    }
#pragma unroll
    for (int k = 0; k < ICP_NSUM; ++k) {
        const double s = wave_sum(v[k]);
        if (lane_id() == 0) red[wave_id()][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < ICP_NSUM) partials[blockIdx.x * ICP_NSUM + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// sums the workgroup partials in workgroup order and hands them to the host (pinned, device-visible memory)
__global__ void __launch_bounds__(64) k_icp_final(const double* __restrict__ partials, int nb, double* __restrict__ host_out) {
    if (threadIdx.x < ICP_NSUM) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) s += partials[b * ICP_NSUM + threadIdx.x];
        host_out[threadIdx.x] = s;
    }
}

struct Mat34f {
    float m[12];
};
__global__ void __launch_bounds__(256) k_icp_apply(const float4* __restrict__ in, float4* __restrict__ out, int n, Mat34f T) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    float4 o;
    o.x = ((T.m[0] * p.x + T.m[1] * p.y) + T.m[2] * p.z) + T.m[3];
    o.y = ((T.m[4] * p.x + T.m[5] * p.y) + T.m[6] * p.z) + T.m[7];
    o.z = ((T.m[8] * p.x + T.m[9] * p.y) + T.m[10] * p.z) + T.m[11];
    o.w = p.w;
    out[i] = o;
}

__global__ void k_icp_fill(unsigned long long* __restrict__ best, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) best[i] = ~0ull;
}

namespace {
// largest eigenpair of a symmetric 4x4 by cyclic Jacobi -> unit quaternion (w, x, y, z)
void eig4_max(double a[4][4], double q[4]) {
    double v[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) (i == j ? diag : off) += a[i][j] * a[i][j];
        if (off <= 1e-40 * diag || off == 0.0) break;
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                if (a[p][r] == 0.0) continue;
                const double theta = (a[r][r] - a[p][p]) / (2.0 * a[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) {
                    const double akp = a[k][p], akr = a[k][r];
                    a[k][p] = c * akp - s * akr, a[k][r] = s * akp + c * akr;
                }
                for (int k = 0; k < 4; ++k) {
                    const double apk = a[p][k], ark = a[r][k];
                    a[p][k] = c * apk - s * ark, a[r][k] = s * apk + c * ark;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = v[k][p], vkr = v[k][r];
                    v[k][p] = c * vkp - s * vkr, v[k][r] = s * vkp + c * vkr;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < 4; ++i)
        if (a[i][i] > a[best][best]) best = i;
    double nrm = 0;
    for (int k = 0; k < 4; ++k) nrm += v[k][best] * v[k][best];
    nrm = std::sqrt(nrm);
    for (int k = 0; k < 4; ++k) q[k] = v[k][best] / nrm;
}

// rigid increment (as the Matrix4f PCL keeps) from the correspondence sums: Horn's quaternion form of the Umeyama problem
void transform_from_sums(const double* sums, float* T) {
    const double n = sums[0];
    double mp[3], mq[3], S[3][3];
    for (int k = 0; k < 3; ++k) mp[k] = sums[1 + k] / n, mq[k] = sums[4 + k] / n;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S[c][r] = sums[7 + 3 * r + c] / n - mq[r] * mp[c];
    double N[4][4] = {{S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
                      {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
                      {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
                      {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double q[4];
    eig4_max(N, q);
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                            {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                            {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    for (int r = 0; r < 3; ++r) {
        double t = mq[r];
        for (int c = 0; c < 3; ++c) {
            T[4 * r + c] = static_cast<float>(R[r][c]);
            t -= R[r][c] * mp[c];
        }
        T[4 * r + 3] = static_cast<float>(t);
    }
    T[12] = T[13] = T[14] = 0.f, T[15] = 1.f;
}
}  // namespace

}  // namespace scal

using namespace scal;

struct scal_icp {
    scal_icp_config cfg;
    hipStream_t stream = nullptr;
    DevBuf<float4> src, cur, tgt;
    const float4* tgt_cur = nullptr;  // the target of the alignment in flight: tgt, or the caller's device cloud
    DevBuf<unsigned long long> best;
    DevBuf<double> partials;
    PinBuf<double> h_sums;
    // cell grid over the target
    int search = 1;  // 0: dense sweep, 1: cell grid with the dense sweep for unresolved queries
    bool grid_built = false;
    DevBuf<IcpGrid> grid;
    DevBuf<unsigned> mm;
    DevBuf<int> count, start, bsum, cell_of, un_list, d_nun;
    DevBuf<float4> sorted;
};

extern "C" int scal_icp_create(const scal_icp_config* cfg, scal_icp_t** out) {
    if (!cfg || !out || cfg->max_source <= 0 || cfg->max_target <= 0 || cfg->max_iterations <= 0 || !(cfg->max_corr_dist > 0)) {
        set_error("scal_icp_create: bad argument");
        return SCAL_E_ARG;
    }
    *out = nullptr;
    SCAL_TRY(select_device(cfg->device));
    auto* c = new scal_icp();
    c->cfg = *cfg;
    int rc = SCAL_OK;
    auto A = [&](int r) { if (rc == SCAL_OK) rc = r; };
    A(c->src.alloc(cfg->max_source)); A(c->cur.alloc(cfg->max_source)); A(c->tgt.alloc(cfg->max_target));
    A(c->best.alloc(cfg->max_source));
    A(c->partials.alloc(static_cast<size_t>(ICP_NSUM) * (div_up(cfg->max_source, 256) + 1)));
    A(c->h_sums.alloc(ICP_NSUM));
    A(c->grid.alloc(1)); A(c->mm.alloc(6)); A(c->count.alloc(ICP_NCELL)); A(c->start.alloc(ICP_NCELL + 4)); A(c->bsum.alloc(256));
    A(c->cell_of.alloc(cfg->max_target)); A(c->sorted.alloc(cfg->max_target)); A(c->un_list.alloc(cfg->max_source)); A(c->d_nun.alloc(2));
    if (rc == SCAL_OK && acquire_stream(cfg->device, &c->stream) != SCAL_OK) rc = SCAL_E_HIP;
    if (rc == SCAL_OK) {  // initialised on the context's own stream (the legacy null stream is not ordered against it)
        const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
        if (hipMemcpyAsync(c->mm.p, init, sizeof init, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemsetAsync(c->d_nun.p, 0, 2 * sizeof(int), c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
            rc = SCAL_E_HIP;
    }
    if (rc != SCAL_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return SCAL_OK;
}

extern "C" int scal_icp_set_search(scal_icp_t* c, int mode) {
    if (!c || (mode != 0 && mode != 1)) {
        set_error("scal_icp_set_search: bad argument");
        return SCAL_E_ARG;
    }
    c->search = mode;
    return SCAL_OK;
}

extern "C" void scal_icp_destroy(scal_icp_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        release_stream(c->cfg.device);
    }
    delete c;
}

// one correspondence sweep of `pts` against the target: sums -> c->h_sums (waits)
static int icp_sweep(scal_icp* c, const float4* pts, int n_src, int n_tgt, float max2) {
    hipStream_t s = c->stream;
    const int nb = div_up(n_src, 256);
    const int* no_list = nullptr;
    if (c->grid_built) {
        SCAL_LAUNCH_PROF("k_icp_grid_nn", k_icp_grid_nn, dim3(div_up(n_src, 4)), dim3(256), 0, s, pts, n_src, c->grid.p, c->start.p, c->sorted.p,
                         c->best.p, c->un_list.p, c->d_nun.p);
        SCAL_LAUNCH_PROF("k_icp_nn", k_icp_nn, dim3(div_up(n_src, ICP_QT), div_up(n_tgt, ICP_TC)), dim3(256), 0, s, pts, n_src, c->tgt_cur, n_tgt,
                         c->best.p, c->un_list.p, c->d_nun.p);
    } else {
        SCAL_LAUNCH_PROF("k_icp_nn", k_icp_nn, dim3(div_up(n_src, ICP_QT), div_up(n_tgt, ICP_TC)), dim3(256), 0, s, pts, n_src, c->tgt_cur, n_tgt,
                         c->best.p, no_list, no_list);
    }
    hipLaunchKernelGGL(k_icp_reduce, dim3(nb), dim3(256), 0, s, pts, n_src, c->tgt_cur, c->best.p, max2, c->partials.p, c->d_nun.p);
    hipLaunchKernelGGL(k_icp_final, dim3(1), dim3(64), 0, s, c->partials.p, nb, c->h_sums.p);
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(hipStreamSynchronize(s));
    return SCAL_OK;
}

// clouds in host memory (on_device = false: staged into the context) or already in HBM (the target is then used in place)
static int icp_align(scal_icp* c, const float* src_xyzi, int n_src, const float* tgt_xyzi, int n_tgt, bool on_device, scal_icp_result* res) {
    if (!c || !res || n_src < 0 || n_tgt < 0 || (n_src && !src_xyzi) || (n_tgt && !tgt_xyzi)) {
        set_error("scal_icp_align: bad argument");
        return SCAL_E_ARG;
    }
    if (n_src > c->cfg.max_source || n_tgt > c->cfg.max_target) {
        set_error("scal_icp_align: cloud larger than the context capacity (source %d, target %d)", c->cfg.max_source, c->cfg.max_target);
        return SCAL_E_TOO_MANY;
    }
    std::memset(res, 0, sizeof *res);
    float F[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int k = 0; k < 16; ++k) res->T[k] = F[k];
    res->fitness = std::numeric_limits<double>::max();
    if (n_src == 0 || n_tgt == 0) {  // pcl::Registration::align refuses empty clouds: not converged
        res->state = 5;
        return SCAL_OK;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    SCAL_HIP(hipMemcpyAsync(c->src.p, src_xyzi, sizeof(float) * 4 * n_src, kind, s));
    SCAL_HIP(hipMemcpyAsync(c->cur.p, src_xyzi, sizeof(float) * 4 * n_src, kind, s));
    const float4* tgt = c->tgt.p;
    if (on_device)
        tgt = reinterpret_cast<const float4*>(tgt_xyzi);
    else
        SCAL_HIP(hipMemcpyAsync(c->tgt.p, tgt_xyzi, sizeof(float) * 4 * n_tgt, hipMemcpyHostToDevice, s));
    c->tgt_cur = tgt;
    hipLaunchKernelGGL(k_icp_fill, dim3(div_up(n_src, 256)), dim3(256), 0, s, c->best.p, n_src);
    c->grid_built = false;
    if (c->search == 1 && n_tgt >= 4 * ICP_TC) {  // small targets: the dense sweep is a handful of workgroups anyway
        SCAL_HIP(hipMemsetAsync(c->count.p, 0, sizeof(int) * ICP_NCELL, s));
        hipLaunchKernelGGL(k_icp_bbox, dim3(std::min(div_up(n_tgt, 256), 1024)), dim3(256), 0, s, tgt, n_tgt, c->mm.p);
        hipLaunchKernelGGL(k_icp_grid_setup, dim3(1), dim3(1), 0, s, c->mm.p, c->grid.p, 1.0f);
        hipLaunchKernelGGL(k_icp_cell_count, dim3(div_up(n_tgt, 256)), dim3(256), 0, s, tgt, n_tgt, c->grid.p, c->count.p, c->cell_of.p);
        hipLaunchKernelGGL(k_icp_scan_sums, dim3(ICP_NCELL / 1024), dim3(256), 0, s, c->count.p, c->bsum.p);
        hipLaunchKernelGGL(k_icp_scan_apply, dim3(ICP_NCELL / 1024), dim3(256), 0, s, c->count.p, c->bsum.p, c->start.p, c->d_nun.p + 1);
        SCAL_LAUNCH_PROF("k_icp_cell_fill", k_icp_cell_fill, dim3(div_up(n_tgt, 256)), dim3(256), 0, s, tgt, n_tgt, c->cell_of.p, c->start.p,
                         c->count.p, c->sorted.p);
        SCAL_HIP(hipGetLastError());
        c->grid_built = true;
    }
    const float max2 = static_cast<float>(c->cfg.max_corr_dist * c->cfg.max_corr_dist);
    const double te = c->cfg.transformation_epsilon, fe = c->cfg.fitness_epsilon;
    int it = 0, st = 0;
    bool converged = false;
    double prev_mse = std::numeric_limits<double>::max();
    while (!converged) {
        SCAL_TRY(icp_sweep(c, c->cur.p, n_src, n_tgt, max2));
        const double* sums = c->h_sums.p;
        res->n_correspondences = static_cast<int>(sums[0]);
        if (sums[0] < 3) {  // icp.hpp: "Not enough correspondences found"
            st = 5;
            break;
        }
        float T[16];
        transform_from_sums(sums, T);
        Mat34f M;
        std::memcpy(M.m, T, sizeof M.m);
        hipLaunchKernelGGL(k_icp_apply, dim3(div_up(n_src, 256)), dim3(256), 0, s, c->cur.p, c->cur.p, n_src, M);
        float G[16];
        for (int r = 0; r < 4; ++r)
            for (int cc = 0; cc < 4; ++cc) G[4 * r + cc] = ((T[4 * r] * F[cc] + T[4 * r + 1] * F[4 + cc]) + T[4 * r + 2] * F[8 + cc]) + T[4 * r + 3] * F[12 + cc];
        std::memcpy(F, G, sizeof F);
        ++it;
        // DefaultConvergenceCriteria::hasConverged (PCL 1.8)
        if (it >= c->cfg.max_iterations) {
            st = 1, converged = true;
            break;
        }
        const double cos_angle = 0.5 * (static_cast<double>(T[0]) + T[5] + T[10] - 1.0);
        const double tr2 = static_cast<double>(T[3]) * T[3] + static_cast<double>(T[7]) * T[7] + static_cast<double>(T[11]) * T[11];
        if (cos_angle >= 1.0 - te && tr2 <= te) {
            st = 2, converged = true;
            break;
        }
        const double mse = sums[16] / sums[0];
        if (std::fabs(mse - prev_mse) < 1e-12) {
            st = 3, converged = true;
            break;
        }
        if (std::fabs(mse - prev_mse) / prev_mse < fe) {
            st = 4, converged = true;
            break;
        }
        prev_mse = mse;
    }
    // getFitnessScore(): the original source moved by the final transform, no distance cap
    Mat34f M;
    std::memcpy(M.m, F, sizeof M.m);
    hipLaunchKernelGGL(k_icp_apply, dim3(div_up(n_src, 256)), dim3(256), 0, s, c->src.p, c->cur.p, n_src, M);
    SCAL_TRY(icp_sweep(c, c->cur.p, n_src, n_tgt, 3.4e38f));
    res->converged = converged ? 1 : 0;
    res->iterations = it;
    res->state = st;
    res->fitness = c->h_sums.p[0] > 0 ? c->h_sums.p[16] / c->h_sums.p[0] : std::numeric_limits<double>::max();
    for (int k = 0; k < 16; ++k) res->T[k] = F[k];
    return SCAL_OK;
}

extern "C" int scal_icp_align(scal_icp_t* c, const float* src_xyzi, int n_src, const float* tgt_xyzi, int n_tgt, scal_icp_result* res) {
    return icp_align(c, src_xyzi, n_src, tgt_xyzi, n_tgt, false, res);
}

extern "C" int scal_icp_align_device(scal_icp_t* c, const float* d_src_xyzi, int n_src, const float* d_tgt_xyzi, int n_tgt, scal_icp_result* res) {
    return icp_align(c, d_src_xyzi, n_src, d_tgt_xyzi, n_tgt, true, res);
}
