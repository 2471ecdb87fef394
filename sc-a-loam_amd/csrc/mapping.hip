// Stage C on gfx950: scan-to-map registration.  Replaces process(), /root/reference/src/laserMapping.cpp:310-802,
// :845-849 (see include/scaloam_hip.h).
//
// Map layout in HBM.  The reference keeps 21x21x11 cubes of 50 m, each a separate PointCloud (:74-104).  Here each
// feature class (corner, surf) is ONE SoA array x[] y[] z[] i[] plus a packed absolute cube coordinate per point;
// the rolling window (:313-508) is index arithmetic on the host (cen offsets), never a copy: points whose cube
// leaves the window are dropped by the next re-filter pass, which is exactly the slab the reference clears.
//
// Per scan, one stream, one host sync at the end (pose + statistics read-back):
//   stack downsample (:543-551)        VoxelFilter (voxel.hip) on the incoming corner / surf clouds
//   submap gather + kd-tree (:510-560) replaced by a 1 m uniform cell grid over the 5x5x3 valid cubes
//                                      (250x250x150 cells): k_grid_count / k_grid_alloc / k_grid_fill bin the valid
//                                      map points with atomics; both acceptance gates need the 5th neighbour
//                                      within 1 m (:585, :653) so the 27 surrounding cells hold every candidate
//   association (:578-688)             k_assoc_edge / k_assoc_plane: exact 5-NN by f32 (dx^2+dy^2)+dz^2, ties by map
//                                      index; 3x3 Jacobi eigen-decomposition (edge) or pivoted Householder QR
//                                      (plane) in f64 registers -> residual blocks
//   solve (:713-721) x2 (:563)         lm_dev.hpp, all on the device
//   insert + per-cube voxel (:738-802) k_insert_keys -> radix sort by (cube, voxel) -> k_map_reduce; cubes outside
//                                      the 5x5x3 valid set keep raw points in arrival order, as the reference does
//   registration (:845-849)            k_transform_cloud
#include "common.hpp"
#include <mutex>
#include "device_utils.hpp"
#include "voxel_dev.hpp"
#include "radix_sort.hpp"
#include "lm_dev.hpp"
#include "features_dev.hpp"
#include <cmath>
#include <algorithm>

namespace scal {

constexpr int CW = 21, CH = 21, CD = 11;       // laserCloudWidth/Height/Depth (:77-79)
constexpr int GX = 250, GY = 250, GZ = 150;    // 1 m cells over the 5x5x3 valid cubes
constexpr int GCELLS = GX * GY * GZ;
constexpr unsigned long long MAP_NOMERGE = 127ull;

struct MapParams {
    int cenW, cenH, cenD;       // laserCloudCenWidth/Height/Depth after this scan's shifts
    int cI, cJ, cK;             // centerCube indices
    int ox, oy, oz;             // grid origin [m] = 50*(c-2-cen)-25 (integers)
    float inv_line, inv_plane;  // 1/leaf as PCL computes it (f32)
};

struct MapCounters {
    int n_corner_in, n_surf_in;          // incoming clouds
    int n_corner_stack, n_surf_stack;    // after the stack voxel grid
    int n_valid[2];                      // map points inside the 5x5x3 window (corner, surf)
    int cursor[2];                       // grid fill cursors
    int solve_on;                        // laserCloudCornerFromMapNum > 10 && SurfFromMapNum > 50 (:555)
    int n_slots;                         // residual-block slots = n_corner_stack + n_surf_stack
    int n_live[2];                       // live residual blocks per outer iteration
    int n_edge[2], n_plane[2];
    int n_total[2];                      // map points to sort in the re-filter pass (old + new)
    int n_map_new[2];                    // map size after the re-filter
    int error;
    int merge_fail;                      // the merge insert met a case it does not handle: redo with the full sort
    int merge_neff[2];                   // new points inside the cube window (merge insert)
};

__device__ __forceinline__ int pack_cube(int ai, int aj, int ak) { return (ai + 512) | ((aj + 512) << 10) | ((ak + 512) << 20); }
__device__ __forceinline__ void unpack_cube(int p, int& ai, int& aj, int& ak) {
    ai = (p & 1023) - 512, aj = ((p >> 10) & 1023) - 512, ak = ((p >> 20) & 1023) - 512;
}
// cube coordinate of a map-frame coordinate: int((p + 25)/50) with the reference's negative fix (:742-751), minus cen
__device__ __forceinline__ int cube_abs(float p) {
    const double v = static_cast<double>(p) + 25.0;
    int c = static_cast<int>(v / 50.0);
    if (v < 0) c--;
    return c;
}

__device__ __forceinline__ int grid_cell(const MapParams& mp, float x, float y, float z) {
    int cx = static_cast<int>(floorf(x)) - mp.ox, cy = static_cast<int>(floorf(y)) - mp.oy, cz = static_cast<int>(floorf(z)) - mp.oz;
    cx = min(max(cx, 0), GX - 1), cy = min(max(cy, 0), GY - 1), cz = min(max(cz, 0), GZ - 1);
    return cx + GX * (cy + GY * cz);
}

__device__ __forceinline__ bool cube_valid(const MapParams& mp, int packed) {
    int ai, aj, ak;
    unpack_cube(packed, ai, aj, ak);
    const int I = ai + mp.cenW, J = aj + mp.cenH, K = ak + mp.cenD;
    return abs(I - mp.cI) <= 2 && abs(J - mp.cJ) <= 2 && abs(K - mp.cK) <= 1 && I >= 0 && I < CW && J >= 0 && J < CH && K >= 0 && K < CD;
}

struct MapCloud {
    float *x, *y, *z, *w;
    int* cube;
};

// ---------------------------------------------------------------------------------------------- grid build
// Both feature classes per launch: blocks [0, nb0) bin the corner map, the rest the surf map.
struct GridPts {
    float4* p;  // (x, y, z, map index as bits): one 16-byte gather per candidate instead of four 4-byte ones
};
struct GridArgs {
    MapCloud m[2];
    int n[2], nb0;
    int* cnt[2];
    int* rank[2];
    int* start[2];
    GridPts g[2];
};
__device__ __forceinline__ bool grid_part(const GridArgs& a, int& cls, int& i) {
    int b = blockIdx.x;
    cls = b < a.nb0 ? 0 : 1;
    if (cls) b -= a.nb0;
    i = b * blockDim.x + threadIdx.x;
    return i < a.n[cls];
}

__global__ void __launch_bounds__(256) k_grid_count(GridArgs a, MapParams mp, MapCounters* C) {
    __shared__ int s_valid;
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    int cls, i;
    const bool in = grid_part(a, cls, i);
    const MapCloud& m = a.m[cls];
    bool v = false;
    if (in && cube_valid(mp, m.cube[i])) {
        v = true;
        a.rank[cls][i] = atomicAdd(&a.cnt[cls][grid_cell(mp, m.x[i], m.y[i], m.z[i])], 1);
    } else if (in) {
        a.rank[cls][i] = -1;
    }
    // one global atomic per workgroup: a per-wave atomic on the same word was most of this kernel's time
    const uint64_t b = __ballot(v);
    if (lane_id() == 0 && b) atomicAdd(&s_valid, __popcll(b));
    __syncthreads();
    if (threadIdx.x == 0 && s_valid) atomicAdd(&C->n_valid[cls], s_valid);
}

// every non-empty cell gets a slice of the point pool: the cells' sizes are summed per workgroup (the point with rank 0
// speaks for its cell), one cursor atomic per workgroup
__global__ void __launch_bounds__(256) k_grid_alloc(GridArgs a, MapParams mp, MapCounters* C) {
    __shared__ int s_scan[17];
    __shared__ int s_base;
    int cls, i;
    const bool in = grid_part(a, cls, i);
    const MapCloud& m = a.m[cls];
    int c = -1, mine = 0;
    if (in && a.rank[cls][i] == 0) {
        c = grid_cell(mp, m.x[i], m.y[i], m.z[i]);
        mine = a.cnt[cls][c];
    }
    int total = 0;
    const int off = block_exclusive_scan(mine, s_scan, &total);
    if (threadIdx.x == 0) s_base = total ? atomicAdd(&C->cursor[cls], total) : 0;
    __syncthreads();
    if (c >= 0) a.start[cls][c] = s_base + off;
    if (blockIdx.x == 0 && threadIdx.x == 0) C->solve_on = (C->n_valid[0] > 10 && C->n_valid[1] > 50) ? 1 : 0;  // :555
}

__global__ void __launch_bounds__(256) k_grid_fill(GridArgs a, MapParams mp) {
    int cls, i;
    const bool in = grid_part(a, cls, i);
    const MapCloud& m = a.m[cls];
    if (in && a.rank[cls][i] >= 0) {
        const int c = grid_cell(mp, m.x[i], m.y[i], m.z[i]);
        const int p = a.start[cls][c] + a.rank[cls][i];
        const GridPts& g = a.g[cls];
        g.p[p] = make_float4(m.x[i], m.y[i], m.z[i], __int_as_float(i));
    }
}

__global__ void __launch_bounds__(256) k_grid_clear(GridArgs a, MapParams mp) {
    int cls, i;
    const bool in = grid_part(a, cls, i);
    const MapCloud& m = a.m[cls];
    if (in && a.rank[cls][i] >= 0) a.cnt[cls][grid_cell(mp, m.x[i], m.y[i], m.z[i])] = 0;
}

// ---------------------------------------------------------------------------------------------- association

// Exact 5 nearest map points of q among the 27 cells around it, ascending (distance, map index), by ONE WAVE:
// lanes 0..26 each own a neighbour cell (one round trip for the 27 (count, start) pairs); the candidates of all cells form
// one virtual list that the 64 lanes read side by side (lane j finds the cell of candidate j from the cells' running
// counts, 27 uniform compares), so a chunk of 256 candidates costs one more round trip; their 64-bit (f32 distance bits,
// map index) keys stay in registers and five wave-argmin rounds per chunk pick the result.
// Every lane returns the same ascending (key, grid position) list; position -1 = fewer than five candidates.
constexpr int KNN_CHUNK = 256;
__device__ __forceinline__ void knn5_wave(const MapParams& mp, const int* __restrict__ cnt, const int* __restrict__ start, const GridPts& g,
                                          float qx, float qy, float qz, unsigned long long (&bk)[5], int (&bp)[5]) {
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < 5; ++k) bk[k] = ~0ull, bp[k] = -1;
    const int cx = static_cast<int>(floorf(qx)) - mp.ox, cy = static_cast<int>(floorf(qy)) - mp.oy, cz = static_cast<int>(floorf(qz)) - mp.oz;
    int my_cnt = 0, my_start = 0;
    // a query further than one cell outside the grid has no map point within 1 m
    const bool inside = !(cx < -1 || cx > GX || cy < -1 || cy > GY || cz < -1 || cz > GZ);
    if (inside && lane < 27) {
        const int xx = cx + (lane % 3) - 1, yy = cy + ((lane / 3) % 3) - 1, zz = cz + (lane / 9) - 1;
        if (xx >= 0 && xx < GX && yy >= 0 && yy < GY && zz >= 0 && zz < GZ) {
            const int c = xx + GX * (yy + GY * zz);
            my_cnt = cnt[c];
            if (my_cnt) my_start = start[c];
        }
    }
    const int incl = wave_inclusive_scan(my_cnt);
    const int rel = my_start - (incl - my_cnt);  // grid position of candidate j of my cell = rel + j
    const int T = __shfl(incl, 63, 64);
    for (int base = 0; base < T; base += KNN_CHUNK) {
        unsigned long long k0[4];
        int p0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = base + lane + 64 * u;
            int c = 0;  // cell of candidate j: the first one whose running count exceeds j
#pragma unroll
            for (int cc = 0; cc < 26; ++cc) c += __builtin_amdgcn_readlane(incl, cc) <= j ? 1 : 0;
            const int t = __shfl(rel, c, 64) + j;
            k0[u] = ~0ull, p0[u] = -1;
            if (j < T) {
                // FLANN L2_Simple<float>: ((0 + dx^2) + dy^2) + dz^2
                const float4 pt = g.p[t];
                const float dx = qx - pt.x, dy = qy - pt.y, dz = qz - pt.z;
                float dist = dx * dx;
                dist += dy * dy;
                dist += dz * dz;
                k0[u] = (static_cast<unsigned long long>(__float_as_uint(dist)) << 32) | static_cast<unsigned>(__float_as_int(pt.w));
                p0[u] = t;
            }
        }
        for (int round = 0; round < 5; ++round) {
            unsigned long long mine = k0[0];
            int mu = 0;
#pragma unroll
            for (int u = 1; u < 4; ++u)
                if (k0[u] < mine) mine = k0[u], mu = u;
            const unsigned long long best = wave_min_u64(mine);
            if (best == ~0ull) break;
            const uint64_t own = __ballot(mine == best);
            const int owner = __ffsll(static_cast<long long>(own)) - 1;
            int pm = p0[0];  // compile-time indices: the candidate list stays in registers
#pragma unroll
            for (int u = 1; u < 4; ++u)
                if (mu == u) pm = p0[u];
            const int pos = __shfl(pm, owner, 64);
            if (lane == owner) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u == mu) k0[u] = ~0ull;
            }
            if (!(best < bk[4])) break;  // chunk keys come out ascending: nothing smaller is left in this chunk
            // insert (best, pos) into the running ascending list, branch-free and statically indexed
            const bool c0 = best < bk[0], c1 = best < bk[1], c2 = best < bk[2], c3 = best < bk[3];
            bk[4] = c3 ? bk[3] : best, bp[4] = c3 ? bp[3] : pos;
            bk[3] = c2 ? bk[2] : (c3 ? best : bk[3]), bp[3] = c2 ? bp[2] : (c3 ? pos : bp[3]);
            bk[2] = c1 ? bk[1] : (c2 ? best : bk[2]), bp[2] = c1 ? bp[1] : (c2 ? pos : bp[2]);
            bk[1] = c0 ? bk[0] : (c1 ? best : bk[1]), bp[1] = c0 ? bp[0] : (c1 ? pos : bp[1]);
            bk[0] = c0 ? best : bk[0], bp[0] = c0 ? pos : bp[0];
        }
    }
}

// symmetric 3x3 eigen-decomposition by cyclic Jacobi (stands in for Eigen::SelfAdjointEigenSolver, :606);
// returns the largest eigenvalue's unit eigenvector and the two largest eigenvalues
__device__ __forceinline__ void eig3_largest(double a00, double a01, double a02, double a11, double a12, double a22, double* w1, double* w2,
                                             double* dir) {
    double a[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-40 * diag || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    // ascending order of the diagonal; ties keep the lower index first (as a stable sort of three)
    int o0 = 0, o1 = 1, o2 = 2;
    if (a[o1][o1] < a[o0][o0]) { int t = o0; o0 = o1; o1 = t; }
    if (a[o2][o2] < a[o1][o1]) { int t = o1; o1 = o2; o2 = t; }
    if (a[o1][o1] < a[o0][o0]) { int t = o0; o0 = o1; o1 = t; }
    *w1 = a[o1][o1];
    *w2 = a[o2][o2];
    double n = sqrt(v[0][o2] * v[0][o2] + v[1][o2] * v[1][o2] + v[2][o2] * v[2][o2]);
    dir[0] = v[0][o2] / n, dir[1] = v[1][o2] / n, dir[2] = v[2][o2] / n;
}

// least squares A n = b for a 5x3 A by column-pivoted Householder QR (stands in for colPivHouseholderQr().solve, :664)
__device__ __forceinline__ void colpiv_qr_5x3(double A[5][3], double b[5], double x[3]) {
    int perm[3] = {0, 1, 2};
    double rdiag[3] = {0, 0, 0};
    double maxpivot = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int best = k;
        double bestn = -1;
        for (int j = k; j < 3; ++j) {
            double s = 0;
            for (int i = k; i < 5; ++i) s += A[i][j] * A[i][j];
            if (s > bestn) bestn = s, best = j;
        }
        if (best != k) {
            for (int i = 0; i < 5; ++i) {
                const double t = A[i][k];
                A[i][k] = A[i][best];
                A[i][best] = t;
            }
            const int t = perm[k];
            perm[k] = perm[best];
            perm[best] = t;
        }
        double tail = 0;
        for (int i = k + 1; i < 5; ++i) tail += A[i][k] * A[i][k];
        const double c0 = A[k][k];
        double tau, beta;
        if (tail <= 2.2250738585072014e-308) {
            tau = 0, beta = c0;
            for (int i = k + 1; i < 5; ++i) A[i][k] = 0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
            for (int i = k + 1; i < 5; ++i) A[i][k] /= (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        rdiag[k] = beta;
        maxpivot = fmax(maxpivot, fabs(beta));
        if (tau != 0) {
            for (int c = k + 1; c < 3; ++c) {
                double s = A[k][c];
                for (int i = k + 1; i < 5; ++i) s += A[i][k] * A[i][c];
                s *= tau;
                A[k][c] -= s;
                for (int i = k + 1; i < 5; ++i) A[i][c] -= s * A[i][k];
            }
            double s = b[k];
            for (int i = k + 1; i < 5; ++i) s += A[i][k] * b[i];
            s *= tau;
            b[k] -= s;
            for (int i = k + 1; i < 5; ++i) b[i] -= s * A[i][k];
        }
    }
    const double thr = 2.220446049250313e-16 * 3.0 * maxpivot;
    int rank = 0;
    for (int k = 0; k < 3; ++k)
        if (fabs(rdiag[k]) > thr) ++rank;
    double y[3] = {0, 0, 0};
    for (int k = rank - 1; k >= 0; --k) {
        double s = b[k];
        for (int c = k + 1; c < rank; ++c) s -= A[k][c] * y[c];
        y[k] = s / A[k][k];
    }
    x[0] = x[1] = x[2] = 0;
    for (int k = 0; k < 3; ++k) x[perm[k]] = y[k];
}

// pointAssociateToMap (:155-164): f64 rotate + translate, stored back as f32
__device__ __forceinline__ void associate_to_map(const double* x7, float px, float py, float pz, float* o) {
    double r[3];
    quat_rotate(x7, static_cast<double>(px), static_cast<double>(py), static_cast<double>(pz), r);
    o[0] = static_cast<float>(r[0] + x7[4]);
    o[1] = static_cast<float>(r[1] + x7[5]);
    o[2] = static_cast<float>(r[2] + x7[6]);
}

struct NNBuf {
    float* px;  // [5][cap] neighbour coordinates, ascending (distance, map index)
    float* py;
    float* pz;
    float* d5;  // [cap] squared distance of the 5th neighbour
    int cap;
};

// slots [0, n_corner_stack): edge candidates; [n_corner_stack, n_corner_stack + n_surf_stack): plane candidates
// k_assoc_knn: one WAVE per stack point (the neighbour search is a handful of dependent memory round trips, so it wants
// many waves in flight); k_assoc_fit: one THREAD per stack point (the PCA / plane fit is ~3k dependent f64 operations,
// so it wants every lane busy with a different point).
__global__ void __launch_bounds__(256) k_assoc_knn(CSoA4 cs, CSoA4 ss, MapParams mp, const int* __restrict__ ccnt, const int* __restrict__ cstart,
                                                   GridPts cg, const int* __restrict__ scnt, const int* __restrict__ sstart, GridPts sg,
                                                   const LMState* __restrict__ st, const MapCounters* __restrict__ C, NNBuf nb) {
    if (!C->solve_on) return;
    const int nc = C->n_corner_stack, ns = C->n_surf_stack;
    const int n = min(nc + ns, nb.cap);
    double x7[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) x7[k] = st->x[k];
    const int lane = lane_id();
    // bounded grid, wave-stride loop over the slots: a grid sized for the capacity would be ~95 % empty workgroups
    for (int i = blockIdx.x * 4 + wave_id(); i < n; i += gridDim.x * 4) {
        const bool is_edge = i < nc;
        const int j = is_edge ? i : i - nc;
        const float ox = is_edge ? cs.x[j] : ss.x[j], oy = is_edge ? cs.y[j] : ss.y[j], oz = is_edge ? cs.z[j] : ss.z[j];
        float sel[3];
        associate_to_map(x7, ox, oy, oz, sel);
        unsigned long long bk[5];
        int bp[5];
        const GridPts& g = is_edge ? cg : sg;
        if (is_edge)
            knn5_wave(mp, ccnt, cstart, cg, sel[0], sel[1], sel[2], bk, bp);
        else
            knn5_wave(mp, scnt, sstart, sg, sel[0], sel[1], sel[2], bk, bp);
        // lane k < 5 fetches and stores neighbour k (statically indexed selects: the lists stay in registers)
        int mine = bp[0];
#pragma unroll
        for (int k = 1; k < 5; ++k)
            if (lane == k) mine = bp[k];
        if (lane < 5) {
            const bool have = mine >= 0;
            const float4 pt = have ? g.p[mine] : make_float4(0.f, 0.f, 0.f, 0.f);
            nb.px[lane * nb.cap + i] = pt.x;
            nb.py[lane * nb.cap + i] = pt.y;
            nb.pz[lane * nb.cap + i] = pt.z;
        }
        if (lane == 0) nb.d5[i] = bp[4] >= 0 ? __uint_as_float(static_cast<unsigned>(bk[4] >> 32)) : 3.4e38f;
    }
}

__global__ void __launch_bounds__(64) k_assoc_fit(CSoA4 cs, CSoA4 ss, NNBuf nb, MapCounters* C, int outer, FactorSoA f) {
    if (!C->solve_on) return;
    const int nc = C->n_corner_stack, ns = C->n_surf_stack;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int valid = 0;
    bool is_edge = false;
    if (i < nc + ns && i < f.cap) {
        is_edge = i < nc;
        const int j = is_edge ? i : i - nc;
        const float ox = is_edge ? cs.x[j] : ss.x[j], oy = is_edge ? cs.y[j] : ss.y[j], oz = is_edge ? cs.z[j] : ss.z[j];
        double pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
        if (static_cast<double>(nb.d5[i]) < 1.0) {  // :585 / :653
            float px[5], py[5], pz[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) px[k] = nb.px[k * nb.cap + i], py[k] = nb.py[k * nb.cap + i], pz[k] = nb.pz[k * nb.cap + i];
            if (is_edge) {
                double cx = 0, cy = 0, cz = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k) cx = cx + px[k], cy = cy + py[k], cz = cz + pz[k];  // :594
                cx = cx / 5.0, cy = cy / 5.0, cz = cz / 5.0;
                double m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k) {  // raw scatter sum, not divided (:599-604)
                    const double zx = px[k] - cx, zy = py[k] - cy, zz = pz[k] - cz;
                    m00 = m00 + zx * zx, m01 = m01 + zx * zy, m02 = m02 + zx * zz;
                    m11 = m11 + zy * zy, m12 = m12 + zy * zz, m22 = m22 + zz * zz;
                }
                double w1, w2, dir[3];
                eig3_largest(m00, m01, m02, m11, m12, m22, &w1, &w2, dir);
                if (w2 > 3 * w1) {  // :612
                    valid = 1;
                    pa[0] = 0.1 * dir[0] + cx, pa[1] = 0.1 * dir[1] + cy, pa[2] = 0.1 * dir[2] + cz;     // :616
                    pb[0] = -0.1 * dir[0] + cx, pb[1] = -0.1 * dir[1] + cy, pb[2] = -0.1 * dir[2] + cz;  // :617
                }
            } else {
                double A[5][3], b[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) A[k][0] = px[k], A[k][1] = py[k], A[k][2] = pz[k], b[k] = -1.0;
                double nv[3];
                colpiv_qr_5x3(A, b, nv);
                const double nrm = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
                const double d = 1 / nrm;  // negative_OA_dot_norm (:665)
                nv[0] /= nrm, nv[1] /= nrm, nv[2] /= nrm;
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if (fabs(nv[0] * px[k] + nv[1] * py[k] + nv[2] * pz[k] + d) > 0.2) ok = false;  // :673-676
                if (ok && nrm == nrm) {
                    valid = 1;
                    pa[0] = nv[0], pa[1] = nv[1], pa[2] = nv[2];
                    pb[0] = d;
                }
            }
        }
        f.valid[i] = valid;
        f.kind[i] = is_edge ? 0 : 2;
        f.cp[i] = ox, f.cp[f.cap + i] = oy, f.cp[2 * f.cap + i] = oz;
        f.pa[i] = pa[0], f.pa[f.cap + i] = pa[1], f.pa[2 * f.cap + i] = pa[2];
        f.pb[i] = pb[0], f.pb[f.cap + i] = pb[1], f.pb[2 * f.cap + i] = pb[2];
    }
}

__global__ void k_keep_error(const VoxMeta* m, MapCounters* C) {
    if (m->error) C->error = m->error;
}
// after both stack filters: propagate their verdict, fix the number of residual-block slots
__global__ void k_after_stack(const VoxMeta* m, MapCounters* C, int cap) {
    if (m->error) C->error = m->error;
    C->n_slots = min(C->n_corner_stack + C->n_surf_stack, cap);
}

// ---------------------------------------------------------------------------------------------- insert + re-filter
// sort key of a map point: [slot 9: 0..74 valid cube, 127 = not re-filtered][vz 9][vy 9][vx 9]; ~0 = outside the cube window
__device__ __forceinline__ unsigned long long map_key(const MapParams& mp, float inv_leaf, float x, float y, float z, int pc, MapCounters* C) {
    int ai, aj, ak;
    unpack_cube(pc, ai, aj, ak);
    const int I = ai + mp.cenW, J = aj + mp.cenH, K = ak + mp.cenD;
    if (I < 0 || I >= CW || J < 0 || J >= CH || K < 0 || K >= CD) return ~0ull;  // cleared slab (:346-347 ...) or rejected insert (:753-759)
    if (abs(I - mp.cI) <= 2 && abs(J - mp.cJ) <= 2 && abs(K - mp.cK) <= 1) {
        // slot of the cube inside the 5x5x3 valid set, then voxel coordinates relative to one cell below the cube's
        // lower face; lexicographic (vz,vy,vx) order equals PCL's idx order inside the cube
        const unsigned long long slot = static_cast<unsigned long long>((I - mp.cI + 2) + 5 * (J - mp.cJ + 2) + 25 * (K - mp.cK + 1));
        const int bx = static_cast<int>(floorf((50.0f * ai - 26.0f) * inv_leaf));
        const int by = static_cast<int>(floorf((50.0f * aj - 26.0f) * inv_leaf));
        const int bz = static_cast<int>(floorf((50.0f * ak - 26.0f) * inv_leaf));
        int vx = static_cast<int>(floorf(x * inv_leaf)) - bx, vy = static_cast<int>(floorf(y * inv_leaf)) - by,
            vz = static_cast<int>(floorf(z * inv_leaf)) - bz;
        if (vx < 0 || vx > 511 || vy < 0 || vy > 511 || vz < 0 || vz > 511) C->error = SCAL_E_CAPACITY;
        vx = min(max(vx, 0), 511), vy = min(max(vy, 0), 511), vz = min(max(vz, 0), 511);
        return (slot << 27) | (static_cast<unsigned long long>(vz) << 18) | (static_cast<unsigned long long>(vy) << 9) | static_cast<unsigned long long>(vx);
    }
    return MAP_NOMERGE << 27;  // cube not re-filtered this scan: keep every point; one shared key + stable sort = arrival order
}

// appends the stack points (map frame, final pose) behind the old map points and builds the sort keys
//   key layout (36 sorted bits = 4 passes): [slot 9: 0..74 valid cube, 127 = not re-filtered][vz 9][vy 9][vx 9]; dropped points get ~0
__global__ void __launch_bounds__(256) k_insert_keys(MapCloud m, int n_old, CSoA4 stack, const int* __restrict__ d_nstack, const LMState* __restrict__ st,
                                                     MapParams mp, float inv_leaf, int cap, unsigned long long* __restrict__ keys,
                                                     int* __restrict__ vals, MapCounters* C, int cls) {
    const int ns = *d_nstack;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        C->n_total[cls] = min(n_old + ns, cap);
        if (n_old + ns > cap) C->error = SCAL_E_CAPACITY;
    }
    if (i >= n_old + ns || i >= cap) return;
    float x, y, z;
    int pc;
    if (i < n_old) {
        x = m.x[i], y = m.y[i], z = m.z[i], pc = m.cube[i];
    } else {
        const int j = i - n_old;
        double x7[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) x7[k] = st->x[k];
        float sel[3];
        associate_to_map(x7, stack.x[j], stack.y[j], stack.z[j], sel);  // :740 / :764
        x = sel[0], y = sel[1], z = sel[2];
        pc = pack_cube(cube_abs(x) , cube_abs(y), cube_abs(z));
        // cube_abs already excludes cen; see unpack below
        m.x[i] = x, m.y[i] = y, m.z[i] = z, m.w[i] = stack.w[j], m.cube[i] = pc;
    }
    const unsigned long long k = map_key(mp, inv_leaf, x, y, z, pc, C);
    keys[i] = k;
    vals[i] = i;
}

__global__ void __launch_bounds__(256) k_map_heads(SortedPairs sp, const int* __restrict__ d_n, int* __restrict__ blockcnt) {
    const unsigned long long* keys = sp.keys[sorted_sel(sp)];
    const int n = *d_n;
    const int nb = (n + 255) / 256;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    __shared__ int s[17];
    const int i = blockIdx.x * 256 + threadIdx.x;
    int head = 0;
    if (i < n) {
        const unsigned long long k = keys[i];
        head = (k != ~0ull) && (((k >> 27) == MAP_NOMERGE) || i == 0 || keys[i - 1] != k);
    }
    int total;
    block_exclusive_scan(head, s, &total);
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_map_reduce(SortedPairs sp, const int* __restrict__ d_n, const int* __restrict__ blockoff, MapCloud in,
                                                    MapCloud out) {
    const unsigned long long* keys = sp.keys[sorted_sel(sp)];
    const int* vals = sp.vals[sorted_sel(sp)];
    const int n = *d_n;
    const int nb = (n + 255) / 256;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    __shared__ int s[17];
    const int i = blockIdx.x * 256 + threadIdx.x;
    int head = 0;
    unsigned long long k = 0;
    if (i < n) {
        k = keys[i];
        head = (k != ~0ull) && (((k >> 27) == MAP_NOMERGE) || i == 0 || keys[i - 1] != k);
    }
    int total;
    const int rank = block_exclusive_scan(head, s, &total);
    if (!head) return;
    const int o = blockoff[blockIdx.x] + rank;
    if ((k >> 27) == MAP_NOMERGE) {
        const int g = vals[i];
        out.x[o] = in.x[g], out.y[o] = in.y[g], out.z[o] = in.z[g], out.w[o] = in.w[g], out.cube[o] = in.cube[g];
        return;
    }
    float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
    int u = i;
    while (u < n && keys[u] == k) {
        const int g = vals[u];
        ax += in.x[g], ay += in.y[g], az += in.z[g], aw += in.w[g];
        ++u;
    }
    const float c = static_cast<float>(u - i);
    out.x[o] = ax / c, out.y[o] = ay / c, out.z[o] = az / c, out.w[o] = aw / c;
    out.cube[o] = in.cube[vals[i]];
}

// ---------------------------------------------------------------------------------------------- merge insert
// The re-filter of :738-802 without sorting the map.  After a re-filter every valid cube holds one point per voxel in key
// order, and re-filtering such points alone returns them unchanged ((0 + x) / 1 in f32).  So when the cube window did not
// move since the previous scan, only the scan's new points need sorting (<= MERGE_MAX, one workgroup); they are then merged
// into the old sequence: a new voxel run whose key equals an old point's joins that point's centroid (old point first, then
// the new points in arrival order - exactly the order the stable full sort would give), the other runs are inserted.
// Anything unusual - old keys not strictly increasing inside a valid cube (a centroid rounded across a voxel face, a cube
// that collected unfiltered points while it was outside the 5x5x3 window), an old point outside the window, too many new
// points - raises MapCounters::merge_fail and the host redoes the insertion with the full sort.
constexpr int MERGE_MAX = 8192;
constexpr int MERGE_IDX_BITS = 13;
constexpr int MERGE_SAMPLES = 4096;  // old keys staged in LDS for the two-level lookup

struct MergeNew {            // per class, MERGE_MAX entries
    float *x, *y, *z, *w;    // new points in the map frame, arrival order
    int* cube;
    unsigned long long* key;     // their map keys
    unsigned long long* sorted;  // (key << 13 | arrival index), ascending; entries >= n_eff are ~0
    int* pre;                    // [MERGE_MAX + 1] inserted (unmatched) runs that start before sorted position t
    int* lb;                     // run heads: number of old points in front of the run
    unsigned char* hm;           // bit 0: run head, bit 1: run joins an old point
};
struct MergeArgs {
    MapCloud in[2], out[2];
    int n_old[2];
    CSoA4 stack[2];
    const int* d_ns[2];
    float inv_leaf[2];
    unsigned long long* okeys[2];  // keys of the old points
    MergeNew nw[2];
    int nbo[2];                    // blocks over the old points
    int cap;
    int* grid_cnt[2];              // cell counters of the neighbour grid: restored to zero here (saves a launch)
    const int* grid_rank[2];
};
constexpr int MERGE_NEW_BLOCKS = MERGE_MAX / 256;

__device__ __forceinline__ bool key_nomerge(unsigned long long k) { return (k >> 27) == MAP_NOMERGE; }

__global__ void __launch_bounds__(256) k_merge_keys(MergeArgs a, const LMState* __restrict__ st, MapParams mp, MapCounters* C) {
    int b = blockIdx.x;
    int cls, part;  // part 0: old points, 1: new points
    if (b < a.nbo[0]) cls = 0, part = 0;
    else if ((b -= a.nbo[0]) < a.nbo[1]) cls = 1, part = 0;
    else if ((b -= a.nbo[1]) < MERGE_NEW_BLOCKS) cls = 0, part = 1;
    else b -= MERGE_NEW_BLOCKS, cls = 1, part = 1;
    const int i = b * 256 + threadIdx.x;
    const MapCloud m = a.in[cls];
    if (part == 0) {
        if (i >= a.n_old[cls]) return;
        const unsigned long long k = map_key(mp, a.inv_leaf[cls], m.x[i], m.y[i], m.z[i], m.cube[i], C);
        a.okeys[cls][i] = k;
        if (a.grid_rank[cls][i] >= 0) a.grid_cnt[cls][grid_cell(mp, m.x[i], m.y[i], m.z[i])] = 0;  // zero invariant of the cell grid
        bool bad = k == ~0ull;
        if (i > 0) {
            const unsigned long long kp = map_key(mp, a.inv_leaf[cls], m.x[i - 1], m.y[i - 1], m.z[i - 1], m.cube[i - 1], C);
            bad |= kp > k || (kp == k && !key_nomerge(k));
        }
        if (bad) C->merge_fail = 1;
        return;
    }
    const int ns = *a.d_ns[cls];
    if (i == 0 && (ns > MERGE_MAX || a.n_old[cls] + ns > a.cap)) C->merge_fail = 1;
    if (i >= min(ns, MERGE_MAX)) return;
    double x7[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) x7[k] = st->x[k];
    float sel[3];
    const CSoA4 stack = a.stack[cls];
    associate_to_map(x7, stack.x[i], stack.y[i], stack.z[i], sel);  // :740 / :764
    const int pc = pack_cube(cube_abs(sel[0]), cube_abs(sel[1]), cube_abs(sel[2]));
    const MergeNew nw = a.nw[cls];
    nw.x[i] = sel[0], nw.y[i] = sel[1], nw.z[i] = sel[2], nw.w[i] = stack.w[i], nw.cube[i] = pc;
    nw.key[i] = map_key(mp, a.inv_leaf[cls], sel[0], sel[1], sel[2], pc, C);
}

// one workgroup per class: sort the new keys, find the run heads, look each head up among the old keys, count the inserts
__global__ void __launch_bounds__(1024) k_merge_prepare(MergeArgs a, MapCounters* C) {
    extern __shared__ __align__(16) unsigned long long sk[];  // MERGE_MAX keys
    __shared__ unsigned char s_hm[MERGE_MAX];
    __shared__ unsigned long long s_samp[MERGE_SAMPLES];
    __shared__ int s_scan[17];
    const int cls = blockIdx.x;
    const MergeNew nw = a.nw[cls];
    const int tid = threadIdx.x;
    const int n_new = min(*a.d_ns[cls], MERGE_MAX);
    const int n_old = a.n_old[cls];
    const unsigned long long* okeys = a.okeys[cls];
    const int np2 = max(512, next_pow2(n_new));
    if (cls == 1 && tid == 0) SCAL_STAMP(26);
    int mine = 0;
    for (int t = tid; t < np2; t += 1024) {
        unsigned long long k = ~0ull;
        if (t < n_new) {
            const unsigned long long kk = nw.key[t];
            if (kk != ~0ull) k = (kk << MERGE_IDX_BITS) | static_cast<unsigned long long>(t), ++mine;
        }
        sk[t] = k;
    }
    int n_eff = 0;
    block_exclusive_scan(mine, s_scan, &n_eff);  // also the barrier after the fill
    if (cls == 1 && tid == 0) SCAL_STAMP(27);
    block_sort_u64(sk, np2, n_new);
    if (cls == 1 && tid == 0) SCAL_STAMP(28);
    // heads + lookups; element t = e * 1024 + tid, so the eight binary searches of a thread advance in lock step
    constexpr int PER = MERGE_MAX / 1024;
    unsigned long long key[PER];
    int lo[PER], hi[PER];
    bool head[PER], nom[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int t = e * 1024 + tid;
        head[e] = false, nom[e] = false, key[e] = 0, lo[e] = 0, hi[e] = 0;
        if (t < n_eff) {
            key[e] = sk[t] >> MERGE_IDX_BITS;
            nom[e] = key_nomerge(key[e]);
            head[e] = nom[e] || t == 0 || (sk[t - 1] >> MERGE_IDX_BITS) != key[e];
            if (head[e] && !nom[e]) hi[e] = n_old;       // lower bound among the old keys
            if (nom[e]) lo[e] = hi[e] = n_old;           // behind every old point (NOMERGE is the largest old key)
        }
    }
    if (cls == 1 && tid == 0) SCAL_STAMP(29);
    // Two-level lower bound: every `stride`-th old key is staged in LDS (one coalesced pass), the search over the samples
    // runs at LDS latency and leaves a window of <= stride old keys for the last few global steps.
    int stride = 32;
    while ((n_old + stride - 1) / stride > MERGE_SAMPLES) stride <<= 1;
    const int n_samp = (n_old + stride - 1) / stride;
    for (int j = tid; j < n_samp; j += 1024) s_samp[j] = okeys[static_cast<size_t>(j) * stride];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        if (lo[e] < hi[e]) {  // heads that need a lookup
            int a = 0, b = n_samp;  // number of samples < key
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (s_samp[mid] < key[e]) a = mid + 1;
                else b = mid;
            }
            lo[e] = a > 0 ? (a - 1) * stride + 1 : 0;
            hi[e] = min(n_old, a * stride);
            if (lo[e] > hi[e]) lo[e] = hi[e];
        }
    }
    for (int step = 0; step < 32; ++step) {
        bool any = false;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            if (lo[e] < hi[e]) {
                const int mid = (lo[e] + hi[e]) >> 1;
                if (okeys[mid] < key[e]) lo[e] = mid + 1;
                else hi[e] = mid;
                any = true;
            }
        }
        if (!any) break;
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const int t = e * 1024 + tid;
        if (t < n_eff) {
            const bool matched = head[e] && !nom[e] && lo[e] < n_old && okeys[lo[e]] == key[e];
            s_hm[t] = (head[e] ? 1 : 0) | (matched ? 2 : 0);
            nw.lb[t] = lo[e];
        }
    }
    __syncthreads();
    if (cls == 1 && tid == 0) SCAL_STAMP(30);
    // inserted runs in front of every sorted position
    const int c0 = min(n_eff, tid * PER), c1 = min(n_eff, c0 + PER);
    int ins = 0;
    for (int t = c0; t < c1; ++t) ins += (s_hm[t] & 3) == 1;
    int total = 0;
    int run = block_exclusive_scan(ins, s_scan, &total);
    for (int t = c0; t < c1; ++t) {
        nw.pre[t] = run;
        run += (s_hm[t] & 3) == 1;
        nw.hm[t] = s_hm[t];
        nw.sorted[t] = sk[t];
    }
    if (cls == 1 && tid == 0) SCAL_STAMP(31);
    if (tid == 0) {
        nw.pre[n_eff] = total;
        C->merge_neff[cls] = n_eff;
        C->n_map_new[cls] = n_old + total;
        if (n_old + total > a.cap) C->merge_fail = 1;
    }
}

__global__ void __launch_bounds__(256) k_merge_write(MergeArgs a, const MapCounters* __restrict__ C) {
    int b = blockIdx.x;
    int cls, part;
    if (b < a.nbo[0]) cls = 0, part = 0;
    else if ((b -= a.nbo[0]) < a.nbo[1]) cls = 1, part = 0;
    else if ((b -= a.nbo[1]) < MERGE_NEW_BLOCKS) cls = 0, part = 1;
    else b -= MERGE_NEW_BLOCKS, cls = 1, part = 1;
    const int i = b * 256 + threadIdx.x;
    const MapCloud in = a.in[cls], out = a.out[cls];
    const MergeNew nw = a.nw[cls];
    const int n_eff = C->merge_neff[cls];
    const int n_old = a.n_old[cls];
    if (part == 0) {
        if (i >= n_old) return;
        const unsigned long long k = a.okeys[cls][i];
        int lo = 0, hi = n_eff;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((nw.sorted[mid] >> MERGE_IDX_BITS) < k) lo = mid + 1;
            else hi = mid;
        }
        const int o = min(i + nw.pre[lo], a.cap - 1);
        // (0 + x) / 1: what the re-filter computes for a voxel holding this point alone
        float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
        ax += in.x[i], ay += in.y[i], az += in.z[i], aw += in.w[i];
        int cnt = 1;
        if (!key_nomerge(k)) {
            int u = lo;
            while (u < n_eff && (nw.sorted[u] >> MERGE_IDX_BITS) == k) {  // new points of the same voxel, arrival order
                const int j = static_cast<int>(nw.sorted[u] & (MERGE_MAX - 1));
                ax += nw.x[j], ay += nw.y[j], az += nw.z[j], aw += nw.w[j];
                ++u, ++cnt;
            }
        }
        const float c = static_cast<float>(cnt);
        out.x[o] = ax / c, out.y[o] = ay / c, out.z[o] = az / c, out.w[o] = aw / c, out.cube[o] = in.cube[i];
        return;
    }
    if (i >= n_eff || (nw.hm[i] & 3) != 1) return;  // only heads of inserted runs
    const unsigned long long k = nw.sorted[i] >> MERGE_IDX_BITS;
    const int o = min(nw.lb[i] + nw.pre[i], a.cap - 1);
    float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
    int u = i;
    do {
        const int j = static_cast<int>(nw.sorted[u] & (MERGE_MAX - 1));
        ax += nw.x[j], ay += nw.y[j], az += nw.z[j], aw += nw.w[j];
        ++u;
    } while (!key_nomerge(k) && u < n_eff && (nw.sorted[u] >> MERGE_IDX_BITS) == k);
    const float c = static_cast<float>(u - i);
    out.x[o] = ax / c, out.y[o] = ay / c, out.z[o] = az / c, out.w[o] = aw / c;
    out.cube[o] = nw.cube[static_cast<int>(nw.sorted[i] & (MERGE_MAX - 1))];
}

__global__ void __launch_bounds__(256) k_transform_cloud(CSoA4 in, const int* __restrict__ d_n, int cap, const LMState* __restrict__ st, SoA4 out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*d_n, cap)) return;
    double x7[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) x7[k] = st->x[k];
    float sel[3];
    associate_to_map(x7, in.x[i], in.y[i], in.z[i], sel);
    out.x[i] = sel[0], out.y[i] = sel[1], out.z[i] = sel[2], out.w[i] = in.w[i];
}

// Start of a device-resident step in ONE launch: clears the per-scan counters and copies the lessSharp cloud (xyzi records)
// and the lessFlat cloud (SoA) of a features context into corner_in / surf_in.  Blocks [0,nbc) corner, the rest surf.
__global__ void __launch_bounds__(256) k_map_gather(const float* __restrict__ less_aos, const int* __restrict__ d_n_less, CSoA4 less_flat,
                                                    const int* __restrict__ d_n_less_flat, SoA4 corner_in, SoA4 surf_in, MapCounters* C, int cap,
                                                    int nbc) {
    const bool corner = static_cast<int>(blockIdx.x) < nbc;
    const int b = corner ? blockIdx.x : blockIdx.x - nbc;
    const int n = min(corner ? *d_n_less : *d_n_less_flat, cap);
    const int i = b * 256 + threadIdx.x;
    if (blockIdx.x == 0) {  // every counter except the two input counts, which have exactly one writer each below
        int* w = reinterpret_cast<int*>(C);
        for (int t = 2 + threadIdx.x; t < static_cast<int>(sizeof(MapCounters) / sizeof(int)); t += 256) w[t] = 0;
    }
    if (i == 0) (corner ? C->n_corner_in : C->n_surf_in) = n;
    if (i >= n) return;
    if (corner) {
        const float4 p = reinterpret_cast<const float4*>(less_aos)[i];
        corner_in.x[i] = p.x, corner_in.y[i] = p.y, corner_in.z[i] = p.z, corner_in.w[i] = p.w;
    } else {
        surf_in.x[i] = less_flat.x[i], surf_in.y[i] = less_flat.y[i], surf_in.z[i] = less_flat.z[i], surf_in.w[i] = less_flat.w[i];
    }
}

__global__ void __launch_bounds__(256) k_export_valid(MapCloud m, int n, MapParams mp, int* __restrict__ counter, float* __restrict__ out, int cap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && cube_valid(mp, m.cube[i])) {
        const int p = atomicAdd(counter, 1);
        if (p < cap) reinterpret_cast<float4*>(out)[p] = make_float4(m.x[i], m.y[i], m.z[i], m.w[i]);
    }
}

struct SoAStore {
    DevBuf<float> x, y, z, w;
    int alloc(size_t n) {
        SCAL_TRY(x.alloc(n));
        SCAL_TRY(y.alloc(n));
        SCAL_TRY(z.alloc(n));
        SCAL_TRY(w.alloc(n));
        return SCAL_OK;
    }
    SoA4 v() { return SoA4{x.p, y.p, z.p, w.p}; }
    CSoA4 cv() const { return CSoA4{x.p, y.p, z.p, w.p}; }
};

struct MapStore {
    SoAStore pts[2];  // double buffered
    DevBuf<int> cube[2];
    int cur = 0;
    int n = 0;  // host copy of the point count
    MapCloud cloud(int b) { return MapCloud{pts[b].x.p, pts[b].y.p, pts[b].z.p, pts[b].w.p, cube[b].p}; }
};

struct GridStore {
    DevBuf<int> cnt, start, rank;
    DevBuf<float4> g;
    GridPts pts() { return GridPts{g.p}; }
};

SCAL_DEFINE_STAMP_READER(scal_debug_stamps_map)
}  // namespace scal

using namespace scal;

struct scal_map {
    scal_map_config cfg;
    int lane = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;          // side stream for scal_map_prefetch_features (lazily acquired)
    hipEvent_t ev = nullptr;
    // prefetches queued ahead of their step (at most two: the bench queues scan k+1's before scan k's step starts)
    static constexpr int NSETS = 3;
    struct Prefetch {
        scal_features_t* feat;
        int set;
    };
    std::mutex pf_mu;  // prefetches may come from a second host thread
    Prefetch pf[2];
    int n_pf = 0;
    hipEvent_t ev_pre[NSETS] = {};
    // asynchronous step: pose first (ev_pose), insertion + registration behind it (ev_done)
    hipEvent_t ev_pose = nullptr, ev_done = nullptr;
    bool pose_pending = false, insert_pending = false, insert_try_merge = false, pending_prefetched = false;
    double pend_q_wodom[4] = {0, 0, 0, 1}, pend_t_wodom[3] = {0, 0, 0};
    MapParams pend_mp{};
    int deferred_error = SCAL_OK;
    int scan_cap = 0, map_cap = 0, slot_cap = 0;
    // host-side pose state (laserMapping.cpp:110-120)
    double q_wmap_wodom[4] = {0, 0, 0, 1}, t_wmap_wodom[3] = {0, 0, 0};
    int cenW = 10, cenH = 10, cenD = 5;  // :74-76
    MapParams last_mp{};
    bool have_mp = false;
    // device
    DevBuf<float> aos;  // upload staging
    // Per-step inputs live in a ring of three "sets": the side stream may gather + downsample scans k+1 and k+2 while scan k's
    // association and insertion still read theirs; a step takes the set its prefetch filled (or the next free one).
    int set = 0;
    SoAStore corner_in2[NSETS], surf_in2[NSETS], corner_stack2[NSETS], surf_stack2[NSETS];
    SoAStore full_in, full_out;
    SoAStore& corner_in(int st = -1) { return corner_in2[st < 0 ? set : st]; }
    SoAStore& surf_in(int st = -1) { return surf_in2[st < 0 ? set : st]; }
    SoAStore& corner_stack(int st = -1) { return corner_stack2[st < 0 ? set : st]; }
    SoAStore& surf_stack(int st = -1) { return surf_stack2[st < 0 ? set : st]; }
    VoxelFilter vf;
    MapStore map[2];  // corner, surf
    GridStore grid[2];
    RadixSort sorter;
    DevBuf<unsigned long long> keys;
    DevBuf<int> vals, blockcnt;
    // merge insert
    bool merge_insert = true;
    int last_insert_path = 0;  // 0 full sort, 1 merge
    SoAStore mnew[2];
    DevBuf<int> mcube[2], mpre[2], mlb[2];
    DevBuf<unsigned long long> mkey[2], msorted[2];
    DevBuf<unsigned char> mhm[2];
    MergeNew merge_new(int k) {
        return MergeNew{mnew[k].x.p, mnew[k].y.p, mnew[k].z.p, mnew[k].w.p, mcube[k].p, mkey[k].p, msorted[k].p, mpre[k].p, mlb[k].p, mhm[k].p};
    }
    DevBuf<int> fvalid, fkind;
    DevBuf<double> fcp, fpa, fpb, partials;
    DevBuf<LMSync> lm_sync;
    DevBuf<float> nnx, nny, nnz, nnd5;
    NNBuf nnbuf() { return NNBuf{nnx.p, nny.p, nnz.p, nnd5.p, slot_cap}; }
    DevBuf<LMState> d_st;
    DevBuf<MapCounters> d_C2[NSETS];
    DevBuf<MapCounters>& d_C(int st = -1) { return d_C2[st < 0 ? set : st]; }
    DevBuf<double> d_x0;
    DevBuf<int> d_nfull;
    PinBuf<MapCounters> h_C, h_C1;  // counters at the end of the step / when the pose is ready
    PinBuf<int> h_misc;
    PinBuf<double> h_x0;
    PinBuf<LMState> h_st;
    FactorSoA factors() { return FactorSoA{fvalid.p, fkind.p, fcp.p, fpa.p, fpb.p, slot_cap}; }
};

extern "C" int scal_map_create(const scal_map_config* cfg, scal_map_t** out) {
    if (!cfg || !out || cfg->max_scan_points <= 0 || cfg->max_map_points <= 0 || !(cfg->line_res > 0) || !(cfg->plane_res > 0)) {
        set_error("scal_map_create: bad argument");
        return SCAL_E_ARG;
    }
    if (cfg->line_res < 0.11f || cfg->plane_res < 0.11f) {
        set_error("mapping resolutions below 0.11 m exceed the 512 voxel/cube key layout");
        return SCAL_E_ARG;
    }
    *out = nullptr;
    SCAL_TRY(select_device(cfg->device));
    auto* c = new scal_map();
    c->cfg = *cfg;
    c->scan_cap = cfg->max_scan_points;
    c->map_cap = cfg->max_map_points;
    c->slot_cap = cfg->max_scan_points;
    int rc = SCAL_OK;
    auto A = [&](int r) { if (rc == SCAL_OK) rc = r; };
    const size_t sc = c->scan_cap, mc = c->map_cap;
    A(c->aos.alloc(sc * 4));
    for (int k = 0; k < scal_map::NSETS; ++k) {
        A(c->corner_in2[k].alloc(sc)); A(c->surf_in2[k].alloc(sc)); A(c->corner_stack2[k].alloc(sc)); A(c->surf_stack2[k].alloc(sc));
        A(c->d_C2[k].alloc(1));
    }
    A(c->full_in.alloc(sc)); A(c->full_out.alloc(sc));
    A(c->vf.init(c->scan_cap));
    for (int k = 0; k < 2; ++k) {
        for (int b = 0; b < 2; ++b) {
            A(c->map[k].pts[b].alloc(mc));
            A(c->map[k].cube[b].alloc(mc));
        }
        A(c->grid[k].cnt.alloc(GCELLS)); A(c->grid[k].start.alloc(GCELLS)); A(c->grid[k].rank.alloc(mc));
        A(c->grid[k].g.alloc(mc));
    }
    A(c->sorter.init(c->map_cap));
    A(c->keys.alloc(mc)); A(c->vals.alloc(mc)); A(c->blockcnt.alloc(div_up(c->map_cap, 256) + 1));
    for (int k = 0; k < 2; ++k) {
        A(c->mnew[k].alloc(MERGE_MAX)); A(c->mcube[k].alloc(MERGE_MAX)); A(c->mpre[k].alloc(MERGE_MAX + 1)); A(c->mlb[k].alloc(MERGE_MAX));
        A(c->mkey[k].alloc(MERGE_MAX)); A(c->msorted[k].alloc(MERGE_MAX)); A(c->mhm[k].alloc(MERGE_MAX));
    }
    A(c->fvalid.alloc(sc)); A(c->fkind.alloc(sc)); A(c->fcp.alloc(3 * sc)); A(c->fpa.alloc(3 * sc)); A(c->fpb.alloc(3 * sc));
    A(c->nnx.alloc(5 * sc)); A(c->nny.alloc(5 * sc)); A(c->nnz.alloc(5 * sc)); A(c->nnd5.alloc(sc));
    A(c->partials.alloc((size_t)2 * LM_GRID * LM_NACC));
    A(c->lm_sync.alloc(1));
    if (rc == SCAL_OK && hipMemset(c->lm_sync.p, 0, sizeof(LMSync)) != hipSuccess) rc = SCAL_E_HIP;
    A(c->d_st.alloc(1)); A(c->d_x0.alloc(8)); A(c->d_nfull.alloc(4)); A(c->h_C1.alloc(1));
    A(c->h_C.alloc(1)); A(c->h_st.alloc(1)); A(c->h_misc.alloc(4)); A(c->h_x0.alloc(8));
    c->lane = stage_lane(STAGE_MAP);
    if (rc == SCAL_OK && acquire_stream(c->cfg.device, &c->stream, c->lane) != SCAL_OK) rc = SCAL_E_HIP;
    if (rc == SCAL_OK && hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) rc = SCAL_E_HIP;
    if (rc == SCAL_OK) {
        // the cell counters obey a zero invariant: every step clears exactly the cells it touched
        for (int k = 0; k < 2 && rc == SCAL_OK; ++k) rc = c->grid[k].cnt.zero(c->stream);
        if (rc == SCAL_OK && hipMemsetAsync(c->d_st.p, 0, sizeof(LMState), c->stream) != hipSuccess) rc = SCAL_E_HIP;
        if (rc == SCAL_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = SCAL_E_HIP;
    }
    if (rc != SCAL_OK) {
        if (rc == SCAL_E_HIP) set_error("scal_map_create: HIP resource creation failed");
        delete c;
        return rc;
    }
    *out = c;
    return SCAL_OK;
}

extern "C" void scal_map_destroy(scal_map_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        release_stream(c->cfg.device, c->lane);
    }
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        release_stream(c->cfg.device, 2);
    }
    if (c->ev) (void)hipEventDestroy(c->ev);
    for (int k = 0; k < scal_map::NSETS; ++k)
        if (c->ev_pre[k]) (void)hipEventDestroy(c->ev_pre[k]);
    if (c->ev_pose) (void)hipEventDestroy(c->ev_pose);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    delete c;
}

namespace {

// Eigen-equivalent host quaternion helpers, storage (x,y,z,w)
void h_qmul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
void h_rot(const double* q, const double* v, double* o) {
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux, uy += uy, uz += uz;
    const double cx = q[1] * uz - q[2] * uy, cy = q[2] * ux - q[0] * uz, cz = q[0] * uy - q[1] * ux;
    o[0] = (v[0] + q[3] * ux) + cx, o[1] = (v[1] + q[3] * uy) + cy, o[2] = (v[2] + q[3] * uz) + cz;
}

// stack downsample (:543-551) of corner_in / surf_in into corner_stack / surf_stack; independent of the pose
static int enqueue_stack_filters(scal_map* c, hipStream_t s, int n_corner_bound, int n_surf_bound, int st) {
    MapCounters* C = c->d_C(st).p;
    SCAL_TRY(c->vf.run(s, c->corner_in(st).cv(), &C->n_corner_in, n_corner_bound, c->cfg.line_res, 36, c->corner_stack(st).v(), &C->n_corner_stack));
    if (n_corner_bound <= 8192 && n_surf_bound > 8192) {
        // the two filters share one VoxMeta: keep the small-path verdict of the corner cloud
        hipLaunchKernelGGL(k_keep_error, dim3(1), dim3(1), 0, s, c->vf.meta.p, C);
    }
    SCAL_TRY(c->vf.run(s, c->surf_in(st).cv(), &C->n_surf_in, n_surf_bound, c->cfg.plane_res, 36, c->surf_stack(st).v(), &C->n_surf_stack));
    hipLaunchKernelGGL(k_after_stack, dim3(1), dim3(1), 0, s, c->vf.meta.p, C, c->slot_cap);
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

// insert + re-filter (:738-802) by a full stable sort of the pool
static int insert_full_sort(scal_map* c, const MapParams& mp) {
    hipStream_t s = c->stream;
    MapCounters* C = c->d_C().p;
    LMState* st = c->d_st.p;
    for (int k = 0; k < 2; ++k) {
        MapStore& M = c->map[k];
        const int n_tot_max = std::min(c->map_cap, M.n + c->scan_cap);
        const int nb = std::max(1, div_up(n_tot_max, 256));
        MapCloud in = M.cloud(M.cur), outc = M.cloud(M.cur ^ 1);
        const CSoA4 stack = k == 0 ? c->corner_stack().cv() : c->surf_stack().cv();
        const int* d_ns = k == 0 ? &C->n_corner_stack : &C->n_surf_stack;
        hipLaunchKernelGGL(k_insert_keys, dim3(nb), dim3(256), 0, s, in, M.n, stack, d_ns, st, mp, k == 0 ? mp.inv_line : mp.inv_plane, c->map_cap,
                           c->keys.p, c->vals.p, C, k);
        SortedPairs sp;
        SCAL_TRY(c->sorter.sort(s, c->keys.p, c->vals.p, &C->n_total[k], n_tot_max, 36, nullptr, &sp));
        hipLaunchKernelGGL(k_map_heads, dim3(nb), dim3(256), 0, s, sp, &C->n_total[k], c->blockcnt.p);
        launch_scan_inplace(s, c->blockcnt.p, &C->n_total[k], 256, 1, &C->n_map_new[k]);
        hipLaunchKernelGGL(k_map_reduce, dim3(nb), dim3(256), 0, s, sp, &C->n_total[k], c->blockcnt.p, in, outc);
    }
    return SCAL_OK;
}

// waits for the insertion of the previous step (if any), redoes it with the full sort when the merge gave up, and publishes
// the new map sizes.  Every entry point that touches the map calls this first.
static int map_finish(scal_map* c) {
    if (!c->insert_pending) return SCAL_OK;
    c->insert_pending = false;
    hipStream_t s = c->stream;
    SCAL_HIP(hipEventSynchronize(c->ev_done));
    c->last_insert_path = c->insert_try_merge ? 1 : 0;
    if (c->insert_try_merge && c->h_C.p->merge_fail && !c->h_C.p->error) {  // a case the merge does not cover: redo with the full sort
        SCAL_TRY(insert_full_sort(c, c->pend_mp));
        SCAL_HIP(hipGetLastError());
        SCAL_HIP(hipMemcpyAsync(c->h_C.p, c->d_C().p, sizeof(MapCounters), hipMemcpyDeviceToHost, s));
        SCAL_HIP(hipStreamSynchronize(s));
        c->last_insert_path = 0;
    }
    for (int k = 0; k < 2; ++k) c->map[k].cur ^= 1;
    const MapCounters& H = *c->h_C.p;
    for (int k = 0; k < 2; ++k) c->map[k].n = H.n_map_new[k];
    if (H.error) {
        set_error("scal_map_step: device capacity exceeded (map pool of %d points per class, or a voxel outside its cube)", c->map_cap);
        return H.error;
    }
    return SCAL_OK;
}

// Enqueues one process() pass: everything after the inputs sit in corner_in / surf_in (/ full_in) with their counts in d_C.
// ev_pose fires when the optimised pose has reached the host buffers, ev_done after insertion + registration.
static int map_enqueue(scal_map* c, const double* q_wodom, const double* t_wodom, bool have_full, CSoA4 full_view, const int* d_n_full,
                       int n_corner_bound, int n_surf_bound, bool filters_done) {
    SCAL_TRY(map_finish(c));
    if (c->pose_pending) {
        set_error("scal_map: the pose of the previous step has not been collected");
        return SCAL_E_STATE;
    }
    if (!c->ev_pose) SCAL_HIP(hipEventCreateWithFlags(&c->ev_pose, hipEventDisableTiming));
    if (!c->ev_done) SCAL_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    for (int i = 0; i < 4; ++i) c->pend_q_wodom[i] = q_wodom[i];
    for (int i = 0; i < 3; ++i) c->pend_t_wodom[i] = t_wodom[i];
    hipStream_t s = c->stream;
    // transformAssociateToMap (:143-147)
    double x0[8] = {0};
    h_qmul(c->q_wmap_wodom, q_wodom, x0);
    double rt[3];
    h_rot(c->q_wmap_wodom, t_wodom, rt);
    x0[4] = rt[0] + c->t_wmap_wodom[0], x0[5] = rt[1] + c->t_wmap_wodom[1], x0[6] = rt[2] + c->t_wmap_wodom[2];
    // :313-322
    int cI = int((x0[4] + 25.0) / 50.0) + c->cenW, cJ = int((x0[5] + 25.0) / 50.0) + c->cenH, cK = int((x0[6] + 25.0) / 50.0) + c->cenD;
    if (x0[4] + 25.0 < 0) cI--;
    if (x0[5] + 25.0 < 0) cJ--;
    if (x0[6] + 25.0 < 0) cK--;
    // :324-508 — the pointer shuffles become offset updates; the cleared slabs are dropped by k_insert_keys
    while (cI < 3) cI++, c->cenW++;
    while (cI >= CW - 3) cI--, c->cenW--;
    while (cJ < 3) cJ++, c->cenH++;
    while (cJ >= CH - 3) cJ--, c->cenH--;
    while (cK < 3) cK++, c->cenD++;
    while (cK >= CD - 3) cK--, c->cenD--;
    MapParams mp;
    mp.cenW = c->cenW, mp.cenH = c->cenH, mp.cenD = c->cenD;
    mp.cI = cI, mp.cJ = cJ, mp.cK = cK;
    mp.ox = 50 * (cI - 2 - c->cenW) - 25, mp.oy = 50 * (cJ - 2 - c->cenH) - 25, mp.oz = 50 * (cK - 1 - c->cenD) - 25;
    mp.inv_line = 1.0f / c->cfg.line_res, mp.inv_plane = 1.0f / c->cfg.plane_res;
    const bool window_same = c->have_mp && c->last_mp.cenW == mp.cenW && c->last_mp.cenH == mp.cenH && c->last_mp.cenD == mp.cenD &&
                             c->last_mp.cI == mp.cI && c->last_mp.cJ == mp.cJ && c->last_mp.cK == mp.cK;
    c->last_mp = mp;
    c->have_mp = true;

    MapCounters* C = c->d_C().p;
    LMState* st = c->d_st.p;
    for (int i = 0; i < 7; ++i) c->h_x0.p[i] = x0[i];
    SCAL_HIP(hipMemcpyAsync(st->x, c->h_x0.p, sizeof(double) * 7, hipMemcpyHostToDevice, s));  // LMState::x is the first member

    if (!filters_done) SCAL_TRY(enqueue_stack_filters(c, s, n_corner_bound, n_surf_bound, c->set));

    // cell grids over the valid cubes (both classes per launch)
    GridArgs ga;
    for (int k = 0; k < 2; ++k) {
        MapStore& M = c->map[k];
        GridStore& G = c->grid[k];
        ga.m[k] = M.cloud(M.cur), ga.n[k] = M.n;
        ga.cnt[k] = G.cnt.p, ga.rank[k] = G.rank.p, ga.start[k] = G.start.p, ga.g[k] = G.pts();
    }
    ga.nb0 = std::max(1, div_up(c->map[0].n, 256));
    const int grid_blocks = ga.nb0 + std::max(1, div_up(c->map[1].n, 256));
    hipLaunchKernelGGL(k_grid_count, dim3(grid_blocks), dim3(256), 0, s, ga, mp, C);
    hipLaunchKernelGGL(k_grid_alloc, dim3(grid_blocks), dim3(256), 0, s, ga, mp, C);
    hipLaunchKernelGGL(k_grid_fill, dim3(grid_blocks), dim3(256), 0, s, ga, mp);
    // two outer iterations (:563)
    FactorSoA F = c->factors();
    const int assoc_blocks = std::max(1, std::min(2048, div_up(c->slot_cap, 4)));
    for (int outer = 0; outer < 2; ++outer) {
        {
            SCAL_LAUNCH_PROF("k_assoc_knn", k_assoc_knn, dim3(assoc_blocks), dim3(256), 0, s, c->corner_stack().cv(), c->surf_stack().cv(), mp, c->grid[0].cnt.p,
                               c->grid[0].start.p, c->grid[0].pts(), c->grid[1].cnt.p, c->grid[1].start.p, c->grid[1].pts(), st, C, c->nnbuf());
        }
        {
            SCAL_LAUNCH_PROF("k_assoc_fit", k_assoc_fit, dim3(std::max(1, div_up(c->slot_cap, 64))), dim3(64), 0, s, c->corner_stack().cv(), c->surf_stack().cv(), c->nnbuf(),
                               C, outer, F);
        }
        {
                        launch_lm_solve(s, F, &C->n_slots, st, &C->solve_on, c->partials.p, c->lm_sync.p, outer);
        }
    }
    // the pose and the statistics known so far go to the host now; the map update follows behind
    SCAL_HIP(hipGetLastError());
    launch_publish(s, st, c->h_st.p, C, c->h_C1.p);
    SCAL_HIP(hipEventRecord(c->ev_pose, s));
    c->pose_pending = true;
    // insert + re-filter (:738-802)
    const bool try_merge = c->merge_insert && window_same && c->map[0].n + MERGE_MAX <= c->map_cap && c->map[1].n + MERGE_MAX <= c->map_cap;
    // restore the zero invariant of the cell counters (the merge insert does it in its key kernel)
    if (!try_merge) hipLaunchKernelGGL(k_grid_clear, dim3(grid_blocks), dim3(256), 0, s, ga, mp);
    if (try_merge) {
        MergeArgs a;
        for (int k = 0; k < 2; ++k) {
            MapStore& M = c->map[k];
            a.in[k] = M.cloud(M.cur), a.out[k] = M.cloud(M.cur ^ 1);
            a.n_old[k] = M.n;
            a.stack[k] = k == 0 ? c->corner_stack().cv() : c->surf_stack().cv();
            a.d_ns[k] = k == 0 ? &C->n_corner_stack : &C->n_surf_stack;
            a.inv_leaf[k] = k == 0 ? mp.inv_line : mp.inv_plane;
            a.okeys[k] = k == 0 ? c->keys.p : c->sorter.keys_alt.p;
            a.nw[k] = c->merge_new(k);
            a.nbo[k] = std::max(1, div_up(M.n, 256));
            a.grid_cnt[k] = c->grid[k].cnt.p, a.grid_rank[k] = c->grid[k].rank.p;
        }
        a.cap = c->map_cap;
        const int grid = a.nbo[0] + a.nbo[1] + 2 * MERGE_NEW_BLOCKS;
        static bool attr_set = false;
        const int lds = sizeof(unsigned long long) * MERGE_MAX;
        if (!attr_set) {
            SCAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_prepare), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(k_merge_keys, dim3(grid), dim3(256), 0, s, a, st, mp, C);
        hipLaunchKernelGGL(k_merge_prepare, dim3(2), dim3(1024), lds, s, a, C);
        hipLaunchKernelGGL(k_merge_write, dim3(grid), dim3(256), 0, s, a, C);
    } else {
        SCAL_TRY(insert_full_sort(c, mp));
    }
    if (have_full) {
        const int nb = std::max(1, div_up(c->scan_cap, 256));
        hipLaunchKernelGGL(k_transform_cloud, dim3(nb), dim3(256), 0, s, full_view, d_n_full, c->scan_cap, st, c->full_out.v());
    }
    launch_publish(s, C, c->h_C.p, static_cast<const MapCounters*>(nullptr), static_cast<MapCounters*>(nullptr));
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(hipEventRecord(c->ev_done, s));
    c->insert_pending = true, c->insert_try_merge = try_merge, c->pend_mp = mp;
    return SCAL_OK;
}

// waits for the pose of the enqueued step; map sizes in `stats` are those before this step's insertion (insert_path = -1)
static int map_collect_pose(scal_map* c, double* q_out, double* t_out, scal_map_stats* stats) {
    if (!c->pose_pending) {
        set_error("scal_map_collect: no step enqueued");
        return SCAL_E_STATE;
    }
    c->pose_pending = false;
    SCAL_HIP(hipEventSynchronize(c->ev_pose));
    const MapCounters& H = *c->h_C1.p;
    if (H.error) {
        set_error("scal_map_step: device capacity exceeded (map pool of %d points per class, or a voxel outside its cube)", c->map_cap);
        return H.error;
    }
    if (c->h_st.p->termination == 5) {  // a grid barrier of the LM solve ran out of polls: never seen, but do not trust the pose
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemset(c->lm_sync.p, 0, sizeof(LMSync));
        set_error("LM solve abandoned: grid barrier timed out");
        return SCAL_E_HIP;
    }
    const double* q_wodom = c->pend_q_wodom;
    const double* t_wodom = c->pend_t_wodom;
    const double* xf = c->h_st.p->x;
    for (int i = 0; i < 4; ++i) q_out[i] = xf[i];
    for (int i = 0; i < 3; ++i) t_out[i] = xf[4 + i];
    {  // transformUpdate (:149-153): q_wmap_wodom = q_w_curr * q_wodom_curr^-1 ; t_wmap_wodom = t_w_curr - q_wmap_wodom * t_wodom_curr
        const double n2 = q_wodom[0] * q_wodom[0] + q_wodom[1] * q_wodom[1] + q_wodom[2] * q_wodom[2] + q_wodom[3] * q_wodom[3];
        const double qi[4] = {-q_wodom[0] / n2, -q_wodom[1] / n2, -q_wodom[2] / n2, q_wodom[3] / n2};
        h_qmul(xf, qi, c->q_wmap_wodom);
        double r2[3];
        h_rot(c->q_wmap_wodom, t_wodom, r2);
        for (int i = 0; i < 3; ++i) c->t_wmap_wodom[i] = xf[4 + i] - r2[i];
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->n_corner_stack = H.n_corner_stack, stats->n_surf_stack = H.n_surf_stack;
        stats->n_corner_map = H.n_valid[0], stats->n_surf_map = H.n_valid[1];
        for (int o = 0; o < 2; ++o) {
            const LMState& L = *c->h_st.p;
            stats->n_edge[o] = H.solve_on ? L.log_n_edge[o] : 0, stats->n_plane[o] = H.solve_on ? L.log_n_plane[o] : 0;
            stats->lm_iters[o] = H.solve_on ? L.log_iters[o] : 0, stats->lm_success[o] = H.solve_on ? L.log_success[o] : 0;
            stats->cost_init[o] = H.solve_on ? L.log_cost_init[o] : 0.0, stats->cost_final[o] = H.solve_on ? L.log_cost_final[o] : 0.0;
        }
        stats->solved = H.solve_on;
        stats->n_map_corner_total = c->map[0].n, stats->n_map_surf_total = c->map[1].n;
        stats->insert_path = -1;
    }
    return SCAL_OK;
}

// synchronous form: enqueue, pose, insertion finished, final map sizes
int run_step(scal_map* c, const double* q_wodom, const double* t_wodom, bool have_full, CSoA4 full_view, const int* d_n_full, int n_corner_bound,
             int n_surf_bound, bool filters_done, double* q_out, double* t_out, scal_map_stats* stats) {
    SCAL_TRY(map_enqueue(c, q_wodom, t_wodom, have_full, full_view, d_n_full, n_corner_bound, n_surf_bound, filters_done));
    SCAL_TRY(map_collect_pose(c, q_out, t_out, stats));
    SCAL_TRY(map_finish(c));
    if (stats) {
        stats->n_map_corner_total = c->map[0].n, stats->n_map_surf_total = c->map[1].n;
        stats->insert_path = c->last_insert_path;
    }
    return SCAL_OK;
}

int reset_counters(scal_map* c, int n_corner, int n_surf, int n_full) {
    MapCounters z;
    std::memset(&z, 0, sizeof z);
    z.n_corner_in = n_corner, z.n_surf_in = n_surf;
    *c->h_C.p = z;
    SCAL_HIP(hipMemcpyAsync(c->d_C().p, c->h_C.p, sizeof(MapCounters), hipMemcpyHostToDevice, c->stream));
    c->h_misc.p[0] = n_full;
    SCAL_HIP(hipMemcpyAsync(c->d_nfull.p, c->h_misc.p, sizeof(int), hipMemcpyHostToDevice, c->stream));
    return SCAL_OK;
}

}  // namespace

extern "C" int scal_map_step(scal_map_t* c, const float* corner_last, int n_corner, const float* surf_last, int n_surf, const float* full_res,
                             int n_full, const double* q_wodom, const double* t_wodom, double* q_w_curr, double* t_w_curr, float* registered,
                             scal_map_stats* stats) {
    if (!c || !q_wodom || !t_wodom || !q_w_curr || !t_w_curr || n_corner < 0 || n_surf < 0 || n_full < 0 || (n_corner > 0 && !corner_last) ||
        (n_surf > 0 && !surf_last)) {
        set_error("scal_map_step: bad argument");
        return SCAL_E_ARG;
    }
    if (n_corner > c->scan_cap || n_surf > c->scan_cap || n_full > c->scan_cap) {
        set_error("scal_map_step: input cloud larger than max_scan_points (%d)", c->scan_cap);
        return SCAL_E_TOO_MANY;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const bool have_full = full_res != nullptr && n_full > 0;
    int nf = have_full ? n_full : 0;
    SCAL_TRY(map_finish(c));  // a pending insertion of an asynchronous step still owns the current set
    {
        std::lock_guard<std::mutex> lk(c->pf_mu);
        c->n_pf = 0;  // queued prefetches belong to a features context: not used by this entry point
        c->set = (c->set + 1) % scal_map::NSETS;
    }
    c->pending_prefetched = false;
    // n_full lives in pinned-less host memory for the async copy: stage through the counters struct instead
    SCAL_TRY(reset_counters(c, n_corner, n_surf, nf));
    SCAL_HIP(hipStreamSynchronize(s));  // &nf must not be read after return
    auto up = [&](const float* src, int n, SoAStore& dst) -> int {
        if (n > 0) {
            SCAL_HIP(hipMemcpyAsync(c->aos.p, src, sizeof(float) * 4 * n, hipMemcpyHostToDevice, s));
            launch_deinterleave(s, c->aos.p, n, dst.v());
        }
        return SCAL_OK;
    };
    SCAL_TRY(up(corner_last, n_corner, c->corner_in()));
    SCAL_TRY(up(surf_last, n_surf, c->surf_in()));
    if (have_full) SCAL_TRY(up(full_res, n_full, c->full_in));
    SCAL_TRY(run_step(c, q_wodom, t_wodom, have_full, c->full_in.cv(), c->d_nfull.p, n_corner, n_surf, false, q_w_curr, t_w_curr, stats));
    if (have_full && registered) {
        launch_interleave(s, c->d_nfull.p, n_full, c->full_out.cv(), c->aos.p);
        SCAL_HIP(hipMemcpyAsync(registered, c->aos.p, sizeof(float) * 4 * n_full, hipMemcpyDeviceToHost, s));
        SCAL_HIP(hipStreamSynchronize(s));
    }
    return SCAL_OK;
}

// laserCloudCornerLast = lessSharp cloud, laserCloudSurfLast = lessFlat cloud (laserOdometry.cpp:554-563); the full-res cloud is
// read in place by the registration transform
static int enqueue_gather(scal_map* c, const FeatDeviceView& v, hipStream_t s, int ls_cap, int cap, int st) {
    const int nbc = std::max(1, div_up(ls_cap, 256));
    hipLaunchKernelGGL(k_map_gather, dim3(nbc + std::max(1, div_up(cap, 256))), dim3(256), 0, s, v.less_xyzi, &v.P->n_less_sharp,
                       CSoA4{v.lfx, v.lfy, v.lfz, v.lfi}, &v.P->n_less_flat, c->corner_in(st).v(), c->surf_in(st).v(), c->d_C(st).p, c->scan_cap, nbc);
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

extern "C" int scal_map_prefetch_features(scal_map_t* c, scal_features_t* feat) {
    if (!c || !feat) {
        set_error("scal_map_prefetch_features: null argument");
        return SCAL_E_ARG;
    }
    FeatDeviceView v = features_view(feat);
    if (v.device != c->cfg.device) {
        set_error("features context lives on device %d, map context on %d", v.device, c->cfg.device);
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    if (!c->side) SCAL_TRY(acquire_stream(c->cfg.device, &c->side, 2));
    const int ls_cap = std::min(c->scan_cap, v.n_scans * 120);
    const int cap = std::min(c->scan_cap, v.cap);
    std::lock_guard<std::mutex> lk(c->pf_mu);
    if (c->n_pf >= 2) {
        set_error("scal_map_prefetch_features: two prefetches are already queued ahead of their steps");
        return SCAL_E_STATE;
    }
    // the step in flight owns set `c->set`, queued prefetches own the following ones; this one takes the next in the ring
    const int nset = ((c->n_pf ? c->pf[c->n_pf - 1].set : c->set) + 1) % scal_map::NSETS;
    if (!c->ev_pre[nset]) SCAL_HIP(hipEventCreateWithFlags(&c->ev_pre[nset], hipEventDisableTiming));
    SCAL_TRY(features_wait_done(feat, c->side));
    // The filter scratch (c->vf) is shared with a step that ran its filters on the main stream: wait for that step in that case.
    if (c->insert_pending && !c->pending_prefetched) SCAL_HIP(hipStreamWaitEvent(c->side, c->ev_done, 0));
    SCAL_TRY(enqueue_gather(c, v, c->side, ls_cap, cap, nset));
    SCAL_TRY(enqueue_stack_filters(c, c->side, ls_cap, cap, nset));
    SCAL_HIP(hipEventRecord(c->ev_pre[nset], c->side));
    SCAL_TRY(features_note_reader(feat, c->side));
    c->pf[c->n_pf].feat = feat, c->pf[c->n_pf].set = nset;
    c->n_pf++;
    return SCAL_OK;
}

static int map_enqueue_features(scal_map* c, scal_features_t* feat, const double* q_wodom, const double* t_wodom) {
    FeatDeviceView v = features_view(feat);
    if (v.device != c->cfg.device) {
        set_error("features context lives on device %d, map context on %d", v.device, c->cfg.device);
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_TRY(map_finish(c));  // the previous insertion still owns the counters and the stack buffers
    hipStream_t s = c->stream;
    const int ls_cap = std::min(c->scan_cap, v.n_scans * 120);
    const int cap = std::min(c->scan_cap, v.cap);
    bool pre = false;
    {
        std::lock_guard<std::mutex> lk(c->pf_mu);
        if (c->n_pf > 0 && c->pf[0].feat == feat) {  // the oldest queued prefetch belongs to this step: take its set
            pre = true;
            c->set = c->pf[0].set;
            c->pf[0] = c->pf[1];
            c->n_pf--;
        } else {  // no (matching) prefetch: drop what was queued and take the next set of the ring
            c->set = ((c->n_pf ? c->pf[c->n_pf - 1].set : c->set) + 1) % scal_map::NSETS;
            c->n_pf = 0;
        }
    }
    c->pending_prefetched = pre;
    if (pre) {
        SCAL_HIP(hipStreamWaitEvent(s, c->ev_pre[c->set], 0));  // inputs gathered and downsampled on the side stream
    } else {
        SCAL_TRY(features_wait_done(feat, s));
        SCAL_TRY(enqueue_gather(c, v, s, ls_cap, cap, c->set));
    }
    SCAL_TRY(map_enqueue(c, q_wodom, t_wodom, true, CSoA4{v.x, v.y, v.z, v.i}, &v.P->n_kept, ls_cap, cap, pre));
    return features_note_reader(feat, s);  // the registration transform reads the full-resolution cloud last
}

extern "C" int scal_map_step_features(scal_map_t* c, scal_features_t* feat, const double* q_wodom, const double* t_wodom, double* q_w_curr,
                                      double* t_w_curr, scal_map_stats* stats) {
    if (!c || !feat || !q_wodom || !t_wodom || !q_w_curr || !t_w_curr) {
        set_error("scal_map_step_features: null argument");
        return SCAL_E_ARG;
    }
    SCAL_TRY(map_enqueue_features(c, feat, q_wodom, t_wodom));
    SCAL_TRY(map_collect_pose(c, q_w_curr, t_w_curr, stats));
    SCAL_TRY(map_finish(c));
    if (stats) {
        stats->n_map_corner_total = c->map[0].n, stats->n_map_surf_total = c->map[1].n;
        stats->insert_path = c->last_insert_path;
    }
    return SCAL_OK;
}

extern "C" int scal_map_enqueue_features(scal_map_t* c, scal_features_t* feat, const double* q_wodom, const double* t_wodom) {
    if (!c || !feat || !q_wodom || !t_wodom) {
        set_error("scal_map_enqueue_features: null argument");
        return SCAL_E_ARG;
    }
    return map_enqueue_features(c, feat, q_wodom, t_wodom);
}

extern "C" int scal_map_collect(scal_map_t* c, double* q_w_curr, double* t_w_curr, scal_map_stats* stats) {
    if (!c || !q_w_curr || !t_w_curr) {
        set_error("scal_map_collect: null argument");
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    return map_collect_pose(c, q_w_curr, t_w_curr, stats);
}

extern "C" int scal_map_finish(scal_map_t* c) {
    if (!c) return SCAL_E_ARG;
    SCAL_HIP(hipSetDevice(c->cfg.device));
    return map_finish(c);
}

extern "C" int scal_map_export(scal_map_t* c, int which, float* out_xyzi, int cap) {
    if (!c || (which != 0 && which != 1) || cap < 0) {
        set_error("scal_map_export: bad argument");
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    if (map_finish(c) != SCAL_OK) return -1;
    if (!c->have_mp || c->map[which].n == 0) return 0;
    hipStream_t s = c->stream;
    MapStore& M = c->map[which];
    int zero = 0;
    SCAL_HIP(hipMemcpyAsync(c->d_nfull.p + 1, &zero, sizeof(int), hipMemcpyHostToDevice, s));
    SCAL_HIP(hipStreamSynchronize(s));
    const int room = std::min(cap, c->scan_cap);
    hipLaunchKernelGGL(k_export_valid, dim3(std::max(1, div_up(M.n, 256))), dim3(256), 0, s, M.cloud(M.cur), M.n, c->last_mp, c->d_nfull.p + 1, c->aos.p,
                       out_xyzi ? room : 0);
    int n = 0;
    SCAL_HIP(hipMemcpyAsync(&n, c->d_nfull.p + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    SCAL_HIP(hipStreamSynchronize(s));
    if (out_xyzi && room > 0) {
        const int m = std::min(n, room);
        SCAL_HIP(hipMemcpyAsync(out_xyzi, c->aos.p, sizeof(float) * 4 * m, hipMemcpyDeviceToHost, s));
        SCAL_HIP(hipStreamSynchronize(s));
        return m;
    }
    return n;
}

extern "C" int scal_map_set_merge_insert(scal_map_t* c, int enable) {
    if (!c) return SCAL_E_ARG;
    c->merge_insert = enable != 0;
    return SCAL_OK;
}

extern "C" int scal_map_get_wmap_wodom(scal_map_t* c, double* q, double* t) {
    if (!c || !q || !t) return SCAL_E_ARG;
    for (int i = 0; i < 4; ++i) q[i] = c->q_wmap_wodom[i];
    for (int i = 0; i < 3; ++i) t[i] = c->t_wmap_wodom[i];
    return SCAL_OK;
}
