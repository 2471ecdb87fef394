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
//                                      index; closed-form largest eigenpair (edge) or pivoted Householder QR
//                                      (plane) in f64 registers -> residual blocks
//   solve (:713-721) x2 (:563)         lm_dev.hpp, all on the device
//   insert + per-cube voxel (:738-802) k_insert_keys -> radix sort by (cube, voxel) -> k_map_reduce; cubes outside
//                                      the 5x5x3 valid set keep raw points in arrival order, as the reference does
//   registration (:845-849)            k_transform_cloud
#include "common.hpp"
#include <mutex>
#include <atomic>
#include "device_utils.hpp"
#include "voxel_dev.hpp"
#include "radix_sort.hpp"
#include "lm_dev.hpp"
#include "features_dev.hpp"
#include <cmath>
#include <cstdlib>
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

// Persistent device-side state of the mapper.  The pose algebra of :143-153, the rolling-window decision of :313-508 and the
// map sizes live here, so that a step can be queued behind the previous one without the host reading anything back.
enum { MAP_ABORT_NONE = 0, MAP_ABORT_WINDOW = 1, MAP_ABORT_MERGE = 2, MAP_ABORT_GRID = 3, MAP_ABORT_LM = 4 };
static_assert(MAP_ABORT_LM == LM_ABORT_CODE, "the LM solve raises the abort word itself when a workgroup gives up");
struct MapState {
    double q_wmap_wodom[4], t_wmap_wodom[3];  // :113-114
    double q_wodom[4], t_wodom[3];            // this step's /laser_odom_to_init pose (kept for transformUpdate)
    MapParams mp;                             // window of the current step
    int have_mp;
    int window_same;                          // this step's window equals the previous step's (merge insert possible)
    int n_map[2];                             // points per class in the current map buffers
    int abort;                                // sticky: while set, every kernel of a speculatively queued step returns at once
    int seq;                                  // steps begun
    // left by a merge write that also binned its output into the NEXT step's cell grid (MergeArgs::prebuild), taken over and
    // cleared by that step's k_map_begin: points inside the window, "a stored point no longer has the key it was placed under"
    // (the next merge insert must not trust the order: merge_fail), "a cell overflowed its fixed slice" (MAP_ABORT_GRID)
    int next_valid[2];
    int next_unsorted;
    int next_over;
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

struct MapCounters;
__device__ __forceinline__ unsigned long long map_key(const MapParams& mp, float inv_leaf, float x, float y, float z, int pc, MapCounters* C);
constexpr unsigned long long MAP_NOMERGE_SLOT = 127ull;
__device__ __forceinline__ bool key_nomerge(unsigned long long k) { return (k >> 27) == MAP_NOMERGE_SLOT; }

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
    int2* cell[2];  // per cell (count, start in the point pool): one 8-byte load per neighbour cell
    int* rank[2];
    GridPts g[2];
    // one-launch build only: the merge insert's keys of the old points and its sortedness check ride on this walk over the map
    unsigned long long* okeys[2];
    float inv_leaf[2];
};
// Fixed launch shape (the map sizes are device words): a quarter of the blocks walks the corner map, the rest the surf map,
// each with a block-stride loop.  `i0` is the first point of this block's current tile, uniform over the block.
constexpr int GRID_BLOCKS = 2048;
__device__ __forceinline__ void grid_part(int& cls, int& first, int& stride) {
    const int nb0 = gridDim.x / 4;
    const int b = blockIdx.x;
    cls = b < nb0 ? 0 : 1;
    first = (cls ? b - nb0 : b) * 256;
    stride = (cls ? static_cast<int>(gridDim.x) - nb0 : nb0) * 256;
}

__device__ __forceinline__ void k_grid_count_body(const GridArgs& a, const MapState* __restrict__ S, MapCounters* C) {
    if (S->abort) return;
    __shared__ int s_valid;
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    int cls, first, stride;
    grid_part(cls, first, stride);
    const MapParams mp = S->mp;
    const int n = S->n_map[cls];
    const MapCloud& m = a.m[cls];
    for (int i0 = first; i0 < n; i0 += stride) {
        const int i = i0 + threadIdx.x;
        const bool in = i < n;
        bool v = false;
        if (in && cube_valid(mp, m.cube[i])) {
            v = true;
            a.rank[cls][i] = atomicAdd(&a.cell[cls][grid_cell(mp, m.x[i], m.y[i], m.z[i])].x, 1);
        } else if (in) {
            a.rank[cls][i] = -1;
        }
        // one global atomic per workgroup: a per-wave atomic on the same word was most of this kernel's time
        const uint64_t b = __ballot(v);
        if (lane_id() == 0 && b) atomicAdd(&s_valid, __popcll(b));
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_valid) atomicAdd(&C->n_valid[cls], s_valid);
}
SCAL_KERNEL(256, k_grid_count)

// The cell grid in ONE launch for the speculative chain: every cell owns a fixed slice of `cap` entries of a (large, sparsely
// touched) pool, so a point's place is known as soon as it has its rank in the cell - no allocation pass, no second walk over the
// map.  `cap` = the number of filter voxels that can intersect a 1 m cell (27 at 0.4 m, 8 at 0.8 m): on this path the window did
// not move, so every binned point is the only one of its voxel (the re-filter of the previous scan, :775-791).  A centroid rounded
// across a cell face could still make a full cell overflow: that stops the chain (MAP_ABORT_GRID), the host clears the counters
// and redoes the step with the three general launches below.
__device__ __forceinline__ void k_grid_build_body(const GridArgs& a, int cap0, int cap1, MapState* S, MapCounters* C) {
    if (S->abort) return;
    __shared__ int s_valid;
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    int cls, first, stride;
    grid_part(cls, first, stride);
    const MapParams mp = S->mp;
    const int n = S->n_map[cls];
    const int cap = cls ? cap1 : cap0;
    const MapCloud& m = a.m[cls];
    bool over = false, unsorted = false;
    for (int i0 = first; i0 < n; i0 += stride) {
        const int i = i0 + threadIdx.x;
        const bool in = i < n;
        bool v = false;
        if (in) {
            const float x = m.x[i], y = m.y[i], z = m.z[i];
            const int pc = m.cube[i];
            // the merge insert's key of this (old) point and its sortedness check: the window is fixed for the step, so they are
            // known here, on a walk over the map that happens anyway (round 2 walked the map a second time for them in k_merge_keys)
            const unsigned long long k = map_key(mp, a.inv_leaf[cls], x, y, z, pc, C);
            a.okeys[cls][i] = k;
            bool bad = k == ~0ull;
            if (i > 0) {
                const unsigned long long kp = map_key(mp, a.inv_leaf[cls], m.x[i - 1], m.y[i - 1], m.z[i - 1], m.cube[i - 1], C);
                bad |= kp > k || (kp == k && !key_nomerge(k));
            }
            unsorted |= bad;
            if (cube_valid(mp, pc)) {
                v = true;
                const int c = grid_cell(mp, x, y, z);
                const int r = atomicAdd(&a.cell[cls][c].x, 1);
                a.rank[cls][i] = r;  // >= 0: this point's cell counter has to be cleared again
                if (r < cap) {
                    a.g[cls].p[static_cast<size_t>(c) * cap + r] = make_float4(x, y, z, __int_as_float(i));
                    if (r == 0) a.cell[cls][c].y = c * cap;
                } else {
                    over = true;
                }
            } else {
                a.rank[cls][i] = -1;
            }
        }
        const uint64_t b = __ballot(v);
        if (lane_id() == 0 && b) atomicAdd(&s_valid, __popcll(b));
    }
    if (unsorted) C->merge_fail = 1;
    if (over) __hip_atomic_store(&S->abort, static_cast<int>(MAP_ABORT_GRID), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0 && s_valid) atomicAdd(&C->n_valid[cls], s_valid);
}
SCAL_KERNEL(256, k_grid_build)

// every non-empty cell gets a slice of the point pool: the cells' sizes are summed per workgroup (the point with rank 0
// speaks for its cell), one cursor atomic per workgroup and tile
__device__ __forceinline__ void k_grid_alloc_body(const GridArgs& a, const MapState* __restrict__ S, MapCounters* C) {
    if (S->abort) return;
    __shared__ int s_scan[17];
    __shared__ int s_base;
    int cls, first, stride;
    grid_part(cls, first, stride);
    const MapParams mp = S->mp;
    const int n = S->n_map[cls];
    const MapCloud& m = a.m[cls];
    for (int i0 = first; i0 < n; i0 += stride) {
        const int i = i0 + threadIdx.x;
        int c = -1, mine = 0;
        if (i < n && a.rank[cls][i] == 0) {
            c = grid_cell(mp, m.x[i], m.y[i], m.z[i]);
            mine = a.cell[cls][c].x;
        }
        int total = 0;
        const int off = block_exclusive_scan(mine, s_scan, &total);
        if (threadIdx.x == 0) s_base = total ? atomicAdd(&C->cursor[cls], total) : 0;
        __syncthreads();
        if (c >= 0) a.cell[cls][c].y = s_base + off;
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) C->solve_on = (C->n_valid[0] > 10 && C->n_valid[1] > 50) ? 1 : 0;  // :555
}
SCAL_KERNEL(256, k_grid_alloc)

__device__ __forceinline__ void k_grid_fill_body(const GridArgs& a, const MapState* __restrict__ S) {
    if (S->abort) return;
    int cls, first, stride;
    grid_part(cls, first, stride);
    const MapParams mp = S->mp;
    const int n = S->n_map[cls];
    const MapCloud& m = a.m[cls];
    const GridPts& g = a.g[cls];
    for (int i = first + threadIdx.x; i < n; i += stride) {
        if (a.rank[cls][i] >= 0) {
            const int c = grid_cell(mp, m.x[i], m.y[i], m.z[i]);
            const int p = a.cell[cls][c].y + a.rank[cls][i];
            g.p[p] = make_float4(m.x[i], m.y[i], m.z[i], __int_as_float(i));
        }
    }
}
SCAL_KERNEL(256, k_grid_fill)

__device__ __forceinline__ void k_grid_clear_body(const GridArgs& a, const MapState* __restrict__ S) {
    int cls, first, stride;
    grid_part(cls, first, stride);
    const MapParams mp = S->mp;
    const int n = S->n_map[cls];
    const MapCloud& m = a.m[cls];
    for (int i = first + threadIdx.x; i < n; i += stride)
        if (a.rank[cls][i] >= 0) a.cell[cls][grid_cell(mp, m.x[i], m.y[i], m.z[i])].x = 0;
}
SCAL_KERNEL(256, k_grid_clear)

// ---------------------------------------------------------------------------------------------- association

// Exact 5 nearest map points of q among the 27 cells around it, ascending (distance, map index), by HALF A WAVE: lanes 0-31 serve
// one query, lanes 32-63 another.  Lane c < 27 of a half owns neighbour cell c (one round trip for the 27 (count, start) pairs) and
// walks its points, two loads in flight; every lane keeps the five best of ITS candidates as an ascending register list of 64-bit
// (f32 distance bits, map index) keys.  The half then takes the minimum of the list heads five times (DPP reduction), the owning
// lane popping its head each time.  (Rounds 1-3 gave a query a whole wave, two lanes per cell: the 9-13 k queries of a scan did not
// fit the device at once at 6 waves per SIMD, and a wave is a chain of dependent round trips either way - 16 us per launch.)
// Every lane of a half returns the same ascending (key, grid position) list; position -1 = fewer than five candidates.
__device__ __forceinline__ void knn5_insert(unsigned long long (&k)[5], int (&p)[5], unsigned long long key, int pos) {
    // branch-free, statically indexed insertion into the ascending list (keys are distinct: they carry the map index)
    const bool c0 = key < k[0], c1 = key < k[1], c2 = key < k[2], c3 = key < k[3], c4 = key < k[4];
    k[4] = c3 ? k[3] : (c4 ? key : k[4]), p[4] = c3 ? p[3] : (c4 ? pos : p[4]);
    k[3] = c2 ? k[2] : (c3 ? key : k[3]), p[3] = c2 ? p[2] : (c3 ? pos : p[3]);
    k[2] = c1 ? k[1] : (c2 ? key : k[2]), p[2] = c1 ? p[1] : (c2 ? pos : p[2]);
    k[1] = c0 ? k[0] : (c1 ? key : k[1]), p[1] = c0 ? p[0] : (c1 ? pos : p[1]);
    k[0] = c0 ? key : k[0], p[0] = c0 ? pos : p[0];
}
__device__ __forceinline__ unsigned long long knn_key(const float4 pt, float qx, float qy, float qz) {
    // FLANN L2_Simple<float>: ((0 + dx^2) + dy^2) + dz^2
    const float dx = qx - pt.x, dy = qy - pt.y, dz = qz - pt.z;
    float dist = dx * dx;
    dist += dy * dy;
    dist += dz * dz;
    return (static_cast<unsigned long long>(__float_as_uint(dist)) << 32) | static_cast<unsigned>(__float_as_int(pt.w));
}
__device__ __forceinline__ void knn5_half(const MapParams& mp, const int2* __restrict__ cell, const float4* __restrict__ pool, bool active,
                                          float qx, float qy, float qz, unsigned long long (&bk)[5], int (&bp)[5]) {
    const int lane = lane_id(), hl = lane & 31, half = lane >> 5;
    const int cx = static_cast<int>(floorf(qx)) - mp.ox, cy = static_cast<int>(floorf(qy)) - mp.oy, cz = static_cast<int>(floorf(qz)) - mp.oz;
    // a query further than one cell outside the grid has no map point within 1 m
    const bool inside = !(cx < -1 || cx > GX || cy < -1 || cy > GY || cz < -1 || cz > GZ);
    int cnt = 0, start = 0;
    if (active && inside && hl < 27) {
        const int xx = cx + (hl % 3) - 1, yy = cy + ((hl / 3) % 3) - 1, zz = cz + (hl / 9) - 1;
        if (xx >= 0 && xx < GX && yy >= 0 && yy < GY && zz >= 0 && zz < GZ) {
            const int2 h = cell[xx + GX * (yy + GY * zz)];
            cnt = h.x, start = h.y;
        }
    }
    unsigned long long mk[5];
    int mpos[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) mk[k] = ~0ull, mpos[k] = -1;
    for (int t = start, end = start + cnt; __ballot(t < end) != 0; t += 2) {
        const bool a = t < end, b = t + 1 < end;
        float4 pa, pb;
        if (a) pa = pool[t];
        if (b) pb = pool[t + 1];
        if (a) knn5_insert(mk, mpos, knn_key(pa, qx, qy, qz), t);
        if (b) knn5_insert(mk, mpos, knn_key(pb, qx, qy, qz), t + 1);
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const unsigned long long best = half_min_u64_dpp(mk[0]);
        const bool have = best != ~0ull;
        const uint64_t ownb = __ballot(have && mk[0] == best);  // keys are distinct inside a query: at most one lane per half
        const unsigned own0 = static_cast<unsigned>(ownb), own1 = static_cast<unsigned>(ownb >> 32);
        const int o0 = own0 ? __ffs(static_cast<int>(own0)) - 1 : 0, o1 = own1 ? 32 + __ffs(static_cast<int>(own1)) - 1 : 32;
        const int p0 = __builtin_amdgcn_readlane(mpos[0], o0), p1 = __builtin_amdgcn_readlane(mpos[0], o1);
        bk[r] = best;
        bp[r] = have ? (half ? p1 : p0) : -1;
        if (have && lane == (half ? o1 : o0)) {
            mk[0] = mk[1], mk[1] = mk[2], mk[2] = mk[3], mk[3] = mk[4], mk[4] = ~0ull;
            mpos[0] = mpos[1], mpos[1] = mpos[2], mpos[2] = mpos[3], mpos[3] = mpos[4], mpos[4] = -1;
        }
    }
}

// Largest eigenpair and second-largest eigenvalue of a symmetric positive semi-definite 3x3 matrix (stands in for
// Eigen::SelfAdjointEigenSolver, :606-612): eigenvalues in closed form (trigonometric solution of the characteristic cubic of the
// shifted, scaled matrix), the eigenvector as the largest cross product of two rows of A - w I.  One sqrt, one reciprocal, one acos,
// two cos, one more sqrt and division on the critical path - a cyclic Jacobi iteration to 1e-20 (rounds 1-2) took 18 rotations of two
// square roots and three divisions each and was 10-16 us of k_assoc_fit's 17.  Where the reference goes on (largest > 3 x second, :612) the
// largest eigenvalue is simple and well separated, which is where both the closed form and the cross product are accurate to rounding.
__device__ __forceinline__ void eig3_largest(double a00, double a01, double a02, double a11, double a12, double a22, double* w1, double* w2,
                                             double* dir) {
    const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
    const double q = (a00 + a11 + a22) / 3.0;
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
    dir[0] = 1.0, dir[1] = 0.0, dir[2] = 0.0;
    if (!(p2 > 0.0)) {  // a multiple of the identity (or not a number): no direction stands out
        *w1 = q, *w2 = q;
        return;
    }
    const double p = sqrt(p2 / 6.0);
    const double ip = 1.0 / p;
    const double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
    double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
    r = fmin(1.0, fmax(-1.0, r));
    const double phi = acos(r) / 3.0;
    const double l3 = q + 2.0 * p * cos(phi);                          // largest
    const double l1 = q + 2.0 * p * cos(phi + 2.0943951023931953);     // smallest (phi + 2 pi / 3)
    *w2 = l3;
    *w1 = 3.0 * q - l1 - l3;
    const double r0[3] = {a00 - l3, a01, a02}, r1[3] = {a01, a11 - l3, a12}, r2[3] = {a02, a12, a22 - l3};
    const double x0[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    const double x1[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
    const double x2[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    const double n0 = x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2];
    const double n1 = x1[0] * x1[0] + x1[1] * x1[1] + x1[2] * x1[2];
    const double n2 = x2[0] * x2[0] + x2[1] * x2[1] + x2[2] * x2[2];
    const bool use1 = n1 > n0 && n1 >= n2, use2 = n2 > n0 && n2 > n1;
    const double vx = use2 ? x2[0] : (use1 ? x1[0] : x0[0]), vy = use2 ? x2[1] : (use1 ? x1[1] : x0[1]), vz = use2 ? x2[2] : (use1 ? x1[2] : x0[2]);
    const double nn = use2 ? n2 : (use1 ? n1 : n0);
    if (nn > 0.0) {
        const double n = sqrt(nn);
        dir[0] = vx / n, dir[1] = vy / n, dir[2] = vz / n;
    }
}

// least squares A n = b for a 5x3 A by column-pivoted Householder QR (stands in for colPivHouseholderQr().solve, :664)
// Every index is a compile-time constant after unrolling (column swaps, the permutation and the rank-limited back substitution are
// selects): the arrays stay in registers.  With run-time indices they lived in scratch memory, 7 of k_assoc_fit's 14 us.
__device__ __forceinline__ void colpiv_qr_5x3(double (&A)[5][3], double (&b)[5], double (&x)[3]) {
    int perm[3] = {0, 1, 2};
    double rdiag[3] = {0, 0, 0};
    double maxpivot = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int best = k;
        double bestn = -1;
#pragma unroll
        for (int j = k; j < 3; ++j) {
            double s = 0;
#pragma unroll
            for (int i = k; i < 5; ++i) s += A[i][j] * A[i][j];
            if (s > bestn) bestn = s, best = j;
        }
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            const bool sw = best == j;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const double t = A[i][k];
                A[i][k] = sw ? A[i][j] : t;
                A[i][j] = sw ? t : A[i][j];
            }
            const int t = perm[k];
            perm[k] = sw ? perm[j] : t;
            perm[j] = sw ? t : perm[j];
        }
        double tail = 0;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) tail += A[i][k] * A[i][k];
        const double c0 = A[k][k];
        double tau, beta;
        if (tail <= 2.2250738585072014e-308) {
            tau = 0, beta = c0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) A[i][k] = 0;
        } else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0) beta = -beta;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) A[i][k] /= (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        rdiag[k] = beta;
        maxpivot = fmax(maxpivot, fabs(beta));
        if (tau != 0) {
#pragma unroll
            for (int c = k + 1; c < 3; ++c) {
                double s = A[k][c];
#pragma unroll
                for (int i = k + 1; i < 5; ++i) s += A[i][k] * A[i][c];
                s *= tau;
                A[k][c] -= s;
#pragma unroll
                for (int i = k + 1; i < 5; ++i) A[i][c] -= s * A[i][k];
            }
            double s = b[k];
#pragma unroll
            for (int i = k + 1; i < 5; ++i) s += A[i][k] * b[i];
            s *= tau;
            b[k] -= s;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) b[i] -= s * A[i][k];
        }
    }
    const double thr = 2.220446049250313e-16 * 3.0 * maxpivot;
    int rank = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (fabs(rdiag[k]) > thr) ++rank;
    double y[3] = {0, 0, 0};
#pragma unroll
    for (int k = 2; k >= 0; --k)
        if (k < rank) {
            double s = b[k];
#pragma unroll
            for (int c = k + 1; c < 3; ++c)
                if (c < rank) s -= A[k][c] * y[c];
            y[k] = s / A[k][k];
        }
    x[0] = x[1] = x[2] = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (perm[k] == d) x[d] = y[k];
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
    float* rec;  // [cap][16]: the five neighbours of a slot as x y z triples, ascending (distance, map index), then the squared distance
                 // of the 5th in [15] - one 64-byte line per slot, written by five lanes of the search, read by the one thread that fits
                 // (SoA columns until round 3: 16 scattered 4-byte stores per slot, 1.8 MB of write traffic for 0.7 MB of payload)
    int cap;
};

__host__ __device__ inline void m_qmul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
__host__ __device__ inline void m_rot(const double* q, const double* v, double* o) {
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux, uy += uy, uz += uz;
    const double cx = q[1] * uz - q[2] * uy, cy = q[2] * ux - q[0] * uz, cz = q[0] * uy - q[1] * ux;
    o[0] = (v[0] + q[3] * ux) + cx, o[1] = (v[1] + q[3] * uy) + cy, o[2] = (v[2] + q[3] * uz) + cz;
}

struct MapPoseIn {
    double q_wodom[4], t_wodom[3];
};

// Start of a step: transformAssociateToMap (:143-147) and the rolling-window decision (:313-508) on the device.
// The pointer shuffles of :324-508 are offset updates; the slabs they clear are dropped by the next re-filter.
// allow_window_change = 0 (a step queued speculatively behind another one): a window that differs from the previous step's
// stops the chain (MAP_ABORT_WINDOW) - the grid, the keys and the counts the previous step's merge write left are for that window.
// Two halves: map_begin_eval reads the state and decides (any thread may run it: it writes nothing), map_begin_commit stores the
// outcome (one thread).  k_map_begin runs both; a queued step whose grid is in place has no launch between this and the first
// neighbour search, so there every workgroup of k_assoc_knn evaluates for itself and workgroup 0 commits (MapBeginArgs::active).
struct MapBeginArgs {
    int active;
    MapPoseIn in;
    int allow_window_change;
    float inv_line, inv_plane;
    int slot_cap, prebuilt;
    MapState* S;
};
struct MapBeginOut {
    int abort;      // MAP_ABORT_NONE / _WINDOW / _GRID
    MapParams mp;
    double x0[7];
    int nv0, nv1, drifted, same, n_slots;
};
__device__ __forceinline__ void map_begin_eval(const MapState* S, const MapCounters* C, const MapBeginArgs& a, MapBeginOut& o) {
    const MapPoseIn& in = a.in;
    o.n_slots = min(C->n_corner_stack + C->n_surf_stack, a.slot_cap);  // residual-block slots (both stack filters have finished)
    // what the previous step's merge write left for this one (MergeArgs::prebuild)
    o.nv0 = S->next_valid[0], o.nv1 = S->next_valid[1], o.drifted = S->next_unsorted;
    const int over = S->next_over;
    m_qmul(S->q_wmap_wodom, in.q_wodom, o.x0);
    double rt[3];
    m_rot(S->q_wmap_wodom, in.t_wodom, rt);
    o.x0[4] = rt[0] + S->t_wmap_wodom[0], o.x0[5] = rt[1] + S->t_wmap_wodom[1], o.x0[6] = rt[2] + S->t_wmap_wodom[2];
    int cenW = S->mp.cenW, cenH = S->mp.cenH, cenD = S->mp.cenD;
    // :313-322
    int cI = int((o.x0[4] + 25.0) / 50.0) + cenW, cJ = int((o.x0[5] + 25.0) / 50.0) + cenH, cK = int((o.x0[6] + 25.0) / 50.0) + cenD;
    if (o.x0[4] + 25.0 < 0) cI--;
    if (o.x0[5] + 25.0 < 0) cJ--;
    if (o.x0[6] + 25.0 < 0) cK--;
    while (cI < 3) cI++, cenW++;
    while (cI >= CW - 3) cI--, cenW--;
    while (cJ < 3) cJ++, cenH++;
    while (cJ >= CH - 3) cJ--, cenH--;
    while (cK < 3) cK++, cenD++;
    while (cK >= CD - 3) cK--, cenD--;
    MapParams& mp = o.mp;
    mp.cenW = cenW, mp.cenH = cenH, mp.cenD = cenD;
    mp.cI = cI, mp.cJ = cJ, mp.cK = cK;
    mp.ox = 50 * (cI - 2 - cenW) - 25, mp.oy = 50 * (cJ - 2 - cenH) - 25, mp.oz = 50 * (cK - 1 - cenD) - 25;
    mp.inv_line = a.inv_line, mp.inv_plane = a.inv_plane;
    const MapParams p = S->mp;
    const bool same = S->have_mp && p.cenW == mp.cenW && p.cenH == mp.cenH && p.cenD == mp.cenD && p.cI == mp.cI && p.cJ == mp.cJ && p.cK == mp.cK;
    o.same = same ? 1 : 0;
    o.abort = MAP_ABORT_NONE;
    if (!a.allow_window_change && !same) o.abort = MAP_ABORT_WINDOW;
    else if (a.prebuilt && over) o.abort = MAP_ABORT_GRID;  // a cell of the prebuilt grid overflowed its fixed slice: as k_grid_build would have reported it
}
// reset_next: the words the previous merge write left are cleared here (k_map_begin) or, when other workgroups may still be reading
// them (the evaluation inside k_assoc_knn), by this step's k_merge_keys - in front of the merge write that fills them again
__device__ __forceinline__ void map_begin_commit(MapState* S, MapCounters* C, LMState* st, const MapBeginArgs& a, const MapBeginOut& o, bool reset_next) {
    C->n_slots = o.n_slots;
    if (reset_next) S->next_valid[0] = 0, S->next_valid[1] = 0, S->next_unsorted = 0, S->next_over = 0;
    if (o.abort) {
        S->abort = o.abort;
        return;
    }
    if (a.prebuilt) {
        C->n_valid[0] = o.nv0, C->n_valid[1] = o.nv1;
        if (o.drifted) C->merge_fail = 1;
    }
    S->mp = o.mp;
    S->have_mp = 1;
    S->window_same = o.same;
    S->seq++;
#pragma unroll
    for (int k = 0; k < 4; ++k) S->q_wodom[k] = a.in.q_wodom[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) S->t_wodom[k] = a.in.t_wodom[k];
#pragma unroll
    for (int k = 0; k < 7; ++k) st->x[k] = o.x0[k];
}
__device__ __forceinline__ void k_map_begin_body(const MapBeginArgs& a, LMState* st, MapCounters* C) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (a.S->abort) return;
    MapBeginOut o;
    map_begin_eval(a.S, C, a, o);
    map_begin_commit(a.S, C, st, a, o, true);
}
SCAL_KERNEL(1024, k_map_begin)

// slots [0, n_corner_stack): edge candidates; [n_corner_stack, n_corner_stack + n_surf_stack): plane candidates
// k_assoc_knn: half a WAVE per stack point (the neighbour search is a handful of dependent memory round trips, so it wants
// many waves in flight); k_assoc_fit: one THREAD per stack point (the PCA / plane fit is ~3k dependent f64 operations,
// so it wants every lane busy with a different point).
__device__ __forceinline__ void k_assoc_knn_body(const CSoA4& cs, const CSoA4& ss, const MapState* __restrict__ S, const int2* __restrict__ ccell, const GridPts& cg,
                                                   const int2* __restrict__ scell, const GridPts& sg, LMState* st,
                                                   MapCounters* C, const NNBuf& nb, const MapBeginArgs& mb) {
    // everything the kernel decides on, fetched together: the checks below would otherwise be a chain of dependent scalar round trips
    const int stop = S->abort;
    int nv0 = C->n_valid[0], nv1 = C->n_valid[1];
    const int nc = C->n_corner_stack, ns = C->n_surf_stack;
    MapParams mp = S->mp;
    double x7[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) x7[k] = st->x[k];
    if (stop) return;
    if (mb.active) {
        // the step's start (k_map_begin's work) rides on this launch: every workgroup evaluates the same decision from the same
        // state - nothing the evaluation reads is changed by the commit - and workgroup 0 stores it for the kernels behind
        MapBeginOut o;
        map_begin_eval(S, C, mb, o);
        if (blockIdx.x == 0 && threadIdx.x == 0) map_begin_commit(mb.S, C, st, mb, o, false);
        if (o.abort) return;
        mp = o.mp, nv0 = o.nv0, nv1 = o.nv1;
#pragma unroll
        for (int k = 0; k < 7; ++k) x7[k] = o.x0[k];
    }
    // :555 - evaluated here (the map counts are complete once the grid is): the one-launch grid build has no later
    // launch of its own that could do it; workgroup 0 publishes the decision for the fit, the solve and the host
    const bool solve_on = nv0 > 10 && nv1 > 50;
    if (blockIdx.x == 0 && threadIdx.x == 0) C->solve_on = solve_on ? 1 : 0;
    if (!solve_on) return;
    const int n = min(nc + ns, nb.cap);
    const int lane = lane_id(), hl = lane & 31, half = lane >> 5;
    // bounded grid, wave-stride loop over pairs of slots: a grid sized for the capacity would be ~95 % empty workgroups
    for (int i0 = (blockIdx.x * 4 + wave_id()) * 2; i0 < n; i0 += gridDim.x * 8) {
        const int i = i0 + half;
        const bool active = i < n;
        const bool is_edge = i < nc;
        const int j = active ? (is_edge ? i : i - nc) : 0;
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (active) ox = is_edge ? cs.x[j] : ss.x[j], oy = is_edge ? cs.y[j] : ss.y[j], oz = is_edge ? cs.z[j] : ss.z[j];
        float sel[3];
        associate_to_map(x7, ox, oy, oz, sel);
        unsigned long long bk[5];
        int bp[5];
        const float4* pool = is_edge ? cg.p : sg.p;
        if (lane == 0 && i == 0) SCAL_STAMP(28);
        knn5_half(mp, is_edge ? ccell : scell, pool, active, sel[0], sel[1], sel[2], bk, bp);
        if (lane == 0 && i == 0) SCAL_STAMP(29);
        // lane k < 5 of a half fetches and stores neighbour k (statically indexed selects: the lists stay in registers)
        int mine = bp[0];
#pragma unroll
        for (int k = 1; k < 5; ++k)
            if (hl == k) mine = bp[k];
        if (active && hl < 5) {
            const bool have = mine >= 0;
            const float4 pt = have ? pool[mine] : make_float4(0.f, 0.f, 0.f, 0.f);
            float* r = nb.rec + static_cast<size_t>(i) * 16 + 3 * hl;
            r[0] = pt.x, r[1] = pt.y, r[2] = pt.z;
        }
        if (active && hl == 5) nb.rec[static_cast<size_t>(i) * 16 + 15] = bp[4] >= 0 ? __uint_as_float(static_cast<unsigned>(bk[4] >> 32)) : 3.4e38f;
    }
}
SCAL_KERNEL(256, k_assoc_knn)

// The neighbours of a residual-block slot become factor parameters: PCA of the five neighbours for an edge candidate (:594-622),
// plane fit for a surf candidate (:651-687).  ~3k dependent f64 operations per slot.
struct AssocFit {
    CSoA4 cs, ss;
    NNBuf nb;
    const MapCounters* C;
    FactorSoA f;
    __device__ __forceinline__ void operator()(int first, int stride, int n_slots) const {
        const int nc = C->n_corner_stack, ns = C->n_surf_stack;
        for (int i = first; i < nc + ns && i < f.cap; i += stride) fit(i, nc);
    }
    __device__ __forceinline__ void fit(int i, int nc) const {
        int valid = 0;
        const bool is_edge = i < nc;
        const int j = is_edge ? i : i - nc;
        const float ox = is_edge ? cs.x[j] : ss.x[j], oy = is_edge ? cs.y[j] : ss.y[j], oz = is_edge ? cs.z[j] : ss.z[j];
        double pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
        const float4* rec = reinterpret_cast<const float4*>(nb.rec + static_cast<size_t>(i) * 16);
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
        if (static_cast<double>(r3.w) < 1.0) {  // :585 / :653
            const float px[5] = {r0.x, r0.w, r1.z, r2.y, r3.x}, py[5] = {r0.y, r1.x, r1.w, r2.z, r3.y}, pz[5] = {r0.z, r1.y, r2.x, r2.w, r3.z};
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
                // No finiteness test, as in the reference: a degenerate fit (n = 0 -> nrm = 0, d = inf, n/0 = NaN) passes the
                // 0.2 m test because NaN comparisons are false (:673-676), the NaN block reaches the solver, every step of the
                // solve is then invalid and the pose stays as it was (Ceres: "Residual and Jacobian evaluation failed").
                if (ok) {
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
};

// one THREAD per stack point (the PCA / plane fit is ~3k dependent f64 operations, so it wants every lane busy with a different
// point and many small workgroups spread over the machine; inside the solve kernel's 64 workgroups it ran slower)
__device__ __forceinline__ void k_assoc_fit_body(const AssocFit& fit, const MapState* __restrict__ S) {
    if (S->abort || !fit.C->solve_on) return;
    const int nc = fit.C->n_corner_stack, ns = fit.C->n_surf_stack;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int surf_block = nc / 64 + 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) SCAL_STAMP(24);
    if (threadIdx.x == 0 && static_cast<int>(blockIdx.x) == surf_block) SCAL_STAMP(26);
    if (i < nc + ns && i < fit.f.cap) fit.fit(i, nc);
    if (threadIdx.x == 0 && blockIdx.x == 0) SCAL_STAMP(25);
    if (threadIdx.x == 0 && static_cast<int>(blockIdx.x) == surf_block) SCAL_STAMP(27);
}
SCAL_KERNEL(64, k_assoc_fit)

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
//   key layout (36 sorted bits = 3 passes of 12): [slot 9: 0..74 valid cube, 127 = not re-filtered][vz 9][vy 9][vx 9]; dropped points get ~0
__device__ __forceinline__ void k_insert_keys_body(const MapCloud& m, const MapState* __restrict__ S, const CSoA4& stack, const int* __restrict__ d_nstack,
                                                     const LMState* __restrict__ st, float inv_leaf, int cap, unsigned long long* __restrict__ keys,
                                                     int* __restrict__ vals, MapCounters* C, int cls) {
    const MapParams mp = S->mp;
    const int n_old = S->n_map[cls];
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
SCAL_KERNEL(256, k_insert_keys)

__device__ __forceinline__ void k_map_heads_body(const SortedPairs& sp, const int* __restrict__ d_n, int* __restrict__ blockcnt) {
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
SCAL_KERNEL(256, k_map_heads)

__device__ __forceinline__ void k_map_reduce_body(const SortedPairs& sp, const int* __restrict__ d_n, const int* __restrict__ blockoff, const MapCloud& in,
                                                    const MapCloud& out) {
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
SCAL_KERNEL(256, k_map_reduce)

// ---------------------------------------------------------------------------------------------- merge insert
// The re-filter of :738-802 without sorting the map.  After a re-filter every valid cube holds one point per voxel in key
// order, and re-filtering such points alone returns them unchanged ((0 + x) / 1 in f32).  So when the cube window did not
// move since the previous scan, only the scan's new points need sorting (<= MERGE_MAX, one workgroup); they are then merged
// into the old sequence: a new voxel run whose key equals an old point's joins that point's centroid (old point first, then
// the new points in arrival order - exactly the order the stable full sort would give), the other runs are inserted.
// Anything unusual - old keys not strictly increasing inside a valid cube (a centroid rounded across a voxel face, a cube
// that collected unfiltered points while it was outside the 5x5x3 window), an old point outside the window, too many new
// points - raises MapCounters::merge_fail: k_merge_write then stops the chain and the host redoes the insertion with the full sort.
// 16384 new points per class and scan: the HDL-64 surf stacks of config #2 hold 6.5-8.5 k points depending on the world (seed 205 stays
// just below 8192, other seeds do not - with the round-2 limit of 8192 every step of such a sequence fell back to the full sort).
constexpr int MERGE_MAX = 16384;
constexpr int MERGE_IDX_BITS = 14;
constexpr int MERGE_SAMPLES = 4096;  // old keys staged in LDS for the two-level lookup

struct MergeNew {            // per class, MERGE_MAX entries
    float *x, *y, *z, *w;    // new points in the map frame, arrival order
    int* cube;
    unsigned long long* pkey;    // (key << 13 | arrival index) in arrival order, ~0 = outside the cube window (k_map_pose_done)
    unsigned long long* sorted;  // the same, ascending; n_eff entries
    unsigned long long* samp;    // [MERGE_MAX / 8] key of every 8th sorted entry (top level of k_merge_write's search)
    int* pre;                    // inserted (unmatched) runs that start before sorted position t, counted inside t's 512-block
    int* lb;                     // run heads: number of old points in front of the run
    unsigned char* hm;           // bit 0: run head, bit 1: run joins an old point
    int* blocktot;               // [MERGE_LOOKUP_BLOCKS] inserted runs per 512-block of sorted positions
};
struct MergeArgs {
    MapCloud in[2], out[2];
    CSoA4 stack[2];
    const int* d_ns[2];
    float inv_leaf[2];
    unsigned long long* okeys[2];  // keys of the old points
    MergeNew nw[2];
    int cap;
    int2* grid_cell[2];            // cell headers of the neighbour grid: counts restored to zero here (saves a launch)
    const int* grid_rank[2];
    // prebuild = 1 (speculative chain): k_merge_write bins every point it writes into the cell grid of the NEXT step (the other grid
    // set, all zero on entry) and stores the key the point has now - the window of a queued step cannot move (k_map_begin stops the
    // chain if it would), so the next step starts with its grid, its old keys and its counts in place instead of a 20 us walk over
    // the map (k_grid_build, which remains for the first queued step behind a general one)
    int prebuild;
    int2* next_cell[2];
    int* next_rank[2];
    float4* next_pool[2];
    unsigned long long* next_okeys[2];
    int next_cap[2];
};
// Launch shapes (the map sizes are device words, so the old points are walked with block-stride loops):
//   k_merge_keys   1024 threads: blocks [0, KB0) old corner points, [KB0, KB0 + KB1) old surf points, then MERGE_CHUNKS blocks
//                  per class that sort the new keys
//   k_merge_lookup 512 threads, MERGE_CHUNKS blocks per class
//   k_merge_write  256 threads: blocks [0, WB0) / [WB0, WB0 + WB1) old points, then MERGE_MAX / 256 blocks per class for the new runs
constexpr int MERGE_CHUNKS = MERGE_MAX / 512;
constexpr int MERGE_KB0 = 24, MERGE_KB1 = 88;
constexpr int MERGE_KEYS_GRID = 2 * MERGE_CHUNKS;
constexpr int MERGE_NEW_BLOCKS = MERGE_MAX / 256;
constexpr int MERGE_WB0 = 256, MERGE_WB1 = 1536;
constexpr int MERGE_WRITE_GRID = MERGE_WB0 + MERGE_WB1 + 2 * MERGE_NEW_BLOCKS;

static_assert(MAP_NOMERGE_SLOT == MAP_NOMERGE, "one constant");

// Old points: their keys, the sortedness check, the zero invariant of the cell grid.  New points: every sorting block stages ALL
// new keys of its class in LDS (written by k_map_pose_done), sorts the 512-chunks in registers (one wave each), then ranks the
// keys of ITS chunk against the other chunks with binary searches - rank = sorted position, the keys are distinct because they
// carry the arrival index - and scatters them.  No merge network across workgroups, no second launch.
// keys of the old points + sortedness check as a launch of their own: the general path, whose three-launch grid build does not
// compute them (the speculative chain's k_grid_build does)
__device__ __forceinline__ void k_merge_okeys_body(const MergeArgs& a, const MapState* __restrict__ S, MapCounters* C) {
    if (S->abort) return;
    const MapParams mp = S->mp;
    int b = blockIdx.x;
    const int cls = b < MERGE_KB0 ? 0 : 1;
    const int nblk = cls ? MERGE_KB1 : MERGE_KB0;
    if (cls) b -= MERGE_KB0;
    const MapCloud m = a.in[cls];
    const int n_old = S->n_map[cls];
    for (int i = b * 1024 + threadIdx.x; i < n_old; i += nblk * 1024) {
        const unsigned long long k = map_key(mp, a.inv_leaf[cls], m.x[i], m.y[i], m.z[i], m.cube[i], C);
        a.okeys[cls][i] = k;
        bool bad = k == ~0ull;
        if (i > 0) {
            const unsigned long long kp = map_key(mp, a.inv_leaf[cls], m.x[i - 1], m.y[i - 1], m.z[i - 1], m.cube[i - 1], C);
            bad |= kp > k || (kp == k && !key_nomerge(k));
        }
        if (bad) C->merge_fail = 1;
    }
}
SCAL_KERNEL(1024, k_merge_okeys)

__device__ __forceinline__ void k_merge_keys_body(const MergeArgs& a, const LMState* __restrict__ st, MapState* S, MapCounters* C) {
    if (S->abort) return;
    // what the previous step's merge write left for this step has been taken over (k_map_begin, or the evaluation inside the first
    // k_assoc_knn, which leaves the clearing to a kernel behind all its workgroups): cleared in front of this step's merge write
    if (blockIdx.x == 0 && threadIdx.x == 0) S->next_valid[0] = 0, S->next_valid[1] = 0, S->next_unsorted = 0, S->next_over = 0;
    extern __shared__ __align__(16) unsigned long long sk[];  // MERGE_MAX packed keys
    __shared__ int s_scan[17];
    int b = blockIdx.x;
    const int cls = b / MERGE_CHUNKS, tile = b % MERGE_CHUNKS;
    const int ns = *a.d_ns[cls];
    const int n_new = min(ns, MERGE_MAX);
    const int nc = (n_new + 511) >> 9;
    if (tile == 0 && threadIdx.x == 0 && (ns > MERGE_MAX || S->n_map[cls] + ns > a.cap || !S->window_same)) C->merge_fail = 1;
    if (tile > 0 && tile >= nc) return;
    const MergeNew nw = a.nw[cls];
    int mine = 0;
    for (int i = threadIdx.x; i < nc * 512; i += 1024) {
        const unsigned long long pk = i < n_new ? nw.pkey[i] : ~0ull;
        mine += pk != ~0ull;
        sk[i] = pk;
    }
    int n_eff = 0;
    block_exclusive_scan(mine, s_scan, &n_eff);  // also the barrier after the fill
    if (tile == 0 && threadIdx.x == 0) C->merge_neff[cls] = n_eff;
    if (nc == 0) return;
    for (int w = wave_id(); w < nc; w += 16) {  // 16 waves, up to MERGE_CHUNKS chunks
        unsigned long long v[8];
        chunk_load(sk, w, v);
        wave_sort512(v);
        chunk_store(sk, w, v);
    }
    __syncthreads();
    // two threads per key of this block's chunk: each searches half of the other chunks
    const int j = threadIdx.x >> 1, half = threadIdx.x & 1;
    const unsigned long long key = sk[tile * 512 + j];
    int rank = half == 0 ? j : 0;  // position inside the own chunk
    for (int c = half; c < nc; c += 2) {
        if (c == tile) continue;
        const unsigned long long* ch = sk + c * 512;
        int lo = 0, hi = 512;  // lower bound: 513 possible answers, at most 10 probes
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ch[mid] < key) lo = mid + 1;
            else hi = mid;
        }
        rank += lo;
    }
    rank += __shfl_xor(rank, 1, 64);
    if (half == 0 && key != ~0ull) nw.sorted[rank] = key;
}
SCAL_KERNEL(1024, k_merge_keys)

// Run heads of the sorted new keys, each head looked up among the old keys, inserted runs counted per 512-block.
// Two-level lower bound: every `stride`-th old key is staged in LDS (one pass), the search over the samples runs at LDS
// latency and leaves a window of <= stride old keys for the last few global steps.
__device__ __forceinline__ void k_merge_lookup_body(const MergeArgs& a, const MapState* __restrict__ S, const MapCounters* __restrict__ C) {
    if (S->abort) return;
    __shared__ unsigned long long s_samp[MERGE_SAMPLES];
    __shared__ int s_scan[17];
    const int cls = blockIdx.x / MERGE_CHUNKS, blk = blockIdx.x % MERGE_CHUNKS;
    const MergeNew nw = a.nw[cls];
    const int n_eff = C->merge_neff[cls];
    if (blk * 512 >= n_eff) {
        if (threadIdx.x == 0) nw.blocktot[blk] = 0;
        return;
    }
    const int n_old = S->n_map[cls];
    const unsigned long long* okeys = a.okeys[cls];
    int stride = 32;
    while ((n_old + stride - 1) / stride > MERGE_SAMPLES) stride <<= 1;
    const int n_samp = (n_old + stride - 1) / stride;
    for (int j = threadIdx.x; j < n_samp; j += 512) s_samp[j] = okeys[static_cast<size_t>(j) * stride];
    const int t = blk * 512 + threadIdx.x;
    unsigned long long key = 0;
    bool head = false, nom = false;
    int lo = 0, hi = 0;
    if (t < n_eff) {
        key = nw.sorted[t] >> MERGE_IDX_BITS;
        nom = key_nomerge(key);
        head = nom || t == 0 || (nw.sorted[t - 1] >> MERGE_IDX_BITS) != key;
        if (head && !nom) hi = n_old;       // lower bound among the old keys
        if (nom) lo = hi = n_old;           // behind every old point (NOMERGE is the largest old key)
    }
    __syncthreads();
    if (lo < hi) {
        int x = 0, y = n_samp;  // number of samples < key
        while (x < y) {
            const int mid = (x + y) >> 1;
            if (s_samp[mid] < key) x = mid + 1;
            else y = mid;
        }
        lo = x > 0 ? (x - 1) * stride + 1 : 0;
        hi = min(n_old, x * stride);
        if (lo > hi) lo = hi;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (okeys[mid] < key) lo = mid + 1;
            else hi = mid;
        }
    }
    const bool matched = head && !nom && lo < n_old && okeys[lo] == key;
    const int ins = (head && !matched) ? 1 : 0;
    int total = 0;
    const int pre = block_exclusive_scan(ins, s_scan, &total);
    if (t < n_eff) {
        nw.hm[t] = (head ? 1 : 0) | (matched ? 2 : 0);
        nw.lb[t] = lo;
        nw.pre[t] = pre;
        if ((t & 7) == 0) nw.samp[t >> 3] = key;
    }
    if (threadIdx.x == 0) nw.blocktot[blk] = total;
}
SCAL_KERNEL(512, k_merge_lookup)

// A merge that cannot be done (merge_fail, map pool full) stops the speculative chain here: nothing has been committed yet, the
// host redoes this insertion with the full sort and replays the steps queued behind it.
__device__ __forceinline__ void merge_write_body(const MergeArgs& a, MapState* S, MapCounters* C) {
    if (S->abort) return;
    if (static_cast<int>(blockIdx.x) < MERGE_WB0 + MERGE_WB1) {
        // zero invariant of the cell grid: the blocks that walk the old points clear the cells their points were binned into -
        // first of all, also when the merge is about to be called off (the full-sort redo does not touch the grid)
        const int cls0 = static_cast<int>(blockIdx.x) < MERGE_WB0 ? 0 : 1;
        const int b0 = cls0 ? blockIdx.x - MERGE_WB0 : blockIdx.x, nblk0 = cls0 ? MERGE_WB1 : MERGE_WB0;
        const MapParams mp0 = S->mp;
        const MapCloud m0 = a.in[cls0];
        const int n0 = S->n_map[cls0];
        for (int i = b0 * 256 + threadIdx.x; i < n0; i += nblk0 * 256)
            if (a.grid_rank[cls0][i] >= 0) a.grid_cell[cls0][grid_cell(mp0, m0.x[i], m0.y[i], m0.z[i])].x = 0;
    }
    if (C->merge_fail) {
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&S->abort, static_cast<int>(MAP_ABORT_MERGE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (static_cast<int>(blockIdx.x) >= MERGE_WRITE_GRID) return;  // registration blocks of the fused form
    __shared__ int s_base[MERGE_CHUNKS + 1];
    __shared__ unsigned long long s_samp[MERGE_MAX / 8];
    int b = blockIdx.x;
    int cls, part, nblk;
    if (b < MERGE_WB0) cls = 0, part = 0, nblk = MERGE_WB0;
    else if ((b -= MERGE_WB0) < MERGE_WB1) cls = 1, part = 0, nblk = MERGE_WB1;
    else if ((b -= MERGE_WB1) < MERGE_NEW_BLOCKS) cls = 0, part = 1, nblk = MERGE_NEW_BLOCKS;
    else b -= MERGE_NEW_BLOCKS, cls = 1, part = 1, nblk = MERGE_NEW_BLOCKS;
    const MapCloud in = a.in[cls], out = a.out[cls];
    const MergeNew nw = a.nw[cls];
    const int n_eff = C->merge_neff[cls];
    const int n_old = S->n_map[cls];
    if (part == 0 && b * 256 >= n_old && b > 0) return;  // nothing to do (block 0 of each class always commits the new size)
    const int n_samp = (n_eff + 7) >> 3;
    if (part == 0)
        for (int q = threadIdx.x; q < n_samp; q += 256) s_samp[q] = nw.samp[q];
    if (threadIdx.x < MERGE_CHUNKS) s_base[threadIdx.x + 1] = nw.blocktot[threadIdx.x];
    __shared__ int s_valid;
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    if (threadIdx.x == 0) {  // inserted runs in front of every 512-block
        int run = 0;
        for (int q = 0; q < MERGE_CHUNKS; ++q) {
            const int v = s_base[q + 1];
            s_base[q] = run;
            run += v;
        }
        s_base[MERGE_CHUNKS] = run;
    }
    __syncthreads();
    const int total = s_base[MERGE_CHUNKS];
    if (n_old + total > a.cap) {  // uniform over the grid
        if (b == 0 && part == 0 && threadIdx.x == 0) __hip_atomic_store(&S->abort, static_cast<int>(MAP_ABORT_MERGE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (b == 0 && part == 0 && threadIdx.x == 0) __hip_atomic_store(&C->n_map_new[cls], n_old + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto pre_at = [&](int t) { return t < n_eff ? nw.pre[t] + s_base[t >> 9] : total; };
    const MapParams mp = S->mp;
    bool drifted = false, over = false;
    int n_in = 0;
    // the written point o goes into the next step's grid; `placed` = the key it was merged under
    auto prebuild = [&](int o, float x, float y, float z, int cube, unsigned long long placed) {
        const unsigned long long k = map_key(mp, a.inv_leaf[cls], x, y, z, cube, C);
        a.next_okeys[cls][o] = k;
        drifted |= k != placed;
        if (cube_valid(mp, cube)) {
            ++n_in;
            const int cap = a.next_cap[cls];
            const int c = grid_cell(mp, x, y, z);
            const int r = atomicAdd(&a.next_cell[cls][c].x, 1);
            a.next_rank[cls][o] = r;
            if (r < cap) {
                a.next_pool[cls][static_cast<size_t>(c) * cap + r] = make_float4(x, y, z, __int_as_float(o));
                if (r == 0) a.next_cell[cls][c].y = c * cap;
            } else {
                over = true;
            }
        } else {
            a.next_rank[cls][o] = -1;
        }
    };
    if (part == 0) {
        for (int i = b * 256 + threadIdx.x; i < n_old; i += nblk * 256) {
            const unsigned long long k = a.okeys[cls][i];
            // the point itself, fetched with its key: nothing below depends on the search
            const float ix = in.x[i], iy = in.y[i], iz = in.z[i], iw = in.w[i];
            const int cube = in.cube[i];
            int lo, hi;
            {   // two-level lower bound among the sorted new keys: every 8th key in LDS, then a window of <= 7 entries and the sample
                // behind it, fetched together (the 16-entry windows of rounds 2-3 took four dependent probes)
                int x = 0, y = n_samp;
                while (x < y) {
                    const int mid = (x + y) >> 1;
                    if (s_samp[mid] < k) x = mid + 1;
                    else y = mid;
                }
                lo = x > 0 ? (x - 1) * 8 + 1 : 0;
                hi = min(n_eff, x * 8);
                if (lo > hi) lo = hi;
            }
            unsigned long long e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = lo + j < n_eff ? nw.sorted[lo + j] : ~0ull;
            int cn = 0;  // entries of the window below k: the lower bound is lo + cn <= hi <= lo + 7
#pragma unroll
            for (int j = 0; j < 7; ++j) cn += (lo + j < hi && (e[j] >> MERGE_IDX_BITS) < k) ? 1 : 0;
            const int at = lo + cn;
            const int o = i + pre_at(at);
            // (0 + x) / 1: what the re-filter computes for a voxel holding this point alone
            float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
            ax += ix, ay += iy, az += iz, aw += iw;
            int cnt = 1;
            if (!key_nomerge(k)) {  // new points of the same voxel, arrival order: first the fetched entries, then (rarely) beyond them
                bool run = true;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j >= cn) {
                        run = run && (e[j] >> MERGE_IDX_BITS) == k;  // ~0 (past the end) never equals a key
                        if (run) {
                            const int jn = static_cast<int>(e[j] & (MERGE_MAX - 1));
                            ax += nw.x[jn], ay += nw.y[jn], az += nw.z[jn], aw += nw.w[jn];
                            ++cnt;
                        }
                    }
                if (run) {
                    int u = lo + 8;
                    while (u < n_eff && (nw.sorted[u] >> MERGE_IDX_BITS) == k) {
                        const int jn = static_cast<int>(nw.sorted[u] & (MERGE_MAX - 1));
                        ax += nw.x[jn], ay += nw.y[jn], az += nw.z[jn], aw += nw.w[jn];
                        ++u, ++cnt;
                    }
                }
            }
            const float c = static_cast<float>(cnt);
            const float ox = ax / c, oy = ay / c, oz = az / c;
            out.x[o] = ox, out.y[o] = oy, out.z[o] = oz, out.w[o] = aw / c, out.cube[o] = cube;
            if (a.prebuild) prebuild(o, ox, oy, oz, cube, k);
        }
    } else {
        const int i = b * 256 + threadIdx.x;
        if (i < n_eff && (nw.hm[i] & 3) == 1) {  // only heads of inserted runs
            const unsigned long long k = nw.sorted[i] >> MERGE_IDX_BITS;
            const int o = nw.lb[i] + pre_at(i);
            float ax = 0.f, ay = 0.f, az = 0.f, aw = 0.f;
            int u = i;
            do {
                const int j = static_cast<int>(nw.sorted[u] & (MERGE_MAX - 1));
                ax += nw.x[j], ay += nw.y[j], az += nw.z[j], aw += nw.w[j];
                ++u;
            } while (!key_nomerge(k) && u < n_eff && (nw.sorted[u] >> MERGE_IDX_BITS) == k);
            const float c = static_cast<float>(u - i);
            const int cube = nw.cube[static_cast<int>(nw.sorted[i] & (MERGE_MAX - 1))];
            const float ox = ax / c, oy = ay / c, oz = az / c;
            out.x[o] = ox, out.y[o] = oy, out.z[o] = oz, out.w[o] = aw / c;
            out.cube[o] = cube;
            if (a.prebuild) prebuild(o, ox, oy, oz, cube, k);
        }
    }
    if (!a.prebuild) return;
    if (drifted) S->next_unsorted = 1;
    if (over) S->next_over = 1;
    // one global atomic per workgroup for the count of points inside the window
    for (int o = 32; o >= 1; o >>= 1) n_in += __shfl_xor(n_in, o, 64);
    if (lane_id() == 0 && n_in) atomicAdd(&s_valid, n_in);
    __syncthreads();
    if (threadIdx.x == 0 && s_valid) atomicAdd(&S->next_valid[cls], s_valid);
}

// What the host reads per step, in pinned memory: state + counters when the pose is ready (1) and after the insertion (2)
struct MapResult {
    LMState st;
    MapCounters C1, C2;
    MapState S1, S2;
    // Written last, behind a system-scope fence, by the kernels that fill (1) and (2): the sequence number of the step.  On the
    // speculative chain the host polls these words instead of waiting for events - an event record behind a kernel is a packet of
    // its own on the chain and, for an event the host reads memory behind, a cache write-back (~6 us each, two per step).
    unsigned seq_pose, seq_done;
};

// k_merge_write.  tail.fused = 0: the merge write alone.  tail.fused = 1 (speculative chain): the launch also carries the registration
// of the full-resolution cloud (:845-849) as extra blocks - what k_transform_cloud does in a launch of its own on the general path.
// (Committing the sizes from the last block to finish was tried as well: a ticket counter bumped by 2,300 workgroups costs more
// than the 5 us launch of k_map_end it saves.)
struct MergeTail {
    int fused;
    CSoA4 full;
    const int* d_nfull;
    int full_cap;
    const LMState* st;
    SoA4 full_out;
};
__device__ __forceinline__ void k_merge_write_body(const MergeArgs& a, MapState* S, MapCounters* C, const MergeTail& t) {
    merge_write_body(a, S, C);
    if (!t.fused) return;
    if (static_cast<int>(blockIdx.x) >= MERGE_WRITE_GRID && !S->abort) {
        const int i = (blockIdx.x - MERGE_WRITE_GRID) * 256 + threadIdx.x;
        if (i < min(*t.d_nfull, t.full_cap)) {
            double x7[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) x7[k] = t.st->x[k];
            float sel[3];
            associate_to_map(x7, t.full.x[i], t.full.y[i], t.full.z[i], sel);
            t.full_out.x[i] = sel[0], t.full_out.y[i] = sel[1], t.full_out.z[i] = sel[2], t.full_out.w[i] = t.full.w[i];
        }
    }
}
SCAL_KERNEL(256, k_merge_write)

__device__ __forceinline__ void k_transform_cloud_body(const CSoA4& in, const int* __restrict__ d_n, int cap, const LMState* __restrict__ st,
                                                         const MapState* __restrict__ S, const SoA4& out) {
    if (S->abort) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*d_n, cap)) return;
    double x7[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) x7[k] = st->x[k];
    float sel[3];
    associate_to_map(x7, in.x[i], in.y[i], in.z[i], sel);
    out.x[i] = sel[0], out.y[i] = sel[1], out.z[i] = sel[2], out.w[i] = in.w[i];
}
SCAL_KERNEL(256, k_transform_cloud)

// Start of a device-resident step in ONE launch: clears the per-scan counters and copies the lessSharp cloud (xyzi records)
// and the lessFlat cloud (SoA) of a features context into corner_in / surf_in.  Blocks [0,nbc) corner, the rest surf.
// surf_parts != null: every surf block stores the bounding box of its points there (stands in for k_vox_bbox, see VoxelFilter::run).
__device__ __forceinline__ void k_map_gather_body(const float* __restrict__ less_aos, const int* __restrict__ d_n_less, const CSoA4& less_flat,
                                                    const int* __restrict__ d_n_less_flat, const SoA4& corner_in, const SoA4& surf_in, MapCounters* C, int cap,
                                                    int nbc, unsigned* surf_parts) {
    const bool corner = static_cast<int>(blockIdx.x) < nbc;
    const int b = corner ? blockIdx.x : blockIdx.x - nbc;
    const int n = min(corner ? *d_n_less : *d_n_less_flat, cap);
    const int i = b * 256 + threadIdx.x;
    if (blockIdx.x == 0) {  // every counter except the two input counts, which have exactly one writer each below
        int* w = reinterpret_cast<int*>(C);
        for (int t = 2 + threadIdx.x; t < static_cast<int>(sizeof(MapCounters) / sizeof(int)); t += 256) w[t] = 0;
    }
    if (i == 0) (corner ? C->n_corner_in : C->n_surf_in) = n;
    if (corner) {
        if (i >= n) return;
        const float4 p = reinterpret_cast<const float4*>(less_aos)[i];
        corner_in.x[i] = p.x, corner_in.y[i] = p.y, corner_in.z[i] = p.z, corner_in.w[i] = p.w;
    } else {
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < n) {
            x = less_flat.x[i], y = less_flat.y[i], z = less_flat.z[i];
            surf_in.x[i] = x, surf_in.y[i] = y, surf_in.z[i] = z, surf_in.w[i] = less_flat.w[i];
        }
        if (surf_parts) vox_bbox_block_store(surf_parts, b, i < n, x, y, z);
    }
}
SCAL_KERNEL(256, k_map_gather)

__device__ __forceinline__ void k_export_valid_body(const MapCloud& m, const MapState* __restrict__ S, int cls, int* __restrict__ counter,
                                                      float* __restrict__ out, int cap) {
    const MapParams mp = S->mp;
    const int n = S->n_map[cls];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (cube_valid(mp, m.cube[i])) {
            const int p = atomicAdd(counter, 1);
            if (p < cap) reinterpret_cast<float4*>(out)[p] = make_float4(m.x[i], m.y[i], m.z[i], m.w[i]);
        }
}
SCAL_KERNEL(256, k_export_valid)

// Eigen-equivalent quaternion helpers, storage (x,y,z,w); the same code on both sides of the launch

__device__ __forceinline__ void copy_words(void* dst, const void* src, int bytes) {
    const unsigned* s = static_cast<const unsigned*>(src);
    unsigned* d = static_cast<unsigned*>(dst);
    for (int i = threadIdx.x; i < bytes / 4; i += blockDim.x) d[i] = s[i];
}

// Epilogue of the second solve (lm_dev.hpp, Post hook), with the final pose in hand.  Block 0: transformUpdate (:149-153) on the
// device, then pose, statistics and state go to the host slot.  All threads: the insertion is prepared - the scan's stack points
// in the map frame (:740 / :764), their cube and their packed sort key, once, for every consumer of the merge insert.
struct PoseDoneNew {
    CSoA4 stack[2];
    const int* d_ns[2];
    float inv_leaf[2];
    float *x[2], *y[2], *z[2], *w[2];
    int* cube[2];
    unsigned long long* pkey[2];
};
struct MapPoseDone {
    int active;  // the hook runs behind the second outer iteration only
    unsigned seq;  // the step's sequence number, published in MapResult::seq_pose when the host slot is complete
    MapState* S;
    const LMState* st;
    MapCounters* C;
    MapResult* host;
    PoseDoneNew nw;
    __device__ __forceinline__ void operator()(const double* x, int first, int stride, bool aborted) const {
        if (!active) return;
        const bool stop = aborted || S->abort;
        if (!stop) {
            const MapParams mp = S->mp;
            double x7[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) x7[k] = x[k];
#pragma unroll
            for (int cls = 0; cls < 2; ++cls) {
                const int n_new = min(*nw.d_ns[cls], MERGE_MAX);
                const CSoA4 stack = nw.stack[cls];
                for (int i = first; i < n_new; i += stride) {
                    float sel[3];
                    associate_to_map(x7, stack.x[i], stack.y[i], stack.z[i], sel);
                    const int pc = pack_cube(cube_abs(sel[0]), cube_abs(sel[1]), cube_abs(sel[2]));
                    const unsigned long long k = map_key(mp, nw.inv_leaf[cls], sel[0], sel[1], sel[2], pc, C);
                    nw.x[cls][i] = sel[0], nw.y[cls][i] = sel[1], nw.z[cls][i] = sel[2], nw.w[cls][i] = stack.w[i], nw.cube[cls][i] = pc;
                    nw.pkey[cls][i] = k == ~0ull ? ~0ull : ((k << MERGE_IDX_BITS) | static_cast<unsigned long long>(i));
                }
            }
        }
        if (blockIdx.x != 0) return;  // block 0 only from here on
        if (first == 0 && !stop) {
            // q_wmap_wodom = q_w_curr * q_wodom_curr^-1 ; t_wmap_wodom = t_w_curr - q_wmap_wodom * t_wodom_curr
            const double* q_wodom = S->q_wodom;
            const double n2 = q_wodom[0] * q_wodom[0] + q_wodom[1] * q_wodom[1] + q_wodom[2] * q_wodom[2] + q_wodom[3] * q_wodom[3];
            const double qi[4] = {-q_wodom[0] / n2, -q_wodom[1] / n2, -q_wodom[2] / n2, q_wodom[3] / n2};
            double qn[4];
            m_qmul(x, qi, qn);
            double r2[3];
            m_rot(qn, S->t_wodom, r2);
#pragma unroll
            for (int i = 0; i < 4; ++i) S->q_wmap_wodom[i] = qn[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) S->t_wmap_wodom[i] = x[4 + i] - r2[i];
        }
        __syncthreads();
        copy_words(&host->st, st, sizeof(LMState));
        copy_words(&host->C1, C, sizeof(MapCounters));
        copy_words(&host->S1, S, sizeof(MapState));
        __threadfence_system();  // every thread's part of the slot is out before ...
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&host->seq_pose, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // ... the word the host polls
    }
    __device__ __forceinline__ void operator()(int, int, int) const {}
};

// the same work as a launch of its own (Ceres-adapter mode: the caller's solver produced the pose)
__device__ __forceinline__ void k_map_pose_done_body(const MapPoseDone& pd) {
    pd(pd.st->x, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, false);
}
SCAL_KERNEL(256, k_map_pose_done)

// End of a step: the new map sizes are committed (unless the speculative chain was stopped), counters and state go to the host.
__device__ __forceinline__ void k_map_end_body(MapState* S, const MapCounters* C, MapResult* host, unsigned seq) {
    if (threadIdx.x == 0 && !S->abort) S->n_map[0] = C->n_map_new[0], S->n_map[1] = C->n_map_new[1];
    __syncthreads();
    copy_words(&host->C2, C, sizeof(MapCounters));
    copy_words(&host->S2, S, sizeof(MapState));
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&host->seq_done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
SCAL_KERNEL(256, k_map_end)

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
    int n = 0;  // host mirror of the point count (valid after scal_map_finish)
    MapCloud cloud(int b) { return MapCloud{pts[b].x.p, pts[b].y.p, pts[b].z.p, pts[b].w.p, cube[b].p}; }
};

// Two sets, one per map buffer parity: set p describes the map in buffers p.  On the speculative chain the merge write that fills
// buffers p ^ 1 bins its output into set p ^ 1 while it restores the zero invariant of set p.
struct GridStore {
    DevBuf<int2> cell[2];
    DevBuf<int> rank[2];
    DevBuf<float4> g;          // compact point pool of the three-launch build (general path: one step at a time)
    DevBuf<float4> fixed[2];   // one-launch build: `fixed_cap` entries per cell
    DevBuf<unsigned long long> okeys[2];  // merge-insert keys of the map points
    int fixed_cap = 0;
    GridPts pts(int par, bool fixed_pool = false) { return GridPts{fixed_pool ? fixed[par].p : g.p}; }
};

SCAL_DEFINE_STAMP_READER(scal_debug_stamps_map)
}  // namespace scal

using namespace scal;
#include <deque>

// filter voxels (leaf `leaf`, lattice = multiples of the leaf) that can intersect a 1 m cell, per axis
int voxels_per_cell_axis(float leaf) {
    const double inv = 1.0 / static_cast<double>(leaf);
    int most = 1;
    for (int c = 0; c < 1000; ++c) {  // 1e-4: the f32 leaf is off its decimal value by a few 1e-8, times c
        const int lo = static_cast<int>(std::floor(c * inv + 1e-4)), hi = static_cast<int>(std::floor((c + 1) * inv - 1e-4));
        most = std::max(most, hi - lo + 1);
    }
    return most;
}


// A step in flight.  Steps are queued speculatively ("fast": merge insert, window unchanged, nothing read back) behind each other;
// a step that meets a case the fast chain does not cover stops the chain on the device (MapState::abort) and is redone by
// the host on the general path, after which the steps queued behind it are replayed.
struct MapStep {
    scal_features_t* feat = nullptr;  // null: inputs came from host arrays
    MapPoseIn pose{};
    int set = 0;          // input set (corner_in / surf_in / stacks / counters)
    int slot = 0;         // result slot + events
    int par = 0;          // map buffer parity before this step's insertion
    bool prefetched = false, have_full = false, fast = false;
    bool pose_collected = false;  // the caller has the pose
    bool confirmed = false;       // the insertion is known to have completed; a step leaves the queue when both hold
    bool failed = false;          // the LM solve was abandoned twice (MAP_ABORT_LM): nothing of this step was committed
    bool prebuilt = false;        // the previous step's merge write left this step's cell grid, keys and counts in place
    unsigned seq = 0;             // set at every (re)launch: what the step's kernels publish in MapResult::seq_pose / seq_done
    unsigned feat_generation = 0; // run of `feat` this step was enqueued for: a replay must find the same scan in the context
    int n_corner_bound = 0, n_surf_bound = 0;
    int insert_path = 0;
};

struct scal_map {
    scal_map_config cfg;
    int lane = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;          // side stream for scal_map_prefetch_features (lazily acquired)
    int side_lane = 2;
    static constexpr int NSETS = 8;      // input sets: steps in flight (<= MAX_STEPS) + prefetches queued ahead (<= MAX_PF) + 1
    static constexpr int MAX_STEPS = 4;  // steps whose insertion has not been confirmed
    static constexpr int MAX_PF = 3;
    static constexpr int NSLOTS = MAX_STEPS + 1;
    struct Prefetch {
        scal_features_t* feat;
        int set;
        unsigned generation;  // run of `feat` the inputs were taken from
    };
    std::mutex pf_mu;  // prefetches may come from a second host thread
    Prefetch pf[MAX_PF];
    int n_pf = 0;
    Prefetch pf_half[MAX_PF];  // scal_map_prefetch_begin done, scal_map_prefetch_finish still to come (FIFO)
    int n_half = 0;
    int next_set = 0;  // ring allocation of input sets
    hipEvent_t ev_pre[NSETS] = {};    // side stream: surf stack ready
    hipEvent_t ev_pre_a[NSETS] = {};  // features stream: corner stack ready
    std::deque<MapStep> steps;
    hipEvent_t ev_pose[NSLOTS] = {}, ev_done[NSLOTS] = {};
    int next_slot = 0;
    PinBuf<MapResult> res;
    // Ceres-adapter mode (scal_map_adapter_begin ... _finish): the step whose solve the caller drives
    bool adapter_active = false;
    MapStep adapter_step;
    DevBuf<int> bl_live, bl_rowoff, bl_counts;
    DevBuf<double> d_x7, d_res, d_jac, d_blocks;
    PinBuf<int> h_counts;
    BlockList block_list() { return BlockList{bl_live.p, bl_rowoff.p, bl_counts.p}; }
    int n_fast = 0, n_general = 0, n_recover_pose = 0, n_recover_insert = 0;  // scal_map_get_path_counters
    bool poll_on_enqueue = true;  // scal_map_set_poll
    int cur = 0;                // parity of the map buffers once every queued step has been inserted
    bool initialised = false;   // at least one step went through the general path
    // map <- odometry correction after the last step whose insertion is confirmed: what an abandoned solve is rolled back to
    double good_q[4] = {0, 0, 0, 1}, good_t[3] = {0, 0, 0};
    int scan_cap = 0, map_cap = 0, slot_cap = 0;
    // host mirrors (laserMapping.cpp:110-120), refreshed when a pose is collected
    double q_wmap_wodom[4] = {0, 0, 0, 1}, t_wmap_wodom[3] = {0, 0, 0};
    bool have_mp = false;
    // device
    DevBuf<MapState> d_S;
    DevBuf<float> aos;  // upload staging
    DevBuf<float> export_buf;  // map-sized staging of scal_map_export / _export_all (allocated by the first export)
    SoAStore corner_in2[NSETS], surf_in2[NSETS], corner_stack2[NSETS], surf_stack2[NSETS];
    SoAStore full_in, full_out;
    SoAStore& corner_in(int st) { return corner_in2[st]; }
    SoAStore& surf_in(int st) { return surf_in2[st]; }
    SoAStore& corner_stack(int st) { return corner_stack2[st]; }
    SoAStore& surf_stack(int st) { return surf_stack2[st]; }
    VoxelFilter vf, vf_side, vf_corner;  // main stream / prefetch (surf, side stream) / prefetch (corner, behind stage A): no shared scratch
    MapStore map[2];  // corner, surf
    GridStore grid[2];
    HostStage hs_reg;               // pinned landing area of the registered cloud (host-array entry points, first use)
    bool grid_fixed = false;        // both fixed pools exist: speculative steps build the grid in one launch
    unsigned seq_counter = 0;       // MapStep::seq of the last launch
    bool grid_prebuilt = false;     // the last queued step was a speculative one: its merge write leaves the next step's grid
    int grid_cap_now[2] = {0, 0};   // <= fixed_cap (scal_map_debug_set_grid_cap lowers it to force the overflow path in tests)
    RadixSort sorter;
    DevBuf<unsigned long long> keys;
    DevBuf<int> vals, blockcnt;
    // merge insert
    bool merge_insert = true;
    int last_insert_path = 0;  // 0 full sort, 1 merge
    SoAStore mnew[2];
    DevBuf<int> mcube[2], mpre[2], mlb[2], mblocktot[2];
    DevBuf<unsigned long long> msorted[2], mpkey[2], msamp[2];
    DevBuf<unsigned char> mhm[2];
    MergeNew merge_new(int k) {
        return MergeNew{mnew[k].x.p, mnew[k].y.p, mnew[k].z.p, mnew[k].w.p, mcube[k].p, mpkey[k].p, msorted[k].p, msamp[k].p, mpre[k].p, mlb[k].p, mhm[k].p, mblocktot[k].p};
    }
    DevBuf<int> fvalid, fkind;
    DevBuf<double> fcp, fpa, fpb, partials;
    DevBuf<LMSync> lm_sync;
    DevBuf<float> nnrec;
    NNBuf nnbuf() { return NNBuf{nnrec.p, slot_cap}; }
    DevBuf<LMState> d_st;
    DevBuf<MapCounters> d_C2[NSETS];
    DevBuf<unsigned> surf_parts[NSETS];  // per-block bounding boxes of surf_in (written by the prefetch's gather)
    DevBuf<MapCounters>& d_C(int st) { return d_C2[st]; }
    DevBuf<int> d_nfull, d_done;
    PinBuf<MapCounters> h_C;
    PinBuf<MapState> h_S;
    PinBuf<int> h_misc;
    FactorSoA factors() { return FactorSoA{fvalid.p, fkind.p, fcp.p, fpa.p, fpb.p, slot_cap}; }
    int alloc_set() {
        const int st = next_set;
        next_set = (next_set + 1) % NSETS;
        return st;
    }
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
        A(c->surf_parts[k].alloc((size_t)6 * (div_up(c->scan_cap, 256) + 1)));
    }
    A(c->full_in.alloc(sc)); A(c->full_out.alloc(sc));
    A(c->vf.init(c->scan_cap));
    A(c->vf_side.init(c->scan_cap));
    A(c->vf_corner.init(c->scan_cap));
    c->vf.set_tag(".C"), c->vf_side.set_tag(".C"), c->vf_corner.set_tag(".C");
    for (int k = 0; k < 2; ++k) {
        for (int b = 0; b < 2; ++b) {
            A(c->map[k].pts[b].alloc(mc));
            A(c->map[k].cube[b].alloc(mc));
        }
        for (int par = 0; par < 2; ++par) { A(c->grid[k].cell[par].alloc(GCELLS)); A(c->grid[k].rank[par].alloc(mc)); A(c->grid[k].okeys[par].alloc(mc)); }
        A(c->grid[k].g.alloc(mc));
    }
    A(c->sorter.init(c->map_cap));
    A(c->keys.alloc(mc)); A(c->vals.alloc(mc)); A(c->blockcnt.alloc(div_up(c->map_cap, 256) + 1));
    for (int k = 0; k < 2; ++k) {
        A(c->mnew[k].alloc(MERGE_MAX)); A(c->mcube[k].alloc(MERGE_MAX)); A(c->mpre[k].alloc(MERGE_MAX + 1)); A(c->mlb[k].alloc(MERGE_MAX));
        A(c->msorted[k].alloc(MERGE_MAX)); A(c->mhm[k].alloc(MERGE_MAX)); A(c->mblocktot[k].alloc(MERGE_CHUNKS));
        A(c->mpkey[k].alloc(MERGE_MAX)); A(c->msamp[k].alloc(MERGE_MAX / 8));
    }
    A(c->fvalid.alloc(sc)); A(c->fkind.alloc(sc)); A(c->fcp.alloc(3 * sc)); A(c->fpa.alloc(3 * sc)); A(c->fpb.alloc(3 * sc));
    A(c->nnrec.alloc(16 * sc));
    A(c->partials.alloc(LM_PARTIAL_WORDS));
    A(c->lm_sync.alloc(1));
    A(c->d_st.alloc(1)); A(c->d_S.alloc(1)); A(c->d_nfull.alloc(4)); A(c->d_done.alloc(1));
    A(c->bl_live.alloc(sc)); A(c->bl_rowoff.alloc(sc + 1)); A(c->bl_counts.alloc(2)); A(c->h_counts.alloc(2)); A(c->d_x7.alloc(8));
    A(c->d_res.alloc(3 * sc)); A(c->d_jac.alloc(21 * sc)); A(c->d_blocks.alloc(10 * sc));
    A(c->h_C.alloc(1)); A(c->h_S.alloc(1)); A(c->h_misc.alloc(4)); A(c->res.alloc(scal_map::NSLOTS));
    // Optional, and therefore LAST (after every mandatory buffer, so that a device where they do not fit still gets a working context
    // with the three-launch build): fixed-slice pools of the one-launch grid build, two sets of 16 B x voxels per cell x 9.4 M cells = 10.5 GB at
    // the reference's 0.4 / 0.8 m, sparsely touched.  SCALOAM_MAP_FIXED_GRID=0 switches the shortcut (and its footprint) off.
    const char* fixed_env = std::getenv("SCALOAM_MAP_FIXED_GRID");
    if (rc == SCAL_OK && !(fixed_env && fixed_env[0] == '0')) {
        int cap[2];
        for (int k = 0; k < 2; ++k) {
            const int a = voxels_per_cell_axis(k == 0 ? c->cfg.line_res : c->cfg.plane_res);
            cap[k] = a * a * a;
        }
        if (cap[0] + cap[1] <= 48) {  // 10.5 GB at the reference's 0.4 / 0.8 m; finer filters keep the three-launch build
            c->grid_fixed = true;
            for (int k = 0; k < 2 && c->grid_fixed; ++k) {
                for (int par = 0; par < 2; ++par)
                    if (c->grid[k].fixed[par].alloc(static_cast<size_t>(GCELLS) * cap[k]) != SCAL_OK) c->grid_fixed = false;  // not fatal: no memory, no shortcut
                c->grid[k].fixed_cap = c->grid_cap_now[k] = cap[k];
            }
            if (!c->grid_fixed) {
                (void)hipGetLastError();
                for (int k = 0; k < 2; ++k) c->grid[k].fixed[0].release(), c->grid[k].fixed[1].release();
            }
        }
    }
    if (rc == SCAL_OK) std::memset(static_cast<void*>(c->res.p), 0, sizeof(MapResult) * scal_map::NSLOTS);  // seq_pose / seq_done: no step has reported
    c->lane = stage_lane(STAGE_MAP);
    if (rc == SCAL_OK) rc = lm_check_residency<AssocFit, MapPoseDone>(c->cfg.device);
    // per device, hence here and not behind a process-wide flag at the first launch
    if (rc == SCAL_OK && hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_keys), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(sizeof(unsigned long long) * MERGE_MAX)) != hipSuccess)
        rc = SCAL_E_HIP;
    if (rc == SCAL_OK && acquire_stream(c->cfg.device, &c->stream, c->lane) != SCAL_OK) rc = SCAL_E_HIP;
    // the prefetch's side stream is fixed HERE, under the stream mode the context is created in (a pipeline restores the caller's mode
    // after creating its contexts: a lane looked up at the first prefetch would be a fifth busy stream - measured: -20 % throughput)
    if (rc == SCAL_OK && acquire_stream(c->cfg.device, &c->side, c->side_lane = stage_lane(STAGE_MAP_PREFETCH)) != SCAL_OK) rc = SCAL_E_HIP;
    for (int k = 0; k < scal_map::NSLOTS && rc == SCAL_OK; ++k) {
        if (hipEventCreateWithFlags(&c->ev_pose[k], hipEventDisableTiming) != hipSuccess) rc = SCAL_E_HIP;
        if (rc == SCAL_OK && hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming) != hipSuccess) rc = SCAL_E_HIP;
    }
    if (rc == SCAL_OK) {
        // Everything is initialised on the context's own stream (the legacy null stream is not ordered against it).
        // The cell counters obey a zero invariant: every step clears exactly the cells it touched.
        for (int k = 0; k < 4 && rc == SCAL_OK; ++k) rc = c->grid[k & 1].cell[k >> 1].zero(c->stream);
        if (rc == SCAL_OK && op_memset_async(c->d_st.p, 0, sizeof(LMState), c->stream) != hipSuccess) rc = SCAL_E_HIP;
        if (rc == SCAL_OK && op_memset_async(c->lm_sync.p, 0, sizeof(LMSync), c->stream) != hipSuccess) rc = SCAL_E_HIP;
        if (rc == SCAL_OK) rc = c->partials.zero(c->stream);  // sequence number 0 = never published
        if (rc == SCAL_OK) rc = c->d_done.zero(c->stream);
        MapState& H = *c->h_S.p;
        std::memset(&H, 0, sizeof H);
        H.q_wmap_wodom[3] = 1.0;
        H.mp.cenW = 10, H.mp.cenH = 10, H.mp.cenD = 5;  // :74-76
        if (rc == SCAL_OK && op_memcpy_async(c->d_S.p, c->h_S.p, sizeof(MapState), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = SCAL_E_HIP;
        if (rc == SCAL_OK && op_stream_synchronize(c->stream) != hipSuccess) rc = SCAL_E_HIP;
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
        (void)op_stream_synchronize(c->stream);
        release_stream(c->cfg.device, c->lane);
    }
    if (c->side) {
        (void)op_stream_synchronize(c->side);
        release_stream(c->cfg.device, c->side_lane);
    }
    for (int k = 0; k < scal_map::NSETS; ++k) {
        if (c->ev_pre[k]) (void)hipEventDestroy(c->ev_pre[k]);
        if (c->ev_pre_a[k]) (void)hipEventDestroy(c->ev_pre_a[k]);
    }
    for (int k = 0; k < scal_map::NSLOTS; ++k) {
        if (c->ev_pose[k]) (void)hipEventDestroy(c->ev_pose[k]);
        if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
    }
    delete c;
}

namespace {

// stack downsample (:543-551) of corner_in / surf_in into corner_stack / surf_stack; independent of the pose.  The filters'
// verdicts land in the counters from inside their last kernels.
// surf_box_done: the gather has already folded the surf cloud's bounding box into vfs.meta.
// Key bits: a stack cloud spans the sensor's range, <= 1024 voxels per axis at the launch files' resolutions -> three radix passes.
int enqueue_corner_filter(scal_map* c, VoxelFilter& vf, hipStream_t s, int n_corner_bound, int st) {
    MapCounters* C = c->d_C(st).p;
    VoxTail tc;
    tc.err_out = &C->error;
    const int bits_c = c->cfg.line_res >= 0.35f ? 30 : 40;
    return vf.run(s, c->corner_in(st).cv(), &C->n_corner_in, n_corner_bound, c->cfg.line_res, bits_c, c->corner_stack(st).v(), &C->n_corner_stack, &tc);
}
int enqueue_surf_filter(scal_map* c, VoxelFilter& vf, hipStream_t s, int n_surf_bound, int st, bool surf_box_done) {
    MapCounters* C = c->d_C(st).p;
    VoxTail ts;
    ts.err_out = &C->error;
    const int bits_s = c->cfg.plane_res >= 0.35f ? 30 : 40;
    const bool box = surf_box_done && n_surf_bound > 8192;  // the gather stored per-block boxes of the surf cloud
    return vf.run(s, c->surf_in(st).cv(), &C->n_surf_in, n_surf_bound, c->cfg.plane_res, bits_s, c->surf_stack(st).v(), &C->n_surf_stack, &ts,
                  box ? c->surf_parts[st].p : nullptr, box ? std::max(1, div_up(n_surf_bound, 256)) : 0);
}
int enqueue_stack_filters(scal_map* c, VoxelFilter& vfc, VoxelFilter& vfs, hipStream_t s, int n_corner_bound, int n_surf_bound, int st,
                          bool surf_box_done) {
    SCAL_TRY(enqueue_corner_filter(c, vfc, s, n_corner_bound, st));
    return enqueue_surf_filter(c, vfs, s, n_surf_bound, st, surf_box_done);
}

GridArgs grid_args(scal_map* c, int par, bool fixed_pool = false) {
    GridArgs ga;
    for (int k = 0; k < 2; ++k) {
        GridStore& G = c->grid[k];
        ga.m[k] = c->map[k].cloud(par);
        ga.cell[k] = G.cell[par].p, ga.rank[k] = G.rank[par].p, ga.g[k] = G.pts(par, fixed_pool);
        ga.okeys[k] = G.okeys[par].p;
        ga.inv_leaf[k] = 1.0f / (k == 0 ? c->cfg.line_res : c->cfg.plane_res);
    }
    return ga;
}

// insert + re-filter (:738-802) by a full stable sort of the pool; general path only: n_map[] = host copy of the map sizes
int insert_full_sort(scal_map* c, const MapStep& e, const int* n_map) {
    hipStream_t s = c->stream;
    MapCounters* C = c->d_C(e.set).p;
    LMState* st = c->d_st.p;
    for (int k = 0; k < 2; ++k) {
        MapStore& M = c->map[k];
        const int n_tot_max = std::min(c->map_cap, n_map[k] + c->scan_cap);
        const int nb = std::max(1, div_up(n_tot_max, 256));
        MapCloud in = M.cloud(e.par), outc = M.cloud(e.par ^ 1);
        const CSoA4 stack = k == 0 ? c->corner_stack(e.set).cv() : c->surf_stack(e.set).cv();
        const int* d_ns = k == 0 ? &C->n_corner_stack : &C->n_surf_stack;
        SCAL_LAUNCH("k_insert_keys", k_insert_keys, dim3(nb), dim3(256), 0, s, in, c->d_S.p, stack, d_ns, st, 1.0f / (k == 0 ? c->cfg.line_res : c->cfg.plane_res),
                           c->map_cap, c->keys.p, c->vals.p, C, k);
        SortedPairs sp;
        SCAL_TRY(c->sorter.sort(s, c->keys.p, c->vals.p, &C->n_total[k], n_tot_max, 36, nullptr, &sp));
        SCAL_LAUNCH("k_map_heads", k_map_heads, dim3(nb), dim3(256), 0, s, sp, &C->n_total[k], c->blockcnt.p);
        launch_scan_inplace(s, c->blockcnt.p, &C->n_total[k], 256, 1, &C->n_map_new[k]);
        SCAL_LAUNCH("k_map_reduce", k_map_reduce, dim3(nb), dim3(256), 0, s, sp, &C->n_total[k], c->blockcnt.p, in, outc);
    }
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

// merge insert of the step's stack points (k_merge_*): no host-side sizes, stops the chain (MapState::abort) if it cannot be done.
// fused: registration of the full-resolution cloud, commit and host copy ride in the last launch (see k_merge_write)
int launch_insert_merge(scal_map* c, const MapStep& e, bool fused = false) {
    hipStream_t s = c->stream;
    MapCounters* C = c->d_C(e.set).p;
    MergeArgs a;
    for (int k = 0; k < 2; ++k) {
        MapStore& M = c->map[k];
        a.in[k] = M.cloud(e.par), a.out[k] = M.cloud(e.par ^ 1);
        a.stack[k] = k == 0 ? c->corner_stack(e.set).cv() : c->surf_stack(e.set).cv();
        a.d_ns[k] = k == 0 ? &C->n_corner_stack : &C->n_surf_stack;
        a.inv_leaf[k] = 1.0f / (k == 0 ? c->cfg.line_res : c->cfg.plane_res);
        GridStore& G = c->grid[k];
        a.okeys[k] = G.okeys[e.par].p;
        a.nw[k] = c->merge_new(k);
        a.grid_cell[k] = G.cell[e.par].p, a.grid_rank[k] = G.rank[e.par].p;
        a.next_cell[k] = G.cell[e.par ^ 1].p, a.next_rank[k] = G.rank[e.par ^ 1].p, a.next_pool[k] = G.fixed[e.par ^ 1].p;
        a.next_okeys[k] = G.okeys[e.par ^ 1].p, a.next_cap[k] = c->grid_cap_now[k];
    }
    a.prebuild = fused && c->grid_fixed ? 1 : 0;
    a.cap = c->map_cap;
    const int lds = sizeof(unsigned long long) * MERGE_MAX;  // attribute set per device in scal_map_create
    // the old points' keys come with the one-launch grid (k_grid_build, or the previous step's merge write); wherever the grid was
    // built by the three general launches - the general path, and queued steps of a context without fixed pools (filters finer than
    // 48 voxels per cell pair) - they are computed here
    if (!(fused && c->grid_fixed)) SCAL_LAUNCH("k_merge_okeys", k_merge_okeys, dim3(MERGE_KB0 + MERGE_KB1), dim3(1024), 0, s, a, c->d_S.p, C);
    SCAL_LAUNCH("k_merge_keys", k_merge_keys, dim3(MERGE_KEYS_GRID), dim3(1024), lds, s, a, c->d_st.p, c->d_S.p, C);
    SCAL_LAUNCH("k_merge_lookup", k_merge_lookup, dim3(2 * MERGE_CHUNKS), dim3(512), 0, s, a, c->d_S.p, C);
    MergeTail t{};
    int grid = MERGE_WRITE_GRID;
    if (fused) {
        t.fused = 1, t.st = c->d_st.p, t.full_out = c->full_out.v(), t.full_cap = c->scan_cap;
        if (e.have_full && e.feat) {
            FeatDeviceView v = features_view(e.feat);
            t.full = CSoA4{v.x, v.y, v.z, v.i}, t.d_nfull = &v.P->n_kept;
            grid += std::max(1, div_up(c->scan_cap, 256));
        } else if (e.have_full) {
            t.full = c->full_in.cv(), t.d_nfull = c->d_nfull.p;
            grid += std::max(1, div_up(c->scan_cap, 256));
        } else {
            t.full = c->full_in.cv(), t.d_nfull = c->d_nfull.p, t.full_cap = 0;
        }
    }
    SCAL_LAUNCH("k_merge_write", k_merge_write, dim3(grid), dim3(256), 0, s, a, c->d_S.p, C, t);
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

// registration of the full-resolution cloud (:845-849), commit of the new map sizes, counters to the host; fires ev_done
int launch_tail(scal_map* c, const MapStep& e) {
    hipStream_t s = c->stream;
    if (e.have_full) {
        const int nb = std::max(1, div_up(c->scan_cap, 256));
        if (e.feat) {
            FeatDeviceView v = features_view(e.feat);
            SCAL_LAUNCH("k_transform_cloud", k_transform_cloud, dim3(nb), dim3(256), 0, s, CSoA4{v.x, v.y, v.z, v.i}, &v.P->n_kept, c->scan_cap, c->d_st.p, c->d_S.p,
                               c->full_out.v());
        } else {
            SCAL_LAUNCH("k_transform_cloud", k_transform_cloud, dim3(nb), dim3(256), 0, s, c->full_in.cv(), c->d_nfull.p, c->scan_cap, c->d_st.p, c->d_S.p,
                               c->full_out.v());
        }
    }
    SCAL_LAUNCH("k_map_end", k_map_end, dim3(1), dim3(256), 0, s, c->d_S.p, c->d_C(e.set).p, c->res.p + e.slot, e.seq);
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(op_event_record(c->ev_done[e.slot], s));
    if (e.feat) SCAL_TRY(features_note_reader(e.feat, s));  // the registration transform reads the full-resolution cloud last
    return SCAL_OK;
}

MapPoseDone make_pose_done(scal_map* c, const MapStep& e) {
    MapCounters* C = c->d_C(e.set).p;
    MapPoseDone pd;
    pd.active = 1, pd.seq = e.seq;
    pd.S = c->d_S.p, pd.st = c->d_st.p, pd.C = C, pd.host = c->res.p + e.slot;
    for (int k = 0; k < 2; ++k) {
        pd.nw.stack[k] = k == 0 ? c->corner_stack(e.set).cv() : c->surf_stack(e.set).cv();
        pd.nw.d_ns[k] = k == 0 ? &C->n_corner_stack : &C->n_surf_stack;
        pd.nw.inv_leaf[k] = 1.0f / (k == 0 ? c->cfg.line_res : c->cfg.plane_res);
        pd.nw.x[k] = c->mnew[k].x.p, pd.nw.y[k] = c->mnew[k].y.p, pd.nw.z[k] = c->mnew[k].z.p, pd.nw.w[k] = c->mnew[k].w.p;
        pd.nw.cube[k] = c->mcube[k].p, pd.nw.pkey[k] = c->mpkey[k].p;
    }
    return pd;
}

// Everything of a step up to the pose: inputs (unless prefetched), k_map_begin, cell grids, 2 x (association + solve), transformUpdate.
// ev_pose fires when pose and statistics have reached the host slot.
int launch_pose_part(scal_map* c, const MapStep& e, bool prepare_only = false) {
    hipStream_t s = c->stream;
    const int st_ = e.set;
    MapCounters* C = c->d_C(st_).p;
    LMState* st = c->d_st.p;
    MapState* S = c->d_S.p;
    if (e.feat) {
        if (e.prefetched) {
            // inputs gathered and downsampled: the surf cloud on the side stream, the corner cloud behind stage A - the side stream
            // recorded its event behind a wait for the corner filter's (scal_map_prefetch_finish)
            SCAL_HIP(op_stream_wait_event(s, c->ev_pre[st_], 0));
        } else {
            FeatDeviceView v = features_view(e.feat);
            SCAL_TRY(features_wait_done(e.feat, s));
            const int nbc = std::max(1, div_up(e.n_corner_bound, 256));
            SCAL_LAUNCH("k_map_gather", k_map_gather, dim3(nbc + std::max(1, div_up(e.n_surf_bound, 256))), dim3(256), 0, s, v.less_xyzi, &v.P->n_less_sharp,
                               CSoA4{v.lfx, v.lfy, v.lfz, v.lfi}, &v.P->n_less_flat, c->corner_in(st_).v(), c->surf_in(st_).v(), C, c->scan_cap, nbc,
                               static_cast<unsigned*>(nullptr));
        }
    }
    if (!e.prefetched) SCAL_TRY(enqueue_stack_filters(c, c->vf, c->vf, s, e.n_corner_bound, e.n_surf_bound, st_, false));
    MapBeginArgs mb;
    mb.active = 0, mb.in = e.pose, mb.allow_window_change = e.fast ? 0 : 1, mb.inv_line = 1.0f / c->cfg.line_res, mb.inv_plane = 1.0f / c->cfg.plane_res;
    mb.slot_cap = c->slot_cap, mb.prebuilt = e.prebuilt ? 1 : 0, mb.S = S;
    // a queued step whose grid the previous merge write left has no launch between its start and the first neighbour search: the
    // start rides on that launch
    const bool begin_in_knn = e.fast && e.prebuilt && c->grid_fixed && !prepare_only;
    if (!begin_in_knn) SCAL_LAUNCH("k_map_begin", k_map_begin, dim3(1), dim3(64), 0, s, mb, st, C);
    // cell grids over the valid cubes (both classes per launch): one launch on the speculative chain, three in general
    const bool fixed_pool = e.fast && c->grid_fixed && !prepare_only;
    const GridArgs ga = grid_args(c, e.par, fixed_pool);
    if (fixed_pool) {
        if (!e.prebuilt) SCAL_LAUNCH("k_grid_build", k_grid_build, dim3(GRID_BLOCKS), dim3(256), 0, s, ga, c->grid_cap_now[0], c->grid_cap_now[1], S, C);
    } else {
        SCAL_LAUNCH("k_grid_count", k_grid_count, dim3(GRID_BLOCKS), dim3(256), 0, s, ga, S, C);
        SCAL_LAUNCH("k_grid_alloc", k_grid_alloc, dim3(GRID_BLOCKS), dim3(256), 0, s, ga, S, C);
        SCAL_LAUNCH("k_grid_fill", k_grid_fill, dim3(GRID_BLOCKS), dim3(256), 0, s, ga, S);
    }
    if (prepare_only) {  // Ceres-adapter mode: the caller drives association and solve (scal_map_associate / scal_map_eval_blocks)
        SCAL_HIP(hipGetLastError());
        return SCAL_OK;
    }
    // two outer iterations (:563)
    FactorSoA F = c->factors();
    const int assoc_blocks = std::max(1, std::min(3072, div_up(c->slot_cap, 8)));
    MapPoseDone pd = make_pose_done(c, e);
    const AssocFit fit{c->corner_stack(st_).cv(), c->surf_stack(st_).cv(), c->nnbuf(), C, F};
    for (int outer = 0; outer < 2; ++outer) {
        {
            mb.active = (begin_in_knn && outer == 0) ? 1 : 0;
            SCAL_LAUNCH(mb.active ? "k_assoc_knn.0" : "k_assoc_knn", k_assoc_knn, dim3(assoc_blocks), dim3(256), 0, s, c->corner_stack(st_).cv(), c->surf_stack(st_).cv(), S,
                             c->grid[0].cell[e.par].p, c->grid[0].pts(e.par, fixed_pool), c->grid[1].cell[e.par].p, c->grid[1].pts(e.par, fixed_pool), st, C, c->nnbuf(), mb);
        }
        // PCA / plane fit of every slot (by the thread that evaluates it), the solve and (second iteration) transformUpdate + the host
        // copy + the insertion keys: one launch.  (The fit had a launch of its own while it took 17 us at one thread per slot; at 3 us
        // - closed-form eigenpair, register-only QR - it fits in front of the first evaluation.)
        pd.active = outer == 1;
        launch_lm_solve(s, F, &C->n_slots, st, &C->solve_on, c->partials.p, c->lm_sync.p, outer, &S->abort, fit, pd, "k_lm_solve_map");
    }
    SCAL_HIP(hipGetLastError());
    if (!e.fast) SCAL_HIP(op_event_record(c->ev_pose[e.slot], s));  // a queued step announces its pose through MapResult::seq_pose
    return SCAL_OK;
}

// The host's view of a queued (speculative) step: the two words its kernels publish last.  A blocking wait spins on the word; while
// it does, it checks now and then that the stream still has work - a stream that has gone idle without the word means the kernels
// never ran (a launch error): SCAL_E_HIP instead of a hang.
static bool step_flag_set(const volatile unsigned* w, unsigned seq) {
    if (*w != seq) return false;
    std::atomic_thread_fence(std::memory_order_acquire);
    return true;
}
static int step_flag_wait(scal_map* c, const volatile unsigned* w, unsigned seq) {
    if (g_recorder) SCAL_HIP(op_flush_for_wait(__FILE__, __LINE__));  // launches still sitting in this thread's list would never run
    for (unsigned spins = 0;; ++spins) {
        if (step_flag_set(w, seq)) return SCAL_OK;
        if ((spins & 0xfff) == 0xfff && hipStreamQuery(c->stream) == hipSuccess) {
            if (step_flag_set(w, seq)) return SCAL_OK;
            set_error("scal_map: the stream is idle and the step never reported (a kernel launch failed?)");
            return SCAL_E_HIP;
        }
        __builtin_ia32_pause();
    }
}
static int wait_pose(scal_map* c, const MapStep& e) {
    if (e.fast) return step_flag_wait(c, &c->res.p[e.slot].seq_pose, e.seq);
    SCAL_HIP(op_event_synchronize(c->ev_pose[e.slot]));
    return SCAL_OK;
}
static int wait_done(scal_map* c, const MapStep& e) {
    if (e.fast) return step_flag_wait(c, &c->res.p[e.slot].seq_done, e.seq);
    SCAL_HIP(op_event_synchronize(c->ev_done[e.slot]));
    return SCAL_OK;
}
static bool done_ready(scal_map* c, const MapStep& e) {
    if (e.fast) return step_flag_set(&c->res.p[e.slot].seq_done, e.seq);
    return op_event_query(c->ev_done[e.slot]) == hipSuccess;
}

int report_device_error(scal_map* c, int err) {
    set_error("scal_map_step: device capacity exceeded (map pool of %d points per class, or a voxel outside its cube)", c->map_cap);
    return err;
}

// ---- an abandoned LM solve (lm_dev.hpp: a workgroup ran out of polls - another process hogging the machine).  The solve raised
// LMSync::abandoned and MapState::abort = MAP_ABORT_LM itself: the step's insertion and everything queued behind it drained as
// no-ops, nothing was committed.  The host clears the exchange, rolls the map <- odometry correction back to the last confirmed
// step (workgroup 0 may have applied transformUpdate before some other workgroup gave up) and runs the step once more; a second
// failure is reported to the caller (SCAL_E_HIP, "LM solve abandoned") and the scan is dropped - the map never sees it.
constexpr int MAP_RC_LM = 1000;  // internal: launch sequence fine, solve abandoned
bool lm_gave_up(const MapResult& R) { return R.st.termination == 5 || R.S1.abort == MAP_ABORT_LM; }
int lm_reset(scal_map* c, bool restore_pose) {
    hipStream_t s = c->stream;
    SCAL_HIP(op_stream_synchronize(s));
    SCAL_HIP(op_memset_async(c->lm_sync.p, 0, sizeof(LMSync), s));
    SCAL_TRY(c->partials.zero(s));
    SCAL_HIP(op_memset_async(&c->d_S.p->abort, 0, sizeof(int), s));
    if (restore_pose) {
        double* h = reinterpret_cast<double*>(c->h_S.p);  // pinned scratch
        for (int i = 0; i < 4; ++i) h[i] = c->good_q[i];
        for (int i = 0; i < 3; ++i) h[4 + i] = c->good_t[i];
        static_assert(offsetof(MapState, t_wmap_wodom) == offsetof(MapState, q_wmap_wodom) + 4 * sizeof(double), "pose fields are contiguous");
        SCAL_HIP(op_memcpy_async(c->d_S.p->q_wmap_wodom, h, 7 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    SCAL_HIP(op_stream_synchronize(s));
    return SCAL_OK;
}

int general_insert(scal_map* c, MapStep& e);
// A speculative step's merge write has binned the map for a successor on the chain; a step that builds its own grid (general path,
// Ceres-adapter mode, a changed slice size) first restores the zero invariant of that grid set.  All queued steps have finished.
int drop_prebuilt_grid(scal_map* c) {
    if (!c->grid_prebuilt) return SCAL_OK;
    SCAL_LAUNCH("k_grid_clear", k_grid_clear, dim3(GRID_BLOCKS), dim3(256), 0, c->stream, grid_args(c, c->cur), c->d_S.p);
    SCAL_HIP(hipGetLastError());
    c->grid_prebuilt = false;
    return SCAL_OK;
}
// General path, synchronous: the window may move, the insertion falls back to the full sort.  All earlier steps have finished.
// Returns MAP_RC_LM (nothing committed, exchange not yet cleared) when the solve was abandoned.
int run_general_once(scal_map* c, MapStep& e) {
    e.fast = false, e.prebuilt = false;
    e.par = c->cur;
    e.seq = ++c->seq_counter;
    SCAL_TRY(drop_prebuilt_grid(c));  // a general step starts and ends with both grid sets empty
    c->n_general++;
    SCAL_TRY(launch_pose_part(c, e));
    SCAL_HIP(op_event_synchronize(c->ev_pose[e.slot]));
    if (lm_gave_up(c->res.p[e.slot])) return MAP_RC_LM;
    return general_insert(c, e);
}
int run_general(scal_map* c, MapStep& e) {
    int rc = run_general_once(c, e);
    for (int attempt = 0; rc == MAP_RC_LM; ++attempt) {
        SCAL_TRY(lm_reset(c, true));
        // a full cell grid was left behind by the stopped pose part: restore the zero invariant before the grid is built again
        SCAL_LAUNCH("k_grid_clear", k_grid_clear, dim3(GRID_BLOCKS), dim3(256), 0, c->stream, grid_args(c, e.par), c->d_S.p);
        if (attempt == 1) {
            SCAL_HIP(op_stream_synchronize(c->stream));
            e.failed = true;
            set_error("LM solve abandoned: grid barrier timed out");
            return SCAL_E_HIP;
        }
        rc = run_general_once(c, e);
    }
    return rc;
}

// second half of the general path: the pose stands (slot S1 / C1 published), insertion + registration + commit, synchronous
int general_insert(scal_map* c, MapStep& e) {
    hipStream_t s = c->stream;
    const MapResult& R = c->res.p[e.slot];
    if (R.C1.error) return report_device_error(c, R.C1.error);
    const int n_map[2] = {R.S1.n_map[0], R.S1.n_map[1]};
    bool try_merge = c->merge_insert && R.S1.window_same && n_map[0] + MERGE_MAX <= c->map_cap && n_map[1] + MERGE_MAX <= c->map_cap;
    MapCounters* C = c->d_C(e.set).p;
    e.insert_path = 1;
    if (try_merge) {
        SCAL_TRY(launch_insert_merge(c, e));
        SCAL_LAUNCH("k_map_end", k_map_end, dim3(1), dim3(256), 0, s, c->d_S.p, C, c->res.p + e.slot, e.seq);
        SCAL_HIP(op_stream_synchronize(s));
        if (R.S2.abort) {  // a case the merge does not cover: redo with the full sort
            SCAL_HIP(op_memset_async(&c->d_S.p->abort, 0, sizeof(int), s));
            try_merge = false;
        } else if (R.C2.error) {
            return report_device_error(c, R.C2.error);
        }
    } else {
        // restore the zero invariant of the cell counters (the merge insert does it in its key kernel)
        SCAL_LAUNCH("k_grid_clear", k_grid_clear, dim3(GRID_BLOCKS), dim3(256), 0, s, grid_args(c, e.par), c->d_S.p);
    }
    if (!try_merge) {
        e.insert_path = 0;
        SCAL_TRY(insert_full_sort(c, e, n_map));
    }
    SCAL_TRY(launch_tail(c, e));
    SCAL_HIP(op_event_synchronize(c->ev_done[e.slot]));
    if (R.C2.error) return report_device_error(c, R.C2.error);
    c->cur = e.par ^ 1;
    c->initialised = true;
    return SCAL_OK;
}

// queues a step on the speculative chain: nothing is read back, the next step can be queued right behind
int launch_fast(scal_map* c, MapStep& e) {
    e.fast = true;
    e.par = c->cur;
    e.insert_path = 1;
    e.prebuilt = c->grid_prebuilt && c->grid_fixed;
    e.seq = ++c->seq_counter;  // a fresh number at every launch: a replayed step must not be taken for its stopped first run
    c->n_fast++;
    SCAL_TRY(launch_pose_part(c, e));
    SCAL_TRY(launch_insert_merge(c, e, true));
    SCAL_LAUNCH("k_map_end", k_map_end, dim3(1), dim3(256), 0, c->stream, c->d_S.p, c->d_C(e.set).p, c->res.p + e.slot, e.seq);
    SCAL_HIP(hipGetLastError());
    if (e.feat) SCAL_TRY(features_note_reader(e.feat, c->stream));  // the registration transform reads the full-resolution cloud last
    c->cur = e.par ^ 1;
    c->grid_prebuilt = c->grid_fixed;
    return SCAL_OK;
}

void confirm(scal_map* c, MapStep& e) {
    const MapResult& R = c->res.p[e.slot];
    for (int k = 0; k < 2; ++k) c->map[k].n = R.S2.n_map[k];
    for (int i = 0; i < 4; ++i) c->good_q[i] = R.S1.q_wmap_wodom[i];
    for (int i = 0; i < 3; ++i) c->good_t[i] = R.S1.t_wmap_wodom[i];
    c->last_insert_path = e.insert_path;
    e.confirmed = true;
}
// a step whose features context has been run again since it was enqueued cannot be redone: its inputs are gone
int check_generation(const MapStep& e, bool needs_inputs) {
    if (!e.feat || features_view(e.feat).generation == e.feat_generation) return SCAL_OK;
    if (!needs_inputs && !e.have_full) return SCAL_OK;
    set_error("scal_map: a stopped step has to be redone, but its features context has been run again since scal_map_enqueue_features "
              "(the caller must not reuse a features context before its step has been collected)");
    return SCAL_E_STATE;
}
void pop_done(scal_map* c) {
    while (!c->steps.empty() && c->steps.front().confirmed && c->steps.front().pose_collected) c->steps.pop_front();
}

// The speculative chain stopped on the device.  Everything queued has drained as no-ops behind the stop; the step that raised the
// flag is redone on the general path (from its start, or only its insertion), the steps behind it are queued again.
int recover(scal_map* c) {
    hipStream_t s = c->stream;
    SCAL_HIP(op_stream_synchronize(s));
    size_t origin = c->steps.size();
    bool at_pose = false;
    for (size_t i = 0; i < c->steps.size(); ++i) {
        MapStep& e = c->steps[i];
        if (e.confirmed) continue;
        const MapResult& R = c->res.p[e.slot];
        if (R.S1.abort || R.st.termination == 5) { origin = i, at_pose = true; break; }
        if (R.S2.abort) { origin = i, at_pose = false; break; }
        if (R.C2.error) return report_device_error(c, R.C2.error);
        confirm(c, e);  // in front of the stop: completed
    }
    if (origin == c->steps.size()) return SCAL_OK;  // nothing stopped
    MapStep& e = c->steps[origin];
    {
        static const bool trace = std::getenv("SCALOAM_PIPE_TIMING") != nullptr;
        static int shown = 0;
        if (trace && shown < 12) {
            const MapResult& R = c->res.p[e.slot];
            std::fprintf(stderr, "[scal_map] chain stopped: S1.abort %d S2.abort %d termination %d merge_fail %d neff %d/%d n_map %d/%d stack %d/%d window_same %d\n",
                         R.S1.abort, R.S2.abort, R.st.termination, R.C2.merge_fail, R.C2.merge_neff[0], R.C2.merge_neff[1], R.S2.n_map[0], R.S2.n_map[1],
                         R.C2.n_corner_stack, R.C2.n_surf_stack, R.S2.window_same);
            ++shown;
        }
    }
    c->cur = e.par;
    c->grid_prebuilt = false;  // whatever is redone below ends with both grid sets empty
    bool dropped = false;
    if (at_pose) {
        c->n_recover_pose++;
        const MapResult& R0 = c->res.p[e.slot];
        const bool lm = lm_gave_up(R0);
        if (lm) SCAL_TRY(lm_reset(c, true));  // clears the abort word too
        else SCAL_HIP(op_memset_async(&c->d_S.p->abort, 0, sizeof(int), s));
        // the grid of this step exists - built by its own k_grid_build (unless the window check stopped the chain before it) or left
        // by the previous step's merge write - and the kernel that clears its counters never ran.  MapState::mp is still the window
        // it was built under: a stopped k_map_begin does not store the new one.
        if (lm || R0.S1.abort == MAP_ABORT_GRID || e.prebuilt)
            SCAL_LAUNCH("k_grid_clear", k_grid_clear, dim3(GRID_BLOCKS), dim3(256), 0, s, grid_args(c, e.par), c->d_S.p);
        SCAL_TRY(check_generation(e, !e.prefetched));
        const int rc = run_general(c, e);
        if (rc != SCAL_OK && !e.failed) return rc;
        dropped = e.failed;  // abandoned again: the scan is dropped, the caller gets SCAL_E_HIP from collect
    } else {
        c->n_recover_insert++;  // the pose of this step stands; only its insertion is redone, with the full sort
        if (c->res.p[e.slot].S2.abort == MAP_ABORT_LM) SCAL_TRY(lm_reset(c, false));  // a workgroup gave up in the last round, after workgroup 0 had finished
        else SCAL_HIP(op_memset_async(&c->d_S.p->abort, 0, sizeof(int), s));
        if (c->res.p[e.slot].S2.abort == MAP_ABORT_LM)
            SCAL_LAUNCH("k_grid_clear", k_grid_clear, dim3(GRID_BLOCKS), dim3(256), 0, s, grid_args(c, e.par), c->d_S.p);
        SCAL_TRY(check_generation(e, false));
        const MapResult& R = c->res.p[e.slot];
        const int n_map[2] = {R.S2.n_map[0], R.S2.n_map[1]};  // not committed: still the sizes before the insertion
        e.fast = false, e.insert_path = 0;
        SCAL_TRY(insert_full_sort(c, e, n_map));
        SCAL_TRY(launch_tail(c, e));
        SCAL_HIP(op_event_synchronize(c->ev_done[e.slot]));
        if (R.C2.error) return report_device_error(c, R.C2.error);
        c->cur = e.par ^ 1;
    }
    if (dropped) e.confirmed = true;  // nothing to confirm: map sizes and the last good correction stay as they were
    else confirm(c, e);
    for (size_t i = origin + 1; i < c->steps.size(); ++i) {
        SCAL_TRY(check_generation(c->steps[i], !c->steps[i].prefetched));
        SCAL_TRY(launch_fast(c, c->steps[i]));
    }
    return SCAL_OK;
}

// confirms the insertions of the queued steps in order; wait = false: only those that have already finished
int confirm_steps(scal_map* c, bool wait) {
    for (size_t i = 0; i < c->steps.size(); ++i) {
        if (c->steps[i].confirmed) continue;
        const int slot = c->steps[i].slot;
        if (wait) {
            SCAL_TRY(wait_done(c, c->steps[i]));
        } else if (!done_ready(c, c->steps[i])) {
            break;
        }
        const MapResult& R = c->res.p[slot];
        if (R.S1.abort || R.S2.abort || R.st.termination == 5) {
            SCAL_TRY(recover(c));  // confirms at least this step
            if (!c->steps[i].confirmed) {
                set_error("scal_map: internal error (recovery did not complete the stopped step)");
                return SCAL_E_STATE;
            }
            continue;
        }
        if (R.C2.error) {
            confirm(c, c->steps[i]);
            return report_device_error(c, R.C2.error);
        }
        confirm(c, c->steps[i]);
    }
    return SCAL_OK;
}
int map_finish(scal_map* c) {
    SCAL_TRY(confirm_steps(c, true));
    pop_done(c);
    return SCAL_OK;
}
int map_poll(scal_map* c) {
    SCAL_TRY(confirm_steps(c, false));
    pop_done(c);
    return SCAL_OK;
}
// room for one more step: MAX_STEPS bounds the steps whose pose has not been collected; collected ones whose insertion is still
// running are waited for here
int map_make_room(scal_map* c) {
    if (c->poll_on_enqueue) SCAL_TRY(map_poll(c));
    while (static_cast<int>(c->steps.size()) >= scal_map::MAX_STEPS) {
        if (!c->steps.front().pose_collected) {
            set_error("scal_map: %d steps are queued and not collected", scal_map::MAX_STEPS);
            return SCAL_E_STATE;
        }
        SCAL_TRY(wait_done(c, c->steps.front()));
        SCAL_TRY(map_poll(c));
    }
    return SCAL_OK;
}

int new_step(scal_map* c, MapStep* out) {
    MapStep e;
    e.slot = c->next_slot;
    c->next_slot = (c->next_slot + 1) % scal_map::NSLOTS;
    *out = e;
    return SCAL_OK;
}

// queues one process() pass; `e` describes the inputs.  Speculative when the previous steps allow it, general (synchronous) otherwise.
int map_enqueue(scal_map* c, MapStep& e) {
    // room for this step's points whatever the queued ones add (host mirror + worst case per queued step)
    const int queued = static_cast<int>(c->steps.size()) + 1;
    const bool room = c->map[0].n + queued * MERGE_MAX <= c->map_cap && c->map[1].n + queued * MERGE_MAX <= c->map_cap;
    const bool fast = c->initialised && c->merge_insert && room && e.feat != nullptr;
    if (fast) {
        SCAL_TRY(launch_fast(c, e));
        c->steps.push_back(e);
        return SCAL_OK;
    }
    SCAL_TRY(map_finish(c));
    SCAL_TRY(run_general(c, e));
    c->steps.push_back(e);
    confirm(c, c->steps.back());
    return SCAL_OK;
}

// waits for the pose of the oldest uncollected step; map sizes in `stats` are those before this step's insertion (insert_path = -1)
int map_collect_pose(scal_map* c, double* q_out, double* t_out, scal_map_stats* stats) {
    MapStep* pe = nullptr;
    for (auto& e : c->steps)
        if (!e.pose_collected) {
            pe = &e;
            break;
        }
    if (!pe) {
        set_error("scal_map_collect: no step enqueued");
        return SCAL_E_STATE;
    }
    const int slot = pe->slot;
    for (int round = 0;; ++round) {
        SCAL_TRY(wait_pose(c, *pe));
        if (pe->failed || (!c->res.p[slot].S1.abort && c->res.p[slot].st.termination != 5)) break;
        if (round > scal_map::MAX_STEPS + 1) {
            set_error("scal_map_collect: internal error (recovery does not converge)");
            return SCAL_E_STATE;
        }
        SCAL_TRY(recover(c));  // redoes the stopped step on the general path and queues the ones behind it again
    }
    pe->pose_collected = true;
    const MapResult& R = c->res.p[slot];
    const MapCounters& H = R.C1;
    if (pe->failed) {  // abandoned twice (recover / run_general): the scan was dropped, the exchange is clean again
        pop_done(c);
        set_error("LM solve abandoned: grid barrier timed out");
        return SCAL_E_HIP;
    }
    if (H.error) return report_device_error(c, H.error);
    const double* xf = R.st.x;
    for (int i = 0; i < 4; ++i) q_out[i] = xf[i];
    for (int i = 0; i < 3; ++i) t_out[i] = xf[4 + i];
    for (int i = 0; i < 4; ++i) c->q_wmap_wodom[i] = R.S1.q_wmap_wodom[i];
    for (int i = 0; i < 3; ++i) c->t_wmap_wodom[i] = R.S1.t_wmap_wodom[i];
    c->have_mp = true;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->n_corner_stack = H.n_corner_stack, stats->n_surf_stack = H.n_surf_stack;
        stats->n_corner_map = H.n_valid[0], stats->n_surf_map = H.n_valid[1];
        for (int o = 0; o < 2; ++o) {
            const LMState& L = R.st;
            stats->n_edge[o] = H.solve_on ? L.log_n_edge[o] : 0, stats->n_plane[o] = H.solve_on ? L.log_n_plane[o] : 0;
            stats->lm_iters[o] = H.solve_on ? L.log_iters[o] : 0, stats->lm_success[o] = H.solve_on ? L.log_success[o] : 0;
            stats->cost_init[o] = H.solve_on ? L.log_cost_init[o] : 0.0, stats->cost_final[o] = H.solve_on ? L.log_cost_final[o] : 0.0;
        }
        stats->solved = H.solve_on;
        stats->n_map_corner_total = R.S1.n_map[0], stats->n_map_surf_total = R.S1.n_map[1];
        stats->insert_path = -1;
    }
    pop_done(c);
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
    SCAL_TRY(map_finish(c));  // synchronous entry point: nothing may be in flight
    if (!c->steps.empty()) {
        set_error("scal_map_step: a queued step has not been collected");
        return SCAL_E_STATE;
    }
    MapStep e;
    SCAL_TRY(new_step(c, &e));
    {
        std::lock_guard<std::mutex> lk(c->pf_mu);
        c->n_pf = 0, c->n_half = 0;  // queued prefetches belong to a features context: not used by this entry point
        e.set = c->alloc_set();
    }
    // per-scan counters and inputs, staged through pinned memory
    MapCounters z;
    std::memset(&z, 0, sizeof z);
    z.n_corner_in = n_corner, z.n_surf_in = n_surf;
    *c->h_C.p = z;
    SCAL_HIP(op_memcpy_async(c->d_C(e.set).p, c->h_C.p, sizeof(MapCounters), hipMemcpyHostToDevice, s));
    c->h_misc.p[0] = have_full ? n_full : 0;
    SCAL_HIP(op_memcpy_async(c->d_nfull.p, c->h_misc.p, sizeof(int), hipMemcpyHostToDevice, s));
    auto up = [&](const float* src, int n, SoAStore& dst) -> int {
        if (n > 0) {
            SCAL_HIP(op_memcpy_async(c->aos.p, src, sizeof(float) * 4 * n, hipMemcpyHostToDevice, s));
            launch_deinterleave(s, c->aos.p, n, dst.v());
            SCAL_HIP(op_stream_synchronize(s));  // the staging buffer is reused by the next upload
        }
        return SCAL_OK;
    };
    SCAL_TRY(up(corner_last, n_corner, c->corner_in(e.set)));
    SCAL_TRY(up(surf_last, n_surf, c->surf_in(e.set)));
    if (have_full) SCAL_TRY(up(full_res, n_full, c->full_in));
    for (int i = 0; i < 4; ++i) e.pose.q_wodom[i] = q_wodom[i];
    for (int i = 0; i < 3; ++i) e.pose.t_wodom[i] = t_wodom[i];
    e.have_full = have_full, e.n_corner_bound = n_corner, e.n_surf_bound = n_surf;
    SCAL_TRY(run_general(c, e));
    c->steps.push_back(e);
    confirm(c, c->steps.back());
    SCAL_TRY(map_collect_pose(c, q_w_curr, t_w_curr, stats));
    SCAL_TRY(map_finish(c));
    if (stats) {
        stats->n_map_corner_total = c->map[0].n, stats->n_map_surf_total = c->map[1].n;
        stats->insert_path = c->last_insert_path;
    }
    if (have_full && registered) {
        launch_interleave(s, c->d_nfull.p, n_full, c->full_out.cv(), c->aos.p);
        (void)c->hs_reg.reserve((size_t)c->scan_cap * 16 + 4096);
        SCAL_HIP(c->hs_reg.d2h(registered, c->aos.p, sizeof(float) * 4 * n_full, s));
        SCAL_HIP(op_stream_synchronize(s));
        c->hs_reg.finish();
    }
    return SCAL_OK;
}

// laserCloudCornerLast = lessSharp cloud, laserCloudSurfLast = lessFlat cloud (laserOdometry.cpp:554-563); the full-res cloud is
// read in place by the registration transform
extern "C" int scal_map_prefetch_begin(scal_map_t* c, scal_features_t* feat) {
    if (!c || !feat) {
        set_error("scal_map_prefetch_begin: null argument");
        return SCAL_E_ARG;
    }
    FeatDeviceView v = features_view(feat);
    if (v.device != c->cfg.device) {
        set_error("features context lives on device %d, map context on %d", v.device, c->cfg.device);
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    const int ls_cap = std::min(c->scan_cap, v.n_scans * 120);
    const int cap = std::min(c->scan_cap, v.cap);
    std::lock_guard<std::mutex> lk(c->pf_mu);
    if (c->n_pf + c->n_half >= scal_map::MAX_PF) {
        set_error("scal_map_prefetch_features: %d prefetches are already queued ahead of their steps", scal_map::MAX_PF);
        return SCAL_E_STATE;
    }
    const int nset = c->alloc_set();
    if (!c->ev_pre[nset]) {
        SCAL_HIP(hipEventCreateWithFlags(&c->ev_pre[nset], EV_DEVICE_ONLY));
        SCAL_HIP(hipEventCreateWithFlags(&c->ev_pre_a[nset], EV_DEVICE_ONLY));
    }
    // The gather (it also stores per-block bounding boxes of the surf cloud for the filter) and the small corner filter ride on the
    // features context's own stream, right behind stage A; the surf filter runs on the side stream, which it shares with
    // ScanContext's keyframe filter.  (More than four busy streams slow every stream down on this GPU; see stage_lane().)
    hipStream_t sa = v.stream;
    const int nbc = std::max(1, div_up(ls_cap, 256));
    SCAL_LAUNCH("k_map_gather", k_map_gather, dim3(nbc + std::max(1, div_up(cap, 256))), dim3(256), 0, sa, v.less_xyzi, &v.P->n_less_sharp,
                     CSoA4{v.lfx, v.lfy, v.lfz, v.lfi}, &v.P->n_less_flat, c->corner_in(nset).v(), c->surf_in(nset).v(), c->d_C(nset).p, c->scan_cap, nbc,
                     c->surf_parts[nset].p);
    SCAL_HIP(hipGetLastError());
    SCAL_TRY(enqueue_corner_filter(c, c->vf_corner, sa, ls_cap, nset));
    SCAL_HIP(op_event_record(c->ev_pre_a[nset], sa));
    c->pf_half[c->n_half].feat = feat, c->pf_half[c->n_half].set = nset, c->pf_half[c->n_half].generation = v.generation;
    c->n_half++;
    return SCAL_OK;
}

extern "C" int scal_map_prefetch_finish(scal_map_t* c, scal_features_t* feat) {
    if (!c || !feat) {
        set_error("scal_map_prefetch_finish: null argument");
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    FeatDeviceView v = features_view(feat);
    const int cap = std::min(c->scan_cap, v.cap);
    std::lock_guard<std::mutex> lk(c->pf_mu);
    if (c->n_half == 0 || c->pf_half[0].feat != feat) {
        set_error("scal_map_prefetch_finish: no scal_map_prefetch_begin of this features context is waiting (halves are finished in the order they were begun)");
        return SCAL_E_STATE;
    }
    const scal_map::Prefetch h = c->pf_half[0];
    for (int i = 1; i < c->n_half; ++i) c->pf_half[i - 1] = c->pf_half[i];
    c->n_half--;
    // ONE event from the features stream, recorded behind gather AND corner filter: the surf filter starts ~35 us later than it could
    // (it has that slack: the side lane is busy 190 of every ~217 us and the step needs its output a whole stage B later), the
    // features stream saves an event record between two kernels of the chain the pipeline is bound by, and the step waits for one
    // event that covers both halves
    SCAL_HIP(op_stream_wait_event(c->side, c->ev_pre_a[h.set], 0));
    SCAL_TRY(enqueue_surf_filter(c, c->vf_side, c->side, cap, h.set, true));
    SCAL_HIP(op_event_record(c->ev_pre[h.set], c->side));
    c->pf[c->n_pf] = h;
    c->n_pf++;
    return SCAL_OK;
}

extern "C" int scal_map_prefetch_features(scal_map_t* c, scal_features_t* feat) {
    SCAL_TRY(scal_map_prefetch_begin(c, feat));
    return scal_map_prefetch_finish(c, feat);
}

static int map_enqueue_features(scal_map* c, scal_features_t* feat, const double* q_wodom, const double* t_wodom) {
    FeatDeviceView v = features_view(feat);
    if (v.device != c->cfg.device) {
        set_error("features context lives on device %d, map context on %d", v.device, c->cfg.device);
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_TRY(map_make_room(c));
    MapStep e;
    SCAL_TRY(new_step(c, &e));
    e.feat = feat;
    e.feat_generation = v.generation;
    e.n_corner_bound = std::min(c->scan_cap, v.n_scans * 120);
    e.n_surf_bound = std::min(c->scan_cap, v.cap);
    e.have_full = true;
    for (int i = 0; i < 4; ++i) e.pose.q_wodom[i] = q_wodom[i];
    for (int i = 0; i < 3; ++i) e.pose.t_wodom[i] = t_wodom[i];
    {
        std::lock_guard<std::mutex> lk(c->pf_mu);
        if (c->n_pf > 0 && c->pf[0].feat == feat) {  // the oldest queued prefetch belongs to this step: take its set
            if (c->pf[0].generation != v.generation) {
                // the prefetched inputs are of an EARLIER scan than the one now in `feat` (its full-resolution cloud, read by the
                // registration at the end of the step, has been overwritten): refuse instead of mixing two scans
                c->n_pf = 0, c->n_half = 0;
                set_error("scal_map_enqueue_features: the features context was run again between scal_map_prefetch_features and this call");
                return SCAL_E_STATE;
            }
            e.prefetched = true;
            e.set = c->pf[0].set;
            for (int i = 1; i < c->n_pf; ++i) c->pf[i - 1] = c->pf[i];
            c->n_pf--;
        } else {  // no (matching, complete) prefetch: drop what was queued and take a fresh set
            c->n_pf = 0, c->n_half = 0;
            e.set = c->alloc_set();
        }
    }
    return map_enqueue(c, e);
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
    c->h_misc.p[1] = 0;
    SCAL_HIP(op_memcpy_async(c->d_nfull.p + 1, c->h_misc.p + 1, sizeof(int), hipMemcpyHostToDevice, s));
    // staging for the map-sized exports (the scan-sized upload staging cut a 5x5x3 window larger than one scan short): once, on demand
    if (out_xyzi && c->export_buf.n < static_cast<size_t>(c->map_cap) * 4) SCAL_TRY(c->export_buf.alloc(static_cast<size_t>(c->map_cap) * 4));
    const int room = std::min(cap, c->map_cap);
    SCAL_LAUNCH("k_export_valid", k_export_valid, dim3(std::max(1, std::min(1024, div_up(M.n, 256)))), dim3(256), 0, s, M.cloud(c->cur), c->d_S.p, which,
                       c->d_nfull.p + 1, out_xyzi ? c->export_buf.p : nullptr, out_xyzi ? room : 0);
    SCAL_HIP(op_memcpy_async(c->h_misc.p + 2, c->d_nfull.p + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    const int n = c->h_misc.p[2];
    if (out_xyzi && room > 0) {
        const int m = std::min(n, room);
        SCAL_HIP(op_memcpy_async(out_xyzi, c->export_buf.p, sizeof(float) * 4 * m, hipMemcpyDeviceToHost, s));
        SCAL_HIP(op_stream_synchronize(s));
        return m;
    }
    return n;
}

// Every point of the 21 x 21 x 11 cube grid, one feature class: what /laser_cloud_map carries (laserMapping.cpp:824-837 adds up
// laserCloudCornerArray[i] and laserCloudSurfArray[i] over all 4,851 cubes every 20 frames).  Here that is the class's whole map
// array: the next re-filter drops what has left the grid, nothing else is stored.  Order: this library's (valid cubes in voxel
// order, then the cubes outside the 5x5x3 window in arrival order), not the reference's cube-index order - a point set.
extern "C" int scal_map_export_all(scal_map_t* c, int which, float* out_xyzi, int cap) {
    if (!c || (which != 0 && which != 1) || cap < 0) {
        set_error("scal_map_export_all: bad argument");
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    if (map_finish(c) != SCAL_OK) return -1;
    MapStore& M = c->map[which];
    if (!out_xyzi || cap == 0) return M.n;
    const int m = std::min(M.n, cap);
    if (m == 0) return 0;
    hipStream_t s = c->stream;
    if (c->export_buf.n < static_cast<size_t>(c->map_cap) * 4) SCAL_TRY(c->export_buf.alloc(static_cast<size_t>(c->map_cap) * 4));
    c->h_misc.p[1] = m;
    SCAL_HIP(op_memcpy_async(c->d_nfull.p + 1, c->h_misc.p + 1, sizeof(int), hipMemcpyHostToDevice, s));
    const MapCloud mc = M.cloud(c->cur);
    launch_interleave(s, c->d_nfull.p + 1, m, CSoA4{mc.x, mc.y, mc.z, mc.w}, c->export_buf.p);
    SCAL_HIP(op_memcpy_async(out_xyzi, c->export_buf.p, sizeof(float) * 4 * m, hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    return m;
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

extern "C" int scal_map_get_path_counters(scal_map_t* c, int* out4) {
    if (!c || !out4) return SCAL_E_ARG;
    out4[0] = c->n_fast, out4[1] = c->n_general, out4[2] = c->n_recover_pose, out4[3] = c->n_recover_insert;
    return SCAL_OK;
}

extern "C" int scal_map_set_poll(scal_map_t* c, int enable) {
    if (!c) return SCAL_E_ARG;
    c->poll_on_enqueue = enable != 0;
    return SCAL_OK;
}

// ------------------------------------------------------------------------------------------------ Ceres-adapter mode
// The host keeps ceres::Problem / ceres::Solve (laserMapping.cpp:566-573, :713-721); the device does everything around it.
static int upload_step_inputs(scal_map* c, MapStep& e, const float* corner_last, int n_corner, const float* surf_last, int n_surf, const float* full_res,
                              int n_full) {
    hipStream_t s = c->stream;
    MapCounters z;
    std::memset(&z, 0, sizeof z);
    z.n_corner_in = n_corner, z.n_surf_in = n_surf;
    *c->h_C.p = z;
    SCAL_HIP(op_memcpy_async(c->d_C(e.set).p, c->h_C.p, sizeof(MapCounters), hipMemcpyHostToDevice, s));
    c->h_misc.p[0] = n_full;
    SCAL_HIP(op_memcpy_async(c->d_nfull.p, c->h_misc.p, sizeof(int), hipMemcpyHostToDevice, s));
    auto up = [&](const float* src, int n, SoAStore& dst) -> int {
        if (n > 0) {
            SCAL_HIP(op_memcpy_async(c->aos.p, src, sizeof(float) * 4 * n, hipMemcpyHostToDevice, s));
            launch_deinterleave(s, c->aos.p, n, dst.v());
            SCAL_HIP(op_stream_synchronize(s));  // the staging buffer is reused by the next upload
        }
        return SCAL_OK;
    };
    SCAL_TRY(up(corner_last, n_corner, c->corner_in(e.set)));
    SCAL_TRY(up(surf_last, n_surf, c->surf_in(e.set)));
    if (n_full > 0) SCAL_TRY(up(full_res, n_full, c->full_in));
    e.have_full = n_full > 0, e.n_corner_bound = n_corner, e.n_surf_bound = n_surf;
    return SCAL_OK;
}

extern "C" int scal_map_adapter_begin(scal_map_t* c, const float* corner_last, int n_corner, const float* surf_last, int n_surf, const float* full_res,
                                      int n_full, const double* q_wodom, const double* t_wodom, double* q_w_curr, double* t_w_curr) {
    if (!c || !q_wodom || !t_wodom || !q_w_curr || !t_w_curr || n_corner < 0 || n_surf < 0 || n_full < 0 || (n_corner > 0 && !corner_last) ||
        (n_surf > 0 && !surf_last)) {
        set_error("scal_map_adapter_begin: bad argument");
        return SCAL_E_ARG;
    }
    if (n_corner > c->scan_cap || n_surf > c->scan_cap || n_full > c->scan_cap) {
        set_error("scal_map_adapter_begin: input cloud larger than max_scan_points (%d)", c->scan_cap);
        return SCAL_E_TOO_MANY;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_TRY(map_finish(c));
    if (!c->steps.empty() || c->adapter_active) {
        set_error("scal_map_adapter_begin: a step is still open");
        return SCAL_E_STATE;
    }
    MapStep e;
    SCAL_TRY(new_step(c, &e));
    {
        std::lock_guard<std::mutex> lk(c->pf_mu);
        c->n_pf = 0, c->n_half = 0;
        e.set = c->alloc_set();
    }
    SCAL_TRY(upload_step_inputs(c, e, corner_last, n_corner, surf_last, n_surf, (full_res && n_full > 0) ? full_res : nullptr, full_res ? n_full : 0));
    for (int i = 0; i < 4; ++i) e.pose.q_wodom[i] = q_wodom[i];
    for (int i = 0; i < 3; ++i) e.pose.t_wodom[i] = t_wodom[i];
    e.fast = false, e.par = c->cur;
    SCAL_TRY(drop_prebuilt_grid(c));
    c->n_general++;
    SCAL_TRY(launch_pose_part(c, e, true));  // stack filters, transformAssociateToMap + window, cell grids
    hipStream_t s = c->stream;
    SCAL_HIP(op_memcpy_async(c->h_S.p, c->d_S.p, sizeof(MapState), hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_memcpy_async(c->res.p + e.slot, c->d_st.p, sizeof(LMState), hipMemcpyDeviceToHost, s));  // MapResult::st is its first member
    SCAL_HIP(op_stream_synchronize(s));
    const double* x = c->res.p[e.slot].st.x;
    for (int i = 0; i < 4; ++i) q_w_curr[i] = x[i];
    for (int i = 0; i < 3; ++i) t_w_curr[i] = x[4 + i];
    c->adapter_step = e;
    c->adapter_active = true;
    return SCAL_OK;
}

static int adapter_set_pose(scal_map* c, const double* q, const double* t) {
    double* h = reinterpret_cast<double*>(c->h_S.p);  // pinned scratch (MapState is larger than 7 doubles)
    for (int i = 0; i < 4; ++i) h[i] = q[i];
    for (int i = 0; i < 3; ++i) h[4 + i] = t[i];
    SCAL_HIP(op_memcpy_async(c->d_st.p->x, h, sizeof(double) * 7, hipMemcpyHostToDevice, c->stream));
    return SCAL_OK;
}

extern "C" int scal_map_associate(scal_map_t* c, const double* q_w_curr, const double* t_w_curr, int* n_blocks, int* n_residuals) {
    if (!c || !q_w_curr || !t_w_curr || !n_blocks || !n_residuals) {
        set_error("scal_map_associate: null argument");
        return SCAL_E_ARG;
    }
    if (!c->adapter_active) {
        set_error("scal_map_associate: no scal_map_adapter_begin");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const MapStep& e = c->adapter_step;
    MapCounters* C = c->d_C(e.set).p;
    SCAL_TRY(adapter_set_pose(c, q_w_curr, t_w_curr));
    SCAL_HIP(op_stream_synchronize(s));  // the pinned scratch is reused
    FactorSoA F = c->factors();
    const int assoc_blocks = std::max(1, std::min(3072, div_up(c->slot_cap, 8)));
    SCAL_LAUNCH("k_assoc_knn", k_assoc_knn, dim3(assoc_blocks), dim3(256), 0, s, c->corner_stack(e.set).cv(), c->surf_stack(e.set).cv(), c->d_S.p,
                     c->grid[0].cell[e.par].p, c->grid[0].pts(e.par), c->grid[1].cell[e.par].p, c->grid[1].pts(e.par), c->d_st.p, C, c->nnbuf(), MapBeginArgs{});
    const AssocFit fit{c->corner_stack(e.set).cv(), c->surf_stack(e.set).cv(), c->nnbuf(), C, F};
    SCAL_LAUNCH("k_assoc_fit", k_assoc_fit, dim3(std::max(1, div_up(c->slot_cap, 64))), dim3(64), 0, s, fit, c->d_S.p);
    SCAL_LAUNCH("k_blocks_compact", k_blocks_compact, dim3(1), dim3(1024), 0, s, F, &C->n_slots, c->block_list());
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(op_memcpy_async(c->h_counts.p, c->bl_counts.p, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_memcpy_async(c->h_C.p, C, sizeof(MapCounters), hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    if (!c->h_C.p->solve_on) c->h_counts.p[0] = c->h_counts.p[1] = 0;  // map too small (:555): no residual blocks, the caller skips the solve
    *n_blocks = c->h_counts.p[0], *n_residuals = c->h_counts.p[1];
    return SCAL_OK;
}

extern "C" int scal_map_get_blocks(scal_map_t* c, scal_block* out, int cap) {
    if (!c || (!out && cap > 0) || cap < 0) {
        set_error("scal_map_get_blocks: bad argument");
        return SCAL_E_ARG;
    }
    if (!c->adapter_active) {
        set_error("scal_map_get_blocks: no scal_map_adapter_begin");
        return SCAL_E_STATE;
    }
    static_assert(sizeof(scal_block) == 80, "scal_block layout");
    SCAL_HIP(hipSetDevice(c->cfg.device));
    const int n = std::min(cap, c->h_counts.p[0]);
    if (n <= 0) return 0;
    hipStream_t s = c->stream;
    SCAL_LAUNCH("k_blocks_export", k_blocks_export, dim3(div_up(c->h_counts.p[0], 256)), dim3(256), 0, s, c->factors(), c->block_list(), c->d_blocks.p);
    SCAL_HIP(op_memcpy_async(out, c->d_blocks.p, sizeof(scal_block) * n, hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    return n;
}

extern "C" int scal_map_eval_blocks(scal_map_t* c, const double* x7, int want_jac, double* residuals, double* jacobians) {
    if (!c || !x7 || !residuals || (want_jac && !jacobians)) {
        set_error("scal_map_eval_blocks: bad argument");
        return SCAL_E_ARG;
    }
    if (!c->adapter_active) {
        set_error("scal_map_eval_blocks: no scal_map_adapter_begin");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    const int nb = c->h_counts.p[0], nr = c->h_counts.p[1];
    if (nb == 0) return SCAL_OK;
    hipStream_t s = c->stream;
    SCAL_HIP(op_memcpy_async(c->d_x7.p, x7, sizeof(double) * 7, hipMemcpyHostToDevice, s));
    SCAL_LAUNCH("k_blocks_eval", k_blocks_eval, dim3(div_up(nb, 256)), dim3(256), 0, s, c->factors(), c->block_list(), c->d_x7.p, want_jac ? 1 : 0,
                     c->d_res.p, c->d_jac.p);
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(op_memcpy_async(residuals, c->d_res.p, sizeof(double) * nr, hipMemcpyDeviceToHost, s));
    if (want_jac) SCAL_HIP(op_memcpy_async(jacobians, c->d_jac.p, sizeof(double) * 7 * nr, hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    return SCAL_OK;
}

extern "C" int scal_map_adapter_finish(scal_map_t* c, const double* q_w_curr, const double* t_w_curr, float* registered, scal_map_stats* stats) {
    if (!c || !q_w_curr || !t_w_curr) {
        set_error("scal_map_adapter_finish: null argument");
        return SCAL_E_ARG;
    }
    if (!c->adapter_active) {
        set_error("scal_map_adapter_finish: no scal_map_adapter_begin");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    MapStep& e = c->adapter_step;
    c->adapter_active = false;
    SCAL_TRY(adapter_set_pose(c, q_w_curr, t_w_curr));
    // transformUpdate (:735), host copy, insertion keys: what the second solve's launch does on the all-device path
    SCAL_LAUNCH("k_map_pose_done", k_map_pose_done, dim3(64), dim3(256), 0, s, make_pose_done(c, e));
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(op_event_record(c->ev_pose[e.slot], s));
    SCAL_HIP(op_event_synchronize(c->ev_pose[e.slot]));
    SCAL_TRY(general_insert(c, e));
    const MapResult& R = c->res.p[e.slot];
    for (int i = 0; i < 4; ++i) c->q_wmap_wodom[i] = R.S1.q_wmap_wodom[i];
    for (int i = 0; i < 3; ++i) c->t_wmap_wodom[i] = R.S1.t_wmap_wodom[i];
    c->have_mp = true;
    for (int k = 0; k < 2; ++k) c->map[k].n = R.S2.n_map[k];
    c->last_insert_path = e.insert_path;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->n_corner_stack = R.C1.n_corner_stack, stats->n_surf_stack = R.C1.n_surf_stack;
        stats->n_corner_map = R.C1.n_valid[0], stats->n_surf_map = R.C1.n_valid[1];
        stats->solved = R.C1.solve_on;
        stats->n_map_corner_total = c->map[0].n, stats->n_map_surf_total = c->map[1].n;
        stats->insert_path = e.insert_path;
    }
    if (e.have_full && registered) {
        const int n_full = c->h_misc.p[0];
        launch_interleave(s, c->d_nfull.p, n_full, c->full_out.cv(), c->aos.p);
        (void)c->hs_reg.reserve((size_t)c->scan_cap * 16 + 4096);
        SCAL_HIP(c->hs_reg.d2h(registered, c->aos.p, sizeof(float) * 4 * n_full, s));
        SCAL_HIP(op_stream_synchronize(s));
        c->hs_reg.finish();
    }
    return SCAL_OK;
}

extern "C" void* scal_map_stream(scal_map_t* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

extern "C" int scal_map_debug_set_grid_cap(scal_map_t* c, int cap_corner, int cap_surf) {
    if (!c || cap_corner < 1 || cap_surf < 1) {
        set_error("scal_map_debug_set_grid_cap: bad argument");
        return SCAL_E_ARG;
    }
    if (!c->grid_fixed) return SCAL_OK;
    SCAL_TRY(map_finish(c));
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_TRY(drop_prebuilt_grid(c));  // binned under the old slice size: the next step builds its own
    c->grid_cap_now[0] = std::min(cap_corner, c->grid[0].fixed_cap);
    c->grid_cap_now[1] = std::min(cap_surf, c->grid[1].fixed_cap);
    return SCAL_OK;
}

extern "C" int scal_map_debug_set_lm_polls(scal_map_t* c, int polls) {
    if (!c || polls < 0) return SCAL_E_ARG;
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_HIP(op_stream_synchronize(c->stream));
    return lm_set_poll_budget(polls);
}
