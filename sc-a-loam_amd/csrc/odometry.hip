// Stage B on gfx950: scan-to-scan odometry.  Replaces the main-loop body of /root/reference/src/laserOdometry.cpp
// (:267-291 problem set-up, :299-384 edge correspondences, :387-483 plane correspondences, :494-506 solve and pose
// integration, :554-568 hand-over of the lessSharp / lessFlat clouds).  DISTORTION is 0 (:59): s == 1 everywhere.
//
// Kernels (one stream; one host sync per scan for the pose read-back):
//   k_odom_assoc   one WORKGROUP per query point (sharp then flat).  TransformToStart (:111-129) in f64 -> f32.
//                  The kd-tree NN(1) (:302, :390) is an exact argmin over the previous scan's cloud (f32 ((dx^2+dy^2)+dz^2),
//                  ties -> lower index) through a two-level hashed cell index that the hand-over builds (CellIndex below);
//                  the reference's sequential walks over neighbouring rings (:312-361, :402-455) become
//                  chunked wave scans: per-ring index tables give the first index past +-2.5 rings, and a 64-bit
//                  (distance bits, visit order) key reproduces "first strictly smaller wins".
//   k_odom_handover / k_odom_cellscan / k_odom_cellfill   the lessSharp / lessFlat clouds become the next scan's targets
//                  (:554-568): copy + ring tables + bucket counts, bucket offsets, placement.
//   lm_dev.hpp     the Ceres-equivalent solve, state persists across scans (para_q / para_t, :97-101).
// Clouds are SoA x[] y[] z[] intensity[] in HBM; intensity carries the ring id in its integer part (:308).
#include "common.hpp"
#include <deque>
#include "device_utils.hpp"
#include "voxel_dev.hpp"
#include "lm_dev.hpp"
#include "features_dev.hpp"
#include <cmath>
#include <algorithm>

namespace scal {

struct OdomCounters {
    // persistent across scans (previous scan's clouds, device-resident)
    int n_corner_last, n_surf_last;
    // per scan: zeroed / refreshed at the start of every step (everything from n_sharp on)
    int n_sharp, n_flat, n_less_sharp, n_less_flat;
    int n_slots, enable;
    int n_live[2], n_edge[2], n_plane[2];
};

__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    // (a.x-b.x)*(a.x-b.x) + (a.y-b.y)*(a.y-b.y) + (a.z-b.z)*(a.z-b.z), all f32 (:322-327)
    return (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz);
}

__device__ __forceinline__ unsigned long long make_key(float d, unsigned seq) {
    return (static_cast<unsigned long long>(__float_as_uint(d)) << 32) | seq;
}

constexpr int RING_TAB = 96;   // per-ring first/last index tables of a target cloud

// TransformToStart (:111-129): q_last_curr * p + t_last_curr in f64, stored to f32
__device__ __forceinline__ void transform_to_start(const double* x7, float ox, float oy, float oz, float* o) {
    double r[3];
    quat_rotate(x7, static_cast<double>(ox), static_cast<double>(oy), static_cast<double>(oz), r);
    o[0] = static_cast<float>(r[0] + 1.0 * x7[4]);
    o[1] = static_cast<float>(r[1] + 1.0 * x7[5]);
    o[2] = static_cast<float>(r[2] + 1.0 * x7[6]);
}

// NN(1) index over a target cloud: cells hashed into OD_H buckets, points stored bucket by bucket (counting sort done by the
// hand-over: count + rank in k_odom_handover, offsets in k_odom_cellscan, placement in k_odom_cellfill), at two cell sizes.  A query
// looks at the 27 buckets of the cells around it; two cells sharing a bucket only add candidates, every candidate's distance is
// computed.  A point closer than one cell edge lies in a neighbouring cell on every axis, so if the best candidate is closer than that,
// it is the exact NN of the whole cloud (ties included: they were all candidates).  Level 0 (1 m cells) answers almost every query;
// level 1 (5 m cells) covers DISTANCE_SQ_THRESHOLD = 25 (:65): a query it cannot answer has no usable neighbour at all.
constexpr int OD_H = 8192;
constexpr int OD_LEVELS = 2;
__device__ constexpr float OD_INV_CELL[OD_LEVELS] = {1.0f, 0.2f};
// squared distances below which a level's answer is exact; the margins keep the f32 rounding of the sum and of v * 0.2f clear of the edge
__device__ constexpr float OD_EXACT_D2[OD_LEVELS] = {0.99f, 24.9f};
struct CellIndex {
    const int* start[OD_LEVELS];   // [OD_H + 1]
    const float4* pts[OD_LEVELS];  // x, y, z, original index (bit pattern)
};
__device__ __forceinline__ int od_cell(float v, int level) { return static_cast<int>(floorf(v * OD_INV_CELL[level])); }
__device__ __forceinline__ unsigned od_hash(int cx, int cy, int cz) {
    return (static_cast<unsigned>(cx) * 73856093u ^ static_cast<unsigned>(cy) * 19349663u ^ static_cast<unsigned>(cz) * 83492791u) & (OD_H - 1);
}
__device__ __forceinline__ unsigned od_bucket(float x, float y, float z, int level) { return od_hash(od_cell(x, level), od_cell(y, level), od_cell(z, level)); }
// best (f32 distance bits, original index) key among the 27 buckets around (sx, sy, sz); called by all 256 threads of a workgroup
__device__ __forceinline__ unsigned long long od_cell_nn(const int* __restrict__ start, const float4* __restrict__ pts, int level, float sx, float sy, float sz,
                                                          int* s_beg, int* s_pre, unsigned long long* s_nn) {
    const int lane = lane_id(), wv = wave_id();
    if (wv == 0) {
        int b = 0, cnt = 0;
        if (lane < 27) {
            const unsigned h = od_hash(od_cell(sx, level) + lane % 3 - 1, od_cell(sy, level) + (lane / 3) % 3 - 1, od_cell(sz, level) + lane / 9 - 1);
            b = start[h];
            cnt = start[h + 1] - b;
        }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane < 27) s_beg[lane] = b, s_pre[lane + 1] = incl;
        if (lane == 0) s_pre[0] = 0;
    }
    __syncthreads();
    unsigned long long best = ~0ull;
    const int n_cand = s_pre[27];
    for (int m = threadIdx.x; m < n_cand; m += 256) {
        int j = 0;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1)
            if (j + o <= 26 && s_pre[j + o] <= m) j += o;
        const float4 p = pts[s_beg[j] + (m - s_pre[j])];
        const float dx = sx - p.x, dy = sy - p.y, dz = sz - p.z;
        float d = dx * dx;  // FLANN L2_Simple<float>
        d += dy * dy;
        d += dz * dz;
        best = min(best, make_key(d, __float_as_uint(p.w)));
    }
    best = wave_min_u64(best);
    if (lane == 0) s_nn[wv] = best;
    __syncthreads();
    best = min(min(s_nn[0], s_nn[1]), min(s_nn[2], s_nn[3]));
    __syncthreads();  // the LDS arrays may be rewritten by the next level
    return best;
}

// one workgroup per query
__device__ __forceinline__ void k_odom_assoc_body(const CSoA4& sharp, const CSoA4& flat, const CSoA4& CL, const CSoA4& SL, const LMState* __restrict__ st, OdomCounters* C,
                                                    int outer, const FactorSoA& f, const CellIndex& ci_corner, const CellIndex& ci_surf,
                                                    const int* __restrict__ ring_tab) {
    if (!C->enable) return;
    const int ns = C->n_sharp, nf = C->n_flat;
    if (blockIdx.x == 0 && threadIdx.x == 0) C->n_slots = min(ns + nf, f.cap);  // read by the LM launches that follow
    __shared__ unsigned long long s_k2[4], s_k3[4];
    const int q = blockIdx.x;  // one workgroup per query: its four waves interleave over the ring window (latency hiding)
    if (q >= ns + nf || q >= f.cap) return;
    const int lane = lane_id(), wv = wave_id();
    const bool is_edge = q < ns;
    const int j = is_edge ? q : q - ns;
    const CSoA4& Q = is_edge ? sharp : flat;
    const CSoA4& T = is_edge ? CL : SL;
    const int nT = is_edge ? C->n_corner_last : C->n_surf_last;
    double x7[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) x7[k] = st->x[k];
    const float ox = Q.x[j], oy = Q.y[j], oz = Q.z[j];
    float sel[3];
    transform_to_start(x7, ox, oy, oz, sel);
    const float sx = sel[0], sy = sel[1], sz = sel[2];

    // ---- NN(1) (:302, :390): exact argmin of FLANN's L2_Simple<float> over the target cloud, ties -> lower index (see CellIndex)
    __shared__ int s_beg[27], s_pre[28];
    __shared__ unsigned long long s_nn[4];
    const CellIndex& ci = is_edge ? ci_corner : ci_surf;
    unsigned long long best = od_cell_nn(ci.start[0], ci.pts[0], 0, sx, sy, sz, s_beg, s_pre, s_nn);
    auto dist_of = [](unsigned long long k) { return __uint_as_float(static_cast<unsigned>(k >> 32)); };
    if (best == ~0ull || !(dist_of(best) < OD_EXACT_D2[0])) {  // uniform branches: one query per workgroup
        best = od_cell_nn(ci.start[1], ci.pts[1], 1, sx, sy, sz, s_beg, s_pre, s_nn);
        if (best == ~0ull || !(dist_of(best) < OD_EXACT_D2[1])) {
            // nothing certified within 5 m: the answer only matters if some point's f32 distance lands in [24.9, 25) - the whole cloud
            // decides, four independent loads per thread and step
            best = ~0ull;
            for (int t0 = threadIdx.x; t0 < nT; t0 += 1024) {
                float txx[4], tyy[4], tzz[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = min(t0 + 256 * u, nT - 1);
                    txx[u] = T.x[t], tyy[u] = T.y[t], tzz[u] = T.z[t];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = min(t0 + 256 * u, nT - 1);  // a clamped repeat of the last point cannot change the minimum
                    const float dx = sx - txx[u], dy = sy - tyy[u], dz = sz - tzz[u];
                    float d = dx * dx;
                    d += dy * dy;
                    d += dz * dz;
                    best = min(best, make_key(d, static_cast<unsigned>(t)));
                }
            }
            best = wave_min_u64(best);
            if (lane == 0) s_nn[wv] = best;
            __syncthreads();
            best = min(min(s_nn[0], s_nn[1]), min(s_nn[2], s_nn[3]));
        }
    }
    int valid = 0;
    int kind = is_edge ? 0 : 1;
    double pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
    const float nnd = __uint_as_float(static_cast<unsigned>(best >> 32));
    if (best != ~0ull && static_cast<double>(nnd) < 25.0) {  // DISTANCE_SQ_THRESHOLD (:65, :305, :393)
        const int c = static_cast<int>(best & 0xffffffffu);
        const int id = static_cast<int>(T.w[c]);  // closestPointScanID = int(intensity) (:308, :398)
        unsigned long long k2 = ~0ull, k3 = ~0ull;
        // window bounds from the per-ring first/last index tables of this target cloud (see k_odom_handover)
        const int* first_idx = ring_tab + (is_edge ? 0 : 2 * RING_TAB);
        const int* last_idx = first_idx + RING_TAB;
        int fi = 0x7f7f7f7f, li = -1;
        for (int r = lane; r < RING_TAB; r += 64) {
            if (r >= id + 3) fi = min(fi, first_idx[r]);  // first index whose ring exceeds id + NEARBY_SCAN (:319, :405)
            if (r <= id - 3) li = max(li, last_idx[r]);   // last index whose ring is below id - NEARBY_SCAN (:345, :433)
        }
        const int hi = min(wave_min_i(fi), nT);
        const int lo = wave_max_i(li);
        // Both walks are unrolled by four with the loads hoisted: the 16 loads of a step are independent, so a step costs one
        // memory latency instead of four.
        // ---- towards increasing index: j in (c, hi)
        for (int t0 = c + 1 + lane + 64 * wv; t0 < hi; t0 += 1024) {
            float tw[4], txx[4], tyy[4], tzz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 256 * u;
                if (t < hi) tw[u] = T.w[t], txx[u] = T.x[t], tyy[u] = T.y[t], tzz[u] = T.z[t];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 256 * u;
                if (t < hi) {
                    const int rj = static_cast<int>(tw[u]);
                    const float d = sqdist(txx[u], tyy[u], tzz[u], sx, sy, sz);
                    if (static_cast<double>(d) < 25.0) {
                        const unsigned seq = static_cast<unsigned>(t - c - 1);
                        if (is_edge) {
                            if (rj > id) k2 = min(k2, make_key(d, seq));  // skip same-or-lower ring (:315)
                        } else {
                            if (rj <= id) k2 = min(k2, make_key(d, seq));  // :416
                            else k3 = min(k3, make_key(d, seq));           // :422
                        }
                    }
                }
            }
        }
        // ---- towards decreasing index: j in (lo, c)
        for (int t0 = c - 1 - lane - 64 * wv; t0 > lo; t0 -= 1024) {
            float tw[4], txx[4], tyy[4], tzz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 - 256 * u;
                if (t > lo) tw[u] = T.w[t], txx[u] = T.x[t], tyy[u] = T.y[t], tzz[u] = T.z[t];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 - 256 * u;
                if (t > lo) {
                    const int rj = static_cast<int>(tw[u]);
                    const float d = sqdist(txx[u], tyy[u], tzz[u], sx, sy, sz);
                    if (static_cast<double>(d) < 25.0) {
                        const unsigned seq = 0x40000000u + static_cast<unsigned>(c - 1 - t);  // visited after every upward candidate
                        if (is_edge) {
                            if (rj < id) k2 = min(k2, make_key(d, seq));  // skip same-or-higher ring (:341)
                        } else {
                            if (rj >= id) k2 = min(k2, make_key(d, seq));  // :444
                            else k3 = min(k3, make_key(d, seq));           // :449
                        }
                    }
                }
            }
        }
        k2 = wave_min_u64(k2);
        k3 = wave_min_u64(k3);
        if (lane == 0) s_k2[wv] = k2, s_k3[wv] = k3;
        __syncthreads();  // uniform: every wave of the block takes this branch (same query, same NN)
        k2 = min(min(s_k2[0], s_k2[1]), min(s_k2[2], s_k2[3]));
        k3 = min(min(s_k3[0], s_k3[1]), min(s_k3[2], s_k3[3]));
        auto decode = [&](unsigned long long k) {
            const unsigned seq = static_cast<unsigned>(k & 0xffffffffu);
            return (seq & 0x40000000u) ? c - 1 - static_cast<int>(seq & 0x3fffffffu) : c + 1 + static_cast<int>(seq);
        };
        if (is_edge) {
            if (k2 != ~0ull) {  // :363-383
                const int i2 = decode(k2);
                valid = 1;
                pa[0] = T.x[c], pa[1] = T.y[c], pa[2] = T.z[c];
                pb[0] = T.x[i2], pb[1] = T.y[i2], pb[2] = T.z[i2];
            }
        } else if (k2 != ~0ull && k3 != ~0ull) {  // :457-481
            const int i2 = decode(k2), i3 = decode(k3);
            valid = 1;
            const double jx = T.x[c], jy = T.y[c], jz = T.z[c];
            const double lx = T.x[i2], ly = T.y[i2], lz = T.z[i2];
            const double mx = T.x[i3], my = T.y[i3], mz = T.z[i3];
            // ljm_norm = (j - l) x (j - m), normalised (lidarFactor.hpp:64-65)
            const double ax = jx - lx, ay = jy - ly, az = jz - lz, bx = jx - mx, by = jy - my, bz = jz - mz;
            double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
            const double z = nx * nx + ny * ny + nz * nz;
            if (z > 0) {
                const double s = sqrt(z);
                nx /= s, ny /= s, nz /= s;
            }
            pa[0] = jx, pa[1] = jy, pa[2] = jz;
            pb[0] = nx, pb[1] = ny, pb[2] = nz;
        }
    }
    if (lane == 0 && wv == 0) {
        f.valid[q] = valid;
        f.kind[q] = kind;
        f.cp[q] = ox, f.cp[f.cap + q] = oy, f.cp[2 * f.cap + q] = oz;
        f.pa[q] = pa[0], f.pa[f.cap + q] = pa[1], f.pa[2 * f.cap + q] = pa[2];
        f.pb[q] = pb[0], f.pb[f.cap + q] = pb[1], f.pb[2 * f.cap + q] = pb[2];
    }
}
SCAL_KERNEL(256, k_odom_assoc)

__device__ __forceinline__ void k_odom_init_pose_body(LMState* st, int* ring_tab) {
    if (threadIdx.x == 0) {
        st->x[0] = st->x[1] = st->x[2] = 0.0, st->x[3] = 1.0;  // para_q = {0,0,0,1}, para_t = {0,0,0} (:97-98)
        st->x[4] = st->x[5] = st->x[6] = 0.0;
    }
    for (int t = threadIdx.x; t < 8 * RING_TAB; t += blockDim.x) ring_tab[t] = ((t / RING_TAB) & 1) ? -1 : 0x7f7f7f7f;
}
SCAL_KERNEL(1024, k_odom_init_pose)
// hand-over copy of both target clouds that also records, per ring id r = int(intensity), the first and last index holding r.
// The tables are double-buffered: first_idx must be pre-filled with a value >= n (0x7f7f7f7f), last_idx with -1, which the
// previous hand-over did for the set written now.  The walks of :312-361 / :402-455 stop at
// the first index past +-2.5 rings; with these tables that index is min_{r' >= id+3} first_idx[r'] (resp. the max of
// last_idx below id-3): int(intensity) never drops by more than one ring along the cloud, so no earlier index qualifies.
struct HandoverArgs {
    CSoA4 in[2];        // current lessSharp, lessFlat
    const int* d_n[2];
    SoA4 out[2];        // next scan's corner_last, surf_last
    int* d_n_out[2];
    int cap[2];
    int nb0;            // blocks of the corner part
    int* tab_write;     // [corner first | corner last | surf first | surf last] x RING_TAB, pre-filled
    int* tab_reset;     // the set the association of this scan used: re-filled here for the next hand-over
    const LMState* st;  // the solved state goes to the host slot from here (no launch of its own)
    LMState* host_st;
    const LMSync* sync; // an abandoned solve (any workgroup, this step or one queued in front of it) reaches the host as termination 5
    int* cell_cnt[2][OD_LEVELS];   // [OD_H] per target cloud and level, zero on entry (k_odom_cellscan leaves them so)
    int* cell_rank[2][OD_LEVELS];  // arrival order of point i within its bucket
};
__device__ __forceinline__ void k_odom_handover_body(const HandoverArgs& a) {
    int b = blockIdx.x;
    const int k = b < a.nb0 ? 0 : 1;
    if (k) b -= a.nb0;
    if (blockIdx.x == 0) {
        for (int t = threadIdx.x; t < 4 * RING_TAB; t += 256) a.tab_reset[t] = ((t / RING_TAB) & 1) ? -1 : 0x7f7f7f7f;
        const unsigned* src = reinterpret_cast<const unsigned*>(a.st);
        unsigned* dst = reinterpret_cast<unsigned*>(a.host_st);
        const bool abandoned = a.sync->abandoned != 0;
        constexpr int TERM_WORD = static_cast<int>(offsetof(LMState, termination) / 4);
        for (int t = threadIdx.x; t < static_cast<int>(sizeof(LMState) / 4); t += 256) dst[t] = (abandoned && t == TERM_WORD) ? 5u : src[t];
    }
    int* first_idx = a.tab_write + 2 * k * RING_TAB;
    int* last_idx = first_idx + RING_TAB;
    const CSoA4 in = a.in[k];
    const SoA4 out = a.out[k];
    const int n = min(*a.d_n[k], a.cap[k]);
    const int i = b * 256 + threadIdx.x;
    if (i == 0) *a.d_n_out[k] = n;
    if (b * 256 >= n) return;
    const int lane = lane_id();
    float x = 0, y = 0, z = 0;
    if (i < n) {
        const float w = in.w[i];
        x = in.x[i], y = in.y[i], z = in.z[i];
        out.x[i] = x, out.y[i] = y, out.z[i] = z, out.w[i] = w;
        const int r = min(max(static_cast<int>(w), 0), RING_TAB - 1);
        // consecutive points mostly share a ring: only ring boundaries (and the wave's first lane) touch the tables
        const int rp = __shfl_up(r, 1, 64);
        const int rn = __shfl_down(r, 1, 64);
        if (lane == 0 || rp != r) atomicMin(&first_idx[r], i);
        if (lane == 63 || rn != r || i == n - 1) atomicMax(&last_idx[r], i);
    }
    // bucket counts: neighbours along the cloud mostly share a cell, so one atomic per run of equal buckets in the wave
#pragma unroll
    for (int l = 0; l < OD_LEVELS; ++l) {
        const unsigned h = i < n ? od_bucket(x, y, z, l) : 0xffffffffu;
        const unsigned hp = __shfl_up(h, 1, 64);
        const unsigned long long heads = __ballot(lane == 0 || hp != h);
        const int head = 63 - __clzll(heads & (~0ull >> (63 - lane)));           // start of this lane's run
        const unsigned long long above = lane == 63 ? 0ull : heads & (~0ull << (lane + 1));
        const int next = above ? __ffsll(static_cast<long long>(above)) - 1 : 64;  // start of the next run
        int base = 0;
        if (lane == head && i < n) {
            // the run ends at the next head; the tail of the last run may hold lanes past n, which carry the sentinel and form their own run
            base = atomicAdd(&a.cell_cnt[k][l][h], next - head);
        }
        base = __shfl(base, head, 64);
        if (i < n) a.cell_rank[k][l][i] = base + (lane - head);
    }
}
SCAL_KERNEL(256, k_odom_handover)
// bucket offsets, one block per table (cloud x level): exclusive scan of the counts, which are left zero for the next hand-over
constexpr int OD_START_STRIDE = OD_H + 4;
__device__ __forceinline__ void k_odom_cellscan_body(int* cnt_all, int* start_all) {
    constexpr int PER = OD_H / 1024;
    static_assert(PER % 4 == 0, "int4 loads");
    int* cnt = cnt_all + blockIdx.x * OD_H;
    int* start = start_all + blockIdx.x * OD_START_STRIDE;
    __shared__ int s_w[16];
    int v[PER];
    int4* c4 = reinterpret_cast<int4*>(cnt + threadIdx.x * PER);
    int sum = 0;
#pragma unroll
    for (int u = 0; u < PER / 4; ++u) {
        const int4 q = c4[u];
        v[4 * u] = q.x, v[4 * u + 1] = q.y, v[4 * u + 2] = q.z, v[4 * u + 3] = q.w;
        sum += q.x + q.y + q.z + q.w;
        c4[u] = make_int4(0, 0, 0, 0);
    }
    int incl = sum;
    const int lane = lane_id(), wv = wave_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wv; ++w) base += s_w[w];
    int4* s4 = reinterpret_cast<int4*>(start + threadIdx.x * PER);
#pragma unroll
    for (int u = 0; u < PER / 4; ++u) {
        int4 o;
        o.x = base, base += v[4 * u];
        o.y = base, base += v[4 * u + 1];
        o.z = base, base += v[4 * u + 2];
        o.w = base, base += v[4 * u + 3];
        s4[u] = o;
    }
    if (threadIdx.x == 1023) start[OD_H] = base;
}
SCAL_KERNEL(1024, k_odom_cellscan)
// places the points of both target clouds bucket by bucket; same block layout as k_odom_handover
struct CellFillArgs {
    CSoA4 in[2];        // corner_last, surf_last as the hand-over wrote them
    const int* d_n[2];
    const int* rank[2][OD_LEVELS];
    const int* start[2][OD_LEVELS];
    float4* pts[2][OD_LEVELS];
    int nb0;
};
__device__ __forceinline__ void k_odom_cellfill_body(const CellFillArgs& a) {
    int b = blockIdx.x;
    const int k = b < a.nb0 ? 0 : 1;
    if (k) b -= a.nb0;
    const int i = b * 256 + threadIdx.x;
    if (i >= *a.d_n[k]) return;
    const float x = a.in[k].x[i], y = a.in[k].y[i], z = a.in[k].z[i];
    const float4 rec = make_float4(x, y, z, __uint_as_float(static_cast<unsigned>(i)));
#pragma unroll
    for (int l = 0; l < OD_LEVELS; ++l) a.pts[k][l][a.start[k][l][od_bucket(x, y, z, l)] + a.rank[k][l][i]] = rec;
}
SCAL_KERNEL(256, k_odom_cellfill)
// Start of a device-resident step in ONE launch: refreshes the per-scan counters and copies the four feature clouds of a
// features context (sharp / flat / lessSharp as xyzi records, lessFlat as SoA) into this context's SoA clouds.
// Block ranges: [0,nbs) sharp, [nbs,2nbs) flat, [2nbs,2nbs+nbf) lessSharp, the rest lessFlat.
struct OdomGatherArgs {
    const float *sharp, *flat, *less;   // xyzi records
    CSoA4 less_flat;
    const int *n_sharp, *n_flat, *n_less, *n_less_flat;
    SoA4 o_sharp, o_flat, o_less, o_less_flat;
    int slot_cap, feat_cap, cap, nbs, nbf;
};
__device__ __forceinline__ void k_odom_gather_body(const OdomGatherArgs& a, OdomCounters* C) {
    int b = blockIdx.x;
    const float* aos = nullptr;
    const int* d_n;
    int* d_n_out;
    SoA4 out;
    int cap;
    if (b < a.nbs) {
        aos = a.sharp, d_n = a.n_sharp, d_n_out = &C->n_sharp, out = a.o_sharp, cap = a.slot_cap;
        if (b == 0 && threadIdx.x == 0) {
            C->n_slots = 0, C->enable = 1;
            C->n_live[0] = C->n_live[1] = C->n_edge[0] = C->n_edge[1] = C->n_plane[0] = C->n_plane[1] = 0;
        }
    } else if (b < 2 * a.nbs) {
        b -= a.nbs, aos = a.flat, d_n = a.n_flat, d_n_out = &C->n_flat, out = a.o_flat, cap = a.slot_cap;
    } else if (b < 2 * a.nbs + a.nbf) {
        b -= 2 * a.nbs, aos = a.less, d_n = a.n_less, d_n_out = &C->n_less_sharp, out = a.o_less, cap = a.feat_cap;
    } else {
        b -= 2 * a.nbs + a.nbf, d_n = a.n_less_flat, d_n_out = &C->n_less_flat, out = a.o_less_flat, cap = a.cap;
    }
    const int n = min(*d_n, cap);
    const int i = b * 256 + threadIdx.x;
    if (i == 0) *d_n_out = n;
    if (i >= n) return;
    if (aos) {
        const float4 p = reinterpret_cast<const float4*>(aos)[i];
        out.x[i] = p.x, out.y[i] = p.y, out.z[i] = p.z, out.w[i] = p.w;
    } else {
        out.x[i] = a.less_flat.x[i], out.y[i] = a.less_flat.y[i], out.z[i] = a.less_flat.z[i], out.w[i] = a.less_flat.w[i];
    }
}
SCAL_KERNEL(256, k_odom_gather)

struct OSoA {
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

}  // namespace scal

using namespace scal;

struct scal_odom {
    scal_odom_config cfg;
    hipStream_t stream = nullptr;
    int cap = 0, feat_cap = 0, slot_cap = 0;
    int lane = 0;
    bool systemInited = false;
    // steps enqueued and not collected yet: B(k+1) only needs B(k)'s device state, so it can be queued right behind it
    static constexpr int MAX_STEPS = 4;
    struct Pending {
        int slot;
        bool solve;
    };
    std::deque<Pending> pending;
    hipEvent_t ev[MAX_STEPS] = {};
    int next_slot = 0;
    double q_w_curr[4] = {0, 0, 0, 1}, t_w_curr[3] = {0, 0, 0};  // :93-94
    DevBuf<float> aos;
    OSoA sharp, flat, less_sharp, less_flat;  // current scan
    OSoA corner_last, surf_last;              // previous scan (kd-tree inputs, :567-568)
    DevBuf<int> fvalid, fkind;
    DevBuf<double> fcp, fpa, fpb, partials;
    DevBuf<LMSync> lm_sync;
    DevBuf<int> cell_cnt, cell_start, cell_rank;  // NN(1) bucket index of corner_last | surf_last (see CellIndex)
    DevBuf<float4> cell_pts;
    // tables are laid out [cloud][level]; point arrays [level][corner | surf]
    int* cell_cnt_of(int k, int l) const { return cell_cnt.p + (k * OD_LEVELS + l) * OD_H; }
    int* cell_start_of(int k, int l) const { return cell_start.p + (k * OD_LEVELS + l) * OD_START_STRIDE; }
    int* cell_rank_of(int k, int l) const { return cell_rank.p + (size_t)l * (feat_cap + cap) + (k ? feat_cap : 0); }
    float4* cell_pts_of(int k, int l) const { return cell_pts.p + (size_t)l * (feat_cap + cap) + (k ? feat_cap : 0); }
    CellIndex cell_index(int k) const {
        CellIndex ci;
        for (int l = 0; l < OD_LEVELS; ++l) ci.start[l] = cell_start_of(k, l), ci.pts[l] = cell_pts_of(k, l);
        return ci;
    }
    DevBuf<int> ring_tab;  // 2 sets of [corner first | corner last | surf first | surf last] x RING_TAB (double-buffered)
    int tab_cur = 0;       // set read by this scan's association
    DevBuf<LMState> d_st;
    DevBuf<OdomCounters> d_C;
    PinBuf<LMState> h_st;       // [MAX_STEPS]
    PinBuf<OdomCounters> h_up;  // upload staging of scal_odom_step
    HostStage hs;               // pinned staging of its four host clouds (first call)
    // Ceres-adapter mode (scal_odom_adapter_begin ... _finish)
    bool adapter_active = false, adapter_solve = false;
    DevBuf<int> bl_live, bl_rowoff, bl_counts;
    DevBuf<double> d_x7, d_res, d_jac, d_blocks;
    PinBuf<int> h_counts;
    PinBuf<double> h_x7;
    int outer_next = 0;
    BlockList block_list() { return BlockList{bl_live.p, bl_rowoff.p, bl_counts.p}; }
    FactorSoA factors() { return FactorSoA{fvalid.p, fkind.p, fcp.p, fpa.p, fpb.p, slot_cap}; }
};

extern "C" int scal_odom_create(const scal_odom_config* cfg, scal_odom_t** out) {
    if (!cfg || !out || cfg->max_points <= 0) {
        set_error("scal_odom_create: bad argument");
        return SCAL_E_ARG;
    }
    *out = nullptr;
    SCAL_TRY(select_device(cfg->device));
    auto* c = new scal_odom();
    c->cfg = *cfg;
    c->cap = cfg->max_points;
    c->feat_cap = 120 * 64;  // lessSharp <= 20 per segment, 6 segments per ring (scanRegistration.cpp:314-328)
    c->slot_cap = 36 * 64;   // sharp <= 2, flat <= 4 per segment
    int rc = SCAL_OK;
    auto A = [&](int r) { if (rc == SCAL_OK) rc = r; };
    A(c->aos.alloc((size_t)c->cap * 4));
    A(c->sharp.alloc(c->slot_cap)); A(c->flat.alloc(c->slot_cap));
    A(c->less_sharp.alloc(c->feat_cap)); A(c->corner_last.alloc(c->feat_cap));
    A(c->less_flat.alloc(c->cap)); A(c->surf_last.alloc(c->cap));
    A(c->fvalid.alloc(c->slot_cap)); A(c->fkind.alloc(c->slot_cap));
    A(c->fcp.alloc(3 * (size_t)c->slot_cap)); A(c->fpa.alloc(3 * (size_t)c->slot_cap)); A(c->fpb.alloc(3 * (size_t)c->slot_cap));
    A(c->partials.alloc(LM_PARTIAL_WORDS));
    A(c->lm_sync.alloc(1));
    A(c->cell_cnt.alloc(2 * OD_LEVELS * OD_H)); A(c->cell_start.alloc(2 * OD_LEVELS * OD_START_STRIDE));
    A(c->cell_rank.alloc(OD_LEVELS * ((size_t)c->feat_cap + c->cap))); A(c->cell_pts.alloc(OD_LEVELS * ((size_t)c->feat_cap + c->cap)));
    A(c->ring_tab.alloc(8 * RING_TAB));
    A(c->bl_live.alloc(c->slot_cap)); A(c->bl_rowoff.alloc(c->slot_cap + 1)); A(c->bl_counts.alloc(2)); A(c->h_counts.alloc(2)); A(c->h_x7.alloc(8));
    A(c->d_x7.alloc(8)); A(c->d_res.alloc(3 * (size_t)c->slot_cap)); A(c->d_jac.alloc(21 * (size_t)c->slot_cap)); A(c->d_blocks.alloc(10 * (size_t)c->slot_cap));
    A(c->d_st.alloc(1)); A(c->d_C.alloc(1)); A(c->h_st.alloc(scal_odom::MAX_STEPS)); A(c->h_up.alloc(1));
    c->lane = stage_lane(STAGE_ODOM);
    if (rc == SCAL_OK) rc = lm_check_residency<LMNoHook, LMNoHook>(c->cfg.device);
    if (rc == SCAL_OK && acquire_stream(c->cfg.device, &c->stream, c->lane) != SCAL_OK) rc = SCAL_E_HIP;
    for (int k = 0; k < scal_odom::MAX_STEPS && rc == SCAL_OK; ++k)
        if (hipEventCreateWithFlags(&c->ev[k], hipEventDisableTiming) != hipSuccess) rc = SCAL_E_HIP;
    if (rc == SCAL_OK) {
        // everything is initialised on the context's own stream (the legacy null stream is not ordered against it)
        if (op_memset_async(c->lm_sync.p, 0, sizeof(LMSync), c->stream) != hipSuccess) rc = SCAL_E_HIP;
        if (rc == SCAL_OK) rc = c->partials.zero(c->stream);  // sequence number 0 = never published
        if (rc == SCAL_OK && op_memset_async(c->d_st.p, 0, sizeof(LMState), c->stream) != hipSuccess) rc = SCAL_E_HIP;
        if (rc == SCAL_OK && op_memset_async(c->d_C.p, 0, sizeof(OdomCounters), c->stream) != hipSuccess) rc = SCAL_E_HIP;
        if (rc == SCAL_OK) rc = c->cell_cnt.zero(c->stream);
        if (rc == SCAL_OK) rc = c->cell_start.zero(c->stream);
        SCAL_LAUNCH("k_odom_init_pose", k_odom_init_pose, dim3(1), dim3(256), 0, c->stream, c->d_st.p, c->ring_tab.p);
        if (rc == SCAL_OK && op_stream_synchronize(c->stream) != hipSuccess) rc = SCAL_E_HIP;
    }
    if (rc != SCAL_OK) {
        if (rc == SCAL_E_HIP) set_error("scal_odom_create: HIP resource creation failed");
        delete c;
        return rc;
    }
    *out = c;
    return SCAL_OK;
}

extern "C" void scal_odom_destroy(scal_odom_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) {
        (void)op_stream_synchronize(c->stream);
        release_stream(c->cfg.device, c->lane);
    }
    for (int k = 0; k < scal_odom::MAX_STEPS; ++k)
        if (c->ev[k]) (void)hipEventDestroy(c->ev[k]);
    delete c;
}

namespace {

void o_qmul(const double* a, const double* b, double* o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
void o_rot(const double* q, const double* v, double* o) {
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux, uy += uy, uz += uz;
    const double cx = q[1] * uz - q[2] * uy, cy = q[2] * ux - q[0] * uz, cz = q[0] * uy - q[1] * ux;
    o[0] = (v[0] + q[3] * ux) + cx, o[1] = (v[1] + q[3] * uy) + cy, o[2] = (v[2] + q[3] * uz) + cz;
}

int launch_handover(scal_odom* c, int slot, hipEvent_t ev);
// inputs already in sharp / flat / less_sharp / less_flat with counts in d_C
int odom_enqueue(scal_odom* c) {
    hipStream_t s = c->stream;
    OdomCounters* C = c->d_C.p;
    LMState* st = c->d_st.p;
    FactorSoA F = c->factors();
    const bool solve = c->systemInited;  // first frame: no optimisation (:267-271)
    c->systemInited = true;
    const int slot = c->next_slot;
    c->next_slot = (c->next_slot + 1) % scal_odom::MAX_STEPS;
    if (solve) {
        for (int outer = 0; outer < 2; ++outer) {  // :278
            {
                SCAL_LAUNCH("k_odom_assoc", k_odom_assoc, dim3(std::max(1, c->slot_cap)), dim3(256), 0, s, c->sharp.cv(), c->flat.cv(),
                                   c->corner_last.cv(), c->surf_last.cv(), st, C, outer, F, c->cell_index(0), c->cell_index(1), c->ring_tab.p + c->tab_cur * 4 * RING_TAB);
            }
            {
                                launch_lm_solve(s, F, &C->n_slots, st, &C->enable, c->partials.p, c->lm_sync.p, outer, nullptr, LMNoHook(), LMNoHook(), "k_lm_solve_odom");
            }
        }
    }
    SCAL_TRY(launch_handover(c, slot, c->ev[slot]));
    c->pending.push_back({slot, solve});
    return SCAL_OK;
}

// hand-over (:554-563): the current lessSharp / lessFlat clouds become the next scan's targets; the state goes to the host slot
int launch_handover(scal_odom* c, int slot, hipEvent_t ev) {
    hipStream_t s = c->stream;
    OdomCounters* C = c->d_C.p;
    LMState* st = c->d_st.p;
    {
        HandoverArgs h;
        h.in[0] = c->less_sharp.cv(), h.in[1] = c->less_flat.cv();
        h.d_n[0] = &C->n_less_sharp, h.d_n[1] = &C->n_less_flat;
        h.out[0] = c->corner_last.v(), h.out[1] = c->surf_last.v();
        h.d_n_out[0] = &C->n_corner_last, h.d_n_out[1] = &C->n_surf_last;
        h.cap[0] = c->feat_cap, h.cap[1] = c->cap;
        h.nb0 = std::max(1, div_up(c->feat_cap, 256));
        h.tab_write = c->ring_tab.p + (c->tab_cur ^ 1) * 4 * RING_TAB;
        h.tab_reset = c->ring_tab.p + c->tab_cur * 4 * RING_TAB;
        h.st = st, h.host_st = c->h_st.p + slot, h.sync = c->lm_sync.p;
        for (int k = 0; k < 2; ++k)
            for (int l = 0; l < OD_LEVELS; ++l) h.cell_cnt[k][l] = c->cell_cnt_of(k, l), h.cell_rank[k][l] = c->cell_rank_of(k, l);
        const dim3 grid(h.nb0 + std::max(1, div_up(c->cap, 256)));
        SCAL_LAUNCH("k_odom_handover", k_odom_handover, grid, dim3(256), 0, s, h);
        c->tab_cur ^= 1;
        // the pose is out with the hand-over; the bucket index of the new targets is only needed by the next step's association
        if (ev) SCAL_HIP(op_event_record(ev, s));
        SCAL_LAUNCH("k_odom_cellscan", k_odom_cellscan, dim3(2 * OD_LEVELS), dim3(1024), 0, s, c->cell_cnt.p, c->cell_start.p);
        CellFillArgs f;
        f.in[0] = c->corner_last.cv(), f.in[1] = c->surf_last.cv();
        f.d_n[0] = &C->n_corner_last, f.d_n[1] = &C->n_surf_last;
        for (int k = 0; k < 2; ++k)
            for (int l = 0; l < OD_LEVELS; ++l) f.rank[k][l] = c->cell_rank_of(k, l), f.start[k][l] = c->cell_start_of(k, l), f.pts[k][l] = c->cell_pts_of(k, l);
        f.nb0 = h.nb0;
        SCAL_LAUNCH("k_odom_cellfill", k_odom_cellfill, grid, dim3(256), 0, s, f);
    }
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

// waits for the enqueued step and integrates the pose on the host (:504-505)
int odom_collect(scal_odom* c, double* q_lc, double* t_lc, double* q_w, double* t_w, scal_odom_stats* stats) {
    if (c->pending.empty()) {
        set_error("scal_odom_collect: no step enqueued");
        return SCAL_E_STATE;
    }
    const int slot = c->pending.front().slot;
    const bool solve = c->pending.front().solve;
    c->pending.pop_front();
    SCAL_HIP(op_event_synchronize(c->ev[slot]));
    const LMState& L = c->h_st.p[slot];
    if (L.termination == 5) {
        // A workgroup of this step's solve - or of a step queued in front of it: the flag is sticky on the device and every solve
        // behind it returned at once - ran out of polls.  Never seen outside the test hook, but the pose cannot be trusted: the step
        // is reported as failed (the pose keeps its last good value), and the exchange is cleared once everything queued has drained.
        (void)op_stream_synchronize(c->stream);
        (void)op_memset_async(c->lm_sync.p, 0, sizeof(LMSync), c->stream);
        (void)c->partials.zero(c->stream);
        (void)op_stream_synchronize(c->stream);
        set_error("LM solve abandoned: grid barrier timed out");
        return SCAL_E_HIP;
    }
    const double* x = L.x;
    if (solve) {  // :504-505
        double r[3];
        o_rot(c->q_w_curr, x + 4, r);
        for (int i = 0; i < 3; ++i) c->t_w_curr[i] = c->t_w_curr[i] + r[i];
        double qn[4];
        o_qmul(c->q_w_curr, x, qn);
        for (int i = 0; i < 4; ++i) c->q_w_curr[i] = qn[i];
    }
    for (int i = 0; i < 4; ++i) q_lc[i] = x[i], q_w[i] = c->q_w_curr[i];
    for (int i = 0; i < 3; ++i) t_lc[i] = x[4 + i], t_w[i] = c->t_w_curr[i];
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        if (solve) {
            for (int o = 0; o < 2; ++o) {
                stats->n_edge[o] = L.log_n_edge[o], stats->n_plane[o] = L.log_n_plane[o];
                stats->lm_iters[o] = L.log_iters[o], stats->lm_success[o] = L.log_success[o];
                stats->cost_init[o] = L.log_cost_init[o], stats->cost_final[o] = L.log_cost_final[o];
            }
        }
    }
    return SCAL_OK;
}

}  // namespace

static int odom_upload(scal_odom* c, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp, const float* flat, int n_flat,
                       const float* less_flat, int n_less_flat) {
    hipStream_t s = c->stream;
    SCAL_HIP(op_stream_synchronize(s));  // the upload staging may still feed the previous call's copy
    c->hs.finish();
    (void)c->hs.reserve(((size_t)2 * c->slot_cap + c->feat_cap + c->cap) * 16 + 4096);
    OdomCounters& H = *c->h_up.p;
    // keep the device-resident n_corner_last / n_surf_last, refresh the per-scan part from pinned memory
    OdomCounters fresh;
    std::memset(&fresh, 0, sizeof fresh);
    fresh.n_sharp = n_sharp, fresh.n_flat = n_flat, fresh.n_less_sharp = n_less_sharp, fresh.n_less_flat = n_less_flat, fresh.enable = 1;
    H = fresh;
    const size_t off = offsetof(OdomCounters, n_sharp);
    SCAL_HIP(op_memcpy_async(reinterpret_cast<char*>(c->d_C.p) + off, reinterpret_cast<char*>(&H) + off, sizeof(OdomCounters) - off,
                            hipMemcpyHostToDevice, s));
    auto up = [&](const float* src, int n, OSoA& dst) -> int {
        if (n > 0) {
            SCAL_HIP(c->hs.h2d(c->aos.p, src, sizeof(float) * 4 * n, s));
            launch_deinterleave(s, c->aos.p, n, dst.v());
        }
        return SCAL_OK;
    };
    SCAL_TRY(up(sharp, n_sharp, c->sharp));
    SCAL_TRY(up(flat, n_flat, c->flat));
    SCAL_TRY(up(less_sharp, n_less_sharp, c->less_sharp));
    SCAL_TRY(up(less_flat, n_less_flat, c->less_flat));
    return SCAL_OK;
}

extern "C" int scal_odom_step(scal_odom_t* c, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp, const float* flat,
                              int n_flat, const float* less_flat, int n_less_flat, double* q_lc, double* t_lc, double* q_w, double* t_w,
                              scal_odom_stats* stats) {
    if (!c || !q_lc || !t_lc || !q_w || !t_w || n_sharp < 0 || n_less_sharp < 0 || n_flat < 0 || n_less_flat < 0 || (n_sharp && !sharp) ||
        (n_less_sharp && !less_sharp) || (n_flat && !flat) || (n_less_flat && !less_flat)) {
        set_error("scal_odom_step: bad argument");
        return SCAL_E_ARG;
    }
    if (n_sharp + n_flat > c->slot_cap || n_sharp > c->slot_cap || n_flat > c->slot_cap || n_less_sharp > c->feat_cap || n_less_flat > c->cap) {
        set_error("scal_odom_step: cloud larger than the context capacity (sharp+flat <= %d, lessSharp <= %d, lessFlat <= %d)", c->slot_cap,
                  c->feat_cap, c->cap);
        return SCAL_E_TOO_MANY;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    if (!c->pending.empty()) {
        set_error("scal_odom_step: a queued step has not been collected");
        return SCAL_E_STATE;
    }
    SCAL_TRY(odom_upload(c, sharp, n_sharp, less_sharp, n_less_sharp, flat, n_flat, less_flat, n_less_flat));
    SCAL_TRY(odom_enqueue(c));
    return odom_collect(c, q_lc, t_lc, q_w, t_w, stats);
}

extern "C" int scal_odom_enqueue_features(scal_odom_t* c, scal_features_t* feat) {
    if (!c || !feat) {
        set_error("scal_odom_enqueue_features: null argument");
        return SCAL_E_ARG;
    }
    if (static_cast<int>(c->pending.size()) >= scal_odom::MAX_STEPS) {
        set_error("scal_odom_enqueue_features: %d steps are queued and not collected", scal_odom::MAX_STEPS);
        return SCAL_E_STATE;
    }
    FeatDeviceView v = features_view(feat);
    if (v.device != c->cfg.device) {
        set_error("features context lives on device %d, odometry context on %d", v.device, c->cfg.device);
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    OdomGatherArgs a;
    a.sharp = v.sharp_xyzi, a.flat = v.flat_xyzi, a.less = v.less_xyzi;
    a.less_flat = CSoA4{v.lfx, v.lfy, v.lfz, v.lfi};
    a.n_sharp = &v.P->n_sharp, a.n_flat = &v.P->n_flat, a.n_less = &v.P->n_less_sharp, a.n_less_flat = &v.P->n_less_flat;
    a.o_sharp = c->sharp.v(), a.o_flat = c->flat.v(), a.o_less = c->less_sharp.v(), a.o_less_flat = c->less_flat.v();
    a.slot_cap = c->slot_cap, a.feat_cap = c->feat_cap, a.cap = std::min(c->cap, v.cap);
    a.nbs = std::max(1, div_up(c->slot_cap, 256)), a.nbf = std::max(1, div_up(c->feat_cap, 256));
    SCAL_TRY(features_wait_done(feat, s));
    SCAL_LAUNCH("k_odom_gather", k_odom_gather, dim3(2 * a.nbs + a.nbf + std::max(1, div_up(a.cap, 256))), dim3(256), 0, s, a, c->d_C.p);
    SCAL_TRY(features_note_reader(feat, s));  // everything stage B needs has been copied out of the features context
    return odom_enqueue(c);
}

extern "C" int scal_odom_collect(scal_odom_t* c, double* q_lc, double* t_lc, double* q_w, double* t_w, scal_odom_stats* stats) {
    if (!c || !q_lc || !t_lc || !q_w || !t_w) {
        set_error("scal_odom_collect: null argument");
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    return odom_collect(c, q_lc, t_lc, q_w, t_w, stats);
}

extern "C" int scal_odom_step_features(scal_odom_t* c, scal_features_t* feat, double* q_lc, double* t_lc, double* q_w, double* t_w,
                                       scal_odom_stats* stats) {
    if (!c || !feat || !q_lc || !t_lc || !q_w || !t_w) {
        set_error("scal_odom_step_features: null argument");
        return SCAL_E_ARG;
    }
    SCAL_TRY(scal_odom_enqueue_features(c, feat));
    return odom_collect(c, q_lc, t_lc, q_w, t_w, stats);
}

// ------------------------------------------------------------------------------------------------ Ceres-adapter mode
// laserOdometry.cpp:278-501 with the host's own ceres::Problem / ceres::Solve: the device associates at the solver's current
// q_last_curr / t_last_curr and evaluates the residual blocks in batches.
extern "C" int scal_odom_adapter_begin(scal_odom_t* c, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp, const float* flat,
                                       int n_flat, const float* less_flat, int n_less_flat, double* q_last_curr, double* t_last_curr, int* need_solve) {
    if (!c || !q_last_curr || !t_last_curr || !need_solve || n_sharp < 0 || n_less_sharp < 0 || n_flat < 0 || n_less_flat < 0 || (n_sharp && !sharp) ||
        (n_less_sharp && !less_sharp) || (n_flat && !flat) || (n_less_flat && !less_flat)) {
        set_error("scal_odom_adapter_begin: bad argument");
        return SCAL_E_ARG;
    }
    if (n_sharp + n_flat > c->slot_cap || n_less_sharp > c->feat_cap || n_less_flat > c->cap) {
        set_error("scal_odom_adapter_begin: cloud larger than the context capacity");
        return SCAL_E_TOO_MANY;
    }
    if (!c->pending.empty() || c->adapter_active) {
        set_error("scal_odom_adapter_begin: a step is still open");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    SCAL_HIP(op_stream_synchronize(s));
    SCAL_TRY(odom_upload(c, sharp, n_sharp, less_sharp, n_less_sharp, flat, n_flat, less_flat, n_less_flat));
    SCAL_HIP(op_memcpy_async(c->h_st.p, c->d_st.p, sizeof(LMState), hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    for (int i = 0; i < 4; ++i) q_last_curr[i] = c->h_st.p->x[i];   // para_q / para_t persist across scans (:97-101)
    for (int i = 0; i < 3; ++i) t_last_curr[i] = c->h_st.p->x[4 + i];
    c->adapter_solve = c->systemInited;  // first frame: no optimisation (:267-271)
    c->systemInited = true;
    *need_solve = c->adapter_solve ? 1 : 0;
    c->adapter_active = true;
    c->outer_next = 0;
    return SCAL_OK;
}

static int odom_set_pose(scal_odom* c, const double* q, const double* t) {
    SCAL_HIP(op_stream_synchronize(c->stream));
    for (int i = 0; i < 4; ++i) c->h_x7.p[i] = q[i];
    for (int i = 0; i < 3; ++i) c->h_x7.p[4 + i] = t[i];
    SCAL_HIP(op_memcpy_async(c->d_st.p->x, c->h_x7.p, sizeof(double) * 7, hipMemcpyHostToDevice, c->stream));
    return SCAL_OK;
}

extern "C" int scal_odom_associate(scal_odom_t* c, const double* q_last_curr, const double* t_last_curr, int* n_blocks, int* n_residuals) {
    if (!c || !q_last_curr || !t_last_curr || !n_blocks || !n_residuals) {
        set_error("scal_odom_associate: null argument");
        return SCAL_E_ARG;
    }
    if (!c->adapter_active || !c->adapter_solve) {
        set_error("scal_odom_associate: no open step that needs a solve");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    OdomCounters* C = c->d_C.p;
    LMState* st = c->d_st.p;
    FactorSoA F = c->factors();
    SCAL_TRY(odom_set_pose(c, q_last_curr, t_last_curr));
    SCAL_LAUNCH("k_odom_assoc", k_odom_assoc, dim3(std::max(1, c->slot_cap)), dim3(256), 0, s, c->sharp.cv(), c->flat.cv(), c->corner_last.cv(),
                     c->surf_last.cv(), st, C, std::min(c->outer_next, 1), F, c->cell_index(0), c->cell_index(1), c->ring_tab.p + c->tab_cur * 4 * RING_TAB);
    c->outer_next++;
    SCAL_LAUNCH("k_blocks_compact", k_blocks_compact, dim3(1), dim3(1024), 0, s, F, &C->n_slots, c->block_list());
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(op_memcpy_async(c->h_counts.p, c->bl_counts.p, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    *n_blocks = c->h_counts.p[0], *n_residuals = c->h_counts.p[1];
    return SCAL_OK;
}

extern "C" int scal_odom_get_blocks(scal_odom_t* c, scal_block* out, int cap) {
    if (!c || (!out && cap > 0) || cap < 0) {
        set_error("scal_odom_get_blocks: bad argument");
        return SCAL_E_ARG;
    }
    if (!c->adapter_active) {
        set_error("scal_odom_get_blocks: no scal_odom_adapter_begin");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    const int n = std::min(cap, c->h_counts.p[0]);
    if (n <= 0) return 0;
    hipStream_t s = c->stream;
    SCAL_LAUNCH("k_blocks_export", k_blocks_export, dim3(div_up(c->h_counts.p[0], 256)), dim3(256), 0, s, c->factors(), c->block_list(), c->d_blocks.p);
    SCAL_HIP(op_memcpy_async(out, c->d_blocks.p, sizeof(scal_block) * n, hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    return n;
}

extern "C" int scal_odom_eval_blocks(scal_odom_t* c, const double* x7, int want_jac, double* residuals, double* jacobians) {
    if (!c || !x7 || !residuals || (want_jac && !jacobians)) {
        set_error("scal_odom_eval_blocks: bad argument");
        return SCAL_E_ARG;
    }
    if (!c->adapter_active) {
        set_error("scal_odom_eval_blocks: no scal_odom_adapter_begin");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    const int nb = c->h_counts.p[0], nr = c->h_counts.p[1];
    if (nb == 0) return SCAL_OK;
    hipStream_t s = c->stream;
    SCAL_HIP(op_stream_synchronize(s));
    for (int i = 0; i < 7; ++i) c->h_x7.p[i] = x7[i];
    SCAL_HIP(op_memcpy_async(c->d_x7.p, c->h_x7.p, sizeof(double) * 7, hipMemcpyHostToDevice, s));
    SCAL_LAUNCH("k_blocks_eval", k_blocks_eval, dim3(div_up(nb, 256)), dim3(256), 0, s, c->factors(), c->block_list(), c->d_x7.p, want_jac ? 1 : 0,
                     c->d_res.p, c->d_jac.p);
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(op_memcpy_async(residuals, c->d_res.p, sizeof(double) * nr, hipMemcpyDeviceToHost, s));
    if (want_jac) SCAL_HIP(op_memcpy_async(jacobians, c->d_jac.p, sizeof(double) * 7 * nr, hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    return SCAL_OK;
}

extern "C" int scal_odom_adapter_finish(scal_odom_t* c, const double* q_last_curr, const double* t_last_curr, double* q_w_curr, double* t_w_curr) {
    if (!c || !q_last_curr || !t_last_curr || !q_w_curr || !t_w_curr) {
        set_error("scal_odom_adapter_finish: null argument");
        return SCAL_E_ARG;
    }
    if (!c->adapter_active) {
        set_error("scal_odom_adapter_finish: no scal_odom_adapter_begin");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    c->adapter_active = false;
    SCAL_TRY(odom_set_pose(c, q_last_curr, t_last_curr));  // para_q / para_t of the next scan's initial guess
    const int slot = c->next_slot;
    c->next_slot = (c->next_slot + 1) % scal_odom::MAX_STEPS;
    SCAL_TRY(launch_handover(c, slot, c->ev[slot]));
    c->pending.push_back({slot, c->adapter_solve});
    double q_lc[4], t_lc[3];
    return odom_collect(c, q_lc, t_lc, q_w_curr, t_w_curr, nullptr);  // pose integration (:504-505)
}

extern "C" void* scal_odom_stream(scal_odom_t* c) { return c ? static_cast<void*>(c->stream) : nullptr; }
