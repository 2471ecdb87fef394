// Stage A on gfx950: per-scan ring assignment, ring-major reorder, curvature and edge/surf feature picks.
// Replaces laserCloudHandler, /root/reference/src/scanRegistration.cpp:134-421 (see include/scaloam_hip.h).
//
// Kernel chain (one HIP stream, no host round trip until fetch):
//   k_pre       1 block      first/last surviving point -> startOri/endOri (:142-155), counters reset
//   k_classify  N/1024       NaN/range filter (:87-114,:138), ring id (:168-218), -atan2 (:221), first point past
//                            half sweep (the sequential `halfPassed` of :222-237 is a min-index reduction),
//                            per-block ring histogram
//   k_ringscan  1 block      histogram -> stable scatter offsets, ring offsets, scanStartInd/scanEndInd (:259-265)
//   k_scatter   N/1024       stable ring-major reorder (wave-match ranking), relTime/intensity (:239-252) -> SoA
//   k_curv      N'/256       11-tap curvature, exact left-to-right f32 order (:269-279); neighbour-gap bits (:334-337)
//   k_ring      1 block/ring six LDS bitonic sorts by (curvature, index) (:290-302), greedy picks with +-5
//                            suppression by one wave scanning 64 candidates per step (:304-403), lessFlat
//                            collection (:405-411) and the per-ring 0.2 m voxel grid (:414-420) as an LDS sort +
//                            ordered segmented mean
//   k_compact   N'/256       per-segment pick slots -> the reference's emission order (every block derives the offsets itself);
//                            ring-staged lessFlat centroids -> contiguous cloud; xyzi records of the picks
// Data layout: SoA x[] y[] z[] intensity[] in HBM (coalesced 4 B/lane loads; the whole scan is < 2 MB and lives
// in L2).  All f32 arithmetic that feeds a comparison is compiled without FMA contraction.
#include "common.hpp"
#include <cstring>
#include <mutex>
#include "device_utils.hpp"
#include "features_dev.hpp"
#include "voxel_dev.hpp"
#include <cmath>
#include <climits>

namespace scal {

constexpr int TILE = 1024;      // points per block in classify/scatter: 4 waves x 4 items x 64 lanes
constexpr int RING_THREADS = 1024;
constexpr int RING_MAX = 16384; // longest ring the LDS sort accepts

__device__ __forceinline__ bool point_kept(float x, float y, float z, float thres2, int check_finite) {
    if (check_finite && !(isfinite(x) && isfinite(y) && isfinite(z))) return false;
    return !(x * x + y * y + z * z < thres2);  // scanRegistration.cpp:101
}

// -atan2(y, x) with std::atan2(float,float) semantics (:56, :143, :221), correctly rounded via f64
__device__ __forceinline__ float neg_atan2f_cr(float y, float x) { return -static_cast<float>(atan2(static_cast<double>(y), static_cast<double>(x))); }

__device__ __forceinline__ int ring_of(float x, float y, float z, int lidar_type, int n_scans, int float_math) {
    float angle;
    if (!float_math) {
        angle = static_cast<float>(atan(static_cast<double>(z) / sqrt(static_cast<double>(x * x + y * y))) * 180 / M_PI);  // :168
    } else {
        const float a = static_cast<float>(atan(static_cast<double>(z / sqrtf(x * x + y * y))));
        angle = static_cast<float>(static_cast<double>(a * 180.0f) / M_PI);
    }
    if (angle != angle) return -1;  // int(NaN) is INT_MIN on the reference's x86-64 => rejected
    int scanID;
    if (lidar_type == SCAL_VLP16) {
        scanID = static_cast<int>(static_cast<double>((angle + 15.0f) / 2.0f) + 0.5);  // :173
        if (scanID > (n_scans - 1) || scanID < 0) return -1;
    } else if (lidar_type == SCAL_HDL32) {
        scanID = static_cast<int>((static_cast<double>(angle) + 92.0 / 3.0) * 3.0 / 4.0);  // :182
        if (scanID > (n_scans - 1) || scanID < 0) return -1;
    } else if (lidar_type == SCAL_HDL64) {
        if (static_cast<double>(angle) >= -8.83)  // :192-195
            scanID = static_cast<int>(static_cast<double>(2.0f - angle) * 3.0 + 0.5);
        else
            scanID = n_scans / 2 + static_cast<int>((-8.83 - static_cast<double>(angle)) * 2.0 + 0.5);
        if (angle > 2.0f || static_cast<double>(angle) < -24.33 || scanID > 50 || scanID < 0) return -1;  // :198
    } else {
        scanID = static_cast<int>((static_cast<double>(angle) + 22.5) / 2 + 0.5);  // :207
        if (scanID > (n_scans - 1) || scanID < 0) return -1;
    }
    return scanID;
}

struct KCfg {
    int lidar_type, n_scans, float_math, check_finite;
    float thres2;
};

__device__ __forceinline__ void k_pre_body(const float* __restrict__ in, int n, int stride, const KCfg& c, FeatParams* P) {
    __shared__ int s_first, s_last;
    if (threadIdx.x == 0) s_first = INT_MAX, s_last = -1;
    __syncthreads();
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + threadIdx.x;
        bool k = false;
        if (i < n) k = point_kept(in[(size_t)i * stride], in[(size_t)i * stride + 1], in[(size_t)i * stride + 2], c.thres2, c.check_finite);
        if (k) atomicMin(&s_first, i);
        if (__syncthreads_or(k)) break;
    }
    for (int base = n - 1; base >= 0; base -= blockDim.x) {
        const int i = base - static_cast<int>(threadIdx.x);
        bool k = false;
        if (i >= 0) k = point_kept(in[(size_t)i * stride], in[(size_t)i * stride + 1], in[(size_t)i * stride + 2], c.thres2, c.check_finite);
        if (k) atomicMax(&s_last, i);
        if (__syncthreads_or(k)) break;
    }
    __syncthreads();
    if (threadIdx.x < 64) P->lf_ring_cnt[threadIdx.x] = 0;  // k_ring only visits rings < n_scans
    if (threadIdx.x == 0) {
        P->n_sharp = P->n_less_sharp = P->n_flat = P->n_less_flat = 0;
        P->first_idx = s_first;
        P->last_idx = s_last;
        P->flip_idx = INT_MAX;
        P->empty = (s_last < 0);
        P->error = 0;
        P->n_tied = 0;
        P->n_kept = 0;
        if (s_last >= 0) {
            const size_t f = (size_t)s_first * stride, l = (size_t)s_last * stride;
            float startOri = neg_atan2f_cr(in[f + 1], in[f]);                                      // :143
            float endOri = static_cast<float>(static_cast<double>(neg_atan2f_cr(in[l + 1], in[l])) + 2 * M_PI);  // :144-146
            if (static_cast<double>(endOri - startOri) > 3 * M_PI)
                endOri = static_cast<float>(static_cast<double>(endOri) - 2 * M_PI);
            else if (static_cast<double>(endOri - startOri) < M_PI)
                endOri = static_cast<float>(static_cast<double>(endOri) + 2 * M_PI);
            P->start_ori = startOri;
            P->end_ori = endOri;
        }
    }
}
SCAL_KERNEL(1024, k_pre)

// element handled by (wave w, item j, lane l) of a block: base + w*256 + j*64 + l  => (j,l) order is arrival order
__device__ __forceinline__ void k_classify_body(const float* __restrict__ in, int n, int stride, const KCfg& c, FeatParams* P,
                                                  signed char* __restrict__ ring, float* __restrict__ ori, int* __restrict__ block_hist,
                                                  int nb) {
    __shared__ int hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    if (!P->empty) {
        const float startOri = P->start_ori;
        const int w = wave_id(), l = lane_id();
        int flip = INT_MAX;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = blockIdx.x * TILE + w * 256 + j * 64 + l;
            if (e < n) {
                const float x = in[(size_t)e * stride], y = in[(size_t)e * stride + 1], z = in[(size_t)e * stride + 2];
                int r = -1;
                float o = 0.f;
                if (point_kept(x, y, z, c.thres2, c.check_finite)) r = ring_of(x, y, z, c.lidar_type, c.n_scans, c.float_math);
                if (r >= 0) {
                    o = neg_atan2f_cr(y, x);  // :221
                    // first-half branch of :222-237; the first surviving point where it trips sets halfPassed
                    float oa = o;
                    if (static_cast<double>(oa) < static_cast<double>(startOri) - M_PI / 2)
                        oa = static_cast<float>(static_cast<double>(oa) + 2 * M_PI);
                    else if (static_cast<double>(oa) > static_cast<double>(startOri) + M_PI * 3 / 2)
                        oa = static_cast<float>(static_cast<double>(oa) - 2 * M_PI);
                    if (static_cast<double>(oa - startOri) > M_PI) flip = min(flip, e);
                    atomicAdd(&hist[r], 1);
                }
                ring[e] = static_cast<signed char>(r);
                ori[e] = o;
            }
        }
        flip = wave_min_i(flip);
        if (l == 0 && flip != INT_MAX) atomicMin(&P->flip_idx, flip);
    }
    __syncthreads();
    if (threadIdx.x < 64) block_hist[threadIdx.x * nb + blockIdx.x] = hist[threadIdx.x];
}
SCAL_KERNEL(256, k_classify)

__device__ __forceinline__ void k_ringscan_body(int* __restrict__ block_hist, int nb, int n_scans, FeatParams* P) {
    __shared__ int cnt[64];
    const int w = wave_id(), l = lane_id();
    for (int r = w; r < 64; r += 16) {
        int running = 0;
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const int b = c0 + l;
            const int v = b < nb ? block_hist[r * nb + b] : 0;
            const int incl = wave_inclusive_scan(v);
            if (b < nb) block_hist[r * nb + b] = running + incl - v;
            running += __shfl(incl, 63, 64);
        }
        if (l == 0) cnt[r] = running;
    }
    __syncthreads();
    if (w == 0) {
        const int v = cnt[l];
        const int incl = wave_inclusive_scan(v);
        const int off = incl - v;
        P->ring_count[l] = v;
        P->ring_off[l] = off;
        if (l < n_scans) {
            P->scan_start[l] = off + 5;       // :262
            P->scan_end[l] = off + v - 6;     // :264
        }
        if (l == 63) {
            P->ring_off[64] = incl;
            P->n_kept = incl;
        }
    }
}
SCAL_KERNEL(1024, k_ringscan)

__device__ __forceinline__ void k_scatter_body(const float* __restrict__ in, int n, int stride, const FeatParams* __restrict__ P,
                                                 const signed char* __restrict__ ring, const float* __restrict__ ori,
                                                 const int* __restrict__ block_off, int nb, float* __restrict__ ox, float* __restrict__ oy,
                                                 float* __restrict__ oz, float* __restrict__ oi, int* __restrict__ src) {
    __shared__ int cnt[4][64];
    if (P->empty) return;
    const int w = wave_id(), l = lane_id();
    cnt[w][l] = 0;
    __syncthreads();
    int rnk[4], rg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = blockIdx.x * TILE + w * 256 + j * 64 + l;
        const int r = e < n ? ring[e] : -1;
        const bool valid = r >= 0;
        const uint64_t m = wave_match<6>(static_cast<uint32_t>(r & 63), valid);
        int prev = 0;
        if (valid) prev = cnt[w][r];
        rnk[j] = prev + __popcll(m & lanemask_lt());
        rg[j] = r;
        __builtin_amdgcn_wave_barrier();
        if (valid && (m & lanemask_lt()) == 0) cnt[w][r] = prev + __popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const float startOri = P->start_ori, endOri = P->end_ori;
    const int flip = P->flip_idx;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = blockIdx.x * TILE + w * 256 + j * 64 + l;
        const int r = rg[j];
        if (r < 0) continue;
        int wprefix = 0;
        for (int ww = 0; ww < w; ++ww) wprefix += cnt[ww][r];
        const int pos = P->ring_off[r] + block_off[r * nb + blockIdx.x] + wprefix + rnk[j];
        float o = ori[e];
        if (e <= flip) {  // halfPassed still false when this point is processed (:222-237)
            if (static_cast<double>(o) < static_cast<double>(startOri) - M_PI / 2)
                o = static_cast<float>(static_cast<double>(o) + 2 * M_PI);
            else if (static_cast<double>(o) > static_cast<double>(startOri) + M_PI * 3 / 2)
                o = static_cast<float>(static_cast<double>(o) - 2 * M_PI);
        } else {  // :238-249
            o = static_cast<float>(static_cast<double>(o) + 2 * M_PI);
            if (static_cast<double>(o) < static_cast<double>(endOri) - M_PI * 3 / 2)
                o = static_cast<float>(static_cast<double>(o) + 2 * M_PI);
            else if (static_cast<double>(o) > static_cast<double>(endOri) + M_PI / 2)
                o = static_cast<float>(static_cast<double>(o) - 2 * M_PI);
        }
        const float relTime = (o - startOri) / (endOri - startOri);                           // :251
        const float inten = static_cast<float>(static_cast<double>(r) + 0.1 * static_cast<double>(relTime));  // :252, scanPeriod = 0.1
        ox[pos] = in[(size_t)e * stride];
        oy[pos] = in[(size_t)e * stride + 1];
        oz[pos] = in[(size_t)e * stride + 2];
        oi[pos] = inten;
        src[pos] = e;
    }
}
SCAL_KERNEL(256, k_scatter)

__device__ __forceinline__ void k_curv_body(const FeatParams* __restrict__ P, const float* __restrict__ x, const float* __restrict__ y,
                                              const float* __restrict__ z, float* __restrict__ curv, int* __restrict__ label,
                                              unsigned char* __restrict__ gap, unsigned* __restrict__ box_parts) {
    const int n = P->n_kept;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // bounding box of the ordered cloud on the way, one part per block (the ScanContext keyframe filter starts from the parts
    // instead of two launches of its own)
    vox_bbox_block_store(box_parts, blockIdx.x, i < n, i < n ? x[i] : 0.f, i < n ? y[i] : 0.f, i < n ? z[i] : 0.f);
    if (i >= n) return;
    float c = 0.f;
    if (i >= 5 && i < n - 5) {  // :269-275, summed left to right exactly as written
        const float dx = x[i - 5] + x[i - 4] + x[i - 3] + x[i - 2] + x[i - 1] - 10 * x[i] + x[i + 1] + x[i + 2] + x[i + 3] + x[i + 4] + x[i + 5];
        const float dy = y[i - 5] + y[i - 4] + y[i - 3] + y[i - 2] + y[i - 1] - 10 * y[i] + y[i + 1] + y[i + 2] + y[i + 3] + y[i + 4] + y[i + 5];
        const float dz = z[i - 5] + z[i - 4] + z[i - 3] + z[i - 2] + z[i - 1] - 10 * z[i] + z[i + 1] + z[i + 2] + z[i + 3] + z[i + 4] + z[i + 5];
        c = dx * dx + dy * dy + dz * dz;
    }
    curv[i] = c;
    label[i] = 0;
    unsigned char g = 1;
    if (i + 1 < n) {  // squared gap between consecutive points > 0.05 stops the +-5 suppression (:334-340)
        const float dx = x[i + 1] - x[i], dy = y[i + 1] - y[i], dz = z[i + 1] - z[i];
        g = static_cast<double>(dx * dx + dy * dy + dz * dz) > 0.05 ? 1 : 0;
    }
    gap[i] = g;
}
SCAL_KERNEL(256, k_curv)

// Greedy picks of one segment (:304-403), executed by ONE wave.  `keys` = the segment's (curvature, ring-relative index)
// keys sorted ascending, flags[] = per-point LDS bytes: bit 0 = cloudNeighborPicked, bit 1 = "the squared gap to the NEXT
// point exceeds 0.05" (:334-340 / :344-351).
//
// A window of 64 candidates is held in registers together with everything a pick needs: each lane knows how far its own
// +-5 suppression would reach (the gap flags never change), so one pick is ballot -> readlane -> range test on registers.
// LDS is only written (for later windows and segments), never read back inside the window.
struct PickWindow {
    int li;    // ring-relative index of the candidate
    int nf;    // forward reach of its suppression: l = 1..nf
    int nb;    // backward reach: l = -1..-nb
    unsigned g; // gap flags of li-5 .. li+5 (bit d+5), so a pick can rewrite its neighbours' bytes without reading them
    bool cand; // passes the curvature test
    bool ok;   // cand and not picked yet
};

template <bool SHARP>
__device__ __forceinline__ PickWindow load_window(const unsigned long long* keys, int L, int base, const unsigned char* flags, int lane) {
    PickWindow w;
    const int p = base + lane;
    const bool valid = p < L;
    const unsigned long long k = valid ? keys[SHARP ? L - 1 - p : p] : 0ull;
    w.li = valid ? static_cast<int>(k & 0xffffffffu) : 5;
    const float c = __uint_as_float(static_cast<unsigned>(k >> 32));
    w.cand = valid && (SHARP ? static_cast<double>(c) > 0.1 : static_cast<double>(c) < 0.1);
    unsigned g = 0, own = 0;
#pragma unroll
    for (int d = -5; d <= 5; ++d) {
        const unsigned f = flags[w.li + d];
        g |= ((f >> 1) & 1u) << (d + 5);
        if (d == 0) own = f & 1u;
    }
    w.g = g;
    // forward l = 1..5 stops at the first gap between (li+l-1, li+l) => flag of li+l-1
    const unsigned gm = (g >> 5) & 0x1fu;
    w.nf = gm ? __ffs(static_cast<int>(gm)) - 1 : 5;
    // backward l = -1..-5 stops at the first gap between (li+l, li+l+1) => flag of li+l
    int nb = 5;
#pragma unroll
    for (int l = 5; l >= 1; --l)
        if ((g >> (5 - l)) & 1u) nb = l - 1;
    w.nb = nb;
    w.ok = w.cand && own == 0;
    return w;
}

// marks the picked point and its +-reach neighbours in LDS (one store: lane d + 5 owns offset d) and in the register window
__device__ __forceinline__ void apply_pick(PickWindow& w, int pli, int pnf, int pnb, unsigned pg, unsigned char* flags, int lane) {
    const int d = lane - 5;
    if (lane <= 10 && d <= pnf && -d <= pnb) flags[pli + d] = static_cast<unsigned char>(1u | (((pg >> lane) & 1u) << 1));
    if (w.li >= pli - pnb && w.li <= pli + pnf) w.ok = false;
}

// The picks of a phase stay in registers while the phase runs - lane n keeps the n-th pick - and reach `label` and the per-segment
// slot arrays afterwards, all at once: the loop body of a pick is ballot -> readlane -> one LDS store -> range test (rounds 1-3 also
// stored label and slots from lane 0 inside it, behind three exec-mask branches: 0.27 us per pick, 39 of k_ring's 73 us).
__device__ __forceinline__ void pick_segment(const unsigned long long* keys, int L, int rs, int seg, unsigned char* picked, int* __restrict__ label, int* __restrict__ seg_sharp, int* __restrict__ seg_less,
                                             int* __restrict__ seg_flat, int* __restrict__ seg_cnt, int lane) {
    // ---- sharp / lessSharp: largest curvature first (:304-357).  The 21st acceptable candidate ends the scan of the segment
    // without being marked (:327-330), so the loop simply stops after 20 picks.
    int n_pick = 0, my_pick = 0;
    for (int base = 0; base < L && n_pick < 20; base += 64) {
        PickWindow w = load_window<true>(keys, L, base, picked, lane);
        const uint64_t stopm = __ballot(base + lane < L && !w.cand);  // sorted: nothing after the first failure can pass
        const uint64_t live = stopm ? ((1ull << (__ffsll(static_cast<long long>(stopm)) - 1)) - 1ull) : ~0ull;
        while (n_pick < 20) {
            const uint64_t okm = __ballot(w.ok) & live;
            if (!okm) break;
            const int f = __ffsll(static_cast<long long>(okm)) - 1;
            const int pli = __builtin_amdgcn_readlane(w.li, f);
            const int pnf = __builtin_amdgcn_readlane(w.nf, f), pnb = __builtin_amdgcn_readlane(w.nb, f);
            const unsigned pg = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(w.g), f));
            my_pick = lane == n_pick ? pli : my_pick;
            ++n_pick;
            apply_pick(w, pli, pnf, pnb, pg, picked, lane);
        }
        __builtin_amdgcn_wave_barrier();
        if (stopm) break;
    }
    const int n_sh = min(n_pick, 2), n_ls = n_pick;
    if (lane < n_pick) {
        label[rs + my_pick] = lane < 2 ? 2 : 1;
        seg_less[seg * 20 + lane] = rs + my_pick;
        if (lane < 2) seg_sharp[seg * 2 + lane] = rs + my_pick;
    }
    // ---- flat: smallest curvature first (:359-403)
    int smallestPickedNum = 0, my_flat = 0;
    for (int base = 0; base < L; base += 64) {
        PickWindow w = load_window<false>(keys, L, base, picked, lane);
        const uint64_t stopm = __ballot(base + lane < L && !w.cand);
        const uint64_t live = stopm ? ((1ull << (__ffsll(static_cast<long long>(stopm)) - 1)) - 1ull) : ~0ull;
        bool done = false;
        while (true) {
            const uint64_t okm = __ballot(w.ok) & live;
            if (!okm) break;
            const int f = __ffsll(static_cast<long long>(okm)) - 1;
            const int pli = __builtin_amdgcn_readlane(w.li, f);
            const int pnf = __builtin_amdgcn_readlane(w.nf, f), pnb = __builtin_amdgcn_readlane(w.nb, f);
            const unsigned pg = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(w.g), f));
            my_flat = lane == smallestPickedNum ? pli : my_flat;
            smallestPickedNum++;
            if (smallestPickedNum >= 4) {  // the 4th is appended but neither marked nor suppressing (:371-375)
                done = true;
                break;
            }
            apply_pick(w, pli, pnf, pnb, pg, picked, lane);
        }
        __builtin_amdgcn_wave_barrier();
        if (done || stopm) break;
    }
    if (lane < smallestPickedNum) {
        label[rs + my_flat] = -1;
        seg_flat[seg * 4 + lane] = rs + my_flat;
    }
    if (lane == 0) {
        seg_cnt[seg * 3 + 0] = n_sh;
        seg_cnt[seg * 3 + 1] = n_ls;
        seg_cnt[seg * 3 + 2] = smallestPickedNum;
    }
}

__device__ __forceinline__ void k_ring_body(FeatParams* P, const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ z, const float* __restrict__ inten,
                                                       const float* __restrict__ curv, int* __restrict__ label,
                                                       const unsigned char* __restrict__ gap, int* __restrict__ seg_sharp,
                                                       int* __restrict__ seg_less, int* __restrict__ seg_flat, int* __restrict__ seg_cnt,
                                                       float* __restrict__ sx, float* __restrict__ sy, float* __restrict__ sz,
                                                       float* __restrict__ si) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds_raw);
    unsigned char* picked = lds_raw + sizeof(unsigned long long) * RING_MAX;  // RING_MAX + 16 bytes
    __shared__ int s_scan[17];
    __shared__ int s_misc[8];
    __shared__ float s_red[6][16];

    const int r = blockIdx.x;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    if (r == 32 && tid == 0) SCAL_STAMP(7);
    const int rs = P->ring_off[r], cnt = P->ring_count[r];
    const int start = rs + 5, end = rs + cnt - 6;
    if (tid < 18) seg_cnt[r * 18 + tid] = 0;
    if (tid == 0) P->lf_ring_cnt[r] = 0;
    if (end - start < 6) return;  // :292
    if (cnt > RING_MAX) {
        if (tid == 0) atomicExch(&P->error, SCAL_E_CAPACITY);
        return;
    }
    for (int t = tid; t < cnt + 16; t += blockDim.x) picked[t] = (t < cnt && gap[rs + t]) ? 2 : 0;
    __syncthreads();
    if (r == 32 && tid == 0) SCAL_STAMP(0);

    // segment bounds (:297-298)
    int seg_sp[6], seg_L[6];
    int maxL = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        seg_sp[j] = start + (end - start) * j / 6;
        seg_L[j] = start + (end - start) * (j + 1) / 6 - 1 - seg_sp[j] + 1;
        maxL = max(maxL, seg_L[j]);
    }
    const bool fast = maxL <= 512;  // every segment fits one wave's registers (8 keys per lane)
    if (fast) {
        // six waves sort the six segments concurrently in registers (:290-302), keys land in LDS at j*512
        if (wv < 6) {
            const int sp = seg_sp[wv], L = seg_L[wv];
            unsigned long long v[8];
            bool tie = false;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int t = lane * 8 + r;
                v[r] = t < L ? ((static_cast<unsigned long long>(__float_as_uint(curv[sp + t])) << 32) | static_cast<unsigned>(sp + t - rs)) : ~0ull;
            }
            wave_sort512(v);
            chunk_store(keys, wv, v);
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            for (int t = lane; t + 1 < L; t += 64) tie |= (keys[wv * 512 + t] >> 32) == (keys[wv * 512 + t + 1] >> 32);
            if (__ballot(tie) && lane == 0) atomicAdd(&P->n_tied, 1);
        }
        __syncthreads();
        if (r == 32 && tid == 0) SCAL_STAMP(1);
    }
    for (int j = 0; j < 6; ++j) {
        const int sp = seg_sp[j];
        const int L = seg_L[j];
        unsigned long long* skeys = keys + (fast ? j * 512 : 0);
        if (!fast) {
            const int Lp = max(512, next_pow2(L));
            for (int t = tid; t < Lp; t += blockDim.x) {
                unsigned long long k = ~0ull;
                if (t < L) k = (static_cast<unsigned long long>(__float_as_uint(curv[sp + t])) << 32) | static_cast<unsigned>(sp + t - rs);
                keys[t] = k;  // curvature >= 0, so its bit pattern orders like the float; ties fall back to the index
            }
            __syncthreads();
            block_sort_u64(keys, Lp, L);
            bool tie = false;
            for (int t = tid; t + 1 < L; t += blockDim.x) tie |= (keys[t] >> 32) == (keys[t + 1] >> 32);
            if (__syncthreads_or(tie) && tid == 0) atomicAdd(&P->n_tied, 1);
        }

        if (wv == 0) pick_segment(skeys, L, rs, r * 6 + j, picked, label, seg_sharp, seg_less, seg_flat, seg_cnt, lane);
        if (!fast) __syncthreads();  // the next segment's block-wide sort reuses the key buffer
    }
    __threadfence_block();
    __syncthreads();

    if (r == 32 && tid == 0) SCAL_STAMP(2);
    // ---- lessFlat of this ring (:405-411): every k in [start, end-1] with label <= 0, arrival order
    const int span = end - start;  // segments tile [start, end-1]
    const int per = (span + blockDim.x - 1) / blockDim.x;
    const int b0 = min(span, tid * per), b1 = min(span, b0 + per);
    int mine = 0;
    for (int t = b0; t < b1; ++t) mine += label[start + t] <= 0;
    int total = 0;
    int pos = block_exclusive_scan(mine, s_scan, &total);
    const int m = total;
    if (m == 0) return;  // uniform: nothing to downsample in this ring
    // bounding box of the candidates -> pcl::VoxelGrid min_b / overflow guard
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int t = b0; t < b1; ++t) {
        const int g = start + t;
        if (label[g] <= 0) {
            mn[0] = fminf(mn[0], x[g]), mx[0] = fmaxf(mx[0], x[g]);
            mn[1] = fminf(mn[1], y[g]), mx[1] = fmaxf(mx[1], y[g]);
            mn[2] = fminf(mn[2], z[g]), mx[2] = fmaxf(mx[2], z[g]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, 64));
        }
        if (lane == 0) s_red[a][wv] = mn[a], s_red[3 + a][wv] = mx[a];
    }
    __syncthreads();
    if (tid == 0) {
        float bmn[3], bmx[3];
        const int nw = blockDim.x >> 6;
        for (int a = 0; a < 3; ++a) {
            bmn[a] = s_red[a][0], bmx[a] = s_red[3 + a][0];
            for (int q = 1; q < nw; ++q) bmn[a] = fminf(bmn[a], s_red[a][q]), bmx[a] = fmaxf(bmx[a], s_red[3 + a][q]);
        }
        const float inv = 1.0f / 0.2f;  // setLeafSize(0.2,0.2,0.2) (:417) -> inverse_leaf_size_
        long long d[3];
        int mb[3];
        bool wide = false;
        for (int a = 0; a < 3; ++a) {
            d[a] = static_cast<long long>((bmx[a] - bmn[a]) * inv) + 1;
            mb[a] = static_cast<int>(floorf(bmn[a] * inv));
            const int xb = static_cast<int>(floorf(bmx[a] * inv));
            if (xb - mb[a] + 1 > 16383) wide = true;
        }
        const bool guard = m > 0 && d[0] * d[1] * d[2] > 2147483647ll;  // "Leaf size is too small": output = input
        s_misc[0] = mb[0], s_misc[1] = mb[1], s_misc[2] = mb[2];
        s_misc[3] = guard ? 1 : 0;
        if (!guard && wide && m > 0) atomicExch(&P->error, SCAL_E_CAPACITY);
    }
    __syncthreads();
    if (r == 32 && tid == 0) SCAL_STAMP(3);
    const int mb0 = s_misc[0], mb1 = s_misc[1], mb2 = s_misc[2];
    const bool guard = s_misc[3] != 0;
    const int Mp = max(512, next_pow2(m));
    for (int t = tid; t < Mp; t += blockDim.x) keys[t] = ~0ull;
    __syncthreads();
    {
        const float inv = 1.0f / 0.2f;
        for (int t = b0; t < b1; ++t) {
            const int g = start + t;
            if (label[g] <= 0) {
                unsigned long long k;
                if (guard) {
                    k = static_cast<unsigned long long>(pos) << 14;  // every point its own voxel, arrival order
                } else {
                    const unsigned long long i0 = static_cast<unsigned long long>(static_cast<int>(floorf(x[g] * inv)) - mb0) & 0x3fffu;
                    const unsigned long long i1 = static_cast<unsigned long long>(static_cast<int>(floorf(y[g] * inv)) - mb1) & 0x3fffu;
                    const unsigned long long i2 = static_cast<unsigned long long>(static_cast<int>(floorf(z[g] * inv)) - mb2) & 0x3fffu;
                    k = (i2 << 42) | (i1 << 28) | (i0 << 14);  // idx = i0 + i1*dx + i2*dx*dy orders like (i2, i1, i0)
                }
                keys[pos] = k | static_cast<unsigned long long>(g - rs);  // ring-relative index: arrival order inside a voxel
                ++pos;
            }
        }
    }
    __syncthreads();
    if (r == 32 && tid == 0) SCAL_STAMP(4);
    block_sort_u64(keys, Mp, m);
    if (r == 32 && tid == 0) SCAL_STAMP(5);
    // heads of voxel runs -> output slots
    const int per2 = (m + blockDim.x - 1) / blockDim.x;
    const int c0 = min(m, tid * per2), c1 = min(m, c0 + per2);
    int heads = 0;
    for (int t = c0; t < c1; ++t) heads += (t == 0) || ((keys[t] >> 14) != (keys[t - 1] >> 14));
    int n_out = 0;
    int opos = block_exclusive_scan(heads, s_scan, &n_out);
    for (int t = c0; t < c1; ++t) {
        if ((t == 0) || ((keys[t] >> 14) != (keys[t - 1] >> 14))) {
            const unsigned long long vk = keys[t] >> 14;
            float ax = 0.f, ay = 0.f, az = 0.f, ai = 0.f;
            int u = t;
            while (u < m && (keys[u] >> 14) == vk) {  // CentroidPoint: f32 sums in sorted order
                const int g = rs + static_cast<int>(keys[u] & 0x3fffu);
                ax += x[g], ay += y[g], az += z[g], ai += inten[g];
                ++u;
            }
            const float cntf = static_cast<float>(u - t);
            sx[rs + opos] = ax / cntf;
            sy[rs + opos] = ay / cntf;
            sz[rs + opos] = az / cntf;
            si[rs + opos] = ai / cntf;
            ++opos;
        }
    }
    if (tid == 0) P->lf_ring_cnt[r] = n_out;
    if (r == 32 && tid == 0) SCAL_STAMP(6);
}
SCAL_KERNEL(RING_THREADS, k_ring)

SCAL_DEFINE_STAMP_READER(scal_debug_stamps_features)

// one block: turn per-segment pick slots into the reference's emission order (segments in (ring, sixth) order)
// Last kernel of stage A.  Every workgroup derives the emission order itself - exclusive prefixes of the per-segment pick counts (<= 512
// segments, two per thread) and of the per-ring lessFlat counts, in LDS - instead of reading them from a one-workgroup launch in front
// (k_finalize until round 3: a launch, its gap and 9 us, against ~5 us more in this kernel).  Then: ring-staged lessFlat centroids -> the
// contiguous cloud; the i-th sharp / lessSharp / flat pick in the reference's order (segment by segment, :304-403) -> the index lists
// and its xyzi record; workgroup 0 also writes the counts.
constexpr int SEG_MAX = 512;
__device__ __forceinline__ void k_compact_body(FeatParams* P, int n_scans, const int* __restrict__ seg_sharp, const int* __restrict__ seg_less,
                                                 const int* __restrict__ seg_flat, const int* __restrict__ seg_cnt, int* __restrict__ sharp,
                                                 int* __restrict__ less, int* __restrict__ flat, const float* __restrict__ sx,
                                                 const float* __restrict__ sy, const float* __restrict__ sz, const float* __restrict__ si,
                                                 float* __restrict__ lx, float* __restrict__ ly, float* __restrict__ lz, float* __restrict__ li,
                                                 const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                 const float* __restrict__ inten, float* __restrict__ fsharp, float* __restrict__ fless,
                                                 float* __restrict__ fflat) {
    __shared__ int s_scan[17];
    __shared__ int s_o[3][SEG_MAX + 1];  // exclusive prefixes of the sharp / lessSharp / flat counts per segment, total at [nseg]
    __shared__ int s_lf[65];             // exclusive prefix of the lessFlat counts per ring
    const int nseg = min(n_scans * 6, SEG_MAX);
    const int t = threadIdx.x;
    // two block scans carry the four prefix sums: (lessSharp | flat << 14) and (sharp | lessFlat << 11) - a scan is three barriers,
    // and every workgroup pays them.  Totals: lessSharp <= 20 * 512 < 2^14, flat <= 4 * 512, sharp <= 2 * 512 < 2^11, lessFlat < 2^21.
    int tot[3], tl;
    {
        const int i0 = 2 * t, i1 = 2 * t + 1;
        const int a1 = i0 < nseg ? seg_cnt[i0 * 3 + 1] : 0, b1 = i1 < nseg ? seg_cnt[i1 * 3 + 1] : 0;
        const int a2 = i0 < nseg ? seg_cnt[i0 * 3 + 2] : 0, b2 = i1 < nseg ? seg_cnt[i1 * 3 + 2] : 0;
        const int a0 = i0 < nseg ? seg_cnt[i0 * 3 + 0] : 0, b0 = i1 < nseg ? seg_cnt[i1 * 3 + 0] : 0;
        const int lc = t < 64 ? P->lf_ring_cnt[t] : 0;
        int totA, totB;
        const int exA = block_exclusive_scan((a1 + b1) | ((a2 + b2) << 14), s_scan, &totA);
        const int exB = block_exclusive_scan((a0 + b0) | (lc << 11), s_scan, &totB);
        const int e1 = exA & 0x3fff, e2 = exA >> 14, e0 = exB & 0x7ff, el = static_cast<int>(static_cast<unsigned>(exB) >> 11);
        if (i0 < nseg) s_o[0][i0] = e0, s_o[1][i0] = e1, s_o[2][i0] = e2;
        if (i1 < nseg) s_o[0][i1] = e0 + a0, s_o[1][i1] = e1 + a1, s_o[2][i1] = e2 + a2;
        if (t < 64) s_lf[t] = el;
        tot[1] = totA & 0x3fff, tot[2] = totA >> 14, tot[0] = totB & 0x7ff, tl = static_cast<int>(static_cast<unsigned>(totB) >> 11);
        if (t == 0) s_o[0][nseg] = tot[0], s_o[1][nseg] = tot[1], s_o[2][nseg] = tot[2], s_lf[64] = tl;
    }
    __syncthreads();
    if (blockIdx.x == 0) {  // ring offsets and counts for the host and the later stages
        if (t < 64) P->lf_ring_off[t] = s_lf[t];
        if (t == 0) {
            P->n_sharp = tot[0], P->n_less_sharp = tot[1], P->n_flat = tot[2], P->n_less_flat = tl;
            P->lf_ring_off[64] = tl;
        }
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = P->n_kept;
    if (i < n) {
        int lo = 0, hi = 63;  // ring of ordered index i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (P->ring_off[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const int loc = i - P->ring_off[lo];
        if (loc < s_lf[lo + 1] - s_lf[lo]) {
            const int o = s_lf[lo] + loc;
            lx[o] = sx[i], ly[o] = sy[i], lz[o] = sz[i], li[o] = si[i];
        }
    }
    // xyzi records of the picked points for the downstream stages: the i-th pick of a class lives in segment upper_bound(prefix, i) - 1
    auto pick = [&](int q, int per_seg, const int* __restrict__ seg_list) {
        int lo = 0, hi = nseg;  // largest sgi with s_o[q][sgi] <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_o[q][mid] <= i) lo = mid; else hi = mid;
        }
        return seg_list[lo * per_seg + (i - s_o[q][lo])];
    };
    if (i < tot[0]) {
        const int g = pick(0, 2, seg_sharp);
        sharp[i] = g;
        fsharp[i * 4 + 0] = x[g], fsharp[i * 4 + 1] = y[g], fsharp[i * 4 + 2] = z[g], fsharp[i * 4 + 3] = inten[g];
    }
    if (i < tot[1]) {
        const int g = pick(1, 20, seg_less);
        less[i] = g;
        fless[i * 4 + 0] = x[g], fless[i * 4 + 1] = y[g], fless[i * 4 + 2] = z[g], fless[i * 4 + 3] = inten[g];
    }
    if (i < tot[2]) {
        const int g = pick(2, 4, seg_flat);
        flat[i] = g;
        fflat[i * 4 + 0] = x[g], fflat[i * 4 + 1] = y[g], fflat[i * 4 + 2] = z[g], fflat[i * 4 + 3] = inten[g];
    }
}
SCAL_KERNEL(256, k_compact)

__device__ __forceinline__ void k_interleave_body(const int* __restrict__ d_n, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                             const float* __restrict__ w, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *d_n) reinterpret_cast<float4*>(out)[i] = make_float4(x[i], y[i], z[i], w[i]);
}
SCAL_KERNEL(1024, k_interleave)

}  // namespace scal

using namespace scal;

struct scal_features {
    scal_features_config cfg;
    hipStream_t stream = nullptr;
    // last read of this context's buffers by each consumer stream (B, C, C's prefetch, D may all run on their own)
    static constexpr int MAX_READERS = 8;
    hipStream_t reader_stream[MAX_READERS] = {};
    hipEvent_t reader_ev[MAX_READERS] = {};
    bool reader_pending[MAX_READERS] = {};
    hipEvent_t done_ev = nullptr;    // end of the most recent run, recorded on demand
    bool done_recorded = false;
    unsigned generation = 0;  // runs so far (FeatDeviceView::generation)
    bool cross_stream_consumers = false;
    std::mutex ev_mu;  // consumers may register from different host threads  // seen once: record done_ev right behind every run, before later main-stream work
    int cap = 0, nb_cap = 0;
    DevBuf<float> d_in;
    // pinned staging of host scans (two slots, each guarded by the event of the upload that last read it): the upload is an
    // ordinary stream-ordered copy and the caller's buffer is free again when scal_features_run returns
    PinBuf<float> h_in[2];
    hipEvent_t up_ev[2] = {};
    bool up_used[2] = {};
    int up_next = 0;
    DevBuf<signed char> d_ring;
    DevBuf<float> d_ori;
    DevBuf<int> d_hist;
    DevBuf<float> ox, oy, oz, oi;
    DevBuf<int> d_src;
    DevBuf<float> d_curv;
    DevBuf<int> d_label;
    DevBuf<unsigned char> d_gap;
    DevBuf<int> seg_sharp, seg_less, seg_flat, seg_cnt;
    DevBuf<int> d_sharp, d_less, d_flat;
    DevBuf<float> sx, sy, sz, si;  // ring-staged lessFlat centroids
    DevBuf<float> lx, ly, lz, li;  // lessFlat cloud
    DevBuf<float> f_sharp, f_less, f_flat;  // xyzi of the picked points
    DevBuf<float> d_aos;  // AoS staging for fetch
    DevBuf<float> d_aos2; // the lessFlat cloud's, so that both clouds leave in one go (first fetch)
    HostStage hs;         // pinned landing area of the fetch (first fetch)
    DevBuf<FeatParams> d_P;
    DevBuf<unsigned> d_boxparts;  // [ceil(cap / 256)][6]
    int n_box_parts = 0;
    PinBuf<FeatParams> h_P;
    bool ran = false;
    int last_n = 0;
};

namespace scal {
FeatDeviceView features_view(scal_features* c) {
    FeatDeviceView v;
    v.P = c->d_P.p;
    v.x = c->ox.p, v.y = c->oy.p, v.z = c->oz.p, v.i = c->oi.p;
    v.lfx = c->lx.p, v.lfy = c->ly.p, v.lfz = c->lz.p, v.lfi = c->li.p;
    v.sharp_xyzi = c->f_sharp.p, v.less_xyzi = c->f_less.p, v.flat_xyzi = c->f_flat.p;
    v.box_parts = c->d_boxparts.p, v.n_box_parts = c->n_box_parts;
    v.cap = c->cap;
    v.stream = c->stream;
    v.device = c->cfg.device;
    v.n_scans = c->cfg.n_scans;
    {
        std::lock_guard<std::mutex> lk(c->ev_mu);
        v.generation = c->generation;
    }
    return v;
}
int features_wait_done(scal_features* c, hipStream_t consumer_stream) {
    if (consumer_stream == c->stream) return SCAL_OK;
    std::lock_guard<std::mutex> lk(c->ev_mu);
    if (!c->done_ev) SCAL_HIP(hipEventCreateWithFlags(&c->done_ev, EV_DEVICE_ONLY));
    c->cross_stream_consumers = true;
    if (!c->done_recorded) {
        // At once, also under a recorder: `done_recorded` tells the consumers on OTHER host threads (the pipeline queues stage B, the
        // side lane and stage C from different threads) that they may wait for the event, and a record still sitting in this
        // thread's list would let them wait for the event's previous state.  Recording later than stage A's last kernel is harmless.
        SCAL_HIP(hipEventRecord(c->done_ev, c->stream));
        c->done_recorded = true;
    }
    SCAL_HIP(op_stream_wait_event(consumer_stream, c->done_ev, 0));
    return SCAL_OK;
}
int features_note_reader(scal_features* c, hipStream_t consumer_stream) {
    if (consumer_stream == c->stream) return SCAL_OK;
    std::lock_guard<std::mutex> lk(c->ev_mu);
    int slot = -1;
    for (int i = 0; i < scal_features::MAX_READERS && slot < 0; ++i)
        if (c->reader_stream[i] == consumer_stream || c->reader_stream[i] == nullptr) slot = i;
    if (slot < 0) {
        set_error("features context: more than %d consumer streams", scal_features::MAX_READERS);
        return SCAL_E_STATE;
    }
    c->reader_stream[slot] = consumer_stream;
    if (!c->reader_ev[slot]) SCAL_HIP(hipEventCreateWithFlags(&c->reader_ev[slot], EV_DEVICE_ONLY));
    SCAL_HIP(op_event_record(c->reader_ev[slot], consumer_stream));
    c->reader_pending[slot] = true;
    return SCAL_OK;
}
}  // namespace scal

extern "C" int scal_features_create(const scal_features_config* cfg, scal_features_t** out) {
    if (!cfg || !out) {
        set_error("scal_features_create: null argument");
        return SCAL_E_ARG;
    }
    *out = nullptr;
    if (cfg->n_scans != 16 && cfg->n_scans != 32 && cfg->n_scans != 64) {
        set_error("only 16, 32 or 64 scan lines are supported (scanRegistration.cpp:486-490), got %d", cfg->n_scans);
        return SCAL_E_SCAN_LINE;
    }
    const bool type_ok = (cfg->lidar_type == SCAL_VLP16 && cfg->n_scans == 16) || (cfg->lidar_type == SCAL_HDL32 && cfg->n_scans == 32) ||
                         (cfg->lidar_type == SCAL_HDL64 && cfg->n_scans == 64) || (cfg->lidar_type == SCAL_OS1_64 && cfg->n_scans == 64);
    if (!type_ok) {
        set_error("lidar_type %d does not match scan_line %d (scanRegistration.cpp:171-218: wrong scan number)", cfg->lidar_type, cfg->n_scans);
        return SCAL_E_LIDAR_TYPE;
    }
    if (cfg->max_points <= 0 || cfg->max_points > 400000) {
        set_error("max_points must be in (0, 400000] (scanRegistration.cpp:68-71)");
        return SCAL_E_ARG;
    }
    SCAL_TRY(select_device(cfg->device));
    auto* c = new scal_features();
    c->cfg = *cfg;
    c->cap = cfg->max_points;
    c->nb_cap = div_up(c->cap, TILE);
    const int cap = c->cap, ns = cfg->n_scans;
    int rc = SCAL_OK;
    auto A = [&](int r) { if (rc == SCAL_OK) rc = r; };
    A(c->d_in.alloc((size_t)cap * 8));  // up to 32-byte point stride
    A(c->h_in[0].alloc((size_t)cap * 8));
    A(c->h_in[1].alloc((size_t)cap * 8));
    A(c->d_ring.alloc(cap));
    A(c->d_ori.alloc(cap));
    A(c->d_hist.alloc((size_t)64 * c->nb_cap));
    A(c->ox.alloc(cap)); A(c->oy.alloc(cap)); A(c->oz.alloc(cap)); A(c->oi.alloc(cap));
    A(c->d_src.alloc(cap));
    A(c->d_curv.alloc(cap));
    A(c->d_label.alloc(cap));
    A(c->d_gap.alloc(cap + 16));
    A(c->seg_sharp.alloc(ns * 6 * 2)); A(c->seg_less.alloc(ns * 6 * 20)); A(c->seg_flat.alloc(ns * 6 * 4)); A(c->seg_cnt.alloc(64 * 18));
    A(c->d_sharp.alloc(ns * 6 * 2)); A(c->d_less.alloc(ns * 6 * 20)); A(c->d_flat.alloc(ns * 6 * 4));
    A(c->sx.alloc(cap)); A(c->sy.alloc(cap)); A(c->sz.alloc(cap)); A(c->si.alloc(cap));
    A(c->lx.alloc(cap)); A(c->ly.alloc(cap)); A(c->lz.alloc(cap)); A(c->li.alloc(cap));
    A(c->f_sharp.alloc(ns * 6 * 2 * 4)); A(c->f_less.alloc(ns * 6 * 20 * 4)); A(c->f_flat.alloc(ns * 6 * 4 * 4));
    A(c->d_aos.alloc((size_t)cap * 4));
    A(c->d_P.alloc(1));
    A(c->d_boxparts.alloc((size_t)6 * (div_up(cap, 256) + 1)));
    A(c->h_P.alloc(1));
    if (rc == SCAL_OK && acquire_stream(c->cfg.device, &c->stream) != SCAL_OK) {
        set_error("hipStreamCreate failed");
        rc = SCAL_E_HIP;
    }
    // Initialisation goes through the context's own stream.  hipMemset on the legacy null stream returns before the fill has
    // run and is not ordered against a hipStreamNonBlocking stream: a fill landing after the first k_pre cleared
    // FeatParams::empty again (the E_EMPTY that test_errors once missed).
    if (rc == SCAL_OK && (op_memset_async(c->d_P.p, 0, sizeof(FeatParams), c->stream) != hipSuccess || op_stream_synchronize(c->stream) != hipSuccess))
        rc = SCAL_E_HIP;
    for (int k = 0; k < 2 && rc == SCAL_OK; ++k)
        if (hipEventCreateWithFlags(&c->up_ev[k], hipEventDisableTiming) != hipSuccess) rc = SCAL_E_HIP;
    if (rc == SCAL_OK) {
        const int lds = sizeof(unsigned long long) * RING_MAX + RING_MAX + 16;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("cannot reserve %d bytes of LDS for k_ring", lds);
            rc = SCAL_E_HIP;
        }
    }
    if (rc != SCAL_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return SCAL_OK;
}

extern "C" void scal_features_destroy(scal_features_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    for (int i = 0; i < scal_features::MAX_READERS; ++i)
        if (c->reader_ev[i]) {  // consumers on other streams must be done with the buffers
            (void)op_event_synchronize(c->reader_ev[i]);
            (void)hipEventDestroy(c->reader_ev[i]);
        }
    if (c->done_ev) (void)hipEventDestroy(c->done_ev);
    for (int k = 0; k < 2; ++k)
        if (c->up_ev[k]) (void)hipEventDestroy(c->up_ev[k]);
    if (c->stream) {
        (void)op_stream_synchronize(c->stream);
        release_stream(c->cfg.device);
    }
    delete c;
}

static int launch_chain(scal_features* c, const float* d_xyz, int n, int stride) {
    const scal_features_config& g = c->cfg;
    hipStream_t s = c->stream;
    KCfg k;
    k.lidar_type = g.lidar_type, k.n_scans = g.n_scans, k.float_math = g.float_math, k.check_finite = g.check_finite;
    const float thres = static_cast<float>(g.minimum_range);
    k.thres2 = thres * thres;
    // Launch shapes follow the point count rounded up to 8192 (blocks beyond the cloud find nothing to do), so that scans of similar
    // size - the sequences of a multi-sequence pipeline - get identical shapes and can share launches (batch.hpp).
    const int nq = min(c->cap, div_up(max(n, 1), 8192) * 8192);
    const int nb = min(c->nb_cap, max(1, div_up(nq, TILE)));
    FeatParams* P = c->d_P.p;
    std::unique_lock<std::mutex> ev_lk(c->ev_mu);
    for (int i = 0; i < scal_features::MAX_READERS; ++i)
        if (c->reader_pending[i]) {  // a consumer on another stream may still be reading the previous scan's outputs
            SCAL_HIP(op_stream_wait_event(s, c->reader_ev[i], 0));
            c->reader_pending[i] = false;
        }
    c->done_recorded = false;
    c->generation++;
    ev_lk.unlock();
    SCAL_LAUNCH("k_pre", k_pre, dim3(1), dim3(256), 0, s, d_xyz, n, stride, k, P);
    SCAL_LAUNCH("k_classify", k_classify, dim3(nb), dim3(256), 0, s, d_xyz, n, stride, k, P, c->d_ring.p, c->d_ori.p, c->d_hist.p, nb);
    SCAL_LAUNCH("k_ringscan", k_ringscan, dim3(1), dim3(1024), 0, s, c->d_hist.p, nb, g.n_scans, P);
    SCAL_LAUNCH("k_scatter", k_scatter, dim3(nb), dim3(256), 0, s, d_xyz, n, stride, P, c->d_ring.p, c->d_ori.p, c->d_hist.p, nb, c->ox.p, c->oy.p,
                       c->oz.p, c->oi.p, c->d_src.p);
    const int nb256 = max(1, div_up(nq, 256));
    SCAL_LAUNCH("k_curv", k_curv, dim3(nb256), dim3(256), 0, s, P, c->ox.p, c->oy.p, c->oz.p, c->d_curv.p, c->d_label.p, c->d_gap.p, c->d_boxparts.p);
    c->n_box_parts = nb256;
    const int lds = sizeof(unsigned long long) * RING_MAX + RING_MAX + 16;
    {
        SCAL_LAUNCH("k_ring", k_ring, dim3(g.n_scans), dim3(RING_THREADS), lds, s, P, c->ox.p, c->oy.p, c->oz.p, c->oi.p, c->d_curv.p, c->d_label.p,
                       c->d_gap.p, c->seg_sharp.p, c->seg_less.p, c->seg_flat.p, c->seg_cnt.p, c->sx.p, c->sy.p, c->sz.p, c->si.p);
    }
    SCAL_LAUNCH("k_compact", k_compact, dim3(nb256), dim3(256), 0, s, P, g.n_scans, c->seg_sharp.p, c->seg_less.p, c->seg_flat.p, c->seg_cnt.p, c->d_sharp.p,
                       c->d_less.p, c->d_flat.p, c->sx.p, c->sy.p, c->sz.p, c->si.p, c->lx.p, c->ly.p, c->lz.p, c->li.p, c->ox.p, c->oy.p, c->oz.p, c->oi.p,
                       c->f_sharp.p, c->f_less.p, c->f_flat.p);
    SCAL_HIP(hipGetLastError());
    ev_lk.lock();
    if (c->cross_stream_consumers) {  // the event must sit right behind stage A, not behind whatever the stream gets next
        SCAL_HIP(op_event_record(c->done_ev, s));
        c->done_recorded = true;
    }
    ev_lk.unlock();
    c->ran = true;
    c->last_n = n;
    return SCAL_OK;
}

extern "C" int scal_features_run_device(scal_features_t* c, const float* d_xyz, int n, int stride_floats) {
    if (!c || (!d_xyz && n > 0) || n < 0 || stride_floats < 3) {
        set_error("scal_features_run_device: bad argument");
        return SCAL_E_ARG;
    }
    if (n > c->cap) {
        set_error("scan has %d points, capacity is %d (the reference's scratch holds 400000)", n, c->cap);
        return SCAL_E_TOO_MANY;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    return launch_chain(c, d_xyz, n, stride_floats);
}

extern "C" int scal_features_sync(scal_features_t* c) {
    if (!c) return SCAL_E_ARG;
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_HIP(op_stream_synchronize(c->stream));
    return SCAL_OK;
}

extern "C" int scal_features_fetch(scal_features_t* c, scal_features_out* o) {
    if (!c || !o) {
        set_error("scal_features_fetch: null argument");
        return SCAL_E_ARG;
    }
    if (!c->ran) {
        set_error("scal_features_fetch before any run");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    SCAL_HIP(op_memcpy_async(c->h_P.p, c->d_P.p, sizeof(FeatParams), hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    const FeatParams& P = *c->h_P.p;
    o->n_kept = o->n_sharp = o->n_less_sharp = o->n_flat = o->n_less_flat = 0;
    o->n_tied_segments = 0;
    if (P.empty) {
        set_error("no point survives the NaN / minimum_range filters");
        return SCAL_E_EMPTY;
    }
    if (P.error) {
        set_error("device capacity exceeded (a ring longer than %d points or a voxel extent beyond 16383 cells)", RING_MAX);
        return P.error;
    }
    o->n_kept = P.n_kept, o->n_sharp = P.n_sharp, o->n_less_sharp = P.n_less_sharp, o->n_flat = P.n_flat, o->n_less_flat = P.n_less_flat;
    o->n_tied_segments = P.n_tied;
    const int nk = P.n_kept;
    const int nb = max(1, div_up(nk, 256));
    // everything the caller asked for lands in pinned memory with one asynchronous copy each and goes to the caller's arrays after ONE
    // synchronisation (four, and copies into pageable memory, until round 3: 1.3 of the registration node's 1.5 ms per scan)
    if (!c->d_aos2.p && o->cloud && o->less_flat) SCAL_TRY(c->d_aos2.alloc((size_t)c->cap * 4));
    (void)c->hs.reserve((size_t)c->cap * 44 + (size_t)c->cfg.n_scans * 156 * 4 + 4096);
    float* aos_lf = (o->cloud && c->d_aos2.p) ? c->d_aos2.p : c->d_aos.p;
    if (o->cloud && nk) {
        SCAL_LAUNCH("k_interleave", k_interleave, dim3(nb), dim3(256), 0, s, &c->d_P.p->n_kept, c->ox.p, c->oy.p, c->oz.p, c->oi.p, c->d_aos.p);
        SCAL_HIP(c->hs.d2h(o->cloud, c->d_aos.p, sizeof(float) * 4 * nk, s));
    }
    if (o->less_flat && P.n_less_flat) {
        SCAL_LAUNCH("k_interleave", k_interleave, dim3(max(1, div_up(P.n_less_flat, 256))), dim3(256), 0, s, &c->d_P.p->n_less_flat, c->lx.p, c->ly.p,
                           c->lz.p, c->li.p, aos_lf);
        SCAL_HIP(c->hs.d2h(o->less_flat, aos_lf, sizeof(float) * 4 * P.n_less_flat, s));
    }
    if (o->src_index && nk) SCAL_HIP(c->hs.d2h(o->src_index, c->d_src.p, sizeof(int) * nk, s));
    if (o->curvature && nk) SCAL_HIP(c->hs.d2h(o->curvature, c->d_curv.p, sizeof(float) * nk, s));
    if (o->label && nk) SCAL_HIP(c->hs.d2h(o->label, c->d_label.p, sizeof(int) * nk, s));
    if (o->sharp && P.n_sharp) SCAL_HIP(c->hs.d2h(o->sharp, c->d_sharp.p, sizeof(int) * P.n_sharp, s));
    if (o->less_sharp && P.n_less_sharp) SCAL_HIP(c->hs.d2h(o->less_sharp, c->d_less.p, sizeof(int) * P.n_less_sharp, s));
    if (o->flat && P.n_flat) SCAL_HIP(c->hs.d2h(o->flat, c->d_flat.p, sizeof(int) * P.n_flat, s));
    SCAL_HIP(op_stream_synchronize(s));
    c->hs.finish();
    if (o->ring_start) std::memcpy(o->ring_start, P.scan_start, sizeof(int) * c->cfg.n_scans);
    if (o->ring_end) std::memcpy(o->ring_end, P.scan_end, sizeof(int) * c->cfg.n_scans);
    return SCAL_OK;
}

static int features_run_host(scal_features_t* c, const void* xyz, int n, int stride_bytes, scal_features_out* out, bool no_wait) {
    if (!c || (!xyz && n > 0) || n < 0 || stride_bytes < 12 || (stride_bytes % 4) != 0 || stride_bytes > 32) {
        set_error("scal_features_run: bad argument (stride_bytes must be a multiple of 4 in [12, 32])");
        return SCAL_E_ARG;
    }
    if (n > c->cap) {
        set_error("scan has %d points, capacity is %d (the reference's scratch holds 400000)", n, c->cap);
        return SCAL_E_TOO_MANY;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    if (n > 0) {
        // The caller's (usually pageable) buffer is staged through pinned memory, so the upload is an ordinary stream-ordered
        // copy.  Two slots: the only wait is for the upload issued two calls ago, never for the work queued on the stream.
        const int slot = c->up_next;
        c->up_next ^= 1;
        if (c->up_used[slot]) SCAL_HIP(op_event_synchronize(c->up_ev[slot]));
        std::memcpy(c->h_in[slot].p, xyz, (size_t)n * stride_bytes);
        SCAL_HIP(op_memcpy_async(c->d_in.p, c->h_in[slot].p, (size_t)n * stride_bytes, hipMemcpyHostToDevice, c->stream));
        SCAL_HIP(op_event_record(c->up_ev[slot], c->stream));
        c->up_used[slot] = true;
    }
    SCAL_TRY(launch_chain(c, c->d_in.p, n, stride_bytes / 4));
    if (no_wait) return SCAL_OK;
    if (out) return scal_features_fetch(c, out);
    return scal_features_sync(c);
}

extern "C" int scal_features_run(scal_features_t* c, const void* xyz, int n, int stride_bytes, scal_features_out* out) {
    return features_run_host(c, xyz, n, stride_bytes, out, false);
}

extern "C" int scal_features_enqueue_host(scal_features_t* c, const void* xyz, int n, int stride_bytes) {
    return features_run_host(c, xyz, n, stride_bytes, nullptr, true);
}

extern "C" void* scal_features_stream(scal_features_t* c) { return c ? static_cast<void*>(c->stream) : nullptr; }
