// Stage D on gfx950: ScanContext descriptor construction, ring-key top-3 and column-shift cosine distance.
// Replaces SCManager, /root/reference/include/scancontext/Scancontext.cpp (:151-195 makeScancontext, :198-227 keys,
// :69-148 distances, :336-427 detectLoopClosureID); call sites laserPosegraphOptimization.cpp:639, :718.
//
// Database layout in HBM (one slot per keyframe, slot = global index / n_shards):
//   desc   [cap][1200] f64, column-major 20x60 exactly as Eigen stores it (9.6 KB per keyframe)
//   rkey   [cap][20]   f32 ring key (row means cast to float, Scancontext.cpp:62-66) — what the kd-tree indexes
//   skey   [cap][60]   f64 sector key (column means)
//   cnorm  [cap][60]   f64 column norms (hoisted out of distDirectSC, :78-81)
// Kernels: k_sc_bin + k_sc_finish (LDS atomic-max polar binning on an order-preserving integer image of the f32
// height, merged across workgroups with global atomic-max, then keys/norms), k_sc_topk (brute-force ring-key distances in nanoflann's f32 accumulation
// order, per-block top-3), k_sc_detect (merge top-3, one wave per candidate runs the sector-key alignment and the
// 7-shift cosine distance), k_sc_pairs / k_sc_matrix (batched distances).
// All reductions that the reference does through Eigen follow Eigen 3.3's SSE2 order: four interleaved partial
// sums, (s0+s2)+(s1+s3) — see oracle/scancontext.cpp.
#include "common.hpp"
#include "device_utils.hpp"
#include "voxel_dev.hpp"
#include "features_dev.hpp"
#include <cmath>
#include <cfloat>

namespace scal {

constexpr int NR = 20, NS = 60, DESC = NR * NS;

struct SCRec {  // per-candidate record produced on the device
    float key_dist;
    int idx;
    double sc_dist;
    int shift;
    int pad;
};

// Eigen redux order for sizes divisible by 4: lanes j = i mod 4 summed sequentially, (s0+s2)+(s1+s3)
template <class F>
__device__ __forceinline__ double eigen_sum4(int size, F at) {
    double s0 = at(0), s1 = at(1), s2 = at(2), s3 = at(3);
    for (int i = 4; i < size; i += 4) {
        s0 += at(i), s1 += at(i + 1);
        s2 += at(i + 2), s3 += at(i + 3);
    }
    return (s0 + s2) + (s1 + s3);
}

// xy2theta, Scancontext.cpp:23-36 (float in, float out)
__device__ __forceinline__ float xy2theta(float x, float y, int float_math) {
    auto at = [&](float v) -> double {
        const double a = atan(static_cast<double>(v));
        return float_math ? static_cast<double>(static_cast<float>(a)) : a;
    };
    if ((x >= 0) & (y >= 0)) return static_cast<float>((180 / M_PI) * at(y / x));
    if ((x < 0) & (y >= 0)) return static_cast<float>(180 - ((180 / M_PI) * at(y / (-x))));
    if ((x < 0) & (y < 0)) return static_cast<float>(180 + ((180 / M_PI) * at(y / x)));
    if ((x >= 0) & (y < 0)) return static_cast<float>(360 - ((180 / M_PI) * at((-y) / x)));
    return NAN;
}

__device__ __forceinline__ int ceil_to_int(double v) {  // int(ceil(v)); NaN -> INT_MIN like cvttsd2si
    const double c = ceil(v);
    if (!(c == c)) return INT_MIN;
    if (c >= 2147483648.0 || c < -2147483648.0) return INT_MIN;
    return static_cast<int>(c);
}

// points may be AoS xyzi (stride 4, y=z=null) or SoA.  Two launches: k_sc_bin (many workgroups: LDS atomic-max polar
// binning on an order-preserving integer image of the f32 height, merged into a global 1200-cell image) and k_sc_finish
// (one workgroup: descriptor, keys, column norms; clears the global image again so it is always zero between scans).
__device__ __forceinline__ void k_sc_bin_body(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz, int stride,
                                                const int* __restrict__ d_n, int n_host, double max_radius, int float_math,
                                                unsigned* __restrict__ gcell) {
    __shared__ unsigned cell[DESC];
    const int n = d_n ? *d_n : n_host;
    if (static_cast<int>(blockIdx.x * blockDim.x) >= n) return;
    for (int i = threadIdx.x; i < DESC; i += blockDim.x) cell[i] = 0u;
    __syncthreads();
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        float x, y, z;
        if (py) {
            x = px[k], y = py[k], z = pz[k];
        } else {
            x = px[(size_t)k * stride], y = px[(size_t)k * stride + 1], z = px[(size_t)k * stride + 2];
        }
        const float zz = static_cast<float>(static_cast<double>(z) + 2.0);  // LIDAR_HEIGHT (Scancontext.h:83, cpp:168)
        const float azim_range = sqrtf(x * x + y * y);                      // :171
        const float azim_angle = xy2theta(x, y, float_math);                // :172
        if (static_cast<double>(azim_range) > max_radius) continue;         // :175
        if (zz != zz) continue;                                             // desc < NaN is false
        const int ring_idx = max(min(NR, ceil_to_int((static_cast<double>(azim_range) / max_radius) * NR)), 1);    // :178
        const int sctor_idx = max(min(NS, ceil_to_int((static_cast<double>(azim_angle) / 360.0) * NS)), 1);        // :179
        atomicMax(&cell[(ring_idx - 1) + NR * (sctor_idx - 1)], float_to_ordered(zz));                             // :182-183
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DESC; i += blockDim.x)
        if (cell[i]) atomicMax(&gcell[i], cell[i]);
}
SCAL_KERNEL(256, k_sc_bin)

struct SCSlot {  // one keyframe's record: descriptor (column-major 20x60), ring key, sector key, column norms
    double* desc;
    float* rkey;
    double* skey;
    double* cnorm;
};

// writes the record to slot a and, when b.desc != null, to slot b as well (query staging + database slot in one pass)
__device__ __forceinline__ void k_sc_finish_body(unsigned* __restrict__ gcell, const SCSlot& a, const SCSlot& b) {
    __shared__ double d[DESC];
    for (int i = threadIdx.x; i < DESC; i += blockDim.x) {
        const unsigned c = gcell[i];
        gcell[i] = 0u;
        double v = 0.0;  // empty cell: -1000 -> 0 (:187-190)
        if (c != 0u) {
            const float f = ordered_to_float(c);
            if (static_cast<double>(f) > -1000.0) v = static_cast<double>(f);
        }
        d[i] = v;
        a.desc[i] = v;
        if (b.desc) b.desc[i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NR) {  // makeRingkeyFromScancontext (:198-211) + eig2stdvec cast (:62-66)
        const int r = threadIdx.x;
        const float m = static_cast<float>(eigen_sum4(NS, [&](int c) { return d[r + NR * c]; }) / NS);
        a.rkey[r] = m;
        if (b.desc) b.rkey[r] = m;
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + NS) {  // makeSectorkeyFromScancontext (:214-227), column norms
        const int c = threadIdx.x - 64;
        const double sk = eigen_sum4(NR, [&](int r) { return d[r + NR * c]; }) / NR;
        const double cn = sqrt(eigen_sum4(NR, [&](int r) { return d[r + NR * c] * d[r + NR * c]; }));
        a.skey[c] = sk, a.cnorm[c] = cn;
        if (b.desc) b.skey[c] = sk, b.cnorm[c] = cn;
    }
}
SCAL_KERNEL(256, k_sc_finish)

// copies one record (saveScancontextAndKeys path: the staged descriptor becomes a database slot)
__device__ __forceinline__ void k_sc_store_body(const SCSlot& from, const SCSlot& to) {
    for (int i = threadIdx.x; i < DESC; i += blockDim.x) to.desc[i] = from.desc[i];
    if (threadIdx.x < NR) to.rkey[threadIdx.x] = from.rkey[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + NS) to.skey[threadIdx.x - 64] = from.skey[threadIdx.x - 64], to.cnorm[threadIdx.x - 64] = from.cnorm[threadIdx.x - 64];
}
SCAL_KERNEL(256, k_sc_store)

// keys / norms of a descriptor supplied by the caller (saveScancontextAndKeys, :236-246)
__device__ __forceinline__ void k_sc_keys_body(const double* __restrict__ desc, float* __restrict__ rkey, double* __restrict__ skey,
                                                 double* __restrict__ cnorm) {
    desc += static_cast<size_t>(blockIdx.x) * DESC, rkey += blockIdx.x * NR, skey += blockIdx.x * NS, cnorm += blockIdx.x * NS;  // one descriptor per block
    if (threadIdx.x < NR) {
        const int r = threadIdx.x;
        rkey[r] = static_cast<float>(eigen_sum4(NS, [&](int c) { return desc[r + NR * c]; }) / NS);
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + NS) {
        const int c = threadIdx.x - 64;
        skey[c] = eigen_sum4(NR, [&](int r) { return desc[r + NR * c]; }) / NR;
        cnorm[c] = sqrt(eigen_sum4(NR, [&](int r) { return desc[r + NR * c] * desc[r + NR * c]; }));
    }
}
SCAL_KERNEL(128, k_sc_keys)

// batch of descriptors (one per block): keys as in k_sc_keys, stored straight into the database slot slots[b] (< 0: not owned)
struct SCSlotList {
    int slot[64];
};
__device__ __forceinline__ void k_sc_store_batch_body(const double* __restrict__ descs, const SCSlotList& sl, SCSlot db /* slot 0 */) {
    const int s = sl.slot[blockIdx.x];
    if (s < 0) return;
    const double* desc = descs + static_cast<size_t>(blockIdx.x) * DESC;
    double* od = db.desc + static_cast<size_t>(s) * DESC;
    for (int i = threadIdx.x; i < DESC; i += blockDim.x) od[i] = desc[i];
    if (threadIdx.x < NR) {
        const int r = threadIdx.x;
        db.rkey[static_cast<size_t>(s) * NR + r] = static_cast<float>(eigen_sum4(NS, [&](int c) { return desc[r + NR * c]; }) / NS);
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + NS) {
        const int c = threadIdx.x - 64;
        db.skey[static_cast<size_t>(s) * NS + c] = eigen_sum4(NR, [&](int r) { return desc[r + NR * c]; }) / NR;
        db.cnorm[static_cast<size_t>(s) * NS + c] = sqrt(eigen_sum4(NR, [&](int r) { return desc[r + NR * c] * desc[r + NR * c]; }));
    }
}
SCAL_KERNEL(128, k_sc_store_batch)

// nanoflann L2_Adaptor<float>::evalMetric (nanoflann.hpp:383-408): five groups of four, f32
__device__ __forceinline__ float key_dist(const float* __restrict__ a, const float* __restrict__ b) {
    float result = 0.f;
#pragma unroll
    for (int g = 0; g < NR; g += 4) {
        const float d0 = a[g] - b[g], d1 = a[g + 1] - b[g + 1], d2 = a[g + 2] - b[g + 2], d3 = a[g + 3] - b[g + 3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    return result;
}

// block-wide three smallest of one u64 key per thread (key = dist bits << 32 | index); smem: 16 u64
__device__ __forceinline__ void block_top3(unsigned long long mine, unsigned long long* out3, unsigned long long* smem) {
    const int nw = blockDim.x >> 6;
    for (int round = 0; round < 3; ++round) {
        unsigned long long m = wave_min_u64(mine);
        if (lane_id() == 0) smem[wave_id()] = m;
        __syncthreads();
        unsigned long long best = smem[0];
        for (int q = 1; q < nw; ++q) best = smem[q] < best ? smem[q] : best;
        __syncthreads();
        out3[round] = best;
        if (mine == best) mine = ~0ull;
    }
}

// local slot s holds global keyframe index s * n_shards + shard
__device__ __forceinline__ void k_sc_topk_body(const float* __restrict__ rkey, const float* __restrict__ query, int n_local, int n_shards,
                                                 int shard, int global_limit, unsigned long long* __restrict__ block_best,
                                                 const int* __restrict__ limits = nullptr) {
    __shared__ unsigned long long smem[16];
    __shared__ float q[NR];
    // blockIdx.y = query of a batch (its own key, limit and result rows)
    query += blockIdx.y * NR, block_best += static_cast<size_t>(blockIdx.y) * gridDim.x * 3;
    if (limits) global_limit = limits[blockIdx.y];
    if (threadIdx.x < NR) q[threadIdx.x] = query[threadIdx.x];
    __syncthreads();
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long mine = ~0ull;
    if (s < n_local) {
        const int g = s * n_shards + shard;
        if (g < global_limit) {
            const float d = key_dist(q, rkey + (size_t)s * NR);
            if (d == d) mine = (static_cast<unsigned long long>(__float_as_uint(d)) << 32) | static_cast<unsigned>(g);
        }
    }
    unsigned long long best[3];
    block_top3(mine, best, smem);
    if (threadIdx.x == 0) {
        block_best[blockIdx.x * 3 + 0] = best[0];
        block_best[blockIdx.x * 3 + 1] = best[1];
        block_best[blockIdx.x * 3 + 2] = best[2];
    }
}
SCAL_KERNEL(256, k_sc_topk)

// distanceBtnScanContext (:116-148) for one pair, executed by ONE wave.  scratch: 7*60 doubles of LDS per wave.
__device__ __forceinline__ void wave_pair_distance(const double* __restrict__ d1, const double* __restrict__ n1, const double* __restrict__ v1,
                                                   const double* __restrict__ d2, const double* __restrict__ n2, const double* __restrict__ v2,
                                                   double* scratch, double* out_dist, int* out_shift) {
    const int lane = lane_id();
    // fastAlignUsingVkey (:93-113): Frobenius norm of vkey1 - circshift(vkey2, s); first strict minimum
    double nrm = 1e300;
    if (lane < NS) {
        const int s = lane;
        nrm = sqrt(eigen_sum4(NS, [&](int j) {
            const double df = v1[j] - v2[(j - s + NS) % NS];
            return df * df;
        }));
        if (!(nrm == nrm)) nrm = 1e300;
    }
    double best = nrm;
    int bs = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int os = __shfl_xor(bs, o, 64);
        if (ob < best || (ob == best && os < bs)) best = ob, bs = os;
    }
    const int argmin_vkey_shift = (best < 10000000) ? bs : 0;
    // SEARCH_RADIUS = round(0.5 * 0.1 * 60) = 3 (:123): shifts a, a+-1, a+-2, a+-3 mod 60, sorted ascending (:124-130)
    int sh[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) sh[k] = (argmin_vkey_shift + (k - 3) + NS) % NS;
#pragma unroll
    for (int a = 1; a < 7; ++a) {  // insertion sort of 7 ints
        const int v = sh[a];
        int b = a - 1;
        while (b >= 0 && sh[b] > v) {
            sh[b + 1] = sh[b];
            --b;
        }
        sh[b + 1] = v;
    }
    // distDirectSC (:69-90) on (sc1, circshift(sc2, s)): shifted.col(j) = sc2.col((j - s) mod 60)
    if (lane < NS) {
        const int col = lane;
        const double na = n1[col];
        for (int k = 0; k < 7; ++k) {
            const int c2 = (col - sh[k] + NS) % NS;
            const double nb = n2[c2];
            double sim = NAN;  // NaN marks "skipped"
            if (!((na == 0) | (nb == 0))) {
                const double* a = d1 + NR * col;
                const double* b = d2 + NR * c2;
                sim = eigen_sum4(NR, [&](int i) { return a[i] * b[i]; }) / (na * nb);
                if (!(sim == sim)) sim = INFINITY;  // keep a genuine NaN distinguishable from "skipped": poisons the sum below
            }
            scratch[k * NS + col] = sim;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    double dist = 1e300;
    if (lane < 7) {
        double sum = 0;
        int eff = 0;
        for (int c = 0; c < NS; ++c) {
            const double s = scratch[lane * NS + c];
            if (s == s) {
                sum = sum + (isinf(s) ? NAN : s);
                eff = eff + 1;
            }
        }
        dist = 1.0 - sum / eff;  // eff == 0 -> NaN, never selected (:139)
    }
    // first strict minimum in ascending shift order, starting from 10000000 (:133-144)
    double bd = 10000000;
    int bshift = 0;
    for (int k = 0; k < 7; ++k) {
        const double dk = __shfl(dist, k, 64);
        if (dk < bd) bd = dk, bshift = sh[k];
    }
    *out_dist = bd;
    *out_shift = bshift;
    __builtin_amdgcn_wave_barrier();
}

struct SCDb {
    const double* desc;
    const float* rkey;
    const double* skey;
    const double* cnorm;
};

// one block of 256 threads: merge per-block top-3, then waves 0..2 evaluate the three candidates (:385-400)
__device__ __forceinline__ void k_sc_detect_body(const unsigned long long* __restrict__ block_best, int n_blocks, const SCDb& db, int n_shards,
                                                   int shard, const double* __restrict__ qdesc, const double* __restrict__ qskey,
                                                   const double* __restrict__ qnorm, int fill_missing_with_zero, SCRec* __restrict__ out) {
    __shared__ unsigned long long smem[16];
    __shared__ unsigned long long top[3];
    // blockIdx.y = query of a batch
    block_best += static_cast<size_t>(blockIdx.y) * n_blocks * 3, qdesc += static_cast<size_t>(blockIdx.y) * DESC;
    qskey += blockIdx.y * NS, qnorm += blockIdx.y * NS, out += blockIdx.y * 3;
    __shared__ double scratch[3][7 * NS];
    unsigned long long mine = ~0ull;
    unsigned long long second = ~0ull, third = ~0ull;
    // each thread scans a strided share of the candidates and keeps its own three smallest
    for (int i = threadIdx.x; i < n_blocks * 3; i += blockDim.x) {
        unsigned long long v = block_best[i];
        if (v < mine) {
            third = second, second = mine, mine = v;
        } else if (v < second) {
            third = second, second = v;
        } else if (v < third) {
            third = v;
        }
    }
    // three rounds of block argmin over each thread's current head
    for (int round = 0; round < 3; ++round) {
        unsigned long long m = wave_min_u64(mine);
        if (lane_id() == 0) smem[wave_id()] = m;
        __syncthreads();
        unsigned long long best = smem[0];
        for (int q = 1; q < 4; ++q) best = smem[q] < best ? smem[q] : best;
        if (threadIdx.x == 0) top[round] = best;
        __syncthreads();
        if (mine == best && best != ~0ull) mine = second, second = third, third = ~0ull;
    }
    __syncthreads();
    const int w = wave_id();
    if (w < 3) {
        const unsigned long long t = top[w];
        SCRec r;
        r.key_dist = FLT_MAX, r.idx = -1, r.sc_dist = 10000000, r.shift = 0, r.pad = 0;
        int g = -1;
        if (t != ~0ull) {
            g = static_cast<int>(t & 0xffffffffu);
            r.key_dist = __uint_as_float(static_cast<unsigned>(t >> 32));
        } else if (fill_missing_with_zero && shard == 0) {
            g = 0;  // unused KNN slots keep the caller's zero-initialised index (Scancontext.cpp:372, nanoflann.hpp:193-198)
        }
        if (g >= 0) {
            const int s = g / n_shards;
            double dist;
            int shift;
            wave_pair_distance(qdesc, qnorm, qskey, db.desc + (size_t)s * DESC, db.cnorm + (size_t)s * NS, db.skey + (size_t)s * NS, scratch[w],
                               &dist, &shift);
            r.idx = g, r.sc_dist = dist, r.shift = shift;
        }
        if (lane_id() == 0) out[w] = r;
    }
}
SCAL_KERNEL(256, k_sc_detect)

__device__ __forceinline__ void k_sc_pairs_body(const SCDb& db, const int* __restrict__ ia, const int* __restrict__ ib, int n_pairs,
                                                  double* __restrict__ dist, int* __restrict__ shift) {
    __shared__ double scratch[4][7 * NS];
    const int w = wave_id();
    const int p = blockIdx.x * 4 + w;
    if (p >= n_pairs) return;
    const int a = ia[p], b = ib[p];
    double d;
    int s;
    wave_pair_distance(db.desc + (size_t)a * DESC, db.cnorm + (size_t)a * NS, db.skey + (size_t)a * NS, db.desc + (size_t)b * DESC,
                       db.cnorm + (size_t)b * NS, db.skey + (size_t)b * NS, scratch[w], &d, &s);
    if (lane_id() == 0) dist[p] = d, shift[p] = s;
}
SCAL_KERNEL(256, k_sc_pairs)

// dense block of the pair grid, one wave per (query, database) pair.
// mode 0: the reference's 7-shift search; mode 1: exhaustive over all 60 shifts (first strict minimum).
__device__ __forceinline__ void k_sc_matrix_body(const SCDb& db, int q0, int nq, int d0, int nd, int mode, double* __restrict__ dist,
                                                   int* __restrict__ shift) {
    __shared__ double scratch[4][7 * NS];
    const int w = wave_id(), lane = lane_id();
    const long long p = static_cast<long long>(blockIdx.x) * 4 + w;
    if (p >= static_cast<long long>(nq) * nd) return;
    const int a = q0 + static_cast<int>(p / nd), b = d0 + static_cast<int>(p % nd);
    const double* d1 = db.desc + (size_t)a * DESC;
    const double* d2 = db.desc + (size_t)b * DESC;
    const double* n1 = db.cnorm + (size_t)a * NS;
    const double* n2 = db.cnorm + (size_t)b * NS;
    double bd;
    int bs;
    if (mode == 0) {
        wave_pair_distance(d1, n1, db.skey + (size_t)a * NS, d2, n2, db.skey + (size_t)b * NS, scratch[w], &bd, &bs);
    } else {
        // lane = shift: sequential sum over columns exactly as distDirectSC does
        double dd = 1e300;
        if (lane < NS) {
            const int s = lane;
            double sum = 0;
            int eff = 0;
            for (int col = 0; col < NS; ++col) {
                const int c2 = (col - s + NS) % NS;
                const double na = n1[col], nb = n2[c2];
                if ((na == 0) | (nb == 0)) continue;
                const double* x = d1 + NR * col;
                const double* y = d2 + NR * c2;
                sum = sum + eigen_sum4(NR, [&](int i) { return x[i] * y[i]; }) / (na * nb);
                eff = eff + 1;
            }
            dd = 1.0 - sum / eff;
            if (!(dd == dd)) dd = 1e300;
        }
        bd = dd, bs = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(bd, o, 64);
            const int os = __shfl_xor(bs, o, 64);
            if (ob < bd || (ob == bd && os < bs)) bd = ob, bs = os;
        }
        if (!(bd < 10000000)) bd = 10000000, bs = 0;
    }
    if (lane == 0) dist[p] = bd, shift[p] = bs;
}
SCAL_KERNEL(256, k_sc_matrix)


// ---- dense matrix on the matrix cores (mode 2) --------------------------------------------------------------------------------
// distDirectSC (Scancontext.cpp:83-110) summed over a column shift s is  sum_col <A[:,col], B[:,col-s]> / (|A col| |B col-s|):
// with unit columns Ahat, Bhat this is a product  C[q][s] = sum_k Ahat_q[k] * X_b[k][s],  k = (ring, col), K = 1200, where
// X_b[(r,col)][s] = Bhat_b[r][(col - s) mod 60] is the circulant expansion of one database descriptor.  16 queries x 64 shifts
// (60 used) x K = 1200 per database entry is 300 k-steps of v_mfma_f64_16x16x4_f64 per 16-shift tile; the expansion is never
// materialised - each lane reads its B operand from an LDS copy of Bhat_b whose rows are extended to 124 entries so that
// (col - s) needs no modulo.  The effective-column count of the reference (columns where either norm is zero are skipped and
// not counted) is popcount(maskA & rotl60(maskB, s)).
constexpr int GQ = 64;      // queries per workgroup, 16 per wave
constexpr int GROW = 124;   // extended row: x = c + 64 for c in [-64, 59]
constexpr int GKS = DESC / 4;
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

// f64: v_mfma_f64_16x16x4_f64, 64 cycles, C row = (lane >> 4) + 4 reg; 4 database entries per workgroup.
// f32 (mode 3): v_mfma_f32_16x16x4_f32, 32 cycles, C row = 4 (lane >> 4) + reg; 8 entries per workgroup (same LDS bytes,
// same 128 accumulator registers, twice the pairs per query-operand load).
template <typename T>
struct GramT;
template <>
struct GramT<double> {
    typedef d4_t acc_t;
    typedef double2 vec_t;
    static constexpr int GE = 4, VEC = 2;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct GramT<float> {
    typedef f4_t acc_t;
    typedef float4 vec_t;
    static constexpr int GE = 8, VEC = 4;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// operand A in fragment order: frag[tile][ks][lane] = Ahat_q[r][col], q = 16 tile + (lane & 15), k = 4 ks + (lane >> 4) = 60 r + col
template <typename T>
__global__ void __launch_bounds__(256) k_sc_gram_prep(SCDb db, int q0, int nq, T* __restrict__ frag, unsigned long long* __restrict__ qmask) {
    const int tile = blockIdx.x;
    T* out = frag + static_cast<size_t>(tile) * GKS * 64;
    for (int i = threadIdx.x; i < GKS * 64; i += blockDim.x) {
        const int ks = i >> 6, lane = i & 63;
        const int q = tile * 16 + (lane & 15), k = 4 * ks + (lane >> 4);
        const int r = k / NS, col = k - r * NS;
        double v = 0;
        if (q < nq) {
            const double n = db.cnorm[static_cast<size_t>(q0 + q) * NS + col];
            if (n != 0) v = db.desc[static_cast<size_t>(q0 + q) * DESC + col * NR + r] / n;
        }
        out[i] = static_cast<T>(v);
    }
    // column masks of this tile's 16 queries: wave w takes queries w, w+4, ...
    for (int j = wave_id(); j < 16; j += 4) {
        const int q = tile * 16 + j;
        const int lane = lane_id();
        const bool nz = q < nq && lane < NS && db.cnorm[static_cast<size_t>(q0 + q) * NS + lane] != 0;
        const unsigned long long m = __ballot(nz);
        if (lane == 0) qmask[q] = m;
    }
}

// database operand: unit columns, ring-major (bhat[b][60 r + c]), and the 60-bit column masks; k_sc_gram's workgroups then fill
// their LDS rows with plain 16-byte copies instead of each re-deriving the quotients
template <typename T>
__global__ void __launch_bounds__(256) k_sc_gram_prep_db(SCDb db, int d0, int nd, T* __restrict__ bhat, unsigned long long* __restrict__ dmask) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < DESC; i += blockDim.x) {
        const int c = i / NR, r = i - c * NR;
        const double n = db.cnorm[static_cast<size_t>(d0 + b) * NS + c];
        bhat[static_cast<size_t>(b) * DESC + r * NS + c] = static_cast<T>(n != 0 ? db.desc[static_cast<size_t>(d0 + b) * DESC + i] / n : 0.0);
    }
    if (wave_id() == 0) {
        const int lane = lane_id();
        const bool nz = lane < NS && db.cnorm[static_cast<size_t>(d0 + b) * NS + lane] != 0;
        const unsigned long long m = __ballot(nz);
        if (lane == 0) dmask[b] = m;
    }
}

template <typename T>
__global__ void __launch_bounds__(256, 2) k_sc_gram(const T* __restrict__ bhat, const unsigned long long* __restrict__ dmask,
                                                    const T* __restrict__ frag, const unsigned long long* __restrict__ qmask, int nq, int nd,
                                                    double* __restrict__ dist, int* __restrict__ shift) {
    typedef GramT<T> G;
    typedef typename G::acc_t acc_t;
    typedef typename G::vec_t vec_t;
    constexpr int GE = G::GE, VEC = G::VEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char ext_raw[];
    T* ext = reinterpret_cast<T*>(ext_raw);  // [GE][NR][GROW]
    __shared__ unsigned long long bmask[GE];
    __shared__ double inv_eff[NS + 1];  // f32 mode only: 1 / eff, so its 128 quotients per lane are products
    const int chunk = blockIdx.x, lane = lane_id(), w = wave_id();
    if (sizeof(T) == 4 && threadIdx.x <= NS) inv_eff[threadIdx.x] = threadIdx.x ? 1.0 / threadIdx.x : 0.0;
    // extended rows, 16 bytes per copy: x a multiple of VEC -> c = (x - 64) mod 60 is one too, so a copy never straddles the wrap
    for (int i = threadIdx.x; i < GE * NR * GROW / VEC; i += blockDim.x) {
        const int e = i / (NR * GROW / VEC), rem = i - e * (NR * GROW / VEC);
        const int r = rem / (GROW / VEC), x = VEC * (rem - r * (GROW / VEC));
        const int b = chunk * GE + e;
        const int c = (x + 56) % NS;
        vec_t v = {};
        if (b < nd) v = *reinterpret_cast<const vec_t*>(bhat + static_cast<size_t>(b) * DESC + r * NS + c);
        *reinterpret_cast<vec_t*>(ext + e * (NR * GROW) + r * GROW + x) = v;
    }
    if (threadIdx.x < GE) {
        const int b = chunk * GE + threadIdx.x;
        bmask[threadIdx.x] = b < nd ? dmask[b] : 0ull;
    }
    __syncthreads();
    const int tile = blockIdx.y * 4 + w;
    const T* fa = frag + static_cast<size_t>(tile) * GKS * 64 + lane;
    // The (entry, shift) pairs of the GE entries form ONE axis of GE x 60 columns, cut into 16-column MFMA tiles: 15 tiles for four
    // entries, 30 for eight - no padded shifts (rounds 1-2 gave every entry four tiles of its own, 64 columns for 60 shifts: 6 % of the
    // MFMAs computed nothing).  Column c = 16 t + n (n = lane & 15) belongs to entry e = c / 60 at shift s = c - 60 e; its operand
    // sits at ext[e][r][col - s + 64] = base(lane) + e * (entry stride + 60) - 16 t + r * GROW + col.  For most tiles e is the same
    // for all lanes (the offset is an immediate); the tile that holds an entry's last columns and the next entry's first ones - c
    // crosses 60 (e + 1) at n = thr_e = 60 (e + 1) mod 16 - adds a per-lane adjustment kept in one register per entry.
    constexpr int NT = GE * NS / 16;
    constexpr int ESTR = NR * GROW + NS;
    static_assert(GE * NS % 16 == 0, "the packed shift axis must be whole tiles");
    const int n16 = lane & 15;
    const T* bp = ext + ((lane >> 4) - n16 + 64 - 48);  // - 48: the offsets below stay non-negative
    int xadj[GE];
#pragma unroll
    for (int e = 0; e < GE; ++e) xadj[e] = n16 >= ((NS * (e + 1)) & 15) ? ESTR : 0;
    auto toff = [&](int t) {  // element offset of tile t's operand from bp (+ r * GROW + 4 j): compile-time part + per-lane part
        const int e_lo = (16 * t) / NS;
        const int thr = NS * (e_lo + 1) - 16 * t;  // lanes n >= thr belong to the next entry (only if thr < 16)
        return e_lo * ESTR - 16 * t + 48 + (thr < 16 ? xadj[e_lo] : 0);
    };
    acc_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = acc_t{0, 0, 0, 0};
    // one ring (15 k-steps) of the query operand lives in registers; each value is reloaded for the next ring right after its
    // MFMAs have been issued, so the load has the other 14 k-steps to land
    T a[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) a[j] = fa[j * 64];
    // the circulant operands of a k-step are read from LDS one k-step ahead (bn) and the issue order is pinned to one
    // ds_read per MFMA, so no MFMA waits on the read issued just in front of it
    T bc[NT], bn[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bc[t] = bp[toff(t)];
    for (int r = 0; r < NR; ++r) {
        const T* br = bp + r * GROW;
        const T* fn = fa + (r + 1 < NR ? r + 1 : r) * 15 * 64;
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            // next k-step: j + 1 of this ring, or the first of the next ring (the last ring re-reads its own: unused)
            const T* bx = j + 1 < 15 ? br + 4 * (j + 1) : (r + 1 < NR ? br + GROW : br);
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[t] = bx[toff(t)];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = G::mma(a[j], bc[t], acc[t]);
            a[j] = fn[j * 64];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) bc[i] = bn[i];
        }
    }
    // acc[t][reg]: query = 16 tile + row(lane, reg), column c = 16 t + n16 -> (entry, shift)
    const unsigned long long m60 = (1ull << NS) - 1;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int q = tile * 16 + G::row(lane, reg);
        const unsigned long long ma = qmask[q];
        double bd[GE];
        int bs[GE];
#pragma unroll
        for (int e = 0; e < GE; ++e) bd[e] = 1e300, bs[e] = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int e_lo = (16 * t) / NS;
            const int thr = NS * (e_lo + 1) - 16 * t;
            const bool next = thr < 16 && n16 >= thr;   // this lane's column of tile t belongs to entry e_lo + 1
            const int s = 16 * t + n16 - NS * e_lo - (next ? NS : 0);
            const unsigned long long mb = (thr < 16 && e_lo + 1 < GE) ? (next ? bmask[e_lo + 1] : bmask[e_lo]) : bmask[e_lo];
            const unsigned long long rot = ((mb << s) | (mb >> (NS - s))) & m60;
            const int eff = __popcll(ma & rot);
            double dd;
            if (sizeof(T) == 4)
                dd = 1.0 - static_cast<double>(acc[t][reg]) * inv_eff[eff];
            else
                dd = 1.0 - acc[t][reg] / eff;
            if (eff == 0 || !(dd == dd)) dd = 1e300;
            // ascending t = ascending shift inside an entry, so "first strict minimum" keeps the smallest shift
            if (!next && dd < bd[e_lo]) bd[e_lo] = dd, bs[e_lo] = s;
            if (thr < 16 && e_lo + 1 < GE) {
                if (next && dd < bd[e_lo + 1]) bd[e_lo + 1] = dd, bs[e_lo + 1] = s;
            }
        }
#pragma unroll
        for (int e = 0; e < GE; ++e) {
            double b1 = bd[e];
            int s1 = bs[e];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const double ob = __shfl_xor(b1, o, 64);
                const int os = __shfl_xor(s1, o, 64);
                if (ob < b1 || (ob == b1 && os < s1)) b1 = ob, s1 = os;
            }
            if (!(b1 < 10000000)) b1 = 10000000, s1 = 0;
            const int b = chunk * GE + e;
            if (n16 == 0 && q < nq && b < nd) {
                dist[static_cast<size_t>(q) * nd + b] = b1;
                shift[static_cast<size_t>(q) * nd + b] = s1;
            }
        }
    }
}

}  // namespace scal

using namespace scal;

struct scal_sc {
    scal_sc_config cfg;
    hipStream_t stream = nullptr;
    HostStage hs;  // pinned staging of scal_sc_insert_cloud's host array (first call)
    // keyframe filter of the *_features entry points: on its own lane in the stage-pipelined mode (the filter is three quarters of a
    // keyframe's device time, descriptor + search are short), on `stream` otherwise.  One set of filter outputs: ev_ds = filter
    // done (the main stream waits for it), ev_tail = descriptor built from them (the next filter waits for it).
    hipStream_t fstream = nullptr;
    int flane = -1;
    hipEvent_t ev_ds = nullptr, ev_tail = nullptr;
    bool tail_recorded = false;
    std::mutex mu;  // insert and detect come from two threads in the reference with no common lock
    int cap = 0;
    int n_global = 0;  // keyframes inserted (global count)
    int n_local = 0;   // slots used on this shard
    int tree_making_period_conter = 0;
    int size_at_rebuild = 0;
    DevBuf<double> desc, skey, cnorm;
    DevBuf<float> rkey;
    // query staging (newest keyframe may live on another shard)
    DevBuf<double> qdesc, qskey, qnorm;
    DevBuf<float> qrkey;
    DevBuf<float> pts;
    int pts_cap = 0;
    DevBuf<unsigned> gcell;  // global polar image of k_sc_bin, zero between scans
    DevBuf<unsigned long long> block_best;
    DevBuf<SCRec> d_rec;
    PinBuf<SCRec> h_rec;
    DevBuf<int> d_pairs;
    DevBuf<double> d_dist;
    DevBuf<int> d_shift;
    DevBuf<double> g_frag;  // dense matrix, mode 2: query operand in MFMA fragment order
    DevBuf<unsigned long long> g_qmask;
    DevBuf<double> g_bhat;  // database operand: unit columns, ring-major
    DevBuf<float> g_frag32, g_bhat32;  // mode 3 (f32 MFMA)
    DevBuf<unsigned long long> g_dmask;
    size_t pair_cap = 0;
    // keyframe downsampling for scal_sc_insert_features (lazy)
    VoxelFilter vf;
    DevBuf<float> dsx, dsy, dsz, dsw;
    DevBuf<int> d_nds;
    int vf_cap = 0;
    hipEvent_t ev = nullptr;
    int lane = 0;
    // batched shard queries
    DevBuf<float> bq_rkey;
    DevBuf<double> bq_skey, bq_norm;
    DevBuf<unsigned long long> bq_best;
    DevBuf<int> bq_limits;
    PinBuf<int> bq_hlimits;
    int bq_cap = 0, bq_nb = 0;
    // detections enqueued and not collected: 1 = search launched, 2 = database too small (nothing launched); up to DET_DEPTH in flight,
    // each with its own record slot and event, so collecting one does not wait for anything queued on the stream behind it
    static constexpr int DET_DEPTH = 4;
    int det_mode[DET_DEPTH] = {};
    hipEvent_t det_ev[DET_DEPTH] = {};
    int det_head = 0, det_count = 0;
    hipEvent_t made_ev[4] = {};  // descriptors queued by scal_sc_make_features_enqueue, oldest first
    unsigned made_head = 0, made_tail = 0;
    SCDb db() const { return SCDb{desc.p, rkey.p, skey.p, cnorm.p}; }
    SCSlot staging() const { return SCSlot{qdesc.p, qrkey.p, qskey.p, qnorm.p}; }
    SCSlot slot(size_t sl) const { return SCSlot{desc.p + sl * DESC, rkey.p + sl * NR, skey.p + sl * NS, cnorm.p + sl * NS}; }
    bool owns(int g) const { return cfg.n_shards <= 1 || (g % cfg.n_shards) == cfg.shard; }
};

extern "C" int scal_sc_create(const scal_sc_config* cfg, scal_sc_t** out) {
    if (!cfg || !out || cfg->max_keyframes <= 0 || !(cfg->max_radius > 0)) {
        set_error("scal_sc_create: bad argument");
        return SCAL_E_ARG;
    }
    *out = nullptr;
    SCAL_TRY(select_device(cfg->device));
    auto* c = new scal_sc();
    c->cfg = *cfg;
    if (c->cfg.n_shards < 1) c->cfg.n_shards = 1, c->cfg.shard = 0;
    if (c->cfg.shard < 0 || c->cfg.shard >= c->cfg.n_shards) {
        delete c;
        set_error("shard %d out of range for %d shards", cfg->shard, cfg->n_shards);
        return SCAL_E_ARG;
    }
    c->cap = cfg->max_keyframes;
    int rc = SCAL_OK;
    auto A = [&](int r) { if (rc == SCAL_OK) rc = r; };
    A(c->desc.alloc((size_t)c->cap * DESC));
    A(c->skey.alloc((size_t)c->cap * NS));
    A(c->cnorm.alloc((size_t)c->cap * NS));
    A(c->rkey.alloc((size_t)c->cap * NR));
    A(c->qdesc.alloc(DESC)); A(c->qskey.alloc(NS)); A(c->qnorm.alloc(NS)); A(c->qrkey.alloc(NR));
    A(c->block_best.alloc((size_t)3 * div_up(c->cap, 256) + 3));
    A(c->d_rec.alloc(4 * scal_sc::DET_DEPTH));
    A(c->gcell.alloc(DESC));
    A(c->h_rec.alloc(4 * scal_sc::DET_DEPTH));
    if (rc == SCAL_OK && acquire_stream(c->cfg.device, &c->stream, c->lane = (c->cfg.side_stream > 0 ? std::min(c->cfg.side_stream, 5) : stage_lane(STAGE_SC))) != SCAL_OK) {
        set_error("hipStreamCreate failed");
        rc = SCAL_E_HIP;
    }
    if (rc == SCAL_OK && c->cfg.side_stream <= 0 && stage_lane(STAGE_SC_FILTER) != c->lane) {
        if (acquire_stream(c->cfg.device, &c->fstream, c->flane = stage_lane(STAGE_SC_FILTER)) != SCAL_OK ||
            hipEventCreateWithFlags(&c->ev_ds, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess) {
            set_error("hipStreamCreate failed");
            rc = SCAL_E_HIP;
        }
    }
    // initialised on the context's own stream (the legacy null stream is not ordered against it)
    if (rc == SCAL_OK && (op_memset_async(c->gcell.p, 0, sizeof(unsigned) * DESC, c->stream) != hipSuccess || op_stream_synchronize(c->stream) != hipSuccess))
        rc = SCAL_E_HIP;
    if (rc != SCAL_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return SCAL_OK;
}

extern "C" void scal_sc_destroy(scal_sc_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->fstream) {
        (void)op_stream_synchronize(c->fstream);
        release_stream(c->cfg.device, c->flane);
    }
    if (c->stream) {
        (void)op_stream_synchronize(c->stream);
        release_stream(c->cfg.device, c->lane);
    }
    if (c->ev_ds) (void)hipEventDestroy(c->ev_ds);
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    if (c->ev) (void)hipEventDestroy(c->ev);
    for (int k = 0; k < scal_sc::DET_DEPTH; ++k)
        if (c->det_ev[k]) (void)hipEventDestroy(c->det_ev[k]);
    delete c;
}

extern "C" int scal_sc_size(scal_sc_t* c) {
    if (!c) return SCAL_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return c->n_global;
}

// the staged descriptor/keys become keyframe n_global; stored only when this shard owns that index
static int commit_staged(scal_sc* c) {
    hipStream_t s = c->stream;
    const int g = c->n_global;
    if (c->owns(g)) {
        if (c->n_local >= c->cap) {
            set_error("ScanContext database full (%d keyframes on this shard)", c->cap);
            return SCAL_E_CAPACITY;
        }
        SCAL_LAUNCH("k_sc_store", k_sc_store, dim3(1), dim3(256), 0, s, c->staging(), c->slot(c->n_local));
        SCAL_HIP(hipGetLastError());
        c->n_local++;
    }
    c->n_global++;
    return SCAL_OK;
}

// build the descriptor of a cloud into the query staging slot; when this shard owns the new global index, also
// into its database slot (same kernel)
static int make_into(scal_sc* c, const float* px, const float* py, const float* pz, int stride, const int* d_n, int n_host, bool insert) {
    hipStream_t s = c->stream;
    SCSlot db{nullptr, nullptr, nullptr, nullptr};
    const bool store = insert && c->owns(c->n_global);
    if (store) {
        if (c->n_local >= c->cap) {
            set_error("ScanContext database full (%d keyframes on this shard)", c->cap);
            return SCAL_E_CAPACITY;
        }
        db = c->slot(c->n_local);
    }
    const int nblk = std::max(1, std::min(64, div_up(n_host, 1024)));
    SCAL_LAUNCH("k_sc_bin", k_sc_bin, dim3(nblk), dim3(256), 0, s, px, py, pz, stride, d_n, n_host, c->cfg.max_radius, c->cfg.float_math, c->gcell.p);
    SCAL_LAUNCH("k_sc_finish", k_sc_finish, dim3(1), dim3(256), 0, s, c->gcell.p, c->staging(), db);
    SCAL_HIP(hipGetLastError());
    if (store) c->n_local++;
    if (insert) c->n_global++;
    return SCAL_OK;
}
static int upload_points(scal_sc* c, const float* xyzi, int n) {
    if (n > c->pts_cap) {
        const int nc = std::max(n, 65536);
        SCAL_TRY(c->pts.alloc((size_t)nc * 4));
        c->pts_cap = nc;
    }
    if (n > 0) {
        // through pinned staging (four clouds' worth; when it is full the stream is synchronised once and the area starts over)
        const size_t bytes = sizeof(float) * 4 * n;
        (void)c->hs.reserve(std::max<size_t>(4 * bytes, (size_t)1 << 22));
        if (c->hs.pin.p && c->hs.used + bytes + 256 > c->hs.pin.n && bytes + 256 <= c->hs.pin.n) {
            SCAL_HIP(op_stream_synchronize(c->stream));
            c->hs.finish();
        }
        SCAL_HIP(c->hs.h2d(c->pts.p, xyzi, bytes, c->stream));
    }
    return SCAL_OK;
}

extern "C" int scal_sc_insert_cloud(scal_sc_t* c, const float* xyzi, int n) {
    if (!c || n < 0 || (n > 0 && !xyzi)) {
        set_error("scal_sc_insert_cloud: bad argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_TRY(upload_points(c, xyzi, n));
    SCAL_TRY(make_into(c, c->pts.p, nullptr, nullptr, 4, nullptr, n, true));
    SCAL_HIP(op_stream_synchronize(c->stream));
    return SCAL_OK;
}

extern "C" int scal_sc_insert_cloud_device(scal_sc_t* c, const float* d_x, const float* d_y, const float* d_z, const int* d_n, int n_max) {
    if (!c || !d_x || !d_y || !d_z || n_max < 0) {
        set_error("scal_sc_insert_cloud_device: bad argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    return make_into(c, d_x, d_y, d_z, 1, d_n, n_max, true);
}

extern "C" int scal_sc_make_descriptor(scal_sc_t* c, const float* xyzi, int n, double* desc) {
    if (!c || !desc || n < 0 || (n > 0 && !xyzi)) {
        set_error("scal_sc_make_descriptor: bad argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_TRY(upload_points(c, xyzi, n));
    SCAL_TRY(make_into(c, c->pts.p, nullptr, nullptr, 4, nullptr, n, false));
    SCAL_HIP(op_memcpy_async(desc, c->qdesc.p, sizeof(double) * DESC, hipMemcpyDeviceToHost, c->stream));
    SCAL_HIP(op_stream_synchronize(c->stream));
    return SCAL_OK;
}

extern "C" int scal_sc_insert_descriptor(scal_sc_t* c, const double* desc) {
    if (!c || !desc) {
        set_error("scal_sc_insert_descriptor: bad argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    SCAL_HIP(op_memcpy_async(c->qdesc.p, desc, sizeof(double) * DESC, hipMemcpyHostToDevice, s));
    SCAL_LAUNCH("k_sc_keys", k_sc_keys, dim3(1), dim3(128), 0, s, c->qdesc.p, c->qrkey.p, c->qskey.p, c->qnorm.p);
    SCAL_TRY(commit_staged(c));
    SCAL_HIP(op_stream_synchronize(s));
    return SCAL_OK;
}

static int ds_features(scal_sc* c, scal_features_t* feat, const int** d_n, int* n_cap) {
    FeatDeviceView v = features_view(feat);
    if (v.device != c->cfg.device) {
        set_error("features context lives on device %d, ScanContext on %d", v.device, c->cfg.device);
        return SCAL_E_ARG;
    }
    if (c->vf_cap < v.cap) {
        SCAL_TRY(c->vf.init(v.cap));
        c->vf.set_tag(".D");
        SCAL_TRY(c->dsx.alloc(v.cap));
        SCAL_TRY(c->dsy.alloc(v.cap));
        SCAL_TRY(c->dsz.alloc(v.cap));
        SCAL_TRY(c->dsw.alloc(v.cap));
        SCAL_TRY(c->d_nds.alloc(2));
        c->vf_cap = v.cap;
    }
    hipStream_t fs = c->fstream ? c->fstream : c->stream;
    SCAL_TRY(features_wait_done(feat, fs));  // start after stage A of this scan
    if (fs != c->stream && c->tail_recorded) SCAL_HIP(op_stream_wait_event(fs, c->ev_tail, 0));  // the previous descriptor has been built from the buffers
    // downSizeFilterScancontext: leaf 0.4 m (laserPosegraphOptimization.cpp:890-891); tightly packed keys, up to 36 bits
    // the bounding box comes with the features context (per-block parts from k_curv): no reset / bounding-box launches here
    SCAL_TRY(c->vf.run(fs, CSoA4{v.x, v.y, v.z, v.i}, &v.P->n_kept, v.cap, 0.4f, 36, SoA4{c->dsx.p, c->dsy.p, c->dsz.p, c->dsw.p}, c->d_nds.p,
                       nullptr, v.box_parts, v.n_box_parts));
    if (v.stream != fs) SCAL_TRY(features_note_reader(feat, fs));  // the filter was the last reader of feat's buffers
    if (fs != c->stream) {
        SCAL_HIP(op_event_record(c->ev_ds, fs));
        SCAL_HIP(op_stream_wait_event(c->stream, c->ev_ds, 0));
    }
    *d_n = c->d_nds.p;
    *n_cap = v.cap;
    return SCAL_OK;
}

// the descriptor has been built from the filter's output buffers: the next filter may overwrite them
static int ds_consumed(scal_sc* c) {
    if (c->fstream && c->fstream != c->stream) {
        SCAL_HIP(op_event_record(c->ev_tail, c->stream));
        c->tail_recorded = true;
    }
    return SCAL_OK;
}

extern "C" int scal_sc_insert_features(scal_sc_t* c, scal_features_t* feat) {
    if (!c || !feat) {
        set_error("scal_sc_insert_features: null argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    const int* d_n;
    int cap;
    SCAL_TRY(ds_features(c, feat, &d_n, &cap));
    SCAL_TRY(make_into(c, c->dsx.p, c->dsy.p, c->dsz.p, 1, d_n, cap, true));
    return ds_consumed(c);
}

static int make_features(scal_sc_t* c, scal_features_t* feat, double* d_desc, bool wait) {
    if (!c || !feat || !d_desc) {
        set_error("scal_sc_make_features: null argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    const int* d_n;
    int cap;
    SCAL_TRY(ds_features(c, feat, &d_n, &cap));
    SCAL_TRY(make_into(c, c->dsx.p, c->dsy.p, c->dsz.p, 1, d_n, cap, false));
    SCAL_TRY(ds_consumed(c));
    SCAL_HIP(op_memcpy_async(d_desc, c->qdesc.p, sizeof(double) * DESC, hipMemcpyDeviceToDevice, c->stream));
    if (wait) {
        SCAL_HIP(op_stream_synchronize(c->stream));
    } else {
        if (c->made_tail - c->made_head >= 4) {
            set_error("scal_sc_make_features_enqueue: four descriptors are queued and none has been waited for");
            return SCAL_E_STATE;
        }
        hipEvent_t& ev = c->made_ev[c->made_tail % 4];
        if (!ev) SCAL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        SCAL_HIP(op_event_record(ev, c->stream));
        c->made_tail++;
    }
    return SCAL_OK;
}
// waits for the OLDEST descriptor queued by scal_sc_make_features_enqueue (not for younger work on the stream)
extern "C" int scal_sc_wait_descriptor(scal_sc_t* c) {
    if (!c) return SCAL_E_ARG;
    std::unique_lock<std::mutex> lk(c->mu);
    if (c->made_head == c->made_tail) {
        set_error("scal_sc_wait_descriptor: nothing queued");
        return SCAL_E_STATE;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipEvent_t ev = c->made_ev[c->made_head % 4];
    lk.unlock();  // further descriptors may be queued from another thread while this one is waited for
    const hipError_t e = op_event_synchronize(ev);
    lk.lock();
    SCAL_HIP(e);
    c->made_head++;
    return SCAL_OK;
}
extern "C" int scal_sc_make_features(scal_sc_t* c, scal_features_t* feat, double* d_desc) { return make_features(c, feat, d_desc, true); }
extern "C" int scal_sc_make_features_enqueue(scal_sc_t* c, scal_features_t* feat, double* d_desc) { return make_features(c, feat, d_desc, false); }

extern "C" int scal_sc_insert_descriptor_device(scal_sc_t* c, const double* d_desc) {
    if (!c || !d_desc) {
        set_error("scal_sc_insert_descriptor_device: null argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    SCAL_HIP(op_memcpy_async(c->qdesc.p, d_desc, sizeof(double) * DESC, hipMemcpyDeviceToDevice, s));
    SCAL_LAUNCH("k_sc_keys", k_sc_keys, dim3(1), dim3(128), 0, s, c->qdesc.p, c->qrkey.p, c->qskey.p, c->qnorm.p);
    SCAL_TRY(commit_staged(c));
    SCAL_HIP(op_stream_synchronize(s));
    return SCAL_OK;
}

extern "C" int scal_sc_shard_query_device(scal_sc_t* c, const double* d_queries, int nq, int global_size_at_rebuild, scal_sc_cand* d_out) {
    if (!c || !d_queries || !d_out || nq < 0) {
        set_error("scal_sc_shard_query_device: bad argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const int nb = std::max(1, div_up(c->n_local, 256));
    for (int q = 0; q < nq; ++q) {
        SCAL_HIP(op_memcpy_async(c->qdesc.p, d_queries + (size_t)q * DESC, sizeof(double) * DESC, hipMemcpyDeviceToDevice, s));
        SCAL_LAUNCH("k_sc_keys", k_sc_keys, dim3(1), dim3(128), 0, s, c->qdesc.p, c->qrkey.p, c->qskey.p, c->qnorm.p);
        SCAL_LAUNCH("k_sc_topk", k_sc_topk, dim3(nb), dim3(256), 0, s, c->rkey.p, c->qrkey.p, c->n_local, c->cfg.n_shards, c->cfg.shard,
                           global_size_at_rebuild - 30, c->block_best.p, static_cast<const int*>(nullptr));
        SCAL_LAUNCH("k_sc_detect", k_sc_detect, dim3(1), dim3(256), 0, s, c->block_best.p, nb, c->db(), c->cfg.n_shards, c->cfg.shard, c->qdesc.p, c->qskey.p,
                           c->qnorm.p, 1, reinterpret_cast<SCRec*>(d_out) + 3 * (size_t)q);
    }
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(op_stream_synchronize(s));
    return SCAL_OK;
}

// saveScancontextAndKeys for a batch of device-resident descriptors in global order (the N ranks' descriptors of one step):
// one launch, no host synchronisation; every shard keeps the ones it owns
extern "C" int scal_sc_insert_descriptors_device(scal_sc_t* c, const double* d_descs, int n) {
    if (!c || !d_descs || n < 0 || n > 64) {
        set_error("scal_sc_insert_descriptors_device: bad argument (at most 64 descriptors per call)");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCSlotList sl;
    int n_local = c->n_local;
    for (int r = 0; r < 64; ++r) sl.slot[r] = -1;
    for (int r = 0; r < n; ++r)
        if (c->owns(c->n_global + r)) {
            if (n_local >= c->cap) {
                set_error("ScanContext database full (%d keyframes on this shard)", c->cap);
                return SCAL_E_CAPACITY;
            }
            sl.slot[r] = n_local++;
        }
    if (n > 0) SCAL_LAUNCH("k_sc_store_batch", k_sc_store_batch, dim3(n), dim3(128), 0, c->stream, d_descs, sl, c->slot(0));
    SCAL_HIP(hipGetLastError());
    c->n_local = n_local;
    c->n_global += n;
    return SCAL_OK;
}

// scal_sc_shard_query_device for a batch: query q uses limits[q] (the tree size at its rebuild, Scancontext.cpp:353-365);
// three launches for the whole batch, results stay on the device (d_out[3 * nq]), no host synchronisation
extern "C" int scal_sc_shard_query_batch_device(scal_sc_t* c, const double* d_queries, int nq, const int* limits, scal_sc_cand* d_out) {
    if (!c || !d_queries || !d_out || !limits || nq < 0 || nq > 64) {
        set_error("scal_sc_shard_query_batch_device: bad argument (at most 64 queries per call)");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const int nb = std::max(1, div_up(c->n_local, 256));
    if (c->bq_cap < nq || c->bq_nb < nb) {
        SCAL_HIP(op_stream_synchronize(s));
        const int cq = std::max(nq, 8), cb = std::max(nb, c->bq_nb);
        SCAL_TRY(c->bq_rkey.alloc((size_t)cq * NR));
        SCAL_TRY(c->bq_skey.alloc((size_t)cq * NS));
        SCAL_TRY(c->bq_norm.alloc((size_t)cq * NS));
        SCAL_TRY(c->bq_best.alloc((size_t)cq * cb * 3));
        SCAL_TRY(c->bq_limits.alloc(cq));
        SCAL_TRY(c->bq_hlimits.alloc(64));
        c->bq_cap = cq, c->bq_nb = cb;
    }
    for (int q = 0; q < nq; ++q) c->bq_hlimits.p[q] = limits[q] - 30;  // NUM_EXCLUDE_RECENT
    SCAL_HIP(op_memcpy_async(c->bq_limits.p, c->bq_hlimits.p, sizeof(int) * nq, hipMemcpyHostToDevice, s));
    if (nq > 0) {
        SCAL_LAUNCH("k_sc_keys", k_sc_keys, dim3(nq), dim3(128), 0, s, d_queries, c->bq_rkey.p, c->bq_skey.p, c->bq_norm.p);
        SCAL_LAUNCH("k_sc_topk", k_sc_topk, dim3(nb, nq), dim3(256), 0, s, c->rkey.p, c->bq_rkey.p, c->n_local, c->cfg.n_shards, c->cfg.shard, 0,
                           c->bq_best.p, c->bq_limits.p);
        SCAL_LAUNCH("k_sc_detect", k_sc_detect, dim3(1, nq), dim3(256), 0, s, c->bq_best.p, nb, c->db(), c->cfg.n_shards, c->cfg.shard, d_queries, c->bq_skey.p,
                           c->bq_norm.p, 1, reinterpret_cast<SCRec*>(d_out));
    }
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

extern "C" int scal_sc_sync(scal_sc_t* c) {
    if (!c) return SCAL_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_HIP(op_stream_synchronize(c->stream));
    return SCAL_OK;
}

extern "C" int scal_sc_get_descriptor(scal_sc_t* c, int idx, double* desc, float* ringkey20) {
    if (!c) return SCAL_E_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    if (idx < 0 || idx >= c->n_global || !c->owns(idx)) {
        set_error("keyframe %d is not stored on this shard", idx);
        return SCAL_E_ARG;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    const size_t sl = idx / c->cfg.n_shards;
    if (desc) SCAL_HIP(op_memcpy_async(desc, c->desc.p + sl * DESC, sizeof(double) * DESC, hipMemcpyDeviceToHost, c->stream));
    if (ringkey20) SCAL_HIP(op_memcpy_async(ringkey20, c->rkey.p + sl * NR, sizeof(float) * NR, hipMemcpyDeviceToHost, c->stream));
    SCAL_HIP(op_stream_synchronize(c->stream));
    return SCAL_OK;
}

// ring-key top-3 over global indices < limit on this shard + SC distance of the three; records -> h_rec[0..2]
static int search_local(scal_sc* c, int limit, bool fill_zero, SCSlot q, bool wait = true, int slot = 0) {
    hipStream_t s = c->stream;
    SCRec* d_rec = c->d_rec.p + 4 * slot;
    SCRec* h_rec = c->h_rec.p + 4 * slot;
    const int nb = std::max(1, div_up(c->n_local, 256));
    SCAL_LAUNCH("k_sc_topk", k_sc_topk, dim3(nb), dim3(256), 0, s, c->rkey.p, q.rkey, c->n_local, c->cfg.n_shards, c->cfg.shard, limit,
                       c->block_best.p, static_cast<const int*>(nullptr));
    SCAL_LAUNCH("k_sc_detect", k_sc_detect, dim3(1), dim3(256), 0, s, c->block_best.p, nb, c->db(), c->cfg.n_shards, c->cfg.shard, q.desc, q.skey,
                       q.cnorm, fill_zero ? 1 : 0, wait ? d_rec : h_rec);  // queued searches write their records straight to pinned host memory
    SCAL_HIP(hipGetLastError());
    if (wait) {
        SCAL_HIP(op_memcpy_async(h_rec, d_rec, sizeof(SCRec) * 3, hipMemcpyDeviceToHost, s));
        SCAL_HIP(op_stream_synchronize(s));
    }
    return SCAL_OK;
}

static void finish_result(const SCRec* rec, int n, double thres, scal_sc_result* res) {
    // the three smallest key distances overall (ties: lower index), evaluated in that order with strict '<' (:385-400)
    std::vector<SCRec> v;
    for (int i = 0; i < n; ++i)
        if (rec[i].idx >= 0) v.push_back(rec[i]);
    std::stable_sort(v.begin(), v.end(), [](const SCRec& a, const SCRec& b) { return a.key_dist < b.key_dist || (a.key_dist == b.key_dist && a.idx < b.idx); });
    if (v.size() > 3) v.resize(3);
    double min_dist = 10000000;
    int nn_align = 0, nn_idx = 0;
    for (int i = 0; i < 3; ++i) {
        res->cand_idx[i] = 0, res->cand_keydist[i] = 0.f, res->cand_scdist[i] = 10000000, res->cand_shift[i] = 0;
    }
    for (size_t i = 0; i < v.size(); ++i) {
        res->cand_idx[i] = v[i].idx, res->cand_keydist[i] = v[i].key_dist, res->cand_scdist[i] = v[i].sc_dist, res->cand_shift[i] = v[i].shift;
        if (v[i].sc_dist < min_dist) min_dist = v[i].sc_dist, nn_align = v[i].shift, nn_idx = v[i].idx;
    }
    res->loop_id = (min_dist < thres) ? nn_idx : -1;  // :406-408
    res->min_dist = min_dist;
    res->nn_idx = nn_idx;
    res->nn_shift = nn_align;
    const float deg = static_cast<float>(nn_align * (360.0 / 60.0));  // PC_UNIT_SECTORANGLE; deg2rad(float) (:17-20, :422)
    res->yaw_rad = static_cast<float>(deg * M_PI / 180.0);
}

// detectLoopClosureID (:335-427) in two halves so that the search can run behind other work: enqueue launches the ring-key
// search + candidate distances for the newest keyframe, collect waits and applies the threshold.
static int detect_enqueue(scal_sc* c) {
    if (c->cfg.n_shards > 1) {
        set_error("scal_sc_detect needs the whole database on one context; use scal_sc_shard_query + scal_sc_merge_candidates");
        return SCAL_E_STATE;
    }
    if (c->det_count >= scal_sc::DET_DEPTH) {
        set_error("scal_sc_detect_enqueue: %d detections are queued and not collected", scal_sc::DET_DEPTH);
        return SCAL_E_STATE;
    }
    const int NUM_EXCLUDE_RECENT = 30, TREE_MAKING_PERIOD_ = 30;
    const int slot = (c->det_head + c->det_count) % scal_sc::DET_DEPTH;
    if (c->n_global < NUM_EXCLUDE_RECENT + 1) {  // :346-350
        c->det_mode[slot] = 2;  // nothing launched
        c->det_count++;
        return SCAL_OK;
    }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    if (!c->det_ev[slot]) SCAL_HIP(hipEventCreateWithFlags(&c->det_ev[slot], hipEventDisableTiming));
    if (c->tree_making_period_conter % TREE_MAKING_PERIOD_ == 0) c->size_at_rebuild = c->n_global;  // :353-364
    c->tree_making_period_conter++;
    // query = newest keyframe (:340-341), read in place from its database slot
    SCAL_TRY(search_local(c, c->size_at_rebuild - NUM_EXCLUDE_RECENT, true, c->slot(c->n_global - 1), false, slot));
    SCAL_HIP(op_event_record(c->det_ev[slot], c->stream));
    c->det_mode[slot] = 1;
    c->det_count++;
    return SCAL_OK;
}
// `lk` (the context's mutex, held on entry and on return) is released while the search is waited for: inserts and further searches
// from another thread - scal_pipeline's front thread - go on meanwhile.  The slot stays reserved until the answer has been read.
static int detect_collect(scal_sc* c, scal_sc_result* res, std::unique_lock<std::mutex>* lk = nullptr) {
    std::memset(res, 0, sizeof *res);
    res->loop_id = -1;
    res->min_dist = 10000000;
    if (c->det_count == 0) {
        set_error("scal_sc_detect_collect: no detection enqueued");
        return SCAL_E_STATE;
    }
    const int slot = c->det_head;
    const int mode = c->det_mode[slot];
    int rc = SCAL_OK;
    if (mode != 2) {
        hipError_t e = hipSetDevice(c->cfg.device);
        if (lk) lk->unlock();
        if (e == hipSuccess) e = op_event_synchronize(c->det_ev[slot]);
        if (e == hipSuccess) finish_result(c->h_rec.p + 4 * slot, 3, c->cfg.dist_thres, res);
        if (lk) lk->lock();
        if (e != hipSuccess) {
            set_error("scal_sc_detect_collect: %s", hipGetErrorString(e));
            rc = SCAL_E_HIP;
        }
    }
    c->det_head = (c->det_head + 1) % scal_sc::DET_DEPTH;
    c->det_count--;
    return rc;
}

extern "C" int scal_sc_detect_enqueue(scal_sc_t* c) {
    if (!c) {
        set_error("scal_sc_detect_enqueue: null argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    return detect_enqueue(c);
}
extern "C" int scal_sc_detect_collect(scal_sc_t* c, scal_sc_result* res) {
    if (!c || !res) {
        set_error("scal_sc_detect_collect: null argument");
        return SCAL_E_ARG;
    }
    std::unique_lock<std::mutex> lk(c->mu);
    return detect_collect(c, res, &lk);
}
extern "C" int scal_sc_detect(scal_sc_t* c, scal_sc_result* res) {
    if (!c || !res) {
        set_error("scal_sc_detect: null argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    std::memset(res, 0, sizeof *res);
    res->loop_id = -1;
    res->min_dist = 10000000;
    SCAL_TRY(detect_enqueue(c));
    return detect_collect(c, res);
}

extern "C" int scal_sc_shard_query(scal_sc_t* c, const double* query_desc, int global_size_at_rebuild, scal_sc_cand out[3]) {
    if (!c || !query_desc || !out) {
        set_error("scal_sc_shard_query: null argument");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    SCAL_HIP(op_memcpy_async(c->qdesc.p, query_desc, sizeof(double) * DESC, hipMemcpyHostToDevice, s));
    SCAL_LAUNCH("k_sc_keys", k_sc_keys, dim3(1), dim3(128), 0, s, c->qdesc.p, c->qrkey.p, c->qskey.p, c->qnorm.p);
    SCAL_TRY(search_local(c, global_size_at_rebuild - 30, true, c->staging()));
    static_assert(sizeof(SCRec) == sizeof(scal_sc_cand), "record layout");
    std::memcpy(out, c->h_rec.p, sizeof(SCRec) * 3);
    return SCAL_OK;
}

extern "C" int scal_sc_merge_candidates(const scal_sc_cand* gathered, int n_records, double dist_thres, scal_sc_result* res) {
    if (!gathered || !res || n_records < 0) {
        set_error("scal_sc_merge_candidates: bad argument");
        return SCAL_E_ARG;
    }
    std::memset(res, 0, sizeof *res);
    // a shard that found fewer than three real neighbours pads with keyframe 0 only on shard 0; drop padded
    // duplicates when real candidates exist elsewhere
    std::vector<SCRec> v(n_records);
    std::memcpy(v.data(), gathered, sizeof(SCRec) * n_records);
    int real = 0;
    for (auto& r : v)
        if (r.idx >= 0 && r.key_dist < FLT_MAX) ++real;
    if (real >= 3)
        for (auto& r : v)
            if (!(r.key_dist < FLT_MAX)) r.idx = -1;
    finish_result(v.data(), n_records, dist_thres, res);
    return SCAL_OK;
}


namespace scal {
// Batch loop search: the k smallest distances of every row of a dense distance block, among the database entries that are older
// than the query by more than `exclude` keyframes (NUM_EXCLUDE_RECENT, Scancontext.h:92): d0 + j < q0 + row - exclude.
// One workgroup per row; k rounds of a block-wide argmin over (distance, index) keys that must exceed the previous pick:
// deterministic, ties go to the lower index.  Entries with NaN distance (no overlapping sector at any shift) never win.
__device__ __forceinline__ void k_sc_row_topk_body(const double* __restrict__ dist, const int* __restrict__ shift, int nq, int nd, int q0, int d0,
                                                     int exclude, int k, int* __restrict__ out_idx, double* __restrict__ out_dist,
                                                     int* __restrict__ out_shift) {
    __shared__ unsigned long long s_d[4];
    __shared__ int s_j[4];
    const int row = blockIdx.x;
    if (row >= nq) return;
    const double* drow = dist + static_cast<size_t>(row) * nd;
    const int limit = min(nd, q0 + row - exclude - d0);  // eligible columns: [0, limit)
    unsigned long long last_d = 0;
    int last_j = -1;
    bool first = true;
    for (int r = 0; r < k; ++r) {
        unsigned long long bd = ~0ull;
        int bj = 0x7fffffff;
        for (int j = threadIdx.x; j < limit; j += 256) {
            const double v = drow[j];
            if (!(v == v)) continue;
            const unsigned long long u = static_cast<unsigned long long>(__double_as_longlong(v < 0 ? 0.0 : v));  // distances are >= 0: bit order = value order
            const bool after = first || u > last_d || (u == last_d && j > last_j);
            if (after && (u < bd || (u == bd && j < bj))) bd = u, bj = j;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long od = shfl_xor_u64(bd, o);
            const int oj = __shfl_xor(bj, o, 64);
            if (od < bd || (od == bd && oj < bj)) bd = od, bj = oj;
        }
        if (lane_id() == 0) s_d[wave_id()] = bd, s_j[wave_id()] = bj;
        __syncthreads();
        bd = s_d[0], bj = s_j[0];
        for (int w = 1; w < 4; ++w)
            if (s_d[w] < bd || (s_d[w] == bd && s_j[w] < bj)) bd = s_d[w], bj = s_j[w];
        __syncthreads();
        if (threadIdx.x == 0) {
            const bool have = bj != 0x7fffffff;
            out_idx[row * k + r] = have ? d0 + bj : -1;
            out_dist[row * k + r] = have ? drow[bj] : 10000000.0;
            out_shift[row * k + r] = have ? shift[static_cast<size_t>(row) * nd + bj] : 0;
        }
        if (bj == 0x7fffffff) {  // fewer than k eligible entries: the remaining slots are empty too
            for (int r2 = r + 1 + threadIdx.x; r2 < k; r2 += 256) out_idx[row * k + r2] = -1, out_dist[row * k + r2] = 10000000.0, out_shift[row * k + r2] = 0;
            return;
        }
        last_d = bd, last_j = bj, first = false;
    }
}
SCAL_KERNEL(256, k_sc_row_topk)
}  // namespace scal

static int ensure_pairs(scal_sc* c, size_t n) {
    if (n > c->pair_cap) {
        SCAL_TRY(c->d_pairs.alloc(2 * n));
        SCAL_TRY(c->d_dist.alloc(n));
        SCAL_TRY(c->d_shift.alloc(n));
        c->pair_cap = n;
    }
    return SCAL_OK;
}

extern "C" int scal_sc_distance_pairs(scal_sc_t* c, const int* idx_a, const int* idx_b, int n_pairs, double* dist, int* shift) {
    if (!c || n_pairs < 0 || (n_pairs > 0 && (!idx_a || !idx_b || !dist || !shift))) {
        set_error("scal_sc_distance_pairs: bad argument");
        return SCAL_E_ARG;
    }
    if (n_pairs == 0) return SCAL_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->cfg.n_shards > 1) {
        set_error("scal_sc_distance_pairs needs an unsharded context");
        return SCAL_E_STATE;
    }
    for (int i = 0; i < n_pairs; ++i)
        if (idx_a[i] < 0 || idx_a[i] >= c->n_global || idx_b[i] < 0 || idx_b[i] >= c->n_global) {
            set_error("pair %d references a keyframe outside [0, %d)", i, c->n_global);
            return SCAL_E_ARG;
        }
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_TRY(ensure_pairs(c, n_pairs));
    hipStream_t s = c->stream;
    SCAL_HIP(op_memcpy_async(c->d_pairs.p, idx_a, sizeof(int) * n_pairs, hipMemcpyHostToDevice, s));
    SCAL_HIP(op_memcpy_async(c->d_pairs.p + n_pairs, idx_b, sizeof(int) * n_pairs, hipMemcpyHostToDevice, s));
    SCAL_LAUNCH("k_sc_pairs", k_sc_pairs, dim3(div_up(n_pairs, 4)), dim3(256), 0, s, c->db(), c->d_pairs.p, c->d_pairs.p + n_pairs, n_pairs, c->d_dist.p,
                       c->d_shift.p);
    SCAL_HIP(hipGetLastError());
    SCAL_HIP(op_memcpy_async(dist, c->d_dist.p, sizeof(double) * n_pairs, hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_memcpy_async(shift, c->d_shift.p, sizeof(int) * n_pairs, hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    return SCAL_OK;
}

template <typename T>
static int enqueue_gram(scal_sc* c, int q0, int nq, int d0, int nd, DevBuf<T>& frag, DevBuf<T>& bhat, double* d_dist, int* d_shift, hipStream_t s) {
    constexpr int GE = GramT<T>::GE;
    const int qblocks = div_up(nq, GQ), n_tiles = qblocks * 4, chunks = div_up(nd, GE);
    if (qblocks > 65535) {
        set_error("scal_sc_distance_matrix: more than 65535 x 64 queries in one call");
        return SCAL_E_ARG;
    }
    const size_t frag_n = static_cast<size_t>(n_tiles) * GKS * 64;
    if (frag.n < frag_n) SCAL_TRY(frag.alloc(frag_n));
    if (c->g_qmask.n < static_cast<size_t>(n_tiles) * 16) SCAL_TRY(c->g_qmask.alloc(static_cast<size_t>(n_tiles) * 16));
    if (bhat.n < static_cast<size_t>(nd) * DESC) SCAL_TRY(bhat.alloc(static_cast<size_t>(nd) * DESC));
    if (c->g_dmask.n < static_cast<size_t>(nd)) SCAL_TRY(c->g_dmask.alloc(nd));
    const size_t lds = sizeof(T) * GE * NR * GROW;  // 79,360 B: above the 64 KiB default, two workgroups per CU
    SCAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sc_gram<T>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const bool f64 = sizeof(T) == 8;
    SCAL_LAUNCH_PROF(f64 ? "k_sc_gram_prep" : "k_sc_gram_prep_f32", k_sc_gram_prep<T>, dim3(n_tiles), dim3(256), 0, s, c->db(), q0, nq, frag.p,
                     c->g_qmask.p);
    SCAL_LAUNCH_PROF(f64 ? "k_sc_gram_prep_db" : "k_sc_gram_prep_db_f32", k_sc_gram_prep_db<T>, dim3(nd), dim3(256), 0, s, c->db(), d0, nd, bhat.p,
                     c->g_dmask.p);
    SCAL_LAUNCH_PROF(f64 ? "k_sc_gram" : "k_sc_gram_f32", k_sc_gram<T>, dim3(chunks, qblocks), dim3(256), lds, s, bhat.p, c->g_dmask.p, frag.p,
                     c->g_qmask.p, nq, nd, d_dist, d_shift);
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

// mode 0 / 1: one wave per pair on the vector ALUs (summation order of the reference); mode 2: matrix cores (k_sc_gram)
static int enqueue_matrix(scal_sc* c, int q0, int q1, int d0, int d1, int mode, double* d_dist, int* d_shift, hipStream_t s) {
    const int nq = q1 - q0, nd = d1 - d0;
    const size_t np = static_cast<size_t>(nq) * nd;
    if (mode < 2) {
        SCAL_LAUNCH("k_sc_matrix", k_sc_matrix, dim3(static_cast<unsigned>((np + 3) / 4)), dim3(256), 0, s, c->db(), q0, nq, d0, nd, mode,
                         d_dist, d_shift);
        SCAL_HIP(hipGetLastError());
        return SCAL_OK;
    }
    return mode == 2 ? enqueue_gram<double>(c, q0, nq, d0, nd, c->g_frag, c->g_bhat, d_dist, d_shift, s)
                     : enqueue_gram<float>(c, q0, nq, d0, nd, c->g_frag32, c->g_bhat32, d_dist, d_shift, s);
}

static int check_matrix_args(scal_sc* c, const void* dist, const void* shift, int q0, int q1, int d0, int d1, int mode) {
    if (!c || !dist || !shift || q0 < 0 || d0 < 0 || q1 < q0 || d1 < d0 || mode < 0 || mode > 3) {
        set_error("scal_sc_distance_matrix: bad argument");
        return SCAL_E_ARG;
    }
    return SCAL_OK;
}

extern "C" int scal_sc_distance_matrix(scal_sc_t* c, int q0, int q1, int d0, int d1, int mode, double* dist, int* shift) {
    SCAL_TRY(check_matrix_args(c, dist, shift, q0, q1, d0, d1, mode));
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->cfg.n_shards > 1 || q1 > c->n_global || d1 > c->n_global) {
        set_error("scal_sc_distance_matrix: range outside the database (or sharded context)");
        return SCAL_E_ARG;
    }
    const size_t np = static_cast<size_t>(q1 - q0) * (d1 - d0);
    if (np == 0) return SCAL_OK;
    SCAL_HIP(hipSetDevice(c->cfg.device));
    SCAL_TRY(ensure_pairs(c, np));
    hipStream_t s = c->stream;
    SCAL_TRY(enqueue_matrix(c, q0, q1, d0, d1, mode, c->d_dist.p, c->d_shift.p, s));
    SCAL_HIP(op_memcpy_async(dist, c->d_dist.p, sizeof(double) * np, hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_memcpy_async(shift, c->d_shift.p, sizeof(int) * np, hipMemcpyDeviceToHost, s));
    SCAL_HIP(op_stream_synchronize(s));
    return SCAL_OK;
}

extern "C" int scal_sc_distance_matrix_device(scal_sc_t* c, int q0, int q1, int d0, int d1, int mode, double* d_dist, int* d_shift) {
    SCAL_TRY(check_matrix_args(c, d_dist, d_shift, q0, q1, d0, d1, mode));
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->cfg.n_shards > 1 || q1 > c->n_global || d1 > c->n_global) {
        set_error("scal_sc_distance_matrix_device: range outside the database (or sharded context)");
        return SCAL_E_ARG;
    }
    if (q1 == q0 || d1 == d0) return SCAL_OK;
    SCAL_HIP(hipSetDevice(c->cfg.device));
    return enqueue_matrix(c, q0, q1, d0, d1, mode, d_dist, d_shift, c->stream);
}

// Exhaustive loop mining over a stored session: for every query keyframe q in [q0, q1) its k best matches among the keyframes older
// than q - exclude_recent, by the dense distance block of `mode` (2: f64 matrix cores) - row tiles of 512 queries, top-k on the
// device, only k records per query come back.
extern "C" int scal_sc_batch_loop_search(scal_sc_t* c, int q0, int q1, int exclude_recent, int k, int mode, int* idx, double* dist, int* shift) {
    if (!c || !idx || !dist || !shift || q0 < 0 || q1 < q0 || exclude_recent < 0 || k < 1 || k > 16 || mode < 0 || mode > 3) {
        set_error("scal_sc_batch_loop_search: bad argument (1 <= k <= 16)");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->cfg.n_shards > 1 || q1 > c->n_global) {
        set_error("scal_sc_batch_loop_search: range outside the database (or sharded context)");
        return SCAL_E_ARG;
    }
    if (q1 == q0) return SCAL_OK;
    SCAL_HIP(hipSetDevice(c->cfg.device));
    hipStream_t s = c->stream;
    const int TILE_Q = 512;
    const int nd_max = std::max(1, q1 - 1 - exclude_recent);  // the newest query sees [0, q1 - 1 - exclude)
    SCAL_TRY(ensure_pairs(c, static_cast<size_t>(TILE_Q) * nd_max));
    DevBuf<int> o_idx, o_shift;
    DevBuf<double> o_dist;
    SCAL_TRY(o_idx.alloc(static_cast<size_t>(TILE_Q) * k));
    SCAL_TRY(o_shift.alloc(static_cast<size_t>(TILE_Q) * k));
    SCAL_TRY(o_dist.alloc(static_cast<size_t>(TILE_Q) * k));
    for (int t0 = q0; t0 < q1; t0 += TILE_Q) {
        const int t1 = std::min(q1, t0 + TILE_Q), nq = t1 - t0;
        const int nd = std::max(0, t1 - 1 - exclude_recent);  // columns any query of this tile can use
        if (nd > 0) SCAL_TRY(enqueue_matrix(c, t0, t1, 0, nd, mode, c->d_dist.p, c->d_shift.p, s));
        SCAL_LAUNCH("k_sc_row_topk", k_sc_row_topk, dim3(nq), dim3(256), 0, s, c->d_dist.p, c->d_shift.p, nq, nd, t0, 0, exclude_recent, k, o_idx.p,
                         o_dist.p, o_shift.p);
        SCAL_HIP(hipGetLastError());
        const size_t off = static_cast<size_t>(t0 - q0) * k;
        SCAL_HIP(op_memcpy_async(idx + off, o_idx.p, sizeof(int) * nq * k, hipMemcpyDeviceToHost, s));
        SCAL_HIP(op_memcpy_async(dist + off, o_dist.p, sizeof(double) * nq * k, hipMemcpyDeviceToHost, s));
        SCAL_HIP(op_memcpy_async(shift + off, o_shift.p, sizeof(int) * nq * k, hipMemcpyDeviceToHost, s));
        SCAL_HIP(op_stream_synchronize(s));
    }
    return SCAL_OK;
}

extern "C" void* scal_sc_stream(scal_sc_t* c) { return c ? static_cast<void*>(c->stream) : nullptr; }
