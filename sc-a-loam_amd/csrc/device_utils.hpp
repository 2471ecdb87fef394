// Device-side building blocks for gfx950 (wave64): wave/block reductions and scans, LDS bitonic sort,
// order-preserving float<->uint maps, stable multi-bin ranking.  Everything here is compiled with
// -ffp-contract=off: several kernels must reproduce the reference's f32 rounding bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scal {

// Development aid (make STAMPS=1): section time stamps inside single-workgroup kernels.  SCAL_STAMP(i) stores the 100 MHz
// wall clock into this translation unit's g_stamps[i]; SCAL_DEFINE_STAMP_READER(fn) exports a C function copying them out.
#ifdef SCAL_STAMPS
static __device__ long long g_stamps[32];
#define SCAL_STAMP(i)                                                   \
    do {                                                                \
        __builtin_amdgcn_s_waitcnt(0);                                  \
        ::scal::g_stamps[i] = static_cast<long long>(wall_clock64());   \
    } while (0)
#define SCAL_DEFINE_STAMP_READER(fn)                                                                                      \
    extern "C" int fn(long long* out32) {                                                                                 \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                                              \
        return hipMemcpyFromSymbol(out32, HIP_SYMBOL(::scal::g_stamps), sizeof(long long) * 32) == hipSuccess ? 0 : -1;   \
    }
#else
#define SCAL_STAMP(i) \
    do {              \
    } while (0)
#define SCAL_DEFINE_STAMP_READER(fn)
#endif

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// order-preserving map of a float onto uint32 (works for negative values too)
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
// The same minimum through the DPP network (row shifts + the two row broadcasts of gfx9) instead of six LDS-crossbar permutes:
// ~30 VALU operations and no ds_bpermute latency chain.  The result is uniform (read from lane 63).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_min_step_u64(unsigned long long v) {
    const int lo = static_cast<int>(v), hi = static_cast<int>(v >> 32);
    const unsigned tl = static_cast<unsigned>(__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false));
    const unsigned th = static_cast<unsigned>(__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false));
    const unsigned long long t = (static_cast<unsigned long long>(th) << 32) | tl;
    return t < v ? t : v;
}
__device__ __forceinline__ unsigned long long wave_min_u64_dpp(unsigned long long v) {
    v = dpp_min_step_u64<0x111, 0xf>(v);  // row_shr:1
    v = dpp_min_step_u64<0x112, 0xf>(v);  // row_shr:2
    v = dpp_min_step_u64<0x114, 0xf>(v);  // row_shr:4
    v = dpp_min_step_u64<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row minimum
    v = dpp_min_step_u64<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3 take in rows 0 and 2
    v = dpp_min_step_u64<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3 take in lane 31: lane 63 holds the wave minimum
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v >> 32), 63));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}
// minimum over each 32-lane half of the wave (lanes 0-31 / 32-63), returned to every lane of that half
__device__ __forceinline__ unsigned long long half_min_u64_dpp(unsigned long long v) {
    v = dpp_min_step_u64<0x111, 0xf>(v);  // row_shr:1
    v = dpp_min_step_u64<0x112, 0xf>(v);  // row_shr:2
    v = dpp_min_step_u64<0x114, 0xf>(v);  // row_shr:4
    v = dpp_min_step_u64<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row minimum
    v = dpp_min_step_u64<0x142, 0xa>(v);  // row_bcast:15 -> lanes 31 and 63 hold the minima of their halves
    const unsigned lo0 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 31));
    const unsigned hi0 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v >> 32), 31));
    const unsigned lo1 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
    const unsigned hi1 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v >> 32), 63));
    const unsigned long long m0 = (static_cast<unsigned long long>(hi0) << 32) | lo0, m1 = (static_cast<unsigned long long>(hi1) << 32) | lo1;
    return (threadIdx.x & 32) ? m1 : m0;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (l >= o) v += t;
    }
    return v;
}

// exclusive prefix sum over blockDim.x (<=1024) values; returns the exclusive prefix, *total gets the block sum.
// `smem` must hold 17 ints.  Contains __syncthreads().
__device__ __forceinline__ int block_exclusive_scan(int v, int* smem, int* total) {
    const int incl = wave_inclusive_scan(v);
    const int w = wave_id(), l = lane_id();
    const int nw = (blockDim.x + 63) >> 6;
    if (l == 63) smem[w] = incl;
    __syncthreads();
    if (w == 0) {
        int s = l < nw ? smem[l] : 0;
        int si = wave_inclusive_scan(s);
        if (l < nw) smem[l] = si - s;
        if (l == nw - 1) smem[16] = si;
    }
    __syncthreads();
    const int base = smem[w];
    if (total) *total = smem[16];
    __syncthreads();
    return base + incl - v;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    const unsigned lo = static_cast<unsigned>(__shfl_xor(static_cast<int>(v & 0xffffffffu), mask, 64));
    const unsigned hi = static_cast<unsigned>(__shfl_xor(static_cast<int>(v >> 32), mask, 64));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// value of lane (lane ^ X) for the lane masks a bitonic network needs.  X = 1, 2, 3, 7, 15 are DPP modifiers (quad_perm,
// row_half_mirror, row_mirror: full VALU rate, no LDS round trip), 4, 8, 16, 31 use ds_swizzle (no address VGPR), 32 and 63
// fall back to ds_bpermute.
template <int X>
__device__ __forceinline__ int lane_xor_i32(int v) {
    if constexpr (X == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    else if constexpr (X == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    else if constexpr (X == 3) return __builtin_amdgcn_mov_dpp(v, 0x1B, 0xf, 0xf, false);   // quad_perm [3,2,1,0]
    else if constexpr (X == 7) return __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    else if constexpr (X == 15) return __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, false); // row_mirror
    else if constexpr (X == 4 || X == 8 || X == 16 || X == 31) return __builtin_amdgcn_ds_swizzle(v, 0x1F | (X << 10));
    else return __shfl_xor(v, X, 64);
}
template <int X>
__device__ __forceinline__ unsigned long long lane_xor_u64(unsigned long long v) {
    const unsigned lo = static_cast<unsigned>(lane_xor_i32<X>(static_cast<int>(v & 0xffffffffu)));
    const unsigned hi = static_cast<unsigned>(lane_xor_i32<X>(static_cast<int>(v >> 32)));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// ---- ascending bitonic sort of 512 keys held by ONE wave, 8 per lane, element index = lane*8 + r.
// "Flip" form of the network: stage k first compares i with its mirror image inside the k-block, then runs half-cleaners
// j = k/4 .. 1; every compare is ascending, so keys equal to the maximum (padding) never move down.  With this layout the
// distances 1, 2, 4 stay inside a lane (24 of the 45 steps), the lane masks of the others are 1, 2, 3, 7, 15 (DPP),
// 4, 8, 16, 31 (swizzle) and 32, 63 (bpermute): 8 steps through the LDS crossbar instead of 39.
__device__ __forceinline__ void cmpswap(unsigned long long& a, unsigned long long& b) {
    const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo, b = hi;
}
template <int J>
__device__ __forceinline__ void reg_half_cleaner(unsigned long long (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if ((r & J) == 0) cmpswap(v[r], v[r | J]);
}
template <int K>
__device__ __forceinline__ void reg_mirror(unsigned long long (&v)[8]) {  // K = 2, 4, 8 registers per block
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if ((r & (K >> 1)) == 0) cmpswap(v[r], v[r ^ (K - 1)]);
}
__device__ __forceinline__ void reg_tail(unsigned long long (&v)[8]) {
    reg_half_cleaner<4>(v);
    reg_half_cleaner<2>(v);
    reg_half_cleaner<1>(v);
}
// half-cleaner between lanes l and l ^ X (same register)
template <int X>
__device__ __forceinline__ void lane_half_cleaner(unsigned long long (&v)[8]) {
    const bool lower = (lane_id() & X) == 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const unsigned long long a = v[r], p = lane_xor_u64<X>(a);
        v[r] = ((a < p) == lower) ? a : p;
    }
}
// mirror step of a stage spanning M lanes: (lane, r) <-> (lane ^ (M-1), 7 - r)
template <int M>
__device__ __forceinline__ void lane_mirror(unsigned long long (&v)[8]) {
    const bool lower = (lane_id() & (M >> 1)) == 0;
    unsigned long long p[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) p[r] = lane_xor_u64<M - 1>(v[7 - r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = ((v[r] < p[r]) == lower) ? v[r] : p[r];
}
template <int M>
__device__ __forceinline__ void lane_stage(unsigned long long (&v)[8]) {  // stage k = 8*M
    lane_mirror<M>(v);
    if constexpr (M >= 64) lane_half_cleaner<16>(v);
    if constexpr (M >= 32) lane_half_cleaner<8>(v);
    if constexpr (M >= 16) lane_half_cleaner<4>(v);
    if constexpr (M >= 8) lane_half_cleaner<2>(v);
    if constexpr (M >= 4) lane_half_cleaner<1>(v);
    reg_tail(v);
}
__device__ __forceinline__ void wave_sort512(unsigned long long (&v)[8]) {
    reg_mirror<2>(v);
    reg_mirror<4>(v);
    reg_half_cleaner<1>(v);
    reg_mirror<8>(v);
    reg_half_cleaner<2>(v);
    reg_half_cleaner<1>(v);
    lane_stage<2>(v);
    lane_stage<4>(v);
    lane_stage<8>(v);
    lane_stage<16>(v);
    lane_stage<32>(v);
    lane_stage<64>(v);
}
// the in-chunk part (distances 256 .. 1) of a merge stage k >= 1024
__device__ __forceinline__ void wave_merge512(unsigned long long (&v)[8]) {
    lane_half_cleaner<32>(v);
    lane_half_cleaner<16>(v);
    lane_half_cleaner<8>(v);
    lane_half_cleaner<4>(v);
    lane_half_cleaner<2>(v);
    lane_half_cleaner<1>(v);
    reg_tail(v);
}
__device__ __forceinline__ void chunk_load(const unsigned long long* s, int c, unsigned long long (&v)[8]) {
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(s + c * 512 + lane_id() * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const ulonglong2 t = p[q];
        v[2 * q] = t.x, v[2 * q + 1] = t.y;
    }
}
__device__ __forceinline__ void chunk_store(unsigned long long* s, int c, const unsigned long long (&v)[8]) {
    ulonglong2* p = reinterpret_cast<ulonglong2*>(s + c * 512 + lane_id() * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = make_ulonglong2(v[2 * q], v[2 * q + 1]);
}

// Ascending sort of s[0 .. n_pow2) in LDS (16-byte aligned, n_pow2 a power of two >= 512) by the whole block.  Entries at
// index >= n_live must hold the maximum key (~0): they are never touched, so the work follows n_live, not n_pow2.
// 512-chunks are sorted in registers; of each merge stage only the steps that cross chunks go through LDS with barriers.
__device__ __forceinline__ void block_sort_u64(unsigned long long* s, int n_pow2, int n_live) {
    const int nw = blockDim.x >> 6, w = wave_id();
    const int nc = (n_live + 511) >> 9;  // chunks holding live keys
    for (int c = w; c < nc; c += nw) {
        unsigned long long v[8];
        chunk_load(s, c, v);
        wave_sort512(v);
        chunk_store(s, c, v);
    }
    __syncthreads();
    const int half = n_pow2 >> 1;
    for (int k = 1024; k <= n_pow2; k <<= 1) {
        if ((k >> 1) >= n_live) break;  // only padding beyond the first half: already sorted
        // mirror step: o-th element of a k-block's lower half against the o-th from the block's end
        for (int t = threadIdx.x; t < half; t += blockDim.x) {
            const int blk = t / (k >> 1), o = t - blk * (k >> 1);
            const int i = blk * k + o, p = blk * k + k - 1 - o;
            if (p < n_live) {
                const unsigned long long a = s[i], b = s[p];
                if (a > b) s[i] = b, s[p] = a;
            }
        }
        __syncthreads();
        for (int j = k >> 2; j >= 512; j >>= 1) {
            for (int t = threadIdx.x; t < half; t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));
                const int p = i + j;
                if (p < n_live) {
                    const unsigned long long a = s[i], b = s[p];
                    if (a > b) s[i] = b, s[p] = a;
                }
            }
            __syncthreads();
        }
        for (int c = w; c < nc; c += nw) {
            unsigned long long v[8];
            chunk_load(s, c, v);
            wave_merge512(v);
            chunk_store(s, c, v);
        }
        __syncthreads();
    }
}

// lanes of the wave whose `digit` equals mine (among `valid` lanes); BITS ballots.
template <int BITS>
__device__ __forceinline__ uint64_t wave_match(uint32_t digit, bool valid) {
    uint64_t mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t bal = __ballot(bit && valid);
        mask &= bit ? bal : ~bal;
    }
    return mask;
}

// the same with a run-time number of digit bits (uniform over the wave)
__device__ __forceinline__ uint64_t wave_match_bits(uint32_t digit, bool valid, int bits) {
    uint64_t mask = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t bal = __ballot(bit && valid);
        mask &= bit ? bal : ~bal;
    }
    return mask;
}

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace scal
