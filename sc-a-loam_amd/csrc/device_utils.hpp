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

// In-LDS bitonic sort of n_pow2 uint64 keys (ascending) by the whole block.  Contains __syncthreads().
__device__ __forceinline__ void block_bitonic_sort_u64(unsigned long long* s, int n_pow2) {
    const int half = n_pow2 >> 1;
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < half; t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));
                const int ixj = i + j;
                const bool up = (i & k) == 0;
                const unsigned long long a = s[i], b = s[ixj];
                if ((a > b) == up) {
                    s[i] = b;
                    s[ixj] = a;
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    const unsigned lo = static_cast<unsigned>(__shfl_xor(static_cast<int>(v & 0xffffffffu), mask, 64));
    const unsigned hi = static_cast<unsigned>(__shfl_xor(static_cast<int>(v >> 32), mask, 64));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// Bitonic stages k = 2 .. 512 on 512 keys held by ONE wave, 8 per lane, element index = r*64 + lane (r = 0..7).
// No LDS, no barrier: partners closer than 64 are exchanged with lane shuffles, the others live in the same lane.
// The direction of every compare follows the global network: ascending iff ((global_base + index) & k) == 0, so the
// chunk can be one 512-block of a larger bitonic sort that continues in LDS from k = 1024.
__device__ __forceinline__ void wave_bitonic_sort512(unsigned long long (&v)[8], int global_base) {
    const int lane = lane_id();
#pragma unroll
    for (int k = 2; k <= 512; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int rr = j >> 6;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if ((r & rr) == 0) {
                        const bool up = ((global_base + r * 64 + lane) & k) == 0;
                        const unsigned long long a = v[r], b = v[r | rr];
                        if ((a > b) == up) v[r] = b, v[r | rr] = a;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const unsigned long long a = v[r];
                    const unsigned long long p = shfl_xor_u64(a, j);
                    const bool up = ((global_base + r * 64 + lane) & k) == 0;
                    const bool lower = (lane & j) == 0;
                    const unsigned long long mn = a < p ? a : p, mx = a < p ? p : a;
                    v[r] = (lower == up) ? mn : mx;
                }
            }
        }
    }
}

// The steps j = 256 .. 1 of merge stage k (k >= 1024) on one 512-chunk held in registers (same layout as above).
__device__ __forceinline__ void wave_bitonic_merge512(unsigned long long (&v)[8], int global_base, int k) {
    const int lane = lane_id();
#pragma unroll
    for (int j = 256; j > 0; j >>= 1) {
        if (j >= 64) {
            const int rr = j >> 6;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if ((r & rr) == 0) {
                    const bool up = ((global_base + r * 64 + lane) & k) == 0;
                    const unsigned long long a = v[r], b = v[r | rr];
                    if ((a > b) == up) v[r] = b, v[r | rr] = a;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const unsigned long long a = v[r];
                const unsigned long long p = shfl_xor_u64(a, j);
                const bool up = ((global_base + r * 64 + lane) & k) == 0;
                const bool lower = (lane & j) == 0;
                const unsigned long long mn = a < p ? a : p, mx = a < p ? p : a;
                v[r] = (lower == up) ? mn : mx;
            }
        }
    }
}

// Bitonic sort of n_pow2 (>= 512, power of two) uint64 keys in LDS.  Every 512-chunk is sorted in registers by one wave;
// of each merge stage k >= 1024 only the steps that cross chunks (j >= 512) go through LDS with block barriers, the
// nine steps inside a chunk run in registers again.  8192 keys: 10 barrier steps instead of 91.
__device__ __forceinline__ void block_bitonic_sort_u64_fast(unsigned long long* s, int n_pow2) {
    const int nw = blockDim.x >> 6, w = wave_id(), lane = lane_id();
    for (int c = w; c * 512 < n_pow2; c += nw) {
        unsigned long long v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = s[c * 512 + r * 64 + lane];
        wave_bitonic_sort512(v, c * 512);
#pragma unroll
        for (int r = 0; r < 8; ++r) s[c * 512 + r * 64 + lane] = v[r];
    }
    __syncthreads();
    const int half = n_pow2 >> 1;
    for (int k = 1024; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j >= 512; j >>= 1) {
            for (int t = threadIdx.x; t < half; t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));
                const int ixj = i + j;
                const bool up = (i & k) == 0;
                const unsigned long long a = s[i], b = s[ixj];
                if ((a > b) == up) {
                    s[i] = b;
                    s[ixj] = a;
                }
            }
            __syncthreads();
        }
        for (int c = w; c * 512 < n_pow2; c += nw) {
            unsigned long long v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = s[c * 512 + r * 64 + lane];
            wave_bitonic_merge512(v, c * 512, k);
#pragma unroll
            for (int r = 0; r < 8; ++r) s[c * 512 + r * 64 + lane] = v[r];
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

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace scal
