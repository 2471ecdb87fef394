// Device-side building blocks for gfx950 (wave64): wave/block reductions and scans, LDS bitonic sort,
// order-preserving float<->uint maps, stable multi-bin ranking.  Everything here is compiled with
// -ffp-contract=off: several kernels must reproduce the reference's f32 rounding bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scal {

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
