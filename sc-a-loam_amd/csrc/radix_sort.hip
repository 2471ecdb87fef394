// Stable LSD radix sort (8-bit digits) for gfx950.  Three kernels per pass:
//   k_rs_hist     per-block digit histogram (LDS atomics)            -> hist[digit][block]
//   k_scan        one-block exclusive scan over the digit-major histogram
//   k_rs_scatter  stable scatter: per wave, items are ranked with ballot-based digit matching (8 ballots give
//                 the set of lanes holding the same digit), per-wave digit counters live in LDS
// The element handled by (wave w, item j, lane l) of a block is base + w*ITEMS*64 + j*64 + l, so (w, j, l) order is
// arrival order and equal keys keep their relative order.
#include "radix_sort.hpp"
#include "device_utils.hpp"

namespace scal {

constexpr int RS_ITEMS = RadixSort::ITEMS;
constexpr int RS_TILE = RadixSort::TILE;

__global__ void __launch_bounds__(256) k_rs_hist(const unsigned long long* __restrict__ keys, const int* __restrict__ d_n, int shift,
                                                 int* __restrict__ hist) {
    const int n = *d_n;
    const int nb = (n + RS_TILE - 1) / RS_TILE;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int e = base + j * 256 + threadIdx.x;
        if (e < n) atomicAdd(&h[(keys[e] >> shift) & 0xff], 1);
    }
    __syncthreads();
    hist[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(1024) k_scan(int* __restrict__ data, const int* __restrict__ d_n, int tile, int bins, int* __restrict__ d_total) {
    __shared__ int smem[17];
    const int n = *d_n;
    const int m = bins * ((n + tile - 1) / tile);
    const int per = (m + 1023) / 1024;
    const int b0 = min(m, static_cast<int>(threadIdx.x) * per), b1 = min(m, b0 + per);
    int sum = 0;
    for (int i = b0; i < b1; ++i) sum += data[i];
    int total;
    int run = block_exclusive_scan(sum, smem, &total);
    for (int i = b0; i < b1; ++i) {
        const int v = data[i];
        data[i] = run;
        run += v;
    }
    if (d_total && threadIdx.x == 0) *d_total = total;
}

__global__ void __launch_bounds__(256) k_rs_scatter(const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                                    const int* __restrict__ d_n, int shift, const int* __restrict__ hist,
                                                    unsigned long long* __restrict__ okeys, int* __restrict__ ovals) {
    const int n = *d_n;
    const int nb = (n + RS_TILE - 1) / RS_TILE;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    __shared__ int cnt[4][256];
    const int w = wave_id(), l = lane_id();
    for (int i = threadIdx.x; i < 1024; i += 256) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE + w * (RS_ITEMS * 64);
    unsigned long long k[RS_ITEMS];
    int rk[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int e = base + j * 64 + l;
        const bool valid = e < n;
        k[j] = valid ? keys[e] : 0ull;
        const uint32_t d = static_cast<uint32_t>(k[j] >> shift) & 0xffu;
        const uint64_t m = wave_match<8>(d, valid);
        int prev = 0;
        if (valid) prev = cnt[w][d];
        rk[j] = prev + __popcll(m & lanemask_lt());
        __builtin_amdgcn_wave_barrier();
        if (valid && (m & lanemask_lt()) == 0) cnt[w][d] = prev + __popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int e = base + j * 64 + l;
        if (e < n) {
            const uint32_t d = static_cast<uint32_t>(k[j] >> shift) & 0xffu;
            int pre = 0;
            for (int ww = 0; ww < w; ++ww) pre += cnt[ww][d];
            const int pos = hist[d * nb + blockIdx.x] + pre + rk[j];
            okeys[pos] = k[j];
            ovals[pos] = vals[e];
        }
    }
}

int RadixSort::init(int capacity) {
    cap = capacity;
    SCAL_TRY(keys_alt.alloc(cap));
    SCAL_TRY(vals_alt.alloc(cap));
    SCAL_TRY(hist.alloc((size_t)256 * div_up(cap, TILE) + 256));
    return SCAL_OK;
}

int RadixSort::sort(hipStream_t s, unsigned long long* keys, int* vals, const int* d_n, int begin_bit, int end_bit,
                    unsigned long long** out_keys, int** out_vals) {
    unsigned long long* ka = keys;
    unsigned long long* kb = keys_alt.p;
    int* va = vals;
    int* vb = vals_alt.p;
    const int nb_cap = max(1, div_up(cap, TILE));
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        hipLaunchKernelGGL(k_rs_hist, dim3(nb_cap), dim3(256), 0, s, ka, d_n, shift, hist.p);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, hist.p, d_n, TILE, 256, static_cast<int*>(nullptr));
        {
        ProfScope ps("k_rs_scatter", s);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nb_cap), dim3(256), 0, s, ka, va, d_n, shift, hist.p, kb, vb);
        }
        std::swap(ka, kb);
        std::swap(va, vb);
    }
    *out_keys = ka;
    *out_vals = va;
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

void launch_scan_inplace(hipStream_t s, int* data, const int* d_n, int tile, int bins, int* d_total) {
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, data, d_n, tile, bins, d_total);
}

}  // namespace scal
