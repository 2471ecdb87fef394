// Stable LSD radix sort (digits of up to 10 bits for keys of up to 30 bits, up to 12 beyond; width adapted to the key bits in use)
// for gfx950.  Two kernels per pass:
//   k_rs_hist     per-block digit histogram (LDS atomics) -> histogram matrix
//   k_rs_scatter  every block first derives its own scatter bases from the histogram matrix (sum of the earlier blocks'
//                 counts per digit + exclusive prefix of the digit totals, 512 x nb L2-resident words - cheaper than a
//                 separate scan launch while nb is small), then scatters stably: per wave, items are ranked with
//                 ballot-based digit matching (one ballot per digit bit gives the lanes holding the same digit), per-wave
//                 16-bit digit counters live in LDS.
// The element handled by (wave w, item j, lane l) of a block is base + w*ITEMS*64 + j*64 + l, so (w, j, l) order is
// arrival order and equal keys keep their relative order.
#include "radix_sort.hpp"
#include "device_utils.hpp"

namespace scal {

constexpr int RS_ITEMS = RadixSort::ITEMS;
constexpr int RS_TILE = RadixSort::TILE;
constexpr int RS_BINS_MAX = RadixSort::BINS_MAX;

// this pass's digit position; false when the pass has nothing to do.  used = *d_used_bits, or the host's max_bits
__device__ __forceinline__ bool pass_plan(const int* d_used_bits, int max_bits, int pass, int& shift, int& width) {
    const int used = d_used_bits ? *d_used_bits : max_bits;
    int passes;
    rs_plan(used, passes, width);
    shift = pass * width;
    return pass < passes;
}

// Histogram matrix layout: [block][digit] for ordinary sorts (a wave of the scatter kernel then reads 64 consecutive digits of
// one block = one or two cache lines per load), [digit][block] for sorts that go through the hierarchical scan (its flattened
// order must be digit-major).
__device__ __forceinline__ void k_rs_hist_body(const unsigned long long* __restrict__ keys, const int* __restrict__ d_n, int pass, int max_bits,
                                                 const int* __restrict__ d_used_bits, int* __restrict__ hist, int digit_major) {
    int shift, width;
    if (!pass_plan(d_used_bits, max_bits, pass, shift, width)) return;
    const int bins = 1 << width;
    const int n = *d_n;
    const int nb = (n + RS_TILE - 1) / RS_TILE;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    __shared__ int h[RS_BINS_MAX];
    for (int i = threadIdx.x; i < bins; i += 256) h[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int e = base + j * 256 + threadIdx.x;
        if (e < n) atomicAdd(&h[(keys[e] >> shift) & (bins - 1)], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256) hist[digit_major ? i * nb + blockIdx.x : blockIdx.x * RS_BINS_MAX + i] = h[i];
}
SCAL_KERNEL(256, k_rs_hist)

__device__ __forceinline__ void k_scan_body(int* __restrict__ data, const int* __restrict__ d_n, int tile, int bins, int* __restrict__ d_total) {
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
SCAL_KERNEL(1024, k_scan)

__device__ __forceinline__ void k_rs_scatter_body(const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                                    const int* __restrict__ d_n, int pass, int max_bits, const int* __restrict__ d_used_bits,
                                                    const int* __restrict__ hist, unsigned long long* __restrict__ okeys, int* __restrict__ ovals,
                                                    int scanned) {
    int shift, width;
    if (!pass_plan(d_used_bits, max_bits, pass, shift, width)) return;
    const int bins = 1 << width;
    const int n = *d_n;
    const int nb = (n + RS_TILE - 1) / RS_TILE;
    if (static_cast<int>(blockIdx.x) >= nb) return;
    __shared__ unsigned short cnt[4][RS_BINS_MAX];  // per-wave digit counters: at most 8 items x 64 lanes
    __shared__ int sbase[RS_BINS_MAX];
    __shared__ int smem[17];
    const int w = wave_id(), l = lane_id();
    const bool stamp = blockIdx.x == static_cast<unsigned>(nb / 2) && threadIdx.x == 0 && pass == 0;
    if (stamp) SCAL_STAMP(0);
    // scatter base of digit d for this block = (exclusive prefix of the digit totals)[d] + sum_{b < block} hist[d][b]
    if (scanned) {  // large sorts: the histogram matrix has been turned into scatter bases by the hierarchical scan
        for (int d = threadIdx.x; d < bins; d += 256) sbase[d] = hist[d * nb + blockIdx.x];
    } else {
        // [block][digit] layout: thread t owns digits t, t + 256, ... so a wave reads 64 consecutive words of one block's row.
        // The sums over the rows are a chain of L2 round trips (~1 us each), so as many loads as possible go out together: for digits
        // of up to 10 bits (<= 4 digit groups) a thread fetches 8 rows x all its digits per round trip - 32 independent loads - which
        // makes ceil(nb / 8) round trips per pass instead of ceil(nb / 8) x groups (24 -> 6 for a 95 k-point sort).
        const int me = static_cast<int>(blockIdx.x);
        if (bins <= 1024) {
            constexpr int MAXG = 4;
            const int G = (bins + 255) >> 8;
            int before[MAXG] = {0, 0, 0, 0}, total[MAXG] = {0, 0, 0, 0};
            for (int b0 = 0; b0 < nb; b0 += 8) {
                int v[MAXG][8];
#pragma unroll
                for (int g = 0; g < MAXG; ++g) {
                    const int d = g * 256 + static_cast<int>(threadIdx.x);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[g][q] = (g < G && d < bins && b0 + q < nb) ? hist[(b0 + q) * RS_BINS_MAX + d] : 0;
                }
#pragma unroll
                for (int g = 0; g < MAXG; ++g)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        total[g] += v[g][q];
                        if (b0 + q < me) before[g] += v[g][q];
                    }
            }
            int run = 0;  // digits below this group of 256
#pragma unroll
            for (int g = 0; g < MAXG; ++g) {
                if (g < G) {  // uniform over the block
                    const int d = g * 256 + static_cast<int>(threadIdx.x);
                    int sum;
                    const int pre = block_exclusive_scan(total[g], smem, &sum);
                    if (d < bins) sbase[d] = run + pre + before[g];
                    run += sum;
                }
            }
        } else {
            int run = 0;  // digits below this group of 256
            for (int g = 0; g < bins; g += 256) {
                const int d = g + threadIdx.x;
                int before = 0, total = 0;
                if (d < bins) {
                    for (int b0 = 0; b0 < nb; b0 += 8) {
                        int v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = b0 + q < nb ? hist[(b0 + q) * RS_BINS_MAX + d] : 0;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            total += v[q];
                            if (b0 + q < me) before += v[q];
                        }
                    }
                }
                int sum;
                const int pre = block_exclusive_scan(total, smem, &sum);
                if (d < bins) sbase[d] = run + pre + before;
                run += sum;
            }
        }
    }
    if (stamp) SCAL_STAMP(1);
    for (int d = threadIdx.x; d < bins; d += 256) cnt[0][d] = 0, cnt[1][d] = 0, cnt[2][d] = 0, cnt[3][d] = 0;
    __syncthreads();
    if (stamp) SCAL_STAMP(2);
    const int base = blockIdx.x * RS_TILE + w * (RS_ITEMS * 64);
    unsigned long long k[RS_ITEMS];
    int rk[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int e = base + j * 64 + l;
        const bool valid = e < n;
        k[j] = valid ? keys[e] : 0ull;
        const uint32_t d = static_cast<uint32_t>(k[j] >> shift) & (bins - 1);
        const uint64_t m = wave_match_bits(d, valid, width);
        int prev = 0;
        if (valid) prev = cnt[w][d];
        rk[j] = prev + __popcll(m & lanemask_lt());
        __builtin_amdgcn_wave_barrier();
        if (valid && (m & lanemask_lt()) == 0) cnt[w][d] = static_cast<unsigned short>(prev + __popcll(m));
        __builtin_amdgcn_wave_barrier();
    }
    if (stamp) SCAL_STAMP(3);
    __syncthreads();
    if (stamp) SCAL_STAMP(4);
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int e = base + j * 64 + l;
        if (e < n) {
            const uint32_t d = static_cast<uint32_t>(k[j] >> shift) & (bins - 1);
            int prew = 0;
            for (int ww = 0; ww < w; ++ww) prew += cnt[ww][d];
            const int pos = sbase[d] + prew + rk[j];
            okeys[pos] = k[j];
            ovals[pos] = vals[e];
        }
    }
    if (stamp) SCAL_STAMP(5);
}
SCAL_KERNEL(256, k_rs_scatter)
SCAL_DEFINE_STAMP_READER(scal_debug_stamps_radix)

// ---- hierarchical exclusive scan of the flattened histogram matrix (digit-major: exactly the scatter base of (digit, block)),
// used when the sort has too many blocks for every scatter workgroup to sum the matrix rows itself.  m = bins * nb.
constexpr int HS_TILE = 4096;
__device__ __forceinline__ void k_hs_reduce_body(const int* __restrict__ data, const int* __restrict__ d_n, int pass, int max_bits,
                                                     const int* __restrict__ d_used_bits, int* __restrict__ tile_sum) {
    int shift, width;
    if (!pass_plan(d_used_bits, max_bits, pass, shift, width)) return;
    __shared__ int smem[17];
    const int nb = (*d_n + RS_TILE - 1) / RS_TILE;
    const int m = (1 << width) * nb;
    const int base = blockIdx.x * HS_TILE;
    if (base >= m) return;
    int sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = base + q * 1024 + threadIdx.x;
        sum += i < m ? data[i] : 0;
    }
    int total;
    block_exclusive_scan(sum, smem, &total);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
SCAL_KERNEL(1024, k_hs_reduce)
__device__ __forceinline__ void k_hs_tiles_body(int* __restrict__ tile_sum, const int* __restrict__ d_n, int pass, int max_bits,
                                                    const int* __restrict__ d_used_bits) {
    int shift, width;
    if (!pass_plan(d_used_bits, max_bits, pass, shift, width)) return;
    __shared__ int smem[17];
    const int nb = (*d_n + RS_TILE - 1) / RS_TILE;
    const int nt = ((1 << width) * nb + HS_TILE - 1) / HS_TILE;
    const int per = (nt + 1023) / 1024;
    const int b0 = min(nt, static_cast<int>(threadIdx.x) * per), b1 = min(nt, b0 + per);
    int sum = 0;
    for (int i = b0; i < b1; ++i) sum += tile_sum[i];
    int total;
    int run = block_exclusive_scan(sum, smem, &total);
    for (int i = b0; i < b1; ++i) {
        const int v = tile_sum[i];
        tile_sum[i] = run;
        run += v;
    }
}
SCAL_KERNEL(1024, k_hs_tiles)
__device__ __forceinline__ void k_hs_apply_body(int* __restrict__ data, const int* __restrict__ d_n, int pass, int max_bits,
                                                    const int* __restrict__ d_used_bits, const int* __restrict__ tile_off) {
    int shift, width;
    if (!pass_plan(d_used_bits, max_bits, pass, shift, width)) return;
    __shared__ int smem[17];
    const int nb = (*d_n + RS_TILE - 1) / RS_TILE;
    const int m = (1 << width) * nb;
    const int base = blockIdx.x * HS_TILE;
    if (base >= m) return;
    // thread t owns elements base + 4t .. base + 4t + 3 (consecutive), so one block scan orders the whole tile
    int v[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = base + 4 * threadIdx.x + q;
        v[q] = i < m ? data[i] : 0;
        sum += v[q];
    }
    int total;
    int run = tile_off[blockIdx.x] + block_exclusive_scan(sum, smem, &total);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = base + 4 * threadIdx.x + q;
        if (i < m) data[i] = run;
        run += v[q];
    }
}
SCAL_KERNEL(1024, k_hs_apply)

int RadixSort::init(int capacity) {
    cap = capacity;
    SCAL_TRY(keys_alt.alloc(cap));
    SCAL_TRY(vals_alt.alloc(cap));
    SCAL_TRY(hist.alloc((size_t)BINS_MAX * div_up(cap, TILE) + BINS_MAX));
    SCAL_TRY(tile_sum.alloc(div_up(BINS_MAX * div_up(cap, TILE), HS_TILE) + 1));
    return SCAL_OK;
}

int RadixSort::sort(hipStream_t s, unsigned long long* keys, int* vals, const int* d_n, int n_bound, int max_bits, const int* d_used_bits,
                    SortedPairs* out) {
    unsigned long long* kb[2] = {keys, keys_alt.p};
    int* vb[2] = {vals, vals_alt.p};
    const int nb = max(1, div_up(min(cap, max(n_bound, 1)), TILE));
    const bool big = nb > 256;  // every scatter workgroup summing the matrix rows itself stops paying off
    const int nt = div_up(BINS_MAX * nb, HS_TILE);
    const int passes = rs_passes(max_bits);
    for (int pass = 0; pass < passes; ++pass) {
        const int in = pass & 1, o = in ^ 1;
        SCAL_LAUNCH(n_hist.c_str(), k_rs_hist, dim3(nb), dim3(256), 0, s, kb[in], d_n, pass, max_bits, d_used_bits, hist.p, big ? 1 : 0);
        if (big) {
            SCAL_LAUNCH("k_hs_reduce", k_hs_reduce, dim3(nt), dim3(1024), 0, s, hist.p, d_n, pass, max_bits, d_used_bits, tile_sum.p);
            SCAL_LAUNCH("k_hs_tiles", k_hs_tiles, dim3(1), dim3(1024), 0, s, tile_sum.p, d_n, pass, max_bits, d_used_bits);
            SCAL_LAUNCH("k_hs_apply", k_hs_apply, dim3(nt), dim3(1024), 0, s, hist.p, d_n, pass, max_bits, d_used_bits, tile_sum.p);
        }
        SCAL_LAUNCH(n_scatter.c_str(), k_rs_scatter, dim3(nb), dim3(256), 0, s, kb[in], vb[in], d_n, pass, max_bits, d_used_bits, hist.p, kb[o],
                         vb[o], big ? 1 : 0);
    }
    out->keys[0] = kb[0], out->keys[1] = kb[1];
    out->vals[0] = vb[0], out->vals[1] = vb[1];
    out->d_used_bits = d_used_bits;
    out->fixed_sel = d_used_bits ? -1 : (passes & 1);
    SCAL_HIP(hipGetLastError());
    return SCAL_OK;
}

void launch_scan_inplace(hipStream_t s, int* data, const int* d_n, int tile, int bins, int* d_total) {
    SCAL_LAUNCH("k_scan", k_scan, dim3(1), dim3(1024), 0, s, data, d_n, tile, bins, d_total);
}

}  // namespace scal
