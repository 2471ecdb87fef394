// Device-wide stable LSD radix sort of (uint64 key, int32 value) pairs and a small exclusive-scan helper.  Counts live on the
// device (`d_n`): grids are sized for the capacity and surplus blocks exit, so a whole kernel chain can be enqueued without a
// host round trip.  The number of key bits actually in use may also live on the device (`d_used_bits`): the digit width adapts
// to it - ceil(used / 10) passes of ceil(used / passes) bits each up to 30 bits, so a 25..27-bit voxel key takes three passes of
// nine bits and a 28..30-bit one three of ten; wider keys take ceil(used / 12) passes of up to 12 bits (31..36 bits: still three
// passes - slower ones, the scatter bases of 2048-4096 digits cost more - instead of a fourth launch pair that every sort of a
// 36-bit-capable caller would have to enqueue).  Passes beyond the plan return immediately and the result buffer is selected by
// parity on the device.  The host enqueues rs_passes(max_bits) passes.
#pragma once
#include "common.hpp"

namespace scal {

struct SortedPairs {
    // result is in (keys[sel], vals[sel]) with sel = passes_executed & 1; passes_executed = ceil(used_bits / DIGIT_MAX)
    unsigned long long* keys[2];
    int* vals[2];
    int fixed_sel;            // >= 0 when the number of executed passes is known on the host
    const int* d_used_bits;   // else: device word holding the used bits
};

struct RadixSort {
    static constexpr int DIGIT_MAX = 12;
    static constexpr int BINS_MAX = 1 << DIGIT_MAX;
    static constexpr int ITEMS = 8;
    static constexpr int TILE = 256 * ITEMS;  // elements per block
    int cap = 0;
    DevBuf<unsigned long long> keys_alt;
    DevBuf<int> vals_alt;
    DevBuf<int> hist;  // [nb][BINS_MAX] or [bins][nb]
    DevBuf<int> tile_sum;  // hierarchical scan of hist for large sorts
    // names under which the launches are timed (scal_prof_*): a caller's stage tag is appended so that bench.py can charge the
    // passes to the stage that ordered them (".C" stage C's stack filters, ".D" ScanContext's keyframe filter)
    std::string n_hist = "k_rs_hist", n_scatter = "k_rs_scatter";
    void set_tag(const char* tag) { n_hist = std::string("k_rs_hist") + tag, n_scatter = std::string("k_rs_scatter") + tag; }

    int init(int capacity);
    // Sorts the first *d_n pairs (*d_n <= n_bound, host-known) by bits [0, max_bits) of the key, ascending, stable.
    // If d_used_bits is given, only ceil(*d_used_bits / DIGIT_MAX) passes do work.
    int sort(hipStream_t s, unsigned long long* keys, int* vals, const int* d_n, int n_bound, int max_bits, const int* d_used_bits,
             SortedPairs* out);
};

// passes that do work for a key of `used` bits (host and device), and their digit width
__host__ __device__ __forceinline__ int rs_passes(int used) { return used <= 30 ? (used + 9) / 10 : (used + RadixSort::DIGIT_MAX - 1) / RadixSort::DIGIT_MAX; }
__device__ __forceinline__ void rs_plan(int used, int& passes, int& width) {
    passes = rs_passes(used);
    width = passes > 0 ? (used + passes - 1) / passes : 0;
}
__device__ __forceinline__ int sorted_sel(const SortedPairs& p) {
    if (p.fixed_sel >= 0) return p.fixed_sel;
    return rs_passes(*p.d_used_bits) & 1;
}

// in-place exclusive scan of data[0 .. m) with m = bins * ceil(*d_n / tile), by one block; total -> *d_total (may be null)
void launch_scan_inplace(hipStream_t s, int* data, const int* d_n, int tile, int bins, int* d_total);

}  // namespace scal
