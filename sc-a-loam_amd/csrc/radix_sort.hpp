// Device-wide stable LSD radix sort of (uint64 key, int32 value) pairs and a small exclusive-scan helper.
// Counts live on the device (`d_n`): grids are sized for the capacity and surplus blocks exit, so a whole
// kernel chain can be enqueued without a host round trip.
#pragma once
#include "common.hpp"

namespace scal {

struct RadixSort {
    static constexpr int ITEMS = 8;
    static constexpr int TILE = 256 * ITEMS;  // elements per block
    int cap = 0;
    DevBuf<unsigned long long> keys_alt;
    DevBuf<int> vals_alt;
    DevBuf<int> hist;  // [256][nb_cap]

    int init(int capacity);
    // Sorts the first *d_n pairs by bits [begin_bit, end_bit) of the key, ascending, stable.
    // The sorted pairs end up in (*out_keys, *out_vals), which alias either the inputs or the internal buffers.
    int sort(hipStream_t s, unsigned long long* keys, int* vals, const int* d_n, int begin_bit, int end_bit, unsigned long long** out_keys,
             int** out_vals);
};

// in-place exclusive scan of data[0 .. m) with m = bins * ceil(*d_n / tile), by one block; total -> *d_total (may be null)
void launch_scan_inplace(hipStream_t s, int* data, const int* d_n, int tile, int bins, int* d_total);

}  // namespace scal
