// Device-resident results of stage A, shared with the stages that consume them without a host round trip.
#pragma once
#include <hip/hip_runtime.h>

struct scal_features;

namespace scal {

struct FeatParams {
    float start_ori, end_ori;
    int first_idx, last_idx, flip_idx, empty, error, n_tied;
    int n_kept;
    int ring_count[64];
    int ring_off[65];
    int scan_start[64], scan_end[64];
    int n_sharp, n_less_sharp, n_flat, n_less_flat;
    int lf_ring_cnt[64];
    int lf_ring_off[65];
};

struct FeatDeviceView {
    const FeatParams* P;                 // counts live on the device
    const float *x, *y, *z, *i;          // ordered cloud, SoA, P->n_kept points
    const float *lfx, *lfy, *lfz, *lfi;  // lessFlat cloud, SoA, P->n_less_flat points
    const float *sharp_xyzi, *less_xyzi, *flat_xyzi;  // picked points, AoS xyzi
    const unsigned* box_parts;           // per-block bounding boxes of the ordered cloud (vox_bbox_block_store), n_box_parts blocks
    int n_box_parts;
    int cap;
    hipStream_t stream;
    int device;
    int n_scans;
    unsigned generation;                 // runs of the features context so far: a consumer that took its inputs from run g (a prefetch)
                                         // and comes back later checks that the context has not been run again in between
};

FeatDeviceView features_view(scal_features* c);
// A consumer that reads the view's buffers on ANOTHER stream calls this after enqueuing its last read: the next run of the
// features context waits for it before overwriting the buffers.
int features_note_reader(scal_features* c, hipStream_t consumer_stream);
// makes consumer_stream wait for the most recent run of the features context (one event per run, shared by all consumers)
int features_wait_done(scal_features* c, hipStream_t consumer_stream);

}  // namespace scal
