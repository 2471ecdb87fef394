/*
 * scaloam_hip.h — C-ABI of libscaloam_hip.so: the MI355X (gfx950) implementation of SC-A-LOAM's
 * data-parallel hot path.  The reference (swoonge/SC-A-LOAM) has no plugin/FFI interface: its stages
 * are welded into ROS node main()s.  Each entry point below is therefore a *call-site cut*: it replaces
 * the cited lines of the reference, and INTEGRATION.md shows the few lines a maintainer changes in each
 * node to call it.  Plain C: opaque handles, raw pointers and sizes, int status codes; nothing throws
 * or aborts across this boundary.
 *
 * Conventions
 *  - Every context owns one HIP stream, its device buffers and pinned staging.  All arrays passed in or
 *    out are caller-owned and only need to stay valid for the duration of the call.  `_device` variants
 *    take pointers into GPU memory of the context's device instead of host memory.
 *  - Points are `PointXYZI` as the reference uses them (include/aloam_velodyne/common.h:43):
 *    4 packed floats x,y,z,intensity (16 B) on the host side.  On the device everything is SoA.
 *  - A context is single-caller (like the reference's node threads) except the ScanContext context,
 *    which serialises insert/detect internally (the reference calls them from two threads with no
 *    common lock: laserPosegraphOptimization.cpp:633-642 vs :718).
 *  - Status: 0 ok, <0 SCAL_E_*; scal_last_error() gives a message for the calling thread.
 */
#ifndef SCALOAM_HIP_H
#define SCALOAM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
enum {
    SCAL_OK = 0,
    SCAL_E_ARG = -1,          /* bad argument */
    SCAL_E_LIDAR_TYPE = -2,   /* unknown lidar_type (reference: ROS_BREAK, scanRegistration.cpp:214-218) */
    SCAL_E_SCAN_LINE = -3,    /* scan_line not 16/32/64 (reference exits, scanRegistration.cpp:486-490) */
    SCAL_E_TOO_MANY = -4,     /* more points than max_points (reference scratch is 400000, scanRegistration.cpp:68-71) */
    SCAL_E_EMPTY = -5,        /* no point survives the NaN / range filters (reference would read points[0]) */
    SCAL_E_HIP = -6,          /* HIP runtime error */
    SCAL_E_NO_DEVICE = -7,    /* no gfx950 device / code object missing: the product never falls back to a CPU path */
    SCAL_E_CAPACITY = -8,     /* a device-side capacity was exceeded (segment longer than the LDS sort, map pool full) */
    SCAL_E_STATE = -9         /* call order violated (e.g. fetch before run) */
};
const char* scal_last_error(void);
/* number of visible HIP devices, -1 if the runtime cannot initialise */
int scal_device_count(void);
const char* scal_version(void);
/* 0 (default): the stages of a scan share one in-order stream per device (plus the optional side streams).  1: every stage
 * context created afterwards gets its own stream, so that consecutive scans overlap the way the reference's four ROS nodes do
 * (A of scan k+1 during C of scan k ...); hand-overs between contexts are ordered by events.  Four streams carry the work (A + stage
 * C's gather and corner filter; B + ScanContext's descriptor and search; C; one side stream for stage C's surf filter and
 * ScanContext's keyframe filter), split so that the four chains are about equally long.  Use two or more features contexts
 * alternately in that mode: a context's outputs are rewritten by its next run, which waits for all their readers. */
int scal_set_stream_mode(int mode);
/* Optional per-kernel timing (HIP events on the launching stream), used by bench.py's roofline leg.
 * No counterpart in the reference: its TicToc phase timers print nothing. */
int scal_prof_enable(int on);
int scal_prof_filter(const char* kernel_name); /* time only this kernel (NULL/"" = all instrumented kernels) */
int scal_prof_reset(void);
int scal_prof_read(const char* kernel_name, double* total_ms, long* count);
int scal_prof_names(char* buf, int cap);
/* development aid: while on (together with scal_prof_enable), every timed dispatch is also kept as (kernel, start ms, stop ms)
 * relative to one base event; scal_prof_timeline_dump writes them as CSV and clears the list */
int scal_prof_timeline(int on);
int scal_prof_timeline_dump(const char* path);

/* ------------------------------------------------------------------ ordering against the caller's own streams
 * Every `_device` entry point reads (and writes) the caller's device memory on the CONTEXT'S stream, which is not ordered against
 * any stream of the caller: device inputs must be complete before the call, unless the caller orders the context's stream behind
 * its producer itself.  These accessors return that stream (a hipStream_t) for exactly that: hipStreamWaitEvent(ctx_stream, ev)
 * in front of the call, hipEventRecord(ev, ctx_stream) behind it (torch: torch.cuda.ExternalStream(ptr).wait_stream(...)).
 * Typical producer: an RCCL collective on the caller's stream (sharded map filter, sharded ScanContext exchange). */
typedef struct scal_features scal_features_t;
typedef struct scal_voxel scal_voxel_t;
typedef struct scal_sc scal_sc_t;
typedef struct scal_map scal_map_t;
typedef struct scal_odom scal_odom_t;
void* scal_features_stream(scal_features_t* ctx);
void* scal_voxel_stream(scal_voxel_t* ctx);
void* scal_sc_stream(scal_sc_t* ctx); /* the stream of descriptors, keys, database and searches (in stream mode 1 the keyframe filter of
                                       * the *_features entry points runs on a second, internal stream that this one waits for) */
void* scal_map_stream(scal_map_t* ctx);
void* scal_odom_stream(scal_odom_t* ctx);

/* ------------------------------------------------------------------ stage A: feature extraction
 * Replaces laserCloudHandler, src/scanRegistration.cpp:134-421 (NaN/range filter, ring id + relative
 * time, ring-major reorder, curvature, per-(ring,sixth) sort + greedy sharp/lessSharp/flat picks,
 * lessFlat + per-ring 0.2 m voxel grid).  The five output clouds are what the node publishes at :426-454. */
enum { SCAL_VLP16 = 0, SCAL_HDL32 = 1, SCAL_HDL64 = 2, SCAL_OS1_64 = 3 };

typedef struct {
    int lidar_type;        /* SCAL_*  (rosparam lidar_type, scanRegistration.cpp:481) */
    int n_scans;           /* rosparam scan_line (:480) */
    double minimum_range;  /* rosparam minimum_range (:482) */
    int max_points;        /* capacity; <= 400000 */
    int float_math;        /* 0: atan/sqrt of :168 promote to double (the pinned GCC 5 toolchain); 1: float overloads */
    int check_finite;      /* 1: drop non-finite points (pcl::removeNaNFromPointCloud on a non-dense cloud, :138) */
    int device;            /* HIP device ordinal */
} scal_features_config;

typedef struct {
    /* caller-owned host arrays; capacities in points unless noted; NULL = not wanted */
    float* cloud;      /* [max_points][4] ordered cloud, intensity = scanID + 0.1*relTime (/velodyne_cloud_2) */
    int* src_index;    /* [max_points] input index of every ordered point */
    float* curvature;  /* [max_points] cloudCurvature */
    int* label;        /* [max_points] cloudLabel */
    int* ring_start;   /* [n_scans] scanStartInd */
    int* ring_end;     /* [n_scans] scanEndInd */
    int* sharp;        /* [2*6*n_scans]  indices into `cloud`, reference emission order */
    int* less_sharp;   /* [20*6*n_scans] */
    int* flat;         /* [4*6*n_scans] */
    float* less_flat;  /* [max_points][4] downsampled lessFlat cloud */
    /* filled by the call */
    int n_kept, n_sharp, n_less_sharp, n_flat, n_less_flat;
    int n_tied_segments; /* segments whose sort met equal curvatures (std::sort order is unspecified there) */
} scal_features_out;

int scal_features_create(const scal_features_config* cfg, scal_features_t** ctx);
void scal_features_destroy(scal_features_t* ctx);
/* host input: xyz at xyz + i*stride_bytes (PointCloud2 layout). Synchronous. */
int scal_features_run(scal_features_t* ctx, const void* xyz, int n, int stride_bytes, scal_features_out* out);
/* host input, asynchronous: the scan is copied into pinned staging (the caller's buffer is free on return), uploaded and
 * processed on the context's stream; results stay on the GPU as with scal_features_run_device (the PointCloud2 callback of
 * scanRegistration.cpp:116 when the later stages consume the device-resident results) */
int scal_features_enqueue_host(scal_features_t* ctx, const void* xyz, int n, int stride_bytes);
/* device input (float*, stride in floats), asynchronous on the context's stream; results stay on the GPU
 * for scal_odom_step_features / scal_map_step_features / scal_sc_insert_features. */
int scal_features_run_device(scal_features_t* ctx, const float* d_xyz, int n, int stride_floats);
/* wait for the last run and copy results to the host */
int scal_features_fetch(scal_features_t* ctx, scal_features_out* out);
int scal_features_sync(scal_features_t* ctx);

/* ------------------------------------------------------------------ voxel grid
 * Replaces pcl::VoxelGrid<PointXYZI>::filter at scanRegistration.cpp:414-418,
 * laserMapping.cpp:543-551, :793-801 and laserPosegraphOptimization.cpp:629-631. */
int scal_voxel_create(int max_points, int device, scal_voxel_t** ctx);
void scal_voxel_destroy(scal_voxel_t* ctx);
int scal_voxel_downsample(scal_voxel_t* ctx, const float* xyzi, int n, float leaf, float* out_xyzi, int* n_out);
/* the same with the cloud and the output (room for n records; may not overlap the input) in device memory, e.g. between
 * scal_mapmerge_device_points and scal_icp_align_device: loopFindNearKeyframesCloud's VoxelGrid, :491-492; waits, *n_out on the host.
 * The input must be complete (or scal_voxel_stream ordered behind its producer) before the call. */
int scal_voxel_downsample_device(scal_voxel_t* ctx, const float* d_xyzi, int n, float leaf, float* d_out_xyzi, int* n_out);

/* ------------------------------------------------------------------ stage D: ScanContext
 * Replaces SCManager (include/scancontext/Scancontext.h:57-123, Scancontext.cpp), call sites
 * laserPosegraphOptimization.cpp:639 (insert) and :718 (detect). */
typedef struct {
    double max_radius;  /* PC_MAX_RADIUS, setMaximumRadius (Scancontext.cpp:267-270) */
    double dist_thres;  /* SC_DIST_THRES, setSCdistThres (:262-265) */
    int max_keyframes;  /* database capacity on this device */
    int float_math;     /* atan overload at Scancontext.cpp:26-35, as in scal_features_config */
    int device;
    /* sharding (multi-GPU): this context stores keyframes i with i % n_shards == shard; 1/0 = everything */
    int n_shards, shard;
    /* n > 0: run this context on the device's side stream n (1..5) so that insert/detect overlap with stages B and C of the
     * same scan; ordering against the features context (scal_sc_insert_features / scal_sc_make_features) is kept with events.
     * Two contexts on different lanes do not queue behind each other (descriptor builder: 1, sharded database: 5). */
    int side_stream;
} scal_sc_config;

typedef struct {
    int loop_id;      /* -1: no loop (Scancontext.cpp:338, :406-408) */
    float yaw_rad;    /* deg2rad(shift * 6 deg) (:422) */
    double min_dist;  /* best SC distance among the candidates */
    int nn_idx;       /* its keyframe index */
    int nn_shift;     /* its column shift */
    int cand_idx[3];  /* ring-key KNN candidates, ascending key distance */
    float cand_keydist[3];
    double cand_scdist[3];
    int cand_shift[3];
} scal_sc_result;

int scal_sc_create(const scal_sc_config* cfg, scal_sc_t** ctx);
void scal_sc_destroy(scal_sc_t* ctx);
int scal_sc_size(scal_sc_t* ctx);
/* makeAndSaveScancontextAndKeys(cloud) — cloud already downsampled by the caller as the reference does */
int scal_sc_insert_cloud(scal_sc_t* ctx, const float* xyzi, int n);
int scal_sc_insert_cloud_device(scal_sc_t* ctx, const float* d_x, const float* d_y, const float* d_z, const int* d_n, int n_max);
/* saveScancontextAndKeys(desc): 20x60 doubles, column-major */
int scal_sc_insert_descriptor(scal_sc_t* ctx, const double* desc_colmajor);
int scal_sc_get_descriptor(scal_sc_t* ctx, int idx, double* desc_colmajor, float* ringkey20);
/* keyframe cloud of a features context on the same device (its ordered full-resolution cloud, which is what
 * /velodyne_cloud_registered_local carries) -> VoxelGrid 0.4 m (laserPosegraphOptimization.cpp:629-631, :890-891)
 * -> makeAndSaveScancontextAndKeys (:639), nothing leaves the GPU. */
int scal_sc_insert_features(scal_sc_t* ctx, scal_features_t* feat);
/* same front end but the 20x60 descriptor is only written to d_desc (device memory, 1200 doubles, column-major)
 * and NOT inserted: the sharded search exchanges descriptors first. */
int scal_sc_make_features(scal_sc_t* ctx, scal_features_t* feat, double* d_desc);
/* the same without waiting: d_desc is valid after scal_sc_sync, or after scal_sc_wait_descriptor, which waits for the oldest
 * queued descriptor only (up to four may be in flight) */
int scal_sc_make_features_enqueue(scal_sc_t* ctx, scal_features_t* feat, double* d_desc);
int scal_sc_wait_descriptor(scal_sc_t* ctx);
int scal_sc_insert_descriptor_device(scal_sc_t* ctx, const double* d_desc_colmajor);
/* n (<= 64) device-resident descriptors in global order - the ranks' descriptors of one step after the all-gather - in one
 * launch and without a host synchronisation; every shard keeps the ones it owns */
int scal_sc_insert_descriptors_device(scal_sc_t* ctx, const double* d_descs_colmajor, int n);
/* waits for everything queued on the context's stream (before a collective on another stream reads its device outputs) */
int scal_sc_sync(scal_sc_t* ctx);
/* makeScancontext only (no insert) */
int scal_sc_make_descriptor(scal_sc_t* ctx, const float* xyzi, int n, double* desc_colmajor);
/* detectLoopClosureID(): query = newest keyframe; reproduces the >=31 gate, the 30-query tree period,
 * the exclusion of the newest 30 keys and the 3-candidate / 7-shift search (Scancontext.cpp:336-427). */
int scal_sc_detect(scal_sc_t* ctx, scal_sc_result* res);
/* the same in two halves: enqueue launches the search for the newest keyframe and returns, collect waits for the oldest search
 * not yet collected (its own event: nothing queued on the stream behind it is waited for).  scal_sc_detect = enqueue + collect.
 * Up to four searches may be in flight, each for the keyframe that was newest when it was enqueued. */
int scal_sc_detect_enqueue(scal_sc_t* ctx);
int scal_sc_detect_collect(scal_sc_t* ctx, scal_sc_result* res);
/* distanceBtnScanContext for descriptor pairs already in the database */
int scal_sc_distance_pairs(scal_sc_t* ctx, const int* idx_a, const int* idx_b, int n_pairs, double* dist, int* shift);
/* dense block of the pair grid (SURVEY.md 8d "D dense"; distDirectSC Scancontext.cpp:83-110 over every pair): queries [q0,q1) x
 * database [d0,d1), row-major [q][d].  mode 0 = the reference's 7-shift search around the sector-key alignment
 * (distanceBtnScanContext :113-148); 1 = exhaustive 60 shifts, column sums in the reference's order on the vector ALUs;
 * 2 = exhaustive 60 shifts on the matrix cores (v_mfma_f64_16x16x4_f64 over unit columns: same minimum and shift, distances
 * equal to mode 1 within a few ulp - the summation order differs); 3 = the same product on v_mfma_f32_16x16x4_f32 with the
 * unit columns rounded to f32 (twice the rate; distances within 2e-6 of mode 2 - inside the 1e-5 the path is held to - and the
 * shift can differ where two shifts are closer than that).  Unsharded contexts only. */
int scal_sc_distance_matrix(scal_sc_t* ctx, int q0, int q1, int d0, int d1, int mode, double* dist, int* shift);
/* the same with DEVICE outputs (nq*nd doubles / ints), enqueued on the context's stream: scal_sc_sync() before reading them */
int scal_sc_distance_matrix_device(scal_sc_t* ctx, int q0, int q1, int d0, int d1, int mode, double* d_dist, int* d_shift);
/* Batch loop search over a stored session (offline loop mining; the exhaustive counterpart of detectLoopClosureID's ring-key
 * prefilter + 7-shift search, Scancontext.cpp:336-427): for every query keyframe q in [q0, q1) the k (<= 16) smallest distances
 * among the keyframes d < q - exclude_recent (NUM_EXCLUDE_RECENT = 30, Scancontext.h:92), from the dense distance block of `mode`
 * (2 = all 60 shifts on the f64 matrix cores) with the top-k taken on the device.  Outputs [q1 - q0][k]: keyframe index (-1 =
 * fewer than k eligible), distance, column shift; ascending distance, ties to the lower index.  Unsharded contexts only. */
int scal_sc_batch_loop_search(scal_sc_t* ctx, int q0, int q1, int exclude_recent, int k, int mode, int* idx, double* dist, int* shift);
/* sharded search pieces (one context per GPU): local top-3 for the newest GLOBAL key, then a merge of the
 * gathered per-shard records (the all-gather itself is the caller's: RCCL via torch.distributed). */
typedef struct {
    float key_dist;
    int idx; /* global keyframe index, -1 = empty */
    double sc_dist;
    int shift;
    int pad;
} scal_sc_cand;
int scal_sc_shard_query(scal_sc_t* ctx, const double* query_desc_colmajor, int global_size_at_rebuild, scal_sc_cand out[3]);
/* batched form on device pointers: d_queries [nq][1200] doubles, d_out [nq][3] records; synchronous */
int scal_sc_shard_query_device(scal_sc_t* ctx, const double* d_queries, int nq, int global_size_at_rebuild, scal_sc_cand* d_out);
/* the same for a batch of nq (<= 64) queries, query q searched against the tree of size limits[q] (host array); three launches,
 * the 3 * nq records stay in device memory (d_out) and nothing is synchronised: follow with scal_sc_sync */
int scal_sc_shard_query_batch_device(scal_sc_t* ctx, const double* d_queries, int nq, const int* limits, scal_sc_cand* d_out);
int scal_sc_merge_candidates(const scal_sc_cand* gathered, int n_records, double dist_thres, scal_sc_result* res);

/* ------------------------------------------------------------------ stage C: scan-to-map
 * Replaces process(), src/laserMapping.cpp:310-802 and :845-849. */
typedef struct {
    float line_res, plane_res; /* rosparams mapping_line_resolution / mapping_plane_resolution (:915-916) */
    int max_scan_points;       /* capacity of one full-resolution scan */
    int max_map_points;        /* capacity of the rolling 21x21x11 cube window, per feature class */
    int device;
} scal_map_config;

typedef struct {
    int n_corner_stack, n_surf_stack, n_corner_map, n_surf_map;
    int n_edge[2], n_plane[2];      /* residual blocks per outer iteration (:563) */
    int lm_iters[2], lm_success[2]; /* LM iterations / accepted steps per outer iteration */
    double cost_init[2], cost_final[2];
    int solved;                     /* 0: map too small (:555, :731-734), pose = prior */
    int n_map_corner_total, n_map_surf_total;
    int insert_path;                /* how :738-802 ran: 0 = full sort of the map pool, 1 = merge of the scan's points into the sorted map */
} scal_map_stats;

int scal_map_create(const scal_map_config* cfg, scal_map_t** ctx);
void scal_map_destroy(scal_map_t* ctx);
/* one process() pass.  corner_last / surf_last / full_res: PointXYZI host arrays (full_res may be NULL).
 * q_wodom (x,y,z,w), t_wodom: /laser_odom_to_init pose.  Outputs: q_w_curr, t_w_curr (/aft_mapped_to_init),
 * registered (full_res transformed, :845-849; may be NULL). */
int scal_map_step(scal_map_t* ctx, const float* corner_last, int n_corner, const float* surf_last, int n_surf,
                  const float* full_res, int n_full, const double* q_wodom, const double* t_wodom, double* q_w_curr,
                  double* t_w_curr, float* registered, scal_map_stats* stats);
/* same, inputs taken from a features context on the same device (lessSharp / lessFlat / ordered cloud) */
int scal_map_step_features(scal_map_t* ctx, scal_features_t* feat, const double* q_wodom, const double* t_wodom,
                           double* q_w_curr, double* t_w_curr, scal_map_stats* stats);
/* Optional: start the pose-independent part of the next scal_map_step_features(ctx, feat, ...) - input gather and the stack
 * downsample (:543-551) - on the device's side stream, so that it overlaps with stage B.  Returns immediately.  Up to three
 * prefetches may be queued ahead of their steps (rotating input sets); steps consume them in order.  `feat` must not be run again
 * before its step has been enqueued: the step would mix the prefetched inputs of one scan with the full-resolution cloud of the
 * next; the step entry points check the features context's run counter and return SCAL_E_STATE in that case. */
int scal_map_prefetch_features(scal_map_t* ctx, scal_features_t* feat);
/* The same in two halves, for a host that issues them from different threads (scal_pipeline does): begin queues the input gather and
 * the corner stack filter on the features context's stream, right behind stage A; finish queues the surf stack filter on the side
 * stream.  Halves are finished in the order they were begun; a step consumes its prefetch once both halves have been queued. */
int scal_map_prefetch_begin(scal_map_t* ctx, scal_features_t* feat);
int scal_map_prefetch_finish(scal_map_t* ctx, scal_features_t* feat);
/* scal_map_step_features in two halves.  enqueue queues the whole pass; collect returns the oldest uncollected pose as soon as it
 * is on the host, while the map insertion (:738-802) and the registration (:845-849) still run behind it.  Up to four steps may
 * be queued before the first is collected: transformAssociateToMap / transformUpdate (:143-153), the rolling-window decision
 * (:313-508) and the map sizes live on the device, so a step is queued right behind the previous one with nothing read back.
 * Such a step is speculative: it assumes the cube window of the previous step and the merge insert; when that does not hold the
 * device stops the chain, and collect / finish redo the step on the general path (window shift, full-sort insertion) and replay
 * the steps queued behind it - results are identical either way.  A features context handed to enqueue must not be run again
 * before its step has been collected.  scal_map_export and scal_map_finish wait for all queued insertions (and report a
 * capacity error of an insertion).  The map sizes in the statistics of collect are those before this scan's insertion,
 * insert_path is -1. */
int scal_map_enqueue_features(scal_map_t* ctx, scal_features_t* feat, const double* q_wodom, const double* t_wodom);
int scal_map_collect(scal_map_t* ctx, double* q_w_curr, double* t_w_curr, scal_map_stats* stats);
int scal_map_finish(scal_map_t* ctx);
/* current map points of the 5x5x3 window (laserCloudCornerFromMap / SurfFromMap content); returns count */
int scal_map_export(scal_map_t* ctx, int which /*0 corner, 1 surf*/, float* out_xyzi, int cap);
/* every point of the 21x21x11 cube grid, one feature class: the content of /laser_cloud_map (laserMapping.cpp:824-837 adds the corner
 * and surf clouds of all 4,851 cubes every 20 frames); out_xyzi NULL or cap 0: only the count.  Returns the number of points written. */
int scal_map_export_all(scal_map_t* ctx, int which /*0 corner, 1 surf*/, float* out_xyzi, int cap);
int scal_map_get_wmap_wodom(scal_map_t* ctx, double* q_xyzw, double* t);
/* enable (default) / disable the merge insert; both give identical maps, the switch exists for tests and measurements */
int scal_map_set_merge_insert(scal_map_t* ctx, int enable);
/* how the steps ran so far: out4 = {queued speculatively, run on the general path, redone from the start after a stopped chain
 * (window moved), insertion redone with the full sort after a stopped chain}; for tests and measurements */
int scal_map_get_path_counters(scal_map_t* ctx, int* out4);
/* test hook: poll budget of the LM solve's partial-sum exchange between its workgroups (default 2^22).  With 0 every solve of this library's stage C gives up at
 * its first exchange: collect / step return SCAL_E_HIP ("LM solve abandoned"), the pose of that step is its prior, and the context
 * must keep working afterwards.  Exercises the bounded-wait path that would otherwise need a machine hogged by another process. */
int scal_map_debug_set_lm_polls(scal_map_t* ctx, int polls);
/* test hook: entries per 1 m cell of the one-launch neighbour-grid build of the queued steps (default: the number of filter voxels
 * that can intersect a cell - 27 corner / 8 surf at the reference's 0.4 / 0.8 m; values above that are clamped).  A cell that
 * receives more points stops the queued chain, and the step is redone with the general three-launch build: with 1 / 1 every
 * queued step takes that route (path counter 3 counts them) and the poses must not change. */
int scal_map_debug_set_grid_cap(scal_map_t* ctx, int cap_corner, int cap_surf);
/* 1 (default): every enqueue first looks (without waiting) whether queued steps have finished or stopped; 0: a stopped chain is
 * only noticed by collect / finish, so that steps really get queued behind it (test switch for the replay path) */
int scal_map_set_poll(scal_map_t* ctx, int enable);

/* ---- Ceres-adapter mode of stage C: the host keeps ceres::Problem / ceres::Solve exactly as
 * laserMapping.cpp:563-728 builds them; the device does what surrounds the solve and evaluates the residual blocks in batches.
 *   scal_map_adapter_begin   :310-560  transformAssociateToMap, window shift, submap (cell grids), stack downsample; returns the prior pose
 *   per outer iteration (:563):
 *     scal_map_associate     :578-688  kNN + PCA / plane fit at the caller's current pose -> residual blocks, in the order the reference adds
 *                                      them (corner stack points, then surf stack points); n_residuals = 3 per edge block + 1 per plane block
 *     scal_map_get_blocks              the blocks as records (to construct the reference's own LidarEdgeFactor / LidarPlaneNormFactor objects), or
 *     scal_map_eval_blocks             residuals [n_residuals] and ambient Jacobians [n_residuals][7] (qx qy qz qw tx ty tz, row-major) of all
 *                                      blocks at x7, without loss function or local parameterisation (Ceres applies its HuberLoss(0.1) and
 *                                      EigenQuaternionParameterization itself): what a ceres::EvaluationCallback + thin per-block CostFunction serves from
 *   scal_map_adapter_finish  :735-802, :845-849  transformUpdate with the solved pose, map insertion + re-filter, registration
 * kind: 0 LidarEdgeFactor(curr_point = cp, last_point_a = pa, last_point_b = pb, s = 1), 2 LidarPlaneNormFactor(curr_point = cp,
 * plane_unit_norm = pa, negative_OA_dot_norm = pb[0]); (1 = LidarPlaneFactor(cp, j = pa, unit normal = pb) in the odometry's blocks). */
typedef struct {
    int kind, pad;
    double cp[3], pa[3], pb[3];
} scal_block;
int scal_map_adapter_begin(scal_map_t* ctx, const float* corner_last, int n_corner, const float* surf_last, int n_surf, const float* full_res,
                           int n_full, const double* q_wodom, const double* t_wodom, double* q_w_curr, double* t_w_curr);
int scal_map_associate(scal_map_t* ctx, const double* q_w_curr, const double* t_w_curr, int* n_blocks, int* n_residuals);
int scal_map_get_blocks(scal_map_t* ctx, scal_block* out, int cap); /* returns the number of records written */
int scal_map_eval_blocks(scal_map_t* ctx, const double* x7, int want_jac, double* residuals, double* jacobians);
int scal_map_adapter_finish(scal_map_t* ctx, const double* q_w_curr, const double* t_w_curr, float* registered, scal_map_stats* stats);

/* ------------------------------------------------------------------ stage B: scan-to-scan odometry
 * Replaces the main loop body of src/laserOdometry.cpp:267-291, :299-506, :554-568. */
typedef struct {
    int max_points; /* capacity of the lessFlat cloud */
    int device;
} scal_odom_config;
typedef struct {
    int n_edge[2], n_plane[2];
    int lm_iters[2], lm_success[2];
    double cost_init[2], cost_final[2];
} scal_odom_stats;
int scal_odom_create(const scal_odom_config* cfg, scal_odom_t** ctx);
void scal_odom_destroy(scal_odom_t* ctx);
int scal_odom_step(scal_odom_t* ctx, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp,
                   const float* flat, int n_flat, const float* less_flat, int n_less_flat, double* q_last_curr,
                   double* t_last_curr, double* q_w_curr, double* t_w_curr, scal_odom_stats* stats);
int scal_odom_step_features(scal_odom_t* ctx, scal_features_t* feat, double* q_last_curr, double* t_last_curr,
                            double* q_w_curr, double* t_w_curr, scal_odom_stats* stats);
/* the same in two halves (= enqueue + collect): the caller can queue other work while the step runs.  Up to four steps may be
 * queued before the first is collected (a step only needs its predecessor's device state: the last clouds and the pose increment
 * of :97-101); collect returns them in order and integrates the pose (:504-505) on the host. */
int scal_odom_enqueue_features(scal_odom_t* ctx, scal_features_t* feat);
int scal_odom_collect(scal_odom_t* ctx, double* q_last_curr, double* t_last_curr, double* q_w_curr, double* t_w_curr,
                      scal_odom_stats* stats);

/* Ceres-adapter mode of stage B (laserOdometry.cpp:278-501 with the host's own ceres::Problem / ceres::Solve), the counterpart of
 * scal_map_adapter_*: begin uploads the four clouds and returns the initial guess (para_q / para_t persist across scans, :97-101)
 * and whether this frame is solved at all (the first one is not, :267-271); per outer iteration scal_odom_associate runs the
 * correspondence search of :299-483 at the solver's current increment, scal_odom_get_blocks / scal_odom_eval_blocks are as for
 * stage C (kind 0 LidarEdgeFactor, kind 1 LidarPlaneFactor given as (j, unit normal)); finish integrates the pose (:504-505) and
 * hands the clouds over (:554-568). */
int scal_odom_adapter_begin(scal_odom_t* ctx, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp, const float* flat,
                            int n_flat, const float* less_flat, int n_less_flat, double* q_last_curr, double* t_last_curr, int* need_solve);
int scal_odom_associate(scal_odom_t* ctx, const double* q_last_curr, const double* t_last_curr, int* n_blocks, int* n_residuals);
int scal_odom_get_blocks(scal_odom_t* ctx, scal_block* out, int cap);
int scal_odom_eval_blocks(scal_odom_t* ctx, const double* x7, int want_jac, double* residuals, double* jacobians);
int scal_odom_adapter_finish(scal_odom_t* ctx, const double* q_last_curr, const double* t_last_curr, double* q_w_curr, double* t_w_curr);

/* ------------------------------------------------------------------ offline dense map merge
 * Replaces the loop body of utils/python/makeMergedMap.py:83-133: keyframe cloud x SE(3) pose -> global frame (f64), removal of
 * points whose LOCAL range is <= near_thres (:109-116, 2 m in the script), concatenation in keyframe order, f32 xyzi out
 * (:145-147).  In-process analogues: local2global (laserPosegraphOptimization.cpp:338-359), transformPointCloud (:446-470). */
typedef struct {
    long long max_points;  /* capacity of the merged map */
    int max_frame_points;  /* largest keyframe given to scal_mapmerge_add */
    int device;
} scal_mapmerge_config;
typedef struct scal_mapmerge scal_mapmerge_t;
int scal_mapmerge_create(const scal_mapmerge_config* cfg, scal_mapmerge_t** ctx);
void scal_mapmerge_destroy(scal_mapmerge_t* ctx);
int scal_mapmerge_reset(scal_mapmerge_t* ctx);
/* one keyframe: xyzi host records, pose12 = one line of optimized_poses.txt (row-major top 3x4 of the SE(3) matrix, :48-56) */
int scal_mapmerge_add(scal_mapmerge_t* ctx, const float* xyzi, int n, const double* pose12, double near_thres);
/* n_frames keyframes already in device memory, back to back: frame f = records [offsets[f], offsets[f+1]) (host arrays) */
int scal_mapmerge_add_batch_device(scal_mapmerge_t* ctx, const float* d_xyzi, const int* offsets, const double* poses12,
                                   int n_frames, double near_thres);
long long scal_mapmerge_size(scal_mapmerge_t* ctx);                                 /* points merged so far (waits), <0 = error */
int scal_mapmerge_download(scal_mapmerge_t* ctx, float* out_xyzi, long long cap_points);
/* pubMap's VoxelGrid over the merged map (laserPosegraphOptimization.cpp:810-834, leaf = mapviz_filter_size): *n_out centroids
 * in PCL order, the first cap_points of them written to out_xyzi (host).  The merged map itself is kept. */
int scal_mapmerge_downsample(scal_mapmerge_t* ctx, float leaf, float* out_xyzi, long long cap_points, long long* n_out);
const float* scal_mapmerge_device_points(scal_mapmerge_t* ctx);                     /* the merged xyzi records in device memory */

/* ------------------------------------------------------------------ loop-closure verification ICP
 * Replaces the pcl::IterativeClosestPoint call of doICPVirtualRelative, laserPosegraphOptimization.cpp:518-535.  The caller
 * builds the two clouds as the reference does (:504-507: keyframes moved by one root pose, VoxelGrid 0.4) with scal_mapmerge_*
 * and scal_voxel_*, and applies the acceptance test of :532 (converged && fitness <= 0.3) to the result. */
typedef struct {
    double max_corr_dist;           /* setMaxCorrespondenceDistance, 150 (:520) */
    double transformation_epsilon;  /* setTransformationEpsilon, 1e-6 (:522) */
    double fitness_epsilon;         /* setEuclideanFitnessEpsilon, 1e-6 (:523) */
    int max_iterations;             /* setMaximumIterations, 100 (:521) */
    int max_source, max_target;     /* capacities */
    int device;
} scal_icp_config;
typedef struct {
    int converged;          /* icp.hasConverged() */
    int iterations;
    int state;              /* 1 iteration cap, 2 transformation epsilon, 3 absolute MSE, 4 relative MSE, 5 too few correspondences */
    int n_correspondences;  /* of the last iteration */
    double fitness;         /* icp.getFitnessScore() */
    double T[16];           /* icp.getFinalTransformation(), row-major, f32 values */
} scal_icp_result;
typedef struct scal_icp scal_icp_t;
int scal_icp_create(const scal_icp_config* cfg, scal_icp_t** ctx);
void scal_icp_destroy(scal_icp_t* ctx);
/* icp.setInputSource(src); icp.setInputTarget(tgt); icp.align(): xyzi host records */
int scal_icp_align(scal_icp_t* ctx, const float* src_xyzi, int n_src, const float* tgt_xyzi, int n_tgt, scal_icp_result* res);
/* the same with both clouds already in HBM (16-byte xyzi records, e.g. scal_mapmerge_device_points / a device voxel output): the
 * source is copied into the context, the target is read in place and must stay valid until the call returns */
int scal_icp_align_device(scal_icp_t* ctx, const float* d_src_xyzi, int n_src, const float* d_tgt_xyzi, int n_tgt, scal_icp_result* res);
/* nearest-neighbour search: 1 (default) = cell grid over the target, exact, with the dense sweep for queries it cannot settle;
 * 0 = dense sweep only.  Both stand in for pcl::KdTreeFLANN::nearestKSearch(k=1) (pcl/registration/impl/icp.hpp) and give the
 * same correspondences, lowest target index on equal distances. */
int scal_icp_set_search(scal_icp_t* ctx, int mode);

/* ------------------------------------------------------------------ the four stages as ONE pipelined object
 * The reference runs scanRegistration, laserOdometry, laserMapping and laserPosegraphOptimization as four processes that work on
 * different scans at the same time (node main()s: scanRegistration.cpp:475-517, laserOdometry.cpp:186-600, laserMapping.cpp:909-952,
 * laserPosegraphOptimization.cpp:874-906; hand-over by ROS topics with queue size 100).  scal_pipeline is that arrangement inside one
 * process and one GPU: it owns a ring of features contexts, one odometry, one mapping and one ScanContext context, puts each stage on
 * its own stream (scal_set_stream_mode(1) semantics) and keeps the schedule - which scan each stage may start, when a features
 * context may be rewritten, how many stage-C steps are queued on the device - on five internal host threads (stage A + the first half of
 * stage C's prefetch; its second half + ScanContext's filter, insert and search; stage B; stage C; ScanContext's answers).  Scans go in with push, poses come out IN ORDER with pop; every scan passes through
 * A -> B -> C and A -> D with the reference's data dependencies, and the poses are bit-identical to calling the four stage APIs one
 * scan at a time.  A C++ host (host/replay_main.cpp) and bench.py drive the same object. */
enum {
    SCAL_PIPE_SC_OFF = 0,        /* stages A-C only */
    SCAL_PIPE_SC_EVERY_SCAN = 1, /* makeAndSaveScancontextAndKeys + detectLoopClosureID for every scan (BASELINE config #2's "SC loop search" per scan) */
    SCAL_PIPE_SC_DESCRIPTOR = 2  /* only build each scan's descriptor into device memory (sharded search: the caller exchanges it over RCCL) */
};
typedef struct {
    int lidar_type, n_scans;      /* as scal_features_config */
    double minimum_range;
    int max_points;               /* capacity of one scan */
    int float_math, check_finite;
    float line_res, plane_res;    /* as scal_map_config */
    int max_map_points;
    double sc_max_radius, sc_dist_thres;
    int sc_max_keyframes;
    int sc_mode;                  /* SCAL_PIPE_SC_* */
    int device;
    int ring;                     /* features contexts used in turn: 4..16, 0 = 6 */
    int depth;                    /* stage-C steps queued on the device and not collected: 1..3, 0 = 2 */
    double* d_desc_ring;          /* sc_mode 2: caller-owned device memory for ring x 1200 doubles, scan k's descriptor at slot k % ring
                                   * (NULL: allocated by the pipeline) - so that an RCCL all-gather can read it in place */
} scal_pipeline_config;
typedef struct {
    long long seq;                        /* 0, 1, 2 ... in push order */
    double q_w_curr[4], t_w_curr[3];      /* /aft_mapped_to_init (laserMapping.cpp:861-876) */
    double q_odom[4], t_odom[3];          /* /laser_odom_to_init (laserOdometry.cpp:508-522) */
    scal_odom_stats odom;
    scal_map_stats map;
    int have_loop;                        /* sc_mode 1: `loop` is detectLoopClosureID's answer with this scan as the query */
    scal_sc_result loop;
    const double* d_descriptor;           /* sc_mode 2: this scan's 20x60 descriptor in device memory (valid until `ring` more scans were pushed) */
} scal_pipeline_result;
typedef struct scal_pipeline scal_pipeline_t;
int scal_pipeline_create(const scal_pipeline_config* cfg, scal_pipeline_t** p);
void scal_pipeline_destroy(scal_pipeline_t* p);
/* One scan, xyz in device memory (stride in floats); returns as soon as the scan is registered - the buffer must stay valid until the
 * scan's result has been popped.  Blocks only while `ring` scans are in flight; SCAL_E_STATE when 32 results wait to be popped. */
int scal_pipeline_push_device(scal_pipeline_t* p, const float* d_xyz, int n, int stride_floats);
/* The same from host memory (PointCloud2 layout, as scal_features_run): the scan is copied into pinned staging and uploaded on
 * stage A's stream; the caller's buffer is free when the call returns. */
int scal_pipeline_push_host(scal_pipeline_t* p, const void* xyz, int n, int stride_bytes);
/* Result of the oldest scan not yet popped; blocks until its mapping pose (and, in sc_mode 1, its loop answer) is on the host.
 * An error of any stage is returned here (and by every later call) with the failing stage's message in scal_last_error(). */
int scal_pipeline_pop(scal_pipeline_t* p, scal_pipeline_result* out);
/* waits until every pushed scan has its result ready AND the last scan's map insertion has finished; results stay queued for pop */
int scal_pipeline_drain(scal_pipeline_t* p);
/* scans pushed and not popped */
int scal_pipeline_in_flight(scal_pipeline_t* p);
/* the contexts inside (owned by the pipeline): pre-fill the ScanContext database, export the map, read statistics.  Only while the
 * pipeline is drained. */
scal_sc_t* scal_pipeline_sc(scal_pipeline_t* p);
scal_map_t* scal_pipeline_map(scal_pipeline_t* p);
scal_odom_t* scal_pipeline_odom(scal_pipeline_t* p);
scal_features_t* scal_pipeline_features(scal_pipeline_t* p, int i);

/* ---- several independent sequences (sensor streams) on ONE GPU, their kernels sharing launches ("batched /
 * streamed numbers - several independent scans or sequences in flight").  One scan's kernels run on 1..50 workgroups of a 256-CU part
 * and the device advances only four dependent chains at full rate, so S sequences go faster only when their kernels share a launch:
 * every hot-path kernel takes up to 4 argument sets (blockIdx.z selects one), the per-stage calls of the S sequences are recorded and
 * merged position by position (csrc/batch.hpp).  The sequences step together: push one scan of EVERY sequence, pop their S results.
 * Each sequence has its own contexts (map, poses, ScanContext database), and its poses are bit-identical to running it alone. */
int scal_pipeline_create_multi(const scal_pipeline_config* cfg, int n_seqs /* 1..4 */, scal_pipeline_t** p);
int scal_pipeline_seqs(scal_pipeline_t* p);
int scal_pipeline_push_device_multi(scal_pipeline_t* p, const float* const* d_xyz /* [n_seqs] */, const int* n /* [n_seqs] */, int stride_floats);
int scal_pipeline_pop_multi(scal_pipeline_t* p, scal_pipeline_result* out /* [n_seqs] */);
scal_sc_t* scal_pipeline_sc_of(scal_pipeline_t* p, int seq);
scal_map_t* scal_pipeline_map_of(scal_pipeline_t* p, int seq);

/* ------------------------------------------------------------------ factor evaluation (Ceres adapter mode)
 * Batched residual / Jacobian / normal-equation evaluation of lidarFactor.hpp:12-138 blocks at a pose,
 * for a host that keeps ceres::Problem orchestration.  kind: 0 LidarEdgeFactor(a,b),
 * 1 LidarPlaneFactor(j, unit normal), 2 LidarPlaneNormFactor(n, d in pb[0]).  Arrays are [n][3] doubles. */
int scal_factors_eval(int device, int n, const int* kind, const double* cp, const double* pa, const double* pb,
                      const double* x7 /* qx qy qz qw tx ty tz */, double* cost, double* gradient6, double* hessian6x6);

#ifdef __cplusplus
}
#endif
#endif /* SCALOAM_HIP_H */
