/* Synthetic LiDAR scan generator (seeded procedural world + ray-cast sensor models).
 * Input data for tests and bench.py; no dataset exists offline.  Not the oracle, not the product. */
#ifndef SCAN_SYNTH_H
#define SCAN_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
enum { SYN_VLP16 = 0, SYN_HDL32 = 1, SYN_HDL64 = 2, SYN_OS1_64 = 3 };
typedef struct {
    int sensor;
    uint64_t seed;
    int n_boxes, n_cyl;
    double region[4]; /* xmin xmax ymin ymax of the object field */
    double noise_sigma;
    int threads;
} SynthConfig;
void* syn_world_create(const SynthConfig* cfg);
void syn_world_destroy(void* w);
/* trajectory pose of scan k (10 Hz, 10 m/s arc, yaw rate 0.1 rad/s, +-1 deg roll/pitch wobble): q xyzw, t */
void syn_world_pose(void* w, int k, double* q_xyzw, double* t);
int syn_world_max_points(void* w);
int syn_world_scan(void* w, int k, float* out_xyz);
int syn_world_scan_pose(void* w, const double* q_xyzw, const double* t, uint64_t noise_seed, float* out_xyz);
#ifdef __cplusplus
}
#endif
#endif
