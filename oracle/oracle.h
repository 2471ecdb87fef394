/* ORACLE — TEST INFRASTRUCTURE ONLY.  C-ABI of the CPU restatement (liboracle.so) so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.
 * The product (libscaloam_hip.so) never includes or links this. */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_VLP16 = 0, ORC_HDL32 = 1, ORC_HDL64 = 2, ORC_OS1_64 = 3 };
enum { ORC_E_SCANLINE = -1, ORC_E_LIDARTYPE = -2, ORC_E_EMPTY = -3, ORC_E_ARG = -4 };

typedef struct {
    int lidar_type;       /* ORC_* */
    int n_scans;          /* scan_line rosparam */
    double minimum_range; /* minimum_range rosparam */
    int float_math;       /* 0: GCC-5 pin (atan/sqrt promote to double), 1: float overloads */
    int cr_libm;          /* 1: atanf/atan2f replaced by correctly rounded values */
    int sort_mode;        /* 0: std::sort by curvature only (literal), 1: (curvature, index) */
    int voxel_order;      /* order_mode of orc_voxel_grid */
    int check_finite;     /* removeNaNFromPointCloud on a non-dense cloud */
} OrcFeatureConfig;

typedef struct {
    /* caller-owned arrays, capacity n (input size) unless noted */
    float* cloud;      /* [n][4] ordered cloud xyzi (/velodyne_cloud_2) */
    int* src_index;    /* [n] optional: input index of each ordered point */
    float* curvature;  /* [n] optional */
    int* label;        /* [n] optional */
    int* ring_start;   /* [n_scans] scanStartInd */
    int* ring_end;     /* [n_scans] scanEndInd */
    int* sharp;        /* [n] indices into cloud, emission order */
    int* less_sharp;   /* [n] */
    int* flat;         /* [n] */
    float* less_flat;  /* [n][4] downsampled lessFlat cloud */
    int n_kept, n_sharp, n_less_sharp, n_flat, n_less_flat;
    int n_ties; /* adjacent equal-curvature pairs inside sorted segments */
} OrcFeatureOut;

int orc_features_run(const OrcFeatureConfig* cfg, const float* xyz, int n, int stride_floats, OrcFeatureOut* out);

int orc_voxel_grid(const float* xyzi, int n, float leaf, int order_mode, float* out_xyzi, int* n_out, int* guard_hit);
/* one keyframe of the offline map merge (makeMergedMap.py:95-133): f64 rigid transform, near-range removal, f32 out; returns count */
/* loop-closure verification ICP (laserPosegraphOptimization.cpp:497-548, pcl::IterativeClosestPoint restated); returns hasConverged() */
int orc_icp_align(const float* src_xyzi, int n_src, const float* tgt_xyzi, int n_tgt, double max_corr, int max_iter, double trans_eps,
                  double fit_eps, double* T16_out, double* fitness, int* iterations, int* state);
void orc_icp_transform_from_sums(const double* sums16, double* T16);
int orc_mapmerge_frame(const float* xyzi, int n, const double* pose12, double near_thres, float* out_xyzi);

/* ---- stage D: ScanContext ---- */
typedef struct {
    double max_radius;   /* PC_MAX_RADIUS */
    double dist_thres;   /* SC_DIST_THRES */
    int float_math;      /* xy2theta atan overload, as above */
    int cr_libm;
} OrcSCConfig;
void* orc_sc_create(const OrcSCConfig* cfg);
void orc_sc_destroy(void* h);
int orc_sc_size(void* h);
/* makeScancontext only: desc 20x60 column-major doubles */
void orc_sc_make(void* h, const float* xyzi, int n, double* desc);
void orc_sc_keys(const double* desc, double* ringkey20, double* sectorkey60);
/* makeAndSaveScancontextAndKeys / saveScancontextAndKeys */
void orc_sc_insert_cloud(void* h, const float* xyzi, int n);
void orc_sc_insert_desc(void* h, const double* desc);
void orc_sc_get(void* h, int idx, double* desc, float* ringkey20);
/* distanceBtnScanContext */
void orc_sc_distance(const double* sc1, const double* sc2, double* dist, int* shift);
/* all 60 shifts of distDirectSC (for the dense mode) */
void orc_sc_distance_full(const double* sc1, const double* sc2, double* dist60);
/* detectLoopClosureID; returns loop_id (-1 none); out: yaw, min_dist, nn_idx, cand[3], cand_keyd[3] */
int orc_sc_detect(void* h, float* yaw, double* min_dist, int* nn_idx, int* cand, float* cand_d);

/* ---- stage C: mapping ---- */
typedef struct {
    float line_res, plane_res;
    int voxel_order;
    int knn_mode; /* 0 kd-tree, 1 brute force */
} OrcMapConfig;
typedef struct {
    int n_corner_stack, n_surf_stack, n_corner_map, n_surf_map;
    int n_edge[2], n_plane[2];
    int lm_iters[2];
    int lm_success[2];
    double cost_init[2], cost_final[2];
    int solved;
    double t_ms[8]; /* shift/gather, ds, tree, assoc, solve, insert, filter, total */
} OrcMapStats;
void* orc_map_create(const OrcMapConfig* cfg);
void orc_map_destroy(void* h);
/* one laserMapping process() pass. q_wodom/t_wodom: odometry pose; out q_w_curr/t_w_curr.
   full_res may be NULL; registered (same size) optional. */
int orc_map_step(void* h, const float* corner_last, int n_corner, const float* surf_last, int n_surf,
                 const float* full_res, int n_full, const double* q_wodom_xyzw, const double* t_wodom,
                 double* q_w_curr_xyzw, double* t_w_curr, float* registered, OrcMapStats* stats);
/* current map content of the valid (5x5x3) window in reference gather order; returns counts */
int orc_map_export(void* h, int which /*0 corner,1 surf*/, float* out_xyzi, int cap);
int orc_map_export_all(void* h, int which, float* out_xyzi, int cap); /* all 4851 cubes (laserMapping.cpp:824-837) */
void orc_map_get_wmap_wodom(void* h, double* q_xyzw, double* t);

/* ---- stage B: odometry ---- */
typedef struct {
    int n_edge[2], n_plane[2];
    int lm_iters[2];
    double cost_init[2], cost_final[2];
    double t_ms[4];
} OrcOdomStats;
void* orc_odom_create(void);
void orc_odom_destroy(void* h);
/* one laserOdometry main-loop pass over the five stage-A clouds */
int orc_odom_step(void* h, const float* sharp, int n_sharp, const float* less_sharp, int n_less_sharp,
                  const float* flat, int n_flat, const float* less_flat, int n_less_flat,
                  double* q_last_curr_xyzw, double* t_last_curr, double* q_w_curr_xyzw, double* t_w_curr,
                  OrcOdomStats* stats);

/* ---- LM evaluation helpers exposed for unit tests (autodiff Jets, lidarFactor.hpp) ---- */
/* kind 0 edge(a,b), 1 plane(j, n), 2 planenorm(n,d). params: 9 doubles per factor (see lm.cpp) */
void orc_factor_eval(int kind, const double* cp, const double* params, const double* x7, double* residual3,
                     double* jac3x7);

/* third-party sub-steps exposed so that tests can cross-check them against numpy/scipy */
void orc_eig3_sym(const double* A9, double* w3, double* V9);            /* Eigen::SelfAdjointEigenSolver stand-in */
void orc_plane_fit_5x3(const double* A15, const double* b5, double* x3); /* colPivHouseholderQr().solve stand-in */
/* ceres::Solve stand-in on explicit factor arrays ([n] kinds, [n][3] cp/pa/pb); x7 in/out; returns iterations */
int orc_ceres_solve(int n, const int* kind, const double* cp, const double* pa, const double* pb, double* x7,
                    double* cost_trace /*[6]*/, int* n_trace, int* termination);

#ifdef __cplusplus
}
#endif
#endif
