"""ctypes binding of libscaloam_hip.so (include/scaloam_hip.h) for tests, bench.py and Python callers.

This is plumbing only: every function forwards to the C-ABI; there is no Python or CPU implementation
behind it, and loading fails loudly when the HIP library has not been built.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCALOAM_LIB: another build of the same library (A/B measurements of two builds on one GPU box); never a different implementation
LIB_PATH = os.environ.get("SCALOAM_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libscaloam_hip.so"))

VLP16, HDL32, HDL64, OS1_64 = 0, 1, 2, 3
SCAN_LINES = {VLP16: 16, HDL32: 32, HDL64: 64, OS1_64: 64}

OK, E_ARG, E_LIDAR_TYPE, E_SCAN_LINE, E_TOO_MANY, E_EMPTY, E_HIP, E_NO_DEVICE, E_CAPACITY, E_STATE = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int)


class ScalError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"scaloam_hip error {code}: {msg}")
        self.code = code


class FeaturesConfig(C.Structure):
    _fields_ = [("lidar_type", C.c_int), ("n_scans", C.c_int), ("minimum_range", C.c_double), ("max_points", C.c_int),
                ("float_math", C.c_int), ("check_finite", C.c_int), ("device", C.c_int)]


class FeaturesOut(C.Structure):
    _fields_ = [("cloud", _f32p), ("src_index", _i32p), ("curvature", _f32p), ("label", _i32p), ("ring_start", _i32p),
                ("ring_end", _i32p), ("sharp", _i32p), ("less_sharp", _i32p), ("flat", _i32p), ("less_flat", _f32p),
                ("n_kept", C.c_int), ("n_sharp", C.c_int), ("n_less_sharp", C.c_int), ("n_flat", C.c_int),
                ("n_less_flat", C.c_int), ("n_tied_segments", C.c_int)]


class SCConfig(C.Structure):
    _fields_ = [("max_radius", C.c_double), ("dist_thres", C.c_double), ("max_keyframes", C.c_int), ("float_math", C.c_int),
                ("device", C.c_int), ("n_shards", C.c_int), ("shard", C.c_int), ("side_stream", C.c_int)]


class SCResult(C.Structure):
    _fields_ = [("loop_id", C.c_int), ("yaw_rad", C.c_float), ("min_dist", C.c_double), ("nn_idx", C.c_int), ("nn_shift", C.c_int),
                ("cand_idx", C.c_int * 3), ("cand_keydist", C.c_float * 3), ("cand_scdist", C.c_double * 3), ("cand_shift", C.c_int * 3)]


class SCCand(C.Structure):
    _fields_ = [("key_dist", C.c_float), ("idx", C.c_int), ("sc_dist", C.c_double), ("shift", C.c_int), ("pad", C.c_int)]


class MapConfig(C.Structure):
    _fields_ = [("line_res", C.c_float), ("plane_res", C.c_float), ("max_scan_points", C.c_int), ("max_map_points", C.c_int),
                ("device", C.c_int)]


class MapStats(C.Structure):
    _fields_ = [("n_corner_stack", C.c_int), ("n_surf_stack", C.c_int), ("n_corner_map", C.c_int), ("n_surf_map", C.c_int),
                ("n_edge", C.c_int * 2), ("n_plane", C.c_int * 2), ("lm_iters", C.c_int * 2), ("lm_success", C.c_int * 2),
                ("cost_init", C.c_double * 2), ("cost_final", C.c_double * 2), ("solved", C.c_int),
                ("n_map_corner_total", C.c_int), ("n_map_surf_total", C.c_int), ("insert_path", C.c_int)]


class ICPConfig(C.Structure):
    _fields_ = [("max_corr_dist", C.c_double), ("transformation_epsilon", C.c_double), ("fitness_epsilon", C.c_double),
                ("max_iterations", C.c_int), ("max_source", C.c_int), ("max_target", C.c_int), ("device", C.c_int)]


class ICPResult(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("state", C.c_int), ("n_correspondences", C.c_int),
                ("fitness", C.c_double), ("T", C.c_double * 16)]


class MapMergeConfig(C.Structure):
    _fields_ = [("max_points", C.c_longlong), ("max_frame_points", C.c_int), ("device", C.c_int)]


class OdomConfig(C.Structure):
    _fields_ = [("max_points", C.c_int), ("device", C.c_int)]


class OdomStats(C.Structure):
    _fields_ = [("n_edge", C.c_int * 2), ("n_plane", C.c_int * 2), ("lm_iters", C.c_int * 2), ("lm_success", C.c_int * 2),
                ("cost_init", C.c_double * 2), ("cost_final", C.c_double * 2)]


class PipelineConfig(C.Structure):
    _fields_ = [("lidar_type", C.c_int), ("n_scans", C.c_int), ("minimum_range", C.c_double), ("max_points", C.c_int),
                ("float_math", C.c_int), ("check_finite", C.c_int), ("line_res", C.c_float), ("plane_res", C.c_float),
                ("max_map_points", C.c_int), ("sc_max_radius", C.c_double), ("sc_dist_thres", C.c_double), ("sc_max_keyframes", C.c_int),
                ("sc_mode", C.c_int), ("device", C.c_int), ("ring", C.c_int), ("depth", C.c_int), ("d_desc_ring", C.c_void_p)]


class PipelineResult(C.Structure):
    _fields_ = [("seq", C.c_longlong), ("q_w_curr", C.c_double * 4), ("t_w_curr", C.c_double * 3), ("q_odom", C.c_double * 4),
                ("t_odom", C.c_double * 3), ("odom", OdomStats), ("map", MapStats), ("have_loop", C.c_int), ("loop", SCResult),
                ("d_descriptor", C.c_void_p)]


# every symbol include/scaloam_hip.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = [
    "scal_last_error", "scal_device_count", "scal_version", "scal_prof_enable", "scal_prof_filter", "scal_prof_reset", "scal_prof_read", "scal_prof_names", "scal_prof_timeline", "scal_prof_timeline_dump",
    "scal_features_create", "scal_features_destroy", "scal_features_run", "scal_features_run_device", "scal_features_enqueue_host", "scal_features_stream", "scal_voxel_stream", "scal_sc_stream", "scal_map_stream", "scal_odom_stream", "scal_features_fetch",
    "scal_features_sync",
    "scal_voxel_create", "scal_voxel_destroy", "scal_voxel_downsample", "scal_voxel_downsample_device",
    "scal_sc_create", "scal_sc_destroy", "scal_sc_size", "scal_sc_insert_cloud", "scal_sc_insert_cloud_device",
    "scal_sc_insert_descriptor", "scal_sc_get_descriptor", "scal_sc_make_descriptor", "scal_sc_detect", "scal_sc_detect_enqueue", "scal_sc_batch_loop_search", "scal_sc_detect_collect", "scal_sc_distance_pairs",
    "scal_sc_distance_matrix", "scal_sc_distance_matrix_device", "scal_sc_shard_query", "scal_sc_merge_candidates", "scal_sc_insert_features", "scal_sc_make_features",
    "scal_sc_insert_descriptor_device", "scal_sc_shard_query_device", "scal_sc_shard_query_batch_device", "scal_sc_insert_descriptors_device", "scal_sc_sync", "scal_sc_make_features_enqueue", "scal_sc_wait_descriptor",
    "scal_map_create", "scal_map_destroy", "scal_map_step", "scal_map_step_features", "scal_map_export", "scal_map_export_all", "scal_map_get_wmap_wodom", "scal_map_set_merge_insert", "scal_map_get_path_counters", "scal_map_set_poll", "scal_map_debug_set_lm_polls", "scal_map_debug_set_grid_cap", "scal_map_adapter_begin", "scal_map_associate", "scal_map_get_blocks", "scal_map_eval_blocks", "scal_map_adapter_finish", "scal_odom_adapter_begin", "scal_odom_associate", "scal_odom_get_blocks", "scal_odom_eval_blocks", "scal_odom_adapter_finish", "scal_map_prefetch_features", "scal_map_prefetch_begin", "scal_map_prefetch_finish", "scal_map_enqueue_features", "scal_map_collect", "scal_map_finish",
    "scal_set_stream_mode", "scal_mapmerge_create", "scal_mapmerge_destroy", "scal_mapmerge_reset", "scal_mapmerge_add",
    "scal_mapmerge_add_batch_device", "scal_mapmerge_size", "scal_mapmerge_download", "scal_mapmerge_device_points", "scal_mapmerge_downsample", "scal_icp_create", "scal_icp_destroy", "scal_icp_align", "scal_icp_align_device", "scal_icp_set_search",
    "scal_odom_create", "scal_odom_destroy", "scal_odom_step", "scal_odom_step_features", "scal_odom_enqueue_features", "scal_odom_collect",
    "scal_factors_eval",
    "scal_pipeline_create", "scal_pipeline_destroy", "scal_pipeline_push_device", "scal_pipeline_push_host", "scal_pipeline_pop", "scal_pipeline_drain",
    "scal_pipeline_in_flight", "scal_pipeline_sc", "scal_pipeline_map", "scal_pipeline_odom", "scal_pipeline_features",
    "scal_pipeline_create_multi", "scal_pipeline_seqs", "scal_pipeline_push_device_multi", "scal_pipeline_pop_multi", "scal_pipeline_sc_of", "scal_pipeline_map_of",
]

_lib = None


def lib():
    """Load libscaloam_hip.so.  Raises if it is missing: there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C sc-a-loam_amd` or __graft_entry__.build()")
    L = C.CDLL(LIB_PATH)
    L.scal_last_error.restype = C.c_char_p
    L.scal_version.restype = C.c_char_p
    vp = C.c_void_p
    L.scal_features_create.argtypes = [C.POINTER(FeaturesConfig), C.POINTER(vp)]
    L.scal_features_destroy.argtypes = [vp]
    L.scal_features_destroy.restype = None
    L.scal_features_run.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(FeaturesOut)]
    L.scal_features_run_device.argtypes = [vp, vp, C.c_int, C.c_int]
    L.scal_features_enqueue_host.argtypes = [vp, vp, C.c_int, C.c_int]
    for fn in ("scal_features_stream", "scal_voxel_stream", "scal_sc_stream", "scal_map_stream", "scal_odom_stream"):
        getattr(L, fn).restype = C.c_void_p
        getattr(L, fn).argtypes = [vp]
    L.scal_features_fetch.argtypes = [vp, C.POINTER(FeaturesOut)]
    L.scal_features_sync.argtypes = [vp]
    L.scal_voxel_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.scal_voxel_destroy.argtypes = [vp]
    L.scal_voxel_destroy.restype = None
    L.scal_voxel_downsample.argtypes = [vp, _f32p, C.c_int, C.c_float, _f32p, _i32p]
    L.scal_voxel_downsample_device.argtypes = [vp, vp, C.c_int, C.c_float, vp, _i32p]
    L.scal_sc_create.argtypes = [C.POINTER(SCConfig), C.POINTER(vp)]
    L.scal_sc_destroy.argtypes = [vp]
    L.scal_sc_destroy.restype = None
    L.scal_sc_size.argtypes = [vp]
    L.scal_sc_insert_cloud.argtypes = [vp, _f32p, C.c_int]
    L.scal_sc_insert_cloud_device.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    L.scal_sc_insert_descriptor.argtypes = [vp, _f64p]
    L.scal_sc_get_descriptor.argtypes = [vp, C.c_int, _f64p, _f32p]
    L.scal_sc_make_descriptor.argtypes = [vp, _f32p, C.c_int, _f64p]
    L.scal_sc_detect.argtypes = [vp, C.POINTER(SCResult)]
    L.scal_sc_detect_enqueue.argtypes = [vp]
    L.scal_sc_detect_collect.argtypes = [vp, C.POINTER(SCResult)]
    L.scal_sc_distance_pairs.argtypes = [vp, _i32p, _i32p, C.c_int, _f64p, _i32p]
    L.scal_sc_distance_matrix.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f64p, _i32p]
    L.scal_sc_distance_matrix_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.scal_sc_shard_query.argtypes = [vp, _f64p, C.c_int, C.POINTER(SCCand)]
    L.scal_sc_merge_candidates.argtypes = [C.POINTER(SCCand), C.c_int, C.c_double, C.POINTER(SCResult)]
    L.scal_sc_insert_features.argtypes = [vp, vp]
    L.scal_sc_make_features.argtypes = [vp, vp, vp]
    L.scal_sc_insert_descriptor_device.argtypes = [vp, vp]
    L.scal_sc_shard_query_device.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.scal_sc_shard_query_batch_device.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int), vp]
    L.scal_sc_insert_descriptors_device.argtypes = [vp, vp, C.c_int]
    L.scal_sc_sync.argtypes = [vp]
    L.scal_sc_make_features_enqueue.argtypes = [vp, vp, vp]
    L.scal_sc_wait_descriptor.argtypes = [vp]
    L.scal_prof_enable.argtypes = [C.c_int]
    L.scal_prof_filter.argtypes = [C.c_char_p]
    L.scal_prof_read.argtypes = [C.c_char_p, _f64p, C.POINTER(C.c_long)]
    L.scal_prof_names.argtypes = [C.c_char_p, C.c_int]
    L.scal_map_create.argtypes = [C.POINTER(MapConfig), C.POINTER(vp)]
    L.scal_map_destroy.argtypes = [vp]
    L.scal_map_destroy.restype = None
    L.scal_map_step.argtypes = [vp, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f64p, _f64p, _f64p, _f64p, _f32p, C.POINTER(MapStats)]
    L.scal_map_step_features.argtypes = [vp, vp, _f64p, _f64p, _f64p, _f64p, C.POINTER(MapStats)]
    L.scal_map_export.argtypes = [vp, C.c_int, _f32p, C.c_int]
    L.scal_map_export_all.argtypes = [vp, C.c_int, _f32p, C.c_int]
    L.scal_map_get_wmap_wodom.argtypes = [vp, _f64p, _f64p]
    L.scal_map_set_merge_insert.argtypes = [vp, C.c_int]
    L.scal_map_get_path_counters.argtypes = [vp, C.POINTER(C.c_int)]
    L.scal_map_adapter_begin.argtypes = [vp, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f64p, _f64p, _f64p, _f64p]
    L.scal_map_associate.argtypes = [vp, _f64p, _f64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.scal_map_get_blocks.argtypes = [vp, vp, C.c_int]
    L.scal_map_eval_blocks.argtypes = [vp, _f64p, C.c_int, _f64p, _f64p]
    L.scal_map_adapter_finish.argtypes = [vp, _f64p, _f64p, _f32p, C.POINTER(MapStats)]
    L.scal_odom_adapter_begin.argtypes = [vp, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f64p, _f64p, C.POINTER(C.c_int)]
    L.scal_odom_associate.argtypes = [vp, _f64p, _f64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.scal_odom_get_blocks.argtypes = [vp, vp, C.c_int]
    L.scal_odom_eval_blocks.argtypes = [vp, _f64p, C.c_int, _f64p, _f64p]
    L.scal_odom_adapter_finish.argtypes = [vp, _f64p, _f64p, _f64p, _f64p]
    L.scal_map_set_poll.argtypes = [vp, C.c_int]
    L.scal_map_debug_set_lm_polls.argtypes = [vp, C.c_int]
    if hasattr(L, "scal_map_debug_set_grid_cap"):  # absent from older builds compared through SCALOAM_LIB (tools/gpu_ab.sh)
        L.scal_map_debug_set_grid_cap.argtypes = [vp, C.c_int, C.c_int]
    L.scal_map_prefetch_features.argtypes = [vp, vp]
    L.scal_map_prefetch_begin.argtypes = [vp, vp]
    L.scal_map_prefetch_finish.argtypes = [vp, vp]
    L.scal_map_enqueue_features.argtypes = [vp, vp, _f64p, _f64p]
    L.scal_map_collect.argtypes = [vp, _f64p, _f64p, C.POINTER(MapStats)]
    L.scal_map_finish.argtypes = [vp]
    L.scal_set_stream_mode.argtypes = [C.c_int]
    L.scal_icp_create.argtypes = [C.POINTER(ICPConfig), C.POINTER(vp)]
    L.scal_icp_destroy.argtypes = [vp]
    L.scal_icp_destroy.restype = None
    L.scal_icp_align.argtypes = [vp, _f32p, C.c_int, _f32p, C.c_int, C.POINTER(ICPResult)]
    L.scal_icp_set_search.argtypes = [vp, C.c_int]
    L.scal_icp_align_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.POINTER(ICPResult)]
    L.scal_mapmerge_create.argtypes = [C.POINTER(MapMergeConfig), C.POINTER(vp)]
    L.scal_mapmerge_destroy.argtypes = [vp]
    L.scal_mapmerge_destroy.restype = None
    L.scal_mapmerge_reset.argtypes = [vp]
    L.scal_mapmerge_add.argtypes = [vp, _f32p, C.c_int, _f64p, C.c_double]
    L.scal_mapmerge_add_batch_device.argtypes = [vp, vp, C.POINTER(C.c_int), _f64p, C.c_int, C.c_double]
    L.scal_mapmerge_size.argtypes = [vp]
    L.scal_mapmerge_size.restype = C.c_longlong
    L.scal_mapmerge_download.argtypes = [vp, _f32p, C.c_longlong]
    L.scal_mapmerge_downsample.argtypes = [vp, C.c_float, _f32p, C.c_longlong, C.POINTER(C.c_longlong)]
    L.scal_mapmerge_device_points.argtypes = [vp]
    L.scal_mapmerge_device_points.restype = vp
    L.scal_odom_create.argtypes = [C.POINTER(OdomConfig), C.POINTER(vp)]
    L.scal_odom_destroy.argtypes = [vp]
    L.scal_odom_destroy.restype = None
    L.scal_odom_step.argtypes = [vp, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f64p, _f64p, _f64p, _f64p,
                                 C.POINTER(OdomStats)]
    L.scal_odom_step_features.argtypes = [vp, vp, _f64p, _f64p, _f64p, _f64p, C.POINTER(OdomStats)]
    L.scal_odom_enqueue_features.argtypes = [vp, vp]
    L.scal_odom_collect.argtypes = [vp, _f64p, _f64p, _f64p, _f64p, C.POINTER(OdomStats)]
    L.scal_factors_eval.argtypes = [C.c_int, C.c_int, _i32p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p]
    L.scal_pipeline_create.argtypes = [C.POINTER(PipelineConfig), C.POINTER(vp)]
    L.scal_pipeline_destroy.argtypes = [vp]
    L.scal_pipeline_destroy.restype = None
    L.scal_pipeline_push_device.argtypes = [vp, vp, C.c_int, C.c_int]
    L.scal_pipeline_push_host.argtypes = [vp, vp, C.c_int, C.c_int]
    L.scal_pipeline_pop.argtypes = [vp, C.POINTER(PipelineResult)]
    L.scal_pipeline_drain.argtypes = [vp]
    L.scal_pipeline_in_flight.argtypes = [vp]
    for fn in ("scal_pipeline_sc", "scal_pipeline_map", "scal_pipeline_odom"):
        getattr(L, fn).restype = C.c_void_p
        getattr(L, fn).argtypes = [vp]
    L.scal_pipeline_features.restype = C.c_void_p
    L.scal_pipeline_features.argtypes = [vp, C.c_int]
    L.scal_pipeline_create_multi.argtypes = [C.POINTER(PipelineConfig), C.c_int, C.POINTER(vp)]
    L.scal_pipeline_seqs.argtypes = [vp]
    L.scal_pipeline_push_device_multi.argtypes = [vp, C.POINTER(C.c_void_p), _i32p, C.c_int]
    L.scal_pipeline_pop_multi.argtypes = [vp, C.POINTER(PipelineResult)]
    for fn in ("scal_pipeline_sc_of", "scal_pipeline_map_of"):
        getattr(L, fn).restype = C.c_void_p
        getattr(L, fn).argtypes = [vp, C.c_int]
    _lib = L
    return L


def _check(rc):
    if rc != OK:
        raise ScalError(rc, lib().scal_last_error().decode())


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def set_stream_mode(mode):
    _check(lib().scal_set_stream_mode(int(mode)))


def device_count():
    return lib().scal_device_count()


def prof_enable(on, kernel=None):
    lib().scal_prof_filter(kernel.encode() if kernel else None)
    lib().scal_prof_enable(1 if on else 0)


def prof_reset():
    lib().scal_prof_reset()


def prof_timeline(on):
    lib().scal_prof_timeline(1 if on else 0)


def prof_timeline_dump(path):
    lib().scal_prof_timeline_dump(path.encode())


def prof_read_all():
    """{kernel name: (total_ms, launches)} measured with HIP events on the launching stream."""
    buf = C.create_string_buffer(4096)
    lib().scal_prof_names(buf, 4096)
    out = {}
    for name in buf.value.decode().split(";"):
        if not name:
            continue
        ms = C.c_double(0)
        cnt = C.c_long(0)
        lib().scal_prof_read(name.encode(), C.byref(ms), C.byref(cnt))
        out[name] = (ms.value, cnt.value)
    return out


# ---------------------------------------------------------------------------------------------- stage A
class ScanRegistration:
    """Mirror of the reference's scanRegistration node state (scan_line, lidar_type, minimum_range rosparams,
    scanRegistration.cpp:480-482) with laserCloudHandler as a method."""

    def __init__(self, lidar_type, minimum_range, max_points=400000, float_math=0, check_finite=1, device=0, n_scans=None):
        self.cfg = FeaturesConfig(lidar_type, SCAN_LINES.get(lidar_type, 0) if n_scans is None else n_scans, float(minimum_range),
                                  max_points, float_math, check_finite, device)
        self.h = C.c_void_p()
        _check(lib().scal_features_create(C.byref(self.cfg), C.byref(self.h)))
        self.n_scans = self.cfg.n_scans
        m = max_points
        ns = self.n_scans
        self._b = dict(cloud=np.zeros((m, 4), np.float32), src_index=np.zeros(m, np.int32), curvature=np.zeros(m, np.float32),
                       label=np.zeros(m, np.int32), ring_start=np.zeros(ns, np.int32), ring_end=np.zeros(ns, np.int32),
                       sharp=np.zeros(12 * ns, np.int32), less_sharp=np.zeros(120 * ns, np.int32), flat=np.zeros(24 * ns, np.int32),
                       less_flat=np.zeros((m, 4), np.float32))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.scal_features_destroy(self.h)
        self.h = None

    __del__ = close

    def _out(self):
        b = self._b
        return FeaturesOut(_p(b["cloud"], _f32p), _p(b["src_index"], _i32p), _p(b["curvature"], _f32p), _p(b["label"], _i32p),
                           _p(b["ring_start"], _i32p), _p(b["ring_end"], _i32p), _p(b["sharp"], _i32p), _p(b["less_sharp"], _i32p),
                           _p(b["flat"], _i32p), _p(b["less_flat"], _f32p))

    def _pack(self, o):
        b = self._b
        k = o.n_kept
        return dict(n_kept=k, cloud=b["cloud"][:k].copy(), src_index=b["src_index"][:k].copy(), curvature=b["curvature"][:k].copy(),
                    label=b["label"][:k].copy(), ring_start=b["ring_start"].copy(), ring_end=b["ring_end"].copy(),
                    sharp=b["sharp"][:o.n_sharp].copy(), less_sharp=b["less_sharp"][:o.n_less_sharp].copy(),
                    flat=b["flat"][:o.n_flat].copy(), less_flat=b["less_flat"][:o.n_less_flat].copy(), n_tied_segments=o.n_tied_segments)

    def laserCloudHandler(self, xyz):
        xyz = _f32(xyz)
        o = self._out()
        stride = xyz.strides[0] if xyz.shape[0] > 0 else 4 * xyz.shape[1]
        _check(lib().scal_features_run(self.h, xyz.ctypes.data, xyz.shape[0], stride, C.byref(o)))
        return self._pack(o)

    def enqueue_host(self, xyz):
        """asynchronous laserCloudHandler: pinned staging + upload + stage A on the context's stream, results stay on the GPU"""
        xyz = _f32(xyz)
        stride = xyz.strides[0] if xyz.shape[0] > 0 else 4 * xyz.shape[1]
        _check(lib().scal_features_enqueue_host(self.h, xyz.ctypes.data, xyz.shape[0], stride))

    def run_device(self, d_ptr, n, stride_floats):
        _check(lib().scal_features_run_device(self.h, d_ptr, n, stride_floats))

    def fetch(self):
        o = self._out()
        _check(lib().scal_features_fetch(self.h, C.byref(o)))
        return self._pack(o)

    def sync(self):
        _check(lib().scal_features_sync(self.h))


# ---------------------------------------------------------------------------------------------- voxel grid
class VoxelGrid:
    def __init__(self, max_points=400000, device=0):
        self.h = C.c_void_p()
        _check(lib().scal_voxel_create(max_points, device, C.byref(self.h)))
        self.cap = max_points

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.scal_voxel_destroy(self.h)
        self.h = None

    __del__ = close

    def filter(self, xyzi, leaf):
        xyzi = _f32(xyzi).reshape(-1, 4)
        out = np.zeros((max(1, xyzi.shape[0]), 4), np.float32)
        n = C.c_int(0)
        _check(lib().scal_voxel_downsample(self.h, _p(xyzi, _f32p), xyzi.shape[0], C.c_float(leaf), _p(out, _f32p), C.byref(n)))
        return out[:n.value].copy()

    def stream_ptr(self):
        """the context's hipStream_t (to order it against the caller's streams, e.g. torch.cuda.ExternalStream(ptr))"""
        return lib().scal_voxel_stream(self.h)

    def filter_device(self, d_in_ptr, n, leaf, d_out_ptr):
        """16-byte xyzi records in device memory in, centroids to d_out_ptr (room for n records); returns their number."""
        m = C.c_int(0)
        _check(lib().scal_voxel_downsample_device(self.h, d_in_ptr, n, C.c_float(leaf), d_out_ptr, C.byref(m)))
        return m.value


# ---------------------------------------------------------------------------------------------- stage D
class SCManager:
    """Mirror of the reference's SCManager public API (Scancontext.h:62-79, :107-108).  Descriptors cross this
    boundary as [ring, sector] numpy arrays; the C-ABI itself uses Eigen's column-major layout."""

    def __init__(self, max_radius=80.0, dist_thres=0.2, max_keyframes=8192, float_math=0, device=0, n_shards=1, shard=0, side_stream=0):
        self.cfg = SCConfig(max_radius, dist_thres, max_keyframes, float_math, device, n_shards, shard, side_stream)
        self.h = C.c_void_p()
        _check(lib().scal_sc_create(C.byref(self.cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.scal_sc_destroy(self.h)
        self.h = None

    __del__ = close

    def size(self):
        return lib().scal_sc_size(self.h)

    def makeScancontext(self, xyzi):
        xyzi = _f32(xyzi).reshape(-1, 4)
        d = np.zeros(1200)
        _check(lib().scal_sc_make_descriptor(self.h, _p(xyzi, _f32p), xyzi.shape[0], _p(d, _f64p)))
        return d.reshape(60, 20).T.copy()

    def makeAndSaveScancontextAndKeys(self, xyzi):
        xyzi = _f32(xyzi).reshape(-1, 4)
        _check(lib().scal_sc_insert_cloud(self.h, _p(xyzi, _f32p), xyzi.shape[0]))

    def saveScancontextAndKeys(self, desc_ring_sector):
        d = _f64(np.asarray(desc_ring_sector).T.reshape(-1))
        _check(lib().scal_sc_insert_descriptor(self.h, _p(d, _f64p)))

    def get(self, idx):
        d = np.zeros(1200)
        k = np.zeros(20, np.float32)
        _check(lib().scal_sc_get_descriptor(self.h, idx, _p(d, _f64p), _p(k, _f32p)))
        return d.reshape(60, 20).T.copy(), k

    def stream_ptr(self):
        """the context's hipStream_t (to order it against the caller's streams, e.g. torch.cuda.ExternalStream(ptr))"""
        return lib().scal_sc_stream(self.h)

    def insert_features(self, feat):
        """keyframe cloud of a ScanRegistration context -> VoxelGrid(0.4) -> makeAndSaveScancontextAndKeys, all on the GPU"""
        _check(lib().scal_sc_insert_features(self.h, feat.h))

    def make_features(self, feat, d_desc_ptr):
        _check(lib().scal_sc_make_features(self.h, feat.h, d_desc_ptr))

    def make_features_enqueue(self, feat, d_desc_ptr):
        _check(lib().scal_sc_make_features_enqueue(self.h, feat.h, d_desc_ptr))

    def wait_descriptor(self):
        _check(lib().scal_sc_wait_descriptor(self.h))

    def insert_descriptor_device(self, d_desc_ptr):
        _check(lib().scal_sc_insert_descriptor_device(self.h, d_desc_ptr))

    def shard_query_batch_device(self, d_queries_ptr, limits, d_out_ptr):
        lim = (C.c_int * len(limits))(*[int(v) for v in limits])
        _check(lib().scal_sc_shard_query_batch_device(self.h, d_queries_ptr, len(limits), lim, d_out_ptr))

    def insert_descriptors_device(self, d_descs_ptr, n):
        _check(lib().scal_sc_insert_descriptors_device(self.h, d_descs_ptr, n))

    def sync(self):
        _check(lib().scal_sc_sync(self.h))

    def shard_query_device(self, d_queries_ptr, nq, global_size_at_rebuild, d_out_ptr):
        _check(lib().scal_sc_shard_query_device(self.h, d_queries_ptr, nq, global_size_at_rebuild, d_out_ptr))

    def detect_enqueue(self):
        _check(lib().scal_sc_detect_enqueue(self.h))

    def detect_collect(self):
        r = SCResult()
        _check(lib().scal_sc_detect_collect(self.h, C.byref(r)))
        return self._result(r)

    def detectLoopClosureID(self):
        r = SCResult()
        _check(lib().scal_sc_detect(self.h, C.byref(r)))
        return self._result(r)

    @staticmethod
    def _result(r):
        return dict(loop_id=r.loop_id, yaw=r.yaw_rad, min_dist=r.min_dist, nn_idx=r.nn_idx, nn_shift=r.nn_shift,
                    cand=np.array(r.cand_idx[:]), cand_d=np.array(r.cand_keydist[:], np.float32), cand_sc=np.array(r.cand_scdist[:]),
                    cand_shift=np.array(r.cand_shift[:]))

    def distance_pairs(self, ia, ib):
        ia = np.ascontiguousarray(ia, np.int32)
        ib = np.ascontiguousarray(ib, np.int32)
        d = np.zeros(ia.shape[0])
        s = np.zeros(ia.shape[0], np.int32)
        _check(lib().scal_sc_distance_pairs(self.h, _p(ia, _i32p), _p(ib, _i32p), ia.shape[0], _p(d, _f64p), _p(s, _i32p)))
        return d, s

    def distance_matrix(self, q0, q1, d0, d1, mode=0):
        d = np.zeros((max(0, q1 - q0), max(0, d1 - d0)))  # inverted ranges are the library's to refuse
        s = np.zeros((max(0, q1 - q0), max(0, d1 - d0)), np.int32)
        _check(lib().scal_sc_distance_matrix(self.h, q0, q1, d0, d1, mode, _p(d, _f64p), _p(s, _i32p)))
        return d, s

    def batch_loop_search(self, q0, q1, exclude_recent=30, k=3, mode=2):
        """(idx, dist, shift) arrays [q1-q0][k]: the k best older keyframes of every query keyframe, exhaustive over the database"""
        n = max(0, q1 - q0)
        idx, dist, shift = np.zeros((n, k), np.int32), np.zeros((n, k)), np.zeros((n, k), np.int32)
        _check(lib().scal_sc_batch_loop_search(self.h, q0, q1, exclude_recent, k, mode, _p(idx, _i32p), _p(dist, _f64p), _p(shift, _i32p)))
        return idx, dist, shift

    def distance_matrix_device(self, q0, q1, d0, d1, mode, d_dist_ptr, d_shift_ptr):
        """Device outputs ((q1-q0)*(d1-d0) doubles / int32), enqueued on the context's stream; sync() before reading."""
        _check(lib().scal_sc_distance_matrix_device(self.h, q0, q1, d0, d1, mode, d_dist_ptr, d_shift_ptr))

    def shard_query(self, query_desc_ring_sector, global_size_at_rebuild):
        d = _f64(np.asarray(query_desc_ring_sector).T.reshape(-1))
        out = (SCCand * 3)()
        _check(lib().scal_sc_shard_query(self.h, _p(d, _f64p), global_size_at_rebuild, out))
        return out


def merge_candidates(cands, dist_thres):
    arr = (SCCand * len(cands))(*cands)
    r = SCResult()
    _check(lib().scal_sc_merge_candidates(arr, len(cands), dist_thres, C.byref(r)))
    return dict(loop_id=r.loop_id, yaw=r.yaw_rad, min_dist=r.min_dist, nn_idx=r.nn_idx, nn_shift=r.nn_shift,
                cand=np.array(r.cand_idx[:]), cand_d=np.array(r.cand_keydist[:], np.float32))


# ---------------------------------------------------------------------------------------------- stage C
class LaserMapping:
    """Mirror of the reference's laserMapping node state (mapping_line_resolution / mapping_plane_resolution,
    laserMapping.cpp:915-916) with one process() pass as `process`."""

    def __init__(self, line_res=0.4, plane_res=0.8, max_scan_points=400000, max_map_points=4000000, device=0):
        self.cfg = MapConfig(line_res, plane_res, max_scan_points, max_map_points, device)
        self.h = C.c_void_p()
        _check(lib().scal_map_create(C.byref(self.cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.scal_map_destroy(self.h)
        self.h = None

    __del__ = close

    def process(self, corner_last, surf_last, full_res, q_wodom, t_wodom, want_registered=False):
        c = _f32(corner_last).reshape(-1, 4)
        s = _f32(surf_last).reshape(-1, 4)
        f = _f32(full_res).reshape(-1, 4) if full_res is not None else None
        q = _f64(q_wodom)
        t = _f64(t_wodom)
        qo = np.zeros(4)
        to = np.zeros(3)
        reg = np.zeros_like(f) if (want_registered and f is not None) else None
        st = MapStats()
        _check(lib().scal_map_step(self.h, _p(c, _f32p), c.shape[0], _p(s, _f32p), s.shape[0], _p(f, _f32p), 0 if f is None else f.shape[0],
                                   _p(q, _f64p), _p(t, _f64p), _p(qo, _f64p), _p(to, _f64p), _p(reg, _f32p), C.byref(st)))
        return qo, to, st, reg

    def process_features(self, feat, q_wodom, t_wodom):
        q = _f64(q_wodom)
        t = _f64(t_wodom)
        qo = np.zeros(4)
        to = np.zeros(3)
        st = MapStats()
        _check(lib().scal_map_step_features(self.h, feat.h, _p(q, _f64p), _p(t, _f64p), _p(qo, _f64p), _p(to, _f64p), C.byref(st)))
        return qo, to, st

    def export(self, which):
        n = lib().scal_map_export(self.h, which, None, 0)
        if n < 0:
            _check(n)
        out = np.zeros((max(n, 1), 4), np.float32)
        m = lib().scal_map_export(self.h, which, _p(out, _f32p), n)
        return out[:m]

    def export_all(self, which):
        """every point of the 21x21x11 cube grid, one feature class: the content of /laser_cloud_map (laserMapping.cpp:824-837)"""
        n = lib().scal_map_export_all(self.h, which, None, 0)
        if n < 0:
            _check(n)
        out = np.zeros((max(n, 1), 4), np.float32)
        m = lib().scal_map_export_all(self.h, which, _p(out, _f32p), n)
        if m < 0:
            _check(m)
        return out[:m]

    def enqueue_features(self, feat, q_wodom, t_wodom):
        _check(lib().scal_map_enqueue_features(self.h, feat.h, _p(_f64(q_wodom), _f64p), _p(_f64(t_wodom), _f64p)))

    def collect(self):
        qo, to, st = np.zeros(4), np.zeros(3), MapStats()
        _check(lib().scal_map_collect(self.h, _p(qo, _f64p), _p(to, _f64p), C.byref(st)))
        return qo, to, st

    def finish(self):
        _check(lib().scal_map_finish(self.h))

    def prefetch_features(self, feat):
        _check(lib().scal_map_prefetch_features(self.h, feat.h))

    # ---- Ceres-adapter mode (the caller owns the solver): begin -> per outer iteration associate / blocks / eval -> finish
    def adapter_begin(self, corner_last, surf_last, full_res, q_wodom, t_wodom):
        c = _f32(corner_last).reshape(-1, 4)
        s = _f32(surf_last).reshape(-1, 4)
        f = _f32(full_res).reshape(-1, 4) if full_res is not None else None
        self._adapter_full = f
        q, t = np.zeros(4), np.zeros(3)
        _check(lib().scal_map_adapter_begin(self.h, _p(c, _f32p), c.shape[0], _p(s, _f32p), s.shape[0], _p(f, _f32p), 0 if f is None else f.shape[0],
                                            _p(_f64(q_wodom), _f64p), _p(_f64(t_wodom), _f64p), _p(q, _f64p), _p(t, _f64p)))
        return q, t

    def associate(self, q_w_curr, t_w_curr):
        nb, nr = C.c_int(0), C.c_int(0)
        _check(lib().scal_map_associate(self.h, _p(_f64(q_w_curr), _f64p), _p(_f64(t_w_curr), _f64p), C.byref(nb), C.byref(nr)))
        self._adapter_n = (nb.value, nr.value)
        return nb.value, nr.value

    def blocks(self):
        """(kind [n], cp [n][3], pa [n][3], pb [n][3]) of the residual blocks of the last associate()"""
        nb = self._adapter_n[0]
        raw = np.zeros((max(nb, 1), 10), np.float64)
        n = lib().scal_map_get_blocks(self.h, raw.ctypes.data, nb)
        if n < 0:
            _check(n)
        raw = raw[:n]
        kind = raw[:, 0].copy().view(np.int32)[::2].copy()
        return kind, raw[:, 1:4].copy(), raw[:, 4:7].copy(), raw[:, 7:10].copy()

    def eval_blocks(self, x7, want_jac=True):
        nb, nr = self._adapter_n
        r = np.zeros(max(nr, 1))
        J = np.zeros((max(nr, 1), 7))
        _check(lib().scal_map_eval_blocks(self.h, _p(_f64(x7), _f64p), 1 if want_jac else 0, _p(r, _f64p), _p(J, _f64p)))
        return r[:nr], (J[:nr] if want_jac else None)

    def adapter_finish(self, q_w_curr, t_w_curr, want_registered=False):
        f = self._adapter_full
        reg = np.zeros_like(f) if (want_registered and f is not None) else None
        st = MapStats()
        _check(lib().scal_map_adapter_finish(self.h, _p(_f64(q_w_curr), _f64p), _p(_f64(t_w_curr), _f64p), _p(reg, _f32p), C.byref(st)))
        return st, reg

    def path_counters(self):
        """(queued speculatively, general path, redone after a moved window, insertion redone with the full sort)"""
        out = (C.c_int * 4)()
        _check(lib().scal_map_get_path_counters(self.h, out))
        return tuple(out)

    def debug_set_lm_polls(self, polls):
        _check(lib().scal_map_debug_set_lm_polls(self.h, polls))

    def debug_set_grid_cap(self, cap_corner, cap_surf):
        _check(lib().scal_map_debug_set_grid_cap(self.h, cap_corner, cap_surf))

    def set_poll(self, enable):
        _check(lib().scal_map_set_poll(self.h, 1 if enable else 0))

    def set_merge_insert(self, enable):
        _check(lib().scal_map_set_merge_insert(self.h, 1 if enable else 0))

    def wmap_wodom(self):
        q = np.zeros(4)
        t = np.zeros(3)
        _check(lib().scal_map_get_wmap_wodom(self.h, _p(q, _f64p), _p(t, _f64p)))
        return q, t


# ---------------------------------------------------------------------------------------------- stage B
class LaserOdometry:
    def __init__(self, max_points=400000, device=0):
        self.cfg = OdomConfig(max_points, device)
        self.h = C.c_void_p()
        _check(lib().scal_odom_create(C.byref(self.cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.scal_odom_destroy(self.h)
        self.h = None

    __del__ = close

    def step(self, sharp, less_sharp, flat, less_flat):
        a = [_f32(x).reshape(-1, 4) for x in (sharp, less_sharp, flat, less_flat)]
        qlc, tlc, qw, tw = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        st = OdomStats()
        _check(lib().scal_odom_step(self.h, _p(a[0], _f32p), a[0].shape[0], _p(a[1], _f32p), a[1].shape[0], _p(a[2], _f32p), a[2].shape[0],
                                    _p(a[3], _f32p), a[3].shape[0], _p(qlc, _f64p), _p(tlc, _f64p), _p(qw, _f64p), _p(tw, _f64p), C.byref(st)))
        return qlc, tlc, qw, tw, st

    def enqueue_features(self, feat):
        _check(lib().scal_odom_enqueue_features(self.h, feat.h))

    def collect(self):
        qlc, tlc, qw, tw = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        st = OdomStats()
        _check(lib().scal_odom_collect(self.h, _p(qlc, _f64p), _p(tlc, _f64p), _p(qw, _f64p), _p(tw, _f64p), C.byref(st)))
        return qlc, tlc, qw, tw, st

    # ---- Ceres-adapter mode (the caller owns the solver)
    def adapter_begin(self, sharp, less_sharp, flat, less_flat):
        a = [_f32(x).reshape(-1, 4) for x in (sharp, less_sharp, flat, less_flat)]
        q, t, need = np.zeros(4), np.zeros(3), C.c_int(0)
        _check(lib().scal_odom_adapter_begin(self.h, _p(a[0], _f32p), a[0].shape[0], _p(a[1], _f32p), a[1].shape[0], _p(a[2], _f32p), a[2].shape[0],
                                             _p(a[3], _f32p), a[3].shape[0], _p(q, _f64p), _p(t, _f64p), C.byref(need)))
        return q, t, bool(need.value)

    def associate(self, q_lc, t_lc):
        nb, nr = C.c_int(0), C.c_int(0)
        _check(lib().scal_odom_associate(self.h, _p(_f64(q_lc), _f64p), _p(_f64(t_lc), _f64p), C.byref(nb), C.byref(nr)))
        self._adapter_n = (nb.value, nr.value)
        return nb.value, nr.value

    def blocks(self):
        nb = self._adapter_n[0]
        raw = np.zeros((max(nb, 1), 10), np.float64)
        n = lib().scal_odom_get_blocks(self.h, raw.ctypes.data, nb)
        if n < 0:
            _check(n)
        raw = raw[:n]
        return raw[:, 0].copy().view(np.int32)[::2].copy(), raw[:, 1:4].copy(), raw[:, 4:7].copy(), raw[:, 7:10].copy()

    def eval_blocks(self, x7, want_jac=True):
        nb, nr = self._adapter_n
        r, J = np.zeros(max(nr, 1)), np.zeros((max(nr, 1), 7))
        _check(lib().scal_odom_eval_blocks(self.h, _p(_f64(x7), _f64p), 1 if want_jac else 0, _p(r, _f64p), _p(J, _f64p)))
        return r[:nr], (J[:nr] if want_jac else None)

    def adapter_finish(self, q_lc, t_lc):
        qw, tw = np.zeros(4), np.zeros(3)
        _check(lib().scal_odom_adapter_finish(self.h, _p(_f64(q_lc), _f64p), _p(_f64(t_lc), _f64p), _p(qw, _f64p), _p(tw, _f64p)))
        return qw, tw

    def step_features(self, feat):
        qlc, tlc, qw, tw = np.zeros(4), np.zeros(3), np.zeros(4), np.zeros(3)
        st = OdomStats()
        _check(lib().scal_odom_step_features(self.h, feat.h, _p(qlc, _f64p), _p(tlc, _f64p), _p(qw, _f64p), _p(tw, _f64p), C.byref(st)))
        return qlc, tlc, qw, tw, st


class LoopICP:
    """pcl::IterativeClosestPoint as doICPVirtualRelative configures it (laserPosegraphOptimization.cpp:518-531)."""

    def __init__(self, max_source=200000, max_target=2000000, max_corr_dist=150.0, max_iterations=100, transformation_epsilon=1e-6,
                 fitness_epsilon=1e-6, device=0):
        self.h = C.c_void_p()
        cfg = ICPConfig(max_corr_dist, transformation_epsilon, fitness_epsilon, max_iterations, max_source, max_target, device)
        _check(lib().scal_icp_create(C.byref(cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.scal_icp_destroy(self.h)
        self.h = None

    __del__ = close

    def align_device(self, d_src_ptr, n_src, d_tgt_ptr, n_tgt):
        r = ICPResult()
        _check(lib().scal_icp_align_device(self.h, d_src_ptr, n_src, d_tgt_ptr, n_tgt, C.byref(r)))
        return dict(converged=bool(r.converged), iterations=r.iterations, state=r.state, n_correspondences=r.n_correspondences,
                    fitness=r.fitness, T=np.array(r.T[:]).reshape(4, 4))

    def set_search(self, mode):
        """1 (default): cell grid + dense sweep for unresolved queries; 0: dense sweep only.  Same result."""
        _check(lib().scal_icp_set_search(self.h, mode))

    def align(self, src, tgt):
        s, t = _f32(src), _f32(tgt)
        r = ICPResult()
        _check(lib().scal_icp_align(self.h, _p(s, _f32p), s.shape[0], _p(t, _f32p), t.shape[0], C.byref(r)))
        return dict(converged=bool(r.converged), iterations=r.iterations, state=r.state, n_correspondences=r.n_correspondences,
                    fitness=r.fitness, T=np.array(r.T[:]).reshape(4, 4))


class MapMerge:
    """Offline dense map merge (utils/python/makeMergedMap.py:83-133): transform, near-range removal, concatenation."""

    def __init__(self, max_points=40000000, max_frame_points=400000, device=0):
        self.h = C.c_void_p()
        cfg = MapMergeConfig(max_points, max_frame_points, device)
        _check(lib().scal_mapmerge_create(C.byref(cfg), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.scal_mapmerge_destroy(self.h)
        self.h = None

    __del__ = close

    def reset(self):
        _check(lib().scal_mapmerge_reset(self.h))

    def add(self, xyzi, pose12, near_thres=2.0):
        a = _f32(xyzi)
        p = _f64(np.asarray(pose12).reshape(-1))
        _check(lib().scal_mapmerge_add(self.h, _p(a, _f32p), a.shape[0], _p(p, _f64p), float(near_thres)))

    def add_batch_device(self, d_ptr, offsets, poses12, near_thres=2.0):
        off = (C.c_int * len(offsets))(*[int(v) for v in offsets])
        p = _f64(np.asarray(poses12).reshape(-1))
        _check(lib().scal_mapmerge_add_batch_device(self.h, d_ptr, off, _p(p, _f64p), len(offsets) - 1, float(near_thres)))

    def size(self):
        n = lib().scal_mapmerge_size(self.h)
        if n < 0:
            _check(int(n))
        return int(n)

    def download(self):
        n = self.size()
        out = np.zeros((max(n, 1), 4), np.float32)
        _check(lib().scal_mapmerge_download(self.h, _p(out, _f32p), n))
        return out[:n]

    def downsample(self, leaf):
        n = self.size()
        out = np.zeros((max(n, 1), 4), np.float32)
        m = C.c_longlong(0)
        _check(lib().scal_mapmerge_downsample(self.h, C.c_float(leaf), _p(out, _f32p), n, C.byref(m)))
        return out[:m.value].copy()

    def device_points(self):
        return lib().scal_mapmerge_device_points(self.h)


def factors_eval(kind, cp, pa, pb, x7, device=0):
    kind = np.ascontiguousarray(kind, np.int32)
    cp, pa, pb, x7 = _f64(cp), _f64(pa), _f64(pb), _f64(x7)
    cost = np.zeros(1)
    g = np.zeros(6)
    H = np.zeros((6, 6))
    _check(lib().scal_factors_eval(device, kind.shape[0], _p(kind, _i32p), _p(cp, _f64p), _p(pa, _f64p), _p(pb, _f64p), _p(x7, _f64p),
                                   _p(cost, _f64p), _p(g, _f64p), _p(H, _f64p)))
    return cost[0], g, H


# ---------------------------------------------------------------------------------------------- the four stages as one object
SC_OFF, SC_EVERY_SCAN, SC_DESCRIPTOR = 0, 1, 2


def _borrow(cls, handle):
    """A wrapper of `cls` around a context that somebody else (a Pipeline) owns: same methods, close() does nothing."""
    sub = type(cls.__name__ + "Borrowed", (cls,), {"close": lambda self: None, "__del__": lambda self: None})
    o = sub.__new__(sub)
    o.h = C.c_void_p(handle)
    return o


class Pipeline:
    """scal_pipeline: the reference's four nodes (scanRegistration, laserOdometry, laserMapping, laserPosegraphOptimization's
    ScanContext part) working on consecutive scans at the same time, scheduled inside the library.  push() a scan, pop() poses in order."""

    def __init__(self, lidar_type, minimum_range, max_points=400000, line_res=0.4, plane_res=0.8, max_map_points=4000000, sc_mode=SC_EVERY_SCAN,
                 sc_max_radius=80.0, sc_dist_thres=0.2, sc_max_keyframes=8192, device=0, ring=0, depth=0, float_math=0, check_finite=1, d_desc_ring=None,
                 n_seqs=1):
        self.cfg = PipelineConfig(lidar_type, SCAN_LINES.get(lidar_type, 0), float(minimum_range), max_points, float_math, check_finite, line_res,
                                  plane_res, max_map_points, sc_max_radius, sc_dist_thres, sc_max_keyframes, sc_mode, device, ring, depth, d_desc_ring)
        self.h = C.c_void_p()
        self.n_seqs = n_seqs
        _check(lib().scal_pipeline_create_multi(C.byref(self.cfg), n_seqs, C.byref(self.h)))
        self.scs = [_borrow(SCManager, lib().scal_pipeline_sc_of(self.h, q)) if sc_mode != SC_OFF else None for q in range(n_seqs)]
        self.maps = [_borrow(LaserMapping, lib().scal_pipeline_map_of(self.h, q)) for q in range(n_seqs)]
        self.sc, self.map = self.scs[0], self.maps[0]
        self.odom = _borrow(LaserOdometry, lib().scal_pipeline_odom(self.h))

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.scal_pipeline_destroy(self.h)
        self.h = None

    __del__ = close

    def push_device(self, d_ptr, n, stride_floats=3):
        _check(lib().scal_pipeline_push_device(self.h, C.c_void_p(d_ptr), n, stride_floats))

    def push(self, xyz):
        a = _f32(xyz)
        _check(lib().scal_pipeline_push_host(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.strides[0]))

    @staticmethod
    def _res(r):
        return dict(seq=r.seq, q=np.array(r.q_w_curr[:]), t=np.array(r.t_w_curr[:]), q_odom=np.array(r.q_odom[:]), t_odom=np.array(r.t_odom[:]),
                    odom=r.odom, map=r.map, loop=SCManager._result(r.loop) if r.have_loop else None, d_descriptor=r.d_descriptor)

    def pop(self):
        r = PipelineResult()
        _check(lib().scal_pipeline_pop(self.h, C.byref(r)))
        return self._res(r)

    def push_device_multi(self, d_ptrs, ns, stride_floats=3):
        """one scan of EVERY sequence (device pointers, point counts)"""
        pa = (C.c_void_p * self.n_seqs)(*[C.c_void_p(int(v)) for v in d_ptrs])
        na = (C.c_int * self.n_seqs)(*[int(v) for v in ns])
        _check(lib().scal_pipeline_push_device_multi(self.h, pa, na, stride_floats))

    def pop_multi(self):
        """the sequences' results of the oldest scan step not yet popped: a list of n_seqs dicts"""
        ra = (PipelineResult * self.n_seqs)()
        _check(lib().scal_pipeline_pop_multi(self.h, ra))
        return [self._res(ra[q]) for q in range(self.n_seqs)]

    def drain(self):
        _check(lib().scal_pipeline_drain(self.h))

    def in_flight(self):
        return lib().scal_pipeline_in_flight(self.h)

    def features(self, i):
        return _borrow(ScanRegistration, lib().scal_pipeline_features(self.h, i))
