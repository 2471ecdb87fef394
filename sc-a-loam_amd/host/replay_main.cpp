// replay_main: a ROS-free C++ host of libscaloam_hip.so.  It includes include/scaloam_hip.h, links the shared library and runs a
// recorded scan sequence through the hot path the way a C++ maintainer of the reference would:
//
//   --mode integrated  "as integrated": the SYNCHRONOUS host-array entry points exactly as INTEGRATION.md sections 1-4 (and the node
//                      shells next to this file) issue them - scal_features_run, scal_odom_step, scal_map_step,
//                      scal_voxel_downsample + scal_sc_insert_cloud + scal_sc_detect - one thread per stage, clouds handed from
//                      thread to thread as host arrays (the reference: four processes, ROS topics with queue size 100;
//                      scanRegistration.cpp:475-517, laserOdometry.cpp:186-600, laserMapping.cpp:909-952,
//                      laserPosegraphOptimization.cpp:874-906).  Every scan goes through every stage (no drop rule).
//   --mode pipeline    scal_pipeline_*: the same four stages scheduled inside the library, scans uploaded from host memory
//                      (--resident 1: copied to the GPU before the timed region, as bench.py's headline does).
//   --mode serial      one scan at a time through the device-resident per-stage calls (the parity schedule).
//
// Input: a scan stream file written by scaloam.formats.write_scan_stream (bench.py, tests): "SCALSCN1", int32 count, then per scan
// int32 n + n x 3 float32 xyz.  Output: one JSON line on stdout; --poses FILE writes "seq qx qy qz qw tx ty tz loop_id" per scan with
// 17 significant digits so that a test can compare them bit for bit with the Python path.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include "scaloam_hip.h"

namespace {

using Clock = std::chrono::steady_clock;
double now_s() { return std::chrono::duration<double>(Clock::now().time_since_epoch()).count(); }

#define CHECK(e)                                                                          \
    do {                                                                                  \
        int rc_ = (e);                                                                    \
        if (rc_ != SCAL_OK) {                                                             \
            std::fprintf(stderr, "%s failed (%d): %s\n", #e, rc_, scal_last_error());      \
            std::exit(2);                                                                 \
        }                                                                                 \
    } while (0)

struct Scan {
    std::vector<float> xyz;  // n x 3
    int n() const { return static_cast<int>(xyz.size() / 3); }
};

std::vector<Scan> read_stream(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path.c_str());
        std::exit(2);
    }
    char magic[8];
    int32_t count = 0;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "SCALSCN1", 8) != 0 || std::fread(&count, 4, 1, f) != 1 || count < 0) {
        std::fprintf(stderr, "%s is not a scan stream file\n", path.c_str());
        std::exit(2);
    }
    std::vector<Scan> out(count);
    for (auto& s : out) {
        int32_t n = 0;
        if (std::fread(&n, 4, 1, f) != 1 || n < 0) std::exit(2);
        s.xyz.resize(static_cast<size_t>(n) * 3);
        if (n && std::fread(s.xyz.data(), sizeof(float) * 3, n, f) != static_cast<size_t>(n)) std::exit(2);
    }
    std::fclose(f);
    return out;
}

// splitmix64, the generator of the synthetic inputs elsewhere in this repository
struct SplitMix {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};

// ScanContext-like descriptors for the pre-filled database (occupancy ~0.5, heights -2..18 m): only their count matters for timing
void prefill_sc(scal_sc_t* sc, int n, uint64_t seed) {
    SplitMix g{seed};
    std::vector<double> d(1200);
    for (int i = 0; i < n; ++i) {
        for (auto& v : d) v = g.uni() < 0.5 ? -2.0 + 20.0 * g.uni() : 0.0;
        CHECK(scal_sc_insert_descriptor(sc, d.data()));
    }
}

struct Pose {
    double q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0};
    int loop_id = -1;
    double t_in = 0, t_out = 0;
};

template <class T>
struct Channel {  // a ROS topic between two nodes: FIFO, bounded like the reference's queue size 100
    std::mutex mu;
    std::condition_variable cv;
    std::deque<T> q;
    bool closed = false;
    void put(T v) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return q.size() < 100; });
        q.push_back(std::move(v));
        cv.notify_all();
    }
    bool get(T& v) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front());
        q.pop_front();
        cv.notify_all();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> lk(mu);
        closed = true;
        cv.notify_all();
    }
};

struct Args {
    std::string scans, mode = "pipeline", poses;
    int lidar = SCAL_HDL64, n_scans = 64, warmup = 5, sc_db = 0, resident = 0, max_points = 0, steps = 0, sc = 1, ahead = 4, stream_mode = 0;
    double min_range = 5.0, sc_thres = 0.4;
    float line = 0.4f, plane = 0.8f;
};

struct FeatMsg {  // the five clouds scanRegistration publishes (:426-454), as host arrays
    int seq = 0;
    double t_in = 0;
    std::vector<float> cloud, sharp, less_sharp, flat, less_flat;
};
struct OdomMsg {  // what laserOdometry publishes for laserMapping (:570-591): corner_last, surf_last, full-res cloud, odometry pose
    int seq = 0;
    double t_in = 0;
    std::vector<float> corner_last, surf_last, full;
    double q[4], t[3];
};
struct MapMsg {  // /aft_mapped_to_init + the keyframe cloud laserPosegraphOptimization consumes (/velodyne_cloud_registered_local)
    int seq = 0;
    double t_in = 0;
    std::vector<float> local;
    Pose pose;
};

void gather(const std::vector<float>& cloud, const int* idx, int n, std::vector<float>& out) {
    out.resize(static_cast<size_t>(n) * 4);
    for (int k = 0; k < n; ++k) std::memcpy(&out[4 * k], &cloud[4 * static_cast<size_t>(idx[k])], 16);
}

// ------------------------------------------------------------------------------------------------ as integrated
double g_stage_s[4] = {0, 0, 0, 0};  // seconds inside the library calls of nodes A..D (integrated mode; reset by main before the timed run)
struct StageClock {
    double& acc;
    double t0;
    explicit StageClock(int k) : acc(g_stage_s[k]), t0(now_s()) {}
    ~StageClock() { acc += now_s() - t0; }
};
std::vector<Pose> run_integrated(const Args& a, const std::vector<Scan>& scans, int first, int last) {
    static scal_features_t* g_feat = nullptr;
    static scal_odom_t* g_odom = nullptr;
    static scal_map_t* g_map = nullptr;
    static scal_sc_t* g_sc = nullptr;
    static scal_voxel_t* g_vox = nullptr;
    const int cap = a.max_points;
    if (!g_feat) {
        scal_features_config fc{};
        fc.lidar_type = a.lidar, fc.n_scans = a.n_scans, fc.minimum_range = a.min_range, fc.max_points = cap, fc.check_finite = 1;
        CHECK(scal_features_create(&fc, &g_feat));
        scal_odom_config oc{cap, 0};
        CHECK(scal_odom_create(&oc, &g_odom));
        scal_map_config mc{a.line, a.plane, cap, 4000000, 0};
        CHECK(scal_map_create(&mc, &g_map));
        if (a.sc) {
            scal_sc_config sc{};
            sc.max_radius = 80.0, sc.dist_thres = a.sc_thres, sc.max_keyframes = a.sc_db + static_cast<int>(scans.size()) + 64, sc.n_shards = 1;
            CHECK(scal_sc_create(&sc, &g_sc));
            prefill_sc(g_sc, a.sc_db, 4242);
            CHECK(scal_voxel_create(cap, 0, &g_vox));
        }
    }
    std::vector<Pose> out(last - first);
    Channel<FeatMsg> ch_ab;
    Channel<OdomMsg> ch_bc;
    Channel<MapMsg> ch_cd;
    // scanRegistration: laserCloudHandler
    std::thread ta([&] {
        std::vector<float> cloud(static_cast<size_t>(cap) * 4), less_flat(static_cast<size_t>(cap) * 4);
        std::vector<int> sharp(12 * a.n_scans), less(120 * a.n_scans), flat(24 * a.n_scans);
        for (int k = first; k < last; ++k) {
            FeatMsg m;
            m.seq = k, m.t_in = now_s();
            scal_features_out o{};
            o.cloud = cloud.data(), o.sharp = sharp.data(), o.less_sharp = less.data(), o.flat = flat.data(), o.less_flat = less_flat.data();
            {
                StageClock clk(0);
                CHECK(scal_features_run(g_feat, scans[k].xyz.data(), scans[k].n(), 12, &o));
            }
            m.cloud.assign(cloud.begin(), cloud.begin() + static_cast<size_t>(o.n_kept) * 4);
            gather(m.cloud, sharp.data(), o.n_sharp, m.sharp);
            gather(m.cloud, less.data(), o.n_less_sharp, m.less_sharp);
            gather(m.cloud, flat.data(), o.n_flat, m.flat);
            m.less_flat.assign(less_flat.begin(), less_flat.begin() + static_cast<size_t>(o.n_less_flat) * 4);
            ch_ab.put(std::move(m));
        }
        ch_ab.close();
    });
    // laserOdometry: main loop body (section 2)
    std::thread tb([&] {
        FeatMsg m;
        while (ch_ab.get(m)) {
            OdomMsg o;
            o.seq = m.seq, o.t_in = m.t_in;
            double q_lc[4], t_lc[3];
            {
                StageClock clk(1);
                CHECK(scal_odom_step(g_odom, m.sharp.data(), static_cast<int>(m.sharp.size() / 4), m.less_sharp.data(), static_cast<int>(m.less_sharp.size() / 4),
                                     m.flat.data(), static_cast<int>(m.flat.size() / 4), m.less_flat.data(), static_cast<int>(m.less_flat.size() / 4), q_lc, t_lc,
                                     o.q, o.t, nullptr));
            }
            o.corner_last = std::move(m.less_sharp), o.surf_last = std::move(m.less_flat), o.full = std::move(m.cloud);
            ch_bc.put(std::move(o));
        }
        ch_bc.close();
    });
    // laserMapping: process() (section 3)
    std::thread tc([&] {
        OdomMsg m;
        while (ch_bc.get(m)) {
            MapMsg o;
            o.seq = m.seq, o.t_in = m.t_in;
            std::vector<float> reg(m.full.size());
            scal_map_stats st;
            {
                StageClock clk(2);
                CHECK(scal_map_step(g_map, m.corner_last.data(), static_cast<int>(m.corner_last.size() / 4), m.surf_last.data(),
                                    static_cast<int>(m.surf_last.size() / 4), m.full.data(), static_cast<int>(m.full.size() / 4), m.q, m.t, o.pose.q, o.pose.t,
                                    reg.data(), &st));
            }
            o.pose.t_in = m.t_in, o.pose.t_out = now_s();
            o.local = std::move(m.full);  // /velodyne_cloud_registered_local (:839-843): the scan in the sensor frame
            ch_cd.put(std::move(o));
        }
        ch_cd.close();
    });
    // laserPosegraphOptimization, ScanContext part (section 4): VoxelGrid 0.4 (:629-631), insert (:639), detect (:718)
    std::thread td([&] {
        MapMsg m;
        std::vector<float> ds(static_cast<size_t>(cap) * 4);
        while (ch_cd.get(m)) {
            Pose p = m.pose;
            if (a.sc) {
                StageClock clk(3);
                int n_ds = 0;
                CHECK(scal_voxel_downsample(g_vox, m.local.data(), static_cast<int>(m.local.size() / 4), 0.4f, ds.data(), &n_ds));
                CHECK(scal_sc_insert_cloud(g_sc, ds.data(), n_ds));
                scal_sc_result r;
                CHECK(scal_sc_detect(g_sc, &r));
                p.loop_id = r.loop_id;
            }
            out[m.seq - first] = p;
        }
    });
    ta.join(), tb.join(), tc.join(), td.join();
    return out;
}

// ------------------------------------------------------------------------------------------------ scal_pipeline
scal_pipeline_t* g_pipe = nullptr;
std::vector<float*> g_resident;

void make_pipeline(const Args& a, const std::vector<Scan>& scans) {
    scal_pipeline_config pc{};
    pc.lidar_type = a.lidar, pc.n_scans = a.n_scans, pc.minimum_range = a.min_range, pc.max_points = a.max_points, pc.check_finite = 1;
    pc.line_res = a.line, pc.plane_res = a.plane, pc.max_map_points = 4000000;
    pc.sc_max_radius = 80.0, pc.sc_dist_thres = a.sc_thres, pc.sc_max_keyframes = a.sc_db + static_cast<int>(scans.size()) + 64;
    pc.sc_mode = a.sc ? SCAL_PIPE_SC_EVERY_SCAN : SCAL_PIPE_SC_OFF;
    CHECK(scal_pipeline_create(&pc, &g_pipe));
    if (a.sc) prefill_sc(scal_pipeline_sc(g_pipe), a.sc_db, 4242);
    if (a.resident) {
        for (const auto& s : scans) {
            float* d = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&d), std::max<size_t>(s.xyz.size(), 3) * sizeof(float)) != hipSuccess ||
                hipMemcpy(d, s.xyz.data(), s.xyz.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                std::fprintf(stderr, "hipMalloc / hipMemcpy of a resident scan failed\n");
                std::exit(2);
            }
            g_resident.push_back(d);
        }
    }
}

std::vector<Pose> run_pipeline(const Args& a, const std::vector<Scan>& scans, int first, int last) {
    if (!g_pipe) make_pipeline(a, scans);
    std::vector<Pose> out(last - first);
    const int ahead = a.ahead;  // scans pushed beyond the one whose pose is awaited
    int popped = first;
    auto pop_one = [&] {
        scal_pipeline_result r;
        CHECK(scal_pipeline_pop(g_pipe, &r));
        Pose& p = out[popped - first];
        std::memcpy(p.q, r.q_w_curr, sizeof p.q), std::memcpy(p.t, r.t_w_curr, sizeof p.t);
        p.loop_id = r.have_loop ? r.loop.loop_id : -1;
        p.t_out = now_s();
        ++popped;
    };
    for (int k = first; k < last; ++k) {
        out[k - first].t_in = now_s();
        if (a.resident) CHECK(scal_pipeline_push_device(g_pipe, g_resident[k], scans[k].n(), 3));
        else CHECK(scal_pipeline_push_host(g_pipe, scans[k].xyz.data(), scans[k].n(), 12));
        if (k - popped >= ahead) pop_one();
    }
    CHECK(scal_pipeline_drain(g_pipe));  // the last scan's map insertion belongs to the work
    while (popped < last) pop_one();
    return out;
}

// ------------------------------------------------------------------------------------------------ serial, device-resident stage calls
std::vector<Pose> run_serial(const Args& a, const std::vector<Scan>& scans, int first, int last) {
    static scal_features_t* feat = nullptr;
    static scal_odom_t* odom = nullptr;
    static scal_map_t* map = nullptr;
    static scal_sc_t* sc = nullptr;
    if (!feat) {
        scal_features_config fc{};
        fc.lidar_type = a.lidar, fc.n_scans = a.n_scans, fc.minimum_range = a.min_range, fc.max_points = a.max_points, fc.check_finite = 1;
        CHECK(scal_features_create(&fc, &feat));
        scal_odom_config oc{a.max_points, 0};
        CHECK(scal_odom_create(&oc, &odom));
        scal_map_config mc{a.line, a.plane, a.max_points, 4000000, 0};
        CHECK(scal_map_create(&mc, &map));
        if (a.sc) {
            scal_sc_config c{};
            c.max_radius = 80.0, c.dist_thres = a.sc_thres, c.max_keyframes = a.sc_db + static_cast<int>(scans.size()) + 64, c.n_shards = 1;
            CHECK(scal_sc_create(&c, &sc));
            prefill_sc(sc, a.sc_db, 4242);
        }
    }
    std::vector<Pose> out(last - first);
    for (int k = first; k < last; ++k) {
        Pose& p = out[k - first];
        p.t_in = now_s();
        CHECK(scal_features_run(feat, scans[k].xyz.data(), scans[k].n(), 12, nullptr));
        double qlc[4], tlc[3], qo[4], to[3];
        CHECK(scal_odom_step_features(odom, feat, qlc, tlc, qo, to, nullptr));
        CHECK(scal_map_step_features(map, feat, qo, to, p.q, p.t, nullptr));
        p.t_out = now_s();
        if (a.sc) {
            CHECK(scal_sc_insert_features(sc, feat));
            scal_sc_result r;
            CHECK(scal_sc_detect(sc, &r));
            p.loop_id = r.loop_id;
        }
    }
    return out;
}

}  // namespace

int main(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "%s needs a value\n", k.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (k == "--scans") a.scans = val();
        else if (k == "--mode") a.mode = val();
        else if (k == "--poses") a.poses = val();
        else if (k == "--warmup") a.warmup = std::atoi(val());
        else if (k == "--steps") a.steps = std::atoi(val());
        else if (k == "--sc-db") a.sc_db = std::atoi(val());
        else if (k == "--sc") a.sc = std::atoi(val());
        else if (k == "--sc-thres") a.sc_thres = std::atof(val());
        else if (k == "--resident") a.resident = std::atoi(val());
        else if (k == "--ahead") a.ahead = std::max(0, std::min(24, std::atoi(val())));
        else if (k == "--stream-mode") a.stream_mode = std::atoi(val()) ? 1 : 0;
        else if (k == "--min-range") a.min_range = std::atof(val());
        else if (k == "--line") a.line = static_cast<float>(std::atof(val()));
        else if (k == "--plane") a.plane = static_cast<float>(std::atof(val()));
        else if (k == "--lidar") {
            const std::string v = val();
            if (v == "vlp16") a.lidar = SCAL_VLP16, a.n_scans = 16;
            else if (v == "hdl32") a.lidar = SCAL_HDL32, a.n_scans = 32;
            else if (v == "hdl64") a.lidar = SCAL_HDL64, a.n_scans = 64;
            else if (v == "os1") a.lidar = SCAL_OS1_64, a.n_scans = 64;
            else {
                std::fprintf(stderr, "unknown --lidar %s\n", v.c_str());
                return 2;
            }
        } else {
            std::fprintf(stderr, "usage: replay_main --scans FILE [--mode pipeline|integrated|serial] [--lidar hdl64|vlp16|hdl32|os1] [--min-range m]\n"
                                 "                   [--line m --plane m] [--sc 0|1] [--sc-db n] [--sc-thres d] [--warmup n] [--steps n] [--resident 0|1] [--ahead n] [--stream-mode 0|1] [--poses FILE]\n"
                                 "  --stream-mode 1 (integrated / serial modes): every stage context on its own stream, as four separate node processes have it\n");
            return 2;
        }
    }
    if (a.scans.empty()) {
        std::fprintf(stderr, "--scans FILE is required\n");
        return 2;
    }
    if (scal_device_count() < 1) {
        std::fprintf(stderr, "replay_main needs a GPU: libscaloam_hip has no CPU fallback\n");
        return 3;
    }
    const std::vector<Scan> scans = read_stream(a.scans);
    const int total = static_cast<int>(scans.size());
    int most = 0;
    for (const auto& s : scans) most = std::max(most, s.n());
    a.max_points = std::min(400000, most + 1024);
    const int W = std::min(a.warmup, total), K = a.steps > 0 ? std::min(a.steps, total - W) : total - W;
    auto run = a.mode == "integrated" ? run_integrated : a.mode == "serial" ? run_serial : a.mode == "pipeline" ? run_pipeline : nullptr;
    if (!run) {
        std::fprintf(stderr, "unknown --mode %s\n", a.mode.c_str());
        return 2;
    }
    if (a.mode != "pipeline") CHECK(scal_set_stream_mode(a.stream_mode));  // the pipeline object picks its own streams
    std::vector<Pose> all = run(a, scans, 0, W);  // warm-up: the same contexts carry on (map, poses, database)
    (void)hipDeviceSynchronize();
    for (double& v : g_stage_s) v = 0;
    const double t0 = now_s();
    std::vector<Pose> timed = run(a, scans, W, W + K);
    (void)hipDeviceSynchronize();
    const double dt = now_s() - t0;
    all.insert(all.end(), timed.begin(), timed.end());
    std::vector<double> lat;
    for (const auto& p : timed) lat.push_back((p.t_out - p.t_in) * 1e3);
    std::sort(lat.begin(), lat.end());
    auto pct = [&](double f) { return lat.empty() ? 0.0 : lat[std::min(lat.size() - 1, static_cast<size_t>(f * lat.size()))]; };
    int loops = 0;
    for (const auto& p : timed) loops += p.loop_id >= 0;
    if (!a.poses.empty()) {
        FILE* f = std::fopen(a.poses.c_str(), "w");
        if (!f) return 2;
        for (size_t i = 0; i < all.size(); ++i)
            std::fprintf(f, "%zu %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d\n", i, all[i].q[0], all[i].q[1], all[i].q[2], all[i].q[3], all[i].t[0], all[i].t[1],
                         all[i].t[2], all[i].loop_id);
        std::fclose(f);
    }
    const Pose& fp = all.back();
    std::printf("{\"mode\": \"%s\", \"scans\": %d, \"warmup\": %d, \"seconds\": %.6f, \"scans_per_s\": %.3f, \"ms_per_scan\": %.6f, "
                "\"latency_ms\": {\"p50\": %.4f, \"p99\": %.4f}, \"loops_detected\": %d, \"sc_db\": %d, \"resident\": %d, \"stream_mode\": %d, "
                "\"final_map_pose\": {\"q\": [%.17g, %.17g, %.17g, %.17g], \"t\": [%.17g, %.17g, %.17g]}, "
                "\"node_call_ms\": {\"A\": %.4f, \"B\": %.4f, \"C\": %.4f, \"D\": %.4f}, \"library\": \"%s\"}\n",
                a.mode.c_str(), K, W, dt, K / dt, dt / K * 1e3, pct(0.5), pct(0.99), loops, a.sc_db, a.resident, a.stream_mode, fp.q[0], fp.q[1], fp.q[2], fp.q[3], fp.t[0],
                fp.t[1], fp.t[2], g_stage_s[0] / K * 1e3, g_stage_s[1] / K * 1e3, g_stage_s[2] / K * 1e3, g_stage_s[3] / K * 1e3, scal_version());
    std::fflush(stdout);
    if (g_pipe) scal_pipeline_destroy(g_pipe);  // joins its host threads; the per-stage contexts of the other modes are left to process exit
    return 0;
}
