// scal_pipeline: the reference's four concurrently running nodes (scanRegistration.cpp:475-517, laserOdometry.cpp:186-600,
// laserMapping.cpp:909-952, laserPosegraphOptimization.cpp:874-906) as one object on one GPU.  See include/scaloam_hip.h.
//
// The schedule (what round 2 kept in bench.py's Python threads) lives here, on five host threads that only ever queue work:
//   front   stage A of scan k on features context k % ring, then the lane-0 half of stage C's pose-independent prefetch (input gather
//           + corner stack filter); a features context is rewritten only when the scan that used it last has left stage C (pose
//           collected), has been queued into stage B and has been handed to ScanContext (the library's reader events then order the
//           device side);
//   side    the side-lane half of the prefetch (surf stack filter) and ScanContext's keyframe filter + descriptor + insert + search;
//   odom    stage B is queued up to two scans ahead of the pose it hands to stage C, its poses are collected (the only thing this
//           thread ever waits for);
//   pose    a collected odometry pose goes straight into stage C's enqueue, which queues behind the stage-C steps still running on
//           the device (`depth` uncollected); finished steps are collected;
//   loop    ScanContext's answers, collected one scan behind their searches.
// Nothing here touches the device except through the per-stage C-ABI, so the poses are those of the per-stage calls.
#include "common.hpp"
#include <condition_variable>
#include <chrono>
#include <cstdlib>
#include <string>
#include <thread>

namespace scal {
void print_forced_flushes();
}
using namespace scal;

namespace {
constexpr int REC_N = 64;      // per-scan records (ring); at most 32 scans may be pushed and not popped
constexpr int MAX_UNPOPPED = 32;
// development knobs (environment), read once: how far the stages may run ahead of each other
int env_int(const char* name, int dflt, int lo, int hi) {
    const char* e = std::getenv(name);
    if (!e) return dflt;
    const int v = std::atoi(e);
    return v < lo ? lo : (v > hi ? hi : v);
}
const int B_AHEAD = env_int("SCALOAM_PIPE_B_AHEAD", 2, 1, 3);     // stage-B steps queued and not collected
const int PF_AHEAD = env_int("SCALOAM_PIPE_PF_AHEAD", 3, 1, 3);   // scal_map::MAX_PF: prefetches queued ahead of their steps

// SCALOAM_PIPE_TIMING=1: host time spent inside each stage call, printed per scan when the pipeline is destroyed (development aid)
const bool TIMING = env_int("SCALOAM_PIPE_TIMING", 0, 0, 1) != 0;
struct HostTimer {
    const char* name[12] = {};
    double total[12] = {};
    long calls[12] = {};
    template <class F>
    int run(int slot, const char* nm, F&& f) {
        if (!TIMING) return f();
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = f();
        name[slot] = nm;
        total[slot] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        calls[slot]++;
        return rc;
    }
    void print(const char* thread) const {
        for (int i = 0; i < 12; ++i)
            if (calls[i]) std::fprintf(stderr, "[scal_pipeline] %-6s %-28s %8.1f us/call x %ld\n", thread, name[i], total[i] / calls[i], calls[i]);
    }
};

constexpr int SMAX = scal::BATCH_MAX;  // sequences that share launches
struct Rec {
    const float* d_xyz[SMAX] = {};
    const void* h_xyz[SMAX] = {};
    int n[SMAX] = {};
    int stride = 0;
    bool host = false;
    scal_pipeline_result res[SMAX] = {};
};
}  // namespace

struct scal_pipeline {
    scal_pipeline_config cfg{};
    int ring = 6, depth = 2;
    int S = 1;  // independent sequences stepping together; S > 1: their per-stage calls are recorded and zipped (batch.hpp)
    std::vector<scal_features_t*> regs[SMAX];
    scal_odom_t* od[SMAX] = {};
    scal_map_t* mp[SMAX] = {};
    scal_sc_t* sc[SMAX] = {};
    double* d_desc = nullptr;  // sc_mode 2: ring x 1200 doubles (S = 1 only)
    bool own_desc = false;
    Rec rec[REC_N];
    std::mutex mu;
    std::condition_variable cv;
    // monotone counters, all guarded by mu: scan k is "done" for a stage when counter > k
    long long pushed = 0, a_run = 0, a_done = 0, pf_done = 0, b_enq = 0, b_coll = 0, c_enq = 0, c_coll = 0, d_ins = 0, d_coll = 0, popped = 0;
    int pop_waiting = 0;
    bool drain_req = false, drained = true;
    bool stop = false;
    int err = SCAL_OK;
    std::string errmsg;
    std::thread t_front, t_side, t_odom, t_pose, t_loop;
    HostTimer tm_front, tm_side, tm_odom, tm_pose, tm_loop;

    bool sc_on() const { return cfg.sc_mode != SCAL_PIPE_SC_OFF; }
    Rec& r(long long k) { return rec[k % REC_N]; }
    // one per-stage call for every sequence.  S = 1: the call itself.  S > 1: each sequence's call runs under a recorder (nothing
    // reaches the device), then the S lists are zipped: same kernel, same shape -> one launch with gridDim.z = S.
    template <class F>
    int for_all(F&& f) {
        if (S == 1) return f(0);
        Recorder recs[SMAX];
        Recorder* ptr[SMAX];
        int rc = SCAL_OK;
        for (int q = 0; q < S && rc == SCAL_OK; ++q) {
            ptr[q] = &recs[q];
            g_recorder = &recs[q];
            rc = f(q);
            g_recorder = nullptr;
        }
        if (rc != SCAL_OK) {  // what the earlier sequences queued still has to run: their host state already counts on it
            for (int q = 0; q < S; ++q) (void)recs[q].flush();
            return rc;
        }
        if (zip_and_launch(ptr, S) != hipSuccess) {
            set_error("scal_pipeline: a batched launch failed: %s", hipGetErrorString(hipGetLastError()));
            return SCAL_E_HIP;
        }
        return SCAL_OK;
    }
    // calls that wait for the device: one after the other
    template <class F>
    int for_each(F&& f) {
        for (int q = 0; q < S; ++q) SCAL_TRY(f(q));
        return SCAL_OK;
    }
    void fail(int rc) {  // called with mu held, from the thread whose stage call failed
        if (err == SCAL_OK) {
            err = rc;
            errmsg = scal_last_error();
        }
        cv.notify_all();
    }
};

namespace {

// Lane 0's thread: stage A of scan k and, right behind it on the same stream, the first half of stage C's prefetch (input gather +
// corner stack filter).
void front_thread(scal_pipeline* p) {
    std::unique_lock<std::mutex> lk(p->mu);
    for (;;) {
        const long long k = p->a_done;
        p->cv.wait(lk, [&] {
            if (p->stop || p->err) return true;
            if (p->pushed <= k) return false;
            const long long old = k - p->ring;  // the scan that used this features context last
            if (old >= 0 && (p->c_coll <= old || p->b_enq <= old || (p->sc_on() && p->d_ins <= old))) return false;
            return k - p->c_enq < PF_AHEAD;     // prefetches queued ahead of their stage-C steps
        });
        if (p->stop || p->err) return;
        const Rec& rc = p->r(k);
        const int slot = static_cast<int>(k % p->ring);
        lk.unlock();
        int st = p->tm_front.run(0, "A: features_run", [&] {
            return p->for_all([&](int q) {
                scal_features_t* reg = p->regs[q][slot];
                return rc.host ? scal_features_enqueue_host(reg, rc.h_xyz[q], rc.n[q], rc.stride) : scal_features_run_device(reg, rc.d_xyz[q], rc.n[q], rc.stride);
            });
        });
        if (st == SCAL_OK) {
            lk.lock();
            p->a_run = k + 1;  // stage B may be queued behind stage A at once
            p->cv.notify_all();
            lk.unlock();
            st = p->tm_front.run(1, "C: prefetch, lane 0 half", [&] { return p->for_all([&](int q) { return scal_map_prefetch_begin(p->mp[q], p->regs[q][slot]); }); });
        }
        lk.lock();
        if (st != SCAL_OK) {
            p->fail(st);
            return;
        }
        p->a_done = k + 1;
        p->cv.notify_all();
    }
}

// The side lane's thread: everything else that only needs stage A - the second half of stage C's prefetch (surf stack filter) and
// ScanContext's keyframe filter + descriptor + insert + search.  The two big voxel filters - same kernels, same stream, different
// clouds - are recorded and SHARE their launches (zip_marked_and_launch): ~200 us of side-stream time per scan become ~110.  With
// several sequences all 2 S filter runs are merged the same way.  (One thread issued lane 0's and the side lane's launches until the
// chains got short enough for its 130-220 us of launch calls per scan to be the longest thing in the period.)
void side_thread(scal_pipeline* p) {
    std::unique_lock<std::mutex> lk(p->mu);
    const bool search = p->cfg.sc_mode == SCAL_PIPE_SC_EVERY_SCAN;
    for (;;) {
        const long long k = p->pf_done;
        p->cv.wait(lk, [&] {
            if (p->stop || p->err) return true;
            if (p->a_done <= k) return false;
            return !(p->sc_on() && k - p->d_coll >= 3);  // the ScanContext context holds four searches / descriptors in flight
        });
        if (p->stop || p->err) return;
        const int slot = static_cast<int>(k % p->ring);
        lk.unlock();
        double* dd = p->d_desc ? p->d_desc + static_cast<size_t>(slot) * 1200 : nullptr;
        int st = p->tm_side.run(1, "C: prefetch, side half + D: insert", [&]() -> int {
            if (!p->sc_on() && p->S == 1) return scal_map_prefetch_finish(p->mp[0], p->regs[0][slot]);
            Recorder recs[2 * SMAX];
            Recorder* ptr[2 * SMAX];
            int nr = 0, rc2 = SCAL_OK;
            rc2 = p->tm_side.run(3, "   record prefetch", [&] {
                int r3 = SCAL_OK;
                for (int q = 0; q < p->S && r3 == SCAL_OK; ++q) {
                    g_recorder = ptr[nr] = &recs[nr];
                    ++nr;
                    r3 = scal_map_prefetch_finish(p->mp[q], p->regs[q][slot]);
                    g_recorder = nullptr;
                }
                return r3;
            });
            if (rc2 == SCAL_OK && p->sc_on())
                rc2 = p->tm_side.run(4, "   record insert", [&] {
                    int r3 = SCAL_OK;
                    for (int q = 0; q < p->S && r3 == SCAL_OK; ++q) {
                        g_recorder = ptr[nr] = &recs[nr];
                        ++nr;
                        r3 = search ? scal_sc_insert_features(p->sc[q], p->regs[q][slot]) : scal_sc_make_features_enqueue(p->sc[q], p->regs[q][slot], dd);
                        g_recorder = nullptr;
                    }
                    return r3;
                });
            if (rc2 != SCAL_OK) {  // what was recorded still has to run: the contexts' host state already counts on it
                for (int i = 0; i < nr; ++i) (void)recs[i].flush();
                return rc2;
            }
            return p->tm_side.run(5, "   zip + launch", [&]() -> int {
                if (zip_marked_and_launch(ptr, nr) != hipSuccess) {
                    set_error("scal_pipeline: a batched launch failed: %s", hipGetErrorString(hipGetLastError()));
                    return SCAL_E_HIP;
                }
                return SCAL_OK;
            });
        });
        if (st == SCAL_OK && search) st = p->tm_side.run(2, "D: sc_detect_enqueue", [&] { return p->for_all([&](int q) { return scal_sc_detect_enqueue(p->sc[q]); }); });
        lk.lock();
        if (st != SCAL_OK) {
            p->fail(st);
            return;
        }
        p->r(k).res[0].d_descriptor = dd;
        p->pf_done = k + 1;
        if (p->sc_on()) p->d_ins = k + 1;
        p->cv.notify_all();
    }
}

// Stage B's thread: queues the odometry steps (up to B_AHEAD in front of the collected ones) and collects their poses.  It only ever
// waits for stage B: while the two stages shared a thread, the wait for a pose (100 us) kept finished stage-C steps from being collected
// and the caller's next push from starting.
void odom_thread(scal_pipeline* p) {
    std::unique_lock<std::mutex> lk(p->mu);
    for (;;) {
        enum { NONE, B_ENQ, B_COLL } what = NONE;
        p->cv.wait(lk, [&] {
            if (p->stop || p->err) return true;
            if (p->b_enq < p->a_run && p->b_enq - p->b_coll < B_AHEAD) { what = B_ENQ; return true; }
            // a pose is only collected when stage C has room to take it soon: the record of an uncollected step is not reused meanwhile
            if (p->b_coll < p->b_enq && p->b_coll - p->c_enq < 2) { what = B_COLL; return true; }
            return false;
        });
        if (p->stop || p->err) return;
        int st = SCAL_OK;
        if (what == B_ENQ) {
            const long long k = p->b_enq;
            const int slot = static_cast<int>(k % p->ring);
            lk.unlock();
            st = p->tm_odom.run(0, "B: odom_enqueue_features", [&] { return p->for_all([&](int q) { return scal_odom_enqueue_features(p->od[q], p->regs[q][slot]); }); });
            lk.lock();
            if (st == SCAL_OK) p->b_enq = k + 1;
        } else {
            const long long k = p->b_coll;
            scal_pipeline_result* R = p->r(k).res;
            lk.unlock();
            st = p->tm_odom.run(1, "B: odom_collect (wait)", [&] {
                return p->for_each([&](int q) {
                    double qlc[4], tlc[3];
                    return scal_odom_collect(p->od[q], qlc, tlc, R[q].q_odom, R[q].t_odom, &R[q].odom);
                });
            });
            lk.lock();
            if (st == SCAL_OK) p->b_coll = k + 1;
        }
        if (st != SCAL_OK) {
            p->fail(st);
            return;
        }
        p->cv.notify_all();
    }
}

// Stage C's thread: a collected stage-B pose goes straight into a stage-C step (at most depth + 1 of them in flight, scal_map allows 4),
// finished steps are collected.
void pose_thread(scal_pipeline* p) {
    std::unique_lock<std::mutex> lk(p->mu);
    for (;;) {
        enum { NONE, C_ENQ, C_COLL, FINISH } what = NONE;
        p->cv.wait(lk, [&] {
            if (p->stop || p->err) return true;
            const long long inflight = p->c_enq - p->c_coll;
            if (p->c_enq < p->b_coll && p->c_enq < p->pf_done && inflight <= p->depth) { what = C_ENQ; return true; }
            // A pose is collected when more than `depth` steps are queued on the device - or at once when the caller waits for it and
            // nothing more can be queued meanwhile (every pushed scan has its step queued): never at the price of an empty queue.
            if (inflight > p->depth || (inflight > 0 && (p->pop_waiting || p->drain_req) && p->c_enq == p->pushed)) { what = C_COLL; return true; }
            if (p->drain_req && p->c_coll == p->pushed) { what = FINISH; return true; }
            return false;
        });
        if (p->stop || p->err) return;
        int st = SCAL_OK;
        if (what == C_ENQ) {
            const long long k = p->c_enq;
            const int slot = static_cast<int>(k % p->ring);
            scal_pipeline_result* R = p->r(k).res;
            lk.unlock();
            st = p->tm_pose.run(2, "C: map_enqueue_features", [&] {
                return p->for_all([&](int q) { return scal_map_enqueue_features(p->mp[q], p->regs[q][slot], R[q].q_odom, R[q].t_odom); });
            });
            lk.lock();
            if (st == SCAL_OK) p->c_enq = k + 1;
        } else if (what == C_COLL) {
            const long long k = p->c_coll;
            scal_pipeline_result* R = p->r(k).res;
            lk.unlock();
            st = p->tm_pose.run(3, "C: map_collect (wait)", [&] {
                return p->for_each([&](int q) { return scal_map_collect(p->mp[q], R[q].q_w_curr, R[q].t_w_curr, &R[q].map); });
            });
            lk.lock();
            if (st == SCAL_OK) p->c_coll = k + 1;
        } else if (what == FINISH) {
            lk.unlock();
            st = p->for_each([&](int q) { return scal_map_finish(p->mp[q]); });  // the last scan's map insertion (:738-802) belongs to the work
            lk.lock();
            if (st == SCAL_OK) p->drain_req = false, p->drained = true;
        }
        if (st != SCAL_OK) {
            p->fail(st);
            return;
        }
        p->cv.notify_all();
    }
}

void loop_thread(scal_pipeline* p) {  // ScanContext's answers (the searches themselves are queued by the front thread)
    std::unique_lock<std::mutex> lk(p->mu);
    const bool search = p->cfg.sc_mode == SCAL_PIPE_SC_EVERY_SCAN;
    for (;;) {
        p->cv.wait(lk, [&] {
            if (p->stop || p->err) return true;
            // an answer is collected one scan behind its search - a whole period to finish - or as soon as somebody waits for it
            const long long infl = p->d_ins - p->d_coll;
            return infl >= 2 || (infl > 0 && (p->pop_waiting || p->drain_req));
        });
        if (p->stop || p->err) return;
        const long long k = p->d_coll;
        scal_pipeline_result* R = p->r(k).res;
        lk.unlock();
        int st;
        if (search) {
            st = p->tm_loop.run(2, "D: sc_detect_collect (wait)", [&] {
                return p->for_each([&](int q) {
                    const int rc = scal_sc_detect_collect(p->sc[q], &R[q].loop);
                    R[q].have_loop = rc == SCAL_OK;
                    return rc;
                });
            });
        } else {
            st = scal_sc_wait_descriptor(p->sc[0]);
        }
        lk.lock();
        if (st != SCAL_OK) {
            p->fail(st);
            return;
        }
        p->d_coll = k + 1;
        p->cv.notify_all();
    }
}

int report(scal_pipeline* p) {  // mu held
    set_error("%s", p->errmsg.c_str());
    return p->err;
}

int push(scal_pipeline* p, const float* const* d_xyz, const void* const* h_xyz, const int* n, int stride, bool host) {
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->err) return report(p);
    if (p->pushed - p->popped >= MAX_UNPOPPED) {
        set_error("scal_pipeline_push: %d results wait to be popped", MAX_UNPOPPED);
        return SCAL_E_STATE;
    }
    const long long k = p->pushed;
    Rec& rc = p->r(k);
    rc = Rec();
    for (int q = 0; q < p->S; ++q) {
        rc.d_xyz[q] = d_xyz ? d_xyz[q] : nullptr, rc.h_xyz[q] = h_xyz ? h_xyz[q] : nullptr, rc.n[q] = n[q];
        rc.res[q].seq = k;
    }
    rc.stride = stride, rc.host = host;
    p->pushed = k + 1;
    p->drained = false;
    p->cv.notify_all();
    // a host buffer belongs to the caller again when this returns: wait until stage A has staged it.  A device buffer is only
    // registered; the call blocks while `ring` scans are in flight (back-pressure instead of an unbounded queue).
    p->cv.wait(lk, [&] { return p->err || (host ? p->a_done > k : p->a_done + p->ring > k); });
    if (p->err) return report(p);
    return SCAL_OK;
}

int pop(scal_pipeline* p, scal_pipeline_result* out) {
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->popped >= p->pushed && !p->err) {
        set_error("scal_pipeline_pop: nothing pushed");
        return SCAL_E_STATE;
    }
    const long long k = p->popped;
    p->pop_waiting++;
    p->cv.notify_all();
    p->cv.wait(lk, [&] { return p->err || (p->c_coll > k && (!p->sc_on() || p->d_coll > k)); });
    p->pop_waiting--;
    if (p->err) return report(p);
    for (int q = 0; q < p->S; ++q) out[q] = p->r(k).res[q];
    p->popped = k + 1;
    p->cv.notify_all();
    return SCAL_OK;
}

int create(const scal_pipeline_config* cfg, int n_seqs, scal_pipeline_t** out) {
    if (!cfg || !out) {
        set_error("scal_pipeline_create: null argument");
        return SCAL_E_ARG;
    }
    *out = nullptr;
    const int ring = cfg->ring == 0 ? 6 : cfg->ring, depth = cfg->depth == 0 ? 2 : cfg->depth;
    if (ring < 4 || ring > 16 || depth < 1 || depth > 3 || cfg->sc_mode < 0 || cfg->sc_mode > 2) {
        set_error("scal_pipeline_create: ring must be 4..16, depth 1..3, sc_mode 0..2");
        return SCAL_E_ARG;
    }
    if (n_seqs < 1 || n_seqs > SMAX || (n_seqs > 1 && cfg->sc_mode == SCAL_PIPE_SC_DESCRIPTOR)) {
        set_error("scal_pipeline_create_multi: 1..%d sequences (sc_mode 2 only with one)", SMAX);
        return SCAL_E_ARG;
    }
    SCAL_TRY(select_device(cfg->device));
    const int mode_before = stream_mode();
    SCAL_TRY(scal_set_stream_mode(1));  // one stream per stage; the contexts created below pick their lanes from it
    auto* p = new scal_pipeline();
    p->cfg = *cfg, p->ring = ring, p->depth = depth, p->S = n_seqs;
    int rc = SCAL_OK;
    scal_features_config fc{};
    fc.lidar_type = cfg->lidar_type, fc.n_scans = cfg->n_scans, fc.minimum_range = cfg->minimum_range, fc.max_points = cfg->max_points;
    fc.float_math = cfg->float_math, fc.check_finite = cfg->check_finite, fc.device = cfg->device;
    for (int q = 0; q < n_seqs && rc == SCAL_OK; ++q) {
        for (int i = 0; i < ring && rc == SCAL_OK; ++i) {
            scal_features_t* f = nullptr;
            rc = scal_features_create(&fc, &f);
            if (rc == SCAL_OK) p->regs[q].push_back(f);
        }
        if (rc == SCAL_OK) {
            scal_odom_config oc{};
            oc.max_points = cfg->max_points, oc.device = cfg->device;
            rc = scal_odom_create(&oc, &p->od[q]);
        }
        if (rc == SCAL_OK) {
            scal_map_config mc{};
            mc.line_res = cfg->line_res, mc.plane_res = cfg->plane_res, mc.max_scan_points = cfg->max_points, mc.max_map_points = cfg->max_map_points;
            mc.device = cfg->device;
            rc = scal_map_create(&mc, &p->mp[q]);
        }
        if (rc == SCAL_OK && p->sc_on()) {
            scal_sc_config sc{};
            sc.max_radius = cfg->sc_max_radius, sc.dist_thres = cfg->sc_dist_thres, sc.max_keyframes = cfg->sc_max_keyframes;
            sc.float_math = cfg->float_math, sc.device = cfg->device, sc.n_shards = 1, sc.shard = 0, sc.side_stream = env_int("SCALOAM_PIPE_SC_LANE", 1, 0, 5);
            // ScanContext entirely on the side stream it shares with stage C's surf filter (keyframe filter, descriptor, search: one in-order
            // chain, no cross-stream wait).  With the descriptor + search behind stage B instead (scal_set_stream_mode(1)'s own split, lane 0
            // here) stage B's chain queues behind a search that waits for the keyframe filter on the other stream: 3250-3360 scans/s against
            // 3660 on one box (tools/gpu_pipe_knobs.sh).
            rc = scal_sc_create(&sc, &p->sc[q]);
            if (rc == SCAL_OK && cfg->sc_mode == SCAL_PIPE_SC_DESCRIPTOR) {
                if (cfg->d_desc_ring) {
                    p->d_desc = cfg->d_desc_ring;
                } else if (hipMalloc(reinterpret_cast<void**>(&p->d_desc), sizeof(double) * 1200 * ring) != hipSuccess) {
                    set_error("scal_pipeline_create: hipMalloc of the descriptor ring failed");
                    rc = SCAL_E_HIP;
                } else {
                    p->own_desc = true;
                }
            }
        }
    }
    (void)scal_set_stream_mode(mode_before);  // the mode only matters while contexts are created
    if (rc != SCAL_OK) {
        const std::string keep = scal_last_error();
        scal_pipeline_destroy(p);
        set_error("%s", keep.c_str());
        return rc;
    }
    p->t_front = std::thread(front_thread, p);
    p->t_side = std::thread(side_thread, p);
    p->t_odom = std::thread(odom_thread, p);
    p->t_pose = std::thread(pose_thread, p);
    if (p->sc_on()) p->t_loop = std::thread(loop_thread, p);
    *out = p;
    return SCAL_OK;
}

}  // namespace

extern "C" int scal_pipeline_create(const scal_pipeline_config* cfg, scal_pipeline_t** out) { return create(cfg, 1, out); }
extern "C" int scal_pipeline_create_multi(const scal_pipeline_config* cfg, int n_seqs, scal_pipeline_t** out) { return create(cfg, n_seqs, out); }
extern "C" int scal_pipeline_seqs(scal_pipeline_t* p) { return p ? p->S : 0; }

extern "C" void scal_pipeline_destroy(scal_pipeline_t* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
        p->cv.notify_all();
    }
    if (p->t_front.joinable()) p->t_front.join();
    if (p->t_side.joinable()) p->t_side.join();
    if (p->t_odom.joinable()) p->t_odom.join();
    if (p->t_pose.joinable()) p->t_pose.join();
    if (p->t_loop.joinable()) p->t_loop.join();
    if (TIMING) {
        p->tm_front.print("front"), p->tm_side.print("side"), p->tm_odom.print("odom"), p->tm_pose.print("pose"), p->tm_loop.print("loop");
        std::fprintf(stderr, "[scal_pipeline] sequences %d, recorded lists flushed in the middle of an entry point: %ld\n", p->S, g_forced_flushes);
        print_forced_flushes();
    }
    (void)hipSetDevice(p->cfg.device);
    // consumers first: their destructors wait for the streams that still read the features contexts
    for (int q = 0; q < SMAX; ++q) {
        if (p->sc[q]) scal_sc_destroy(p->sc[q]);
        if (p->mp[q]) scal_map_destroy(p->mp[q]);
        if (p->od[q]) scal_odom_destroy(p->od[q]);
    }
    for (int q = 0; q < SMAX; ++q)
        for (auto* f : p->regs[q]) scal_features_destroy(f);
    if (p->d_desc && p->own_desc) (void)hipFree(p->d_desc);
    delete p;
}

static int check_scan(scal_pipeline_t* p, const void* ptr, int n) {
    if ((!ptr && n > 0) || n < 0) {
        set_error("scal_pipeline_push: bad argument");
        return SCAL_E_ARG;
    }
    if (n > p->cfg.max_points) {
        set_error("scan has %d points, capacity is %d", n, p->cfg.max_points);
        return SCAL_E_TOO_MANY;
    }
    return SCAL_OK;
}

extern "C" int scal_pipeline_push_device(scal_pipeline_t* p, const float* d_xyz, int n, int stride_floats) {
    if (!p || stride_floats < 3 || p->S != 1) {
        set_error("scal_pipeline_push_device: bad argument (a pipeline of several sequences takes scal_pipeline_push_device_multi)");
        return SCAL_E_ARG;
    }
    SCAL_TRY(check_scan(p, d_xyz, n));
    return push(p, &d_xyz, nullptr, &n, stride_floats, false);
}

extern "C" int scal_pipeline_push_device_multi(scal_pipeline_t* p, const float* const* d_xyz, const int* n, int stride_floats) {
    if (!p || !d_xyz || !n || stride_floats < 3) {
        set_error("scal_pipeline_push_device_multi: bad argument");
        return SCAL_E_ARG;
    }
    for (int q = 0; q < p->S; ++q) SCAL_TRY(check_scan(p, d_xyz[q], n[q]));
    return push(p, d_xyz, nullptr, n, stride_floats, false);
}

extern "C" int scal_pipeline_push_host(scal_pipeline_t* p, const void* xyz, int n, int stride_bytes) {
    if (!p || stride_bytes < 12 || (stride_bytes % 4) != 0 || stride_bytes > 32 || p->S != 1) {
        set_error("scal_pipeline_push_host: bad argument (stride_bytes must be a multiple of 4 in [12, 32]; one sequence)");
        return SCAL_E_ARG;
    }
    SCAL_TRY(check_scan(p, xyz, n));
    return push(p, nullptr, &xyz, &n, stride_bytes, true);
}

extern "C" int scal_pipeline_pop(scal_pipeline_t* p, scal_pipeline_result* out) {
    if (!p || !out || p->S != 1) {
        set_error("scal_pipeline_pop: bad argument (a pipeline of several sequences takes scal_pipeline_pop_multi)");
        return SCAL_E_ARG;
    }
    return pop(p, out);
}

extern "C" int scal_pipeline_pop_multi(scal_pipeline_t* p, scal_pipeline_result* out) {
    if (!p || !out) {
        set_error("scal_pipeline_pop_multi: null argument");
        return SCAL_E_ARG;
    }
    return pop(p, out);
}

extern "C" int scal_pipeline_drain(scal_pipeline_t* p) {
    if (!p) return SCAL_E_ARG;
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->err) return report(p);
    if (p->drained) return SCAL_OK;
    p->drain_req = true;
    p->pop_waiting++;  // the loop thread collects its last answer, the pose thread every queued pose
    p->cv.notify_all();
    p->cv.wait(lk, [&] { return p->err || (p->drained && (!p->sc_on() || p->d_coll == p->pushed)); });
    p->pop_waiting--;
    if (p->err) return report(p);
    return SCAL_OK;
}

extern "C" int scal_pipeline_in_flight(scal_pipeline_t* p) {
    if (!p) return 0;
    std::lock_guard<std::mutex> lk(p->mu);
    return static_cast<int>(p->pushed - p->popped);
}

static bool seq_ok(scal_pipeline_t* p, int q) { return p && q >= 0 && q < p->S; }
extern "C" scal_sc_t* scal_pipeline_sc(scal_pipeline_t* p) { return p ? p->sc[0] : nullptr; }
extern "C" scal_map_t* scal_pipeline_map(scal_pipeline_t* p) { return p ? p->mp[0] : nullptr; }
extern "C" scal_odom_t* scal_pipeline_odom(scal_pipeline_t* p) { return p ? p->od[0] : nullptr; }
extern "C" scal_features_t* scal_pipeline_features(scal_pipeline_t* p, int i) {
    return (p && i >= 0 && i < static_cast<int>(p->regs[0].size())) ? p->regs[0][i] : nullptr;
}
extern "C" scal_sc_t* scal_pipeline_sc_of(scal_pipeline_t* p, int seq) { return seq_ok(p, seq) ? p->sc[seq] : nullptr; }
extern "C" scal_map_t* scal_pipeline_map_of(scal_pipeline_t* p, int seq) { return seq_ok(p, seq) ? p->mp[seq] : nullptr; }
