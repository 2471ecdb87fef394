// Error reporting and device selection for libscaloam_hip.so.
#include "common.hpp"
#include <chrono>
#include <cstdlib>
#include <memory>

namespace scal {

// ---------------------------------------------------------------------------------------------- in-launch batching (batch.hpp)
thread_local Recorder* g_recorder = nullptr;
long g_forced_flushes = 0;
static std::mutex g_flush_mu;
static std::vector<std::pair<std::string, long>> g_flush_sites;
void note_forced_flush(const char* file, int line) {
    std::lock_guard<std::mutex> lk(g_flush_mu);
    ++g_forced_flushes;
    const std::string key = std::string(file) + ":" + std::to_string(line);
    for (auto& s : g_flush_sites)
        if (s.first == key) {
            ++s.second;
            return;
        }
    g_flush_sites.push_back({key, 1});
}
void print_forced_flushes() {
    std::lock_guard<std::mutex> lk(g_flush_mu);
    for (auto& s : g_flush_sites) std::fprintf(stderr, "[scal_pipeline] forced flush at %s x %ld\n", s.first.c_str(), s.second);
}

static hipError_t run_op(const RecOp& o) {
    if (o.kind != RecOp::LAUNCH) return o.replay();
    const void* one = o.pack;
    return o.fn(o.name, o.grid, o.block, o.lds, o.stream, 1, &one);
}

hipError_t Recorder::flush() {
    Recorder* keep = g_recorder;
    g_recorder = nullptr;  // the replayed operations go to the device
    hipError_t first = hipSuccess;
    for (const RecOp& o : ops) {
        const hipError_t e = run_op(o);
        if (first == hipSuccess) first = e;
    }
    ops.clear();
    g_recorder = keep;
    return first;
}

namespace {
// name of a launch that carries work of differently tagged callers ("k_rs_scatter.C" + "k_rs_scatter.D" -> "k_rs_scatter.C+D"):
// interned, so that the const char* outlives the call (scal_prof_* keeps names)
const char* merged_name(const char* a, const char* b) {
    static std::mutex mu;
    static std::vector<std::unique_ptr<std::string>> pool;
    const char* dot = std::strrchr(b, '.');
    const std::string key = std::string(a) + "+" + (dot ? dot + 1 : b);
    std::lock_guard<std::mutex> lk(mu);
    for (auto& s : pool)
        if (*s == key) return s->c_str();
    pool.push_back(std::make_unique<std::string>(key));
    return pool.back()->c_str();
}
struct Span {
    Recorder* r;
    size_t b, e;
};
bool same_op(const RecOp& a, const RecOp& b) {
    if (a.kind != b.kind) return false;
    if (a.kind != RecOp::LAUNCH) return true;
    return a.fn == b.fn && a.grid.x == b.grid.x && a.grid.y == b.grid.y && a.block.x == b.block.x && a.lds == b.lds && a.stream == b.stream &&
           a.pack_size == b.pack_size;
}
// Spans with the same structure (same kernels and shapes at the same positions) are merged; a span that differs - a sequence that
// took another path through its entry point, or had to flush on the way - is replayed on its own.
hipError_t zip_spans(const std::vector<Span>& spans) {
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t e) { if (first == hipSuccess) first = e; };
    const int n = static_cast<int>(spans.size());
    std::vector<bool> done(n, false);
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        const Span& lead = spans[i];
        const size_t len = lead.e - lead.b;
        std::vector<const Span*> group{&lead};
        done[i] = true;
        if (!lead.r->broken)
            for (int j = i + 1; j < n && static_cast<int>(group.size()) < BATCH_WIDE; ++j) {
                const Span& c = spans[j];
                if (done[j] || c.r->broken || c.e - c.b != len) continue;
                bool ok = true;
                for (size_t k = 0; k < len && ok; ++k) ok = same_op(lead.r->ops[lead.b + k], c.r->ops[c.b + k]);
                if (ok) group.push_back(&c), done[j] = true;
            }
        for (size_t k = 0; k < len; ++k) {
            const RecOp& o = lead.r->ops[lead.b + k];
            if (o.kind == RecOp::LAUNCH) {
                // a kernel takes o.cap argument sets per launch (eight when they are small, else four): larger groups take several
                for (size_t g0 = 0; g0 < group.size(); g0 += static_cast<size_t>(o.cap)) {
                    const size_t g1 = std::min(group.size(), g0 + static_cast<size_t>(o.cap));
                    const void* packs[BATCH_WIDE];
                    const char* name = o.name;
                    for (size_t g = g0; g < g1; ++g) {
                        const RecOp& og = group[g]->r->ops[group[g]->b + k];
                        packs[g - g0] = og.pack;
                        if (g > g0 && std::strcmp(og.name, o.name) != 0 && !std::strstr(name, "+")) name = merged_name(o.name, og.name);
                    }
                    note(o.fn(name, o.grid, o.block, o.lds, o.stream, static_cast<int>(g1 - g0), packs));
                }
            } else {
                for (const Span* g : group) note(g->r->ops[g->b + k].replay());
            }
        }
    }
    return first;
}
}  // namespace

hipError_t zip_and_launch(Recorder* const* recs, int n) {
    Recorder* keep = g_recorder;
    g_recorder = nullptr;
    std::vector<Span> spans;
    for (int i = 0; i < n; ++i) spans.push_back({recs[i], 0, recs[i]->ops.size()});
    const hipError_t e = zip_spans(spans);
    for (int i = 0; i < n; ++i) recs[i]->ops.clear(), recs[i]->broken = false;
    g_recorder = keep;
    return e;
}

hipError_t zip_marked_and_launch(Recorder* const* recs, int n) {
    Recorder* keep = g_recorder;
    g_recorder = nullptr;
    std::vector<Span> before, seg, after;
    for (int i = 0; i < n; ++i) {
        Recorder* r = recs[i];
        size_t b = r->ops.size(), e = r->ops.size();
        for (size_t k = 0; k < r->ops.size(); ++k)
            if (r->ops[k].kind == RecOp::MARK_BEGIN) {
                b = k;
                break;
            }
        for (size_t k = b; k < r->ops.size(); ++k)
            if (r->ops[k].kind == RecOp::MARK_END) {
                e = k + 1;
                break;
            }
        if (r->broken || b == r->ops.size()) {  // no segment (or an unbatchable list): everything counts as "before"
            before.push_back({r, 0, r->ops.size()});
            continue;
        }
        before.push_back({r, 0, b});
        seg.push_back({r, b, e});
        after.push_back({r, e, r->ops.size()});
    }
    hipError_t first = zip_spans(before);
    hipError_t e2 = zip_spans(seg);
    hipError_t e3 = zip_spans(after);
    if (first == hipSuccess) first = e2 != hipSuccess ? e2 : e3;
    for (int i = 0; i < n; ++i) recs[i]->ops.clear(), recs[i]->broken = false;
    g_recorder = keep;
    return first;
}

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s): libscaloam_hip has no CPU fallback", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        return SCAL_E_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device ordinal %d out of range (%d devices)", device, n);
        return SCAL_E_ARG;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return SCAL_E_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
        return SCAL_E_NO_DEVICE;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        set_error("hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
        return SCAL_E_HIP;
    }
    return SCAL_OK;
}

namespace {
std::mutex g_stream_mu;
struct DevStream {
    hipStream_t s = nullptr;
    int refs = 0;
};
DevStream g_streams[64][6];  // [device][lane], see common.hpp
int g_stream_mode = 0;
}  // namespace

int acquire_stream(int device, hipStream_t* out, int lane) {
    if (device < 0 || device >= 64 || lane < 0 || lane > 5) {
        set_error("device ordinal %d out of range", device);
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(g_stream_mu);
    DevStream& d = g_streams[device][lane];
    if (!d.s) {
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.s, hipStreamNonBlocking);
        if (e != hipSuccess) {
            set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            d.s = nullptr;
            return SCAL_E_HIP;
        }
    }
    d.refs++;
    *out = d.s;
    return SCAL_OK;
}

int stream_mode() {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    return g_stream_mode;
}

int stage_lane(int stage) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    if (g_stream_mode == 0) return stage == STAGE_MAP_PREFETCH ? 2 : 0;
    // Four streams, not five: an MI355X advances four dependent kernel chains at full rate, and with more streams than that they
    // all slow down together (tools/stream_probe.hip: 0.62 M kernels/s with 4 streams, 0.25-0.35 M with 6-8).  The side jobs that
    // only depend on stage A are spread so that the four chains come out about equally long (tools/gpu_chains.sh): stage C's surf
    // filter and ScanContext's keyframe filter share the side stream, its corner filter rides behind stage A, and ScanContext's
    // descriptor + search (short) ride behind stage B.
    return (stage == STAGE_ODOM || stage == STAGE_SC) ? 3 : stage == STAGE_MAP ? 4 : (stage == STAGE_SC_FILTER || stage == STAGE_MAP_PREFETCH) ? 1 : 0;
}

void release_stream(int device, int lane) {
    if (device < 0 || device >= 64 || lane < 0 || lane > 5) return;
    std::lock_guard<std::mutex> lk(g_stream_mu);
    DevStream& d = g_streams[device][lane];
    if (d.refs > 0 && --d.refs == 0 && d.s) {
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(d.s);
        (void)hipStreamDestroy(d.s);
        d.s = nullptr;
    }
}

// ---------------------------------------------------------------------------------------------- event profiling
namespace {
struct ProfEntry {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<const void*> pending_tl;  // keep this dispatch for the timeline: its stream (null = do not keep)
    double total_ms = 0;
    long count = 0;
};
std::mutex g_prof_mu;
std::vector<ProfEntry> g_prof;
std::vector<hipEvent_t> g_event_pool;
bool g_prof_on = false;
std::string g_prof_filter;  // empty = every instrumented kernel
// optional raw timeline: (kernel, start, stop) in ms since a base event, for the dispatches timed while it is on
struct TimelineRow {
    std::string name;
    float t0, t1;
    const void* stream;
};
bool g_timeline_on = false;
hipEvent_t g_timeline_base = nullptr;
std::vector<TimelineRow> g_timeline;

hipEvent_t take_event() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void drain_entry(ProfEntry& p);
void drain_locked() {
    for (auto& p : g_prof) drain_entry(p);
}
void drain_entry(ProfEntry& p) {
    {
        for (size_t q = 0; q < p.pending.size(); ++q) {
            auto& ev = p.pending[q];
            float ms = 0.f;
            if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                p.total_ms += ms;
                p.count++;
                float a = 0.f;
                if (p.pending_tl[q] && g_timeline_base && hipEventElapsedTime(&a, g_timeline_base, ev.first) == hipSuccess)
                    g_timeline.push_back({p.name, a, a + ms, p.pending_tl[q]});
            }
            g_event_pool.push_back(ev.first);
            g_event_pool.push_back(ev.second);
        }
        p.pending.clear();
        p.pending_tl.clear();
    }
}
}  // namespace

bool prof_begin(const char* name, hipStream_t stream, hipEvent_t* start, hipEvent_t* stop) {
    // development aid: SCALOAM_HOST_DELAY_US=n spins n microseconds in front of every launch - a slower host, to see when the
    // host threads rather than the device chains bound the stage pipeline (tools/gpu_threads_ab.sh)
    static const int delay_us = [] {
        const char* e = std::getenv("SCALOAM_HOST_DELAY_US");
        return e ? std::atoi(e) : 0;
    }();
    if (delay_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < delay_us) {
        }
    }
    if (!g_prof_on) return false;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_filter.empty()) {  // one name, or several separated by commas
        const std::string f = "," + g_prof_filter + ",";
        if (f.find("," + std::string(name) + ",") == std::string::npos) return false;
    }
    int slot = -1;
    for (size_t i = 0; i < g_prof.size(); ++i)
        if (g_prof[i].name == name) slot = static_cast<int>(i);
    if (slot < 0) {
        g_prof.push_back(ProfEntry());
        g_prof.back().name = name;
        slot = static_cast<int>(g_prof.size()) - 1;
    }
    if (g_prof[slot].pending.size() >= 64) drain_entry(g_prof[slot]);  // bounded number of outstanding events
    hipEvent_t a = take_event(), b = take_event();
    if (!a || !b) return false;
    g_prof[slot].pending.push_back({a, b});
    g_prof[slot].pending_tl.push_back(g_timeline_on ? (stream ? static_cast<const void*>(stream) : static_cast<const void*>(&g_timeline_on)) : nullptr);
    *start = a, *stop = b;
    return true;
}

}  // namespace scal

extern "C" int scal_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(scal::g_prof_mu);
    scal::g_prof_on = on != 0;
    return SCAL_OK;
}
extern "C" int scal_prof_timeline(int on) {
    std::lock_guard<std::mutex> lk(scal::g_prof_mu);
    if (on && !scal::g_timeline_base) {
        if (hipEventCreate(&scal::g_timeline_base) != hipSuccess) return SCAL_E_HIP;
        if (hipEventRecord(scal::g_timeline_base, nullptr) != hipSuccess) return SCAL_E_HIP;
        if (hipEventSynchronize(scal::g_timeline_base) != hipSuccess) return SCAL_E_HIP;
    }
    scal::g_timeline_on = on != 0;
    return SCAL_OK;
}
extern "C" int scal_prof_timeline_dump(const char* path) {
    std::lock_guard<std::mutex> lk(scal::g_prof_mu);
    scal::drain_locked();
    FILE* f = std::fopen(path, "w");
    if (!f) return SCAL_E_ARG;
    for (auto& r : scal::g_timeline) std::fprintf(f, "%s,%.6f,%.6f,%p\n", r.name.c_str(), r.t0, r.t1, r.stream);
    std::fclose(f);
    scal::g_timeline.clear();
    return SCAL_OK;
}
extern "C" int scal_prof_filter(const char* kernel_name) {
    std::lock_guard<std::mutex> lk(scal::g_prof_mu);
    scal::g_prof_filter = kernel_name ? kernel_name : "";
    return SCAL_OK;
}
extern "C" int scal_prof_reset(void) {
    std::lock_guard<std::mutex> lk(scal::g_prof_mu);
    scal::drain_locked();
    for (auto& p : scal::g_prof) p.total_ms = 0, p.count = 0;
    return SCAL_OK;
}
extern "C" int scal_prof_read(const char* name, double* total_ms, long* count) {
    std::lock_guard<std::mutex> lk(scal::g_prof_mu);
    scal::drain_locked();
    for (auto& p : scal::g_prof)
        if (p.name == name) {
            if (total_ms) *total_ms = p.total_ms;
            if (count) *count = p.count;
            return SCAL_OK;
        }
    if (total_ms) *total_ms = 0;
    if (count) *count = 0;
    return SCAL_OK;
}
extern "C" int scal_prof_names(char* buf, int cap) {
    std::lock_guard<std::mutex> lk(scal::g_prof_mu);
    std::string all;
    for (auto& p : scal::g_prof) all += p.name + ";";
    if (buf && cap > 0) {
        std::strncpy(buf, all.c_str(), cap - 1);
        buf[cap - 1] = 0;
    }
    return static_cast<int>(all.size());
}

extern "C" int scal_set_stream_mode(int mode) {
    if (mode != 0 && mode != 1) {
        scal::set_error("scal_set_stream_mode: mode must be 0 or 1");
        return SCAL_E_ARG;
    }
    std::lock_guard<std::mutex> lk(scal::g_stream_mu);
    scal::g_stream_mode = mode;
    return SCAL_OK;
}

extern "C" const char* scal_last_error(void) { return scal::g_err; }
extern "C" int scal_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}
extern "C" const char* scal_version(void) { return "scaloam_hip 0.1 (gfx950)"; }
