// Host-side plumbing shared by every stage of libscaloam_hip.so: error reporting, RAII device/pinned buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include "../../include/scaloam_hip.h"
#include "batch.hpp"

namespace scal {
// Flags of events that only order streams of the same device against each other (nobody on the host reads memory behind them): no
// system-scope fence when the event completes.  Recording a default event behind a kernel writes the caches back for the host's
// sake; on the pipeline's chains that was several microseconds per record, several records per chain and scan.
constexpr unsigned EV_DEVICE_ONLY = hipEventDisableTiming | hipEventDisableSystemFence;


void set_error(const char* fmt, ...);

#define SCAL_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            ::scal::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SCAL_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define SCAL_TRY(expr)           \
    do {                         \
        int _rc = (expr);        \
        if (_rc != SCAL_OK) return _rc; \
    } while (0)

// selects the device and verifies it is a gfx950 part: the product path must fail loudly otherwise
int select_device(int device);
// Streams are shared per device and reference counted.  Lane 0 is the pipeline's in-order stream: by default stages A -> B -> C
// of a scan all run there, strictly dependent, no cross-stream events.  Lanes 1 and 2 carry work that only depends on stage A
// (1: ScanContext with scal_sc_config::side_stream, 2: scal_map_prefetch_features).  With scal_set_stream_mode(1) (set before
// the contexts are created) every stage gets its own stream - A: 0, D: 1, C prefetch: 2, B: 3, C: 4 - so that consecutive
// scans overlap the way the reference's four ROS nodes do; every hand-over between contexts is ordered by events in both
// directions (features_wait_done / features_note_reader).  Lane 5 is free for a context that must not queue behind another one
// of its kind (scal_sc_config::side_stream = 5: the sharded database next to the descriptor builder, bench.py --gpus N).
enum { STAGE_FEATURES = 0, STAGE_SC = 1, STAGE_ODOM = 3, STAGE_MAP = 4, STAGE_MAP_PREFETCH = 2, STAGE_SC_FILTER = 5 };
int stage_lane(int stage);
int stream_mode();  // the value scal_set_stream_mode last set
int acquire_stream(int device, hipStream_t* out, int lane = 0);
void release_stream(int device, int lane = 0);

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            p = nullptr;
            return SCAL_E_HIP;
        }
        n = count;
        return SCAL_OK;
    }
    int zero(hipStream_t s) {
        hipError_t e = op_memset_async(p, 0, n * sizeof(T), s);
        if (e != hipSuccess) {
            set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
            return SCAL_E_HIP;
        }
        return SCAL_OK;
    }
};

template <class T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf() {
        if (p) (void)hipHostFree(p);
    }
    int alloc(size_t count) {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        if (count == 0) count = 1;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), count * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) {
            set_error("hipHostMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            p = nullptr;
            return SCAL_E_HIP;
        }
        n = count;
        return SCAL_OK;
    }
};

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// Host arrays of the synchronous entry points are the caller's (pageable) memory: a copy straight to or from them is staged by the
// runtime in small synchronous pieces.  This keeps one pinned area per context instead: downloads land there with ONE asynchronous
// copy each and are handed to the caller's arrays after the stream has been synchronised; uploads are copied in first.
struct HostStage {
    PinBuf<unsigned char> pin;
    size_t used = 0;
    struct Item {
        void* dst;
        size_t off, bytes;
    };
    std::vector<Item> items;
    // the staging area holds `total` bytes from now on (only grows while nothing is pending); false: no memory, callers copy directly
    bool reserve(size_t total) {
        if (pin.p && pin.n >= total) return true;
        if (used) return false;
        if (pin.alloc(total) != SCAL_OK) {
            (void)hipGetLastError();
            return false;
        }
        return true;
    }
    hipError_t d2h(void* user, const void* dev, size_t bytes, hipStream_t s) {
        if (!bytes) return hipSuccess;
        const size_t need = (bytes + 255) & ~static_cast<size_t>(255);
        if (!pin.p || used + need > pin.n) return hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, s);
        const hipError_t e = hipMemcpyAsync(pin.p + used, dev, bytes, hipMemcpyDeviceToHost, s);
        items.push_back({user, used, bytes});
        used += need;
        return e;
    }
    hipError_t h2d(void* dev, const void* user, size_t bytes, hipStream_t s) {
        if (!bytes) return hipSuccess;
        const size_t need = (bytes + 255) & ~static_cast<size_t>(255);
        if (!pin.p || used + need > pin.n) return hipMemcpyAsync(dev, user, bytes, hipMemcpyHostToDevice, s);
        std::memcpy(pin.p + used, user, bytes);
        const hipError_t e = hipMemcpyAsync(dev, pin.p + used, bytes, hipMemcpyHostToDevice, s);
        used += need;
        return e;
    }
    // the stream the copies were queued on has been synchronised: deliver the downloads, the area is free again
    void finish() {
        for (const Item& it : items) std::memcpy(it.dst, pin.p + it.off, it.bytes);
        items.clear();
        used = 0;
    }
};

// Optional per-kernel timing (bench.py's roofline leg).  The start/stop events are attached to the kernel's own dispatch
// (hipExtLaunchKernelGGL), so they read the dispatch's begin/end timestamps and put no extra packet on the stream: separate
// hipEventRecord calls cost ~5 us of stream time each, more than many of the kernels they would bracket.
// Disabled by default: SCAL_LAUNCH_PROF is then one branch + a plain launch.
bool prof_begin(const char* name, hipStream_t stream, hipEvent_t* start, hipEvent_t* stop);

#define SCAL_LAUNCH_PROF(name, kernel, grid, block, lds, stream, ...)                                   \
    do {                                                                                                \
        hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                                      \
        if (::scal::prof_begin(name, stream, &pe0_, &pe1_))                                                     \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, pe0_, pe1_, 0, __VA_ARGS__);        \
        else                                                                                            \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                          \
    } while (0)

}  // namespace scal
