// In-launch batching of independent sequences ("report batched/streamed numbers - several independent
// scans or sequences in flight - separately").
//
// One scan's kernels run on 1 to ~50 workgroups of a 256-CU part and are latency-bound, and the device advances no more than four
// dependent chains at full rate, so several sequences only go faster when their kernels share a LAUNCH.
// Two pieces, both in this header:
//
//  * Every hot-path kernel takes its arguments as Batch<Pack>: up to BATCH_MAX argument sets by value in the kernarg segment,
//    blockIdx.z selects the set (a scalar load with a uniform offset; the kernel body is the unchanged __device__ function
//    `<kernel>_body`).  A launch for one sequence is a batch of one.
//
//  * A thread-local Recorder.  While one is installed, SCAL_LAUNCH and the stream-operation wrappers below (event record / wait,
//    async copies) do not touch the device: they append to the recorder's list.  The per-stage entry points of the C-ABI are run
//    once per sequence under a recorder each; zip_and_launch() then walks the S lists position by position: launches of the same
//    kernel with the same shape become ONE launch with gridDim.z = S, everything else (events, copies) is replayed per sequence
//    in list order.  All sequences use the same per-stage streams, so list order is stream order.  An operation that has to
//    wait for the device while recording (a fallback path inside an entry point) first flushes its own list - that sequence then
//    simply runs unbatched for this step.  Results are bit-identical to running every sequence on its own: the kernel bodies and
//    their per-sequence arguments are the same, only the launch that carries them differs.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstring>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace scal {

constexpr int BATCH_MAX = 4;    // sequences that step together; every kernel takes at least this many argument sets
constexpr int BATCH_WIDE = 8;   // kernels with small argument sets take eight: two independent filter runs of four sequences each share launches
constexpr size_t KERNARG_BYTES_MAX = 3840;  // the kernarg segment of a dispatch holds 4 KB
template <class P>
constexpr int batch_cap() {
    return sizeof(P) * BATCH_WIDE <= KERNARG_BYTES_MAX ? BATCH_WIDE : BATCH_MAX;
}

// ---- argument sets ------------------------------------------------------------------------------------------------------------
template <size_t I, class T>
struct PackLeaf {
    T v;
};
template <class Seq, class... T>
struct PackImpl;
template <size_t... I, class... T>
struct PackImpl<std::index_sequence<I...>, T...> : PackLeaf<I, T>... {};
template <class... T>
using Pack = PackImpl<std::index_sequence_for<T...>, T...>;

template <size_t I, class T>
__host__ __device__ __forceinline__ const T& pack_get(const PackLeaf<I, T>& l) {
    return l.v;
}

template <class P>
struct Batch {
    P p[batch_cap<P>()];
};

template <class F>
struct KernelTraits;
template <class... B>
struct KernelTraits<void (*)(B...)> : KernelTraits<void(std::decay_t<B>...)> {};
template <class... A>
struct KernelTraits<void(A...)> {
    using pack = Pack<A...>;
    static constexpr size_t n = sizeof...(A);
    template <class... U>
    static pack make(U&&... u) {
        return make_impl(std::index_sequence_for<A...>(), std::forward<U>(u)...);
    }
    template <size_t... I, class... U>
    static pack make_impl(std::index_sequence<I...>, U&&... u) {
        static_assert(sizeof...(U) == sizeof...(A), "wrong number of kernel arguments");
        return pack{PackLeaf<I, A>{static_cast<A>(std::forward<U>(u))}...};
    }
};

// ---- recorder -----------------------------------------------------------------------------------------------------------------
// n <= the kernel's capacity (RecOp::cap: batch_cap of its argument set)
using BatchLaunchFn = hipError_t (*)(const char* name, dim3 grid, dim3 block, int lds, hipStream_t s, int n, const void* const* packs);

// hipStreamWaitEvent, unless the event has completed already: then the wait would be satisfied the moment the command processor reads
// it, but it would still be a barrier packet on the stream (~3 us each; the pipeline's chains wait for events that were recorded a
// scan or a whole ring of scans ago, several per chain).  A host-side query costs a fraction of a microsecond.
inline hipError_t stream_wait_unless_done(hipStream_t s, hipEvent_t ev, unsigned flags) {
    if (hipEventQuery(ev) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();  // hipErrorNotReady is not an error here
    return hipStreamWaitEvent(s, ev, flags);
}

constexpr int PACK_BYTES_MAX = 1024;  // largest argument set (k_merge_write: ~620 B)
struct RecOp {  // plain data: recording an operation allocates nothing
    enum Kind { LAUNCH, EVENT_RECORD, STREAM_WAIT, MEMCPY, MEMSET, MARK_BEGIN, MARK_END } kind = LAUNCH;
    // launch
    BatchLaunchFn fn = nullptr;
    const char* name = nullptr;  // a literal or a member string of the launching context: outlives the list
    dim3 grid, block;
    int lds = 0, pack_size = 0, cap = BATCH_MAX;
    hipStream_t stream = nullptr;
    // the others, replayed per sequence
    hipEvent_t ev = nullptr;
    unsigned flags = 0;
    void* dst = nullptr;
    const void* src = nullptr;
    size_t bytes = 0;
    int value = 0;
    hipMemcpyKind copy_kind = hipMemcpyDefault;
    alignas(16) unsigned char pack[PACK_BYTES_MAX];
    hipError_t replay() const {
        switch (kind) {
            case EVENT_RECORD: return hipEventRecord(ev, stream);
            case STREAM_WAIT: return stream_wait_unless_done(stream, ev, flags);
            case MEMCPY: return hipMemcpyAsync(dst, src, bytes, copy_kind, stream);
            case MEMSET: return hipMemsetAsync(dst, value, bytes, stream);
            default: return hipSuccess;
        }
    }
};

struct Recorder {
    std::vector<RecOp> ops;
    bool broken = false;  // this sequence had to flush in the middle of its list: it runs unbatched for the rest of the step
    hipError_t flush();   // launches / replays everything recorded so far, in order, as batches of one
    bool will_record(hipEvent_t ev) const {
        for (const RecOp& o : ops)
            if (o.kind == RecOp::EVENT_RECORD && o.ev == ev) return true;
        return false;
    }
    RecOp& add(RecOp::Kind k) {
        if (ops.capacity() == 0) ops.reserve(64);
        ops.emplace_back();
        ops.back().kind = k;
        return ops.back();
    }
};
extern long g_forced_flushes;  // flushes in the middle of a recorded list (development statistic)
void note_forced_flush(const char* file, int line);

extern thread_local Recorder* g_recorder;

// launches the lists of n recorders (all recorded on this thread for the same per-stage entry point of n sequences): same-shaped
// launches of the same kernel are merged, the rest is replayed in order.  Clears the lists.  Returns the first HIP error.
hipError_t zip_and_launch(Recorder* const* recs, int n);
// The same for lists of DIFFERENT entry points that each contain one marked segment of the same kind of work (a voxel filter run:
// stage C's surf stack filter and ScanContext's keyframe filter both sort and reduce a cloud with the same kernels).  Every list is
// cut into (before, segment, after); all the befores are launched (merged where their structure matches), then all the segments
// merged with each other, then the afters.  The lists must be independent of each other up to stream order - different contexts
// working on the same scan - and the segments must run on one stream.
hipError_t zip_marked_and_launch(Recorder* const* recs, int n);
// brackets a segment of the current thread's recorder (no effect when nothing is being recorded)
inline void rec_mark(bool begin) {
    if (Recorder* r = g_recorder) r->add(begin ? RecOp::MARK_BEGIN : RecOp::MARK_END);
}

bool prof_begin(const char* name, hipStream_t stream, hipEvent_t* start, hipEvent_t* stop);  // common.cpp

// ---- stream operations that may be recorded -------------------------------------------------------------------------------------
inline hipError_t op_event_record(hipEvent_t ev, hipStream_t s) {
    if (Recorder* r = g_recorder) {
        RecOp& o = r->add(RecOp::EVENT_RECORD);
        o.ev = ev, o.stream = s;
        return hipSuccess;
    }
    return hipEventRecord(ev, s);
}
inline hipError_t op_stream_wait_event(hipStream_t s, hipEvent_t ev, unsigned flags) {
    if (Recorder* r = g_recorder) {
        RecOp& o = r->add(RecOp::STREAM_WAIT);
        o.ev = ev, o.stream = s, o.flags = flags;
        return hipSuccess;
    }
    return stream_wait_unless_done(s, ev, flags);
}
inline hipError_t op_memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    if (Recorder* r = g_recorder) {
        RecOp& o = r->add(RecOp::MEMCPY);
        o.dst = dst, o.src = src, o.bytes = bytes, o.copy_kind = kind, o.stream = s;
        return hipSuccess;
    }
    return hipMemcpyAsync(dst, src, bytes, kind, s);
}
inline hipError_t op_memset_async(void* dst, int value, size_t bytes, hipStream_t s) {
    if (Recorder* r = g_recorder) {
        RecOp& o = r->add(RecOp::MEMSET);
        o.dst = dst, o.value = value, o.bytes = bytes, o.stream = s;
        return hipSuccess;
    }
    return hipMemsetAsync(dst, value, bytes, s);
}
// Operations that wait for (or look at) the device.  An event whose record is still sitting in this thread's list has to reach
// the device first (the list is flushed and the sequence runs unbatched for the rest of the step); an event recorded by an
// earlier, already launched step can be waited for or queried as it is - the entry points poll such events all the time.
inline hipError_t op_flush_for_wait(const char* file, int line) {
    if (Recorder* r = g_recorder) {
        r->broken = true;
        note_forced_flush(file, line);
        return r->flush();
    }
    return hipSuccess;
}
inline hipError_t op_event_synchronize(hipEvent_t ev, const char* file = __builtin_FILE(), int line = __builtin_LINE()) {
    if (g_recorder && g_recorder->will_record(ev)) {
        const hipError_t e = op_flush_for_wait(file, line);
        if (e != hipSuccess) return e;
    }
    return hipEventSynchronize(ev);
}
inline hipError_t op_stream_synchronize(hipStream_t s, const char* file = __builtin_FILE(), int line = __builtin_LINE()) {
    hipError_t e = op_flush_for_wait(file, line);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
}
inline hipError_t op_event_query(hipEvent_t ev, const char* file = __builtin_FILE(), int line = __builtin_LINE()) {
    if (g_recorder && g_recorder->will_record(ev)) {
        const hipError_t e = op_flush_for_wait(file, line);
        if (e != hipSuccess) return e;
    }
    return hipEventQuery(ev);
}

// ---- kernels ------------------------------------------------------------------------------------------------------------------
// SCAL_KERNEL(bounds, k_name) behind the definition of `__device__ void k_name_body(args...)` defines
//   __global__ k_name(Batch<pack>)          the kernel: argument set blockIdx.z, body unchanged
//   k_name_launch(...)                      BatchLaunchFn: n argument sets in one launch (gridDim.z = n), with the optional timing
//                                           events of scal_prof_* attached to the dispatch
template <class Tag, class P, size_t... I>
__device__ __forceinline__ void batch_call(const P& p, std::index_sequence<I...>) {
    Tag::call(pack_get<I>(p)...);
}

template <class P, class K>
inline hipError_t batch_launch_impl(K kernel, const char* name, dim3 grid, dim3 block, int lds, hipStream_t s, int n, const void* const* packs) {
    Batch<P> b;
    static_assert(sizeof(Batch<P>) <= 4096, "argument sets exceed the kernarg segment");
    for (int i = 0; i < n && i < batch_cap<P>(); ++i) std::memcpy(static_cast<void*>(&b.p[i]), packs[i], sizeof(P));
    grid.z = static_cast<unsigned>(n);
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (prof_begin(name, s, &pe0, &pe1))
        hipExtLaunchKernelGGL(kernel, grid, block, lds, s, pe0, pe1, 0, b);
    else
        hipLaunchKernelGGL(kernel, grid, block, lds, s, b);
    return hipSuccess;
}

#define SCAL_KERNEL(bounds, kname)                                                                                                       \
    using kname##_traits = ::scal::KernelTraits<decltype(&kname##_body)>;                                                                \
    struct kname##_tag {                                                                                                                 \
        template <class... T>                                                                                                            \
        __device__ __forceinline__ static void call(const T&... t) {                                                                     \
            kname##_body(t...);                                                                                                          \
        }                                                                                                                                \
    };                                                                                                                                   \
    static __global__ void __launch_bounds__(bounds) kname(::scal::Batch<kname##_traits::pack> b) {                                      \
        ::scal::batch_call<kname##_tag>(b.p[blockIdx.z], std::make_index_sequence<kname##_traits::n>());                                 \
    }                                                                                                                                    \
    static hipError_t kname##_launch(const char* name, dim3 grid, dim3 block, int lds, hipStream_t s, int n, const void* const* packs) { \
        return ::scal::batch_launch_impl<kname##_traits::pack>(kname, name, grid, block, lds, s, n, packs);                              \
    }

// one sequence's launch: recorded when a recorder is installed on this thread, a batch of one otherwise
template <class Traits, class... U>
inline void launch_or_record(BatchLaunchFn fn, const char* name, dim3 grid, dim3 block, int lds, hipStream_t s, U&&... u) {
    const typename Traits::pack p = Traits::make(std::forward<U>(u)...);
    static_assert(sizeof(p) <= PACK_BYTES_MAX, "argument set larger than a recorded operation holds");
    if (Recorder* r = g_recorder) {
        RecOp& o = r->add(RecOp::LAUNCH);
        o.fn = fn, o.name = name, o.grid = grid, o.block = block, o.lds = lds, o.stream = s, o.pack_size = static_cast<int>(sizeof(p));
        o.cap = batch_cap<typename Traits::pack>();
        std::memcpy(o.pack, static_cast<const void*>(&p), sizeof(p));
        return;
    }
    const void* one = &p;
    (void)fn(name, grid, block, lds, s, 1, &one);
}
#define SCAL_LAUNCH(name, kname, grid, block, lds, stream, ...) \
    ::scal::launch_or_record<kname##_traits>(kname##_launch, name, grid, block, lds, stream, __VA_ARGS__)

}  // namespace scal
