// common.h — internal helpers shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <vector>
#include <stdio.h>
#include <string.h>

#include "../../include/myslam_hip.h"

#define MYSLAM_HIP_CHECK(expr)                                                              \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            fprintf(stderr, "[myslam_hip] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr,    \
                    hipGetErrorString(_e));                                                 \
            return MYSLAM_ERR_HIP;                                                          \
        }                                                                                   \
    } while (0)

// Wave priority of everything that is NOT the grid-FAST kernel (experiment of round 4): FAST is the VALU-issue
// bound kernel and is resident during the whole step; the kernels that run under it are latency / texture-addresser / matrix-pipe bound.
// With a raised priority their (few) ready instructions issue ahead of FAST's, so they leave the CU sooner and FAST fills what is left.
#ifndef MYSLAM_SIDE_PRIO_LEVEL
#define MYSLAM_SIDE_PRIO_LEVEL 0
#endif
#if MYSLAM_SIDE_PRIO_LEVEL > 0
#define MYSLAM_SIDE_PRIO() __builtin_amdgcn_s_setprio(MYSLAM_SIDE_PRIO_LEVEL)
#else
#define MYSLAM_SIDE_PRIO() ((void)0)
#endif

namespace myslam_hip {

constexpr int WAVE = 64;

// ---- per-kernel event timing (bench.py's roofline leg reads it through the C ABI) ----
struct ProfSlot { const char* name; double ms; long calls; };
bool prof_is_on();
void prof_begin(int id, hipStream_t s);
void prof_end(int id, hipStream_t s);
enum ProfId {
    P_RESIZE = 0, P_FAST, P_OCTREE, P_BLUR, P_DESC, P_MATCH, P_TRI, P_LCD_PRE, P_CONV1, P_CONV2,
    P_CONV3, P_DBSCAN, P_BA, P_SCREEN, P_POOL2, P_COUNT
};

struct ScopedProf {
    int id; hipStream_t s;
    ScopedProf(int id_, hipStream_t s_) : id(id_), s(s_) { prof_begin(id, s); }
    ~ScopedProf() { prof_end(id, s); }
};

// ---- table uploads that stay off the legacy stream ----------------------------------------------------------------------------
// Host table -> a device buffer no stream uses yet (plan tables, resize tables), complete on return.  A plain hipMemcpy runs on the legacy NULL stream, which
// synchronises with every BLOCKING stream of the process: while another thread records a graph on such a stream it fails ("operation would make the legacy
// stream depend on a capturing blocking stream") and poisons that thread's recording (tests/test_gpu_threads.py: one thread makes a plan while another
// records its one-frame call).  This one copies on a process-wide NON-BLOCKING stream and waits for that stream only.
int upload_table(void* dst, const void* src, size_t bytes);
// The stream of the synchronous host-pointer entry points that bring no stream of their own (PnP-RANSAC, pose graph, map-point correction): one NON-BLOCKING
// stream per calling thread, created on first use — kernels, copies and memsets of such a call all go through it, and the call synchronises it before it reads
// a result or returns.  nullptr = the stream could not be created.
hipStream_t host_call_stream();
// copy through `st`, complete on return (pageable host memory on either side)
inline int copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    if (bytes == 0) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, st));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(st));
    return MYSLAM_OK;
}

// ---- staging of the host-pointer ("drop-in", B = 1) entry points --------------------------------------------------------------
// One grow-only device block, one pinned host block and one stream per calling THREAD, carved into 256-byte aligned pieces per
// call: no hipMalloc / hipFree per call, one host->device and one device->host copy per call, and nothing to free on an error
// return (the arena owns the memory).  Pieces are laid out [in][inout][out][tmp]; `upload()` copies in + inout, `download()` copies
// inout + out back and synchronises.  Usage: register pieces, upload(), launch on stream() with dev<T>(piece), download().
struct HostArena {
    uint8_t* d = nullptr; uint8_t* h = nullptr; size_t cap = 0; hipStream_t s = nullptr;
    int dev = -1;               // the device block, stream and pinned block belong to: a thread that switches devices gets a fresh arena
    ~HostArena();
    void release();
    int ensure(size_t bytes);
};
HostArena& host_arena();

class HostCall {
  public:
    enum Kind { IN = 0, INOUT = 1, OUT = 2, TMP = 3 };
    HostCall() : A(host_arena()) {}
    template <class T> int in(const T* p, size_t n) { return add(IN, p, nullptr, n * sizeof(T)); }
    template <class T> int inout(T* p, size_t n) { return add(INOUT, p, p, n * sizeof(T)); }
    template <class T> int out(T* p, size_t n) { return add(OUT, nullptr, p, n * sizeof(T)); }
    template <class T> int tmp(size_t n) { return add(TMP, nullptr, nullptr, n * sizeof(T)); }
    int upload();
    int download();
    template <class T> T* dev(int piece) const { return reinterpret_cast<T*>(A.d + pc[piece].off); }
    hipStream_t stream() const { return A.s; }

  private:
    struct Piece { Kind kind; const void* src; void* dst; size_t bytes, off; };
    int add(Kind k, const void* src, void* dst, size_t bytes) {
        if (npc >= MAXP) { overflow = true; return 0; }           // upload() refuses the call; piece 0 keeps dev<T>() in bounds until then
        pc[npc] = {k, src, dst, bytes, 0};
        return npc++;
    }
    static constexpr int MAXP = 24;
    HostArena& A;
    Piece pc[MAXP]; int npc = 0; bool overflow = false;
    size_t endIn = 0, begOut = 0, endOut = 0;
};

// ---- what a recorded step (graph.hip) and a loop-database query context (lcddb.hip) know about each other ------------------------
// A scan recorded into a HIP graph names the descriptor matrix by address and reads its row limits from the context's pinned buffer at
// every replay, on whatever stream the replay is launched.  The link is shared (shared_ptr) by the context and by every step that
// captured one of its scans:
//   * `generation` = generation of the matrix the context scans NOW (UINT64_MAX once the context is gone); a step remembers the
//     generation it was recorded against and myslam_graph_launch refuses to replay it when they differ (MYSLAM_ERR_CAPACITY);
//   * every recorded step owns one event here, recorded behind each of its replays ON THE LAUNCH STREAM; the context waits for all of
//     them (wait()) before it rewrites the pinned limits, frees scratch or lets the matrix move.
struct DbGraphLink {
    std::atomic<uint64_t> generation{0};
    // bumped when the CONTEXT's own scratch (pinned row limits, partial results, shard scratch) is reallocated while a recorded step names it: the
    // matrix generation is unchanged then, so a step remembers this number as well and a launch with another one is refused (round 6)
    std::atomic<uint64_t> scratch_epoch{0};
    // captures (myslam_graph_begin .. _end) that have recorded a scan of this context and are still open: while one is, nobody may synchronise the
    // context's stream (hipStreamSynchronize on a capturing stream invalidates the capture) — growth / scratch changes are refused instead
    std::atomic<int> captures_open{0};
    std::mutex mu;
    std::vector<hipEvent_t> events;        // one per recorded step that is still alive; a destroyed step's slot is nullptr and is reused
    std::vector<char> launched;
    ~DbGraphLink() { for (hipEvent_t e : events) if (e) (void)hipEventDestroy(e); }
    void invalidate() { generation.store(UINT64_MAX); }
    int add_event() {                      // -> index of a new event, or -1
        std::lock_guard<std::mutex> lk(mu);
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return -1; }
        for (size_t i = 0; i < events.size(); i++)
            if (!events[i]) { events[i] = e; launched[i] = 0; return (int)i; }
        events.push_back(e); launched.push_back(0);
        return (int)events.size() - 1;
    }
    // the step that owned this event is gone (myslam_graph_destroy): its last replay is waited for once, then nobody synchronises on it again
    void drop_event(int idx) {
        std::lock_guard<std::mutex> lk(mu);
        if (idx < 0 || idx >= (int)events.size() || !events[idx]) return;
        if (launched[idx]) (void)hipEventSynchronize(events[idx]);
        (void)hipEventDestroy(events[idx]);
        events[idx] = nullptr; launched[idx] = 0;
    }
    int mark_replay(int idx, hipStream_t s) {
        std::lock_guard<std::mutex> lk(mu);
        if (idx < 0 || idx >= (int)events.size() || !events[idx]) return MYSLAM_ERR_INVALID;
        if (hipEventRecord(events[idx], s) != hipSuccess) { (void)hipGetLastError(); return MYSLAM_ERR_HIP; }
        launched[idx] = 1;
        return MYSLAM_OK;
    }
    int wait() {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < events.size(); i++)
            if (events[i] && launched[i] && hipEventSynchronize(events[i]) != hipSuccess) { (void)hipGetLastError(); return MYSLAM_ERR_HIP; }
        return MYSLAM_OK;
    }
    size_t live_events() {
        std::lock_guard<std::mutex> lk(mu);
        size_t n = 0;
        for (hipEvent_t e : events) n += e != nullptr;
        return n;
    }
};
// called by a context whose scan is being captured: the step being recorded on this thread (myslam_graph_begin) takes note.
// MYSLAM_ERR_UNSUPPORTED when the capture was not started by myslam_graph_begin on this thread (nobody would check the generation).
int graph_note_db_link(const std::shared_ptr<DbGraphLink>& link, uint64_t generation);

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace myslam_hip
