// graph.hip — capture of a whole batched step (several *_batch calls on several HIP streams) into ONE HIP graph.
//
// Small batches are launch-bound: a step of the per-frame chain is ~90 kernel launches on 4-5 streams, about 0.6 ms of host time however few
// frames it carries (the reference's call pattern is one frame at a time: src/frontend.cpp:302-328, src/loopclosing.cpp:83-121).  The
// *_batch entry points are asynchronous and allocation-free after their first call with a given shape, so a caller can record one step —
// both extractor handles with their FAST gate events, match + triangulation, the DeepLCD -> loop-DB -> BA chain — between
// myslam_graph_begin and myslam_graph_end and replay it with one hipGraphLaunch per step.  What a replay does NOT re-run is host code:
//   * the extractor's FAST-statistics ping-pong (orb_engine.hip run_fast) is frozen at the parity of the captured call: capture TWO
//     consecutive steps and replay them alternately (the second graph reads what the first wrote);
//   * the loop database's per-query row limits live in a pinned host buffer that the captured copy node reads at every replay:
//     myslam_lcddb_update_query_limits rewrites it when ids or the database change (lcddb.hip).
#include <vector>

#include "common.h"

using namespace myslam_hip;

struct StepDbDep { std::shared_ptr<myslam_hip::DbGraphLink> link; uint64_t generation; uint64_t scratch_epoch; int event; };

struct myslam_step_graph {
    hipGraph_t graph = nullptr;          // kept alive beside the executable (ROCm 7.2: see orb_engine.hip HostGraph)
    hipGraphExec_t exec = nullptr;
    size_t nodes = 0;
    std::vector<StepDbDep> deps;         // loop-database scans inside the step: which matrix generation they name (common.h DbGraphLink)
};

namespace myslam_hip {
int upload_table(void* dst, const void* src, size_t bytes) {
    static std::mutex mu;
    static hipStream_t up = nullptr;                    // lives as long as the process
    if (bytes == 0) return MYSLAM_OK;
    std::lock_guard<std::mutex> lk(mu);
    if (!up) MYSLAM_HIP_CHECK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, up));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(up));
    return MYSLAM_OK;
}
hipStream_t host_call_stream() {
    struct Holder {
        hipStream_t s = nullptr;
        ~Holder() { if (s) (void)hipStreamDestroy(s); }
    };
    static thread_local Holder h;
    if (!h.s && hipStreamCreateWithFlags(&h.s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); h.s = nullptr; }
    return h.s;
}
}  // namespace myslam_hip

namespace {
thread_local std::vector<hipEvent_t> t_events;      // fork / join markers of the capture in flight on this thread
thread_local bool t_capturing = false;
thread_local std::vector<StepDbDep> t_deps;         // loop-database contexts whose scans the capture in flight has recorded

int make_event(hipEvent_t* e) {
    MYSLAM_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    t_events.push_back(*e);
    return MYSLAM_OK;
}
void drop_events() {
    for (hipEvent_t e : t_events) (void)hipEventDestroy(e);
    t_events.clear();
}
// the capture that recorded these scans is over (ended, failed or abandoned): their contexts may synchronise their streams again
void close_deps(std::vector<StepDbDep>& deps) {
    for (StepDbDep& d : deps) d.link->captures_open.fetch_sub(1);
}
}  // namespace

namespace myslam_hip {
int graph_note_db_link(const std::shared_ptr<DbGraphLink>& link, uint64_t generation) {
    if (!t_capturing) return MYSLAM_ERR_UNSUPPORTED;
    for (StepDbDep& d : t_deps)
        if (d.link == link) { d.generation = generation; d.scratch_epoch = link->scratch_epoch.load(); return MYSLAM_OK; }
    link->captures_open.fetch_add(1);                                 // until myslam_graph_end (graph.hip close_deps)
    t_deps.push_back({link, generation, link->scratch_epoch.load(), -1});
    return MYSLAM_OK;
}
}  // namespace myslam_hip

extern "C" {

int myslam_graph_begin(void* origin_stream, void* const* side_streams, int n_side) {
    if (!origin_stream || n_side < 0 || (n_side && !side_streams) || t_capturing) return MYSLAM_ERR_INVALID;
    if (prof_is_on()) return MYSLAM_ERR_UNSUPPORTED;                 // the profiling events are host-side bookkeeping: not replayable
    hipStream_t o = (hipStream_t)origin_stream;
    MYSLAM_HIP_CHECK(hipStreamBeginCapture(o, hipStreamCaptureModeThreadLocal));
    t_capturing = true;
    close_deps(t_deps); t_deps.clear();                                // (a capture abandoned without myslam_graph_end on this thread)
    auto fork = [&]() -> int {
        hipEvent_t e;
        int rc = make_event(&e);
        if (rc) return rc;
        MYSLAM_HIP_CHECK(hipEventRecord(e, o));
        for (int i = 0; i < n_side; i++) {
            if (!side_streams[i] || side_streams[i] == origin_stream) return MYSLAM_ERR_INVALID;
            MYSLAM_HIP_CHECK(hipStreamWaitEvent((hipStream_t)side_streams[i], e, 0));       // the side stream joins the capture
        }
        return MYSLAM_OK;
    };
    const int rc = fork();
    if (rc) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(o, &g);
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        drop_events(); t_capturing = false; close_deps(t_deps); t_deps.clear();
    }
    return rc;
}

int myslam_graph_end(void* origin_stream, void* const* side_streams, int n_side, myslam_step_graph** out) {
    if (!origin_stream || !out || n_side < 0 || (n_side && !side_streams) || !t_capturing) return MYSLAM_ERR_INVALID;
    hipStream_t o = (hipStream_t)origin_stream;
    *out = nullptr;
    int rc = MYSLAM_OK;
    for (int i = 0; i < n_side && rc == MYSLAM_OK; i++) {            // every side stream's work joins the origin before the capture ends
        hipEvent_t e;
        rc = make_event(&e);
        if (rc) break;
        if (hipEventRecord(e, (hipStream_t)side_streams[i]) != hipSuccess || hipStreamWaitEvent(o, e, 0) != hipSuccess) rc = MYSLAM_ERR_HIP;
    }
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(o, &g);
    t_capturing = false;
    drop_events();
    std::vector<StepDbDep> deps;
    deps.swap(t_deps);
    close_deps(deps);
    if (rc != MYSLAM_OK || ec != hipSuccess || !g) {
        (void)hipGetLastError();
        if (g) (void)hipGraphDestroy(g);
        return rc != MYSLAM_OK ? rc : MYSLAM_ERR_HIP;
    }
    for (StepDbDep& d : deps) {
        d.event = d.link->add_event();
        if (d.event < 0) {
            for (StepDbDep& e : deps) if (e.event >= 0) e.link->drop_event(e.event);
            (void)hipGraphDestroy(g);
            return MYSLAM_ERR_HIP;
        }
    }
    myslam_step_graph* sg = new myslam_step_graph();
    sg->graph = g;
    sg->deps = std::move(deps);
    (void)hipGraphGetNodes(g, nullptr, &sg->nodes);
    if (hipGraphInstantiate(&sg->exec, g, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(g);
        for (StepDbDep& d : sg->deps) d.link->drop_event(d.event);
        delete sg;
        return MYSLAM_ERR_HIP;
    }
    *out = sg;
    return MYSLAM_OK;
}

int myslam_graph_launch(myslam_step_graph* g, void* hip_stream) {
    if (!g || !g->exec) return MYSLAM_ERR_INVALID;
    // a loop-database scan inside the step names the descriptor matrix by address: refuse the replay when that matrix has moved since
    // (growth beyond its allocation) or its query context is gone — the step has to be recorded again
    for (const StepDbDep& d : g->deps)
        if (d.link->generation.load() != d.generation || d.link->scratch_epoch.load() != d.scratch_epoch) return MYSLAM_ERR_CAPACITY;
    MYSLAM_HIP_CHECK(hipGraphLaunch(g->exec, (hipStream_t)hip_stream));
    for (const StepDbDep& d : g->deps) {           // the context waits for THIS replay, on the stream it actually went to, before it touches its pinned limits
        const int rc = d.link->mark_replay(d.event, (hipStream_t)hip_stream);
        if (rc) return rc;
    }
    return MYSLAM_OK;
}

int myslam_graph_node_count(const myslam_step_graph* g) { return g ? (int)g->nodes : MYSLAM_ERR_INVALID; }

int myslam_graph_destroy(myslam_step_graph* g) {
    if (!g) return MYSLAM_ERR_INVALID;
    for (StepDbDep& d : g->deps) d.link->drop_event(d.event);         // waits for this step's last replay, then the contexts stop synchronising on it
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return MYSLAM_OK;
}

}  // extern "C"
