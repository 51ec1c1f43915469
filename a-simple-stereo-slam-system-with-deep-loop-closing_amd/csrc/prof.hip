// prof.hip — per-kernel HIP-event timing + library-wide entry points.
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace myslam_hip {

static const char* kNames[P_COUNT] = {"resize", "fast", "octree", "blur7", "describe", "hamming_match", "triangulate",
                                      "lcd_preproc", "calc_conv1", "calc_conv2", "calc_conv3", "lcddb_scan", "ba_build", "screen", "calc_pool2"};
struct Pending { int id; hipEvent_t a, b; };
static std::atomic<bool> g_on{false};
static std::mutex g_mu;
static std::vector<Pending> g_pending;
static std::vector<hipEvent_t> g_pool;
static double g_ms[P_COUNT];
static long g_calls[P_COUNT];
static thread_local hipEvent_t t_start[P_COUNT];

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

bool prof_is_on() { return g_on.load(std::memory_order_relaxed); }

void prof_begin(int id, hipStream_t s) {
    if (!g_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on.load(std::memory_order_relaxed)) return;
    t_start[id] = get_event();
    (void)hipEventRecord(t_start[id], s);
}

void prof_end(int id, hipStream_t s) {
    if (!t_start[id]) return;                 // profiling was off (or switched on mid-call) when the launch began: nothing to pair
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t e = get_event();
    (void)hipEventRecord(e, s);
    g_pending.push_back({id, t_start[id], e});
    t_start[id] = nullptr;
}

// ---- thread-local staging of the host-pointer entry points (common.h) ----
void HostArena::release() {
    if (s) (void)hipStreamSynchronize(s);
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    if (s) (void)hipStreamDestroy(s);
    d = nullptr; h = nullptr; s = nullptr; cap = 0;
}

HostArena::~HostArena() { release(); }

int HostArena::ensure(size_t bytes) {
    int cur = 0;
    MYSLAM_HIP_CHECK(hipGetDevice(&cur));
    if (cur != dev) { release(); dev = cur; }          // the calling thread selected another device since its last call
    if (!s) MYSLAM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (bytes <= cap) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (d) { (void)hipFree(d); d = nullptr; }
    if (h) { (void)hipHostFree(h); h = nullptr; }
    cap = 0;
    const size_t want = (bytes + (bytes >> 2) + 65535) & ~(size_t)65535;
    MYSLAM_HIP_CHECK(hipMalloc((void**)&d, want));
    MYSLAM_HIP_CHECK(hipHostMalloc((void**)&h, want));
    cap = want;
    return MYSLAM_OK;
}

HostArena& host_arena() {
    static thread_local HostArena a;
    return a;
}

int HostCall::upload() {
    if (overflow) return MYSLAM_ERR_CAPACITY;          // more pieces than a call may register
    size_t off = 0;
    for (int k = IN; k <= TMP; k++) {
        if (k == INOUT) begOut = off;
        for (int i = 0; i < npc; i++)
            if (pc[i].kind == k) { pc[i].off = off; off += (pc[i].bytes + 255) & ~(size_t)255; }
        if (k == INOUT) endIn = off;
        if (k == OUT) endOut = off;
    }
    int rc = A.ensure(off ? off : 256);
    if (rc) return rc;
    for (int i = 0; i < npc; i++)
        if (pc[i].kind <= INOUT && pc[i].bytes) memcpy(A.h + pc[i].off, pc[i].src, pc[i].bytes);
    if (endIn) MYSLAM_HIP_CHECK(hipMemcpyAsync(A.d, A.h, endIn, hipMemcpyHostToDevice, A.s));
    return MYSLAM_OK;
}

int HostCall::download() {
    if (endOut > begOut) MYSLAM_HIP_CHECK(hipMemcpyAsync(A.h + begOut, A.d + begOut, endOut - begOut, hipMemcpyDeviceToHost, A.s));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(A.s));
    for (int i = 0; i < npc; i++)
        if ((pc[i].kind == INOUT || pc[i].kind == OUT) && pc[i].bytes && pc[i].dst) memcpy(pc[i].dst, A.h + pc[i].off, pc[i].bytes);
    return MYSLAM_OK;
}

static void drain() {
    for (auto& p : g_pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { g_ms[p.id] += ms; g_calls[p.id]++; }
        g_pool.push_back(p.a); g_pool.push_back(p.b);
    }
    g_pending.clear();
}

// one wave spins for `ticks` of the constant 100 MHz clock and reports how many SHADER cycles went by: the clock the chip runs at right now
__global__ void k_clock_probe(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

}  // namespace myslam_hip

using namespace myslam_hip;

extern "C" {

int myslam_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// build.py passes a digest of every source the library is built from: profiles taken on one build can be told from another's (bench.py: roofline.traffic_stale)
#ifndef MYSLAM_BUILD_ID
#define MYSLAM_BUILD_ID "unidentified"
#endif
const char* myslam_hip_version(void) { return "myslam_hip 0.6 (gfx950) build " MYSLAM_BUILD_ID; }

int myslam_prof_shader_clock_mhz(void* hip_stream, float spin_us, float* mhz) {
    if (!mhz || !(spin_us > 0.f) || spin_us > 1e6f) return MYSLAM_ERR_INVALID;
    static thread_local unsigned long long* d_t = nullptr;            // two counters, allocated once per calling thread
    if (!d_t) MYSLAM_HIP_CHECK(hipMalloc((void**)&d_t, 16));
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, (unsigned long long)(spin_us * 100.f), d_t);
    MYSLAM_HIP_CHECK(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h, d_t, 16, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize((hipStream_t)hip_stream));
    *mhz = h[1] ? (float)((double)h[0] / (double)h[1] * 100.0) : 0.f;      // shader cycles per 10 ns tick x 100 MHz
    return MYSLAM_OK;
}

int myslam_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!on) drain();
    g_on.store(on != 0);
    return MYSLAM_OK;
}

int myslam_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain();
    for (int i = 0; i < P_COUNT; i++) { g_ms[i] = 0; g_calls[i] = 0; }
    return MYSLAM_OK;
}

int myslam_prof_count(void) { return P_COUNT; }

// debug: copy n bytes from a device pointer with this library's HIP runtime (diagnoses runtime/VA mismatches)
int myslam_debug_peek(const void* d_ptr, void* out, size_t n) {
    MYSLAM_HIP_CHECK(hipMemcpy(out, d_ptr, n, hipMemcpyDeviceToHost));
    return MYSLAM_OK;
}

int myslam_prof_get(int i, const char** name, double* total_ms, long* launches) {
    if (i < 0 || i >= P_COUNT) return MYSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_mu);
    drain();
    if (name) *name = kNames[i];
    if (total_ms) *total_ms = g_ms[i];
    if (launches) *launches = g_calls[i];
    return MYSLAM_OK;
}

}  // extern "C"
