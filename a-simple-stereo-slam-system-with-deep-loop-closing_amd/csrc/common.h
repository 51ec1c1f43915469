// common.h — internal helpers shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace myslam_hip
