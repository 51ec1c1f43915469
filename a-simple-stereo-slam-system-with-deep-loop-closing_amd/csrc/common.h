// common.h — internal helpers shared by the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

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

namespace myslam_hip {

constexpr int WAVE = 64;

// ---- per-kernel event timing (bench.py's roofline leg reads it through the C ABI) ----
struct ProfSlot { const char* name; double ms; long calls; };
void prof_begin(int id, hipStream_t s);
void prof_end(int id, hipStream_t s);
enum ProfId {
    P_RESIZE = 0, P_FAST, P_OCTREE, P_BLUR, P_DESC, P_MATCH, P_TRI, P_LCD_PRE, P_CONV1, P_CONV2,
    P_CONV3, P_DBSCAN, P_BA, P_SCREEN, P_COUNT
};

struct ScopedProf {
    int id; hipStream_t s;
    ScopedProf(int id_, hipStream_t s_) : id(id_), s(s_) { prof_begin(id, s); }
    ~ScopedProf() { prof_end(id, s); }
};

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace myslam_hip
