// ta_probe.hip — what one vector-memory instruction costs the texture addresser / L1 of a gfx950 CU, by access shape.
// The gather-heavy kernels of the ORB path (k_describe2, k_resize_strip, k_blur7_strip) sit at 50-78 % TA_BUSY; this tool measures the
// instruction shapes they use (and the alternatives) on a working set that stays in L1 / L2, so DESIGN.md can price them:
//   every wave issues ITERS loads of one shape from a wave-uniform pseudo-random window origin inside a `span`-byte plane (pitch 1280),
//   8 loads in flight per wave, 8 waves per SIMD; reported: wave-instructions per CU-cycle, bytes per CU-cycle (useful bytes).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/build/ta_probe tools/ta_probe.hip     Run: tools/build/ta_probe > gpurun_out/ta_probe.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int PITCH = 1280;

enum Shape {
    X4_ROWQUAD = 0,      // 16 B per lane, 4 lanes = 64 contiguous bytes of one row at a 16-byte aligned column, 16 rows      (k_describe2 phase C)
    X4_ROWQUAD_AL64,     // same, column aligned to 64 bytes (no quad crosses a 64-byte boundary)
    X4_ROWPAIR,          // 16 B per lane, 2 lanes = 32 contiguous bytes of one row at any column, 32 rows                        (k_describe2 phase A)
    X4_LINEAR,           // 16 B per lane, 64 lanes = 1 KB contiguous, 16-byte aligned
    X2_ROW8,             // 8 B per lane, 8 lanes = 64 contiguous bytes of a row (8-byte aligned column), 8 rows
    X1_ROW16,            // 4 B per lane, 16 lanes = 64 contiguous bytes of a row (4-byte aligned column), 4 rows
    X1_ROW11,            // 4 B per lane, 11 lanes per 44-byte row (the pre-v30 window loads), 5.8 rows
    X1_LINEAR,           // 4 B per lane, 256 contiguous bytes
    U8_SCATTER,          // 1 B per lane at a pseudo-random position of a 37 x 37 window (steered BRIEF straight from memory)
    X4_ROWQUAD_3OF4,     // as x4_rowquad16 with the fourth lane of every quad switched off (48 aligned bytes per row)
    X4_ROWPAIR_A4,       // as x4_rowpair with the column at 4 (mod 16)
    X4_ROWPAIR_A8,       // ... at 8 (mod 16)
    X4_ROWPAIR_A16,      // ... at 0 (mod 16)
    X2_GATHER_BYTE,      // the resize gather as the kernel issued it until round 3: 8 B per lane at BYTE alignment (x = 4.75 lane)
    X3_GATHER,           // 12 B per lane, 4-byte aligned, same stride (the aligned replacement)
    X2_GATHER,           // 8 B per lane at x = 5 lane / 4-ish (the resize kernel's horizontal gather: stride 1.2 pixels x 4)
    NSHAPES
};
static const char* NAMES[NSHAPES] = {"x4_rowquad16", "x4_rowquad_aligned64", "x4_rowpair_anycol", "x4_linear_1KB", "x2_row8", "x1_row16", "x1_row11_44B",
                                     "x1_linear_256B", "u8_window_scatter", "x4_rowquad16_3of4_lanes", "x4_rowpair_col4mod16", "x4_rowpair_col8mod16", "x4_rowpair_col0mod16", "x2_stride_gather_bytealigned", "x3_stride_gather_dwordaligned", "x2_stride_gather"};
static const int USEFUL[NSHAPES] = {1024, 1024, 1024, 1024, 512, 256, 256, 256, 64, 768, 1024, 1024, 1024, 512, 768, 512};

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int SHAPE>
__global__ __launch_bounds__(256) void k_probe(const uint8_t* __restrict__ plane, uint32_t rowmask, int iters, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t swid = (uint32_t)__builtin_amdgcn_readfirstlane((int)wid);
    uint32_t acc = 0;
    uint32_t off;                                            // lane offset inside the window
    if (SHAPE == X4_ROWQUAD || SHAPE == X4_ROWQUAD_AL64 || SHAPE == X4_ROWQUAD_3OF4) off = (lane >> 2) * PITCH + 16 * (lane & 3);
    else if (SHAPE == X4_ROWPAIR || SHAPE == X4_ROWPAIR_A4 || SHAPE == X4_ROWPAIR_A8 || SHAPE == X4_ROWPAIR_A16) off = (lane >> 1) * PITCH + 16 * (lane & 1);
    else if (SHAPE == X4_LINEAR) off = 16 * lane;
    else if (SHAPE == X2_ROW8) off = (lane >> 3) * PITCH + 8 * (lane & 7);
    else if (SHAPE == X1_ROW16) off = (lane >> 4) * PITCH + 4 * (lane & 15);
    else if (SHAPE == X1_ROW11) off = (lane / 11) * PITCH + 4 * (lane % 11);
    else if (SHAPE == X1_LINEAR) off = 4 * lane;
    else if (SHAPE == U8_SCATTER) off = (hash32(lane * 7919u + 13u) % 37u) * PITCH + (hash32(lane * 104729u + 7u) % 37u);
    else if (SHAPE == X2_GATHER_BYTE) off = (lane * 19) >> 2;
    else off = ((lane * 19) >> 4) * 4;                      // X2_GATHER: ~4.75 bytes per lane step, 4-byte aligned 8-byte loads
    for (int it = 0; it < iters; it += 8) {
        uint32_t v[8][4];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            // wave-uniform origin on the scalar unit (wid is made scalar below), power-of-two ranges: no divisions in the loop
            const uint32_t h = hash32(swid * 65537u + (uint32_t)(it + u));
            uint32_t row = h & rowmask, col = (h >> 20) & 511u;
            if (SHAPE == X4_ROWQUAD || SHAPE == X4_LINEAR || SHAPE == X4_ROWQUAD_3OF4) col &= ~15u;
            if (SHAPE == X4_ROWQUAD_AL64) col &= ~63u;
            if (SHAPE == X4_ROWPAIR_A4) col = (col & ~15u) | 4u;
            if (SHAPE == X4_ROWPAIR_A8) col = (col & ~15u) | 8u;
            if (SHAPE == X4_ROWPAIR_A16) col &= ~15u;
            if (SHAPE == X2_ROW8) col &= ~7u;
            if (SHAPE == X1_ROW16 || SHAPE == X1_ROW11 || SHAPE == X1_LINEAR || SHAPE == X2_GATHER || SHAPE == X3_GATHER || SHAPE == X2_GATHER_BYTE) col &= ~3u;
            const uint8_t* p = plane + (size_t)row * PITCH + col + off;
            if (SHAPE == X4_ROWQUAD_3OF4) { uint4 t = make_uint4(0, 0, 0, 0); if ((lane & 3) != 3) __builtin_memcpy(&t, p, 16); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
            else if (SHAPE <= X4_LINEAR || SHAPE == X4_ROWPAIR_A4 || SHAPE == X4_ROWPAIR_A8 || SHAPE == X4_ROWPAIR_A16) { uint4 t; __builtin_memcpy(&t, p, 16); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
            else if (SHAPE == X3_GATHER) { uint32_t t[3]; __builtin_memcpy(t, __builtin_assume_aligned(p, 4), 12); v[u][0] = t[0]; v[u][1] = t[1]; v[u][2] = t[2]; v[u][3] = 0; }
            else if (SHAPE == X2_ROW8 || SHAPE == X2_GATHER || SHAPE == X2_GATHER_BYTE) { uint2 t; __builtin_memcpy(&t, p, 8); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = 0; v[u][3] = 0; }
            else if (SHAPE == U8_SCATTER) { v[u][0] = *p; v[u][1] = v[u][2] = v[u][3] = 0; }
            else { uint32_t t; __builtin_memcpy(&t, p, 4); v[u][0] = t; v[u][1] = v[u][2] = v[u][3] = 0; }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x12345678u) out[wid] = acc;
}

template <int SHAPE>
static void run(const uint8_t* plane, uint32_t rowmask, uint32_t* d_out, int cus, double ghz, const char* setname, bool last) {
    const int iters = 4096, grid = cus * 8;                  // 8 blocks x 4 waves = 8 waves per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_probe<SHAPE>, dim3(grid), dim3(256), 0, 0, plane, rowmask, 256, d_out);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe<SHAPE>, dim3(grid), dim3(256), 0, 0, plane, rowmask, iters, d_out);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double insts_per_cu = (double)iters * 32.0;        // 32 waves per CU
    const double cyc = best * 1e-3 * ghz * 1e9;
    printf("  \"%s\": {\"ms\": %.3f, \"cycles_per_wave_instruction_per_cu\": %.2f, \"useful_bytes_per_cu_cycle\": %.1f}%s\n", NAMES[SHAPE], best,
           cyc / insts_per_cu, USEFUL[SHAPE] * insts_per_cu / cyc, last ? "" : ",");
    (void)setname;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    uint32_t* d_out; CHECK(hipMalloc(&d_out, (size_t)cus * 32 * 4 + 4096));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz_reported\": %.3f, \"waves_per_simd\": 8, \"loads_in_flight_per_wave\": 8,\n", p.name, cus, ghz);
    const uint32_t rowsets[3] = {64, 1024, 262144};          // rows of 1280 bytes (+ 40 rows of slack): 0.1 MB (L1 / L2), 1.3 MB (L2), 335 MB (infinity cache / HBM)
    const char* setn[3] = {"plane_0.1MB", "plane_1.3MB", "plane_335MB"};
    for (int s = 0; s < 3; s++) {
        const size_t bytes = (size_t)(rowsets[s] + 40) * PITCH + 4096;
        uint8_t* plane; CHECK(hipMalloc(&plane, bytes)); CHECK(hipMemset(plane, 1, bytes));
        printf(" \"%s\": {\n", setn[s]);
        run<X4_ROWQUAD>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X4_ROWQUAD_AL64>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X4_ROWPAIR>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X4_LINEAR>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X2_ROW8>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X1_ROW16>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X1_ROW11>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X1_LINEAR>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<U8_SCATTER>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X4_ROWQUAD_3OF4>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X4_ROWPAIR_A4>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X4_ROWPAIR_A8>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X4_ROWPAIR_A16>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X2_GATHER_BYTE>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X3_GATHER>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], false);
        run<X2_GATHER>(plane, rowsets[s] - 1, d_out, cus, ghz, setn[s], true);
        printf(" }%s\n", s == 2 ? "" : ",");
        CHECK(hipFree(plane));
    }
    printf("}\n");
    return 0;
}
