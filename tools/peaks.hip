// tools/peaks.hip — measures the peaks bench.py's roofline objects are normalised against, on the box the bench runs on.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/peaks tools/peaks.hip && tools/build/peaks > profiles/r03_peaks.json
//   (tools/peaks.py does both and is what tools/gpu_round.sh calls)
//
// What it measures (all on one MI355X, default stream, HIP-event timed, best of `REPS` launches):
//   * hbm:   float4 copy / read-only / write-only GB/s on buffers far beyond the 256 MiB Infinity Cache
//   * pcie:  pinned host -> device and device -> host GB/s (the streamed-input bench line's ceiling)
//   * valu:  ISSUE rate of the instruction classes the ORB kernels are made of, each as an unrolled stream of 8 independent
//            dependency chains, at 1 / 2 / 4 / 8 waves per SIMD and oversubscribed (8 waves per SIMD x 4 rounds):
//            T lane-op/s (= wave-instructions x 64 / s) and cycles per wave-instruction per SIMD at the measured shader clock
//   * mfma:  dense rate of the matrix instructions the path uses (bf16 32x32x16, i8 16x16x64, fp4 scaled 32x32x64, f64 16x16x4)
//   * clock: shader cycles (s_memtime) per 100 MHz constant-clock tick (s_memrealtime) inside the VALU kernels
// Output: one JSON object on stdout.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #e, hipGetErrorString(_e)); exit(2); } } while (0)

static const int REPS = 5;

// ------------------------------------------------------------------------------------------------------------------------------
// VALU issue rate.  One asm block = 4 x 8 instructions over 8 independent registers (%0..%7), operands %8 / %9 are loop invariant.
#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define R32(I) R8(I) R8(I) R8(I) R8(I)
constexpr int INSTS_PER_BLOCK = 32, BLOCKS_PER_ITER = 4;

#define VALU_KERNEL(NAME, T, I)                                                                                                    \
    __global__ __launch_bounds__(256) void k_##NAME(uint64_t* out, int iters, T seed, T b, T c) {                                  \
        T a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;       \
        asm volatile("s_mov_b32 vcc_lo, 0x55555555\ns_mov_b32 vcc_hi, 0x55555555\ns_mov_b32 s20, 0x33333333\ns_mov_b32 s21, 0x33333333" ::: "vcc", "s20", "s21");           \
        uint64_t t0 = __builtin_readcyclecounter(), w0 = wall_clock64();                                                           \
        for (int i = 0; i < iters; i++) {                                                                                          \
            _Pragma("unroll") for (int u = 0; u < BLOCKS_PER_ITER; u++)                                                            \
                asm volatile(R32(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
        }                                                                                                                          \
        uint64_t t1 = __builtin_readcyclecounter(), w1 = wall_clock64();                                                           \
        T s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                               \
        if (s == (T)0x12345 && threadIdx.x == 999) out[2] = (uint64_t)s;             /* keeps the chains alive */                  \
        if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }                                           \
    }

#define I_PKMAX(d) "v_pk_max_i16 %" #d ", %" #d ", %8\n"
#define I_PKMIN(d) "v_pk_min_i16 %" #d ", %" #d ", %8\n"
#define I_PKMAX3H(d) "v_pk_maximum3_f16 %" #d ", %" #d ", %8, %9\n"
#define I_PKMIN3H(d) "v_pk_minimum3_f16 %" #d ", %" #d ", %8, %9\n"
#define I_PKADD16(d) "v_pk_add_u16 %" #d ", %" #d ", %8\n"
#define I_PKSUB16C(d) "v_pk_sub_u16 %" #d ", %" #d ", %8 clamp\n"
#define I_PERM(d) "v_perm_b32 %" #d ", %" #d ", %8, %9\n"
#define I_ALIGNB(d) "v_alignbyte_b32 %" #d ", %" #d ", %8, 1\n"
#define I_DOT4(d) "v_dot4_u32_u8 %" #d ", %8, %9, %" #d "\n"
#define I_DOT2(d) "v_dot2_u32_u16 %" #d ", %8, %9, %" #d "\n"
#define I_ADD32(d) "v_add_u32 %" #d ", %" #d ", %8\n"
#define I_MAX3U(d) "v_max3_u32 %" #d ", %" #d ", %8, %9\n"
#define I_ANDOR(d) "v_and_or_b32 %" #d ", %" #d ", %8, %9\n"
#define I_LSHLOR(d) "v_lshl_or_b32 %" #d ", %" #d ", 1, %9\n"
#define I_MULLO(d) "v_mul_lo_u32 %" #d ", %" #d ", %8\n"
#define I_MULHI24(d) "v_mul_hi_u32_u24 %" #d ", %" #d ", %8\n"
#define I_MAD24(d) "v_mad_u32_u24 %" #d ", %" #d ", %8, %9\n"
#define I_BCNT(d) "v_bcnt_u32_b32 %" #d ", %8, %" #d "\n"
#define I_FMA32(d) "v_fma_f32 %" #d ", %" #d ", %8, %9\n"
#define I_PKFMA32(d) "v_pk_fma_f32 %" #d ", %" #d ", %8, %9\n"
#define I_FMA64(d) "v_fma_f64 %" #d ", %" #d ", %8, %9\n"
#define I_PKFMA16(d) "v_pk_fma_f16 %" #d ", %" #d ", %8, %9\n"
#define I_CNDMASK(d) "v_cndmask_b32 %" #d ", %" #d ", %8, vcc\n"
#define I_CMPSAT(d) "v_sub_u16 %" #d ", %" #d ", %8 clamp\n"
#define I_MAXU16(d) "v_max_u16 %" #d ", %" #d ", %8\n"
#define I_MINU16(d) "v_min_u16 %" #d ", %" #d ", %8\n"
#define I_MAXU32(d) "v_max_u32 %" #d ", %" #d ", %8\n"
#define I_AND(d) "v_and_b32 %" #d ", %" #d ", %8\n"
#define I_OR(d) "v_or_b32 %" #d ", %" #d ", %8\n"
#define I_XOR(d) "v_xor_b32 %" #d ", %" #d ", %8\n"
#define I_LSHL(d) "v_lshlrev_b32 %" #d ", 1, %" #d "\n"
#define I_LSHR(d) "v_lshrrev_b32 %" #d ", 1, %" #d "\n"
#define I_SUB32(d) "v_sub_u32 %" #d ", %" #d ", %8\n"
#define I_BFE(d) "v_bfe_u32 %" #d ", %" #d ", 1, 31\n"
#define I_ADD3(d) "v_add3_u32 %" #d ", %" #d ", %8, %9\n"
#define I_OR3(d) "v_or3_b32 %" #d ", %" #d ", %8, %9\n"
#define I_LSHLADD(d) "v_lshl_add_u32 %" #d ", %" #d ", 1, %9\n"
#define I_MED3(d) "v_med3_u32 %" #d ", %" #d ", %8, %9\n"
#define I_MAX3U16(d) "v_max3_u16 %" #d ", %" #d ", %8, %9\n"
#define I_MIN3F16(d) "v_min3_f16 %" #d ", %" #d ", %8, %9\n"
#define I_SDWAMAX(d) "v_max_u16_sdwa %" #d ", %" #d ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2\n"
#define I_DPPMOV(d) "v_mov_b32_dpp %" #d ", %" #d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPPADD(d) "v_add_u32_dpp %" #d ", %" #d ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CMPCND(d) "v_cmp_gt_u32 vcc, %" #d ", %8\nv_cndmask_b32 %" #d ", %" #d ", %9, vcc\n"
#define I_CMP(d) "v_cmp_gt_u32 vcc, %" #d ", %8\n"
#define I_CND(d) "v_cndmask_b32 %" #d ", %" #d ", %9, vcc\n"
#define I_CND_E64(d) "v_cndmask_b32_e64 %" #d ", %" #d ", %9, s[20:21]\n"
#define I_CND_NODEP(d) "v_cndmask_b32 %" #d ", %8, %9, vcc\n"
#define I_ADDC(d) "v_addc_co_u32 %" #d ", vcc, %" #d ", %8, vcc\n"
#define I_CMPCND64(d) "v_cmp_gt_u32_e64 s[20:21], %" #d ", %8\nv_cndmask_b32_e64 %" #d ", %" #d ", %9, s[20:21]\n"
#define I_CNDADD(d) "v_cndmask_b32 %" #d ", %" #d ", %9, vcc\nv_add_u32 %" #d ", %" #d ", %8\nv_add_u32 %" #d ", %" #d ", %8\nv_add_u32 %" #d ", %" #d ", %8\n"
#define I_PKMULLO(d) "v_pk_mul_lo_u16 %" #d ", %" #d ", %8\n"
#define I_PKMAD(d) "v_pk_mad_u16 %" #d ", %" #d ", %8, %9\n"
#define I_CVTUB(d) "v_cvt_f32_ubyte0 %" #d ", %" #d "\n"
#define I_SADU8(d) "v_sad_u8 %" #d ", %8, %9, %" #d "\n"
#define I_MUL32F(d) "v_mul_f32 %" #d ", %" #d ", %8\n"
#define I_ADD32F(d) "v_add_f32 %" #d ", %" #d ", %8\n"
#define I_PKMULF(d) "v_pk_mul_f32 %" #d ", %" #d ", %8\n"
#define I_RSQ(d) "v_rsq_f32 %" #d ", %" #d "\n"
#define I_RCP(d) "v_rcp_f32 %" #d ", %" #d "\n"
#define I_SQRT(d) "v_sqrt_f32 %" #d ", %" #d "\n"
#define I_MULF64(d) "v_mul_f64 %" #d ", %" #d ", %8\n"
#define I_ADDF64(d) "v_add_f64 %" #d ", %" #d ", %8\n"
#define I_CVTBF16(d) "v_cvt_pk_bf16_f32 %" #d ", %" #d ", %8\n"
#define I_MOV(d) "v_mov_b32 %" #d ", %8\n"
#define I_READLANE(d) "v_readlane_b32 s20, %" #d ", 3\n"

VALU_KERNEL(pk_max_i16, uint32_t, I_PKMAX)
VALU_KERNEL(pk_min_i16, uint32_t, I_PKMIN)
VALU_KERNEL(pk_maximum3_f16, uint32_t, I_PKMAX3H)
VALU_KERNEL(pk_minimum3_f16, uint32_t, I_PKMIN3H)
VALU_KERNEL(pk_add_u16, uint32_t, I_PKADD16)
VALU_KERNEL(pk_sub_u16_clamp, uint32_t, I_PKSUB16C)
VALU_KERNEL(perm_b32, uint32_t, I_PERM)
VALU_KERNEL(alignbyte_b32, uint32_t, I_ALIGNB)
VALU_KERNEL(dot4_u32_u8, uint32_t, I_DOT4)
VALU_KERNEL(dot2_u32_u16, uint32_t, I_DOT2)
VALU_KERNEL(add_u32, uint32_t, I_ADD32)
VALU_KERNEL(max3_u32, uint32_t, I_MAX3U)
VALU_KERNEL(and_or_b32, uint32_t, I_ANDOR)
VALU_KERNEL(lshl_or_b32, uint32_t, I_LSHLOR)
VALU_KERNEL(mul_lo_u32, uint32_t, I_MULLO)
VALU_KERNEL(mul_hi_u32_u24, uint32_t, I_MULHI24)
VALU_KERNEL(mad_u32_u24, uint32_t, I_MAD24)
VALU_KERNEL(bcnt_u32_b32, uint32_t, I_BCNT)
VALU_KERNEL(cndmask_b32, uint32_t, I_CNDMASK)
VALU_KERNEL(sub_u16_clamp, uint32_t, I_CMPSAT)
VALU_KERNEL(max_u16, uint32_t, I_MAXU16)
VALU_KERNEL(min_u16, uint32_t, I_MINU16)
VALU_KERNEL(max_u32, uint32_t, I_MAXU32)
VALU_KERNEL(and_b32, uint32_t, I_AND)
VALU_KERNEL(or_b32, uint32_t, I_OR)
VALU_KERNEL(xor_b32, uint32_t, I_XOR)
VALU_KERNEL(lshlrev_b32, uint32_t, I_LSHL)
VALU_KERNEL(lshrrev_b32, uint32_t, I_LSHR)
VALU_KERNEL(sub_u32, uint32_t, I_SUB32)
VALU_KERNEL(bfe_u32, uint32_t, I_BFE)
VALU_KERNEL(add3_u32, uint32_t, I_ADD3)
VALU_KERNEL(or3_b32, uint32_t, I_OR3)
VALU_KERNEL(lshl_add_u32, uint32_t, I_LSHLADD)
VALU_KERNEL(med3_u32, uint32_t, I_MED3)
VALU_KERNEL(max3_u16, uint32_t, I_MAX3U16)
VALU_KERNEL(min3_f16, uint32_t, I_MIN3F16)
VALU_KERNEL(max_u16_sdwa, uint32_t, I_SDWAMAX)
VALU_KERNEL(mov_b32_dpp, uint32_t, I_DPPMOV)
VALU_KERNEL(add_u32_dpp, uint32_t, I_DPPADD)
VALU_KERNEL(cmp_gt_u32_plus_cndmask, uint32_t, I_CMPCND)
VALU_KERNEL(cmp_gt_u32, uint32_t, I_CMP)
VALU_KERNEL(cndmask_b32_vcc_set, uint32_t, I_CND)
VALU_KERNEL(cndmask_b32_e64_sgpr_mask, uint32_t, I_CND_E64)
VALU_KERNEL(cndmask_b32_no_dependency, uint32_t, I_CND_NODEP)
VALU_KERNEL(addc_co_u32, uint32_t, I_ADDC)
VALU_KERNEL(cmp_e64_sgpr_plus_cndmask_e64, uint32_t, I_CMPCND64)
VALU_KERNEL(cndmask_vcc_plus_3_add_u32, uint32_t, I_CNDADD)
VALU_KERNEL(pk_mul_lo_u16, uint32_t, I_PKMULLO)
VALU_KERNEL(pk_mad_u16, uint32_t, I_PKMAD)
VALU_KERNEL(cvt_f32_ubyte0, uint32_t, I_CVTUB)
VALU_KERNEL(sad_u8, uint32_t, I_SADU8)
VALU_KERNEL(mov_b32, uint32_t, I_MOV)
VALU_KERNEL(readlane_b32, uint32_t, I_READLANE)
VALU_KERNEL(cvt_pk_bf16_f32, float, I_CVTBF16)
VALU_KERNEL(mul_f32, float, I_MUL32F)
VALU_KERNEL(add_f32, float, I_ADD32F)
VALU_KERNEL(rsq_f32, float, I_RSQ)
VALU_KERNEL(rcp_f32, float, I_RCP)
VALU_KERNEL(sqrt_f32, float, I_SQRT)
VALU_KERNEL(pk_mul_f32, double, I_PKMULF)
VALU_KERNEL(mul_f64, double, I_MULF64)
VALU_KERNEL(add_f64, double, I_ADDF64)
VALU_KERNEL(fma_f32, float, I_FMA32)
VALU_KERNEL(pk_fma_f16, uint32_t, I_PKFMA16)
VALU_KERNEL(pk_fma_f32, double, I_PKFMA32)          // 64-bit register pairs: two f32 lanes per instruction
VALU_KERNEL(fma_f64, double, I_FMA64)

template <class T> using ValuFn = void (*)(uint64_t*, int, T, T, T);

struct ValuRow { std::string name; int lanes_per_inst; std::vector<double> tlaneops; std::vector<double> cyc; double ghz; };

template <class T>
static ValuRow run_valu(const char* name, ValuFn<T> fn, T seed, T b, T c, uint64_t* d_out, int ops_per_lane) {
    const int cfg_waves[5] = {1, 2, 4, 8, 32};          // resident waves per SIMD (last: 8 resident x 4 rounds)
    ValuRow row; row.name = name; row.lanes_per_inst = ops_per_lane; row.ghz = 0;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int ci = 0; ci < 5; ci++) {
        int k = cfg_waves[ci];
        int grid = 256 * k;                                  // 256-thread blocks: one wave per SIMD each, k blocks per CU
        int iters = 20000 / (k > 8 ? 8 : k) / 4 * 4 + 64;
        double best = 1e30; uint64_t h[2] = {0, 0};
        for (int r = 0; r < REPS + 1; r++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, d_out, iters, seed, b, c);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) { best = ms; CHECK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost)); }
        }
        double insts = (double)grid * 4 * iters * BLOCKS_PER_ITER * INSTS_PER_BLOCK;        // wave-instructions
        double rate = insts * 64 / (best * 1e-3);                                            // lane-instructions per second
        double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0;                          // s_memtime ticks per 100 MHz tick
        row.tlaneops.push_back(rate / 1e12);
        // cycles one SIMD needs per wave-instruction = clock x SIMDs / (wave-instructions per second)
        row.cyc.push_back(ghz * 1e9 * 1024 / (insts / (best * 1e-3)));
        if (k == 8) row.ghz = ghz;
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return row;
}

// ------------------------------------------------------------------------------------------------------------------------------
// matrix cores: 4 independent accumulators per wave, operands loop invariant
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4;
typedef __attribute__((__vector_size__(8 * sizeof(int)))) int i32x8;
typedef __attribute__((__vector_size__(4 * sizeof(double)))) double f64x4;

__global__ __launch_bounds__(256) void k_mfma_bf16(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 1.2345f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_mfma_i8(int* out, int iters) {
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, 7};
    i32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 4; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345) out[0] = s;
}

__global__ __launch_bounds__(256) void k_mfma_fp4(float* out, int iters) {
    i32x8 a = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, b = {4, 5, 6, 7, 0, 0, 0, 0};
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {        // cbsz = blgp = 4: both operands FP4 (E2M1), unit block scales (E8M0 127)
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 127, 0, 127);
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 1.2345f) out[0] = s;
}

__global__ __launch_bounds__(256) void k_mfma_f64(double* out, int iters) {
    double a = threadIdx.x * 0.5, b = 1.25;
    f64x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 1.2345) out[0] = s;
}

// dependent accumulate chains: NACC accumulators, RUN back-to-back MFMAs on one accumulator before moving to the next (what a GEMM
// inner loop that finishes one output tile's K step before the next tile's looks like)
template <int NACC, int RUN>
__global__ __launch_bounds__(256) void k_mfma_chain(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    f32x16 c[NACC];
    for (int k = 0; k < NACC; k++) for (int i = 0; i < 16; i++) c[k][i] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < NACC; k++)
#pragma unroll
            for (int r = 0; r < RUN; r++) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k], 0, 0, 0);
    }
    float s = 0;
    for (int k = 0; k < NACC; k++) for (int i = 0; i < 16; i++) s += c[k][i];
    if (s == 1.2345f) out[0] = s;
}

template <class F>
static double time_launch(F launch) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 1e30;
    for (int r = 0; r < REPS + 1; r++) {
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return best;
}

// ------------------------------------------------------------------------------------------------------------------------------
// HBM
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * st < n; i += 4 * st) {
        float4 x0 = a[i], x1 = a[i + st], x2 = a[i + 2 * st], x3 = a[i + 3 * st];
        b[i] = x0; b[i + st] = x1; b[i + 2 * st] = x2; b[i + 3 * st] = x3;
    }
    for (; i < n; i += st) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float s = 0;
    for (; i + 3 * st < n; i += 4 * st) {
        float4 x0 = a[i], x1 = a[i + st], x2 = a[i + 2 * st], x3 = a[i + 3 * st];
        s += x0.x + x0.y + x0.z + x0.w + x1.x + x1.y + x1.z + x1.w + x2.x + x2.y + x2.z + x2.w + x3.x + x3.y + x3.z + x3.w;
    }
    for (; i < n; i += st) s += a[i].x;
    if (s == 1.2345f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ b, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float4 x = {v, v, v, v};
    for (; i < n; i += st) b[i] = x;
}
// block-contiguous forms: every block owns 4 x 256 consecutive float4 (16 KB), all four loads of a lane in flight, no loop
__global__ __launch_bounds__(256) void k_copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 < n) {
        float4 x0 = a[i], x1 = a[i + 256], x2 = a[i + 512], x3 = a[i + 768];
        b[i] = x0; b[i + 256] = x1; b[i + 512] = x2; b[i + 768] = x3;
    }
}
__global__ __launch_bounds__(256) void k_read4(const float4* __restrict__ a, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (i + 768 < n) {
        float4 x0 = a[i], x1 = a[i + 256], x2 = a[i + 512], x3 = a[i + 768];
        float s = x0.x + x0.y + x0.z + x0.w + x1.x + x1.y + x1.z + x1.w + x2.x + x2.y + x2.z + x2.w + x3.x + x3.y + x3.z + x3.w;
        if (s == 1.2345f) out[0] = s;
    }
}
__global__ __launch_bounds__(256) void k_write4(float4* __restrict__ b, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    float4 x = {v, v, v, v};
    if (i + 768 < n) { b[i] = x; b[i + 256] = x; b[i + 512] = x; b[i + 768] = x; }
}
// a byte-granular copy as the extractor's ingest does it (16 bytes per lane, u8 images): the achievable rate of THIS path's access width
__global__ __launch_bounds__(256) void k_copy_u8x16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

int main(int argc, char** argv) {
    int dev = 0; CHECK(hipSetDevice(dev));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, dev));
    uint64_t* d_out; CHECK(hipMalloc(&d_out, 4096)); CHECK(hipMemset(d_out, 0, 4096));

    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d, \"mem_clock_mhz_reported\": %d, \"mem_bus_bits\": %d,\n",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, p.memoryClockRate / 1000, p.memoryBusWidth);

    // ---- HBM ----
    {
        size_t bytes = (size_t)2 << 30, n = bytes / 16;
        float4 *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
        CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
        int grid = 256 * 16;
        double t_copy = time_launch([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
        double t_read = time_launch([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, (float*)d_out, n); });
        double t_write = time_launch([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n, 1.0f); });
        double t_c16 = time_launch([&] { hipLaunchKernelGGL(k_copy_u8x16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n); });
        unsigned g4 = (unsigned)(n / 1024);
        double t_copy4 = time_launch([&] { hipLaunchKernelGGL(k_copy4, dim3(g4), dim3(256), 0, 0, a, b, n); });
        double t_read4 = time_launch([&] { hipLaunchKernelGGL(k_read4, dim3(g4), dim3(256), 0, 0, a, (float*)d_out, n); });
        double t_write4 = time_launch([&] { hipLaunchKernelGGL(k_write4, dim3(g4), dim3(256), 0, 0, b, n, 1.0f); });
        if (t_copy4 < t_copy) t_copy = t_copy4;
        if (t_c16 < t_copy) t_copy = t_c16;
        if (t_read4 < t_read) t_read = t_read4;
        if (t_write4 < t_write) t_write = t_write4;
        double t_d2d = time_launch([&] { CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
        printf(" \"hbm\": {\"buffer_bytes\": %zu, \"copy_GBps\": %.1f, \"read_GBps\": %.1f, \"write_GBps\": %.1f, "
               "\"hipMemcpyDtoD_GBps\": %.1f, \"spec_GBps\": 8000.0, \"note\": \"best of three kernel shapes each (grid-stride, 16 KB per block, 4 KB per block); copy counts read + write bytes\"},\n",
               bytes, 2.0 * bytes / t_copy / 1e6, bytes / t_read / 1e6, bytes / t_write / 1e6, 2.0 * bytes / t_d2d / 1e6);
        // ---- PCIe ----
        size_t pb = (size_t)512 << 20; void* h; CHECK(hipHostMalloc(&h, pb, hipHostMallocDefault)); memset(h, 3, pb);
        double t_h2d = time_launch([&] { CHECK(hipMemcpyAsync(a, h, pb, hipMemcpyHostToDevice, 0)); });
        double t_d2h = time_launch([&] { CHECK(hipMemcpyAsync(h, a, pb, hipMemcpyDeviceToHost, 0)); });
        // the streamed-input line moves one batch (512 pairs = 478 MB) per step in image-sized pieces: 466 616-byte copies back to back
        size_t img = 376 * 1241; int nimg = 1024;
        double t_h2d_img = time_launch([&] { for (int i = 0; i < nimg; i++) CHECK(hipMemcpyAsync((char*)a + i * img, (char*)h + i * img, img, hipMemcpyHostToDevice, 0)); });
        printf(" \"pcie\": {\"pinned_bytes\": %zu, \"h2d_GBps\": %.2f, \"d2h_GBps\": %.2f, \"h2d_image_sized_copies_GBps\": %.2f, \"spec_GBps\": 63.0},\n",
               pb, pb / t_h2d / 1e6, pb / t_d2h / 1e6, (double)img * nimg / t_h2d_img / 1e6);
        CHECK(hipHostFree(h)); CHECK(hipFree(a)); CHECK(hipFree(b));
    }

    // ---- VALU ----
    std::vector<ValuRow> rows;
#define RUNU(NAME, ops) rows.push_back(run_valu<uint32_t>(#NAME, k_##NAME, 0x01020304u, 0x00400040u, 0x07060302u, d_out, ops))
    RUNU(pk_max_i16, 2); RUNU(pk_min_i16, 2); RUNU(pk_maximum3_f16, 2); RUNU(pk_minimum3_f16, 2); RUNU(pk_add_u16, 2); RUNU(pk_sub_u16_clamp, 2);
    RUNU(perm_b32, 1); RUNU(alignbyte_b32, 1); RUNU(dot4_u32_u8, 1); RUNU(dot2_u32_u16, 1); RUNU(add_u32, 1); RUNU(max3_u32, 1);
    RUNU(and_or_b32, 1); RUNU(lshl_or_b32, 1); RUNU(mul_lo_u32, 1); RUNU(mul_hi_u32_u24, 1); RUNU(mad_u32_u24, 1); RUNU(bcnt_u32_b32, 1);
    RUNU(cndmask_b32, 1); RUNU(sub_u16_clamp, 1); RUNU(pk_fma_f16, 2);
    RUNU(max_u16, 1); RUNU(min_u16, 1); RUNU(max_u32, 1); RUNU(and_b32, 1); RUNU(or_b32, 1); RUNU(xor_b32, 1); RUNU(lshlrev_b32, 1); RUNU(lshrrev_b32, 1);
    RUNU(sub_u32, 1); RUNU(bfe_u32, 1); RUNU(add3_u32, 1); RUNU(or3_b32, 1); RUNU(lshl_add_u32, 1); RUNU(med3_u32, 1); RUNU(max3_u16, 1); RUNU(min3_f16, 1);
    RUNU(max_u16_sdwa, 1); RUNU(mov_b32_dpp, 1); RUNU(add_u32_dpp, 1); RUNU(cmp_gt_u32_plus_cndmask, 1); RUNU(cmp_gt_u32, 1); RUNU(cndmask_b32_vcc_set, 1);
    RUNU(cndmask_b32_e64_sgpr_mask, 1); RUNU(cndmask_b32_no_dependency, 1); RUNU(addc_co_u32, 1);
    RUNU(cmp_e64_sgpr_plus_cndmask_e64, 1); RUNU(cndmask_vcc_plus_3_add_u32, 1);
    RUNU(pk_mul_lo_u16, 2); RUNU(pk_mad_u16, 2); RUNU(cvt_f32_ubyte0, 1); RUNU(sad_u8, 1); RUNU(mov_b32, 1); RUNU(readlane_b32, 1);
#define RUNF(NAME) rows.push_back(run_valu<float>(#NAME, k_##NAME, 1.0f, 0.999f, 0.001f, d_out, 1))
#define RUND(NAME, ops) rows.push_back(run_valu<double>(#NAME, k_##NAME, 1.0, 0.999, 0.001, d_out, ops))
    RUNF(cvt_pk_bf16_f32); RUNF(mul_f32); RUNF(add_f32); RUNF(rsq_f32); RUNF(rcp_f32); RUNF(sqrt_f32); RUND(pk_mul_f32, 2); RUND(mul_f64, 1); RUND(add_f64, 1);
    rows.push_back(run_valu<float>("fma_f32", k_fma_f32, 1.0f, 0.999f, 0.001f, d_out, 1));
    rows.push_back(run_valu<double>("pk_fma_f32", k_pk_fma_f32, 1.0, 0.5, 0.25, d_out, 2));
    rows.push_back(run_valu<double>("fma_f64", k_fma_f64, 1.0, 0.999, 0.001, d_out, 1));
    printf(" \"valu\": {\"unit\": \"T lane-instructions/s (wave-instructions x 64); cycles_per_wave_inst_per_simd at the measured shader clock\", "
           "\"waves_per_simd\": [1, 2, 4, 8, \"8 x 4 rounds\"], \"spec_16_lanes_per_cycle\": 39.3, \"spec_32_lanes_per_cycle\": 78.6, \"instructions\": {\n");
    for (size_t i = 0; i < rows.size(); i++) {
        const ValuRow& r = rows[i];
        printf("   \"v_%s\": {\"tlaneops\": [%.2f, %.2f, %.2f, %.2f, %.2f], \"cycles_per_wave_inst_per_simd\": [%.2f, %.2f, %.2f, %.2f, %.2f], \"elements_per_lane\": %d, \"shader_ghz\": %.3f}%s\n",
               r.name.c_str(), r.tlaneops[0], r.tlaneops[1], r.tlaneops[2], r.tlaneops[3], r.tlaneops[4], r.cyc[0], r.cyc[1], r.cyc[2], r.cyc[3], r.cyc[4],
               r.lanes_per_inst, r.ghz, i + 1 < rows.size() ? "," : "");
    }
    printf(" }},\n");

    // ---- MFMA ----
    {
        int iters = 4000;
        struct { const char* name; double flop; double ms; int w; } m[16]; int nm = 0;
        for (int w : {1, 2, 4}) {       // waves per SIMD
            int grid = 256 * w;
            double t;
            t = time_launch([&] { hipLaunchKernelGGL(k_mfma_bf16, dim3(grid), dim3(256), 0, 0, (float*)d_out, iters); });
            m[nm++] = {"mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, t, w};
            t = time_launch([&] { hipLaunchKernelGGL(k_mfma_i8, dim3(grid), dim3(256), 0, 0, (int*)d_out, iters); });
            m[nm++] = {"mfma_i32_16x16x64_i8", 2.0 * 16 * 16 * 64, t, w};
            t = time_launch([&] { hipLaunchKernelGGL(k_mfma_fp4, dim3(grid), dim3(256), 0, 0, (float*)d_out, iters); });
            m[nm++] = {"mfma_scale_f32_32x32x64_f8f6f4(fp4)", 2.0 * 32 * 32 * 64, t, w};
            t = time_launch([&] { hipLaunchKernelGGL(k_mfma_f64, dim3(grid), dim3(256), 0, 0, (double*)d_out, iters); });
            m[nm++] = {"mfma_f64_16x16x4_f64", 2.0 * 16 * 16 * 4, t, w};
        }
        // dependent chains (bf16 32x32x16): accumulators per wave x back-to-back run length x waves per SIMD
        std::string chain = " \"mfma_bf16_dependent_chains\": {\"unit\": \"TFLOP/s\", \"note\": \"NACC accumulators per wave, RUN back-to-back MFMAs on one accumulator before the next\"";
#define CHAIN(NACC, RUN) for (int w : {1, 2, 4}) { \
            double t = time_launch([&] { hipLaunchKernelGGL((k_mfma_chain<NACC, RUN>), dim3(256 * w), dim3(256), 0, 0, (float*)d_out, iters / RUN); }); \
            char buf[160]; snprintf(buf, sizeof buf, ", \"nacc%d_run%d@%dwave_per_simd\": %.1f", NACC, RUN, w, 2.0 * 32 * 32 * 16 * NACC * RUN * (iters / RUN) * (256.0 * w * 4) / (t * 1e-3) / 1e12); chain += buf; }
        CHAIN(1, 1) CHAIN(2, 1) CHAIN(2, 6) CHAIN(4, 1) CHAIN(4, 6) CHAIN(8, 1)
        printf("%s},\n", chain.c_str());
        printf(" \"mfma\": {\"unit\": \"T(FL)OP/s dense\", \"spec\": {\"bf16\": 2500.0, \"i8\": 5000.0, \"fp4\": 10000.0, \"f64\": 78.6}, \"instructions\": {\n");
        for (int i = 0; i < nm; i++) {
            double total = m[i].flop * 4.0 * iters * (256.0 * m[i].w * 4);
            printf("   \"%s@%dwave_per_simd\": %.1f%s\n", m[i].name, m[i].w, total / (m[i].ms * 1e-3) / 1e12, i + 1 < nm ? "," : "");
        }
        printf(" }}\n");
    }
    printf("}\n");
    return 0;
}
