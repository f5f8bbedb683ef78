// MFMA 16x16x4 traits shared by the GEMM-shaped kernels (gfx950).
#pragma once
#include "common.hpp"

namespace ssa {

constexpr int kGemmThreads = 256;
constexpr int BM = 128, BN = 128, KC = 16;

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
    using acc_t = f64x4;
    using vec_t = double2;
    static constexpr int VEC = 2;
    static constexpr int APAD = 2;
    static constexpr int BPAD = 16;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct Mfma<float> {
    using acc_t = f32x4;
    using vec_t = float4;
    static constexpr int VEC = 4;
    static constexpr int APAD = 4;
    static constexpr int BPAD = 16;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D map of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
    static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// XCD-contiguous (bijective for any grid size) id -> linear work id
__device__ __forceinline__ int64_t xcd_contiguous(int64_t pid, int64_t nwg) {
    const int64_t q = nwg / 8, r = nwg % 8;
    const int64_t xcd = pid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pid / 8;
}

// LDS-DMA: 16 bytes per lane, LDS destination = wave-uniform base + lane * 16
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        reinterpret_cast<const __attribute__((address_space(1))) void *>(
            reinterpret_cast<uintptr_t>(gsrc)),
        (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

}  // namespace ssa
