// Shared helpers for the gfx950 kernels of libsuperscreen_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <initializer_list>
#include <utility>

#include "../../include/superscreen_hip.h"

namespace ssa {

constexpr int kWave = 64;  // CDNA wavefront width

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define SSA_RETURN_IF_LAUNCH_FAILED()                 \
    do {                                              \
        if (hipGetLastError() != hipSuccess) {        \
            return SSA_ERR_HIP;                       \
        }                                             \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) (the opt-in above 64 KiB of dynamic LDS) applies to the
// CURRENT device only: one flag per call site and device, so that a process that drives several GPUs
// raises the limit on each of them.
constexpr int kMaxDevices = 64;
struct DeviceFlags {
    bool set[kMaxDevices] = {};
};
inline int current_device(int *dev) {
    if (hipGetDevice(dev) != hipSuccess || *dev < 0 || *dev >= kMaxDevices) return SSA_ERR_HIP;
    return SSA_OK;
}
inline int raise_dynamic_lds(DeviceFlags &flags, std::initializer_list<std::pair<const void *, size_t>> kernels) {
    int dev = 0;
    if (current_device(&dev) != SSA_OK) return SSA_ERR_HIP;
    if (flags.set[dev]) return SSA_OK;
    for (const auto &k : kernels)
        if (hipFuncSetAttribute(k.first, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(k.second)) !=
            hipSuccess)
            return SSA_ERR_HIP;
    flags.set[dev] = true;
    return SSA_OK;
}

// Compute units of the current device (cached per device; 256 on an MI355X), 0 if the query fails.
inline int device_cu_count() {
    static int cus[kMaxDevices] = {};
    int dev = 0;
    if (current_device(&dev) != SSA_OK) return 0;
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cus[dev] = v;
    }
    return cus[dev];
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Carves 256-byte aligned sub-buffers out of a caller-provided workspace.
struct Carver {
    char *base;
    size_t used;
    explicit Carver(void *p) : base(static_cast<char *>(p)), used(0) {}
    template <typename T>
    T *take(size_t count) {
        used = align_up(used, 256);
        T *p = reinterpret_cast<T *>(base + used);
        used += count * sizeof(T);
        return p;
    }
};

// 1/sqrt(x) in full double precision: hardware v_rsq_f64 seed + one cubically convergent
// correction (seed error ~2^-26 or better -> ~2^-78 before rounding).
__device__ __forceinline__ double rsqrt_f64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-(x * y), y, 1.0);
    double p = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y * e, p, y);
}

constexpr double kOneOver4Pi = 0.07957747154594767;  // 1 / (4 pi)

// r2^(-3/2) in full double precision: hardware v_rsq_f64 seed y0 (relative error e0 <= 2^-26) and the series
// (1 - e)^(-3/2) = 1 + e (3/2 + 15/8 e) + O(e^3), e = 1 - r2 y0^2: 6 FP64 operations behind the seed (a corrected
// 1/sqrt followed by its cube takes 7), with the cube of the SEED off the dependent chain.  Measured issue costs
// (tools/probes/valu_probe.hip, 4 waves per SIMD): v_fma/mul/add_f64 2.3 ns per wave-instruction, v_rsq_f64 7.3 ns;
// an f32 seed (v_cvt_f32_f64 2.05 + v_rsq_f32 3.6 + v_cvt_f64_f32 2.06 ns) costs the same 7.7 ns and needs the same
// series, so it buys nothing.
__device__ __forceinline__ double inv_r3(double r2) {
    const double y0 = __builtin_amdgcn_rsq(r2);
    const double y2 = y0 * y0;
    const double e = __builtin_fma(-r2, y2, 1.0);
    const double y3 = y2 * y0;
    const double p = __builtin_fma(1.875, e, 1.5);
    return __builtin_fma(y3 * e, p, y3);
}

// q = (1/4pi) * r2^(-3/2)
__device__ __forceinline__ double inv_r3_over_4pi(double r2) {
    const double y0 = __builtin_amdgcn_rsq(r2);
    const double y2 = y0 * y0;
    const double e = __builtin_fma(-r2, y2, 1.0);
    const double k3 = (kOneOver4Pi * y2) * y0;
    const double p = __builtin_fma(1.875, e, 1.5);
    return __builtin_fma(k3 * e, p, k3);
}

// sum_{k < count} p[k * stride], added in index order (the order is part of the result), eight loads in flight at a
// time (a plain loop waits for every load before it issues the next): the fixed-order reductions of the split sums.
template <typename T>
__device__ __forceinline__ T sum_strided(const T *__restrict__ p, int count, int64_t stride) {
    T s = T(0);
    int k = 0;
    for (; k + 8 <= count; k += 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[static_cast<int64_t>(k + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < count; ++k) s += p[static_cast<int64_t>(k) * stride];
    return s;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace ssa
