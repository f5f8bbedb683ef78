// Optional instrumentation shared by the MFMA GEMM kernels: between ssa_profile_begin() and
// ssa_profile_end() every launch of the f64 full-tile kernels is bracketed by HIP events on its
// own stream (what bench.py's `roofline` object is computed from).  ssa_profile_begin_kinds() restricts the
// bracketing to some kinds: a timing event pair costs a few microseconds on its stream, which the ~ 1300 short
// products of the panel chains per config-H factorization feel (bench step 148 -> 138 ms without them) and the
// ~ 100 trailing updates do not.
#pragma once
#include <vector>

#include "common.hpp"

namespace ssa {

enum ProfileKind : int {
    kProfileGemmNN = 0,    // gemm_kernel<double, true>            (LU trailing / in-panel updates)
    kProfileSyrkLower = 1, // gemm_op_kernel<double, N, T, lower>  (Cholesky trailing update)
    kProfileOpNT = 2,      // gemm_op_kernel<double, N, T, all tiles> (Cholesky strips and panel products)
    kProfileRound = 3,     // chol_tail_round_kernel<double>          (round launches: diagonal blocks + update tiles)
    kProfileSmallBatch = 4, // gemm_nt_small_batch_kernel<double>     (the rounds' panel and strip products)
    kProfileKinds = 5
};

struct GemmProfile {
    bool enabled = false;
    unsigned kinds = ~0u;   // bit k: launches of kind k are bracketed (ssa_profile_begin_kinds)
    std::vector<hipEvent_t> start, stop;
    std::vector<double> flops;
    std::vector<int> kind;
    size_t used = 0;
};
inline GemmProfile g_prof;

struct ProfileScope {
    bool active;
    hipStream_t st;
    ProfileScope(bool wanted, int kind, double flops, hipStream_t s) : active(false), st(s) {
        if (!g_prof.enabled || !wanted || ((g_prof.kinds >> kind) & 1u) == 0) return;
        if (g_prof.used == g_prof.start.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            g_prof.start.push_back(a);
            g_prof.stop.push_back(b);
            g_prof.flops.push_back(0.0);
            g_prof.kind.push_back(0);
        }
        g_prof.flops[g_prof.used] = flops;
        g_prof.kind[g_prof.used] = kind;
        (void)hipEventRecord(g_prof.start[g_prof.used], st);
        active = true;
    }
    ~ProfileScope() {
        if (active) {
            (void)hipEventRecord(g_prof.stop[g_prof.used], st);
            ++g_prof.used;
        }
    }
};

// ssa_shutdown(): the events of the instrumentation
inline int profile_shutdown() {
    int rc = SSA_OK;
    g_prof.enabled = false;
    for (size_t i = 0; i < g_prof.start.size(); ++i)
        if (hipEventDestroy(g_prof.start[i]) != hipSuccess || hipEventDestroy(g_prof.stop[i]) != hipSuccess)
            rc = SSA_ERR_HIP;
    g_prof.start.clear();
    g_prof.stop.clear();
    g_prof.flops.clear();
    g_prof.kind.clear();
    g_prof.used = 0;
    return rc;
}

}  // namespace ssa
