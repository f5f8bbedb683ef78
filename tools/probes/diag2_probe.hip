// Probe (development aid): the second diagonal-block kernel (csrc/chol_diag2.hpp) against a host Cholesky and
// against the first form (csrc/chol_diag.hpp): correctness of L and W = L^-1, time alone.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../superscreen_amd/csrc -o diag2_probe diag2_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHOLK2_TIMING 1
#include "chol_diag.hpp"
#include "chol_diag2.hpp"

using namespace ssa;

static void host_chol_inv(const std::vector<double> &A, int n, std::vector<double> &L, std::vector<double> &W) {
    L.assign(n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        d = std::sqrt(d);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / d;
        }
    }
    W.assign(n * n, 0.0);
    for (int c = 0; c < n; ++c) {
        W[c * n + c] = 1.0 / L[c * n + c];
        for (int i = c + 1; i < n; ++i) {
            double s = 0.0;
            for (int t = c; t < i; ++t) s += L[i * n + t] * W[t * n + c];
            W[i * n + c] = -s / L[i * n + i];
        }
    }
}

int main() {
    const int n = 256, lda = 384, ldw = 4096;
    std::vector<double> A(n * n), L, Wr;
    srand(7);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            const double v = (rand() / double(RAND_MAX) - 0.5) * 0.02 - 0.3 / (1.0 + (i - j) * (i - j));
            A[i * n + j] = A[j * n + i] = v;
        }
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int j = 0; j < n; ++j) s += (j != i) ? std::fabs(A[i * n + j]) : 0.0;
        A[i * n + i] = 1.05 * s + 0.1;   // strictly diagonally dominant like diag(w) A
    }
    host_chol_inv(A, n, L, Wr);
    // device buffers: lower triangle only (the upper one is poisoned: the kernels must not depend on it)
    std::vector<double> hD(n * lda, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) hD[i * lda + j] = (j <= i) ? A[i * n + j] : ((j / 16 == i / 16) ? 1e300 : A[i * n + j]);
    double *dD0, *dD, *dW, *dS;
    long long *dT; hipMalloc(&dT, 64 * 8);
    int32_t *dinfo;
    hipMalloc(&dD0, hD.size() * 8); hipMalloc(&dD, hD.size() * 8); hipMalloc(&dW, size_t(n) * ldw * 8);
    hipMalloc(&dS, 1 << 20); hipMalloc(&dinfo, 4);
    hipMemcpy(dD0, hD.data(), hD.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&cholk2::chol_diag256_v2_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(cholk2::Smem<double>));
    hipFuncSetAttribute(reinterpret_cast<const void *>(&cholk::chol_diag256_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, sizeof(cholk::Ge64Smem<double>));
    auto check = [&](const char *name) {
        std::vector<double> gD(n * lda), gW(size_t(n) * ldw);
        int32_t info = -9;
        hipMemcpy(gD.data(), dD, gD.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(gW.data(), dW, gW.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
        double eL = 0, eW = 0, mL = 0, mW = 0, upW = 0;
        int wi = -1, wj = -1;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                if (j <= i) {
                    const double dl = std::fabs(gD[i * lda + j] - L[i * n + j]);
                    if (!(dl <= eL)) { eL = dl; wi = i; wj = j; }
                    mL = std::fmax(mL, std::fabs(L[i * n + j]));
                    eW = std::fmax(eW, std::fabs(gW[size_t(i) * ldw + j] - Wr[i * n + j]));
                    mW = std::fmax(mW, std::fabs(Wr[i * n + j]));
                } else {
                    upW = std::fmax(upW, std::fabs(gW[size_t(i) * ldw + j]));
                }
            }
        printf("%s: info %d  max|dL|/max|L| = %.3e (at %d,%d)  max|dW|/max|W| = %.3e  max|W above diagonal| = %.3e\n", name, info,
               eL / mL, wi, wj, eW / mW, upW);
    };
    auto run_v2 = [&]() {
        hipLaunchKernelGGL((cholk2::chol_diag256_v2_kernel<double>), dim3(1), dim3(cholk2::kThreads), sizeof(cholk2::Smem<double>), 0, dD, lda, dW, ldw, dS,
                           dinfo, 1, dT);
    };
    auto run_v1 = [&]() {
        hipLaunchKernelGGL((cholk::chol_diag256_kernel<double>), dim3(1), dim3(256), sizeof(cholk::Ge64Smem<double>), 0, dD, lda, dW, ldw,
                           dS, dinfo, 1);
    };
    for (int which = 0; which < 2; ++which) {
        hipMemset(dW, 0, size_t(n) * ldw * 8);
        hipMemset(dinfo, 0, 4);
        hipMemcpy(dD, dD0, hD.size() * 8, hipMemcpyDeviceToDevice);
        if (which == 0) {
            // the first form reads the upper triangle of its 64 x 64 diagonal blocks: give it a full symmetric block
            std::vector<double> full(n * lda, 0.0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) full[i * lda + j] = A[i * n + j];
            hipMemcpy(dD, full.data(), full.size() * 8, hipMemcpyHostToDevice);
            run_v1();
        } else {
            run_v2();
        }
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("kernel %d failed: %s\n", which, hipGetErrorString(e)); return 1; }
        check(which == 0 ? "first form " : "second form");
    }
    // timing alone
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int which = 0; which < 2; ++which) {
        float best = 1e9f, sum = 0;
        const int reps = 20;
        for (int r = 0; r < reps; ++r) {
            hipMemcpyAsync(dD, dD0, hD.size() * 8, hipMemcpyDeviceToDevice, 0);
            hipEventRecord(a, 0);
            if (which == 0) run_v1(); else run_v2();
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            best = std::fmin(best, ms); sum += ms;
        }
        printf("%s: %.1f us best, %.1f us mean (alone)\n", which == 0 ? "first form " : "second form", best * 1e3, sum / reps * 1e3);
    }
    {
        long long t[64];
        hipMemcpy(t, dT, 64 * 8, hipMemcpyDeviceToHost);
        printf("stamps (us since start, s_memtime at 100 MHz):");
        for (int k = 0; k < 16; ++k) printf(" [%d] %.1f", k, (t[k] - t[0]) * 0.01);
        printf("\nbase durations (us):");
        for (int k = 0; k < 16; ++k) printf(" %.2f", t[16 + k] * 0.01);
        printf("\n");
    }
    return 0;
}
