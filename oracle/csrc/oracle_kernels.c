/* CPU ORACLE / BASELINE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the two numba kernels on the hot path, parallelised over rows with
 * OpenMP the way numba's prange does (njit(fastmath=True, parallel=True)):
 *   oracle_q_matrix            <- distance.q_matrix                 distance.py:87-115
 *   oracle_biot_savart         <- biot_savart_film_to_film          solver/solve.py:28-73
 * Built by oracle/build_oracle.py with  gcc -O3 -ffast-math -fopenmp  (fastmath mirrors the
 * reference's numba flag, so like the reference it is reproducible only to a few ulp).
 * Used by tests (cross-check of the numpy oracle) and by bench.py's cpu_baseline leg.
 */
#include <math.h>
#include <stdint.h>

#define ONE_OVER_4PI 0.07957747154594767

void oracle_q_matrix(const double *points, int64_t n, double *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double xi = points[2 * i], yi = points[2 * i + 1];
        double *row = out + i * n;
        for (int64_t j = 0; j < n; ++j) {
            if (i == j) {
                row[j] = 0.0;
            } else {
                const double dx = xi - points[2 * j], dy = yi - points[2 * j + 1];
                row[j] = ONE_OVER_4PI * pow(dx * dx + dy * dy, -1.5);
            }
        }
    }
}

void oracle_biot_savart(const double *film1_sites, double film1_z0, const double *film1_areas,
                        const double *film1_J, int64_t n1, const double *film2_sites,
                        double film2_z0, int64_t n2, double *out) {
    const double dz2 = (film2_z0 - film1_z0) * (film2_z0 - film1_z0);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n2; ++i) {
        const double xi = film2_sites[2 * i], yi = film2_sites[2 * i + 1];
        double tmp = 0.0;
        for (int64_t j = 0; j < n1; ++j) {
            const double dx = xi - film1_sites[2 * j], dy = yi - film1_sites[2 * j + 1];
            tmp += ONE_OVER_4PI * film1_areas[j] * (film1_J[2 * j] * dy - film1_J[2 * j + 1] * dx) *
                   pow(dx * dx + dy * dy + dz2, -1.5);
        }
        out[i] = tmp;
    }
}
