/*
 * gram_oracle.c -- plain C (OpenMP) restatement of the reference's OAK Gram arithmetic, for sizes the
 * NumPy oracle is too slow for and as the multi-core "port" CPU baseline of bench.py.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): never linked into or called by the product.
 *
 * Per (row, column) pair it performs, in the reference's order:
 *   k_d  = variance*exp(-0.5*(|x/l|^2 + |z/l|^2 - 2 (x/l)(z/l))) - c(x) c(z) / var_s   oak/ortho_rbf_kernel.py:157-172
 *          (gpflow square_distance form) or a table gather                             ortho_binary_kernel.py:40-53
 *   s_p  = sum_d k_d^p, p = 0..R                                                       oak/oak_kernel.py:236-239
 *   e_n  = (1/n) sum_{k=1..n} (-1)^(k-1) e_{n-k} s_k                                    oak/oak_kernel.py:240-248
 *   K    = sum_n sigma2_n e_n                                                           oak/oak_kernel.py:256-260
 * The reference materialises each of these as a full matrix; element-wise the arithmetic is identical.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC gram_oracle.c -o _build/libgram_oracle.so -lm   (oracle/build.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 64
#define MAXR 16

/* X1s/X2s: per-dim scaled coordinate x/l (rbf) or category index (discrete), dimension-major [D][n].
 * C1/C2  : per-dim cov_X_s values c(x) (0 for unconstrained / discrete), dimension-major.
 * inv_v  : 1/var_s per dim (0 if unconstrained); var: base variance per dim;
 * type   : 0 rbf, 1 discrete (table lookup, tables[tab_off[d] + i*ncat[d] + j], variance already applied)
 * w      : R+1 order weights.  out: [n1][n2] row-major. */
void oak_oracle_gram(int D, int R, const int* type, const double* var, const double* inv_v, const int* ncat,
                     const int* tab_off, const double* tables, const double* w, const double* X1s,
                     const double* C1, int64_t n1, int64_t ld1, const double* X2s, const double* C2, int64_t n2,
                     int64_t ld2, double* out, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n1; ++i) {
        double xs1[MAXD], x1sq[MAXD], c1[MAXD];
        for (int d = 0; d < D; ++d) {
            xs1[d] = X1s[(int64_t)d * ld1 + i];
            x1sq[d] = xs1[d] * xs1[d];
            c1[d] = C1[(int64_t)d * ld1 + i];
        }
        double* orow = out + i * n2;
        for (int64_t j = 0; j < n2; ++j) {
            double s[MAXR + 1], e[MAXR + 1];
            for (int p = 0; p <= R; ++p) s[p] = 0.0;
            for (int d = 0; d < D; ++d) {
                double k;
                if (type[d] == 0) {
                    const double z = X2s[(int64_t)d * ld2 + j];
                    const double r2 = -2.0 * xs1[d] * z + (x1sq[d] + z * z);
                    k = var[d] * exp(-0.5 * r2) - (c1[d] * C2[(int64_t)d * ld2 + j]) * inv_v[d];
                } else {
                    k = tables[tab_off[d] + (int)xs1[d] * ncat[d] + (int)X2s[(int64_t)d * ld2 + j]];
                }
                double kp = 1.0;                      /* tf.pow(k, p) for integer p */
                for (int p = 0; p <= R; ++p) { s[p] += kp; kp *= k; }
            }
            e[0] = 1.0;
            for (int n = 1; n <= R; ++n) {
                double acc = 0.0;
                for (int k = 1; k <= n; ++k) acc += ((k - 1) % 2 == 0 ? 1.0 : -1.0) * e[n - k] * s[k];
                e[n] = (1.0 / n) * acc;
            }
            double K = 0.0;
            for (int n = 0; n <= R; ++n) K += w[n] * e[n];
            orow[j] = K;
        }
    }
}

void oak_oracle_gram_diag(int D, int R, const int* type, const double* var, const double* inv_v, const int* ncat,
                          const int* tab_off, const double* tables, const double* w, const double* X1s,
                          const double* C1, int64_t n1, int64_t ld1, double* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n1; ++i) {
        double s[MAXR + 1], e[MAXR + 1];
        for (int p = 0; p <= R; ++p) s[p] = 0.0;
        for (int d = 0; d < D; ++d) {
            double k;
            if (type[d] == 0) {
                const double c = C1[(int64_t)d * ld1 + i];
                k = var[d] - (c * c) * inv_v[d];        /* oak/ortho_rbf_kernel.py:174-177 */
            } else {
                k = tables[tab_off[d] + ncat[d] * ncat[d] + (int)X1s[(int64_t)d * ld1 + i]];
            }
            double kp = 1.0;
            for (int p = 0; p <= R; ++p) { s[p] += kp; kp *= k; }
        }
        e[0] = 1.0;
        for (int n = 1; n <= R; ++n) {
            double acc = 0.0;
            for (int k = 1; k <= n; ++k) acc += ((k - 1) % 2 == 0 ? 1.0 : -1.0) * e[n - k] * s[k];
            e[n] = (1.0 / n) * acc;
        }
        double K = 0.0;
        for (int n = 0; n <= R; ++n) K += w[n] * e[n];
        out[i] = K;
    }
}

int oak_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
