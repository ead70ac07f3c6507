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
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC gram_oracle.c -o _build/libgram_oracle.so -lmvec -lm   (oracle/build.py)
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

/* ---------------------------------------------------------------------------------------------------------------------
 * The many-core CPU BASELINE of bench.py: the row-dependent part of gpflow's SGPR.elbo (oak/utils.py:184-195) for a block
 * of rows, the way a tuned CPU build of the reference would run it:
 *   Kuf chunk   all OpenMP threads, the pair loop vectorised over the inducing points (the exponential through glibc's
 *               vector math library, as TensorFlow's Eigen kernels use a vectorised exp; same op order per element)
 *   A = L^-1 Kuf / sigma, AAT += A A^T, Aerr += A y
 *               BLAS dtrsm / dsyrk / dgemv (function pointers handed in by the caller: SciPy's OpenBLAS), one
 *               single-threaded call per 512-row block, blas_threads blocks at a time with thread-local accumulators
 * Everything N-dependent is a row sum, so the chunking changes nothing mathematically.
 * ------------------------------------------------------------------------------------------------------------------- */
#pragma omp declare simd notinbranch
extern double exp(double);

typedef void (*dtrsm_fn)(char*, char*, char*, char*, int*, int*, double*, double*, int*, double*, int*);
typedef void (*dsyrk_fn)(char*, char*, int*, int*, double*, double*, int*, double*, double*, int*);
typedef void (*dgemv_fn)(char*, int*, int*, double*, double*, int*, double*, int*, double*, double*, int*);

/* rows [r0, r1) of K(X1, X2) into out[(i - r0) * n2 + j]; scratch: (R + 3) * n2 doubles */
static void gram_rows_vec(int D, int R, const int* type, const double* var, const double* inv_v, const int* ncat,
                          const int* tab_off, const double* tables, const double* w, const double* X1s, const double* C1,
                          int64_t ld1, const double* X2s, const double* C2, int64_t n2, int64_t ld2, int64_t r0, int64_t r1,
                          double* out, double* scratch) {
    for (int64_t i = r0; i < r1; ++i) {
        double* sp[MAXR + 1];
        double* kb = scratch + (int64_t)(R + 1) * n2;
        double* pw = kb + n2;
        for (int p = 0; p <= R; ++p) { sp[p] = scratch + (int64_t)p * n2; for (int64_t j = 0; j < n2; ++j) sp[p][j] = 0.0; }
        for (int d = 0; d < D; ++d) {
            const double* z2 = X2s + (int64_t)d * ld2;
            if (type[d] == 0) {
                const double x = X1s[(int64_t)d * ld1 + i], xsq = x * x, c1 = C1[(int64_t)d * ld1 + i], vd = var[d], iv = inv_v[d];
                const double* c2 = C2 + (int64_t)d * ld2;
#pragma omp simd
                for (int64_t j = 0; j < n2; ++j) {
                    const double z = z2[j];
                    const double r2 = -2.0 * x * z + (xsq + z * z);
                    kb[j] = vd * exp(-0.5 * r2) - (c1 * c2[j]) * iv;
                }
            } else {
                const int xi = (int)X1s[(int64_t)d * ld1 + i];
                for (int64_t j = 0; j < n2; ++j) kb[j] = tables[tab_off[d] + xi * ncat[d] + (int)z2[j]];
            }
#pragma omp simd
            for (int64_t j = 0; j < n2; ++j) pw[j] = 1.0;
            for (int p = 0; p <= R; ++p) {                 /* s_p += k^p  (tf.pow for integer p: repeated products) */
                double* s = sp[p];
#pragma omp simd
                for (int64_t j = 0; j < n2; ++j) { s[j] += pw[j]; pw[j] *= kb[j]; }
            }
        }
        double* orow = out + (i - r0) * n2;
        for (int64_t j = 0; j < n2; ++j) {
            double e[MAXR + 1];
            e[0] = 1.0;
            for (int n = 1; n <= R; ++n) {
                double acc = 0.0;
                for (int k = 1; k <= n; ++k) acc += ((k - 1) % 2 == 0 ? 1.0 : -1.0) * e[n - k] * sp[k][j];
                e[n] = (1.0 / n) * acc;
            }
            double K = 0.0;
            for (int n = 0; n <= R; ++n) K += w[n] * e[n];
            orow[j] = K;
        }
    }
}

/* AAT (M x M column-major, lower triangle), Aerr (M) and *kdiag_sum are ACCUMULATED into.  Lf: chol(Kuu + jitter I), column-major
 * lower.  chunk rows of Kuf are live at a time (chunk * M doubles of scratch).  Returns 0, or -1 when out of memory. */
int oak_oracle_sgpr_rows(int D, int R, const int* type, const double* var, const double* inv_v, const int* ncat, const int* tab_off,
                         const double* tables, const double* w, const double* X1s, const double* C1, int64_t n1, int64_t ld1,
                         const double* X2s, const double* C2, int64_t M, int64_t ld2, const double* y, const double* Lf,
                         double inv_sigma, int64_t chunk, int blas_threads, void* p_dtrsm, void* p_dsyrk, void* p_dgemv,
                         double* AAT, double* Aerr, double* kdiag_sum, double* seconds /* [3]: gram, blas, reduce */, int nthreads) {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    if (blas_threads > nthreads && nthreads > 0) blas_threads = nthreads;
    dtrsm_fn dtrsm = (dtrsm_fn)p_dtrsm; dsyrk_fn dsyrk = (dsyrk_fn)p_dsyrk; dgemv_fn dgemv = (dgemv_fn)p_dgemv;
    const int64_t BLK = 512;
    if (chunk < BLK) chunk = BLK;
    if (blas_threads < 1) blas_threads = 1;
    double* buf = (double*)malloc(sizeof(double) * (size_t)chunk * (size_t)M);
    double* loc = (double*)calloc((size_t)blas_threads * (size_t)(M * M + M), sizeof(double));
    if (!buf || !loc) { free(buf); free(loc); return -1; }
    double t_gram = 0.0, t_blas = 0.0, t_red = 0.0, kd = 0.0;
    for (int64_t a0 = 0; a0 < n1; a0 += chunk) {
        const int64_t na = (a0 + chunk <= n1) ? chunk : n1 - a0;
        double t0 = omp_get_wtime();
#pragma omp parallel
        {
            double* scratch = (double*)malloc(sizeof(double) * (size_t)(R + 3) * (size_t)M);
#pragma omp for schedule(dynamic, 16) reduction(+ : kd)
            for (int64_t i = a0; i < a0 + na; ++i) {
                gram_rows_vec(D, R, type, var, inv_v, ncat, tab_off, tables, w, X1s, C1, ld1, X2s, C2, M, ld2, i, i + 1,
                              buf + (i - a0) * M, scratch);
                double kdi;
                oak_oracle_gram_diag(D, R, type, var, inv_v, ncat, tab_off, tables, w, X1s + i, C1 + i, 1, ld1, &kdi);
                kd += kdi;
            }
            free(scratch);
        }
        double t1 = omp_get_wtime();
        const int64_t nblk = (na + BLK - 1) / BLK;
#pragma omp parallel num_threads(blas_threads)
        {
            double* aat = loc + (size_t)omp_get_thread_num() * (size_t)(M * M + M);
            double* aerr = aat + M * M;
            int m = (int)M, one_i = 1;
            double one = 1.0;
#pragma omp for schedule(dynamic, 1)
            for (int64_t b = 0; b < nblk; ++b) {
                const int64_t r0 = b * BLK;
                int nb = (int)((r0 + BLK <= na) ? BLK : na - r0);
                double* Ab = buf + r0 * M;                     /* [nb][M] row-major == Kuf block [M x nb] column-major, ld M */
                dtrsm("L", "L", "N", "N", &m, &nb, &inv_sigma, (double*)Lf, &m, Ab, &m);          /* A = L^-1 Kuf / sigma */
                dsyrk("L", "N", &m, &nb, &one, Ab, &m, &one, aat, &m);                           /* AAT += A A^T (lower) */
                dgemv("N", &m, &nb, &one, Ab, &m, (double*)(y + a0 + r0), &one_i, &one, aerr, &one_i);   /* Aerr += A y */
            }
        }
        double t2 = omp_get_wtime();
        t_gram += t1 - t0; t_blas += t2 - t1;
    }
    double t3 = omp_get_wtime();
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < M * M + M; ++e) {
        double acc = 0.0;
        for (int t = 0; t < blas_threads; ++t) acc += loc[(size_t)t * (size_t)(M * M + M) + e];
        if (e < M * M) AAT[e] += acc; else Aerr[e - M * M] += acc;
    }
    t_red = omp_get_wtime() - t3;
    *kdiag_sum += kd;
    if (seconds) { seconds[0] = t_gram; seconds[1] = t_blas; seconds[2] = t_red; }
    free(buf); free(loc);
    return 0;
}

int oak_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
