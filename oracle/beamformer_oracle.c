/*
 * beamformer_oracle.c -- CPU restatement of the first consumer of the receive matrix:
 * covariance, noise subspace and the 2-D MUSIC scan of beamformclient/heatmap2d2.cpp.
 *
 * TEST INFRASTRUCTURE ONLY (see coherent_oracle.h): the checker for crsdr_covariance /
 * crsdr_noisesubspace / crsdr_pmusic2d, never linked or called by the product.
 *
 * PARITY UNPINNED: the reference holds no fixtures for this client, and its decomposition is Eigen's
 * BDCSVD (Eigen 3.3.7 per the build line heatmap2d2.cpp:64-68, absent here).  A singular-vector basis of
 * a degenerate (noise) subspace is not unique, so what is comparable -- and what the tests compare -- is
 * the projector Un Un^H, the singular values, and the pseudo-spectrum, which depend only on the subspace.
 */
#include "coherent_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <complex.h>

/* heatmap2d2.cpp:185-199: X = matrix columns 1.. as (I + jQ)/127 (volk_8i_s32f_convert_32f, :313),
 * X.rowwise() -= X.colwise().mean(), Rxx = (1/rows) X^H X.  fp32 data, sums here in fp64.
 * matrix [nrows][B] int8, rxx [(nrows-1)][(nrows-1)][2] float row-major. */
void orc_covariance(float *rxx, const int8_t *matrix, int nrows, int B)
{
    const int M = nrows - 1, L = B / 2;
    float *x = (float *)malloc(sizeof(float) * (size_t)M * B);
    for (int c = 0; c < M; ++c) {
        const int8_t *row = matrix + (size_t)(c + 1) * B;
        float *xc = x + (size_t)c * B;
        orc_convtofloat(xc, row, B);
        double mr = 0, mi = 0;
        for (int n = 0; n < L; ++n) { mr += xc[2 * n]; mi += xc[2 * n + 1]; }
        const float fr = (float)(mr / L), fi = (float)(mi / L);
        for (int n = 0; n < L; ++n) { xc[2 * n] -= fr; xc[2 * n + 1] -= fi; }
    }
    for (int a = 0; a < M; ++a)
        for (int b = 0; b < M; ++b) {
            const float *xa = x + (size_t)a * B, *xb = x + (size_t)b * B;
            double sr = 0, si = 0;
            for (int n = 0; n < L; ++n) { /* conj(xa) xb */
                sr += (double)xa[2 * n] * xb[2 * n] + (double)xa[2 * n + 1] * xb[2 * n + 1];
                si += (double)xa[2 * n] * xb[2 * n + 1] - (double)xa[2 * n + 1] * xb[2 * n];
            }
            rxx[2 * ((size_t)a * M + b)] = (float)(sr / L);
            rxx[2 * ((size_t)a * M + b) + 1] = (float)(si / L);
        }
    free(x);
}

/* heatmap2d2.cpp:69-79 noisesubspace(Rxx, K): SVD, U sorted by singular value, Un = U.rightCols(M-K).
 * Restated as the classical two-sided cyclic Jacobi eigenvalue iteration for a Hermitian matrix in fp64
 * (for Hermitian A the SVD is A = U |Lambda| (U sgn(Lambda))^H, i.e. U = eigenvectors, s = |lambda|).
 * rxx [M][M][2] float; vec [M][M][2] float row-major, column r <-> sv[r], sv descending.
 * Returns the number of sweeps used, or -1 when not converged. */
int orc_noisesubspace(float *vec, float *sv, const float *rxx, int M)
{
    double complex *A = (double complex *)malloc(sizeof(double complex) * (size_t)M * M);
    double complex *V = (double complex *)calloc((size_t)M * M, sizeof(double complex));
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
            /* use the Hermitian part: (a_ij + conj(a_ji)) / 2 */
            const double re = 0.5 * ((double)rxx[2 * (i * M + j)] + (double)rxx[2 * (j * M + i)]);
            const double im = 0.5 * ((double)rxx[2 * (i * M + j) + 1] - (double)rxx[2 * (j * M + i) + 1]);
            A[i * M + j] = re + I * im;
        }
    for (int i = 0; i < M; ++i) V[i * M + i] = 1.0;
    int sweep, done = 0;
    for (sweep = 0; sweep < 60 && !done; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j) {
                const double m2 = creal(A[i * M + j]) * creal(A[i * M + j]) + cimag(A[i * M + j]) * cimag(A[i * M + j]);
                if (i == j) diag += m2; else off += m2;
            }
        if (off <= 1e-30 * diag) { done = 1; break; }
        for (int p = 0; p < M - 1; ++p)
            for (int q = p + 1; q < M; ++q) {
                const double complex apq = A[p * M + q];
                const double g = cabs(apq);
                if (g == 0.0) continue;
                const double app = creal(A[p * M + p]), aqq = creal(A[q * M + q]);
                const double complex ph = apq / g;                 /* e^{j theta} */
                const double zeta = (aqq - app) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                /* J = [[c, s],[-s conj(ph), c conj(ph)]] acting on columns (p, q): col_p' = c col_p - s conj(ph) col_q,
                 * col_q' = s col_p + c conj(ph) col_q;  A <- J^H A J,  V <- V J */
                for (int k = 0; k < M; ++k) {
                    const double complex x = A[k * M + p], y = A[k * M + q] * conj(ph);
                    A[k * M + p] = c * x - s * y;
                    A[k * M + q] = s * x + c * y;
                }
                for (int k = 0; k < M; ++k) {
                    const double complex x = A[p * M + k], y = A[q * M + k] * ph;
                    A[p * M + k] = c * x - s * y;
                    A[q * M + k] = s * x + c * y;
                }
                for (int k = 0; k < M; ++k) {
                    const double complex x = V[k * M + p], y = V[k * M + q] * conj(ph);
                    V[k * M + p] = c * x - s * y;
                    V[k * M + q] = s * x + c * y;
                }
            }
    }
    /* order by |lambda| descending (ties: lower column first) */
    int *order = (int *)malloc(sizeof(int) * M);
    for (int i = 0; i < M; ++i) {
        int rk = 0;
        const double me = fabs(creal(A[i * M + i]));
        for (int j = 0; j < M; ++j) {
            const double o = fabs(creal(A[j * M + j]));
            rk += (o > me) || (o == me && j < i);
        }
        order[i] = rk;
        if (sv) sv[rk] = (float)me;
    }
    for (int r = 0; r < M; ++r)
        for (int c = 0; c < M; ++c) {
            vec[2 * (r * M + order[c])] = (float)creal(V[r * M + c]);
            vec[2 * (r * M + order[c]) + 1] = (float)cimag(V[r * M + c]);
        }
    free(order); free(A); free(V);
    return done ? sweep : -1;
}

/* heatmap2d2.cpp:103-147: s_vecd2d, pmusic, pmusic2dvec in fp32, expression order as written there.
 * vec [M][M][2] (columns k..M-1 = Un), pm [Cx][Cy]. */
void orc_pmusic2d(float *pm, const float *vec, int M, int k, float d, int Mx, int My, int Cx, int Cy)
{
    const float pi = acosf(-1.0f);                                      /* :60 */
    float complex *a = (float complex *)malloc(sizeof(float complex) * (size_t)M);
    for (int cx = 0; cx < Cx; ++cx)
        for (int cy = 0; cy < Cy; ++cy) {
            const float alpha = (float)cx * pi / (float)Cx, beta = (float)cy * pi / (float)Cy;   /* :141 */
            int rc = 0;
            for (int iy = 0; iy < My; ++iy)
                for (int ix = 0; ix < Mx; ++ix) {                        /* :109-112 */
                    const float px = 2.0f * pi * (float)ix * d * cosf(alpha) * sinf(beta);
                    const float py = 2.0f * pi * (float)iy * d * cosf(beta);
                    const float complex ex = cosf(px) + I * sinf(px), ey = cosf(py) + I * sinf(py);
                    a[rc++] = ex * ey;
                }
            float den = 0.f, a2 = 0.f;
            for (int i = 0; i < M; ++i) a2 += crealf(a[i]) * crealf(a[i]) + cimagf(a[i]) * cimagf(a[i]);
            for (int j = k; j < M; ++j) {                                /* (Un^H a).squaredNorm() :122 */
                float complex y = 0;
                for (int i = 0; i < M; ++i) y += conjf(vec[2 * (i * M + j)] + I * vec[2 * (i * M + j) + 1]) * a[i];
                den += crealf(y) * crealf(y) + cimagf(y) * cimagf(y);
            }
            const float res = a2 / den;                                  /* :123 */
            pm[(size_t)cx * Cy + cy] = res * res;                        /* (res*conj(res)).real() :124 */
        }
    free(a);
}
