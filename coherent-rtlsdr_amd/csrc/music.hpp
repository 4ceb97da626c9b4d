// music.hpp -- the consumer of the covariance: noise subspace + 2-D MUSIC scan.
//
// Reference behaviour (beamformclient/heatmap2d2.cpp):
//   :69-79    noisesubspace(Rxx, K): SVD of the M x M Hermitian covariance, Un = U.rightCols(M - K)
//   :103-115  s_vecd2d(alpha, beta, d, Mx, My): URA steering vector, element (ix, iy) at index iy*Mx + ix,
//             a = exp(2 pi j ix d cos(alpha) sin(beta)) * exp(2 pi j iy d cos(beta))
//   :119-128  pmusic(Un, a) = | |a|^2 / |Un^H a|^2 |^2            (the ratio is squared once more, :123)
//   :137-147  pmusic2dvec: alpha = cx pi / Cx, beta = cy pi / Cy over a Cx x Cy grid (100 x 100, :199)
//
// k_herm_subspace: one workgroup, one-sided (Hestenes) Jacobi in fp64 with the matrix and the accumulated
// rotations resident in LDS.  M/2 disjoint column pairs rotate concurrently (round-robin tournament order),
// 16 lanes per pair.  A V = U S with orthogonal columns; for a Hermitian matrix the columns of V are its
// eigenvectors and the column norms its singular values |lambda|, which is exactly what the reference takes
// from the SVD.  V (not A V / sigma) is published, so a rank-deficient covariance still gives an orthonormal
// noise subspace.
// k_pmusic2d: one thread per grid point, Un and the thread's steering vector staged in LDS, fp32 like the
// reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace music {

constexpr int MAX_M = 64;            // sensors; LDS: 2 x M x M complex fp64 = 128 KiB at M = 64
constexpr int JT = 512;              // threads of the Jacobi workgroup: 32 pairs x 16 lanes
constexpr int LANES = 16;
constexpr int MAX_SWEEPS = 30;

__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum16(double v)
{
    v += shfl_xor_f64(v, 1); v += shfl_xor_f64(v, 2); v += shfl_xor_f64(v, 4); v += shfl_xor_f64(v, 8);
    return v;
}

// rxx [M][M] cf32 row-major (Hermitian).  Outputs: sv [M] singular values, descending;
// vec [M][M] cf32 row-major, column r = the singular vector of sv[r] (so the noise subspace for K sources is
// columns K .. M-1); info[0] = sweeps used, info[1] = 1 if converged.
__global__ __launch_bounds__(JT) void k_herm_subspace(const float2 *__restrict__ rxx, int M, float *__restrict__ sv,
                                                      float2 *__restrict__ vec, int *__restrict__ info)
{
    extern __shared__ double2 jsm[];
    double2 *G = jsm;                       // [col][row], column-major: a pair's lanes walk consecutive rows
    double2 *V = jsm + (size_t)M * M;
    __shared__ int rotated;
    __shared__ double colnorm[MAX_M];
    __shared__ int rank_of[MAX_M];
    const int tid = threadIdx.x;

    for (int e = tid; e < M * M; e += JT) {
        const int r = e / M, c = e - r * M;
        const float2 a = rxx[e];
        G[(size_t)c * M + r] = make_double2((double)a.x, (double)a.y);
        V[(size_t)c * M + r] = make_double2(r == c ? 1.0 : 0.0, 0.0);
    }
    if (tid == 0) rotated = 0;
    __syncthreads();

    const int n = (M + 1) & ~1;             // players of the tournament (a dummy if M is odd)
    const int pair = tid / LANES, lane = tid % LANES;
    int sweeps = 0, converged = 0;
    for (; sweeps < MAX_SWEEPS && !converged; ++sweeps) {
        for (int r = 0; r < n - 1; ++r) {
            // circle method: player n-1 is fixed, the others rotate
            int p = -1, q = -1;
            if (pair < n / 2) {
                if (pair == 0) { p = n - 1; q = r; }
                else { p = (r + pair) % (n - 1); q = (r - pair + (n - 1)) % (n - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                if (q >= M) p = -1;         // the dummy sits out
            }
            if (p >= 0) {
                double2 *gp = G + (size_t)p * M, *gq = G + (size_t)q * M;
                double al = 0, be = 0, gr = 0, gi = 0;
                for (int i = lane; i < M; i += LANES) {
                    const double2 x = gp[i], y = gq[i];
                    al += x.x * x.x + x.y * x.y;
                    be += y.x * y.x + y.y * y.y;
                    gr += x.x * y.x + x.y * y.y;          // conj(x) y
                    gi += x.x * y.y - x.y * y.x;
                }
                al = sum16(al); be = sum16(be); gr = sum16(gr); gi = sum16(gi);
                const double g2 = gr * gr + gi * gi;
                if (g2 > 1e-30 * al * be && g2 > 0.0) {
                    const double ga = sqrt(g2);
                    const double er = gr / ga, ei = gi / ga;          // e^{j theta} of gamma
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    double2 *vp = V + (size_t)p * M, *vq = V + (size_t)q * M;
                    for (int i = lane; i < M; i += LANES) {
                        {   // y~ = y e^{-j theta};  x' = c x - s y~,  y' = s x + c y~
                            const double2 x = gp[i], y = gq[i];
                            const double yr = y.x * er + y.y * ei, yi = y.y * er - y.x * ei;
                            gp[i] = make_double2(c * x.x - s * yr, c * x.y - s * yi);
                            gq[i] = make_double2(s * x.x + c * yr, s * x.y + c * yi);
                        }
                        {
                            const double2 x = vp[i], y = vq[i];
                            const double yr = y.x * er + y.y * ei, yi = y.y * er - y.x * ei;
                            vp[i] = make_double2(c * x.x - s * yr, c * x.y - s * yi);
                            vq[i] = make_double2(s * x.x + c * yr, s * x.y + c * yi);
                        }
                    }
                    if (lane == 0) rotated = 1;
                }
            }
            __syncthreads();
        }
        converged = !rotated;
        __syncthreads();
        if (tid == 0) rotated = 0;
        __syncthreads();
    }

    // singular values = column norms of A V; order them (descending, ties by column index)
    if (tid < M) {
        double s2 = 0;
        for (int i = 0; i < M; ++i) { const double2 x = G[(size_t)tid * M + i]; s2 += x.x * x.x + x.y * x.y; }
        colnorm[tid] = sqrt(s2);
    }
    __syncthreads();
    if (tid < M) {
        int rk = 0;
        const double me = colnorm[tid];
        for (int j = 0; j < M; ++j) rk += (colnorm[j] > me) || (colnorm[j] == me && j < tid);
        rank_of[tid] = rk;
        sv[rk] = (float)me;
    }
    __syncthreads();
    for (int e = tid; e < M * M; e += JT) {
        const int c = e / M, r = e - c * M;
        const double2 x = V[(size_t)c * M + r];
        vec[(size_t)r * M + rank_of[c]] = make_float2((float)x.x, (float)x.y);
    }
    if (tid == 0) { info[0] = sweeps; info[1] = converged; }
}

constexpr int PT = 64;               // grid points per workgroup of the scan

// un [M][ldu] cf32 row-major, noise vectors = columns col0 .. col0 + nn - 1;  pm [Cx][Cy] row-major.
__global__ __launch_bounds__(PT) void k_pmusic2d(const float2 *__restrict__ un, int M, int ldu, int col0, int nn, float d, int Mx,
                                                 int My, int Cx, int Cy, float *__restrict__ pm)
{
    extern __shared__ float2 psm[];
    float2 *U = psm;                         // [M][nn]
    float2 *A = psm + (size_t)M * nn;        // [M][PT]: this thread's steering vector, conflict-free by lane
    const int tid = threadIdx.x;
    for (int e = tid; e < M * nn; e += PT) {
        const int i = e / nn, j = e - i * nn;
        U[e] = un[(size_t)i * ldu + col0 + j];
    }
    const int g = blockIdx.x * PT + tid;
    const bool live = g < Cx * Cy;
    const int cx = live ? g / Cy : 0, cy = live ? g - (g / Cy) * Cy : 0;
    const float pi = 3.14159274101257324f;   // (float) acos(-1), heatmap2d2.cpp:60
    const float alpha = (float)cx * pi / (float)Cx, beta = (float)cy * pi / (float)Cy;
    const float ca = cosf(alpha), sb = sinf(beta), cb = cosf(beta);
    float a2 = 0.f;
    for (int iy = 0, rc = 0; iy < My; ++iy) {
        // exp(2 pi j iy d cos(beta)): real argument evaluated left to right like the reference expression
        const float py = 2.0f * pi * (float)iy * d * cb;
        float sy, cyv;
        sincosf(py, &sy, &cyv);
        for (int ix = 0; ix < Mx; ++ix, ++rc) {
            const float px = 2.0f * pi * (float)ix * d * ca * sb;
            float sx, cxv;
            sincosf(px, &sx, &cxv);
            const float re = cxv * cyv - sx * sy, im = cxv * sy + sx * cyv;
            A[(size_t)rc * PT + tid] = make_float2(re, im);
            a2 += re * re + im * im;
        }
    }
    __syncthreads();
    float den = 0.f;
    for (int j = 0; j < nn; ++j) {
        float yr = 0.f, yi = 0.f;                       // (Un^H a)_j = sum_i conj(U[i][j]) a[i]
        for (int i = 0; i < M; ++i) {
            const float2 u = U[(size_t)i * nn + j], a = A[(size_t)i * PT + tid];
            yr += u.x * a.x + u.y * a.y;
            yi += u.x * a.y - u.y * a.x;
        }
        den += yr * yr + yi * yi;
    }
    if (live) {
        const float res = a2 / den;
        pm[g] = res * res;
    }
}

} // namespace music
