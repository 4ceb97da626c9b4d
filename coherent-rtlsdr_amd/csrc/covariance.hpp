// covariance.hpp -- SURVEY 8(f4): the first downstream consumer of the aligned matrix, and the one step of
// the chain that IS a dense contraction: the beamformer's  Rxx = (1/L) X^H X  over the signal channels with
// the per-channel mean removed (beamformclient/heatmap2d2.cpp:189-199; X(n, c) = sample n of channel c,
// int8 scaled by 1/127 :313).  On MI355X that is an int8 GEMM on the matrix cores:
//     sum_n conj(x_a[n]) x_b[n] = (I_a.I_b + Q_a.Q_b) + j (I_a.Q_b - Q_a.I_b)
// -> three v_mfma_i32_32x32x32_i8 products per K-step straight from the packet's interleaved IQ bytes
//    (a . b,  swap16(a) . even_bytes(b),  swap16(a) . odd_bytes(b)), exact in int32 (|sum| <= 2^15 L),
// -> mean removal and the 1/(127^2 L) scale in fp64 in the epilogue.
// Both MFMA operands are rows of the same row-major matrix, so no transposes and no LDS: each lane
// loads the 16 contiguous bytes of "its" row and K-slice.  Hermitian: only tiles on / above the diagonal are formed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crsdr {
namespace cov {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// per-row integer sums (sum I, sum Q) for the mean removal
__global__ __launch_bounds__(256) void k_row_sums(const int8_t *__restrict__ m, int B, int2 *__restrict__ sums)
{
    __shared__ int sre[4], sim[4];
    const uint32_t *r32 = reinterpret_cast<const uint32_t *>(m + (size_t)blockIdx.x * B);
    int re = 0, im = 0;
    for (int i = threadIdx.x; i < B / 4; i += 256) {
        const uint32_t w = r32[i];
        re = __builtin_amdgcn_sdot4((int)w, 0x00010001, re, false); // bytes 0,2 = I
        im = __builtin_amdgcn_sdot4((int)w, 0x01000100, im, false); // bytes 1,3 = Q
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { re += __shfl_xor(re, off, 64); im += __shfl_xor(im, off, 64); }
    if ((threadIdx.x & 63) == 0) { sre[threadIdx.x >> 6] = re; sim[threadIdx.x >> 6] = im; }
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = make_int2(sre[0] + sre[1] + sre[2] + sre[3], sim[0] + sim[1] + sim[2] + sim[3]);
}

__device__ __forceinline__ int swap16(int w) { return (int)((((uint32_t)w >> 8) & 0x00FF00FFu) | (((uint32_t)w << 8) & 0xFF00FF00u)); }

// grid (ceil(nsig/64), ceil(nsig/64)); 256 threads = 2 x 2 waves, one 32 x 32 output tile per wave.
// matrix: [nrows][B] int8 (row 0 = ref, skipped like the reference's X.rightCols(cols-1)); rxx: [nsig][nsig] cf32.
__global__ __launch_bounds__(256) void k_covariance(const int8_t *__restrict__ matrix, int nrows, int B, const int2 *__restrict__ sums,
                                                    float2 *__restrict__ rxx)
{
    const int nsig = nrows - 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a0 = blockIdx.y * 64 + (wave >> 1) * 32, b0 = blockIdx.x * 64 + (wave & 1) * 32;
    if (a0 >= nsig || b0 >= nsig) return; // wave-uniform
    // Rxx is Hermitian: only the 32 x 32 tiles on or above the diagonal are formed, the rest is their conjugate transpose
    if (b0 < a0) return;
    const int ra = min(a0 + (lane & 31), nsig - 1), rb = min(b0 + (lane & 31), nsig - 1); // clamp: padded rows are masked at the store
    const v4i *pa = reinterpret_cast<const v4i *>(matrix + (size_t)(1 + ra) * B) + (lane >> 5);
    const v4i *pb = reinterpret_cast<const v4i *>(matrix + (size_t)(1 + rb) * B) + (lane >> 5);
    v16i g1 = {}, g2 = {}, g3 = {};
    const int steps = B / 32; // 32 bytes of K per MFMA: 16 per lane half
    for (int s = 0; s < steps; ++s) {
        const v4i a = pa[2 * s], b = pb[2 * s];
        const v4i asw = {swap16(a.x), swap16(a.y), swap16(a.z), swap16(a.w)};
        const v4i be = {b.x & 0x00FF00FF, b.y & 0x00FF00FF, b.z & 0x00FF00FF, b.w & 0x00FF00FF};
        const v4i bo = {b.x & (int)0xFF00FF00, b.y & (int)0xFF00FF00, b.z & (int)0xFF00FF00, b.w & (int)0xFF00FF00};
        g1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, g1, 0, 0, 0);     // I_a I_b + Q_a Q_b
        g2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(asw, be, g2, 0, 0, 0);  // Q_a I_b
        g3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(asw, bo, g3, 0, 0, 0);  // I_a Q_b
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int col = b0 + (lane & 31);
    const double L = (double)(B / 2), scale = 1.0 / (127.0 * 127.0);
    if (col < nsig) {
        const int2 sb = sums[1 + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = a0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < nsig) {
                const int2 sa = sums[1 + row];
                // (1/L) sum conj(x_a) x_b - conj(mean_a) mean_b,   x = (I + jQ)/127
                const double re = ((double)g1[r] / L - ((double)sa.x * sb.x + (double)sa.y * sb.y) / (L * L)) * scale;
                const double im = ((double)(g3[r] - g2[r]) / L - ((double)sa.x * sb.y - (double)sa.y * sb.x) / (L * L)) * scale;
                rxx[(size_t)row * nsig + col] = make_float2((float)re, (float)im);
                if (b0 > a0) rxx[(size_t)col * nsig + row] = make_float2((float)re, -(float)im);   // mirrored tile
            }
        }
    }
}

// ---- LDS-tiled form (r03): 128 x 128 output tiles, K split over the grid ----------------------------------------------------------
// k_covariance above streams both operands of every 32 x 32 tile from L2 (no reuse: 26 TB/s of L2 reads at the matrix cores' rate --
// it runs at 13 % of it).  Here a 256-thread workgroup forms one 128 x 128 tile on or above the diagonal: four waves, 2 x 2 MFMA tiles
// each, operands staged through LDS in 128-byte K chunks (row pitch 144 B: conflict-free 16-byte reads in 8-lane groups), the next
// chunk's global loads in flight under the current chunk's 48 MFMAs per wave.  36 tiles do not fill 256 CUs, so the K range is split
// over the grid (S = 7 for 1024 channels: 252 workgroups, one per CU) and every workgroup leaves its exact int32 partial sums in a
// workspace; k_cov_reduce adds the S partials and applies the mean removal / scaling of the reference's expression in fp64, writing
// both triangles.  Integer sums: order-independent, the result is bit for bit that of k_covariance.
constexpr int CT = 128, KC = 128, CPITCH = 144;            // tile edge, K bytes per chunk, LDS row pitch
constexpr int COV_LDS_BYTES = 2 * 2 * CT * CPITCH;          // two operands, double-buffered: 73 728 B

__device__ __forceinline__ void cov_tile_of(int u, int nt, int &ti, int &tj)
{
    ti = 0;
    while (u >= nt - ti) { u -= nt - ti; ++ti; }
    tj = ti + u;
}

// grid (tiles on / above the diagonal, S); partial: [S][tiles][3][CT][CT] int32
__global__ __launch_bounds__(256, 1) void k_covariance_tiled(const int8_t *__restrict__ matrix, int nrows, int B, int nt, int *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nsig = nrows - 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 1, wx = wave & 1;
    int ti, tj;
    cov_tile_of((int)blockIdx.x, nt, ti, tj);
    const int a0 = ti * CT, b0 = tj * CT;
    const bool diag = ti == tj;
    const int nchunks = B / KC, S = (int)gridDim.y, z = (int)blockIdx.y;
    const int c_lo = (int)(((long long)nchunks * z) / S), c_hi = (int)(((long long)nchunks * (z + 1)) / S);
    unsigned char *As = smem, *Bs = diag ? smem : smem + 2 * CT * CPITCH;      // [buf][row][pitch]; a diagonal tile's operands are the same rows
    // global -> LDS: thread i moves 16 bytes of rows i / 8 + 32 it, it = 0 .. 3 (8 threads = one 128-byte line of a row)
    const int lr = tid >> 3, lc = (tid & 7) * 16;
    // (row offsets as integers and every loop spelled out: with pointer arrays captured by lambdas the arrays went to scratch memory)
    size_t oa0 = (size_t)(1 + min(a0 + lr, nsig - 1)) * B + lc, oa1 = (size_t)(1 + min(a0 + lr + 32, nsig - 1)) * B + lc,
           oa2 = (size_t)(1 + min(a0 + lr + 64, nsig - 1)) * B + lc, oa3 = (size_t)(1 + min(a0 + lr + 96, nsig - 1)) * B + lc;      // clamp: padded rows are masked by the reducer
    size_t ob0 = (size_t)(1 + min(b0 + lr, nsig - 1)) * B + lc, ob1 = (size_t)(1 + min(b0 + lr + 32, nsig - 1)) * B + lc,
           ob2 = (size_t)(1 + min(b0 + lr + 64, nsig - 1)) * B + lc, ob3 = (size_t)(1 + min(b0 + lr + 96, nsig - 1)) * B + lc;
    v16i g1[2][2] = {}, g2[2][2] = {}, g3[2][2] = {};
    uint4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;
    pb0 = pb1 = pb2 = pb3 = make_uint4(0, 0, 0, 0);
#define COV_GLOAD(c)                                                                                         \
    do {                                                                                                     \
        const size_t ko = (size_t)(c) * KC;                                                                  \
        pa0 = *reinterpret_cast<const uint4 *>(matrix + oa0 + ko);                                           \
        pa1 = *reinterpret_cast<const uint4 *>(matrix + oa1 + ko);                                           \
        pa2 = *reinterpret_cast<const uint4 *>(matrix + oa2 + ko);                                           \
        pa3 = *reinterpret_cast<const uint4 *>(matrix + oa3 + ko);                                           \
        if (!diag) {                                                                                         \
            pb0 = *reinterpret_cast<const uint4 *>(matrix + ob0 + ko);                                       \
            pb1 = *reinterpret_cast<const uint4 *>(matrix + ob1 + ko);                                       \
            pb2 = *reinterpret_cast<const uint4 *>(matrix + ob2 + ko);                                       \
            pb3 = *reinterpret_cast<const uint4 *>(matrix + ob3 + ko);                                       \
        }                                                                                                    \
    } while (0)
#define COV_LSTORE(buf)                                                                                      \
    do {                                                                                                     \
        unsigned char *la = As + (size_t)(buf) * CT * CPITCH + (size_t)lr * CPITCH + lc;                     \
        *reinterpret_cast<uint4 *>(la) = pa0;                                                                \
        *reinterpret_cast<uint4 *>(la + 32 * CPITCH) = pa1;                                                  \
        *reinterpret_cast<uint4 *>(la + 64 * CPITCH) = pa2;                                                  \
        *reinterpret_cast<uint4 *>(la + 96 * CPITCH) = pa3;                                                  \
        if (!diag) {                                                                                         \
            unsigned char *lb = Bs + (size_t)(buf) * CT * CPITCH + (size_t)lr * CPITCH + lc;                 \
            *reinterpret_cast<uint4 *>(lb) = pb0;                                                            \
            *reinterpret_cast<uint4 *>(lb + 32 * CPITCH) = pb1;                                              \
            *reinterpret_cast<uint4 *>(lb + 64 * CPITCH) = pb2;                                              \
            *reinterpret_cast<uint4 *>(lb + 96 * CPITCH) = pb3;                                              \
        }                                                                                                    \
    } while (0)
    if (c_lo < c_hi) {
        COV_GLOAD(c_lo);
        COV_LSTORE(0);
    }
    __syncthreads();
    const int frow = (lane & 31) * CPITCH + 16 * (lane >> 5);
    for (int c = c_lo; c < c_hi; ++c) {
        const int buf = (c - c_lo) & 1;
        if (c + 1 < c_hi) COV_GLOAD(c + 1);                   // in flight under this chunk's MFMAs
        const unsigned char *Ab = As + (size_t)buf * CT * CPITCH + (size_t)(wy * 64) * CPITCH + frow;
        const unsigned char *Bb = Bs + (size_t)buf * CT * CPITCH + (size_t)(wx * 64) * CPITCH + frow;
#pragma unroll
        for (int ks = 0; ks < KC / 32; ++ks) {
            v4i a[2], b[2], asw[2], be[2], bo[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                a[m] = *reinterpret_cast<const v4i *>(Ab + (size_t)(m * 32) * CPITCH + ks * 32);
                b[m] = *reinterpret_cast<const v4i *>(Bb + (size_t)(m * 32) * CPITCH + ks * 32);
                asw[m] = v4i{swap16(a[m].x), swap16(a[m].y), swap16(a[m].z), swap16(a[m].w)};
                be[m] = v4i{b[m].x & 0x00FF00FF, b[m].y & 0x00FF00FF, b[m].z & 0x00FF00FF, b[m].w & 0x00FF00FF};
                bo[m] = v4i{b[m].x & (int)0xFF00FF00, b[m].y & (int)0xFF00FF00, b[m].z & (int)0xFF00FF00, b[m].w & (int)0xFF00FF00};
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    g1[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], g1[m][n], 0, 0, 0);      // I_a I_b + Q_a Q_b
                    g2[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(asw[m], be[n], g2[m][n], 0, 0, 0);   // Q_a I_b
                    g3[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(asw[m], bo[n], g3[m][n], 0, 0, 0);   // I_a Q_b
                }
        }
        if (c + 1 < c_hi) COV_LSTORE(buf ^ 1);
        __syncthreads();
    }
#undef COV_GLOAD
#undef COV_LSTORE
    // partial sums out: C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    int *pt = partial + ((size_t)z * gridDim.x + blockIdx.x) * 3 * CT * CT;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wy * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = wx * 64 + n * 32 + (lane & 31);
                pt[(size_t)row * CT + col] = g1[m][n][r];
                pt[(size_t)(CT + row) * CT + col] = g2[m][n][r];
                pt[(size_t)(2 * CT + row) * CT + col] = g3[m][n][r];
            }
}

// grid (tiles, CT rows); 128 threads = the columns of one tile row.  Adds the S partials, applies the epilogue of k_covariance, writes both triangles.
__global__ __launch_bounds__(CT) void k_cov_reduce(const int *__restrict__ partial, int S, int ntiles, int nt, int nrows, int B, const int2 *__restrict__ sums,
                                                   float2 *__restrict__ rxx)
{
    const int nsig = nrows - 1;
    int ti, tj;
    cov_tile_of((int)blockIdx.x, nt, ti, tj);
    const int row = ti * CT + (int)blockIdx.y, col = tj * CT + (int)threadIdx.x;
    long long s1 = 0, s2 = 0, s3 = 0;
    for (int z = 0; z < S; ++z) {
        const int *pt = partial + ((size_t)z * ntiles + blockIdx.x) * 3 * CT * CT + (size_t)blockIdx.y * CT + threadIdx.x;
        s1 += pt[0];
        s2 += pt[(size_t)CT * CT];
        s3 += pt[(size_t)2 * CT * CT];
    }
    if (row >= nsig || col >= nsig) return;
    if (ti == tj && col < row) return;                        // a diagonal tile's lower half is written as the mirror of its upper half
    const double L = (double)(B / 2), scale = 1.0 / (127.0 * 127.0);
    const int2 sa = sums[1 + row], sb = sums[1 + col];
    // (1/L) sum conj(x_a) x_b - conj(mean_a) mean_b,   x = (I + jQ)/127      (beamformclient/heatmap2d2.cpp:189-199)
    const double re = ((double)s1 / L - ((double)sa.x * sb.x + (double)sa.y * sb.y) / (L * L)) * scale;
    const double im = ((double)(s3 - s2) / L - ((double)sa.x * sb.y - (double)sa.y * sb.x) / (L * L)) * scale;
    rxx[(size_t)row * nsig + col] = make_float2((float)re, (float)im);
    if (col != row) rxx[(size_t)col * nsig + row] = make_float2((float)re, -(float)im);
}

} // namespace cov
} // namespace crsdr
