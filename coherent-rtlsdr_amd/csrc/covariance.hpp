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
    const int8_t *row = m + (size_t)blockIdx.x * B;
    int re = 0, im = 0;
    if (((uintptr_t)row & 15) == 0 && (B & 15) == 0) {         // 16 bytes per lane and load (r03: the dword loop ran at 2 TB/s)
        const uint4 *r128 = reinterpret_cast<const uint4 *>(row);
        for (int i = threadIdx.x; i < B / 16; i += 256) {
            const uint4 w = r128[i];
            re = __builtin_amdgcn_sdot4((int)w.x, 0x00010001, re, false); im = __builtin_amdgcn_sdot4((int)w.x, 0x01000100, im, false);
            re = __builtin_amdgcn_sdot4((int)w.y, 0x00010001, re, false); im = __builtin_amdgcn_sdot4((int)w.y, 0x01000100, im, false);
            re = __builtin_amdgcn_sdot4((int)w.z, 0x00010001, re, false); im = __builtin_amdgcn_sdot4((int)w.z, 0x01000100, im, false);
            re = __builtin_amdgcn_sdot4((int)w.w, 0x00010001, re, false); im = __builtin_amdgcn_sdot4((int)w.w, 0x01000100, im, false);
        }
    } else {
        const uint32_t *r32 = reinterpret_cast<const uint32_t *>(row);
        for (int i = threadIdx.x; i < B / 4; i += 256) {
            const uint32_t w = r32[i];
            re = __builtin_amdgcn_sdot4((int)w, 0x00010001, re, false); // bytes 0,2 = I
            im = __builtin_amdgcn_sdot4((int)w, 0x01000100, im, false); // bytes 1,3 = Q
        }
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
// it runs at 13 % of it).  Here a 512-thread workgroup forms one 128 x 128 tile on or above the diagonal: eight waves (two per SIMD,
// each the other's cover), a 64 x 32 piece each (2 x 1 MFMA tiles: three 16-byte LDS reads per four MFMAs),
// operands staged through LDS in 256-byte K chunks (row pitch 272 B: conflict-free 16-byte reads in 8-lane groups; one barrier per
// 48 MFMAs of a wave: with 128-byte chunks the barriers' bubbles were a quarter of the run), two chunks of global loads in flight
// under the MFMAs.  36 tiles do not fill 256 CUs, so the K range is split over the grid in PAIRS of chunks
// (S = 7 for 1024 channels: 252 workgroups) and every workgroup leaves its exact int32 partial sums in a workspace; k_cov_reduce
// adds the S partials and applies the mean removal / scaling of the reference's expression in fp64, writing both triangles.
// Integer sums: order-independent, the result is bit for bit that of k_covariance.
//
// What bounds it is the memory system, not the matrix cores (12 us of MFMA time for 1024 channels): every workgroup reads its 2 x 128
// rows' K slice once (151 MB in all) and the first cut read them at the Infinity Cache's gather rate (33 GB/s per CU: 18 us) because
// the tiles of one K slice were dealt over all eight XCDs.  So the work items are numbered K-slice-major and handed out so that an
// XCD (workgroups L, L + 8, ... under round-robin placement: speed only) gets a contiguous run of them: one or two K slices of all
// rows, 2.4 MB each, stay in that XCD's 4 MiB L2 and every row is fetched from the Infinity Cache once per XCD that needs it.  The
// partials are two planes (re, and the imaginary part from ONE MFMA per operand pair: see COV_COMPUTE), and the row sums of the
// mean removal come from the diagonal tiles' staging registers instead of their own launch.
constexpr int CT = 128, KC = 256, CPITCH = KC + 16;         // tile edge, K bytes per chunk, LDS row pitch
constexpr int COV_THREADS = 512;
constexpr int COV_LDS_BYTES = 2 * 2 * CT * CPITCH;          // two operands, double-buffered: 139 264 B

// (I, Q) bytes of two samples -> (Q, ~I)
__device__ __forceinline__ int swapnot16(int w) { return swap16(w) ^ (int)0xFF00FF00; }

__device__ __forceinline__ void cov_tile_of(int u, int nt, int &ti, int &tj)
{
    ti = 0;
    while (u >= nt - ti) { u -= nt - ti; ++ti; }
    tj = ti + u;
}

// 16 bytes from a 4-byte aligned address (a packet's matrix starts at 16 + 4 N: the hardware takes a dword-aligned global_load_dwordx4)
struct __attribute__((packed, aligned(4))) cov_u4 { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 cov_load16(const int8_t *p)
{
    const cov_u4 u = *reinterpret_cast<const cov_u4 *>(p);
    return make_uint4(u.x, u.y, u.z, u.w);
}

__device__ __forceinline__ void cov_iq_sums(const uint4 w, int &si, int &sq)
{
    si = __builtin_amdgcn_sdot4((int)w.x, 0x00010001, si, false); sq = __builtin_amdgcn_sdot4((int)w.x, 0x01000100, sq, false);
    si = __builtin_amdgcn_sdot4((int)w.y, 0x00010001, si, false); sq = __builtin_amdgcn_sdot4((int)w.y, 0x01000100, sq, false);
    si = __builtin_amdgcn_sdot4((int)w.z, 0x00010001, si, false); sq = __builtin_amdgcn_sdot4((int)w.z, 0x01000100, sq, false);
    si = __builtin_amdgcn_sdot4((int)w.w, 0x00010001, si, false); sq = __builtin_amdgcn_sdot4((int)w.w, 0x01000100, sq, false);
}

// grid: 8 * ceil(ntri * S / 8) workgroups; partial: [S][ntri][2][CT][CT] int32; psum: [S][nt * CT] int2 (per K slice row sums of I and Q);
// B % (2 KC) == 0
__global__ __launch_bounds__(COV_THREADS, 1) void k_covariance_tiled(const int8_t *__restrict__ matrix, int nrows, int B, int nt, int ntri, int S,
                                                                     int *__restrict__ partial, int2 *__restrict__ psum)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nsig = nrows - 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wy = wave >> 2, wx = wave & 3;
    // work item of this workgroup: the XCD's contiguous run (see above)
    const int per = (int)gridDim.x >> 3, item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (item >= ntri * S) return;                           // workgroup-uniform
    const int z = item / ntri, u = item - z * ntri;
    int ti, tj;
    cov_tile_of(u, nt, ti, tj);
    const int a0 = ti * CT, b0 = tj * CT;
    const bool diag = ti == tj;
    const int npairs = B / (2 * KC);
    const int c_lo = 2 * (int)(((long long)npairs * z) / S), c_hi = 2 * (int)(((long long)npairs * (z + 1)) / S);      // an even number of chunks
    unsigned char *As = smem, *Bs = diag ? smem : smem + 2 * CT * CPITCH;      // [buf][row][pitch]; a diagonal tile's operands are the same rows
    // global -> LDS: thread i moves 16 bytes of rows i / 16 + {0, 32, 64, 96} (16 threads = one 256-byte chunk of a row)
    const int lr = tid >> 4, lc = (tid & 15) * 16;
#define COV_ROW(base, k) ((size_t)(1 + min((base) + lr + 32 * (k), nsig - 1)) * B + lc)      /* clamp: padded rows are masked by the reducer */
    const size_t oa0 = COV_ROW(a0, 0), oa1 = COV_ROW(a0, 1), oa2 = COV_ROW(a0, 2), oa3 = COV_ROW(a0, 3);
    const size_t ob0 = COV_ROW(b0, 0), ob1 = COV_ROW(b0, 1), ob2 = COV_ROW(b0, 2), ob3 = COV_ROW(b0, 3);
#undef COV_ROW
    unsigned char *la = As + (size_t)lr * CPITCH + lc, *lb = Bs + (size_t)lr * CPITCH + lc;
    v16i g1a = {}, g1b = {}, g2a = {}, g2b = {};                               // rows wy * 64 + {0, 32} of the wave's 32 columns
    int si0 = 0, sq0 = 0, si1 = 0, sq1 = 0, si2 = 0, sq2 = 0, si3 = 0, sq3 = 0;      // diagonal tiles: I / Q sums of this thread's bytes of its four rows
    // two register sets of one chunk each (named scalars, statements spelled out: as arrays or struct members selected by a run-time
    // parity they were kept in scratch memory)
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    uint4 p0a0 = zero4, p0a1 = zero4, p0a2 = zero4, p0a3 = zero4, p0b0 = zero4, p0b1 = zero4, p0b2 = zero4, p0b3 = zero4;
    uint4 p1a0 = zero4, p1a1 = zero4, p1a2 = zero4, p1a3 = zero4, p1b0 = zero4, p1b1 = zero4, p1b2 = zero4, p1b3 = zero4;
#define COV_GLOAD(P, c)                                                                                              \
    do {                                                                                                             \
        const size_t ko = (size_t)(c) * KC;                                                                          \
        P##a0 = cov_load16(matrix + oa0 + ko);                                                 \
        P##a1 = cov_load16(matrix + oa1 + ko);                                                 \
        P##a2 = cov_load16(matrix + oa2 + ko);                                                 \
        P##a3 = cov_load16(matrix + oa3 + ko);                                                 \
        if (!diag) {                                                                                                 \
            P##b0 = cov_load16(matrix + ob0 + ko);                                             \
            P##b1 = cov_load16(matrix + ob1 + ko);                                             \
            P##b2 = cov_load16(matrix + ob2 + ko);                                             \
            P##b3 = cov_load16(matrix + ob3 + ko);                                             \
        }                                                                                                            \
    } while (0)
#define COV_LSTORE(P, buf)                                                                                           \
    do {                                                                                                             \
        *reinterpret_cast<uint4 *>(la + (size_t)(buf) * CT * CPITCH) = P##a0;                                        \
        *reinterpret_cast<uint4 *>(la + (size_t)(buf) * CT * CPITCH + 32 * CPITCH) = P##a1;                          \
        *reinterpret_cast<uint4 *>(la + (size_t)(buf) * CT * CPITCH + 64 * CPITCH) = P##a2;                          \
        *reinterpret_cast<uint4 *>(la + (size_t)(buf) * CT * CPITCH + 96 * CPITCH) = P##a3;                          \
        if (!diag) {                                                                                                 \
            *reinterpret_cast<uint4 *>(lb + (size_t)(buf) * CT * CPITCH) = P##b0;                                    \
            *reinterpret_cast<uint4 *>(lb + (size_t)(buf) * CT * CPITCH + 32 * CPITCH) = P##b1;                      \
            *reinterpret_cast<uint4 *>(lb + (size_t)(buf) * CT * CPITCH + 64 * CPITCH) = P##b2;                      \
            *reinterpret_cast<uint4 *>(lb + (size_t)(buf) * CT * CPITCH + 96 * CPITCH) = P##b3;                      \
        } else {                                                                                                     \
            cov_iq_sums(P##a0, si0, sq0);                                                                            \
            cov_iq_sums(P##a1, si1, sq1);                                                                            \
            cov_iq_sums(P##a2, si2, sq2);                                                                            \
            cov_iq_sums(P##a3, si3, sq3);                                                                            \
        }                                                                                                            \
    } while (0)
    const int frow = (lane & 31) * CPITCH + 16 * (lane >> 5);
    // Two MFMAs per operand pair, not three: with A' = (Q_a, ~I_a) per sample (~I = -I - 1, always an int8; -I is not for I = -128)
    //   sum A' . B = sum (Q_a I_b + ~I_a Q_b) = sum (Q_a I_b - I_a Q_b) - sum Q_b,
    // and the K slice's sum of Q_b is the row sum the mean removal needs anyway: the reducer adds it back.
#define COV_COMPUTE(buf)                                                                                             \
    do {                                                                                                             \
        const unsigned char *Ab = As + (size_t)(buf) * CT * CPITCH + (size_t)(wy * 64) * CPITCH + frow;              \
        const unsigned char *Bb = Bs + (size_t)(buf) * CT * CPITCH + (size_t)(wx * 32) * CPITCH + frow;              \
        _Pragma("unroll") for (int ks = 0; ks < KC / 32; ++ks) {                                                     \
            const v4i a0v = *reinterpret_cast<const v4i *>(Ab + ks * 32), a1v = *reinterpret_cast<const v4i *>(Ab + 32 * CPITCH + ks * 32); \
            const v4i bv = *reinterpret_cast<const v4i *>(Bb + ks * 32);                                             \
            const v4i s0 = v4i{swapnot16(a0v.x), swapnot16(a0v.y), swapnot16(a0v.z), swapnot16(a0v.w)};              \
            const v4i s1 = v4i{swapnot16(a1v.x), swapnot16(a1v.y), swapnot16(a1v.z), swapnot16(a1v.w)};              \
            g1a = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0v, bv, g1a, 0, 0, 0);      /* I_a I_b + Q_a Q_b */          \
            g2a = __builtin_amdgcn_mfma_i32_32x32x32_i8(s0, bv, g2a, 0, 0, 0);       /* Q_a I_b - I_a Q_b - Q_b */    \
            g1b = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1v, bv, g1b, 0, 0, 0);                                       \
            g2b = __builtin_amdgcn_mfma_i32_32x32x32_i8(s1, bv, g2b, 0, 0, 0);                                        \
        }                                                                                                            \
    } while (0)
    if (c_lo < c_hi) {
        COV_GLOAD(p0, c_lo);
        COV_GLOAD(p1, c_lo + 1);
        COV_LSTORE(p0, 0);
    }
    __syncthreads();
    // two chunks per trip: at the top chunk c is in LDS[0], set p1 holds chunk c + 1, set p0 is free
    for (int c = c_lo; c < c_hi; c += 2) {
        if (c + 2 < c_hi) COV_GLOAD(p0, c + 2);
        COV_COMPUTE(0);
        COV_LSTORE(p1, 1);
        __syncthreads();
        if (c + 3 < c_hi) COV_GLOAD(p1, c + 3);
        COV_COMPUTE(1);
        if (c + 2 < c_hi) COV_LSTORE(p0, 0);
        __syncthreads();
    }
#undef COV_GLOAD
#undef COV_LSTORE
#undef COV_COMPUTE
    // partial sums out: C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    int *pt = partial + (size_t)item * 2 * CT * CT;
    const int col = wx * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wy * 64 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        pt[(size_t)row * CT + col] = g1a[r];
        pt[(size_t)(row + 32) * CT + col] = g1b[r];
        pt[(size_t)(CT + row) * CT + col] = g2a[r];
        pt[(size_t)(CT + row + 32) * CT + col] = g2b[r];
    }
    if (diag) {
        // the sixteen threads of a row are consecutive lanes
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            si0 += __shfl_xor(si0, off, 64); sq0 += __shfl_xor(sq0, off, 64);
            si1 += __shfl_xor(si1, off, 64); sq1 += __shfl_xor(sq1, off, 64);
            si2 += __shfl_xor(si2, off, 64); sq2 += __shfl_xor(sq2, off, 64);
            si3 += __shfl_xor(si3, off, 64); sq3 += __shfl_xor(sq3, off, 64);
        }
        if ((tid & 15) == 0) {
            int2 *ps = psum + (size_t)z * nt * CT + a0 + lr;
            ps[0] = make_int2(si0, sq0);
            ps[32] = make_int2(si1, sq1);
            ps[64] = make_int2(si2, sq2);
            ps[96] = make_int2(si3, sq3);
        }
    }
}

// grid (tiles, CT / CR strips); 256 threads.  Adds the S partials of a CR x CT strip (16-byte loads, lanes along the columns), applies the
// epilogue of k_covariance, writes the strip and -- through LDS, lanes along the rows -- its conjugate transpose.
constexpr int CR = 16;
__global__ __launch_bounds__(256) void k_cov_reduce(const int *__restrict__ partial, const int2 *__restrict__ psum, int S, int ntri, int nt, int nrows, int B,
                                                    float2 *__restrict__ rxx)
{
    __shared__ double2 srow[CR], scol[CT];                  // I / Q sums of the strip's rows and the tile's columns
    __shared__ float2 tr[CR][CT + 1];
    const int nsig = nrows - 1, tid = threadIdx.x;
    int ti, tj;
    cov_tile_of((int)blockIdx.x, nt, ti, tj);
    const int r0 = (int)blockIdx.y * CR;
    const int c4 = (tid & 31) * 4, rr = tid >> 5;           // 4 columns, rows rr and rr + 8 of the strip
    const int *p0 = partial + (size_t)blockIdx.x * 2 * CT * CT + (size_t)(r0 + rr) * CT + c4;
    const size_t zs = (size_t)ntri * 2 * CT * CT;
    long long s1[CR / 8][4] = {}, s2[CR / 8][4] = {};
    // every load of a round is issued before the first use: a rolled loop over the K slices waited out one round trip per slice
    // (13 us for S = 7, whatever the bytes)
    const int idx = tid < CR ? ti * CT + r0 + tid : tj * CT + min(tid - CR, CT - 1);
    long long pa = 0, pb = 0;
    for (int z0 = 0; z0 < S; z0 += 8) {
        int2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = psum[(size_t)min(z0 + k, S - 1) * nt * CT + idx];
        int4 v1[2][CR / 8][4], v2[2][CR / 8][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int h = 0; h < CR / 8; ++h) {
                    const int *q = p0 + min(z0 + 4 * g + k, S - 1) * zs + (size_t)(8 * h) * CT;
                    if (g == 0 || z0 + 4 < S) {             // uniform
                        v1[g][h][k] = *reinterpret_cast<const int4 *>(q);
                        v2[g][h][k] = *reinterpret_cast<const int4 *>(q + CT * CT);
                    }
                }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (z0 + k < S) { pa += v[k].x; pb += v[k].y; }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (z0 + 4 * g + k < S) {
#pragma unroll
                    for (int h = 0; h < CR / 8; ++h) {
                        s1[h][0] += v1[g][h][k].x; s1[h][1] += v1[g][h][k].y; s1[h][2] += v1[g][h][k].z; s1[h][3] += v1[g][h][k].w;
                        s2[h][0] += v2[g][h][k].x; s2[h][1] += v2[g][h][k].y; s2[h][2] += v2[g][h][k].z; s2[h][3] += v2[g][h][k].w;
                    }
                }
    }
    if (tid < CR) srow[tid] = make_double2((double)pa, (double)pb);
    else if (tid < CR + CT) scol[tid - CR] = make_double2((double)pa, (double)pb);
    __syncthreads();
    const double L = (double)(B / 2), scale = 1.0 / (127.0 * 127.0);
#pragma unroll
    for (int h = 0; h < CR / 8; ++h) {
        const int lrow = rr + 8 * h;
        const int row = ti * CT + r0 + lrow;
        const double2 sa = srow[lrow];
        float2 o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double2 sb = scol[c4 + k];
            // (1/L) sum conj(x_a) x_b - conj(mean_a) mean_b,   x = (I + jQ)/127      (beamformclient/heatmap2d2.cpp:189-199)
            const double re = ((double)s1[h][k] / L - (sa.x * sb.x + sa.y * sb.y) / (L * L)) * scale;
            const long long s_im = -(s2[h][k] + (long long)sb.y);      // I_a Q_b - Q_a I_b: the tiles left sum (Q_a I_b - I_a Q_b - Q_b)
            const double im = ((double)s_im / L - (sa.x * sb.y - sa.y * sb.x) / (L * L)) * scale;
            o[k] = make_float2((float)re, (float)im);
            tr[lrow][c4 + k] = make_float2((float)re, -(float)im);
        }
        if (row < nsig) {
            // a diagonal tile's lower half is written as the mirror of its upper half
            const int col0 = tj * CT + c4;
            float2 *dst = rxx + (size_t)row * nsig + col0;
            if (col0 + 3 < nsig && !(ti == tj && col0 < row) && ((uintptr_t)dst & 15) == 0) {
                reinterpret_cast<float4 *>(dst)[0] = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
                reinterpret_cast<float4 *>(dst)[1] = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (col0 + k < nsig && !(ti == tj && col0 + k < row)) dst[k] = o[k];
            }
        }
    }
    __syncthreads();
    // mirror: element (col, row) of rxx = conj of (row, col); 16 lanes = the strip's rows of one column (128 contiguous bytes)
    const int mr = tid & (CR - 1);
    for (int c = tid / CR; c < CT; c += 256 / CR) {
        const int row = ti * CT + r0 + mr, col = tj * CT + c;
        if (row < nsig && col < nsig && col > row) rxx[(size_t)col * nsig + row] = tr[mr][c];
    }
}

} // namespace cov
} // namespace crsdr
