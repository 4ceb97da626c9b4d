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

} // namespace cov
} // namespace crsdr
