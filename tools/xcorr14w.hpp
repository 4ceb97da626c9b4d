// (experiment, not part of libcrsdr.so: it built against the csrc/ headers of round 2 as K1 variant "w" -- one line in kernels.hpp,
// one launch branch in crsdr.hip -- passed test_k1_variants_agree (lags identical to the packed kernel, mag to 1e-5, packets +-1 LSB)
// and measured 3.41 ms per 64-block launch against 2.96 (packed) and 2.73 (two-row), A/B in one call: DESIGN.md, dead end (11))
// xcorr14w.hpp -- K1 for B = 16384 with 1024 threads per row: 16 points per thread, FOUR waves per SIMD (experiment,
// CRSDR_K1_VARIANT=w).
//
// The packed kernel (xcorr14p.hpp) keeps 32 points per thread in ~190 VGPRs: two waves per SIMD, and a row's LDS phases and
// VALU phases add up instead of overlapping; the two-row kernel (xcorr14q.hpp) overlaps them across two rows but leaves the
// image-owning window to ONE wave per SIMD.  Here a row gets 16 waves: the same 32 x 32 x 16 network, the same LDS image and
// the same four exchanges, but every radix-32 transform is shared by the two lanes l and l + 32 of a wave, 16 points each
// (<= 128 VGPRs).  A 32-point DFT over two lanes needs exactly ONE exchange of half the data, and gfx950 has the instruction
// for it: v_permlane32_swap_b32 (lanes 32-63 of one register <-> lanes 0-31 of another).
//   forward (decimation in frequency): lane h loads inputs i in {8h .. 8h+7} and {16+8h ..}: the radix-2 butterflies (i, i+16)
//     are local; u = a + b, v = (a - b) W32^i; the swap hands all u to lane 0 and all v to lane 1; one DFT16 per lane:
//     lane h ends with X[2m + h].
//   inverse (decimation in time): lane h loads inputs k = 2m + h (what the forward pass left): one IDFT16 per lane (E on lane 0,
//     O on lane 1), the swap, then x[i] = E[i] + conj(W32^i) O[i], x[i+16] = E[i] - ... for i = 8h .. 8h+7.
// Inter-pass twiddles: 16 per lane, the SAME set for a forward pass's outputs and the matching inverse pass's inputs
// (w^((2m+h) n)), from the five table values by 15 products (depth 4).  Per DFT32 that is 2 x 145 packed instructions
// against 223 + 62 in the packed kernel: the same arithmetic volume.  LDS access patterns per wave-instruction are those of
// the packed kernel (16 / 32 consecutive columns of one plane or two), so its conflict-free image carries over unchanged;
// P1, J and P1' of planes 2w and 2w + 1 belong to wave w, so only P0 -> P1 and P1' -> P0' are workgroup barriers.
// Same results as the packed kernel up to rounding (different butterfly order): lags identical, mag to ~1e-6.
#pragma once
#include "xcorr14p.hpp"

namespace crsdr {
namespace x14w {

using x14::N; using x14::L; using x14::LDS_ELEMS; using x14::LDS_BYTES; using x14::TWA_STRIDE; using x14::TWB_STRIDE;
using x14::kInvScale2; using x14::p0_base; using x14::j_base; using x14::wave_lds_sync;
constexpr int WT = 1024;

// lanes 32-63 of a <-> lanes 0-31 of b (both components of the pair)
__device__ __forceinline__ void swap32(c2 &a, c2 &b)
{
    const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = mk(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    b = mk(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}

// lane-specific W32^(8h + i'), i' < 8 (forward sign), selected once per kernel
__device__ __forceinline__ void make_kc(c2 *kc, int h)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // cos / sin (pi k / 16): x14::kCos16[k], sin(pi k/16) = cos(pi (8 - k) / 16)
        const float c0 = x14::kCos16[i], s0 = x14::kCos16[8 - i];             // k = i
        const float c1 = x14::kCos16[8 + i], s1 = x14::kCos16[i];             // k = 8 + i: cos(pi(8+i)/16), sin(pi(8+i)/16) = cos(pi i / 16)
        kc[i] = mk(h ? c1 : c0, -(h ? s1 : s0));
    }
}

// forward 32-point DFT over the lane pair: R[i'] = a[8h+i'], R[8+i'] = a[16+8h+i']  ->  R[m] = X[2m+h]
template <bool PRUNED>
__device__ __forceinline__ void dft32_pair_fwd(c2 *R, int h)
{
    c2 kc[8];
    make_kc(kc, h);            // re-selected per use (16 cheap selects): 16 registers less across the junction
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (PRUNED) {                       // a[i+16] == 0 (the zero-padded half of a signal row)
            R[8 + i] = cmul(R[i], kc[i]);
        } else {
            const c2 t = csub(R[i], R[8 + i]);
            R[i] = cadd(R[i], R[8 + i]);
            R[8 + i] = cmul(t, kc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) swap32(R[i], R[8 + i]);
    dft16<-1>(R);
}

// inverse 32-point DFT over the lane pair: R[m] = Y[2m+h]  ->  R[i'] = x[8h+i'], R[8+i'] = x[16+8h+i']
__device__ __forceinline__ void dft32_pair_inv(c2 *R, int h)
{
    c2 kc[8];
    make_kc(kc, h);
    dft16<+1>(R);
#pragma unroll
    for (int i = 0; i < 8; ++i) swap32(R[i], R[8 + i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const c2 t = cmulc(R[8 + i], kc[i]);
        const c2 e = R[i];
        R[i] = cadd(e, t);
        R[8 + i] = csub(e, t);
    }
}

// g[m] = w^(2m + h), m < 16, from the table values w^1, w^2, w^4, w^8, w^16 at tab[j * stride + idx]: 15 products, depth 4
__device__ __forceinline__ void tw_pair(c2 *g, const c2 *__restrict__ tab, int stride, int idx, int h)
{
    const c2 w1 = tab[idx], w2 = tab[stride + idx], w4 = tab[2 * stride + idx], w8 = tab[3 * stride + idx], w16 = tab[4 * stride + idx];
    g[0] = mk(h ? w1.x : 1.0f, h ? w1.y : 0.0f);
    g[1] = cmul(g[0], w2);
    g[2] = cmul(g[0], w4);
    g[3] = cmul(g[1], w4);
#pragma unroll
    for (int m = 0; m < 4; ++m) g[4 + m] = cmul(g[m], w8);
#pragma unroll
    for (int m = 0; m < 8; ++m) g[8 + m] = cmul(g[m], w16);
}

// P1 / P1' element (plane-local) index of radix index i for column c: the packed kernel's p1_off(i) + (c ^ p1_swz(i)) with a
// run-time i
__device__ __forceinline__ int p1_elem(int i, int c) { return ((i ^ ((i >> 3) & 1)) << 4) + (c ^ ((i & 7) << 1)); }

// One row; 1024 threads.
__device__ __forceinline__ void xcorr_row14w(const XcorrArgs &a, unsigned char *smem, const int8_t *__restrict__ src, int row, int t,
                                             const c2 *__restrict__ twA, const c2 *__restrict__ twB)
{
    c2 *A = reinterpret_cast<c2 *>(smem);
    float4 *A4 = reinterpret_cast<float4 *>(smem);
    float *red = reinterpret_cast<float *>(smem + (size_t)LDS_ELEMS * 8);      // 128 floats of scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, j = lane & 31;
    const float4 *__restrict__ refspec4 = reinterpret_cast<const float4 *>(a.refspec) + (size_t)t * (N / 2);

    // ---- P0: column col = 32 wave + j, this lane's 8 samples a = 8h .. 8h+7 (the other 16 of the 32 are the zero pad)
    const int col = 32 * wave + j;
    c2 R[16];
    {
        const uint16_t *s16 = reinterpret_cast<const uint16_t *>(src);
        const uint32_t x16 = a.xor80 & 0xFFFFu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t u = (uint32_t)s16[(8 * h + i) * 512 + col] ^ x16;
            R[i] = mk((float)sext8(u, 0), (float)sext8(u, 1));
        }
        dft32_pair_fwd<true>(R, h);
        c2 gA[16];
        tw_pair(gA, twA, TWA_STRIDE, col, h);
        const int base = p0_base(col) + h * 528;
#pragma unroll
        for (int m = 0; m < 16; ++m) A[base + m * 1056] = cmul(R[m], gA[m]);
    }
    // P1 / P1' twiddles of this lane: planes 2 wave + (j >> 4), column c = j & 15
    const int blk = 2 * wave + (j >> 4), c = j & 15;
    __syncthreads();
    c2 *Ab = A + blk * 528;
    // ---- P1 forward
    {
        c2 gB[16];
        tw_pair(gB, twB, TWB_STRIDE, c, h);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            R[i] = *(const volatile x14p::lds_c2 *)(Ab + p1_elem(8 * h + i, c));
            R[8 + i] = *(const volatile x14p::lds_c2 *)(Ab + p1_elem(16 + 8 * h + i, c));
        }
        dft32_pair_fwd<false>(R, h);
#pragma unroll
        for (int m = 0; m < 16; ++m) Ab[p1_elem(2 * m + h, c)] = cmul(R[m], gB[m]);
    }
    wave_lds_sync();
    // ---- J: DFT16 . conj(ref spectrum) . IDFT16 on group g = tid (planes 2 wave, 2 wave + 1: this wave's own)
    {
        const int g = tid, key = g & 7, base = j_base(g);
        float4 r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = refspec4[q * 1024 + g];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = A4[base + (q ^ key)];
            R[2 * q] = mk(v.x, v.y);
            R[2 * q + 1] = mk(v.z, v.w);
        }
        dft16<-1>(R);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            R[2 * q] = cmul(R[2 * q], mk(r[q].x, r[q].y));
            R[2 * q + 1] = cmul(R[2 * q + 1], mk(r[q].z, r[q].w));
        }
        dft16<+1>(R);
#pragma unroll
        for (int q = 0; q < 8; ++q) A4[base + (q ^ key)] = make_float4(R[2 * q].x, R[2 * q].y, R[2 * q + 1].x, R[2 * q + 1].y);
    }
    wave_lds_sync();
    // ---- P1' inverse (the twiddles are formed again rather than kept across the junction: 128 registers per thread)
    {
        c2 gB[16];
        int c_ = c;
        asm volatile("" : "+v"(c_));          // opaque: otherwise the two tw_pair calls are merged and 32 registers live (spilled) across the junction
        tw_pair(gB, twB, TWB_STRIDE, c_, h);
#pragma unroll
        for (int m = 0; m < 16; ++m) R[m] = cmulc(*(const volatile x14p::lds_c2 *)(Ab + p1_elem(2 * m + h, c)), gB[m]);
    }
    dft32_pair_inv(R, h);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        Ab[p1_elem(8 * h + i, c)] = R[i];
        Ab[p1_elem(16 + 8 * h + i, c)] = R[8 + i];
    }
    __syncthreads();
    // ---- P0' inverse + |.|^2: outputs a = 8h + i (m[i]) and 16 + 8h + i (m[8 + i]); natural index a * 512 + col
    float m[16];
    {
        c2 gA[16];
        int col_ = col;
        asm volatile("" : "+v"(col_));        // opaque, as above
        tw_pair(gA, twA, TWA_STRIDE, col_, h);
        const int base = p0_base(col) + h * 528;
#pragma unroll
        for (int q = 0; q < 16; ++q) R[q] = cmulc(A[base + q * 1056], gA[q]);
        dft32_pair_inv(R, h);
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = fmaf(R[i].x, R[i].x, R[i].y * R[i].y);
    }
    float tm = m[0];
#pragma unroll
    for (int i = 1; i < 15; i += 2) tm = fmaxf(tm, fmaxf(m[i], m[i + 1]));
    tm = fmaxf(tm, m[15]);
    const float wm = x14p::q_wave_max63(tm);
    int *redi = reinterpret_cast<int *>(red);
    if (tid == 0) redi[32] = 0x7fffffff;
    if (lane == 63) red[wave] = wm;
    __syncthreads();
    float gm = red[0];
#pragma unroll
    for (int wv = 1; wv < WT / 64; ++wv) gm = fmaxf(gm, red[wv]);
    // first index of the maximum (volk_32f_index_max_32u: first strict maximum)
    if (tm == gm) {
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            const int aidx = (i < 8) ? (8 * h + i) : (16 + 8 * h + (i - 8));
            const int n = aidx * 512 + col;
            bi = (m[i] == gm && n < bi) ? n : bi;
        }
        atomicMin(&redi[32], bi);
    }
    __syncthreads();
    int gi = redi[32];
    if ((unsigned)gi >= (unsigned)N) gi = 0;
    // parabolic neighbours gi -+ 1: same output a of the columns col -+ 1 -- through LDS (one barrier; 16 waves make the
    // in-wave shortcut of the packed kernel less often applicable: a wave holds only 32 columns)
    const int nl = gi - 1, nr = gi + 1;
    auto mine_at = [&](int n) -> float {
        const int aidx = n >> 9;                           // 0 .. 31
        const int hh = (aidx >> 3) & 1, ii = (aidx & 7) + ((aidx >> 4) << 3);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v = (i == ii) ? m[i] : v;
        return hh == h ? v : 0.f;
    };
    if (gi > 0 && (nl & 511) == col && (((nl >> 9) >> 3) & 1) == h) red[40] = mine_at(nl);
    if (gi < N - 1 && (nr & 511) == col && (((nr >> 9) >> 3) & 1) == h) red[41] = mine_at(nr);
    __syncthreads();
    if (tid == 0) {
        float D = 0.0f;
        if (gi > 0 && gi < N - 1) {
            const float ym = red[40], yp = red[41];
            const float den = (ym - 2.0f * gm) + yp;
            if (den != 0.0f) D = (0.5f * (ym - yp)) / den;
        }
        xcorr_publish(a, row, t, gi - L /* src/ccoherent.cc:232 */, sqrtf(gm / (float)L) * kInvScale2 /* :204 */, D);
    }
}

__global__ __launch_bounds__(WT) void k_xcorr_lag14w(XcorrArgs a, const float2 *__restrict__ twA, const float2 *__restrict__ twB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int row = a.row_begin + blockIdx.x, t = blockIdx.y;
    if (xcorr_skip(a, row, t, threadIdx.x)) return;
    xcorr_row14w(a, smem, a.rows + (size_t)t * a.block_stride + (size_t)row * N, row, t,
                 reinterpret_cast<const c2 *>(twA), reinterpret_cast<const c2 *>(twB));
}

} // namespace x14w
} // namespace crsdr
