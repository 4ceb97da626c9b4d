// kernels.hpp -- gfx950 kernels of the coherent-alignment path (included by crsdr.hip only).
//
//   k_ref_spectrum  K0  int8 ref row -> pad at [L,2L) -> B-point DIF FFT -> conj -> HBM
//                       (crefsdr::convtofloat src/crtlsdr.cc:215-218 + sfft slot 0 of
//                        src/ccoherent.cc:174)
//   k_xcorr_lag     K1  one workgroup per signal row, everything LDS/register resident:
//                       int8 -> pad at [0,L) -> DIF FFT -> x conj(ref spectrum) -> DIT inverse
//                       FFT -> |.|^2 -> workgroup argmax (first maximum) -> lag, mag, frac
//                       (src/crtlsdr.cc:205-207, src/ccoherent.cc:123-142,174-234)
//   k_phase_dot     K2a one workgroup per (row, block): exact integer conjugate dot product of
//                       the (shifted) row against the ref row (src/csdrdevice.cc:62)
//   k_align_quant   K2b one workgroup per (row, block): unit phasor -> EMA chain over the
//                       batch -> rotate (+ integer shift in digital mode) -> x127, saturate,
//                       round-half-even -> int8 row at its packet offset
//                       (src/csdrdevice.cc:63-84, src/cpacketizer.cc:137-172)
// Every kernel takes a batch of T consecutive blocks (grid.y = block index inside the batch):
// the host cost of a submit and the launch gaps are paid once per batch, and a GPU that owns
// only a slab of the rows still gets T x rows workgroups to fill its 256 CUs.
//   k_op_*              single-op kernels behind the per-op C ABI (class cdsp)
#pragma once
#include "arith.hpp"
#include "cpk.hpp"
#include "fft_lds.hpp"
#include "plan_args.hpp"
#include "xcorr14.hpp"
#include "xcorr14p.hpp"
#include "xcorr14q.hpp"
#include "longblock.hpp"
#include "covariance.hpp"
#include "music.hpp"
#include <stdint.h>

namespace crsdr {

// ---- row load: int8 IQ row -> complex fp32 in LDS with the reference's zero-pad placement ----
// signal rows: samples in A[0..L), zeros in A[L..2L)   (crtlsdr::convtofloat  src/crtlsdr.cc:205-207)
// ref row    : zeros in A[0..L), samples in A[L..2L)   (crefsdr::convtofloat  src/crtlsdr.cc:215-218)
// xor80: fuse cdsp::convtosigned (src/cdsp.cc:21-34) for offset-binary input.
template <int LOG2N>
__device__ __forceinline__ void load_row_to_lds(float2 *A, const int8_t *__restrict__ row, bool is_ref,
                                                uint32_t xor80, int tid)
{
    using G = FftGeom<LOG2N>;
    constexpr int N = G::N, L = N / 2;
    const int data_off = is_ref ? L : 0, zero_off = is_ref ? 0 : L;
    // one 32-bit word = two complex samples per lane per step: 256 B coalesced per wave from
    // HBM, one conflict-free 16-byte LDS store per lane
    const uint32_t *src = reinterpret_cast<const uint32_t *>(row);
    float4 *dst = reinterpret_cast<float4 *>(A + data_off);
    for (int c = tid; c < N / 4; c += G::THREADS) {
        const uint32_t w = src[c] ^ xor80;
        dst[c] = make_float4(i8_to_f32(sext8(w, 0)), i8_to_f32(sext8(w, 1)), i8_to_f32(sext8(w, 2)),
                             i8_to_f32(sext8(w, 3)));
    }
    for (int n = tid; n < L; n += G::THREADS) A[zero_off + n] = make_float2(0.f, 0.f);
}

// ---- K0 ---------------------------------------------------------------------------------------
template <int LOG2N>
__global__ __launch_bounds__(FftGeom<LOG2N>::THREADS) void k_ref_spectrum(
    const int8_t *__restrict__ rows, size_t block_stride, const float2 *__restrict__ tw, float2 *__restrict__ refspec_base,
    uint32_t xor80)
{
    using G = FftGeom<LOG2N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *A = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const int8_t *ref_row = rows + (size_t)blockIdx.x * block_stride; // row 0 of block blockIdx.x
    float2 *refspec = refspec_base + (size_t)blockIdx.x * G::N;
    load_row_to_lds<LOG2N>(A, ref_row, true, xor80, tid);
    __syncthreads();
    fft_dif_range<LOG2N, -1, 0, G::NPASS>(A, tw, tid);
    // conj(X_ref) in DIF (digit-reversed) order: K1 multiplies element-wise in the same order
    for (int j = tid; j < G::N; j += G::THREADS) {
        float2 v = A[j];
        refspec[j] = make_float2(v.x, -v.y);
    }
}

// ---- K1 ---------------------------------------------------------------------------------------
struct ArgMax {
    float m;
    int idx;
};
__device__ __forceinline__ void argmax_take(ArgMax &a, float m, int idx)
{
    // volk_32f_index_max_32u generic: first strict maximum -> ties resolve to the lowest index
    if (m > a.m || (m == a.m && idx < a.idx)) { a.m = m; a.idx = idx; }
}

template <int LOG2N>
__global__ __launch_bounds__(FftGeom<LOG2N>::THREADS) void k_xcorr_lag(XcorrArgs a, const float2 *__restrict__ tw)
{
    using G = FftGeom<LOG2N>;
    constexpr int N = G::N, L = N / 2, NP = G::NPASS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *A = reinterpret_cast<float2 *>(smem);
    ArgMax *wred = reinterpret_cast<ArgMax *>(smem + sizeof(float2) * N);
    const int tid = threadIdx.x;
    const int row = a.row_begin + blockIdx.x, t = blockIdx.y;
    if (xcorr_skip(a, row, t, tid)) return; // not requested this batch: lag/mag keep their value
    const float2 *__restrict__ refspec = a.refspec + (size_t)t * N;

    load_row_to_lds<LOG2N>(A, a.rows + (size_t)t * a.block_stride + (size_t)row * N, false, a.xor80, tid);
    __syncthreads();
    // forward passes 0 .. NP-2 (each followed by a barrier)
    fft_dif_range<LOG2N, -1, 0, NP - 1>(A, tw, tid);
    // junction: last forward pass (stride 1, no twiddle) x conj(ref spectrum) -> first inverse
    // pass, all in registers -- the two passes act on the same R contiguous points.
    {
        constexpr int LR = G::log2r(NP - 1), R = 1 << LR, NG = N / R;
        static_assert(G::log2m(NP - 1) == 0, "last pass must have stride 1");
        for (int g = tid; g < NG; g += G::THREADS) {
            float2 v[R];
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = A[g * R + i];
            dft<R, -1>(v);
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = cmul(v[i], refspec[g * R + i]); // sfft[k] * conj(sfft[0])
            dft<R, +1>(v);
#pragma unroll
            for (int i = 0; i < R; ++i) A[g * R + i] = v[i];
        }
        __syncthreads();
    }
    // inverse passes NP-2 .. 1
    fft_dit_range<LOG2N, +1, 1, NP - 1>(A, tw, tid);
    // final inverse pass fused with |.|^2 and the per-thread argmax; |.|^2 is parked in A[].x
    ArgMax best = {-1.0f, 0x7fffffff};
    if constexpr (NP >= 2) {
        constexpr int LR = G::log2r(0), R = 1 << LR, LM = G::log2m(0), NG = N / R;
        for (int g = tid; g < NG; g += G::THREADS) {
            const int n2 = g; // P = 0: single block, n2 = g
            float2 v[R];
#pragma unroll
            for (int i = 0; i < R; ++i) v[i] = A[n2 + (i << LM)];
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = ctw<+1>(v[k], tw[n2 * k]);
            dft<R, +1>(v);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                float m = fmaf(v[i].x, v[i].x, v[i].y * v[i].y);
                A[n2 + (i << LM)].x = m;
                argmax_take(best, m, n2 + (i << LM));
            }
        }
    } else {
        // single-pass transform (N == 16): the junction already produced natural order
        for (int j = tid; j < N; j += G::THREADS) {
            float2 v = A[j];
            float m = fmaf(v.x, v.x, v.y * v.y);
            A[j].x = m;
            argmax_take(best, m, j);
        }
    }
    // wavefront argmax (64 lanes), then across the workgroup's waves through LDS
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float om = __shfl_xor(best.m, off, 64);
        int oi = __shfl_xor(best.idx, off, 64);
        argmax_take(best, om, oi);
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) wred[wave] = best;
    __syncthreads();
    if (tid == 0) {
        constexpr int NW = G::THREADS / 64;
        ArgMax b = wred[0];
        for (int w = 1; w < NW; ++w) argmax_take(b, wred[w].m, wred[w].idx);
        const int idx = ((unsigned)b.idx < (unsigned)N) ? b.idx : 0; // all-NaN row: defined as index 0
        float D = 0.0f;
        if (idx > 0 && idx < N - 1) {
            float ym = A[idx - 1].x, yp = A[idx + 1].x;
            float den = (ym - 2.0f * b.m) + yp;
            if (den != 0.0f) D = (0.5f * (ym - yp)) / den;
        }
        xcorr_publish(a, row, t, idx - L /* src/ccoherent.cc:232 */, sqrtf(b.m / (float)L) /* :204 */, D);
    }
}

// ---- K2a / K2b ------------------------------------------------------------------------------------
constexpr int kAlignThreads = 256;

// lag (in samples) the digital mode shifts `row` of batch block t by
__device__ __forceinline__ int align_shift(const AlignArgs &a, int row, int t)
{
    if (!a.digital) return 0;
    // when the cross-correlation ran, its kernels wrote lag[t][row] for every owned row -- measured, or the carried
    // value republished for a row that was not requested (xcorr_skip) -- so the carried state itself, which the
    // next batch's K1 may already be updating on its own stream, is only read by batches that ran no K1
    return a.xcorr_ran ? a.lag[(size_t)t * a.nrows + row] : a.lag_state[row];
}

// one 32-bit word = samples (2i, 2i+1) of the row shifted by d: y[n] = s[n + d], zero outside [0,L)
__device__ __forceinline__ uint32_t shifted_word(const uint32_t *__restrict__ s32, int i, int d, int L, uint32_t xor80)
{
    const int m0 = 2 * i + d;                 // source sample of the low half
    const int j = m0 >> 1;                    // floor: source word of sample m0 (m0 even) or m0-1 (odd)
    const int nw = L >> 1;
    uint32_t w;
    if (m0 & 1) {
        const uint32_t lo = ((unsigned)j < (unsigned)nw) ? (s32[j] ^ xor80) : 0u;
        const uint32_t hi = ((unsigned)(j + 1) < (unsigned)nw) ? (s32[j + 1] ^ xor80) : 0u;
        w = (lo >> 16) | (hi << 16);
    } else {
        w = ((unsigned)j < (unsigned)nw) ? (s32[j] ^ xor80) : 0u;
    }
    return w; // out-of-range samples are exactly the zero-filled words / halves
}

// K2a: corr[t][row] = sum_n y[n] conj(r[n]) in exact integer arithmetic.  int8 products summed
// exactly (|sum| <= 2^15 L): the fp32 value csdrdevice::est_phasecorrect (src/csdrdevice.cc:62)
// accumulates is a rounding of this, scaled by 1/127^2 -- and the phasor is scale-invariant.
// 16 bytes = 8 complex samples of the row shifted by d, starting at output sample 8*i.  Interior
// vectors are ONE 16-byte load at a 2-byte-aligned address (gfx950 global loads need no natural
// alignment); the at most two vectors that straddle [0,L) fall back to word-wise assembly.
struct __attribute__((packed, aligned(2))) u4_unaligned { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 shifted_vec(const int8_t *__restrict__ srow, int i, int d, int L, uint32_t xor80)
{
    const int m0 = 8 * i + d;
    uint4 v;
    if (m0 >= 0 && m0 + 8 <= L) {
        const u4_unaligned u = *reinterpret_cast<const u4_unaligned *>(srow + 2 * (ptrdiff_t)m0);
        v = make_uint4(u.x ^ xor80, u.y ^ xor80, u.z ^ xor80, u.w ^ xor80);
    } else {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(srow);
        v = make_uint4(shifted_word(s32, 4 * i, d, L, xor80), shifted_word(s32, 4 * i + 1, d, L, xor80),
                       shifted_word(s32, 4 * i + 2, d, L, xor80), shifted_word(s32, 4 * i + 3, d, L, xor80));
    }
    return v;
}

// sum of v over the 64 lanes of a wave, valid in lane 63 (all lanes must be active): four row_shr steps leave each
// 16-lane row's sum in its last lane, row_bcast:15 / row_bcast:31 carry them up to lane 63 -- VALU only, no LDS crossbar
__device__ __forceinline__ int wave_sum_lane63(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// U vectors i0, i0 + stride, ... (those < i_end) with all loads issued back to back: a vector that straddles or lies
// outside [0,L) loads the row start instead and is patched afterwards (a branch around every load would make the
// compiler wait for one before issuing the next)
template <int U>
__device__ __forceinline__ void shifted_vecs(uint4 (&sv)[U], const int8_t *__restrict__ srow, int i0, int stride, int i_end, int d, int L,
                                             uint32_t xor80)
{
    bool edge = false;
#pragma unroll
    for (int q = 0; q < U; ++q) {
        const int i = i0 + q * stride, m0 = 8 * i + d;
        const bool live = i < i_end, inside = live && m0 >= 0 && m0 + 8 <= L;
        const u4_unaligned u = *reinterpret_cast<const u4_unaligned *>(srow + 2 * (ptrdiff_t)(inside ? m0 : 0));
        sv[q] = make_uint4(u.x ^ xor80, u.y ^ xor80, u.z ^ xor80, u.w ^ xor80);
        edge |= live && !inside;
    }
    if (edge) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int i = i0 + q * stride, m0 = 8 * i + d;
            if (i < i_end && !(m0 >= 0 && m0 + 8 <= L)) sv[q] = shifted_vec(srow, i, d, L, xor80);
        }
    }
}

// one word = two samples of s . conj(r).  With the reference word's I / Q bytes swapped, rs = [rQ0 rI0 rQ1 rI1]:
//     re += dot4(s, r)          = I.rI + Q.rQ
//     cr += dot4(s, rs)         = I.rQ + Q.rI
//     nq += dot4(s, rs & 0x00FF00FF) = I.rQ                         ->  im = Q.rI - I.rQ = cr - 2 nq, once, after the sums
// THREE instructions per signal word; the swap (ONE v_perm_b32) and the mask are per REFERENCE word, i.e. they do not depend
// on the row (r02: perm of the signal word + two masks of the reference word + three dot4 = six per signal word).
// Ranges at B <= 16384: |re|, |cr| <= 8192 * 2 * 128^2 = 2^28, |nq| <= 2^27 -- exact in int32, and so is cr - 2 nq.
__device__ __forceinline__ uint32_t ref_swap(uint32_t r) { return __builtin_amdgcn_perm(r, r, 0x02030001u); }   // [rQ0 rI0 rQ1 rI1]
__device__ __forceinline__ void dot_word3(uint32_t s, uint32_t r, uint32_t rs, uint32_t rsq, int &re, int &cr, int &nq)
{
    re = __builtin_amdgcn_sdot4((int)s, (int)r, re, false);
    cr = __builtin_amdgcn_sdot4((int)s, (int)rs, cr, false);
    nq = __builtin_amdgcn_sdot4((int)s, (int)rsq, nq, false);
}
// same sums from the reference word alone (forms its swapped / masked forms here: five instructions per word)
__device__ __forceinline__ void dot_word3(uint32_t s, uint32_t r, int &re, int &cr, int &nq)
{
    const uint32_t rs = ref_swap(r);
    dot_word3(s, r, rs, rs & 0x00FF00FFu, re, cr, nq);
}
__device__ __forceinline__ void dot_word(uint32_t s, uint32_t r, int &re, int &im)
{
    int cr = 0, nq = 0;
    dot_word3(s, r, re, cr, nq);
    im += cr - 2 * nq;
}

// U: 16-byte vectors a thread has in flight per loop iteration (4: one 16 KiB chunk per pass of the workgroup; 8: the long
// rows' 32 KiB chunks in one pass -- 16 loads out before the first use instead of two rounds of 8)
template <bool VEC, int U = 4>
__global__ __launch_bounds__(kAlignThreads) void k_phase_dot(AlignArgs a)
{
    __shared__ long long sred[2 * (kAlignThreads / 64)];
    const int tid = threadIdx.x, t = blockIdx.y;
    const int row = a.row_begin + (int)blockIdx.x;
    const int B = a.B, L = B >> 1;
    const int8_t *blk = a.rows + (size_t)t * a.block_stride;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(blk + (size_t)row * B);
    const uint32_t *r32 = reinterpret_cast<const uint32_t *>(blk);
    const int d = align_shift(a, row, t);
    // int32 partials are safe (<= 2^16 per word, <= 2^13 words per thread)
    int re = 0, im = 0, cr = 0, nq = 0;
    // long rows are split over grid.z chunks (one chunk for B <= 16 KiB)
    const int nchunk = gridDim.z, v_lo = (int)(((long long)(B / 16) * blockIdx.z) / nchunk),
              v_hi = (int)(((long long)(B / 16) * (blockIdx.z + 1)) / nchunk);
    if constexpr (VEC) {
        const int8_t *srow = blk + (size_t)row * B;
        const uint4 *r128 = reinterpret_cast<const uint4 *>(blk);
        for (int i0 = v_lo + tid; i0 < v_hi; i0 += U * kAlignThreads) {
            uint4 sv[U], rv[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int i = i0 + q * kAlignThreads;
                rv[q] = r128[i < v_hi ? i : i0];
            }
            shifted_vecs<U>(sv, srow, i0, kAlignThreads, v_hi, d, L, a.xor80);
#pragma unroll
            for (int q = 0; q < U; ++q) {
                if (i0 + q * kAlignThreads < v_hi) {
                    dot_word3(sv[q].x, rv[q].x ^ a.xor80, re, cr, nq);
                    dot_word3(sv[q].y, rv[q].y ^ a.xor80, re, cr, nq);
                    dot_word3(sv[q].z, rv[q].z ^ a.xor80, re, cr, nq);
                    dot_word3(sv[q].w, rv[q].w ^ a.xor80, re, cr, nq);
                }
            }
        }
        im = cr - 2 * nq;
    } else {
        for (int i = 4 * v_lo + tid; i < 4 * v_hi; i += kAlignThreads) {
            const uint32_t s = (d == 0) ? (s32[i] ^ a.xor80) : shifted_word(s32, i, d, L, a.xor80);
            dot_word(s, r32[i] ^ a.xor80, re, im);
        }
    }
    long long acc_re = re, acc_im = im;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        acc_re += __shfl_xor(acc_re, off, 64);
        acc_im += __shfl_xor(acc_im, off, 64);
    }
    if ((tid & 63) == 0) { sred[2 * (tid >> 6)] = acc_re; sred[2 * (tid >> 6) + 1] = acc_im; }
    __syncthreads();
    if (tid == 0) {
        long long sr = 0, si = 0;
        for (int w = 0; w < kAlignThreads / 64; ++w) { sr += sred[2 * w]; si += sred[2 * w + 1]; }
        long long *c = a.corr + 2 * ((size_t)t * a.nrows + row);
        if (nchunk == 1) { c[0] = sr; c[1] = si; }
        else { // integer partial sums: the order of the adds does not matter (corr zeroed by the host)
            atomicAdd(reinterpret_cast<unsigned long long *>(c), (unsigned long long)sr);
            atomicAdd(reinterpret_cast<unsigned long long *>(c + 1), (unsigned long long)si);
        }
    }
}

// K2b: grid.x = 1 + owned signal rows; block x = 0 writes the packet header and copies the raw
// reference row (cpacketize::write(int8*) src/cpacketizer.cc:137-156); block x >= 1 handles row
// row_begin + x - 1 of batch block t = blockIdx.y.
// rotate + quantise the two samples of one word: csdrdevice::phasecorrect (src/csdrdevice.cc:80-84)
// then cdsp::convto8bit (src/cdsp.cc:51-54), single-rounding ops in the oracle's order
__device__ __forceinline__ uint32_t rotq_word(uint32_t s, float2 p)
{
    // On packed pairs (cpk.hpp): per sample  x = (I, Q) * (1/127)  [v_pk_mul],  t1 = x * p.x, t2 = (Q, I) * p.y  [2 v_pk_mul],
    // y = (t1.x - t2.x, t1.y + t2.y)  [v_pk_add, neg_lo] -- the four products and two sums of rot_rn, each rounded once --
    // then y * 127 [v_pk_mul], round-half-even, and  +128 -> v_cvt_pk_u8_f32 (saturating to [0, 255] = the clamp)
    // straight into the byte lane; one XOR turns the four offset-binary bytes back into two's complement.
    const c2 pp = c2{p.x, p.y};
    uint32_t out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        c2 x = c2{(float)sext8(s, 2 * h), (float)sext8(s, 2 * h + 1)};
        x = x * (1.0f / 127.0f);
        c2 t1, t2, y;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t1) : "v"(x), "v"(pp));                                  // (I px, Q px)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t2) : "v"(x), "v"(pp));                     // (Q py, I py)
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(y) : "v"(t1), "v"(t2));                                    // (I px - Q py, Q px + I py)
        y = y * 127.0f;
        // round first, then let v_cvt_pk_u8_f32 saturate: rint(clamp(x)) == clamp(rint(x)) on [-128, 127] (ties to even keep
        // 127.5 -> 128 -> 127 and -128.5 -> -128), and r + 128 is exact, so the explicit clamp of cdsp::convto8bit is implied.
        // rintf for both halves in ONE packed add: |y| < 2^22, so y + 1.5 * 2^23 lands where the spacing is 1 and its own
        // round-to-nearest-even IS rintf(y) (the constant is even: ties fall on the same parity); taking 1.5 * 2^23 - 128
        // off again is exact and leaves rintf(y) + 128.  (-1 instruction per sample against v_rndne x 2 + one add.)
        const c2 u = (y + 12582912.0f) - 12582784.0f;
        // byte 0 first: its "old" operand may be anything (all four bytes get written), so no zero has to be materialised
        if (h == 0) asm("v_cvt_pk_u8_f32 %0, %1, 0, %1" : "=v"(out) : "v"(u.x));
        else asm("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(out) : "v"(u.x), "n"(2 * h));
        asm("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(out) : "v"(u.y), "n"(2 * h + 1));
    }
    return out ^ 0x80808080u;
}

// K2 chain: one thread per owned row walks the batch in block order.
// csdrdevice::est_phasecorrect (src/csdrdevice.cc:58-69) per block:
//   phasecorr = conj(corr)/|corr| (:63);  phasecorr = 0.5 phasecorr + 0.5 phasecorrprev (:66-67)
// |corr| == 0 holds the previous phasor (defined policy; the reference would go NaN for ever).  With the
// reference noise off the estimate is frozen (src/ccoherent.cc:271) and every block gets phase_in.
// The chain is sequential in t by definition; doing it once per row here (instead of once per (row, block)
// workgroup of k_align_quant) keeps the batch cost linear in T.
__global__ void k_phase_chain(AlignArgs a, int row_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_count) return;
    const int row = a.row_begin + i;
    if (!a.xcorr_ran) {
        // no lag was measured in this batch: get_lagp() still returns the last one (include/csdrdevice.h:161)
        const int32_t lg = a.lag_state[row];
        const float mg = a.mag_state[row], fr = a.frac_state[row];
        for (int t = 0; t < a.nblocks; ++t) {
            const size_t o = (size_t)t * a.nrows + row;
            a.lag_out[o] = lg; a.mag_out[o] = mg; a.frac_out[o] = fr;
        }
    }
    float2 p = a.phase_in[row];
    for (int t = 0; t < a.nblocks; ++t) {
        if (a.refnoise) {
            const long long sr = a.corr[2 * ((size_t)t * a.nrows + row)], si = a.corr[2 * ((size_t)t * a.nrows + row) + 1];
            if (sr != 0 || si != 0) {
                const double cr = (double)sr, ci = (double)si;
                const double inv = 1.0 / sqrt(cr * cr + ci * ci);
                const float pr = (float)(cr * inv), pi = (float)(-ci * inv);
                p = make_float2(__fadd_rn(__fmul_rn(0.5f, pr), __fmul_rn(0.5f, p.x)),
                                __fadd_rn(__fmul_rn(0.5f, pi), __fmul_rn(0.5f, p.y)));
            }
        }
        a.phasor[(size_t)t * a.nrows + row] = p;
    }
    a.phase_out[row] = p;                                   // state carried to the next batch
}

template <bool VEC>
__global__ __launch_bounds__(kAlignThreads) void k_align_quant(AlignArgs a)
{
    const int tid = threadIdx.x, t = blockIdx.y;
    const int B = a.B, L = B >> 1;
    const size_t moff = 16 + 4 * (size_t)a.nrows;
    const int8_t *blk = a.rows + (size_t)t * a.block_stride;
    int8_t *packet = a.packet + (size_t)t * a.packet_stride;
    if (a.slab) {
        if (blockIdx.x == 0 && (t < a.hdr_first || t >= a.hdr_first + a.hdr_count)) return; // no packet for this block here
        packet = a.packet + (size_t)(t - a.hdr_first) * a.packet_stride;
    }
    const int nchunk = gridDim.z, v_lo = (int)(((long long)(B / 16) * blockIdx.z) / nchunk),
              v_hi = (int)(((long long)(B / 16) * (blockIdx.z + 1)) / nchunk);
    if (blockIdx.x == 0) {
        // header hdr0{globalseqn,N,L,unused} src/cpacketizer.cc:112-116 and readcnt words :142,163
        uint32_t *h = reinterpret_cast<uint32_t *>(packet);
        const uint32_t seq = a.seq + (uint32_t)t;
        if (blockIdx.z == 0) {
            if (tid == 0) { h[0] = seq; h[1] = (uint32_t)a.nrows; h[2] = (uint32_t)L; h[3] = 0u; }
            for (int r = tid; r < a.nrows; r += kAlignThreads) h[4 + r] = a.readcnt ? a.readcnt[(size_t)t * a.nrows + r] : seq;
        }
        if constexpr (VEC) {
            const uint4 *src = reinterpret_cast<const uint4 *>(blk);
            uint4 *dst = reinterpret_cast<uint4 *>(packet + moff);
            for (int i = v_lo + tid; i < v_hi; i += kAlignThreads) {
                const uint4 v = src[i];
                dst[i] = make_uint4(v.x ^ a.xor80, v.y ^ a.xor80, v.z ^ a.xor80, v.w ^ a.xor80);
            }
        } else {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(blk);
            uint32_t *dst = reinterpret_cast<uint32_t *>(packet + moff);
            for (int i = 4 * v_lo + tid; i < 4 * v_hi; i += kAlignThreads) dst[i] = src[i] ^ a.xor80;
        }
        return;
    }
    const int row = a.row_begin + (int)blockIdx.x - 1;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(blk + (size_t)row * B);
    const int d = align_shift(a, row, t);
    int8_t *orow = a.slab ? a.slab + (size_t)t * a.slab_stride + (size_t)(row - a.row_begin) * B : packet + moff + (size_t)row * B;

    float2 p;
    if (a.inline_chain) {
        // a one-block batch: the chain is a single step, folded here by every workgroup of the row (same operands, same
        // operations as k_phase_chain: identical bits); chunk 0 publishes it -- one launch and one launch gap less per block
        p = a.phase_in[row];
        if (a.refnoise) {
            const long long sr = a.corr[2 * ((size_t)t * a.nrows + row)], si = a.corr[2 * ((size_t)t * a.nrows + row) + 1];
            if (sr != 0 || si != 0) {
                const double cr = (double)sr, ci = (double)si;
                const double inv = 1.0 / sqrt(cr * cr + ci * ci);
                const float pr = (float)(cr * inv), pi = (float)(-ci * inv);
                p = make_float2(__fadd_rn(__fmul_rn(0.5f, pr), __fmul_rn(0.5f, p.x)), __fadd_rn(__fmul_rn(0.5f, pi), __fmul_rn(0.5f, p.y)));
            }
        }
        if (blockIdx.z == 0 && tid == 0) {
            const size_t o = (size_t)t * a.nrows + row;
            a.phasor[o] = p;
            a.phase_out[row] = p;
            if (!a.xcorr_ran) { a.lag_out[o] = a.lag_state[row]; a.mag_out[o] = a.mag_state[row]; a.frac_out[o] = a.frac_state[row]; }
        }
    } else
        p = a.phasor[(size_t)t * a.nrows + row];      // get_phasecorrect() after block t (k_phase_chain)

    // csdrdevice::phasecorrect (src/csdrdevice.cc:80-84) + cpacketize::write(complex<float>*)
    // (src/cpacketizer.cc:158-172): y * p, x127, saturate, round-half-even, int8 at the row offset
    if constexpr (VEC) {
        const int8_t *srow = blk + (size_t)row * B;
        uint4 *o128 = reinterpret_cast<uint4 *>(orow);
        for (int i0 = v_lo + tid; i0 < v_hi; i0 += 4 * kAlignThreads) {
            uint4 sv[4];
            shifted_vecs<4>(sv, srow, i0, kAlignThreads, v_hi, d, L, a.xor80);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * kAlignThreads;
                if (i < v_hi) o128[i] = make_uint4(rotq_word(sv[q].x, p), rotq_word(sv[q].y, p), rotq_word(sv[q].z, p), rotq_word(sv[q].w, p));
            }
        }
    } else {
        uint32_t *o32 = reinterpret_cast<uint32_t *>(orow);
        for (int i = 4 * v_lo + tid; i < 4 * v_hi; i += kAlignThreads) {
            const uint32_t s = (d == 0) ? (s32[i] ^ a.xor80) : shifted_word(s32, i, d, L, a.xor80);
            o32[i] = rotq_word(s, p);
        }
    }
}

// ---- K2 fused: dot -> phasor chain -> rotate with the row read ONCE ------------------------------------
// One workgroup per (row, block) holds its 16 KiB row in registers (<= 4 x 16 B per thread), forms the exact
// integer dot product, and then needs the phasor of the SAME row one block earlier (the EMA of
// src/csdrdevice.cc:66-67 is a chain in block order).  That value comes from another workgroup, so:
//   * work order = workgroup index, block-major: (row, t) depends only on a LOWER index.  Every hardware
//     dispatch queue hands out workgroups in index order, so the lowest-indexed unfinished workgroup of the grid
//     is always resident and never has to wait -- the grid cannot deadlock (the argument of decoupled look-back
//     scans; a ticket drawn from one atomic counter would make it independent of the dispatcher, but 16 k
//     atomics on one address cost more than the second read of the rows saves: measured, 14 vs 9.6 us / block);
//   * what is handed over is not the chain value but each block's own UNIT phasor (one 64-bit agent-scope atomic
//     per (row, block), published right after the dot product; all-ones = "not yet", 0 = "|corr| = 0, hold");
//     workgroup (row, t) fetches the t earlier ones with one load per lane and folds the EMA itself, in block
//     order, from phase_in -- the same operations in the same order as a sequential chain, but nobody waits
//     for anybody's rotation, only for dot products that started a thousand workgroups earlier;
//   * the wait is bounded and nothing depends on it: after kFusedSpinLimit polls a workgroup forms the missing
//     block's dot product itself (same integers, same phasor), so the kernel terminates and stays exact under any
//     scheduling -- e.g. when another process shares the GPU and this grid's dispatch is paused while its resident
//     waves drain, which is where waiting workgroups would otherwise sit on predecessors that cannot start.
//     *status counts how often that happened (0 in every run on an exclusive GPU).
// Grid: ((1 + owned rows) * nblocks) workgroups; 16-byte aligned rows only, B <= 16384 (one chunk).
// Measured (r01, 16 blocks x 1025 rows): 8.8 us per block against 9.5 us for k_phase_dot + k_phase_chain +
// k_align_quant; the locked cadence (phase path only) 92.7 k vs 81.6 k blocks/s.
struct FusedSync {
    unsigned int *ticket;              // unused (kept for a ticketed variant)
    unsigned int *status;              // [0] number of look-back waits that ran out and were computed locally
    unsigned long long *chain;         // [T][nrows] at stride 2: this batch's unit phasor bits, all-ones = not yet
    unsigned long long *rearm;         // the other slot of the same entries: reset to all-ones for the next batch (or null)
    unsigned long long *chainv;        // [T][nrows] at stride 2: the CHAIN value (phasor after block t) once a workgroup has folded it; all-ones = not yet
    unsigned long long *rearmv;        // its other slot, re-armed like `rearm`
    int row_count;
    int spin_limit;                    // polls before the local fallback (kFusedSpinLimit); < 0: treat every earlier block as missing (tests)
};
constexpr unsigned long long kChainEmpty = ~0ull;
constexpr int kFusedSpinLimit = 2048;      // polls of ~1 us each before a workgroup stops waiting and computes the value itself

// FULL: B == 16384, every thread owns exactly four 16-byte vectors of the row (no bounds checks in the hot loops)
// XOR: the input is offset binary (CRSDR_OFFSET_BINARY); false folds the 40-odd "^ xor80" of a thread away (5 % of its VALU work)
template <bool FULL, bool XOR>
__global__ __launch_bounds__(kAlignThreads, 8) void k_align_fused(AlignArgs a_, FusedSync fs)
{
    AlignArgs a = a_;
    a.xor80 = XOR ? a_.xor80 : 0u;                 // a compile-time zero without XOR
    __shared__ long long sred[2 * (kAlignThreads / 64)];
    __shared__ float2 sp;
    __shared__ unsigned long long smiss, sfix[64];
    __shared__ int sstar;
    const int tid = threadIdx.x;
    const unsigned int per = (unsigned)fs.row_count + 1u;
    const unsigned int ticket = blockIdx.x;
    const int t = (int)(ticket / per), x = (int)(ticket % per);
    const int B = FULL ? 16384 : a.B, L = B >> 1, nvec = B / 16;
    const size_t moff = 16 + 4 * (size_t)a.nrows;
    const int8_t *blk = a.rows + (size_t)t * a.block_stride;
    int8_t *packet = a.packet + (size_t)t * a.packet_stride;
    if (a.slab) {
        if (x == 0 && (t < a.hdr_first || t >= a.hdr_first + a.hdr_count)) return;
        packet = a.packet + (size_t)(t - a.hdr_first) * a.packet_stride;
    }
    if (x == 0) {
        // header hdr0{globalseqn,N,L,unused} + readcnt words + the raw reference row (src/cpacketizer.cc:112-116,137-156)
        uint32_t *h = reinterpret_cast<uint32_t *>(packet);
        const uint32_t seq = a.seq + (uint32_t)t;
        if (tid == 0) { h[0] = seq; h[1] = (uint32_t)a.nrows; h[2] = (uint32_t)L; h[3] = 0u; }
        for (int r = tid; r < a.nrows; r += kAlignThreads) h[4 + r] = a.readcnt ? a.readcnt[(size_t)t * a.nrows + r] : seq;
        const uint4 *src = reinterpret_cast<const uint4 *>(blk);
        uint4 *dst = reinterpret_cast<uint4 *>(packet + moff);
        for (int i = tid; i < nvec; i += kAlignThreads) {
            const uint4 v = src[i];
            dst[i] = make_uint4(v.x ^ a.xor80, v.y ^ a.xor80, v.z ^ a.xor80, v.w ^ a.xor80);
        }
        return;
    }
    const int row = a.row_begin + x - 1;
    const size_t o = (size_t)t * a.nrows + row;
    const int d = align_shift(a, row, t);
    const int8_t *srow = blk + (size_t)row * B;
    // All eight 16-byte loads of a thread (its four row vectors and the reference row's) are issued back to back
    // with no branch between them: an interior vector is ONE load at a 2-byte-aligned address, and a vector that
    // straddles or lies outside [0,L) loads from the row start instead and is patched afterwards (rare: at most two
    // straddle; a branch around each load would make the compiler wait for one before issuing the next).
    uint4 sv[4], rv[4];
    bool edge = false;
    if (a.refnoise) {                            // first: these do not wait for the lag
        const uint4 *r128 = reinterpret_cast<const uint4 *>(blk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * kAlignThreads;
            rv[q] = r128[(FULL || i < nvec) ? i : 0];
        }
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 u32x4_u2 __attribute__((aligned(2)));
    if (a.nt & 2) {            // ONE uniform branch around all four loads (a branch per load would serialise them)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * kAlignThreads;
            const int m0 = 8 * i + d;
            const bool inside = (FULL || i < nvec) && m0 >= 0 && m0 + 8 <= L;
            // the row is read once: non-temporal, out of the way of the reference row in L2 / the memory-side cache
            const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u2 *>(srow + 2 * (ptrdiff_t)(inside ? m0 : 0)));
            sv[q] = make_uint4(u.x, u.y, u.z, u.w);
            edge |= !inside;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * kAlignThreads;
            const int m0 = 8 * i + d;
            const bool inside = (FULL || i < nvec) && m0 >= 0 && m0 + 8 <= L;
            const u4_unaligned u = *reinterpret_cast<const u4_unaligned *>(srow + 2 * (ptrdiff_t)(inside ? m0 : 0));
            sv[q] = make_uint4(u.x, u.y, u.z, u.w);
            edge |= !inside;
        }
    }
    if (edge) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * kAlignThreads;
            const int m0 = 8 * i + d;
            if (!FULL && i >= nvec) sv[q] = make_uint4(a.xor80, a.xor80, a.xor80, a.xor80);
            else if (!(m0 >= 0 && m0 + 8 <= L)) {
                const uint4 e = shifted_vec(srow, i, d, L, a.xor80);
                sv[q] = make_uint4(e.x ^ a.xor80, e.y ^ a.xor80, e.z ^ a.xor80, e.w ^ a.xor80);
            }
        }
    }
    // wave 0 also starts what the fold will need -- the carried phasor and the unit phasors that earlier blocks of
    // this row have already published -- so that their latency hides behind the row loads instead of following them
    float2 p_in = make_float2(0.f, 0.f);
    unsigned long long bits = 0ull;          // wave 0, lane u <= t: unit phasor of block u of this row
    unsigned long long cv = kChainEmpty;     // wave 0, lane u < t: the chain value after block u, where its workgroup has got that far
    if (tid < 64) {
        p_in = a.phase_in[row];
        if (a.refnoise && tid < t) {
            bits = fs.spin_limit < 0 ? kChainEmpty
                                     : __hip_atomic_load(fs.chain + 2 * ((size_t)tid * a.nrows + row), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fs.spin_limit >= 0) cv = __hip_atomic_load(fs.chainv + 2 * ((size_t)tid * a.nrows + row), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sv[q] = make_uint4(sv[q].x ^ a.xor80, sv[q].y ^ a.xor80, sv[q].z ^ a.xor80, sv[q].w ^ a.xor80);
    if (a.refnoise) {
        int re = 0, cr = 0, nq = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * kAlignThreads;
            if (FULL || i < nvec) {
                dot_word3(sv[q].x, rv[q].x ^ a.xor80, re, cr, nq);
                dot_word3(sv[q].y, rv[q].y ^ a.xor80, re, cr, nq);
                dot_word3(sv[q].z, rv[q].z ^ a.xor80, re, cr, nq);
                dot_word3(sv[q].w, rv[q].w ^ a.xor80, re, cr, nq);
            }
        }
        int im = cr - 2 * nq;
        // B <= 16384 here: |sum| <= 8192 * 2 * 127^2 < 2^28, so the wave sums are exact in 32 bits (six DPP adds each)
        re = wave_sum_lane63(re);
        im = wave_sum_lane63(im);
        if ((tid & 63) == 63) { sred[2 * (tid >> 6)] = re; sred[2 * (tid >> 6) + 1] = im; }
    }
    __syncthreads();
    // unit phasor conj(corr)/|corr| from the integer sums; 0 = "|corr| == 0, hold the previous phasor"
    auto unit_bits = [](long long sr, long long si) -> unsigned long long {
        if (sr == 0 && si == 0) return 0ull;
        const double cr = (double)sr, ci = (double)si;
        const double inv = 1.0 / sqrt(cr * cr + ci * ci);
        return (unsigned long long)__float_as_uint((float)(cr * inv)) | ((unsigned long long)__float_as_uint((float)(-ci * inv)) << 32);
    };
    if (tid < 64) {
        if (tid == 0 && !a.xcorr_ran) {   // no lag measured in this batch: republish the carried one (include/csdrdevice.h:161)
            a.lag_out[o] = a.lag_state[row]; a.mag_out[o] = a.mag_state[row]; a.frac_out[o] = a.frac_state[row];
        }
        if (a.refnoise) {
            // csdrdevice::est_phasecorrect (src/csdrdevice.cc:58-69) for THIS block, published at once
            unsigned long long mine = 0ull;
            if (tid == 0) {
                long long sr = 0, si = 0;
                for (int w = 0; w < kAlignThreads / 64; ++w) { sr += sred[2 * w]; si += sred[2 * w + 1]; }
                mine = unit_bits(sr, si);
                __hip_atomic_store(fs.chain + 2 * o, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fs.rearm) fs.rearm[2 * o] = kChainEmpty;    // nobody reads this slot before the next launch
            }
            // lane u < t holds the unit phasor of block u (published right after that block's dot product, long
            // before its own rotation) and, where block u's workgroup has already folded it, the CHAIN value after block
            // u, both from the prefetch above.  The fold starts from the latest chain value there is (block ustar; the
            // carried phasor if none) and needs the unit phasors of the blocks after it only: those are polled for.
            // The chain value is bit for bit what folding from the carried phasor gives, so the start point changes
            // nothing but the length of the fold (t + 1 steps at first: 8 % of the kernel's VALU work at T = 64).
            int ustar = -1;
            {
                const unsigned long long *srcu = fs.chain + 2 * ((size_t)tid * a.nrows + row);
                const unsigned long long *srcv = fs.chainv + 2 * ((size_t)tid * a.nrows + row);
                int spins = 0;
                for (;;) {
                    const unsigned long long have = __ballot(tid < t && cv != kChainEmpty);
                    ustar = have ? 63 - __builtin_clzll(have) : -1;
                    const bool need = tid < t && tid > ustar && bits == kChainEmpty;
                    if (!__ballot(need) || spins >= fs.spin_limit) break;
                    if (need) bits = __hip_atomic_load(srcu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (tid < t && tid > ustar && cv == kChainEmpty) cv = __hip_atomic_load(srcv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_s_sleep(8);
                    ++spins;
                }
            }
            const unsigned long long own = __shfl(mine, 0, 64);
            if (tid == t) bits = own;
            const unsigned long long missing = __ballot(tid < t && tid > ustar && bits == kChainEmpty);
            if (tid == 0) { smiss = missing; sstar = ustar; }
        } else if (tid == 0) { smiss = 0ull; sstar = -1; }
    }
    __syncthreads();
    // Fallback, normally never taken: an earlier block's workgroup has not published within the poll budget (its
    // queue was descheduled, e.g. another process shares the GPU and dispatch of this grid is paused while its
    // resident waves drain).  Nothing may depend on it then: the whole workgroup forms that block's dot product
    // itself -- same integers, same unit phasor -- so the kernel terminates and stays exact under any scheduling.
    for (unsigned long long miss = smiss; miss != 0ull; miss &= miss - 1ull) {
        const int u = __builtin_ctzll(miss);
        const int8_t *blku = a.rows + (size_t)u * a.block_stride;
        const int du = align_shift(a, row, u);
        const uint4 *r128 = reinterpret_cast<const uint4 *>(blku);
        int re = 0, im = 0;
        for (int i = tid; i < nvec; i += kAlignThreads) {
            const uint4 s = shifted_vec(blku + (size_t)row * B, i, du, L, a.xor80);
            const uint4 rv = r128[i];
            dot_word(s.x, rv.x ^ a.xor80, re, im);
            dot_word(s.y, rv.y ^ a.xor80, re, im);
            dot_word(s.z, rv.z ^ a.xor80, re, im);
            dot_word(s.w, rv.w ^ a.xor80, re, im);
        }
        long long acc_re = re, acc_im = im;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            acc_re += __shfl_xor(acc_re, off, 64);
            acc_im += __shfl_xor(acc_im, off, 64);
        }
        __syncthreads();                       // sred / sfix of the previous round are consumed
        if ((tid & 63) == 0) { sred[2 * (tid >> 6)] = acc_re; sred[2 * (tid >> 6) + 1] = acc_im; }
        __syncthreads();
        if (tid == 0) {
            long long sr = 0, si = 0;
            for (int w = 0; w < kAlignThreads / 64; ++w) { sr += sred[2 * w]; si += sred[2 * w + 1]; }
            sfix[u] = unit_bits(sr, si);
            atomicAdd(fs.status, 1u);          // counted, not an error: how often the fallback ran
        }
        __syncthreads();
    }
    if (tid < 64) {
        float2 p = p_in;
        if (a.refnoise) {
            const int ustar = sstar;           // wave-uniform: the block whose chain value the fold starts from (-1: the carried phasor)
            if (tid < t && tid > ustar && bits == kChainEmpty) bits = sfix[tid];
            if (ustar >= 0) {
                const unsigned sl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cv & 0xffffffffull), ustar),
                               sh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cv >> 32), ustar);
                p = make_float2(__uint_as_float(sl), __uint_as_float(sh));
            }
            const int blo = (int)(unsigned)(bits & 0xffffffffull), bhi = (int)(unsigned)(bits >> 32);
            for (int u = ustar + 1; u <= t; ++u) {     // one sequential fold in block order (src/csdrdevice.cc:66-67)
                // u is wave-uniform: v_readlane into scalar registers, not a trip through the LDS crossbar
                const unsigned rl = (unsigned)__builtin_amdgcn_readlane(blo, u), rh = (unsigned)__builtin_amdgcn_readlane(bhi, u);
                if ((rl | rh) != 0u) {
                    const float pr = __uint_as_float(rl), pi = __uint_as_float(rh);
                    p = make_float2(__fadd_rn(__fmul_rn(0.5f, pr), __fmul_rn(0.5f, p.x)),
                                    __fadd_rn(__fmul_rn(0.5f, pi), __fmul_rn(0.5f, p.y)));
                }
            }
        }
        if (tid == 0) {
            a.phasor[o] = p;                                    // get_phasecorrect() after block t
            if (t == a.nblocks - 1) a.phase_out[row] = p;       // state carried to the next batch
            sp = p;
            if (a.refnoise) {                                   // later blocks of this row may start their fold here
                if (t < a.nblocks - 1)
                    __hip_atomic_store(fs.chainv + 2 * o, (unsigned long long)__float_as_uint(p.x) | ((unsigned long long)__float_as_uint(p.y) << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fs.rearmv) fs.rearmv[2 * o] = kChainEmpty;
            }
        }
    }
    __syncthreads();
    const float2 p = sp;
    int8_t *orow = a.slab ? a.slab + (size_t)t * a.slab_stride + (size_t)(row - a.row_begin) * B : packet + moff + (size_t)row * B;
    uint4 *o128 = reinterpret_cast<uint4 *>(orow);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + q * kAlignThreads;
        if (FULL || i < nvec) {
#ifdef CRSDR_K2_EXPERIMENT     // diagnostics only (tools/k2_ab.py with a -DCRSDR_K2_EXPERIMENT build): bit 4 copy instead of rotate, bit 5 no stores
            if (a.nt & 32) { if (sv[q].x == 0x12345678u && p.x == 123.f) o128[i] = sv[q]; continue; }
            const uint4 v = (a.nt & 16) ? sv[q] : make_uint4(rotq_word(sv[q].x, p), rotq_word(sv[q].y, p), rotq_word(sv[q].z, p), rotq_word(sv[q].w, p));
#else
            const uint4 v = make_uint4(rotq_word(sv[q].x, p), rotq_word(sv[q].y, p), rotq_word(sv[q].z, p), rotq_word(sv[q].w, p));
#endif
            if (a.nt & 1) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(o128 + i));   // written once, read by nobody here
            else o128[i] = v;
        }
    }
}

// ---- fractional-delay correction for LDS-resident blocks (B <= 16384; crsdr_plan_set_frac_apply) ---------------------------
// The long-block form (longblock.hpp stages A, B'', C'') for a row that fits one workgroup's LDS: int8 row -> zero-padded forward
// transform -> every bin times H[f] = p / B . exp(+2 pi i f_s (lag + D) / B) (f_s the signed bin index) -> inverse transform -> the
// first L samples quantised like cdsp::convto8bit, over the row the phase kernel has just written (shifted by the integer lag
// only).  One workgroup per (owned row, block), the generic in-LDS network of fft_lds.hpp (DIF forward leaves digit-reversed
// order, the DIT inverse takes it back: the response is looked up at the bin a position holds).  The integer part of the
// exponent is reduced mod B in integers and read from the plan's forward table, the fractional part |f_s D / B| <= 1/2 . |D| by
// polynomial (x14p::cis2pi) -- as in the long-block pass; the CPU checker forms the ramp in double and rounds once: equal except +-1 LSB on a few entries per thousand (tests/test_gpu_fracdelay.py).
struct FracArgs {
    const int8_t *rows;
    size_t block_stride;
    int8_t *packet;
    size_t packet_stride;
    int8_t *slab;              // slab output (sharded plans) or nullptr
    size_t slab_stride;
    int nrows, row_begin;
    uint32_t xor80;
    const int32_t *lag;        // [T][nrows] this batch's lags
    const float *frac;         // [T][nrows] this batch's parabolic estimates
    const float *frac_override; // [nrows] or nullptr: D = gain * frac
    float gain;
    const float2 *phasor;      // [T][nrows] get_phasecorrect() after each block
};

template <int LOG2N>
__global__ __launch_bounds__(FftGeom<LOG2N>::THREADS) void k_frac_apply(FracArgs a, const float2 *__restrict__ tw)
{
    using G = FftGeom<LOG2N>;
    constexpr int N = G::N, L = N / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *A = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x, t = blockIdx.y;
    const int row = a.row_begin + (int)blockIdx.x;
    const size_t o = (size_t)t * a.nrows + row;
    load_row_to_lds<LOG2N>(A, a.rows + (size_t)t * a.block_stride + (size_t)row * N, false, a.xor80, tid);
    __syncthreads();
    fft_dif_range<LOG2N, -1, 0, G::NPASS>(A, tw, tid);
    {
        const uint32_t lagu = (uint32_t)a.lag[o];
        const float D = a.frac_override ? a.frac_override[row] : a.gain * a.frac[o];
        const float2 pr = a.phasor[o];
        const float invN = 1.0f / (float)N;
        const float2 pb = make_float2(pr.x * invN, pr.y * invN);
        const float dstep = D * invN;
        for (int j = tid; j < N; j += G::THREADS) {
            const int f = digit_reverse<LOG2N>(j);                      // A[j] holds bin f
            const int fs = f < L ? f : f - N;
            const float2 w = tw[((uint32_t)f * lagu) & (uint32_t)(N - 1)];      // W_N^m = exp(-2 pi i m / N): its conjugate is the integer part
            const c2 cf = x14p::cis2pi((float)fs * dstep);
            const float2 h = cmul(cmulc(pb, w), make_float2(cf.x, cf.y));
            A[j] = cmul(A[j], h);
        }
    }
    __syncthreads();
    fft_dit_range<LOG2N, +1, 0, G::NPASS>(A, tw, tid);
    // cpacketize::write(complex<float>*) -> cdsp::convto8bit (src/cpacketizer.cc:158-172, src/cdsp.cc:51-54) on the first L samples
    int8_t *orow = a.slab ? a.slab + (size_t)t * a.slab_stride + (size_t)(row - a.row_begin) * N
                          : a.packet + (size_t)t * a.packet_stride + 16 + 4 * (size_t)a.nrows + (size_t)row * N;
    uint32_t *o32 = reinterpret_cast<uint32_t *>(orow);
    for (int i = tid; i < L / 2; i += G::THREADS) {
        const float2 x = A[2 * i], y = A[2 * i + 1];
        const uint32_t b0 = (uint32_t)(uint8_t)f32_to_i8(x.x), b1 = (uint32_t)(uint8_t)f32_to_i8(x.y), b2 = (uint32_t)(uint8_t)f32_to_i8(y.x),
                       b3 = (uint32_t)(uint8_t)f32_to_i8(y.y);
        o32[i] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
}

// ---- slab assembly on a gather root ---------------------------------------------------
// recv [nsrc][nblocks][slab_words] (what an all-to-all of the ranks' slab buffers delivers) -> the matrix rows
// 1 + src*per .. of packet j.  Pure copy, grid (chunks, nblocks, nsrc).
template <typename W>
__global__ void k_assemble_slabs(int8_t *packets, size_t packet_stride, size_t body_off, const W *__restrict__ recv, int nblocks,
                                 size_t slab_words)
{
    const int j = blockIdx.y, src = blockIdx.z;
    const W *s = recv + ((size_t)src * nblocks + j) * slab_words;
    W *d = reinterpret_cast<W *>(packets + (size_t)j * packet_stride + body_off) + (size_t)src * slab_words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slab_words; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

// ---- exchange slots (the gather carries int8 rows PLUS {lag, mag, frac, phasor} per row) ------------
// A sharded plan in slab mode writes block t of a batch into the slot  slab + t*slab_stride:
//   [row_count][B] int8 rows | tail at +tail_offset:  int32 lag[rc] | float mag[rc] | float frac[rc] | float2 phasor[rc] | uint32 readcnt[rc]
// (24 bytes per owned row -- what set_lag (src/ccoherent.cc:232-233), the port-5557 payload (src/cpacketizer.cc:127, 131-134) and the
// packet header's read counters (src/cpacketizer.cc:142,163: each device's own block count, what clients detect drops by) consume on
// the assembling side: the rank that READS a dongle is the one that knows its counter).  The rows come from the phase kernels; the tail
// is packed here, after them.
__global__ void k_pack_tails(int8_t *slab, size_t slab_stride, size_t tail_offset, int row_begin, int row_count, int nrows,
                             const int32_t *__restrict__ lag, const float *__restrict__ mag, const float *__restrict__ frac,
                             const float2 *__restrict__ phasor, const uint32_t *__restrict__ readcnt, uint32_t seq)
{
    const int t = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_count) return;
    const size_t o = (size_t)t * nrows + row_begin + i;
    int8_t *tail = slab + (size_t)t * slab_stride + tail_offset;
    reinterpret_cast<int32_t *>(tail)[i] = lag[o];
    reinterpret_cast<float *>(tail + 4 * (size_t)row_count)[i] = mag[o];
    reinterpret_cast<float *>(tail + 8 * (size_t)row_count)[i] = frac[o];
    const float2 ph = phasor[o];                              // written as two floats: 12*rc need not be 8-byte aligned
    reinterpret_cast<float *>(tail + 12 * (size_t)row_count)[2 * i] = ph.x;
    reinterpret_cast<float *>(tail + 12 * (size_t)row_count)[2 * i + 1] = ph.y;
    reinterpret_cast<uint32_t *>(tail + 20 * (size_t)row_count)[i] = readcnt ? readcnt[o] : seq + (uint32_t)t;      // as the header kernels fill it
}

// On the assembling rank: slots [nsrc][nblocks][slot_stride] (what the exchange delivers: chunk z = rank (src_base + z)'s slots
// of the nblocks blocks assembled here; the rank's own chunk may be read straight from its send buffer: self_src) ->
//   rows  -> matrix rows 1 + src*per .. of packet j            (unless rows_in_place: they already landed there)
//   tails -> scalars block j:  int32 lag[nrows] | float mag[nrows] | float frac[nrows] | float phasor[nrows][2]  (row 0: zeros)
//         -> the read counters of the source rank's rows in packet j's header (the assembling rank's own plan wrote the header from
//            ITS counters: right for row 0 and its own rows, whatever it knew for the others)
// grid (chunks, nblocks, nsrc); W = uint4 (16-byte aligned everything) or uint32_t.
template <typename W>
__global__ void k_assemble_slots(int8_t *packets, size_t packet_stride, size_t body_off, int8_t *scalars, size_t scalars_stride, int nrows, int per, int B,
                                 const int8_t *__restrict__ recv, int nblocks, size_t slot_stride, size_t tail_offset, int has_tail, int src_base,
                                 int skip_rank, int self_rank, const int8_t *__restrict__ self_src, int rows_in_place)
{
    const int j = blockIdx.y, src = src_base + (int)blockIdx.z;
    if (src == skip_rank) return;
    const bool self = src == self_rank && self_src;
    const int8_t *slot = self ? self_src + (size_t)j * slot_stride : recv + ((size_t)blockIdx.z * nblocks + j) * slot_stride;
    if (!rows_in_place) {
        const size_t words = (size_t)per * B / sizeof(W);
        const W *s = reinterpret_cast<const W *>(slot);
        W *d = reinterpret_cast<W *>(packets + (size_t)j * packet_stride + body_off + (size_t)src * per * B);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
    }
    if (has_tail && blockIdx.x == 0) {
        const int8_t *tail = slot + tail_offset;
        const size_t n = (size_t)nrows, r0 = 1 + (size_t)src * per;
        uint32_t *hdr_cnt = reinterpret_cast<uint32_t *>(packets + (size_t)j * packet_stride + 16);      // hdr0 is 16 bytes, then uint32 readcnt[N]
        for (int i = threadIdx.x; i < per; i += blockDim.x) hdr_cnt[r0 + i] = reinterpret_cast<const uint32_t *>(tail + 20 * (size_t)per)[i];
        if (!scalars) return;
        int8_t *out = scalars + (size_t)j * scalars_stride;
        for (int i = threadIdx.x; i < per; i += blockDim.x) {
            reinterpret_cast<int32_t *>(out)[r0 + i] = reinterpret_cast<const int32_t *>(tail)[i];
            reinterpret_cast<float *>(out + 4 * n)[r0 + i] = reinterpret_cast<const float *>(tail + 4 * (size_t)per)[i];
            reinterpret_cast<float *>(out + 8 * n)[r0 + i] = reinterpret_cast<const float *>(tail + 8 * (size_t)per)[i];
            reinterpret_cast<float *>(out + 12 * n)[2 * (r0 + i)] = reinterpret_cast<const float *>(tail + 12 * (size_t)per)[2 * i];
            reinterpret_cast<float *>(out + 12 * n)[2 * (r0 + i) + 1] = reinterpret_cast<const float *>(tail + 12 * (size_t)per)[2 * i + 1];
        }
        if (src == 0 && threadIdx.x == 0) {                   // row 0 (the reference channel) has no lag / phasor: zeros, like crsdr_plan_fetch
            reinterpret_cast<int32_t *>(out)[0] = 0;
            reinterpret_cast<float *>(out + 4 * n)[0] = 0.f;
            reinterpret_cast<float *>(out + 8 * n)[0] = 0.f;
            reinterpret_cast<float *>(out + 12 * n)[0] = 0.f;
            reinterpret_cast<float *>(out + 12 * n)[1] = 0.f;
        }
    }
}

// ---- per-op kernels (class cdsp) ---------------------------------------------------------------
__global__ void k_op_convtosigned(const uint32_t *in, uint32_t *out, int nwords)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nwords) out[i] = in[i] ^ 0x80808080u;
}
__global__ void k_op_convtofloat(float *out, const int8_t *in, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i8_to_f32((int)in[i]);
}
__global__ void k_op_scalarmul(float2 *out, const float2 *in, float2 s, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rot_rn(in[i], s);
}
__global__ void k_op_convto8bit(int8_t *out, const float *in, int nfloats)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nfloats) out[i] = (int8_t)f32_to_i8(in[i]);
}
__global__ void k_op_magsquared(float *out, const float2 *in, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __fadd_rn(__fmul_rn(in[i].x, in[i].x), __fmul_rn(in[i].y, in[i].y));
}
__global__ void k_op_conjugatemul(float2 *out, const float2 *a, const float2 *b, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float2 x = a[i], y = b[i];
        out[i] = make_float2(__fadd_rn(__fmul_rn(x.x, y.x), __fmul_rn(x.y, y.y)),
                             __fsub_rn(__fmul_rn(x.y, y.x), __fmul_rn(x.x, y.y)));
    }
}
// single workgroup: pairwise (tree) sum -- differs from VOLK's sequential accumulator only in
// rounding; compared to tolerance (SURVEY 8 note: a10 is never bit-comparable)
__global__ __launch_bounds__(1024) void k_op_conj_dot(float *res, const float2 *a, const float2 *b, int n)
{
    __shared__ double sre[16], sim[16];
    double re = 0.0, im = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float2 x = a[i], y = b[i];
        re += (double)x.x * y.x + (double)x.y * y.y;
        im += (double)x.y * y.x - (double)x.x * y.y;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        re += __shfl_xor(re, off, 64);
        im += __shfl_xor(im, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { sre[threadIdx.x >> 6] = re; sim[threadIdx.x >> 6] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0, i = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { r += sre[w]; i += sim[w]; }
        res[0] = (float)r;
        res[1] = (float)i;
    }
}
__global__ __launch_bounds__(1024) void k_op_indexofmax(uint32_t *index, const float *in, int n)
{
    __shared__ ArgMax wred[16];
    ArgMax best = {in[0], 0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) argmax_take(best, in[i], i);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float om = __shfl_xor(best.m, off, 64);
        int oi = __shfl_xor(best.idx, off, 64);
        argmax_take(best, om, oi);
    }
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        ArgMax b = wred[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) argmax_take(b, wred[w].m, wred[w].idx);
        *index = (uint32_t)b.idx;
    }
}

// batched standalone FFT (cdsp::fft): one workgroup per transform, natural-order output
template <int LOG2N, int DIR>
__global__ __launch_bounds__(FftGeom<LOG2N>::THREADS) void k_op_fft(float2 *__restrict__ out,
                                                                    const float2 *__restrict__ in,
                                                                    const float2 *__restrict__ tw)
{
    using G = FftGeom<LOG2N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *A = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const float2 *src = in + (size_t)blockIdx.x * G::N;
    float2 *dst = out + (size_t)blockIdx.x * G::N;
    for (int j = tid; j < G::N; j += G::THREADS) A[j] = src[j];
    __syncthreads();
    fft_dif_range<LOG2N, DIR, 0, G::NPASS>(A, tw, tid);
    for (int j = tid; j < G::N; j += G::THREADS) dst[digit_reverse<LOG2N>(j)] = A[j];
}

} // namespace crsdr
