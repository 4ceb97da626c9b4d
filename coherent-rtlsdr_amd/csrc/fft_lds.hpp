// fft_lds.hpp -- LDS-resident, register-tiled complex FFT building blocks for gfx950.
//
// One workgroup transforms one row of N = 2^LOG2N complex fp32 points that lives in LDS
// (N * 8 bytes <= 128 KiB of the CU's 160 KiB).  The transform is an in-place mixed-radix
// decimation-in-frequency network (radix-16 register butterflies, one smaller radix for the
// remainder); its transpose, the decimation-in-time network, consumes the digit-reversed
// order the DIF leaves behind and returns natural order.  Cross-correlation therefore never
// needs a reorder pass:   DIF(signal) . conj(DIF(ref))  ->  DIT  ->  natural-order lags.
//
// What it replaces in the reference: fftwf_plan_many_dft forward / backward
// (src/ccoherent.cc:78-93) executed by cdsp::fft (src/cdsp.cc:110-120).  Same definition:
// X[k] = sum_n x[n] exp(DIR * 2 pi i n k / N), unnormalised; DIR = -1 forward, +1 backward.
#pragma once
#include <hip/hip_runtime.h>

namespace crsdr {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * b
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b)
{
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
// a * w where w is a forward-table twiddle; DIR > 0 uses conj(w)
template <int DIR>
__device__ __forceinline__ float2 ctw(float2 a, float2 w)
{
    return DIR < 0 ? cmul(a, w) : cmulc(a, w);
}
// multiply by exp(DIR * i * pi/2): forward -i, backward +i
template <int DIR>
__device__ __forceinline__ float2 mul_j(float2 a)
{
    return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kC1 = 0.92387953251128675613f; // cos(pi/8)
constexpr float kS1 = 0.38268343236508977173f; // sin(pi/8)

// multiply by W_16^K (forward: exp(-2 pi i K/16); backward: conjugate), K in {1,2,3,6,9}
template <int DIR, int K>
__device__ __forceinline__ float2 mul_w16(float2 a)
{
    if constexpr (K == 0) return a;
    else if constexpr (K == 4) return mul_j<DIR>(a);
    else if constexpr (K == 2) { // (1 -+ i)/sqrt2
        return DIR < 0 ? make_float2((a.x + a.y) * kSqrtHalf, (a.y - a.x) * kSqrtHalf)
                       : make_float2((a.x - a.y) * kSqrtHalf, (a.x + a.y) * kSqrtHalf);
    } else if constexpr (K == 6) { // (-1 -+ i)/sqrt2
        return DIR < 0 ? make_float2((a.y - a.x) * kSqrtHalf, -(a.x + a.y) * kSqrtHalf)
                       : make_float2(-(a.x + a.y) * kSqrtHalf, (a.x - a.y) * kSqrtHalf);
    } else {
        // general: w = (c, -s) forward
        constexpr float c = (K == 1) ? kC1 : (K == 3) ? kS1 : (K == 9) ? -kC1 : 0.f;
        constexpr float s = (K == 1) ? kS1 : (K == 3) ? kC1 : (K == 9) ? -kS1 : 0.f;
        static_assert(K == 1 || K == 3 || K == 9, "unsupported W16 power");
        return DIR < 0 ? make_float2(fmaf(a.x, c, a.y * s), fmaf(a.y, c, -a.x * s))
                       : make_float2(fmaf(a.x, c, -a.y * s), fmaf(a.y, c, a.x * s));
    }
}

template <int DIR>
__device__ __forceinline__ void dft2(float2 &a, float2 &b)
{
    float2 t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// natural-order 4-point DFT of (a,b,c,d) in place
template <int DIR>
__device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d)
{
    float2 t0 = cadd(a, c), t1 = csub(a, c);
    float2 t2 = cadd(b, d), t3 = mul_j<DIR>(csub(b, d));
    a = cadd(t0, t2);
    b = cadd(t1, t3);
    c = csub(t0, t2);
    d = csub(t1, t3);
}

template <int DIR>
__device__ __forceinline__ void dft8(float2 (&v)[8])
{
    // even / odd split, X[k] = E[k] + W8^k O[k], X[k+4] = E[k] - W8^k O[k]
    dft4<DIR>(v[0], v[2], v[4], v[6]);
    dft4<DIR>(v[1], v[3], v[5], v[7]);
    float2 o0 = v[1], o1 = mul_w16<DIR, 2>(v[3]), o2 = mul_j<DIR>(v[5]), o3 = mul_w16<DIR, 6>(v[7]);
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

template <int DIR>
__device__ __forceinline__ void dft16(float2 (&v)[16])
{
    // n = q + 4m, k = s + 4u:  X[s+4u] = sum_q W4^{qu} ( W16^{qs} sum_m v[q+4m] W4^{ms} )
    dft4<DIR>(v[0], v[4], v[8], v[12]);
    dft4<DIR>(v[1], v[5], v[9], v[13]);
    dft4<DIR>(v[2], v[6], v[10], v[14]);
    dft4<DIR>(v[3], v[7], v[11], v[15]);
    // after the column DFTs v[q + 4s] holds column q, output s
    v[5] = mul_w16<DIR, 1>(v[5]);   // q=1,s=1
    v[9] = mul_w16<DIR, 2>(v[9]);   // q=1,s=2
    v[13] = mul_w16<DIR, 3>(v[13]); // q=1,s=3
    v[6] = mul_w16<DIR, 2>(v[6]);   // q=2,s=1
    v[10] = mul_w16<DIR, 4>(v[10]); // q=2,s=2
    v[14] = mul_w16<DIR, 6>(v[14]); // q=2,s=3
    v[7] = mul_w16<DIR, 3>(v[7]);   // q=3,s=1
    v[11] = mul_w16<DIR, 6>(v[11]); // q=3,s=2
    v[15] = mul_w16<DIR, 9>(v[15]); // q=3,s=3
    // row DFTs over q for each s: inputs v[4s + q], outputs X[s + 4u] land in v[4s + u]
    dft4<DIR>(v[0], v[1], v[2], v[3]);
    dft4<DIR>(v[4], v[5], v[6], v[7]);
    dft4<DIR>(v[8], v[9], v[10], v[11]);
    dft4<DIR>(v[12], v[13], v[14], v[15]);
    // transpose 4x4 so that v[k] = X[k]  (v[4s+u] -> v[s+4u])
    float2 t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int R, int DIR>
__device__ __forceinline__ void dft(float2 (&v)[R])
{
    if constexpr (R == 2) dft2<DIR>(v[0], v[1]);
    else if constexpr (R == 4) dft4<DIR>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 8) dft8<DIR>(v);
    else if constexpr (R == 16) dft16<DIR>(v);
    else static_assert(R == 2, "unsupported radix");
}

// ---- pass schedule ------------------------------------------------------------------------
// pass p handles log2 radix 4 while >= 4 bits remain, the last pass takes the remainder.
template <int LOG2N>
struct FftGeom {
    static constexpr int N = 1 << LOG2N;
    static constexpr int NPASS = (LOG2N + 3) / 4;
    static constexpr int THREADS = (N / 16) < 64 ? 64 : ((N / 16) > 1024 ? 1024 : (N / 16));
    static constexpr int log2r(int p) { return (LOG2N - 4 * p) >= 4 ? 4 : (LOG2N - 4 * p); }
    static constexpr int log2m(int p) { return LOG2N - 4 * p - log2r(p); } // stride of pass p
};

// digit reversal of the DIF output order: A[j] = X[rev(j)]
template <int LOG2N>
__device__ __forceinline__ int digit_reverse(int j)
{
    using G = FftGeom<LOG2N>;
    int out = 0, shift = 0;
#pragma unroll
    for (int p = 0; p < G::NPASS; ++p) {
        const int lr = G::log2r(p), lm = G::log2m(p);
        const int digit = (j >> lm) & ((1 << lr) - 1);
        out |= digit << shift;
        shift += lr;
    }
    return out;
}

template <int LOG2N>
constexpr int digit_reverse_c(int j)                     // the same permutation for compile-time arguments
{
    using G = FftGeom<LOG2N>;
    int out = 0, shift = 0;
    for (int p = 0; p < G::NPASS; ++p) {
        out |= ((j >> G::log2m(p)) & ((1 << G::log2r(p)) - 1)) << shift;
        shift += G::log2r(p);
    }
    return out;
}

// One DIF pass (butterfly, then twiddle) or DIT pass (twiddle, then butterfly) over the whole
// row in LDS.  tw = forward table W_N^k, k in [0,N).  Groups of R points with stride M; each
// thread owns whole groups, so the pass is in place and needs a barrier only before / after.
template <int LOG2N, int P, int DIR, bool DIT>
__device__ __forceinline__ void fft_pass(float2 *A, const float2 *__restrict__ tw, int tid)
{
    using G = FftGeom<LOG2N>;
    constexpr int LR = G::log2r(P), R = 1 << LR, LM = G::log2m(P), M = 1 << LM;
    constexpr int NG = G::N / R;
    for (int g = tid; g < NG; g += G::THREADS) {
        const int blk = g >> LM, n2 = g & (M - 1);
        const int base = (blk << (LM + LR)) + n2;
        float2 v[R];
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = A[base + (i << LM)];
        if constexpr (DIT && M > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = ctw<DIR>(v[k], tw[(n2 * k) << (4 * P)]);
        }
        dft<R, DIR>(v);
        if constexpr (!DIT && M > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = ctw<DIR>(v[k], tw[(n2 * k) << (4 * P)]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) A[base + (i << LM)] = v[i];
    }
}

// all DIF passes [P0, P1) with barriers in between (caller syncs before the first pass)
template <int LOG2N, int DIR, int P0, int P1>
__device__ __forceinline__ void fft_dif_range(float2 *A, const float2 *__restrict__ tw, int tid)
{
    if constexpr (P0 < P1) {
        fft_pass<LOG2N, P0, DIR, false>(A, tw, tid);
        __syncthreads();
        fft_dif_range<LOG2N, DIR, P0 + 1, P1>(A, tw, tid);
    }
}

// DIT passes P1-1 down to P0 with a barrier after each
template <int LOG2N, int DIR, int P0, int P1>
__device__ __forceinline__ void fft_dit_range(float2 *A, const float2 *__restrict__ tw, int tid)
{
    if constexpr (P0 < P1) {
        fft_pass<LOG2N, P1 - 1, DIR, true>(A, tw, tid);
        __syncthreads();
        fft_dit_range<LOG2N, DIR, P0, P1 - 1>(A, tw, tid);
    }
}

} // namespace crsdr
