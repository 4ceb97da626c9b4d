// xcorr14_scalar.hpp -- the scalar-fp32 formulation of the B = 16384 kernels (r01): the same 32 x 32 x 16 network, LDS image and
// twiddle tree as csrc/xcorr14p.hpp on float2 values and scalar fp32 instructions.  NOT part of libcrsdr.so (r03): the packed
// kernels have since folded the inverse passes' input twiddles into their first radix-2 stage, so this twin agrees with them to
// rounding (lags identical, mag to 1e-5) and no longer to the bit.  Kept as the readable statement of the network; include it
// after csrc/xcorr14.hpp in a diagnostic build if needed.
#pragma once
#include "../coherent-rtlsdr_amd/csrc/xcorr14.hpp"

namespace crsdr {
namespace x14 {

// a * W_32^K  (forward exp(-2 pi i K / 32); DIR > 0: conjugate), K in [0,16)
template <int DIR, int K>
__device__ __forceinline__ float2 mul_w32(float2 a)
{
    if constexpr (K == 0) return a;
    else if constexpr (K == 8) return mul_j<DIR>(a);
    else if constexpr (K == 4) return mul_w16<DIR, 2>(a);
    else if constexpr (K == 12) return mul_w16<DIR, 6>(a);
    else {
        // sin(pi K/16) = cos(pi |8-K| / 16) for K in (0,16)
        constexpr float c = kCos16[K], sn = kCos16[K < 8 ? 8 - K : K - 8];
        return DIR < 0 ? make_float2(fmaf(a.x, c, a.y * sn), fmaf(a.y, c, -a.x * sn))
                       : make_float2(fmaf(a.x, c, -a.y * sn), fmaf(a.y, c, a.x * sn));
    }
}

template <int DIR>
__device__ __forceinline__ void dft16p(float2 *v)
{
    dft16<DIR>(*reinterpret_cast<float2(*)[16]>(v));
}

template <int DIR, int I>
__device__ __forceinline__ void dft32_stage1(float2 *v)
{
    if constexpr (I < 16) {
        float2 t = csub(v[I], v[I + 16]);
        v[I] = cadd(v[I], v[I + 16]);
        v[I + 16] = mul_w32<DIR, I>(t);
        dft32_stage1<DIR, I + 1>(v);
    }
}

// 32-point DFT, natural-order input v[0..32); output X[k] is left in v[(k & 1) * 16 + (k >> 1)]
template <int DIR>
__device__ __forceinline__ void dft32(float2 *v)
{
    dft32_stage1<DIR, 0>(v);
    dft16p<DIR>(v);
    dft16p<DIR>(v + 16);
}

// pruned first stage for the zero-padded rows: v[0..16) holds the 16 non-zero inputs.
// signal rows (zeros in the upper half):  a_i = x_i,      b_i =  x_i W^i
// ref row     (zeros in the lower half):  a_i = x_{i+16}, b_i = -x_{i+16} W^i
template <int I, bool IS_REF>
__device__ __forceinline__ void dft32_stage1_pruned(float2 *v)
{
    if constexpr (I < 16) {
        float2 b = mul_w32<-1, I>(v[I]);
        v[I + 16] = IS_REF ? make_float2(-b.x, -b.y) : b;
        dft32_stage1_pruned<I + 1, IS_REF>(v);
    }
}

// twiddles w[k] = w1^k, k in [1,32), from the five table values (product depth <= 4)
template <int K>
__device__ __forceinline__ void tw_chain(float2 *w)
{
    if constexpr (K < 32) {
        constexpr int hb = (K >= 16) ? 16 : (K >= 8) ? 8 : (K >= 4) ? 4 : (K >= 2) ? 2 : 1;
        if constexpr (K != hb) w[K] = cmul(w[K - hb], w[hb]);
        tw_chain<K + 1>(w);
    }
}
__device__ __forceinline__ void tw_load(float2 *w, const float2 *__restrict__ tab, int stride, int idx)
{
    w[1] = tab[idx];
    w[2] = tab[stride + idx];
    w[4] = tab[2 * stride + idx];
    w[8] = tab[3 * stride + idx];
    w[16] = tab[4 * stride + idx];
    tw_chain<3>(w);
}
template <int DIR, bool DIF_LAYOUT, int K>
__device__ __forceinline__ void tw_apply(float2 *v, const float2 *w)
{
    if constexpr (K < 32) {
        // DIF: the value for output k sits at xpos(k); DIT: input k sits at k
        constexpr int pos = DIF_LAYOUT ? xpos(K) : K;
        v[pos] = ctw<DIR>(v[pos], w[K]);
        tw_apply<DIR, DIF_LAYOUT, K + 1>(v, w);
    }
}

template <bool IS_REF>
__device__ __forceinline__ void pass0_forward(float2 *A, const int8_t *__restrict__ row, const float2 *__restrict__ twA,
                                              uint32_t xor80, int tid)
{
    float2 v[32];
    const uint16_t *src = reinterpret_cast<const uint16_t *>(row);
    const uint32_t x16 = xor80 & 0xFFFFu;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // integer-valued floats: the 1/127 of cdsp::convtofloat (src/cdsp.cc:41-44) is a common
        // factor of the whole linear chain and is applied once, to the peak (kInvScale4 below)
        const uint32_t u = (uint32_t)src[i * 512 + tid] ^ x16;
        v[i] = make_float2((float)sext8(u, 0), (float)sext8(u, 1));
    }
    dft32_stage1_pruned<0, IS_REF>(v);
    dft16p<-1>(v);
    dft16p<-1>(v + 16);
    float2 w[32];
    tw_load(w, twA, TWA_STRIDE, tid);
    tw_apply<-1, true, 1>(v, w);
    const int base = p0_base(tid);
#pragma unroll
    for (int k = 0; k < 32; ++k) A[base + k * 528] = v[xpos(k)];
}

__device__ __forceinline__ void pass1_forward(float2 *A, const float2 *w, int tid)
{
    const int blk = tid >> 4, n2 = tid & 15;
    float2 *Ab = A + blk * 528;
    float2 v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = Ab[p1_off(i) + (n2 ^ p1_swz(i))];
    dft32<-1>(v);
    tw_apply<-1, true, 1>(v, w);
#pragma unroll
    for (int k = 0; k < 32; ++k) Ab[p1_off(k) + (n2 ^ p1_swz(k))] = v[xpos(k)];
}

__device__ __forceinline__ void pass1_inverse(float2 *A, const float2 *w, int tid)
{
    const int blk = tid >> 4, n2 = tid & 15;
    float2 *Ab = A + blk * 528;
    float2 v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = Ab[p1_off(k) + (n2 ^ p1_swz(k))];
    tw_apply<+1, false, 1>(v, w);
    dft32<+1>(v);
#pragma unroll
    for (int i = 0; i < 32; ++i) Ab[p1_off(i) + (n2 ^ p1_swz(i))] = v[xpos(i)];
}

// ---- K0: reference spectrum, conj, [slot][group] layout ------------------------------------------
__global__ __launch_bounds__(THREADS, 2) void k_ref_spectrum14(const int8_t *__restrict__ rows, size_t block_stride,
                                                               const float2 *__restrict__ twA,
                                                               const float2 *__restrict__ twB,
                                                               float4 *__restrict__ refspec_base, uint32_t xor80)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *A = reinterpret_cast<float2 *>(smem);
    const float4 *A4 = reinterpret_cast<const float4 *>(smem);
    const int tid = threadIdx.x;
    const int8_t *ref_row = rows + (size_t)blockIdx.x * block_stride; // row 0 of batch block blockIdx.x
    float4 *refspec4 = refspec_base + (size_t)blockIdx.x * (N / 2);
    pass0_forward<true>(A, ref_row, twA, xor80, tid);
    float2 wB[32];
    tw_load(wB, twB, TWB_STRIDE, tid & 15);
    __syncthreads();
    pass1_forward(A, wB, tid);
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // the 128 J groups of the 4 sub-blocks this wave produced in P1
        const int g = ((tid >> 6) << 7) + 64 * h + (tid & 63), base = j_base(g), key = g & 7;
        float2 u[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 q = A4[base + (j ^ key)];
            u[2 * j] = make_float2(q.x, q.y);
            u[2 * j + 1] = make_float2(q.z, q.w);
        }
        dft16p<-1>(u);
#pragma unroll
        for (int j = 0; j < 8; ++j) // conj(sfft[0]) for the conjugate multiply of src/ccoherent.cc:177-179
            refspec4[j * 1024 + g] = make_float4(u[2 * j].x, -u[2 * j].y, u[2 * j + 1].x, -u[2 * j + 1].y);
    }
}

// ---- K1 -------------------------------------------------------------------------------------------
// One row of K1; src = the int8 row.
__device__ __forceinline__ void xcorr_row14(const XcorrArgs &a, unsigned char *smem, const int8_t *__restrict__ src, int row, int t,
                                            const float2 *__restrict__ twA, const float2 *__restrict__ twB)
{
    float2 *A = reinterpret_cast<float2 *>(smem);
    float4 *A4 = reinterpret_cast<float4 *>(smem);
    float *red = reinterpret_cast<float *>(smem + (size_t)LDS_ELEMS * 8); // 128 floats of scratch
    const int tid = threadIdx.x;
    const float4 *__restrict__ refspec4 = reinterpret_cast<const float4 *>(a.refspec) + (size_t)t * (N / 2);

    CRSDR_STAMP(0);
    pass0_forward<false>(A, src, twA, a.xor80, tid);
    CRSDR_STAMP(1);
    // P1 / P1' twiddles: one chain per row, computed while the P0 stores drain, alive across J
    float2 wB[32];
    tw_load(wB, twB, TWB_STRIDE, tid & 15);
    __syncthreads();
    CRSDR_STAMP(2);
    // stagger: the two waves of a SIMD run the same program and would hit their LDS bursts and VALU
    // stretches together; holding waves 4..7 back by ~500 cycles at the start of the wave-local section
    // lets one wave's exchange overlap the other's butterflies (measured: -2..3 % K1 time at stagger = 1,
    // +4 % at 8; MI355X_MICROARCH.md "Two waves per SIMD", item 9)
    if (a.stagger > 0 && (tid >> 8))
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(8);
    pass1_forward(A, wB, tid);
    wave_lds_sync();
    CRSDR_STAMP(3);
    // junction: DFT16 . conj(ref spectrum) . IDFT16 on the same 16 contiguous points
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // the 128 J groups of the 4 sub-blocks this wave produced in P1 (and consumes in P1')
        const int g = ((tid >> 6) << 7) + 64 * h + (tid & 63), base = j_base(g), key = g & 7;
        float4 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = refspec4[j * 1024 + g];
        float2 u[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 q = A4[base + (j ^ key)];
            u[2 * j] = make_float2(q.x, q.y);
            u[2 * j + 1] = make_float2(q.z, q.w);
        }
        dft16p<-1>(u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u[2 * j] = cmul(u[2 * j], make_float2(r[j].x, r[j].y));
            u[2 * j + 1] = cmul(u[2 * j + 1], make_float2(r[j].z, r[j].w));
        }
        dft16p<+1>(u);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            A4[base + (j ^ key)] = make_float4(u[2 * j].x, u[2 * j].y, u[2 * j + 1].x, u[2 * j + 1].y);
    }
    wave_lds_sync();
    CRSDR_STAMP(4);
    pass1_inverse(A, wB, tid);
    CRSDR_STAMP(5);
    __syncthreads();
    CRSDR_STAMP(6);
    // final inverse pass fused with |.|^2 (cdsp::magsquared) and the argmax (cdsp::indexofmax)
    float m[32];
    {
        float2 v[32];
        const int base = p0_base(tid);
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = A[base + k * 528];
        float2 w[32];
        tw_load(w, twA, TWA_STRIDE, tid);
        tw_apply<+1, false, 1>(v, w);
        dft32<+1>(v);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float2 x = v[xpos(i)];
            m[i] = fmaf(x.x, x.x, x.y * x.y);
        }
    }
    // maximum VALUE first: 16 x v_max3 per thread, a 64-lane butterfly, 8 waves through LDS ...
    float tm = m[0];
#pragma unroll
    for (int i = 1; i < 31; i += 2) tm = fmaxf(tm, fmaxf(m[i], m[i + 1]));
    tm = fmaxf(tm, m[31]);
    CRSDR_STAMP(7);
    float wm = tm;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
    int *redi = reinterpret_cast<int *>(red);
    if (tid == 0) redi[16] = 0x7fffffff;
    if ((tid & 63) == 0) red[tid >> 6] = wm;
    __syncthreads();
    float gm = red[0];
#pragma unroll
    for (int wv = 1; wv < THREADS / 64; ++wv) gm = fmaxf(gm, red[wv]);
    // ... then its FIRST index (volk_32f_index_max_32u keeps the first strict maximum): only the
    // threads that hold the maximum search their 32 values (natural index of output i: i*512 + tid)
    if (tm == gm) {
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 31; i >= 0; --i) bi = (m[i] == gm) ? i * 512 + tid : bi;
        atomicMin(&redi[16], bi);
    }
    __syncthreads();
    int gi = redi[16];
    if ((unsigned)gi >= (unsigned)N) gi = 0; // all-NaN row: defined as index 0
    // neighbours of the peak for the parabolic estimate: their owners publish them
    {
        const int nl = gi - 1, nr = gi + 1;
        if (gi > 0 && (nl & 511) == tid) {
            float ml = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) ml = (i == (nl >> 9)) ? m[i] : ml;
            red[32] = ml;
        }
        if (gi < N - 1 && (nr & 511) == tid) {
            float mr = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) mr = (i == (nr >> 9)) ? m[i] : mr;
            red[33] = mr;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float D = 0.0f;
        if (gi > 0 && gi < N - 1) {
            const float ym = red[32], yp = red[33];
            const float den = (ym - 2.0f * gm) + yp;
            if (den != 0.0f) D = (0.5f * (ym - yp)) / den;
        }
        xcorr_publish(a, row, t, gi - L /* src/ccoherent.cc:232 */, sqrtf(gm / (float)L) * kInvScale2 /* :204 */, D);
    }
    CRSDR_STAMP(8);
}

__global__ __launch_bounds__(THREADS, 2) void k_xcorr_lag14(XcorrArgs a, const float2 *__restrict__ twA,
                                                            const float2 *__restrict__ twB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int row = a.row_begin + blockIdx.x, t = blockIdx.y;
    if (xcorr_skip(a, row, t, threadIdx.x)) return;
    xcorr_row14(a, smem, a.rows + (size_t)t * a.block_stride + (size_t)row * N, row, t, twA, twB);
}

// ---- 16384-point row transforms on cf32 lines (stage B of the long-block path, longblock.hpp) ---------
// Same three passes as K0 / K1 with a full (un-pruned) first radix-32 stage, cf32 in and out:
//   IS_REF: forward, conj -> refspec [slot][group];   else: forward, x conj(ref), inverse, in place.
// grid (N1 lines, rows); line l of row r at Y + (r * gridDim.x + l) * 16384.
template <bool IS_REF>
__global__ __launch_bounds__(THREADS, 2) void k_rows14_cf32(float2 *__restrict__ Y, const float2 *__restrict__ twA,
                                                             const float2 *__restrict__ twB, float4 *__restrict__ refspec_base)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *A = reinterpret_cast<float2 *>(smem);
    float4 *A4 = reinterpret_cast<float4 *>(smem);
    const int tid = threadIdx.x;
    float2 *line = Y + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * N;
    float4 *refspec4 = refspec_base + (size_t)blockIdx.x * (N / 2);
    {
        float2 v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = line[i * 512 + tid];
        dft32<-1>(v);
        float2 w[32];
        tw_load(w, twA, TWA_STRIDE, tid);
        tw_apply<-1, true, 1>(v, w);
        const int base = p0_base(tid);
#pragma unroll
        for (int k = 0; k < 32; ++k) A[base + k * 528] = v[xpos(k)];
    }
    float2 wB[32];
    tw_load(wB, twB, TWB_STRIDE, tid & 15);
    __syncthreads();
    pass1_forward(A, wB, tid);
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int g = ((tid >> 6) << 7) + 64 * h + (tid & 63), base = j_base(g), key = g & 7;
        float4 r[8];
        if constexpr (!IS_REF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = refspec4[j * 1024 + g];
        }
        float2 u[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 q = A4[base + (j ^ key)];
            u[2 * j] = make_float2(q.x, q.y);
            u[2 * j + 1] = make_float2(q.z, q.w);
        }
        dft16p<-1>(u);
        if constexpr (IS_REF) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                refspec4[j * 1024 + g] = make_float4(u[2 * j].x, -u[2 * j].y, u[2 * j + 1].x, -u[2 * j + 1].y);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u[2 * j] = cmul(u[2 * j], make_float2(r[j].x, r[j].y));
                u[2 * j + 1] = cmul(u[2 * j + 1], make_float2(r[j].z, r[j].w));
            }
            dft16p<+1>(u);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                A4[base + (j ^ key)] = make_float4(u[2 * j].x, u[2 * j].y, u[2 * j + 1].x, u[2 * j + 1].y);
        }
    }
    if constexpr (!IS_REF) {
        wave_lds_sync();
        pass1_inverse(A, wB, tid);
        __syncthreads();
        float2 v[32];
        const int base = p0_base(tid);
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = A[base + k * 528];
        float2 w[32];
        tw_load(w, twA, TWA_STRIDE, tid);
        tw_apply<+1, false, 1>(v, w);
        dft32<+1>(v);
#pragma unroll
        for (int i = 0; i < 32; ++i) line[i * 512 + tid] = v[xpos(i)]; // natural order, coalesced
    }
}

} // namespace x14
} // namespace crsdr
