// (experiment, not part of libcrsdr.so any more: kept for the record of DESIGN.md dead end (8); it built against the csrc/ headers
// of round 1 as K1 variant "half" and measured 15 % slower than the packed kernel)
// xcorr14h.hpp -- K1 for B = 16384 with HALF the row in LDS: two workgroups (two rows) per CU.
//
// Why: with the whole 128 KiB row in LDS (xcorr14.hpp) one workgroup fills the CU, both waves of a SIMD belong to
// the same row and stall together (LDS bursts, barriers, HBM/L2 latency): the SIMDs issue a VALU instruction
// every ~2.1 ns where ~1.0-1.2 ns is possible (tools/valu_rate.hip, tools/k1_corun.hip: a register-light VALU
// spinner beside K1 raises the combined issue rate by 45 %).  Here a row needs 66 KiB, so two rows share a CU and
// one row's exchanges and waits fall under the other row's butterflies.
//
// Same 32 x 32 x 16 factorisation, same LDS plane layout, same arithmetic per point as xcorr14.hpp; what
// changes is the order.  The first radix-32 pass splits by output parity,
//     X[2j]   = DFT16(x)[j]            X[2j+1] = DFT16(x . W32^i)[j]              (decimation in frequency)
// and the last one by input parity,
//     y[n], y[n+16] = E[n] +- W32^(+n) O[n],   E = IDFT16(Y[even]), O = IDFT16(Y[odd])   (decimation in time)
// so the EVEN planes (k = 0, 2, .. 30 of the stride-512 / stride-16 transpose) can make the whole round trip
//     P0(even) -> LDS -> P1 -> J -> P1' -> LDS -> E
// before the ODD planes follow through the same 16 plane slots.  256 threads, each the two "virtual threads"
// t and t + 256 of xcorr14.hpp in the column passes (P0 / P0') and the owner of plane 2s (first trip) and
// 2s + 1 (second trip), s = tid >> 4, in the plane passes (P1 / J / P1').  The int8 inputs stay packed in 16
// registers between the trips; E (2 x 16 complex) is the only state carried across the second trip.
// Five workgroup barriers per row (4 waves each) instead of two (8 waves), no extra LDS traffic, no extra flops.
//
// STATUS (r01): opt-in (CRSDR_K1_VARIANT=half), parity-green, 15 % SLOWER than xcorr14.hpp (1.07 vs 0.93 ms per
// 16-block launch).  Two rows per CU at 256 threads each is still two waves per SIMD, each with the same ~50 %
// issue duty as before -- the slack only fills with MORE waves per SIMD, i.e. <= 128 VGPRs per thread, and this
// kernel sits at 256 with 27 spills (E + packed inputs + the radix-32 working set), which also stops the
// compiler from hoisting the twiddle / reference-spectrum loads.  Kept as the starting point for a register-lean
// (<= 128 VGPR, 512-thread) version: the parity-trip decomposition is what makes a 66 KiB row possible at all.
#pragma once
#include "xcorr14.hpp"

namespace crsdr {
namespace x14h {

using namespace x14;
constexpr int THREADS_H = 256;
constexpr int LDS_ELEMS_H = 16 * 528;              // 16 plane slots of 512 elements + 16 pad
constexpr int LDS_BYTES_H = LDS_ELEMS_H * 8 + 512; // + reduction scratch

// twiddles for the even (PAR = 0: w1^(2j)) or odd (PAR = 1: w1^(2j+1)) outputs, j in [0,16), from the five
// table values -- the same product tree as x14::tw_chain, so the values are bit-identical to xcorr14.hpp
template <int PAR, int K>
__device__ __forceinline__ void tw_half_chain(float2 *w)
{
    if constexpr (K < 32) {
        constexpr int hb = (K >= 16) ? 16 : (K >= 8) ? 8 : (K >= 4) ? 4 : (K >= 2) ? 2 : 1;
        if constexpr ((K & 1) == PAR && K != hb) w[K] = cmul(w[K - hb], w[hb]);
        tw_half_chain<PAR, K + 1>(w);
    }
}
// w[] is indexed by k like in xcorr14.hpp; only the entries of parity PAR (and the loaded powers of two) are valid
template <int PAR>
__device__ __forceinline__ void tw_half_load(float2 *w, const float2 *__restrict__ tab, int idx)
{
    if constexpr (PAR == 1) w[1] = tab[idx];
    w[2] = tab[TWA_STRIDE + idx];
    w[4] = tab[2 * TWA_STRIDE + idx];
    w[8] = tab[3 * TWA_STRIDE + idx];
    w[16] = tab[4 * TWA_STRIDE + idx];
    tw_half_chain<PAR, 3>(w);
}

__device__ __forceinline__ void unpack16(float2 *v, const uint32_t *p)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[2 * i] = make_float2((float)sext8(p[i], 0), (float)sext8(p[i], 1));
        v[2 * i + 1] = make_float2((float)sext8(p[i], 2), (float)sext8(p[i], 3));
    }
}

template <int I>
__device__ __forceinline__ void rot_w32(float2 *v)
{
    if constexpr (I < 16) {
        v[I] = mul_w32<-1, I>(v[I]);
        rot_w32<I + 1>(v);
    }
}

// forward column pass for the planes of parity PAR of column c: 16 outputs -> plane slots 0..15
template <int PAR>
__device__ __forceinline__ void p0_half_forward(float2 *A, const uint32_t *packed, const float2 *__restrict__ twA, int c)
{
    float2 v[16];
    unpack16(v, packed);
    if constexpr (PAR == 1) rot_w32<0>(v);
    dft16p<-1>(v);
    float2 w[32];
    tw_half_load<PAR>(w, twA, c);
    const int base = p0_base(c);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = 2 * j + PAR;
        const float2 x = (k == 0) ? v[j] : ctw<-1>(v[j], w[k]);
        A[base + j * 528] = x;
    }
}

// inverse column pass, first half: the planes of parity PAR of column c -> IDFT16 of the twiddled inputs
template <int PAR>
__device__ __forceinline__ void p0_half_inverse(float2 *out, const float2 *A, const float2 *__restrict__ twA, int c)
{
    const int base = p0_base(c);
#pragma unroll
    for (int j = 0; j < 16; ++j) out[j] = A[base + j * 528];
    float2 w[32];
    tw_half_load<PAR>(w, twA, c);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = 2 * j + PAR;
        if (k != 0) out[j] = ctw<+1>(out[j], w[k]);
    }
    dft16p<+1>(out);
}

template <int I>
__device__ __forceinline__ void dit_combine(float *m, const float2 *E, const float2 *O)
{
    if constexpr (I < 16) {
        const float2 t = mul_w32<+1, I>(O[I]);
        const float2 p = cadd(E[I], t), q = csub(E[I], t);
        m[I] = fmaf(p.x, p.x, p.y * p.y);           // cdsp::magsquared (src/cdsp.cc:100-103)
        m[I + 16] = fmaf(q.x, q.x, q.y * q.y);
        dit_combine<I + 1>(m, E, O);
    }
}

// the plane trip: P1 -> J -> P1' on the 16 plane slots; this thread owns (slot tid >> 4, n2 = tid & 15) in the
// radix-32 passes and 2 x 64 J groups of its wave's 4 slots; the slots hold the planes 2 s + PAR
template <int PAR>
__device__ __forceinline__ void plane_trip(float2 *A, const float4 *__restrict__ refspec4, const float2 *__restrict__ twB, int tid)
{
    float4 *A4 = reinterpret_cast<float4 *>(A);
    float2 wB[32];
    tw_load(wB, twB, TWB_STRIDE, tid & 15);
    pass1_forward(A, wB, tid);
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int gs = ((tid >> 6) << 7) + 64 * h + (tid & 63);          // group index in slot space
        const int g = (((gs >> 5) * 2 + PAR) << 5) + (gs & 31);          // ... in the row: plane 2 s + PAR
        const int base = j_base(gs), key = gs & 7;
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
    pass1_inverse(A, wB, tid);
}

__global__ __launch_bounds__(THREADS_H, 2) void k_xcorr_lag14h(XcorrArgs a, const float2 *__restrict__ twA,
                                                                const float2 *__restrict__ twB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *A = reinterpret_cast<float2 *>(smem);
    float *red = reinterpret_cast<float *>(smem + (size_t)LDS_ELEMS_H * 8);
    const int tid = threadIdx.x;
    const int row = a.row_begin + blockIdx.x, t = blockIdx.y;
    if (xcorr_skip(a, row, t, tid)) return;
    const float4 *__restrict__ refspec4 = reinterpret_cast<const float4 *>(a.refspec) + (size_t)t * (N / 2);
    const int c0 = tid, c1 = tid + 256;

    // the 16 non-zero inputs of each column stay packed (4 int8 = 2 samples per register) across both trips
    uint32_t pk0[8], pk1[8];
    {
        const uint16_t *src = reinterpret_cast<const uint16_t *>(a.rows + (size_t)t * a.block_stride + (size_t)row * N);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pk0[i] = (((uint32_t)src[(2 * i) * 512 + c0]) | ((uint32_t)src[(2 * i + 1) * 512 + c0] << 16)) ^ a.xor80;
            pk1[i] = (((uint32_t)src[(2 * i) * 512 + c1]) | ((uint32_t)src[(2 * i + 1) * 512 + c1] << 16)) ^ a.xor80;
        }
    }

    // ---- first trip: even planes ---------------------------------------------------------------------
    p0_half_forward<0>(A, pk0, twA, c0);
    p0_half_forward<0>(A, pk1, twA, c1);
    __syncthreads();
    plane_trip<0>(A, refspec4, twB, tid);
    __syncthreads();
    float2 E0[16], E1[16];
    p0_half_inverse<0>(E0, A, twA, c0);
    p0_half_inverse<0>(E1, A, twA, c1);
    __syncthreads();
    // ---- second trip: odd planes ----------------------------------------------------------------------
    p0_half_forward<1>(A, pk0, twA, c0);
    p0_half_forward<1>(A, pk1, twA, c1);
    __syncthreads();
    plane_trip<1>(A, refspec4, twB, tid);
    __syncthreads();
    float m0[32], m1[32];
    {
        float2 O[16];
        p0_half_inverse<1>(O, A, twA, c0);
        dit_combine<0>(m0, E0, O);
        p0_half_inverse<1>(O, A, twA, c1);
        dit_combine<0>(m1, E1, O);
    }

    // ---- maximum value first, then its first index (volk_32f_index_max_32u: first strict maximum) -------
    float tm = fmaxf(m0[0], m1[0]);
#pragma unroll
    for (int i = 1; i < 32; ++i) tm = fmaxf(tm, fmaxf(m0[i], m1[i]));
    float wm = tm;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
    int *redi = reinterpret_cast<int *>(red);
    if (tid == 0) redi[16] = 0x7fffffff;
    if ((tid & 63) == 0) red[tid >> 6] = wm;
    __syncthreads();
    float gm = red[0];
#pragma unroll
    for (int wv = 1; wv < THREADS_H / 64; ++wv) gm = fmaxf(gm, red[wv]);
    if (tm == gm) {
        int bi = 0x7fffffff;                       // natural index of output i of column c: i*512 + c
#pragma unroll
        for (int i = 31; i >= 0; --i) {
            bi = (m1[i] == gm) ? i * 512 + c1 : bi;
            bi = (m0[i] == gm) ? i * 512 + c0 : bi;
        }
        atomicMin(&redi[16], bi);
    }
    __syncthreads();
    int gi = redi[16];
    if ((unsigned)gi >= (unsigned)N) gi = 0;      // all-NaN row: defined as index 0
    {
        const int nl = gi - 1, nr = gi + 1;
        if (gi > 0 && (nl & 255) == tid) {
            float ml = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) ml = (i == (nl >> 9)) ? (((nl >> 8) & 1) ? m1[i] : m0[i]) : ml;
            red[32] = ml;
        }
        if (gi < N - 1 && (nr & 255) == tid) {
            float mr = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) mr = (i == (nr >> 9)) ? (((nr >> 8) & 1) ? m1[i] : m0[i]) : mr;
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
}

} // namespace x14h
} // namespace crsdr
