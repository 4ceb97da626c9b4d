// xcorr14p.hpp -- K1 for B = 16384 on packed complex arithmetic (cpk.hpp): the same 32 x 32 x 16 network, LDS
// image, twiddle tree and operation order as xcorr14.hpp (every product and sum rounds identically), expressed
// on (re, im) register pairs so that a complex add is one v_pk_add_f32 and a complex product one v_pk_mul_f32 +
// one v_pk_fma_f32.  Generated from xcorr14.hpp's K1 path by type substitution; see cpk.hpp for the rationale.
#pragma once
#include "cpk.hpp"
#include "xcorr14.hpp"

namespace crsdr {
namespace x14p {

using x14::N; using x14::L; using x14::THREADS; using x14::LDS_ELEMS; using x14::LDS_BYTES; using x14::TWA_STRIDE; using x14::TWB_STRIDE;
using x14::kInvScale2; using x14::kCos16; using x14::xpos; using x14::p0_base; using x14::p1_off; using x14::p1_swz; using x14::j_base;
using x14::wave_lds_sync;
typedef __attribute__((address_space(3))) c2 lds_c2;    // an LDS-qualified access (a volatile one through a generic pointer would be a flat load)

// a * W_32^K  (forward exp(-2 pi i K / 32); DIR > 0: conjugate), K in [0,16)
template <int DIR, int K>
__device__ __forceinline__ c2 mul_w32(c2 a)
{
    if constexpr (K == 0) return a;
    else if constexpr (K == 8) return mul_j<DIR>(a);
    else if constexpr (K == 4) return mul_w16<DIR, 2>(a);
    else if constexpr (K == 12) return mul_w16<DIR, 6>(a);
    else {
        // sin(pi K/16) = cos(pi |8-K| / 16) for K in (0,16)
        constexpr float c = kCos16[K], sn = kCos16[K < 8 ? 8 - K : K - 8];
        return rot_cs<DIR>(a, c, sn);
    }
}

template <int DIR>
__device__ __forceinline__ void dft16p(c2 *v)
{
    dft16<DIR>(v);
}

template <int DIR, int I>
__device__ __forceinline__ void dft32_stage1(c2 *v)
{
    if constexpr (I < 16) {
        c2 t = csub(v[I], v[I + 16]);
        v[I] = cadd(v[I], v[I + 16]);
        v[I + 16] = mul_w32<DIR, I>(t);
        dft32_stage1<DIR, I + 1>(v);
    }
}

// 32-point DFT, natural-order input v[0..32); output X[k] is left in v[(k & 1) * 16 + (k >> 1)]
template <int DIR>
__device__ __forceinline__ void dft32(c2 *v)
{
    dft32_stage1<DIR, 0>(v);
    dft16p<DIR>(v);
    dft16p<DIR>(v + 16);
}

// pruned first stage for the zero-padded rows: v[0..16) holds the 16 non-zero inputs.
// signal rows (zeros in the upper half):  a_i = x_i,      b_i =  x_i W^i
// ref row     (zeros in the lower half):  a_i = x_{i+16}, b_i = -x_{i+16} W^i
template <int I, bool IS_REF>
__device__ __forceinline__ void dft32_stage1_pruned(c2 *v)
{
    if constexpr (I < 16) {
        c2 b = mul_w32<-1, I>(v[I]);
        v[I + 16] = IS_REF ? -b : b;
        dft32_stage1_pruned<I + 1, IS_REF>(v);
    }
}

// twiddles w[k] = w1^k, k in [1,32), from the five table values (product depth <= 4)
template <int K>
__device__ __forceinline__ void tw_chain(c2 *w)
{
    if constexpr (K < 32) {
        constexpr int hb = (K >= 16) ? 16 : (K >= 8) ? 8 : (K >= 4) ? 4 : (K >= 2) ? 2 : 1;
        if constexpr (K != hb) w[K] = cmul(w[K - hb], w[hb]);
        tw_chain<K + 1>(w);
    }
}
__device__ __forceinline__ void tw_load(c2 *w, const c2 *__restrict__ tab, int stride, int idx)
{
    w[1] = tab[idx];
    w[2] = tab[stride + idx];
    w[4] = tab[2 * stride + idx];
    w[8] = tab[3 * stride + idx];
    w[16] = tab[4 * stride + idx];
    tw_chain<3>(w);
}
template <int DIR, bool DIF_LAYOUT, int K>
__device__ __forceinline__ void tw_apply(c2 *v, const c2 *w)
{
    if constexpr (K < 32) {
        // DIF: the value for output k sits at xpos(k); DIT: input k sits at k
        constexpr int pos = DIF_LAYOUT ? xpos(K) : K;
        v[pos] = ctw<DIR>(v[pos], w[K]);
        tw_apply<DIR, DIF_LAYOUT, K + 1>(v, w);
    }
}

// Inverse-pass head: the input twiddles (conjugated) folded into the first radix-2 stage of the 32-point transform.
//   separate:  a' = a conj(wa), b' = b conj(wb)  [2 + 2],  s = a' + b', t = a' - b'  [1 + 1]          = 6 packed instructions
//   fused:     p = a conj(wa) [2],  s = p + b conj(wb)  [2 fused multiply-adds],  t = 2 p - s  [1]        = 5
// (index 0 carries no twiddle: 3 against 4).  16 fewer per pass; s rounds once less, t inherits s's rounding.
#ifndef CRSDR_K1_FUSED_TW
#define CRSDR_K1_FUSED_TW 1
#endif
template <int I>
__device__ __forceinline__ void tw_stage1_inv(c2 *v, const c2 *w)
{
    if constexpr (I < 16) {
        const c2 p = (I == 0) ? v[0] : cmulc(v[I], w[I]);
        const c2 s = cmulc_add(p, v[I + 16], w[I + 16]);
        const c2 t = twice_minus(p, s);
        v[I] = s;
        v[I + 16] = mul_w32<+1, I>(t);
        tw_stage1_inv<I + 1>(v, w);
    }
}
// v[k] = input k (natural order) -> inverse 32-point DFT of (v[k] conj(w[k])), outputs as dft32 leaves them (xpos)
__device__ __forceinline__ void tw_dft32_inv(c2 *v, const c2 *w)
{
#if CRSDR_K1_FUSED_TW
    tw_stage1_inv<0>(v, w);
    dft16p<+1>(v);
    dft16p<+1>(v + 16);
#else
    tw_apply<+1, false, 1>(v, w);
    dft32<+1>(v);
#endif
}

template <bool IS_REF>
__device__ __forceinline__ void pass0_forward(c2 *A, const int8_t *__restrict__ row, const c2 *__restrict__ twA,
                                              uint32_t xor80, int tid, c2 *w)
{
    c2 v[32];
    const uint16_t *src = reinterpret_cast<const uint16_t *>(row);
    const uint32_t x16 = xor80 & 0xFFFFu;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        // integer-valued floats: the 1/127 of cdsp::convtofloat (src/cdsp.cc:41-44) is a common
        // factor of the whole linear chain and is applied once, to the peak (kInvScale4 below)
        const uint32_t u = (uint32_t)src[i * 512 + tid] ^ x16;
        v[i] = mk((float)sext8(u, 0), (float)sext8(u, 1));
    }
    dft32_stage1_pruned<0, IS_REF>(v);
    dft16p<-1>(v);
    dft16p<-1>(v + 16);
    tw_load(w, twA, TWA_STRIDE, tid);           // w: the caller's array -- K1 keeps the column twiddles for its last pass
    tw_apply<-1, true, 1>(v, w);
    const int base = p0_base(tid);
#pragma unroll
    for (int k = 0; k < 32; ++k) A[base + k * 528] = v[xpos(k)];
}

__device__ __forceinline__ void pass1_forward(c2 *A, const c2 *w, int tid)
{
    const int blk = tid >> 4, n2 = tid & 15;
    c2 *Ab = A + blk * 528;
    c2 v[32];
#pragma unroll
    // volatile: keeps these as 32 ds_read_b64 (2 LDS cycles each); merged into ds_read2_b64 they cost 8 cycles per pair
    for (int i = 0; i < 32; ++i) v[i] = *(const volatile lds_c2 *)(Ab + p1_off(i) + (n2 ^ p1_swz(i)));
    dft32<-1>(v);
    tw_apply<-1, true, 1>(v, w);
#pragma unroll
    for (int k = 0; k < 32; ++k) Ab[p1_off(k) + (n2 ^ p1_swz(k))] = v[xpos(k)];
}

__device__ __forceinline__ void pass1_inverse(c2 *A, const c2 *w, int tid)
{
    const int blk = tid >> 4, n2 = tid & 15;
    c2 *Ab = A + blk * 528;
    c2 v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = *(const volatile lds_c2 *)(Ab + p1_off(k) + (n2 ^ p1_swz(k)));   // ds_read_b64, not read2 (see pass1_forward)
    tw_dft32_inv(v, w);
#pragma unroll
    for (int i = 0; i < 32; ++i) Ab[p1_off(i) + (n2 ^ p1_swz(i))] = v[xpos(i)];
}

// One row of K1; src = the int8 row.
__device__ __forceinline__ void xcorr_row14(const XcorrArgs &a, unsigned char *smem, const int8_t *__restrict__ src, int row, int t,
                                            const c2 *__restrict__ twA, const c2 *__restrict__ twB)
{
    c2 *A = reinterpret_cast<c2 *>(smem);
    float4 *A4 = reinterpret_cast<float4 *>(smem);
    float *red = reinterpret_cast<float *>(smem + (size_t)LDS_ELEMS * 8); // 128 floats of scratch
    const int tid = threadIdx.x;
    const float4 *__restrict__ refspec4 = reinterpret_cast<const float4 *>(a.refspec) + (size_t)t * (N / 2);

    CRSDR_STAMP(0);
    // folded launches: is the block's reference spectrum there yet?  ONE wave asks (r03, every wave polling: 1900 waves on one L2
    // channel at launch start, K1 + 0.13 ms per launch whatever its size); its verdict travels through LDS at the row's first barrier
    const unsigned int ref_early = (a.fold && tid < 64) ? word_peek(a.refflag + t) : 0u;
    c2 wA[32];                                   // column twiddles: P0 applies them, P0' their conjugates -- one chain, 62 registers
    pass0_forward<false>(A, src, twA, a.xor80, tid, wA);
    CRSDR_STAMP(1);
    // P1 / P1' twiddles: one chain per row, computed while the P0 stores drain, alive across J
    c2 wB[32];
    tw_load(wB, twB, TWB_STRIDE, tid & 15);
    int *refok = reinterpret_cast<int *>(red) + 48;
    if (a.fold && tid < 64) {
        const bool ok = (unsigned int)__builtin_amdgcn_readfirstlane((int)ref_early) == a.refgen;
        if (ok) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // this CU's L1 holds nothing older than the published spectrum ...
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // ... once the (asynchronous) invalidate is through: before the barrier below lets the other waves load
        }
        if (tid == 0) *refok = ok ? 1 : 0;
    }
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
    // folded launches: the block's reference spectrum comes from a workgroup of this same launch with a LOWER index (dispatched
    // before this one: it is resident or done), published behind refflag[t]
    if (a.fold && !*refok) {                     // workgroup-uniform; only the first rows of a launch ever come here
        if (tid < 64 && !wait_word(a.refflag + t, a.refgen, ref_early, kRefWaitBudget) && tid == 0 && a.errflag) atomicAdd(a.errflag, 1);
        __syncthreads();
    }
    // junction: DFT16 . conj(ref spectrum) . IDFT16 on the same 16 contiguous points
#pragma unroll 1     // not unrolled: both halves in flight need 238 VGPRs, one at a time 193 (room for a phase-kernel wave per SIMD)
    for (int h = 0; h < 2; ++h) {
        // the 128 J groups of the 4 sub-blocks this wave produced in P1 (and consumes in P1')
        const int g = ((tid >> 6) << 7) + 64 * h + (tid & 63), base = j_base(g), key = g & 7;
        float4 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = refspec4[j * 1024 + g];
        c2 u[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 q = A4[base + (j ^ key)];
            u[2 * j] = mk(q.x, q.y);
            u[2 * j + 1] = mk(q.z, q.w);
        }
        dft16p<-1>(u);
#if CRSDR_K1_FUSED_TW
        {
            c2 rr[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) { rr[2 * j] = mk(r[j].x, r[j].y); rr[2 * j + 1] = mk(r[j].z, r[j].w); }
            dft16_inv_mul(u, rr);
        }
#else
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u[2 * j] = cmul(u[2 * j], mk(r[j].x, r[j].y));
            u[2 * j + 1] = cmul(u[2 * j + 1], mk(r[j].z, r[j].w));
        }
        dft16p<+1>(u);
#endif
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
        c2 v[32];
        const int base = p0_base(tid);
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = A[base + k * 528];
        tw_dft32_inv(v, wA);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const c2 x = v[xpos(i)];
            m[i] = fmaf(x.x, x.x, x.y * x.y);
        }
    }
    // maximum VALUE first: 16 x v_max3 per thread, a 64-lane butterfly, 8 waves through LDS ...
    float tm = m[0];
#pragma unroll
    for (int i = 1; i < 31; i += 2) tm = fmaxf(tm, fmaxf(m[i], m[i + 1]));
    tm = fmaxf(tm, m[31]);
    CRSDR_STAMP(7);
    // wave maximum in lane 63: four row_shr steps (lane 15 of each 16-lane row holds the row's maximum), then
    // row_bcast:15 / row_bcast:31 -- six DPP v_max instead of six trips through the LDS crossbar
    float wm = tm;
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x111, 0xf, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x112, 0xf, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x114, 0xf, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x118, 0xf, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x142, 0xa, 0xf, false)));
    wm = fmaxf(wm, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(wm), __float_as_int(wm), 0x143, 0xc, 0xf, false)));
    int *redi = reinterpret_cast<int *>(red);
    if (tid == 0) redi[16] = 0x7fffffff;
    if ((tid & 63) == 63) red[tid >> 6] = wm;
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
    // neighbours of the peak for the parabolic estimate.  gi = i*512 + c: its neighbours are the same output i of the
    // columns c -+ 1, i.e. the adjacent lanes of the owner's wave unless the owner sits on a wave edge -- then (1 row in
    // 32) they go through LDS and one more barrier, otherwise the owner's wave finishes the row on its own
    const int pc = gi & 511, pi_ = gi >> 9;
    const bool in_wave = ((pc & 63) != 0) && ((pc & 63) != 63);       // workgroup-uniform
    if (in_wave) {
        if ((tid >> 6) == (pc >> 6)) {
            float mine = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) mine = (i == pi_) ? m[i] : mine;
            const float ym = __shfl_up(mine, 1, 64), yp = __shfl_down(mine, 1, 64);
            if (tid == pc) {
                float D = 0.0f;
                const float den = (ym - 2.0f * gm) + yp;
                if (den != 0.0f) D = (0.5f * (ym - yp)) / den;
                xcorr_publish(a, row, t, gi - L /* src/ccoherent.cc:232 */, sqrtf(gm / (float)L) * kInvScale2 /* :204 */, D);
            }
        }
    } else {
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
    CRSDR_STAMP(8);
}


// ---- K0: reference spectrum, conj, [slot][group] layout ------------------------------------------
// 16 bytes written THROUGH the XCD's L2 (sc1): what a workgroup of the same launch on another XCD is to read needs no write-back of
// the whole L2 behind it (MI355X_MICROARCH.md, publish-large: write-through stores + vmcnt(0) + flag against plain stores + release fence)
__device__ __forceinline__ void store16_sc1(float4 *p, float4 v)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 x = {v.x, v.y, v.z, v.w};
    // the s_nop is part of the instruction as far as the compiler is concerned: a store of more than 8 bytes needs a wait state before
    // a VALU write to the registers that hold its data, hipcc's hazard recognizer pads nothing around inline asm, and whatever it
    // schedules next may well write them (seen in the two-row kernel: garbage spectra; the packed kernel's schedule happened to be safe)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}
// one block's reference row -> conj(spectrum) in the junction's register order; whole 512-thread workgroup
// SC1: the spectrum is for rows of this same launch (folded K1): write-through stores
template <bool SC1 = false>
__device__ __forceinline__ void ref_spectrum_row14(unsigned char *smem, const int8_t *__restrict__ ref_row, const c2 *__restrict__ twA, const c2 *__restrict__ twB,
                                                   float4 *__restrict__ refspec4, uint32_t xor80)
{
    c2 *A = reinterpret_cast<c2 *>(smem);
    const float4 *A4 = reinterpret_cast<const float4 *>(smem);
    const int tid = threadIdx.x;
    c2 wA0[32];
    pass0_forward<true>(A, ref_row, twA, xor80, tid, wA0);
    c2 wB[32];
    tw_load(wB, twB, TWB_STRIDE, tid & 15);
    __syncthreads();
    pass1_forward(A, wB, tid);
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // the 128 J groups of the 4 sub-blocks this wave produced in P1
        const int g = ((tid >> 6) << 7) + 64 * h + (tid & 63), base = j_base(g), key = g & 7;
        c2 u[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 q = A4[base + (j ^ key)];
            u[2 * j] = mk(q.x, q.y);
            u[2 * j + 1] = mk(q.z, q.w);
        }
        dft16p<-1>(u);
#pragma unroll
        for (int j = 0; j < 8; ++j) { // conj(sfft[0]) for the conjugate multiply of src/ccoherent.cc:177-179
            const float4 c = make_float4(u[2 * j].x, -u[2 * j].y, u[2 * j + 1].x, -u[2 * j + 1].y);
            if constexpr (SC1) store16_sc1(refspec4 + j * 1024 + g, c);
            else refspec4[j * 1024 + g] = c;
        }
    }
}
__global__ __launch_bounds__(THREADS, 2) void k_ref_spectrum14p(const int8_t *__restrict__ rows, size_t block_stride,
                                                               const c2 *__restrict__ twA,
                                                               const c2 *__restrict__ twB,
                                                               float4 *__restrict__ refspec_base, uint32_t xor80)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ref_spectrum_row14(smem, rows + (size_t)blockIdx.x * block_stride /* row 0 of batch block blockIdx.x */, twA, twB,
                       refspec_base + (size_t)blockIdx.x * (N / 2), xor80);
}

// ---- 16384-point row transforms on cf32 lines (stage B of the long-block path, longblock.hpp) ---------
// Same three passes as K0 / K1 with a full (un-pruned) first radix-32 stage, cf32 in and out:
//   IS_REF: forward, conj -> refspec [slot][group];   else: forward, x conj(ref), inverse, in place;
//   RAMP (with !IS_REF): forward, x the row's own fractional-delay response instead of conj(ref), inverse, in place --
//   the apply pass of crsdr_plan_set_frac_apply.  The response for frequency bin k = k1 + N1 k2 of row r is
//       H[k] = p_r / B * exp(+2 pi i k_s (lag_r + D_r) / B),   k_s = k for k < B/2, k - B otherwise
//   (a circular advance of the zero-padded row by lag + D samples: for |lag| <= L exactly the zero-filled shift of the
//   digital mode plus a band-limited interpolation by D; the phasor rotation and the 1/B of the unnormalised transform
//   pair ride along).  k lag / B is reduced mod 1 in integers, k_s D / B is small; v_sin / v_cos take revolutions.
//   k2 of a junction register comes from a table made once per plan by transforming a one-hot line (k2tab).
// grid (N1 lines, rows); line l of row r at Y + (r * gridDim.x + l) * 16384.
// (cos, sin)(2 pi x) for |x| <= ~1: nearest half turn taken out, then odd / even polynomials on [-pi/2, pi/2]
// (terms to theta^13 / theta^12: below 6e-8 absolute).  The hardware v_sin / v_cos are not accurate enough here: a 1e-5
// error in the response moves visibly more int8 outputs across a rounding boundary than the FFT pair's own rounding.
__device__ __forceinline__ c2 cis2pi(float x)
{
    const float q = rintf(2.0f * x);                       // half turns
    const float th = 6.28318530717958647692f * fmaf(-0.5f, q, x);
    const float t2 = th * th;
    float sn = fmaf(t2, 1.6059043836821613e-10f, -2.5052108385441720e-08f);
    sn = fmaf(t2, sn, 2.7557319223985893e-06f);
    sn = fmaf(t2, sn, -1.9841269841269841e-04f);
    sn = fmaf(t2, sn, 8.3333333333333332e-03f);
    sn = fmaf(t2, sn, -1.6666666666666666e-01f);
    sn = fmaf(th * t2, sn, th);
    float cs = fmaf(t2, 2.0876756987868100e-09f, -2.7557319223985888e-07f);
    cs = fmaf(t2, cs, 2.4801587301587302e-05f);
    cs = fmaf(t2, cs, -1.3888888888888889e-03f);
    cs = fmaf(t2, cs, 4.1666666666666664e-02f);
    cs = fmaf(t2, cs, -0.5f);
    cs = fmaf(t2, cs, 1.0f);
    const float sg = ((int)q & 1) ? -1.0f : 1.0f;
    return mk(sg * cs, sg * sn);
}

struct RampArgs {
    const int32_t *lag;        // [nrows] this block's lag per row
    const float *frac;         // [nrows] this block's parabolic estimate
    const float *frac_override; // [nrows] caller-supplied fractional delays, or nullptr: gain * frac
    const float2 *phasor;      // [nrows] get_phasecorrect() after this block
    const uint32_t *k2tab;     // [8192]: k2 of the two complex values of refspec slot q, lo / hi 16 bits
    const float2 *wc, *wf;     // two-level table of W_B^m (longblock.hpp LongTw): coarse W_B^(i << fbits), fine W_B^j
    int fbits;
    float gain;
    int row0, log2n1;          // global row of blockIdx.y == 0
};
// The response factored for the two-line stage-B kernel (k_rows14_cf32q<true>): with k = k1 + N1 k2
//     H[k] = c(k1) . G(k2),   c = p / B . exp(2 pi i k1 (lag + D) / B),   G = exp(2 pi i k2 lag / 16384) . exp(2 pi i (k2 D / 16384 - [k2 >= 8192] D))
// G does not depend on the line: it is written once per row and block as a "spectrum" of the row in the junction's register
// order (128 KiB per row: the 128 lines of a row read it from L2 exactly as the correlation pass reads the reference
// spectrum), and c scales a line's inputs.  The per-bin work of the junction -- two gathers, a polynomial sine / cosine and
// four products, two thirds of the one-line kernel's time -- is then done once per row instead of once per line.
__device__ __forceinline__ c2 ramp_line_scale(const RampArgs &ra, int row, uint32_t k1)
{
    const int log2B = 14 + ra.log2n1;
    const uint32_t bmask = (1u << log2B) - 1u, lagu = (uint32_t)ra.lag[row];
    const float invB = 1.0f / (float)(1u << log2B);
    const float D = ra.frac_override ? ra.frac_override[row] : ra.gain * ra.frac[row];
    const float2 pr = ra.phasor[row];
    const uint32_t m1 = (k1 * lagu) & bmask;
    const float2 a = ra.wc[m1 >> ra.fbits], f = ra.wf[m1 & ((1u << ra.fbits) - 1u)];
    return cmul(cmulc(mk(pr.x * invB, pr.y * invB), cmul(mk(a.x, a.y), mk(f.x, f.y))), cis2pi((float)k1 * (D * invB)));
}
// grid (8192 / 256, rows of the launch); G + (row in launch) * 8192 float4
__global__ __launch_bounds__(256) void k_ramp_rowspec(float4 *__restrict__ G, RampArgs ra)
{
    const int idx = blockIdx.x * 256 + threadIdx.x, row = ra.row0 + (int)blockIdx.y;
    const uint32_t lagu = (uint32_t)ra.lag[row];
    const float D = ra.frac_override ? ra.frac_override[row] : ra.gain * ra.frac[row];
    const uint32_t kk = ra.k2tab[idx];
    float hx[4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const uint32_t k2 = e ? (kk >> 16) : (kk & 0xffffu);
        const uint32_t m2 = ((k2 * lagu) & 16383u) << ra.log2n1;                       // W_16384^x = W_B^(N1 x): the plan's two-level table
        const float2 a = ra.wc[m2 >> ra.fbits], f = ra.wf[m2 & ((1u << ra.fbits) - 1u)];
        const c2 h = cmulc(cis2pi((float)k2 * (D * (1.0f / 16384.0f)) - (k2 >= 8192u ? D : 0.0f)), cmul(mk(a.x, a.y), mk(f.x, f.y)));
        hx[2 * e] = h.x; hx[2 * e + 1] = h.y;
    }
    G[(size_t)blockIdx.y * 8192 + idx] = make_float4(hx[0], hx[1], hx[2], hx[3]);
}

template <bool IS_REF, bool RAMP = false>
__global__ __launch_bounds__(THREADS, 2) void k_rows14_cf32p(c2 *__restrict__ Y, const c2 *__restrict__ twA,
                                                             const c2 *__restrict__ twB, float4 *__restrict__ refspec_base, RampArgs ra = RampArgs{})
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c2 *A = reinterpret_cast<c2 *>(smem);
    float4 *A4 = reinterpret_cast<float4 *>(smem);
    const int tid = threadIdx.x;
    c2 *line = Y + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * N;
    float4 *refspec4 = refspec_base + (size_t)blockIdx.x * (N / 2);
    {
        c2 v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = line[i * 512 + tid];
        dft32<-1>(v);
        c2 w[32];
        tw_load(w, twA, TWA_STRIDE, tid);
        tw_apply<-1, true, 1>(v, w);
        const int base = p0_base(tid);
#pragma unroll
        for (int k = 0; k < 32; ++k) A[base + k * 528] = v[xpos(k)];
    }
    c2 wB[32];
    tw_load(wB, twB, TWB_STRIDE, tid & 15);
    __syncthreads();
    pass1_forward(A, wB, tid);
    wave_lds_sync();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int g = ((tid >> 6) << 7) + 64 * h + (tid & 63), base = j_base(g), key = g & 7;
        float4 r[8];
        if constexpr (!IS_REF && !RAMP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = refspec4[j * 1024 + g];
        }
        if constexpr (RAMP) {
            const int row = ra.row0 + (int)blockIdx.y, log2B = 14 + ra.log2n1;
            const uint32_t bmask = (1u << log2B) - 1u, lagu = (uint32_t)ra.lag[row];
            const float invB = 1.0f / (float)(1u << log2B);
            const float D = ra.frac_override ? ra.frac_override[row] : ra.gain * ra.frac[row];
            const float2 pr = ra.phasor[row];
            const c2 pb = mk(pr.x * invB, pr.y * invB);
            const float dstep = D * invB;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t kk = ra.k2tab[j * 1024 + g];
                float hx[4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint32_t k2 = e ? (kk >> 16) : (kk & 0xffffu);
                    const uint32_t k = (uint32_t)blockIdx.x + (k2 << ra.log2n1);
                    const int ks = (k <= (bmask >> 1)) ? (int)k : (int)k - (int)(bmask + 1u);
                    // integer part: exp(+2 pi i (k lag mod B) / B) = conj(W_B^m) from the plan's two-level table (one product, ~1 ulp);
                    // fractional part: |k_s D / B| <= 1/4 turn, by polynomial
                    const uint32_t m = (k * lagu) & bmask;
                    const float2 a = ra.wc[m >> ra.fbits], f = ra.wf[m & ((1u << ra.fbits) - 1u)];
                    const c2 wi = cmul(mk(a.x, a.y), mk(f.x, f.y));
                    const c2 h = cmul(cmulc(pb, wi), cis2pi((float)ks * dstep));
                    hx[2 * e] = h.x; hx[2 * e + 1] = h.y;
                }
                r[j] = make_float4(hx[0], hx[1], hx[2], hx[3]);
            }
        }
        c2 u[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 q = A4[base + (j ^ key)];
            u[2 * j] = mk(q.x, q.y);
            u[2 * j + 1] = mk(q.z, q.w);
        }
        dft16p<-1>(u);
        if constexpr (IS_REF) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                refspec4[j * 1024 + g] = make_float4(u[2 * j].x, -u[2 * j].y, u[2 * j + 1].x, -u[2 * j + 1].y);
        } else {
#if CRSDR_K1_FUSED_TW
            {
                c2 rr[16];
#pragma unroll
                for (int j = 0; j < 8; ++j) { rr[2 * j] = mk(r[j].x, r[j].y); rr[2 * j + 1] = mk(r[j].z, r[j].w); }
                dft16_inv_mul(u, rr);
            }
#else
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u[2 * j] = cmul(u[2 * j], mk(r[j].x, r[j].y));
                u[2 * j + 1] = cmul(u[2 * j + 1], mk(r[j].z, r[j].w));
            }
            dft16p<+1>(u);
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j)
                A4[base + (j ^ key)] = make_float4(u[2 * j].x, u[2 * j].y, u[2 * j + 1].x, u[2 * j + 1].y);
        }
    }
    if constexpr (!IS_REF) {
        wave_lds_sync();
        pass1_inverse(A, wB, tid);
        __syncthreads();
        c2 v[32];
        const int base = p0_base(tid);
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = A[base + k * 528];
        c2 w[32];
        tw_load(w, twA, TWA_STRIDE, tid);
        tw_dft32_inv(v, w);
#pragma unroll
        for (int i = 0; i < 32; ++i) line[i * 512 + tid] = v[xpos(i)]; // natural order, coalesced
    }
}


// ---- fractional-delay correction of a 16384-point block on K1's network (crsdr_plan_set_frac_apply at the reference's block size) ----
// The generic form (kernels.hpp k_frac_apply<LOG2N>: the radix-16 LDS network of fft_lds.hpp) costs 50 us per row and CU at this size;
// this is the same pass on the 32 x 32 x 16 network: K1's int8 first pass and its P1, the junction as in the long blocks' stage B''
// (DFT16, every bin times H[k] = p / B . exp(+2 pi i k_s (lag + D) / B), inverse DFT16; the frequency of every junction register from
// the plan's k2tab, the response from two 128-entry tables of the row), P1', and a last pass whose first L outputs are quantised like cdsp::convto8bit and laid out in natural order
// in the (by then free) LDS image, so that the row leaves as 16-byte stores.  The 1/127 of convtofloat and the x 127 of convto8bit
// are a common factor of the linear chain and cancel: the transforms run on integer-valued floats, as in K1.
struct FracRowArgs {
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
    const uint32_t *k2tab;     // [8192] frequency index of every junction register pair
    const float2 *tw;          // [16384] W_N^m = exp(-2 pi i m / N)
};

// grid (owned rows, blocks)
__global__ __launch_bounds__(THREADS, 2) void k_frac_apply14(FracRowArgs a, const float2 *__restrict__ twA_, const float2 *__restrict__ twB_)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c2 *A = reinterpret_cast<c2 *>(smem);
    float4 *A4 = reinterpret_cast<float4 *>(smem);
    const c2 *twA = reinterpret_cast<const c2 *>(twA_), *twB = reinterpret_cast<const c2 *>(twB_);
    const int tid = threadIdx.x, t = blockIdx.y, row = a.row_begin + (int)blockIdx.x;
    const size_t o = (size_t)t * a.nrows + row;
    // The response of the row, H[k] = p / N . exp(+2 pi i k_s (lag + D) / N), as a product of two 128-entry tables over k = 128 k_hi + k_lo,
    // built once per row behind the image (formed per bin it was 45 vector instructions + a gather, 1.7 x K1's whole row):
    //   R_hi[h] = conj(W^((128 h lag) mod N)) . cis(128 h D / N) . (h >= 64 ? cis(-D) : 1)      (k >= L: k_s = k - N)
    //   R_lo[l] = p / N . conj(W^((l lag) mod N)) . cis(l D / N)
    // integer parts reduced mod N in integers and read from the plan's forward table, fractional parts by polynomial.
    c2 *Rhi = reinterpret_cast<c2 *>(smem + (size_t)LDS_ELEMS * 8 + 512), *Rlo = Rhi + 128;
    if (tid < 256) {
        const uint32_t lagu = (uint32_t)a.lag[o];
        const float invN = 1.0f / (float)N;
        const float D = a.frac_override ? a.frac_override[row] : a.gain * a.frac[o];
        const float dstep = D * invN;
        const uint32_t idx = (uint32_t)tid & 127u, k = tid < 128 ? idx << 7 : idx;
        const float2 w = a.tw[(k * lagu) & (uint32_t)(N - 1)];
        c2 r = cmulc(cis2pi((float)k * dstep - ((tid < 128 && idx >= 64u) ? D : 0.0f)), mk(w.x, w.y));
        if (tid >= 128) {
            const float2 pr = a.phasor[o];
            r = cmul(r, mk(pr.x * invN, pr.y * invN));
        }
        (tid < 128 ? Rhi : Rlo)[idx] = r;
    }
    c2 wA[32];
    pass0_forward<false>(A, a.rows + (size_t)t * a.block_stride + (size_t)row * N, twA, a.xor80, tid, wA);
    c2 wB[32];
    tw_load(wB, twB, TWB_STRIDE, tid & 15);
    __syncthreads();
    pass1_forward(A, wB, tid);
    wave_lds_sync();
    {
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int g = ((tid >> 6) << 7) + 64 * h + (tid & 63), base = j_base(g), key = g & 7;
            c2 rr[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t kk = a.k2tab[j * 1024 + g];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint32_t k = e ? (kk >> 16) : (kk & 0xffffu);
                    rr[2 * j + e] = cmul(Rhi[k >> 7], Rlo[k & 127u]);
                }
            }
            c2 u[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 q = A4[base + (j ^ key)];
                u[2 * j] = mk(q.x, q.y);
                u[2 * j + 1] = mk(q.z, q.w);
            }
            dft16p<-1>(u);
#if CRSDR_K1_FUSED_TW
            dft16_inv_mul(u, rr);
#else
#pragma unroll
            for (int j = 0; j < 16; ++j) u[j] = cmul(u[j], rr[j]);
            dft16p<+1>(u);
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j)
                A4[base + (j ^ key)] = make_float4(u[2 * j].x, u[2 * j].y, u[2 * j + 1].x, u[2 * j + 1].y);
        }
    }
    wave_lds_sync();
    pass1_inverse(A, wB, tid);
    __syncthreads();
    uint32_t out16[16];
    {
        c2 v[32];
        const int base = p0_base(tid);
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = A[base + k * 528];
        tw_dft32_inv(v, wA);
        // cpacketize::write(complex<float>*) -> cdsp::convto8bit (src/cpacketizer.cc:158-172, src/cdsp.cc:51-54) on the first L samples:
        // outputs i * 512 + tid, i < 16
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const c2 x = v[xpos(i)];
            const int qr = (int)rintf(fminf(fmaxf(x.x, -128.0f), 127.0f)), qi = (int)rintf(fminf(fmaxf(x.y, -128.0f), 127.0f));
            out16[i] = ((uint32_t)qr & 0xffu) | (((uint32_t)qi & 0xffu) << 8);
        }
    }
    __syncthreads();                                             // every thread has taken its part of the image: it becomes the output row
    uint16_t *ob = reinterpret_cast<uint16_t *>(smem);
#pragma unroll
    for (int i = 0; i < 16; ++i) ob[i * 512 + tid] = (uint16_t)out16[i];
    __syncthreads();
    int8_t *orow = a.slab ? a.slab + (size_t)t * a.slab_stride + (size_t)(row - a.row_begin) * N
                          : a.packet + (size_t)t * a.packet_stride + 16 + 4 * (size_t)a.nrows + (size_t)row * N;
    if (((uintptr_t)orow & 15) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(smem);
        uint4 *d4 = reinterpret_cast<uint4 *>(orow);
        d4[tid] = s4[tid];
        d4[tid + THREADS] = s4[tid + THREADS];
    } else {                                                     // a bound packet whose matrix is only 4-byte aligned
        const uint32_t *s1 = reinterpret_cast<const uint32_t *>(smem);
        uint32_t *d1 = reinterpret_cast<uint32_t *>(orow);
#pragma unroll
        for (int q = 0; q < 8; ++q) d1[tid + q * THREADS] = s1[tid + q * THREADS];
    }
}

// grid: fold * blocks + owned rows * blocks workgroups, one-dimensional.  fold = 1: the first `blocks` workgroups are the blocks'
// reference items -- every one of them has a lower index than any row workgroup, so in-order dispatch has them all resident (or done)
// before a row can wait for one, and only the first rows of a launch ever do (with the reference item NEXT to its block's rows,
// index t * (rows + 1), every block's rows were dispatched beside their own reference item and stood still for its ~9 us: measured
// r03, K1 0.117 -> 0.274 ms per 2560-row launch).
__global__ __launch_bounds__(THREADS, 2) void k_xcorr_lag14p(XcorrArgs a, const float2 *__restrict__ twA,
                                                             const float2 *__restrict__ twB, int row_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nref = a.fold ? a.nblocks : 0;
    const int id = (int)blockIdx.x - nref;
    const int t = id < 0 ? (int)blockIdx.x : id / row_count, x = id < 0 ? -1 : id % row_count;
    if (x < 0) {
        ref_spectrum_row14<true>(smem, a.rows + (size_t)t * a.block_stride, reinterpret_cast<const c2 *>(twA), reinterpret_cast<const c2 *>(twB),
                                 reinterpret_cast<float4 *>(a.refspec_w) + (size_t)t * (N / 2), a.xor80);
        // write-through stores: once every wave's have been acknowledged (vmcnt(0)) and all waves have said so (barrier), the word may
        // follow (r03 first cut: plain stores + a release fence per wave, i.e. a write-back of the XCD's whole L2 -- the previous
        // batch's freshly written packet rows -- in front of every reference item's flag)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.refflag + t, a.refgen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int row = a.row_begin + x;
    if (xcorr_skip(a, row, t, threadIdx.x)) return;
    xcorr_row14(a, smem, a.rows + (size_t)t * a.block_stride + (size_t)row * N, row, t,
                reinterpret_cast<const c2 *>(twA), reinterpret_cast<const c2 *>(twB));
}

} // namespace x14p
} // namespace crsdr
