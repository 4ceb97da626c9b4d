// cpk.hpp -- complex fp32 arithmetic on packed register pairs (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32).
//
// Why: a wave of K1 can issue one VALU instruction every ~5 cycles whatever the other wave of its SIMD does
// (tools/valu_rate.hip: 2.1 ns per instruction for a lone wave, scalar or packed), and K1's row time is its
// per-wave instruction stream plus its LDS phases.  A packed instruction does two flops per lane for the same
// issue slot, so the complex add is ONE instruction instead of two and the complex multiply TWO instead of four.
// The compiler's own SLP packing loses that to ~900 v_mov per row (dead end 4 in DESIGN.md); here the data
// simply lives as (re, im) pairs from the LDS / HBM load to the store, and the operand modifiers of VOP3P
// (op_sel / op_sel_hi pick the half each lane reads, neg_lo / neg_hi negate per lane) do the swaps and signs
// that a complex product or a multiplication by +-i needs -- clang folds whole-vector shuffles and negations
// into them, per-lane negations it does not, hence the few inline-asm primitives.
// Every primitive rounds exactly like its scalar counterpart in fft_lds.hpp (checked bitwise, tools/pk_check).
#pragma once
#include <hip/hip_runtime.h>

namespace crsdr {

typedef float c2 __attribute__((ext_vector_type(2)));     // (re, im) in an aligned VGPR pair

__device__ __forceinline__ c2 mk(float x, float y) { return c2{x, y}; }
__device__ __forceinline__ c2 cadd(c2 a, c2 b) { return a + b; }
__device__ __forceinline__ c2 csub(c2 a, c2 b) { return a - b; }

// a * b:  (fma(a.x, b.x, -(a.y b.y)), fma(a.x, b.y, a.y b.x))
__device__ __forceinline__ c2 cmul(c2 a, c2 b)
{
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// a * conj(b):  (fma(a.x, b.x, a.y b.y), fma(a.y, b.x, -(a.x b.y)))
__device__ __forceinline__ c2 cmulc(c2 a, c2 b)
{
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
template <int DIR>
__device__ __forceinline__ c2 ctw(c2 a, c2 w) { return DIR < 0 ? cmul(a, w) : cmulc(a, w); }

// c + a * conj(b) as two fused multiply-adds:  (fma(a.y, b.y, fma(a.x, b.x, c.x)),  fma(-a.x, b.y, fma(a.y, b.x, c.y)))
__device__ __forceinline__ c2 cmulc_add(c2 c, c2 a, c2 b)
{
    c2 r1, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r1) : "v"(a), "v"(b), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(r1));
    return r;
}
// c + a * b:  (fma(-a.y, b.y, fma(a.x, b.x, c.x)),  fma(a.x, b.y, fma(a.y, b.x, c.y)))
__device__ __forceinline__ c2 cmul_add(c2 c, c2 a, c2 b)
{
    c2 r1, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r1) : "v"(a), "v"(b), "v"(c));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(r1));
    return r;
}
// 2 p - s in one instruction: the difference of a radix-2 butterfly whose sum s = p + q is already there (q = s - p)
__device__ __forceinline__ c2 twice_minus(c2 p, c2 s)
{
    c2 r;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(p), "v"(s));
    return r;
}

// a + (-i) x = (a.x + x.y, a.y - x.x)      a - (-i) x = (a.x - x.y, a.y + x.x)
__device__ __forceinline__ c2 add_mi(c2 a, c2 x)
{
    c2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(x));
    return r;
}
__device__ __forceinline__ c2 sub_mi(c2 a, c2 x)
{
    c2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(x));
    return r;
}
// a +- exp(DIR i pi/2) x  (forward: -i, backward: +i)
template <int DIR> __device__ __forceinline__ c2 add_j(c2 a, c2 x) { return DIR < 0 ? add_mi(a, x) : sub_mi(a, x); }
template <int DIR> __device__ __forceinline__ c2 sub_j(c2 a, c2 x) { return DIR < 0 ? sub_mi(a, x) : add_mi(a, x); }
// exp(DIR i pi/2) x on its own: one multiplication by (+-1, -+1) with the halves swapped (exact)
template <int DIR>
__device__ __forceinline__ c2 mul_j(c2 x)
{
    c2 r;
    if constexpr (DIR < 0) asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(x));   // (x.y, -x.x)
    else asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,1]" : "=v"(r) : "v"(x));                     // (-x.y, x.x)
    return r;
}

// a * (c -+ i s): forward (DIR < 0)  (fma(a.x, c, a.y s), fma(a.y, c, -(a.x s)));  backward the conjugate
template <int DIR>
__device__ __forceinline__ c2 rot_cs(c2 a, float c, float s)
{
    const c2 cc = c2{c, c}, ss = c2{s, s};
    c2 u, r;
    if constexpr (DIR < 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(u) : "v"(a), "s"(ss));  // (a.y s, -a.x s)
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[0,1]" : "=v"(u) : "v"(a), "s"(ss));                    // (-a.y s, a.x s)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(cc), "v"(u));
    return r;
}

constexpr float kPkSqrtHalf = 0.70710678118654752440f;
constexpr float kPkC1 = 0.92387953251128675613f; // cos(pi/8)
constexpr float kPkS1 = 0.38268343236508977173f; // sin(pi/8)

// multiply by W_16^K (forward: exp(-2 pi i K/16); backward: conjugate), K in {0,1,2,3,4,6,9}
template <int DIR, int K>
__device__ __forceinline__ c2 mul_w16(c2 a)
{
    if constexpr (K == 0) return a;
    else if constexpr (K == 4) return mul_j<DIR>(a);
    else if constexpr (K == 2) {
        // forward ((a.x + a.y) k, (a.y - a.x) k);  backward ((a.x - a.y) k, (a.x + a.y) k)
        const c2 t = DIR < 0 ? add_mi(a, a) : sub_mi(a, a);
        return t * kPkSqrtHalf;
    } else if constexpr (K == 6) {
        // forward ((a.y - a.x) k, -(a.x + a.y) k);  backward (-(a.x + a.y) k, (a.x - a.y) k)
        const c2 t = DIR < 0 ? sub_mi(a, a) : add_mi(a, a);
        return t * (-kPkSqrtHalf);
    } else {
        constexpr float c = (K == 1) ? kPkC1 : (K == 3) ? kPkS1 : (K == 9) ? -kPkC1 : 0.f;
        constexpr float s = (K == 1) ? kPkS1 : (K == 3) ? kPkC1 : (K == 9) ? -kPkS1 : 0.f;
        static_assert(K == 1 || K == 3 || K == 9, "unsupported W16 power");
        return rot_cs<DIR>(a, c, s);
    }
}

// natural-order 4-point DFT of (a,b,c,d) in place
template <int DIR>
__device__ __forceinline__ void dft4(c2 &a, c2 &b, c2 &c, c2 &d)
{
    const c2 t0 = a + c, t1 = a - c, t2 = b + d, u = b - d;
    a = t0 + t2;
    c = t0 - t2;
    b = add_j<DIR>(t1, u);
    d = sub_j<DIR>(t1, u);
}

template <int DIR>
__device__ __forceinline__ void dft16(c2 *v)
{
    // n = q + 4m, k = s + 4u:  X[s+4u] = sum_q W4^{qu} ( W16^{qs} sum_m v[q+4m] W4^{ms} )
    dft4<DIR>(v[0], v[4], v[8], v[12]);
    dft4<DIR>(v[1], v[5], v[9], v[13]);
    dft4<DIR>(v[2], v[6], v[10], v[14]);
    dft4<DIR>(v[3], v[7], v[11], v[15]);
    v[5] = mul_w16<DIR, 1>(v[5]);
    v[9] = mul_w16<DIR, 2>(v[9]);
    v[13] = mul_w16<DIR, 3>(v[13]);
    v[6] = mul_w16<DIR, 2>(v[6]);
    v[10] = mul_w16<DIR, 4>(v[10]);
    v[14] = mul_w16<DIR, 6>(v[14]);
    v[7] = mul_w16<DIR, 3>(v[7]);
    v[11] = mul_w16<DIR, 6>(v[11]);
    v[15] = mul_w16<DIR, 9>(v[15]);
    dft4<DIR>(v[0], v[1], v[2], v[3]);
    dft4<DIR>(v[4], v[5], v[6], v[7]);
    dft4<DIR>(v[8], v[9], v[10], v[11]);
    dft4<DIR>(v[12], v[13], v[14], v[15]);
    c2 t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

// Junction of the cross-correlation: u[i] * r[i] (the conjugated reference spectrum) followed by the inverse 16-point DFT,
// with the products folded into the first radix-4 stage:  t0 = a ra + c rc as p = a ra, t0 = p + c rc (two fused
// multiply-adds), t1 = a ra - c rc = 2 p - t0 -- 10 packed instructions per radix-4 butterfly head instead of 12.
__device__ __forceinline__ void dft4_inv_mul(c2 &a, c2 &b, c2 &c, c2 &d, c2 ra, c2 rb, c2 rc, c2 rd)
{
    const c2 pa = cmul(a, ra), t0 = cmul_add(pa, c, rc), t1 = twice_minus(pa, t0);
    const c2 pb = cmul(b, rb), t2 = cmul_add(pb, d, rd), u = twice_minus(pb, t2);
    a = t0 + t2;
    c = t0 - t2;
    b = add_j<+1>(t1, u);
    d = sub_j<+1>(t1, u);
}
__device__ __forceinline__ void dft16_inv_mul(c2 *v, const c2 *r)
{
    dft4_inv_mul(v[0], v[4], v[8], v[12], r[0], r[4], r[8], r[12]);
    dft4_inv_mul(v[1], v[5], v[9], v[13], r[1], r[5], r[9], r[13]);
    dft4_inv_mul(v[2], v[6], v[10], v[14], r[2], r[6], r[10], r[14]);
    dft4_inv_mul(v[3], v[7], v[11], v[15], r[3], r[7], r[11], r[15]);
    v[5] = mul_w16<+1, 1>(v[5]);
    v[9] = mul_w16<+1, 2>(v[9]);
    v[13] = mul_w16<+1, 3>(v[13]);
    v[6] = mul_w16<+1, 2>(v[6]);
    v[10] = mul_w16<+1, 4>(v[10]);
    v[14] = mul_w16<+1, 6>(v[14]);
    v[7] = mul_w16<+1, 3>(v[7]);
    v[11] = mul_w16<+1, 6>(v[11]);
    v[15] = mul_w16<+1, 9>(v[15]);
    dft4<+1>(v[0], v[1], v[2], v[3]);
    dft4<+1>(v[4], v[5], v[6], v[7]);
    dft4<+1>(v[8], v[9], v[10], v[11]);
    dft4<+1>(v[12], v[13], v[14], v[15]);
    c2 t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

} // namespace crsdr
