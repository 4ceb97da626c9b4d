// pk_check.hip -- bitwise check of the packed complex primitives of coherent-rtlsdr_amd/csrc/cpk.hpp against their scalar
// formulations (fft_lds.hpp).  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/pk_check tools/pk_check.hip
#include <hip/hip_runtime.h>
typedef float c2 __attribute__((ext_vector_type(2)));
// a * b, rounding like fma(a.x,b.x,-(a.y*b.y)), fma(a.x,b.y,a.y*b.x)
__device__ __forceinline__ c2 cmul(c2 a, c2 b)
{
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// a * conj(b): fma(a.x,b.x,a.y*b.y), fma(a.y,b.x,-(a.x*b.y))
__device__ __forceinline__ c2 cmulc(c2 a, c2 b)
{
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// a + (-i) x = (a.x + x.y, a.y - x.x);  a - (-i) x = (a.x - x.y, a.y + x.x)
__device__ __forceinline__ c2 add_mj(c2 a, c2 x) { c2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(x)); return r; }
__device__ __forceinline__ c2 sub_mj(c2 a, c2 x) { c2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(x)); return r; }
__global__ void k(c2 *p, const c2 *q, float *chk)
{
    int i = threadIdx.x;
    c2 a = p[i], b = q[i];
    c2 m = cmul(a, b), n = cmulc(a, b);
    c2 u = add_mj(a, b), w = sub_mj(a, b);
    p[i] = m; p[i + 64] = n; p[i + 128] = u; p[i + 192] = w;
    // scalar references
    chk[8 * i + 0] = fmaf(a.x, b.x, -(a.y * b.y)); chk[8 * i + 1] = fmaf(a.x, b.y, a.y * b.x);
    chk[8 * i + 2] = fmaf(a.x, b.x, a.y * b.y);    chk[8 * i + 3] = fmaf(a.y, b.x, -(a.x * b.y));
    chk[8 * i + 4] = a.x + b.y; chk[8 * i + 5] = a.y - b.x; chk[8 * i + 6] = a.x - b.y; chk[8 * i + 7] = a.y + b.x;
}
int main()
{
    c2 *p, *q; float *c; hipMalloc(&p, 256 * 8); hipMalloc(&q, 64 * 8); hipMalloc(&c, 64 * 8 * 4);
    c2 hp[256], hq[64]; for (int i = 0; i < 64; ++i) { hp[i] = c2{1.1f * i + 0.3f, -0.7f * i + 2.f}; hq[i] = c2{0.37f * i - 5.f, 0.11f * i * i + 1.f}; }
    hipMemcpy(p, hp, 64 * 8, hipMemcpyHostToDevice); hipMemcpy(q, hq, 64 * 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(p, q, c); hipMemcpy(hp, p, 256 * 8, hipMemcpyDeviceToHost);
    float hc[512]; hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        bad += hp[i].x != hc[8 * i] || hp[i].y != hc[8 * i + 1];
        bad += hp[i + 64].x != hc[8 * i + 2] || hp[i + 64].y != hc[8 * i + 3];
        bad += hp[i + 128].x != hc[8 * i + 4] || hp[i + 128].y != hc[8 * i + 5];
        bad += hp[i + 192].x != hc[8 * i + 6] || hp[i + 192].y != hc[8 * i + 7];
    }
    printf("mismatches %d\n", bad);
    return bad != 0;
}
