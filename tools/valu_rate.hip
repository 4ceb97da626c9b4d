// valu_rate.hip -- microbenchmark: issue rate of f32 VALU forms on gfx950 (decides packed vs scalar
// complex arithmetic in the FFT kernels).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 0.999f;
    v2f pa = {a, a + 1.f}, pb = {b, b};
    float s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3, s4 = a + 4, s5 = a + 5, s6 = a + 6, s7 = a + 7;
    v2f p0 = pa, p1 = pa + 1.f, p2 = pa + 2.f, p3 = pa + 3.f, p4 = pa + 4.f, p5 = pa + 5.f, p6 = pa + 6.f, p7 = pa + 7.f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { // v_fma_f32, 8 independent chains x 16
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(b), "v"(a));)
        } else if (KIND == 1) { // v_pk_fma_f32
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                               "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pa));)
        } else if (KIND == 2) { // v_add_f32
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                               : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(b));)
        } else if (KIND == 3) { // v_pk_add_f32
            REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                               "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));)
        } else if (KIND == 5) { // v_mul_f32 (VOP2)
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                               : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(b));)
        } else if (KIND == 6) { // v_fmac_f32 (VOP2: D = S0*S1 + D)
            REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                               "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                               : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(b), "v"(a));)
        } else if (KIND == 7) { // complex multiply as 2 packed ops with op_sel / neg_lo (x * w, 4 independent chains x 2 instr)
            REP16(asm volatile("v_pk_mul_f32 %4, %0, %8 op_sel_hi:[0,1]\n v_pk_fma_f32 %0, %0, %8, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n"
                               "v_pk_mul_f32 %5, %1, %8 op_sel_hi:[0,1]\n v_pk_fma_f32 %1, %1, %8, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n"
                               "v_pk_mul_f32 %6, %2, %8 op_sel_hi:[0,1]\n v_pk_fma_f32 %2, %2, %8, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n"
                               "v_pk_mul_f32 %7, %3, %8 op_sel_hi:[0,1]\n v_pk_fma_f32 %3, %3, %8, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));)
        } else if (KIND == 8) { // the same complex multiply as 4 scalar ops: 2 v_mul + 2 v_fmac per product, 4 chains
            REP16(asm volatile("v_mul_f32 %4, %0, %8\n v_mul_f32 %5, %0, %9\n v_fmac_f32 %4, %1, %9\n v_fmac_f32 %5, %1, %8\n"
                               "v_mul_f32 %6, %2, %8\n v_mul_f32 %7, %2, %9\n v_fmac_f32 %6, %3, %9\n v_fmac_f32 %7, %3, %8\n"
                               "v_mul_f32 %0, %4, %8\n v_mul_f32 %1, %4, %9\n v_fmac_f32 %0, %5, %9\n v_fmac_f32 %1, %5, %8\n"
                               "v_mul_f32 %2, %6, %8\n v_mul_f32 %3, %6, %9\n v_fmac_f32 %2, %7, %9\n v_fmac_f32 %3, %7, %8\n"
                               : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(b), "v"(a));)
        } else { // v_pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                               "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));)
        }
    }
    float r = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (r == 123.456f) out[0] = r;
}
template <int KIND> void run(const char *name, int wavesPerSimd)
{
    // kinds 7 / 8 issue 8 resp. 16 instructions per REP16 element instead of 8: scaled below
    float *d; hipMalloc(&d, 4);
    const int iters = 2000, blocks = 256 * wavesPerSimd; // 256-thread blocks = 4 waves = 1 per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(a); k<KIND><<<blocks, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)iters * (KIND == 8 ? 256 : 128) * wavesPerSimd; // wave-instructions each SIMD executes
    double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("%-14s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cyc @2.4GHz)\n", name, wavesPerSimd, ms, ns_per_instr, ns_per_instr * 2.4);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_add_f32", w); run<3>("v_pk_add_f32", w); run<4>("v_pk_mul_f32", w);
        run<5>("v_mul_f32", w); run<6>("v_fmac_f32", w); run<7>("cmul 2x pk", w); run<8>("cmul 4x scalar", w);
    }
    return 0;
}
