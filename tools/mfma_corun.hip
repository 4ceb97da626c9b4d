// mfma_corun.hip -- diagnostic for the round-2 review's last question: "is the idle matrix pipe worth anything to K1?"
// K1 (csrc/xcorr14q.hpp) is a stream of PACKED fp32 vector instructions (v_pk_fma / v_pk_mul / v_pk_add: 3 156 of the 4 260 static
// VALU instructions of a row pair) on two waves per SIMD; its junction (DFT16 . conj(ref) . IDFT16, ~28 % of the butterflies)
// could be two 32 x 32 real matrix products on v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles per instruction and SIMD).  Would
// such a matrix stream run BESIDE the packed stream of the other wave of its SIMD, or out of it?  Measured here, no kernel built:
//   mode 0  two waves per SIMD, both the packed stream                       (K1 today)
//   mode 1  one wave per SIMD, the packed stream                              (a wave alone)
//   mode 2  one wave per SIMD, the matrix stream                              (the junction's MFMAs alone)
//   mode 3  two waves per SIMD: one packed, one matrix                        (the question)
//   mode 4  two waves per SIMD, each interleaving one MFMA per G packed instructions (one wave doing both)
//   mode 5  as 3 with the vector wave on SCALAR fp32 (v_fma_f32 / v_add_f32 / v_mul_f32: two per packed instruction)
// Every wave stamps its own duration (s_memtime); one workgroup per CU (LDS forces it), all CUs busy.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_corun tools/mfma_corun.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define REP4(X) X X X X
// 32 packed instructions on 8 independent register pairs: fma, mul, add, fma per pair (K1's mix is ~45 % fma, 25 % mul, 30 % add)
#define PK32(p0, p1, p2, p3, p4, p5, p6, p7, pa, pb)                                                                                   \
    asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
                 "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n" \
                 "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"                 \
                 "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"                 \
                 "v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %9\n v_pk_add_f32 %3, %3, %9\n"                 \
                 "v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9\n"                 \
                 "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
                 "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n" \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pa))
// the same arithmetic volume as 64 scalar instructions on 16 registers
#define SC64(s, b, a)                                                                                                                   \
    asm volatile("v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"   \
                 "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"   \
                 "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n" \
                 "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n" \
                 "v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n"                         \
                 "v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n"                         \
                 "v_mul_f32 %8, %8, %16\n v_mul_f32 %9, %9, %16\n v_mul_f32 %10, %10, %16\n v_mul_f32 %11, %11, %16\n"                     \
                 "v_mul_f32 %12, %12, %16\n v_mul_f32 %13, %13, %16\n v_mul_f32 %14, %14, %16\n v_mul_f32 %15, %15, %16\n"                 \
                 "v_add_f32 %0, %0, %17\n v_add_f32 %1, %1, %17\n v_add_f32 %2, %2, %17\n v_add_f32 %3, %3, %17\n"                         \
                 "v_add_f32 %4, %4, %17\n v_add_f32 %5, %5, %17\n v_add_f32 %6, %6, %17\n v_add_f32 %7, %7, %17\n"                         \
                 "v_add_f32 %8, %8, %17\n v_add_f32 %9, %9, %17\n v_add_f32 %10, %10, %17\n v_add_f32 %11, %11, %17\n"                     \
                 "v_add_f32 %12, %12, %17\n v_add_f32 %13, %13, %17\n v_add_f32 %14, %14, %17\n v_add_f32 %15, %15, %17\n"                 \
                 "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"   \
                 "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"   \
                 "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n" \
                 "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n" \
                 : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(s[8]), "+v"(s[9]),  \
                   "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15]) : "v"(b), "v"(a))

// role: 0 nothing, 1 packed stream (`iters` x 32 instructions), 2 matrix stream (`iters` x 2 MFMA), 3 both interleaved (one wave:
// 32 packed + G-th MFMA), 4 scalar stream (`iters` x 64 instructions)
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float *out, unsigned long long *cyc, int iters, int mfma_every)
{
    extern __shared__ float lds[];       // 100 KiB: one workgroup per CU
    const int wave = threadIdx.x >> 6;    // waves 0..3 land on SIMDs 0..3, waves 4..7 on the same SIMDs again
    int role = 0;
    if (MODE == 0) role = 1;
    if (MODE == 1) role = wave < 4 ? 1 : 0;
    if (MODE == 2) role = wave < 4 ? 2 : 0;
    if (MODE == 3) role = wave < 4 ? 1 : 2;
    if (MODE == 4) role = 3;
    if (MODE == 5) role = wave < 4 ? 4 : 2;
    const float a = threadIdx.x * 1e-3f, b = 0.999f;
    v2f pa = {a, a + 1.f}, pb = {b, b};
    v2f p0 = pa, p1 = pa + 1.f, p2 = pa + 2.f, p3 = pa + 3.f, p4 = pa + 4.f, p5 = pa + 5.f, p6 = pa + 6.f, p7 = pa + 7.f;
    float s[16];
    for (int i = 0; i < 16; ++i) s[i] = a + i;
    v16f acc0 = {0}, acc1 = {0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (role == 1) {
        for (int i = 0; i < iters; ++i) { PK32(p0, p1, p2, p3, p4, p5, p6, p7, pa, pb); }
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
    } else if (role == 3) {
        for (int i = 0; i < iters; ++i) {
            PK32(p0, p1, p2, p3, p4, p5, p6, p7, pa, pb);
            if (i % mfma_every == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        }
    } else if (role == 4) {
        for (int i = 0; i < iters; ++i) { SC64(s, b, a); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = role ? t1 - t0 : 0ull;
    float r = p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + acc0[0] + acc1[3];
    for (int i = 0; i < 16; ++i) r += s[i];
    if (r == 123.456f) out[0] = r + lds[threadIdx.x];
}

template <int MODE>
static void run(const char *name, int iters, int mfma_every, int cus)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, sizeof(unsigned long long) * 8 * cus);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {                       // the last repetition is reported (clocks settled)
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(cus), dim3(512), 100 * 1024, 0, out, cyc, iters, mfma_every);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8 * cus);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double lo = 0, hi = 0; int nlo = 0, nhi = 0;
    for (int c = 0; c < cus; ++c) for (int w = 0; w < 8; ++w) { if (!h[c * 8 + w]) continue; if (w < 4) { lo += h[c * 8 + w]; ++nlo; } else { hi += h[c * 8 + w]; ++nhi; } }
    // s_memtime ticks at a fixed 100 MHz on this part: report wall time per instruction from the kernel's duration instead
    std::printf("%-62s %8.3f ms | waves 0-3: %9.0f ticks  waves 4-7: %9.0f ticks\n", name, ms, nlo ? lo / nlo : 0.0, nhi ? hi / nhi : 0.0);
    hipFree(out); hipFree(cyc);
}

int main(int argc, char **argv)
{
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    // per SIMD and kernel: mode 0: 2 x iters x 32 packed; mode 1: iters x 32 packed; mode 2: iters x 2 MFMA; mode 3: both of 1 and 2
    run<1>("1 one wave/SIMD, packed stream (iters x 32 v_pk)", iters, 1, cus);
    run<0>("0 two waves/SIMD, both packed (2 x iters x 32 v_pk)", iters, 1, cus);
    run<2>("2 one wave/SIMD, matrix stream (iters x 2 mfma_32x32x2_f32)", iters, 1, cus);
    run<3>("3 packed wave + matrix wave per SIMD", iters, 1, cus);
    run<5>("5 scalar-fp32 wave (iters x 64) + matrix wave per SIMD", iters, 1, cus);
    run<4>("4 two waves/SIMD, each 32 v_pk + 1 mfma per iteration", iters, 1, cus);
    run<4>("4 two waves/SIMD, each 32 v_pk + 1 mfma per 2 iterations", iters, 2, cus);
    run<4>("4 two waves/SIMD, each 32 v_pk + 1 mfma per 4 iterations", iters, 4, cus);
    std::printf("reading: if mode 3 ~ max(mode 1, mode 2) the pipes run side by side; if ~ mode 1 + mode 2 the matrix stream comes out of the packed stream\n");
    return 0;
}
