// k1_corun.hip -- diagnostic: how much VALU issue slack do the SIMDs have while K1 (B = 16384) runs?
// Launches K1 on one stream and a register-light, LDS-free VALU spinner (W waves per SIMD) on another and
// compares each one's time alone and together.  Never part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I coherent-rtlsdr_amd/csrc -o tools/k1_corun tools/k1_corun.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "xcorr14.hpp"
using namespace crsdr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X

// 64-thread workgroups, 8 independent chains, half v_add half v_fmac (the K1 mix), 128 VALU per iteration
__global__ __launch_bounds__(64) void k_spin(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 0.999f;
    float s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3, s4 = a + 4, s5 = a + 5, s6 = a + 6, s7 = a + 7;
    for (int i = 0; i < iters; ++i) {
        REP16(asm volatile("v_add_f32 %0, %0, %8\n v_fmac_f32 %1, %8, %9\n v_add_f32 %2, %2, %8\n v_fmac_f32 %3, %8, %9\n"
                           "v_add_f32 %4, %4, %8\n v_fmac_f32 %5, %8, %9\n v_add_f32 %6, %6, %8\n v_fmac_f32 %7, %8, %9\n"
                           : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(b), "v"(a));)
    }
    float r = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    if (r == 123.456f) out[0] = r;
}

int main(int argc, char **argv)
{
    const int rows = 1025, T = 8, N = 16384;
    const int W = argc > 1 ? atoi(argv[1]) : 1;           // spinner waves per SIMD
    const int iters = argc > 2 ? atoi(argv[2]) : 1500;
    std::vector<int8_t> h((size_t)T * rows * N);
    srand(1);
    for (auto &v : h) v = (int8_t)((rand() % 120) - 60);
    int8_t *d_rows; CK(hipMalloc(&d_rows, h.size())); CK(hipMemcpy(d_rows, h.data(), h.size(), hipMemcpyHostToDevice));
    std::vector<float2> a(5 * 512), b(5 * 16);
    for (int j = 0; j < 5; ++j) {
        for (int t = 0; t < 512; ++t) { double ang = 2.0 * M_PI * (double)(t << j) / 16384.0; a[j * 512 + t] = make_float2((float)cos(ang), (float)-sin(ang)); }
        for (int n = 0; n < 16; ++n) { double ang = 2.0 * M_PI * (double)(n << j) / 512.0; b[j * 16 + n] = make_float2((float)cos(ang), (float)-sin(ang)); }
    }
    float2 *twA, *twB, *refspec; CK(hipMalloc(&twA, a.size() * 8)); CK(hipMalloc(&twB, b.size() * 8)); CK(hipMalloc(&refspec, (size_t)T * N * 8));
    CK(hipMemcpy(twA, a.data(), a.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(twB, b.data(), b.size() * 8, hipMemcpyHostToDevice));
    int32_t *lag, *lag_s; float *mag, *frac, *mag_s, *frac_s, *d_out;
    CK(hipMalloc(&lag, T * rows * 4)); CK(hipMalloc(&mag, T * rows * 4)); CK(hipMalloc(&frac, T * rows * 4));
    CK(hipMalloc(&lag_s, rows * 4)); CK(hipMalloc(&mag_s, rows * 4)); CK(hipMalloc(&frac_s, rows * 4)); CK(hipMalloc(&d_out, 4));
    CK(hipFuncSetAttribute((const void *)x14::k_ref_spectrum14, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)x14::k_xcorr_lag14, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES));
    hipLaunchKernelGGL(x14::k_ref_spectrum14, dim3(T), dim3(512), x14::LDS_BYTES, 0, d_rows, (size_t)rows * N, twA, twB, (float4 *)refspec, 0u);
    XcorrArgs xa{};
    xa.rows = d_rows; xa.block_stride = (size_t)rows * N; xa.refspec = refspec; xa.lag_mask = nullptr; xa.row_begin = 1; xa.nrows = rows;
    xa.nblocks = T; xa.xor80 = 0; xa.stagger = 1; xa.lag = lag; xa.mag = mag; xa.frac = frac; xa.lag_state = lag_s; xa.mag_state = mag_s; xa.frac_state = frac_s;
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t ev[4]; for (auto &x : ev) CK(hipEventCreate(&x));
    auto k1 = [&]() { hipLaunchKernelGGL(x14::k_xcorr_lag14, dim3(rows - 1, T), dim3(512), x14::LDS_BYTES, s1, xa, twA, twB); };
    auto spin = [&]() { hipLaunchKernelGGL(k_spin, dim3(1024 * W), dim3(64), 0, s2, d_out, iters); };
    k1(); spin(); CK(hipDeviceSynchronize());
    float tk = 0, ts = 0, tk2 = 0, ts2 = 0;
    CK(hipEventRecord(ev[0], s1)); k1(); CK(hipEventRecord(ev[1], s1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tk, ev[0], ev[1]));
    CK(hipEventRecord(ev[2], s2)); spin(); CK(hipEventRecord(ev[3], s2)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ts, ev[2], ev[3]));
    // together: the spinner first so that its waves are resident when K1's workgroups arrive
    CK(hipEventRecord(ev[2], s2)); spin(); CK(hipEventRecord(ev[3], s2));
    CK(hipEventRecord(ev[0], s1)); k1(); CK(hipEventRecord(ev[1], s1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&tk2, ev[0], ev[1])); CK(hipEventElapsedTime(&ts2, ev[2], ev[3]));
    const double spin_instr = (double)iters * 128 * W;      // per SIMD
    const double k1_instr = 7242.0 * 8 * (rows - 1) * T / 256.0 / 8; // per SIMD: 7242 per row-SIMD, rows*T/256 rows per CU
    printf("W=%d iters=%d\n alone:    K1 %.3f ms (%.2f ns/VALU/SIMD)   spinner %.3f ms (%.2f ns/VALU/SIMD)\n", W, iters, tk, tk * 1e6 / k1_instr, ts, ts * 1e6 / spin_instr);
    printf(" together: K1 %.3f ms (x%.2f)   spinner %.3f ms (x%.2f)\n", tk2, tk2 / tk, ts2, ts2 / ts);
    const double tall = std::max(tk2, ts2);
    printf(" combined VALU rate together: %.2f ns/VALU/SIMD over %.3f ms (sum of the two alone: %.3f ms)\n", tall * 1e6 / (spin_instr + k1_instr), tall, tk + ts);
    return 0;
}
