// k1_phases.hip -- diagnostic build of K1 (B = 16384) with s_memtime stamps at phase boundaries.
// Never part of the product: prints the share of a row's time each phase takes (median over waves).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I coherent-rtlsdr_amd/csrc -o tools/k1_phases tools/k1_phases.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
__device__ unsigned long long *g_stamps; // [nwg][8 waves][NST]
constexpr int NST = 9;
#define CRSDR_STAMP(i)                                                                                   \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        unsigned long long t_;                                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if ((threadIdx.x & 63) == 0)                                                                     \
            g_stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * NST + (i)] = t_; \
    } while (0)
#include "xcorr14p.hpp"
using namespace crsdr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
    const int rows = 1025, T = 2, N = 16384;
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
    int32_t *lag, *lag_s; float *mag, *frac, *mag_s, *frac_s;
    CK(hipMalloc(&lag, T * rows * 4)); CK(hipMalloc(&mag, T * rows * 4)); CK(hipMalloc(&frac, T * rows * 4));
    CK(hipMalloc(&lag_s, rows * 4)); CK(hipMalloc(&mag_s, rows * 4)); CK(hipMalloc(&frac_s, rows * 4));
    const int nwg = (rows - 1) * T;
    unsigned long long *stamps; CK(hipMalloc(&stamps, (size_t)nwg * 8 * NST * 8)); CK(hipMemset(stamps, 0, (size_t)nwg * 8 * NST * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
    CK(hipFuncSetAttribute((const void *)x14::k_ref_spectrum14, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)x14p::k_xcorr_lag14p, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES));
    hipLaunchKernelGGL(x14::k_ref_spectrum14, dim3(T), dim3(512), x14::LDS_BYTES, 0, d_rows, (size_t)rows * N, twA, twB, (float4 *)refspec, 0u);
    XcorrArgs xa{};
    xa.rows = d_rows; xa.block_stride = (size_t)rows * N; xa.refspec = refspec; xa.lag_mask = nullptr; xa.row_begin = 1; xa.nrows = rows;
    xa.nblocks = T; xa.xor80 = 0; xa.lag = lag; xa.mag = mag; xa.frac = frac; xa.lag_state = lag_s; xa.mag_state = mag_s; xa.frac_state = frac_s;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(x14p::k_xcorr_lag14p, dim3(rows - 1, T), dim3(512), x14::LDS_BYTES, 0, xa, twA, twB);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nwg * 8 * NST);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    const char *names[NST - 1] = {"P0 (HBM int8 -> DFT32 -> tw -> LDS)", "barrier 1", "P1 fwd", "J (DFT16 x ref x IDFT16)", "P1 inv", "barrier 2", "P0 inv + |.|^2 + argmax", "reduce + publish"};
    double tot = 0; std::vector<double> med(NST - 1);
    for (int s = 0; s < NST - 1; ++s) {
        std::vector<double> v;
        for (size_t w = 0; w < (size_t)nwg * 8; ++w) v.push_back((double)(st[w * NST + s + 1] - st[w * NST + s]));
        std::sort(v.begin(), v.end()); med[s] = v[v.size() / 2]; tot += med[s];
    }
    for (int s = 0; s < NST - 1; ++s) printf("%-40s median %8.0f cycles  %5.1f %%\n", names[s], med[s], 100.0 * med[s] / tot);
    printf("sum of medians %.0f cycles per row (stamped build; shares, not absolute time)\n", tot);
    return 0;
}
