// k1_pair.hip -- diagnostic build of the two-rows-per-CU K1 (xcorr14q.hpp) beside the packed one: launch time, and how
// long workgroup 0's waves sit in group barriers / wait for the LDS image.  Never part of the product.
// Usage: k1_pair [blocks = 16] [workgroups = 256] [mask every third row = 0] [co-resident hog workgroups = 0]
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I coherent-rtlsdr_amd/csrc -o tools/k1_pair tools/k1_pair.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CRSDR_QDEBUG 1
#include "xcorr14q.hpp"
using namespace crsdr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// a stand-in for a collective's kernel: nwg small workgroups that sit on their CUs for `cycles`
__global__ __launch_bounds__(256) void k_hog(long long cycles, int *sink)
{
    const long long t0 = (long long)__builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (cycles < 0) *sink = 1;
}
int main(int argc, char **argv)
{
    const int rows = 1025, T = argc > 1 ? atoi(argv[1]) : 16, N = 16384, grid = argc > 2 ? atoi(argv[2]) : 256;
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
    unsigned long long *dbg; CK(hipMalloc(&dbg, 128)); CK(hipMemset(dbg, 0, 128));
    int *err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    unsigned int *work; CK(hipMalloc(&work, 4)); CK(hipMemset(work, 0, 4));
    CK(hipFuncSetAttribute((const void *)x14::k_ref_spectrum14, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)x14p::k_xcorr_lag14p, hipFuncAttributeMaxDynamicSharedMemorySize, x14::LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)x14p::k_xcorr_lag14q, hipFuncAttributeMaxDynamicSharedMemorySize, x14p::LDSQ_BYTES));
    hipLaunchKernelGGL(x14::k_ref_spectrum14, dim3(T), dim3(512), x14::LDS_BYTES, 0, d_rows, (size_t)rows * N, twA, twB, (float4 *)refspec, 0u);
    uint8_t *d_mask = nullptr;
    if (argc > 3 && atoi(argv[3])) {
        std::vector<uint8_t> hm(rows);
        for (int r = 0; r < rows; ++r) hm[r] = (r % 3) != 0;
        CK(hipMalloc(&d_mask, rows)); CK(hipMemcpy(d_mask, hm.data(), rows, hipMemcpyHostToDevice));
        CK(hipMemset(lag_s, 0, rows * 4)); CK(hipMemset(mag_s, 0, rows * 4)); CK(hipMemset(frac_s, 0, rows * 4));
    }
    XcorrArgs xa{};
    xa.rows = d_rows; xa.block_stride = (size_t)rows * N; xa.refspec = refspec; xa.lag_mask = d_mask; xa.row_begin = 1; xa.nrows = rows;
    xa.nblocks = T; xa.xor80 = 0; xa.lag = lag; xa.mag = mag; xa.frac = frac; xa.lag_state = lag_s; xa.mag_state = mag_s; xa.frac_state = frac_s;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(x14p::k_xcorr_lag14p, dim3(rows - 1, T), dim3(512), x14::LDS_BYTES, 0, xa, twA, twB); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("packed, one row per CU : %.3f ms (%.2f us per row and CU)\n", ms, ms * 1e3 / ((rows - 1) * T / 256.0));
    }
    std::vector<int32_t> lag_p((size_t)T * rows), lag_q((size_t)T * rows);
    CK(hipMemcpy(lag_p.data(), lag, lag_p.size() * 4, hipMemcpyDeviceToHost));
    const int hogs = argc > 4 ? atoi(argv[4]) : 0;
    hipStream_t hs; CK(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
        if (hogs) { hipLaunchKernelGGL(k_hog, dim3(hogs), dim3(256), 0, hs, 1000000LL /* ~0.45 ms */, err); }
        unsigned long long *null = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(x14p::dbg__), rep == 7 ? &dbg : &null, sizeof(dbg)));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(x14p::k_xcorr_lag14q, dim3(grid), dim3(512), x14p::LDSQ_BYTES, 0, xa, twA, twB, rows - 1, err, work, (unsigned)(rep * (rows - 1) * T), x14p::kQSpinLimit); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && rep < 7) best = std::min(best, ms);
        CK(hipStreamSynchronize(hs));
    }
    printf("pair, two rows per CU  : best of 6 %.3f ms (%.2f us per row and CU)\n", best, best * 1e3 / ((rows - 1) * T / (double)grid));
    CK(hipMemcpy(lag_q.data(), lag, lag_q.size() * 4, hipMemcpyDeviceToHost));
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    unsigned long long hd[16]; CK(hipMemcpy(hd, dbg, 128, hipMemcpyDeviceToHost));
    printf("same lags: %s, wait errors: %d\n", lag_p == lag_q ? "yes" : "NO", herr);
    printf("workgroup 0, summed over its 8 waves: total %.0f cycles, in group barriers %.0f (%.1f %%), waiting for the image %.0f (%.1f %%)\n",
           (double)hd[0], (double)hd[1], 100.0 * hd[1] / hd[0], (double)hd[2], 100.0 * hd[2] / hd[0]);
    printf("barrier sites (P0->P1, P1'->P0', max, index, edge): %.1f %.1f %.1f %.1f %.1f %% of wave time\n", 100.0 * hd[3] / hd[0], 100.0 * hd[4] / hd[0], 100.0 * hd[5] / hd[0], 100.0 * hd[6] / hd[0], 100.0 * hd[7] / hd[0]);
    return 0;
}
