// Diagnostic: times k_covariance_tiled (as is, or a variant patched by gen.py) on a 1025 x 16384 matrix.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include COV_HEADER
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{
    using namespace crsdr;
    const int nsig = argc > 1 ? atoi(argv[1]) : 1024, B = argc > 2 ? atoi(argv[2]) : 16384, S = argc > 3 ? atoi(argv[3]) : 7, nrows = nsig + 1;
    const int nt = (nsig + cov::CT - 1) / cov::CT, ntri = nt * (nt + 1) / 2;
    int8_t *m; int *part; int2 *psum;
    CK(hipMalloc(&m, (size_t)nrows * B));
    std::vector<int8_t> h((size_t)nrows * B);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (int8_t)(rand() & 255);
    CK(hipMemcpy(m, h.data(), h.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&part, sizeof(int) * 3 * cov::CT * cov::CT * (size_t)ntri * S));
    CK(hipMalloc(&psum, sizeof(int2) * nt * cov::CT * S));
    CK(hipFuncSetAttribute((const void *)cov::k_covariance_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, cov::COV_LDS_BYTES));
    const unsigned grid = 8u * ((ntri * S + 7) / 8);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        const int n = 200;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < n; ++i)
            hipLaunchKernelGGL(cov::k_covariance_tiled, dim3(grid), dim3(cov::COV_THREADS), cov::COV_LDS_BYTES, 0, m, nrows, B, nt, ntri, S, part, psum);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("%-28s nsig %d B %d S %d grid %u: %.2f us per launch (back to back)\n", COV_NAME, nsig, B, S, grid, 1e3 * ms / n);
    }
    return 0;
}
