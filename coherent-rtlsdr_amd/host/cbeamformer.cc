#include "cbeamformer.h"

#include <cstdio>
#include <cstring>

namespace cbeamformer {

int covariance(const int8_t *packet, cmatrix &Rxx, int &M)
{
    uint32_t cols, rows;
    std::memcpy(&cols, packet + 4, 4);
    std::memcpy(&rows, packet + 8, 4);
    if (cols < 2 || rows < 16) return CRSDR_EINVAL;
    M = (int)cols - 1;
    Rxx.assign((size_t)M * M, {0.f, 0.f});
    int rc = crsdr_covariance(reinterpret_cast<float *>(Rxx.data()), packet + 16 + 4 * (size_t)cols, (int)cols, 2 * (int)rows, CRSDR_MEM_HOST);
    if (rc) std::fprintf(stderr, "covariance: %s\n", crsdr_last_error());
    return rc;
}

int noisesubspace(const cmatrix &Rxx, int M, cmatrix &U, std::vector<float> *S)
{
    U.assign((size_t)M * M, {0.f, 0.f});
    if (S) S->assign(M, 0.f);
    int rc = crsdr_noisesubspace(reinterpret_cast<float *>(U.data()), S ? S->data() : nullptr,
                                 reinterpret_cast<const float *>(Rxx.data()), M, CRSDR_MEM_HOST);
    if (rc) std::fprintf(stderr, "noisesubspace: %s\n", crsdr_last_error());
    return rc;
}

int pmusic2dvec(const cmatrix &U, int M, int K, float d, int Mx, int My, int Cx, int Cy, std::vector<float> &pm)
{
    pm.assign((size_t)Cx * Cy, 0.f);
    int rc = crsdr_pmusic2d(pm.data(), reinterpret_cast<const float *>(U.data()), M, K, d, Mx, My, Cx, Cy, CRSDR_MEM_HOST);
    if (rc) std::fprintf(stderr, "pmusic2dvec: %s\n", crsdr_last_error());
    return rc;
}

} // namespace cbeamformer
