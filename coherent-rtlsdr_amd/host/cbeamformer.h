// cbeamformer.h -- the first consumer of a published packet, with the function surface of the reference's
// beamformer client (beamformclient/heatmap2d2.cpp): packet -> covariance (:185-197) -> noisesubspace(Rxx, K)
// (:69-79) -> pmusic2dvec(Un, d, Mx, My, Cx, Cy) (:137-147).  Eigen types become plain row-major vectors;
// the arithmetic runs in libcrsdr.so (crsdr_covariance / crsdr_noisesubspace / crsdr_pmusic2d).
#ifndef CBEAMFORMERH
#define CBEAMFORMERH
#include <complex>
#include <cstdint>
#include <vector>

#include "../../include/crsdr.h"

typedef std::vector<std::complex<float>> cmatrix;   // row-major, dimensions carried by the caller

namespace cbeamformer {
// reference defaults: #define MX 7, MY 3 (:41-42), d = (1.225*1.24)/3 wavelengths (:198), 100 x 100 scan (:199)
constexpr int MX = 7, MY = 3, CX = 100, CY = 100;
constexpr float D = (1.225f * 1.24f) / 3.0f;

// Parses the packet header like main() (:296-300: cols = N at word 1, rows = L at word 2, data at 16 + 4*cols)
// and returns Rxx [(N-1)][(N-1)] of the signal channels (the first column is dropped, :190).
int covariance(const int8_t *packet, cmatrix &Rxx, int &M);
// U of the SVD ordered by singular value; Un = columns K .. M-1 (U.rightCols(M - K)).  S optional.
int noisesubspace(const cmatrix &Rxx, int M, cmatrix &U, std::vector<float> *S = nullptr);
// pm [Cx][Cy] row-major, not normalised
int pmusic2dvec(const cmatrix &U, int M, int K, float d, int Mx, int My, int Cx, int Cy, std::vector<float> &pm);
}
#endif
