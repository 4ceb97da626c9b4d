// arith.hpp -- single-rounding fp32 helpers shared by every kernel (bit parity with the oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crsdr {

// ---- bit-parity arithmetic helpers (no FMA contraction: one rounding per op, oracle order) --

// cdsp::convtofloat src/cdsp.cc:41-44: (float)x * (1.0f/127.0f)
__device__ __forceinline__ float i8_to_f32(int x) { return __fmul_rn((float)x, 1.0f / 127.0f); }

// cdsp::convto8bit src/cdsp.cc:51-54: r = x*127; clamp [-128,127]; rintf (half-even); NaN -> 0
__device__ __forceinline__ int f32_to_i8(float x)
{
    float r = __fmul_rn(x, 127.0f);
    if (r > 127.0f) return 127;
    if (r < -128.0f) return -128;
    if (r != r) return 0;
    return (int)rintf(r);
}

// same result for every non-NaN input, without the compare/select chain: clamp (v_med3), round to
// nearest even (v_rndne), convert.  Used where the operand is finite by construction (K2b: int8
// samples times a finite phasor).
__device__ __forceinline__ int f32_to_i8_finite(float x)
{
    const float r = __fmul_rn(x, 127.0f);
    return (int)rintf(__builtin_amdgcn_fmed3f(r, -128.0f, 127.0f));
}

// cdsp::scalarmul src/cdsp.cc:46-49: (ar*sr - ai*si) + j(ar*si + ai*sr), each op rounded once
__device__ __forceinline__ float2 rot_rn(float2 a, float2 s)
{
    return make_float2(__fsub_rn(__fmul_rn(a.x, s.x), __fmul_rn(a.y, s.y)),
                       __fadd_rn(__fmul_rn(a.x, s.y), __fmul_rn(a.y, s.x)));
}

__device__ __forceinline__ int sext8(uint32_t w, int byte) { return (int)(int8_t)((w >> (8 * byte)) & 0xFFu); }

} // namespace crsdr
